#!/usr/bin/env python3
"""Small-batch latency on one MI355X, eager (one Python call + launch per kernel) against a captured HIP graph."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import graphs  # noqa: E402
from cpu_vision_amd.mobilenet import MobileNetV2  # noqa: E402
from cpu_vision_amd.nn import alexnet, vgg11  # noqa: E402


def wall(fn, n=30):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


torch.manual_seed(0)
for name, model in (("mobilenet_v2", MobileNetV2(1000)), ("vgg11", vgg11(1000)), ("alexnet", alexnet(1000))):
    model = model.cuda().eval()
    for b in (1, 8):
        x = torch.rand((b, 3, 224, 224), device="cuda")
        eager = wall(lambda: model(x))
        cap = graphs.capture(model, x)
        assert torch.equal(cap(x), model(x))
        graph = wall(lambda: cap(x))
        print(f"{name:13s} batch {b}: eager {eager:7.3f} ms   HIP graph {graph:7.3f} ms   ({b / graph * 1e3:8.0f} img/s)", flush=True)
