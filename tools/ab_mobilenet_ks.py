#!/usr/bin/env python3
"""MobileNetV2 forward as a replayed HIP graph (kernel time, no per-launch host work), K slices inside the pointwise
workgroups on (the library's choice) against off (MV_PW_KS=1): runs on the tuning library, interleaved rounds."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("MI355VISION_LIB", str(ROOT / "cpu-vision_amd" / "lib" / "libmi355vision_tuning.so"))
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cpu_vision_amd import graphs  # noqa: E402
from cpu_vision_amd.mobilenet import MobileNetV2  # noqa: E402

torch.manual_seed(0)
model = MobileNetV2(1000).cuda().eval()
for b in (1, 8, 64):
    x = torch.rand((b, 3, 224, 224), device="cuda")
    caps = {}
    for name, env in (("one chain", "1"), ("k slices", None)):
        if env is None:
            os.environ.pop("MV_PW_KS", None)
        else:
            os.environ["MV_PW_KS"] = env
        caps[name] = graphs.capture(model, x)  # the knob is read at capture time (launch parameters are baked into the graph)
    os.environ.pop("MV_PW_KS", None)
    a, c = caps["one chain"](x).clone(), caps["k slices"](x).clone()
    rel = float((a - c).abs().max() / a.abs().max())
    res = {k: [] for k in caps}
    for r in range(9):
        for k, cap in caps.items():
            cap(x)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                cap(x)
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t) / 20 * 1e3)
    line = f"mobilenet_v2 batch {b:3d}:"
    for k, v in res.items():
        v.sort()
        line += f"  {k} {v[len(v) // 2]:7.3f} ms ({b / v[len(v) // 2] * 1e3:7.0f} img/s)"
    print(line + f"   max |diff| / max |logit| = {rel:.1e}", flush=True)
