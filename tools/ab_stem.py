#!/usr/bin/env python3
"""MobileNetV2's stem (Conv2dNormActivation 3 -> 32, 3x3 stride 2, BatchNorm fold + ReLU6, 64 x 3 x 224 x 224) under the tuning
library's MV_STEM_MCHUNK (output channels per thread), on the GPU box."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((64, 3, 224, 224), generator=g, device="cuda")
w = torch.randn((32, 3, 3, 3), generator=g, device="cuda") * 0.2
al, be = torch.rand(32, generator=g, device="cuda") + 0.5, torch.rand(32, generator=g, device="cuda")
fn = lambda: F.conv_norm_act(x, w, None, al, be, None, stride=2, affine="fma", activation="relu6")  # noqa: E731
nbytes = (x.numel() + 64 * 32 * 112 * 112) * 4


def timeit():
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    return sorted(ts)[len(ts) // 2]


ref = fn().clone()
print(f"product: {timeit() * 1e3:.1f} us")
with _lib.tuning_library():
    for m in (4, 8, 16, 32):
        os.environ["MV_STEM_MCHUNK"] = str(m)
        same = torch.equal(fn(), ref)
        ms = timeit()
        print(f"MV_STEM_MCHUNK={m:2d}: {ms * 1e3:6.1f} us  {nbytes / ms / 1e6:6.0f} GB/s  {'' if same else 'DIFFERS'}", flush=True)
    os.environ.pop("MV_STEM_MCHUNK")
