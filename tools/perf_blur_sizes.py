#!/usr/bin/env python3
"""gaussian_blur on 32 x 4K frames, float32 and uint8, kernel sizes 3 .. 23: time, HBM rate and the kernel that ran (GPU box)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402
from tools.perf_vgg import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
xf = torch.rand((32, 3, 2160, 3840), generator=g, device="cuda")
xu = torch.randint(0, 256, (32, 3, 2160, 3840), generator=g, device="cuda", dtype=torch.uint8)
for k in (3, 5, 7, 9, 11, 13, 15, 23):
    for x, name in ((xf, "f32"), (xu, "u8 ")):
        ms = timeit(lambda: F.gaussian_blur(x, [k, k]), 5)
        nbytes = 2 * x.numel() * x.element_size()
        print(f"{k:2d} x {k:2d} {name}: {ms:7.3f} ms  {nbytes / ms / 1e6:6.0f} GB/s  {_lib.last_kernel()}", flush=True)
