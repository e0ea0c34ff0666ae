#!/usr/bin/env python3
"""Aligned vs odd image widths (W % 4 != 0: rows are not 16-byte aligned) for the main operators, ~400 MB batches."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402
from tools.perf_configs import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for w in (1024, 1023, 1022, 500, 333):
    h = 768 if w > 600 else 375
    n = max(1, int(100e6 / (3 * h * w)))
    x = torch.rand((n, 3, h, w), generator=g, device="cuda")
    xu = (x * 255).to(torch.uint8)
    line = f"{n}x3x{h}x{w}:"
    for name, fn, nbytes in (("blur3 f32", lambda: F.gaussian_blur(x, [3, 3]), x.numel() * 8), ("blur5 f32", lambda: F.gaussian_blur(x, [5, 5]), x.numel() * 8),
                             ("blur7 f32", lambda: F.gaussian_blur(x, [7, 7]), x.numel() * 8), ("sharp f32", lambda: F.adjust_sharpness(x, 1.5), x.numel() * 8),
                             ("blur3 u8", lambda: F.gaussian_blur(xu, [3, 3]), x.numel() * 2), ("sharp u8", lambda: F.adjust_sharpness(xu, 1.5), x.numel() * 2),
                             ("blur5 u8", lambda: F.gaussian_blur(xu, [5, 5]), x.numel() * 2), ("blur7 u8", lambda: F.gaussian_blur(xu, [7, 7]), x.numel() * 2)):
        ms, _ = timeit(fn, 5)
        line += f"  {name} {nbytes / ms / 1e6:6.0f} GB/s"
    print(line, flush=True)
