#!/usr/bin/env python3
"""fp32 Gaussian blur over the small kernel sizes (direct 2-D path <= 49 taps) on 32 x 4K frames."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402
from tools.perf_configs import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((32, 3, 2160, 3840), generator=g, device="cuda")
for ks in ([3, 3], [5, 5], [7, 7], [3, 5], [5, 3], [7, 3], [3, 7], [5, 7], [9, 5], [1, 9]):
    ms, _ = timeit(lambda: F.gaussian_blur(x, ks), 7)
    print(f"gaussian_blur {ks[0]}x{ks[1]} f32: {ms:7.3f} ms  {x.numel() * 8 / ms / 1e6:7.1f} GB/s ({x.numel() * 8 / ms / 1e6 / 80:4.1f}% HBM)", flush=True)
