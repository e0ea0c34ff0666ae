#!/usr/bin/env python3
"""Generic depthwise_conv2d (arbitrary 2-D taps) over kernel sizes on 32 x 4K fp32 frames."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402
from tools.perf_configs import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((32, 3, 2160, 3840), generator=g, device="cuda")
for ky, kx in ((3, 3), (5, 5), (7, 7), (5, 3), (7, 5), (5, 7), (9, 9), (3, 9), (11, 11), (1, 7)):
    w = torch.rand((ky, kx), generator=g, device="cuda").cpu()
    w /= w.sum()
    ms, _ = timeit(lambda: F.depthwise_conv2d(x, w, "reflect"), 5)
    fl = 2.0 * x.numel() * ky * kx
    print(f"depthwise {ky}x{kx} f32 reflect: {ms:7.3f} ms  {x.numel() * 8 / ms / 1e6:7.1f} GB/s  {fl / 2 / ms / 1e9:6.1f} Tfma/s", flush=True)
