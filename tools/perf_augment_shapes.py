#!/usr/bin/env python3
"""Data-augmentation shapes: a 256 x 3 x 224 x 224 batch (ImageNet crops) and 64 x 3 x 375 x 500 photographs, uint8 and fp32,
through the transforms the reference applies to them."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402
from tools.perf_configs import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for shape in ((256, 3, 224, 224), (64, 3, 375, 500)):
    xf = torch.rand(shape, generator=g, device="cuda")
    xu = (xf * 255).to(torch.uint8)
    for name, x, bpp in (("u8 ", xu, 2), ("f32", xf, 8)):
        line = f"{shape[0]}x3x{shape[2]}x{shape[3]} {name}:"
        for op, fn in (("blur3", lambda: F.gaussian_blur(x, [3, 3])), ("blur5", lambda: F.gaussian_blur(x, [5, 5])),
                       ("blur23", lambda: F.gaussian_blur(x, [23, 23])), ("sharp", lambda: F.adjust_sharpness(x, 1.7))):
            ms, _ = timeit(fn, 7)
            line += f"  {op} {ms * 1e3:6.1f} us ({x.numel() * bpp / ms / 1e6:5.0f} GB/s)"
        print(line, flush=True)
