#!/usr/bin/env python3
"""Exact uint8 Gaussian blur on batches of NARROW images (224 x 224 crops, 500 x 375 photos): the plain 2-D pass against pair + tie
check + fix-up (round 3: the tie instantiation also exists for images that put several strips into one wave), same bits.  GPU box."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402
from tools.perf_invres import graph_time  # noqa: E402

for shape in ((1024, 3, 224, 224), (256, 3, 375, 500)):
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda")
    for k, s in ((5, 1.1), (7, 1.4), (9, 1.7), (23, 3.8)):
        line, ref = f"{shape[0]} x 3 x {shape[2]} x {shape[3]} uint8, {k:2d} x {k:2d}:", None
        for fast in (False, True):
            F.INTEGER_BLUR_EXACT_FAST = fast
            y = F.gaussian_blur(x, [k, k], [s, s])
            kern = _lib.last_kernel()
            if ref is None:
                ref = y
            assert torch.equal(ref, y)
            t = graph_time(lambda: F.gaussian_blur(x, [k, k], [s, s]), 5, inner=3)
            line += f"   {'pair + ties + fix-up' if fast else 'plain 2-D pass'} {t:.4f} ms ({kern})"
        F.INTEGER_BLUR_EXACT_FAST = True
        print(line, flush=True)
