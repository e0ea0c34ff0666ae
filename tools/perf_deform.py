#!/usr/bin/env python3
"""deform_conv2d forward on one MI355X: time per call for typical DCN layers (3x3, 1 offset group, mask) and the kernel that ran."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import ops  # noqa: E402
from tools.perf_vgg import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for n, c, m, hw in ((8, 256, 256, 64), (16, 128, 128, 56), (32, 64, 64, 112), (4, 512, 512, 32)):
    x = torch.rand((n, c, hw, hw), generator=g, device="cuda")
    w = torch.randn((m, c, 3, 3), generator=g, device="cuda") * 0.05
    b = torch.rand(m, generator=g, device="cuda")
    off = torch.randn((n, 18, hw, hw), generator=g, device="cuda") * 1.5
    mask = torch.rand((n, 9, hw, hw), generator=g, device="cuda")
    def five():  # back-to-back calls: the launches queue up behind each other, so the Python layer's ~50 us per call is hidden
        for _ in range(5):
            ops.deform_conv2d(x, off, w, b, padding=(1, 1), mask=mask)

    ms = timeit(five, 7) / 5
    flop = 2.0 * n * m * hw * hw * c * 9
    col = n * c * 9 * hw * hw * 4
    alg = (x.numel() + off.numel() + mask.numel() + n * m * hw * hw) * 4
    from cpu_vision_amd import _lib
    print(f"deform_conv2d {n}x{c}x{hw}x{hw} -> {m}, 3x3: {ms:7.3f} ms  {flop / ms / 1e9:6.1f} TFLOP/s  {_lib.last_kernel():28s} (columns, if "
          f"written: {col / 1e6:.0f} MB; tensors {alg / 1e6:.0f} MB)", flush=True)
