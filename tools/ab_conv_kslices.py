#!/usr/bin/env python3
"""VGG's 3x3 layers at small batch under the tuning library's MV_CONV_KSLICES (K slices across workgroups, GPU box):
time per slice count next to the library's own choice."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402
from tools.perf_vgg import timeit  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator(device="cuda").manual_seed(0)
for cin, cout, hw in ((64, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28), (512, 512, 28), (512, 512, 14)):
    x = torch.rand((batch, cin, hw, hw), generator=g, device="cuda")
    w = torch.randn((cout, cin, 3, 3), generator=g, device="cuda") * 0.02
    b = torch.rand(cout, generator=g, device="cuda")

    def five():
        for _ in range(5):
            F.conv2d_bias_relu(x, w, b)
    own = timeit(five, 7) / 5
    line = f"batch {batch} {cin:4d}->{cout:4d} @{hw:3d}: library {F.conv3x3_k_slices(batch, cin, hw, hw, cout)[0]} slices {own * 1e3:6.1f} us |"
    with _lib.tuning_library():
        for sl in (1, 2, 4, 8, 16):
            os.environ["MV_CONV_KSLICES"] = str(sl)
            try:
                line += f" ks{sl} {timeit(five, 7) / 5 * 1e3:6.1f}"
            finally:
                os.environ.pop("MV_CONV_KSLICES")
    print(line, flush=True)
