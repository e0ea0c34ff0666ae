#!/usr/bin/env python3
"""Target program for rocprofv3 passes on one generic convolution (mv_conv2d_bias_act_f32, implicit GEMM):
    python3 tools/run_conv.py [--layer alexnet2|alexnet1|stem7] [--batch 256] [--iters 10]"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--layer", default="alexnet2")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
shapes = {"alexnet1": ((3, 224), (64, 3, 11, 11), 4, 2), "alexnet2": ((64, 27), (192, 64, 5, 5), 1, 2), "stem7": ((3, 224), (64, 3, 7, 7), 2, 3)}
(cin, side), ws, st, pd = shapes[a.layer]
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((a.batch, cin, side, side), generator=g, device="cuda")
w = torch.randn(ws, generator=g, device="cuda") * 0.05
b = torch.rand(ws[0], generator=g, device="cuda")
for _ in range(a.iters):
    y = F.conv2d_bias_act(x, w, b, stride=st, padding=pd, activation="relu")
torch.cuda.synchronize()
print(a.layer, a.batch, float(y.sum()))
