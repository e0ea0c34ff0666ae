#!/usr/bin/env python3
"""AlexNet's conv1 (11x11 stride 4) and conv2 (5x5) at batch 256 (models/alexnet.py:22-26): the implicit GEMM of
mv_conv2d_bias_act_f32 (no columns in HBM) against the columns form (plain im2col into a workspace + the same GEMM; tuning build,
MV_CONV_COLUMNS), interleaved in one process.     python tools/perf_alexnet_convs.py [--batch 256]"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
n = a.batch
g = torch.Generator(device="cuda").manual_seed(0)
layers = {"conv1 3->64 11x11 s4 p2 @224": ((n, 3, 224, 224), (64, 3, 11, 11), 4, 2),
          "conv2 64->192 5x5 p2 @27": ((n, 64, 27, 27), (192, 64, 5, 5), 1, 2),
          "7x7 s2 3->64 @224 (ResNet stem shape)": ((n, 3, 224, 224), (64, 3, 7, 7), 2, 3)}


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for name, (xs, ws, st, pd) in layers.items():
    x = torch.rand(xs, generator=g, device="cuda")
    w = torch.randn(ws, generator=g, device="cuda") * 0.05
    b = torch.rand(ws[0], generator=g, device="cuda")
    res = {"implicit": [], "columns": []}
    outs = {}
    for r in range(a.rounds + 2):
        t = timed(lambda: outs.__setitem__("implicit", F.conv2d_bias_act(x, w, b, stride=st, padding=pd, activation="relu")))
        k_i = _lib.last_kernel()
        with _lib.tuning_library():
            os.environ["MV_CONV_COLUMNS"] = "1"
            F.CONV2D_COLUMNS_WORKSPACE = True
            t2 = timed(lambda: outs.__setitem__("columns", F.conv2d_bias_act(x, w, b, stride=st, padding=pd, activation="relu")))
            k_c = _lib.last_kernel()
            F.CONV2D_COLUMNS_WORKSPACE = False
            os.environ.pop("MV_CONV_COLUMNS")
        if r >= 2:
            res["implicit"].append(t), res["columns"].append(t2)
    assert torch.equal(outs["implicit"], outs["columns"])
    oh = outs["implicit"].shape[-1]
    flop = 2.0 * n * ws[0] * oh * oh * ws[1] * ws[2] * ws[3]
    mi, mc = sorted(res["implicit"])[len(res["implicit"]) // 2], sorted(res["columns"])[len(res["columns"]) // 2]
    print(f"{name:40s} batch {n}: implicit {mi:7.3f} ms ({flop / mi / 1e9:5.1f} TF, {k_i})   columns {mc:7.3f} ms ({flop / mc / 1e9:5.1f} TF, "
          f"im2col + {k_c})   x{mc / mi:4.2f}  (identical bits)", flush=True)
