#!/usr/bin/env python3
"""One pointwise layer (Conv2dNormActivation 1x1, BatchNorm fold + ReLU6, batch 64) under the tuning library's tile knobs
(runs on the GPU box):  python tools/ab_pointwise.py 16,96,112 [24,144,56 ...]   -> time and HBM rate per (MV_PW_NT, MV_PW_MW)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402

shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(16, 96, 112)]
g = torch.Generator(device="cuda").manual_seed(0)
for cin, cout, hw in shapes:
    n = 64
    x = torch.rand((n, cin, hw, hw), generator=g, device="cuda")
    w = torch.randn((cout, cin, 1, 1), generator=g, device="cuda") * 0.1
    al, be = torch.rand(cout, generator=g, device="cuda") + 0.5, torch.rand(cout, generator=g, device="cuda")
    nbytes = (x.numel() + n * cout * hw * hw) * 4
    flop = 2.0 * n * cout * cin * hw * hw
    fn = lambda: F.conv_norm_act(x, w, None, al, be, None, affine="fma", activation="relu6")  # noqa: E731
    ref = fn().clone()
    configs = [("product", {})] + [(f"nt{nt} mw{mw}", {"MV_PW_NT": str(nt), "MV_PW_MW": str(mw)}) for mw in (4, 2, 1) for nt in (1, 2, 4) if not (mw == 1 and nt == 4)]
    print(f"pointwise {cin} -> {cout} @ {hw}x{hw}, batch {n}: {nbytes / 1e6:.0f} MB, {flop / 1e9:.1f} GFLOP")
    for name, env in configs:
        ctx = _lib.tuning_library() if env else None
        if ctx:
            ctx.__enter__()
        os.environ.update(env)
        try:
            out = fn()
            torch.cuda.synchronize()
            same = torch.equal(out, ref)
            ts = []
            for _ in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
            ms = sorted(ts)[len(ts) // 2]
            print(f"  {name:10s} {_lib.last_kernel():22s} {ms * 1e3:7.1f} us  {nbytes / ms / 1e6:6.0f} GB/s  {flop / ms / 1e9:5.1f} TFLOP/s  {'' if same else 'DIFFERS'}", flush=True)
        finally:
            for k in env:
                os.environ.pop(k)
            if ctx:
                ctx.__exit__(None, None, None)
