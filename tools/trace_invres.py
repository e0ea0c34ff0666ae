#!/usr/bin/env python3
"""Cycle stamps inside k_invres (runs on the GPU box): where workgroup (0, 0)'s four waves spend their time.

    MV_BUILD_VARIANT=irtrace MV_VARIANT_SOURCES=invres.hip MV_HIPCC_EXTRA=-DMV_IR_TRACE python cpu-vision_amd/_build.py
    python tools/trace_invres.py [--batch 64]
The stamps (s_memtime, 100 MHz-independent shader clock) land behind the output tensor."""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--variant", default="irtrace")
ap.add_argument("--wide", action="store_true", help="the 112 / 56-pixel blocks (k_invres_wide) instead of the 28 / 14 / 7-pixel ones")
a = ap.parse_args()
vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
lib = C.CDLL(str(ROOT / "cpu-vision_amd" / "lib" / f"libmi355vision_{a.variant}.so"))
lib.mv_inverted_residual_f32.argtypes = [vp] * 10 + [i32, vp, i64] + [i32] * 7 + [vp, i64, vp]
lib.mv_inverted_residual_workspace_bytes.restype = i64
lib.mv_inverted_residual_workspace_bytes.argtypes = [i64] + [i32] * 6
lib.mv_inverted_residual_k_slices.argtypes = [i64] + [i32] * 6 + [vp]
lib.mv_last_kernel.restype = C.c_char_p
g = torch.Generator(device="cuda").manual_seed(0)
s = torch.cuda.current_stream().cuda_stream
n = a.batch
SMALL = ((32, 32, 28, 1), (32, 64, 28, 2), (64, 64, 14, 1), (96, 96, 14, 1), (96, 160, 14, 2), (160, 160, 7, 1), (160, 320, 7, 1))
for cin, cout, side, stride in (((32, 16, 112, 1), (16, 24, 112, 2), (24, 24, 56, 1), (24, 32, 56, 2)) if a.wide else SMALL):
    hid = cin if (cin, side) == (32, 112) else 6 * cin  # MobileNetV2's first block has no expansion
    o = (side - 1) // stride + 1
    x = torch.rand((n, cin, side, side), generator=g, device="cuda")
    w1 = torch.randn((hid, cin), generator=g, device="cuda") * 0.1
    wd = torch.randn((hid, 9), generator=g, device="cuda") * 0.3
    w2 = torch.randn((cout, hid), generator=g, device="cuda") * 0.05
    th = torch.rand((hid,), generator=g, device="cuda") + 0.5
    tc = torch.rand((cout,), generator=g, device="cuda") + 0.5
    ysz = n * cout * o * o
    y = torch.zeros(ysz + 1024, device="cuda")
    nb = lib.mv_inverted_residual_workspace_bytes(n, cin, hid, cout, side, side, stride)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        e = (None, None, None) if hid == cin else (w1.data_ptr(), th.data_ptr(), th.data_ptr())
        rc = lib.mv_inverted_residual_f32(x.data_ptr(), *e, wd.data_ptr(), th.data_ptr(), th.data_ptr(),
                                          w2.data_ptr(), tc.data_ptr(), tc.data_ptr(), 0, y.data_ptr(), n, cin, hid, cout, side, side, stride, 2,
                                          ws.data_ptr(), nb, s)
        assert rc == 0, rc
        torch.cuda.synchronize()
    t = y[ysz:ysz + 1024].view(torch.int64).cpu().reshape(8, 64)
    print(f"== {cin}->{hid}->{cout} @{side} s{stride} batch {n}: {lib.mv_last_kernel().decode()}")
    base = int(t[0, 0])
    for wv in range(8):
        st = [int(v) - base for v in t[wv] if int(v) != 0]
        if len(st) < 3:
            continue
        chunks = (len(st) - 3) // 6
        line = f"  wave {wv}: start {st[0]:+6d}  region in LDS {st[1]:6d}"
        for c in range(chunks):
            q = st[2 + 6 * c: 8 + 6 * c]
            line += f"\n     chunk {c}: operands {q[0]:6d} | expand +{q[1] - q[0]:5d} (barrier +{q[2] - q[1]:4d}) | depthwise +{q[3] - q[2]:5d} (barrier +{q[4] - q[3]:4d}) | project +{q[5] - q[4]:5d}"
        line += f"\n     end {st[-1]:6d} cycles (epilogue +{st[-1] - st[-2]})"
        print(line, flush=True)
