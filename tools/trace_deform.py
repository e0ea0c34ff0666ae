#!/usr/bin/env python3
"""Cycle stamps inside k_deform_fused (runs on the GPU box): where a producer wave and a consumer wave of block 0 spend a chunk.

    MV_BUILD_VARIANT=dftrace MV_VARIANT_SOURCES=deform_fused.hip MV_HIPCC_EXTRA=-DMV_DF_TRACE python cpu-vision_amd/_build.py
    python tools/trace_deform.py [variant ...]
The stamps land behind the output tensor."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

n, cin, cout, hw = 8, 256, 256, 64
vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((n, cin, hw, hw), generator=g, device="cuda")
w = torch.randn((cout, cin, 3, 3), generator=g, device="cuda") * 0.05
b = torch.rand(cout, generator=g, device="cuda")
off = torch.randn((n, 18, hw, hw), generator=g, device="cuda") * 1.5
mask = torch.rand((n, 9, hw, hw), generator=g, device="cuda")
s = torch.cuda.current_stream().cuda_stream
P = ["top", "corner reads issued", "loads issued / W stored", "-", "-", "window stored", "gather done", "after barrier"]
for name in sys.argv[1:] or ["dftrace"]:
    lib = C.CDLL(str(ROOT / "cpu-vision_amd" / "lib" / f"libmi355vision_{name}.so"))
    lib.mv_deform_conv2d_f32.argtypes = [vp] * 6 + [i64] + [i32] * 15 + [vp, i64, vp]
    y = torch.zeros(n * cout * hw * hw + 4096, device="cuda")
    for _ in range(3):
        y.zero_()
        rc = lib.mv_deform_conv2d_f32(x.data_ptr(), w.data_ptr(), off.data_ptr(), mask.data_ptr(), b.data_ptr(), y.data_ptr(), n, cin, hw, hw,
                                      cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, None, 0, s)
        assert rc == 0
        torch.cuda.synchronize()
    t = y[n * cout * hw * hw:n * cout * hw * hw + 2048].view(torch.int64).cpu()
    print(f"== {name}")
    for it in range(4):
        pr = [int(t[256 + it * 16 + k]) for k in range(8)]
        co = [int(t[it * 16 + k]) for k in range(3)]
        base = pr[0]
        print(f"chunk {10 + it}: producer " + ", ".join(f"{P[k]} +{pr[k] - base}" for k in (1, 2, 5, 6, 7)))
        print(f"          consumer top {co[0] - base:+d}, MFMAs done {co[1] - base:+d}, after barrier {co[2] - base:+d}   (MFMA phase {co[1] - co[0]} cycles)")
