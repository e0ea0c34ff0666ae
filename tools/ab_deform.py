#!/usr/bin/env python3
"""A/B of library variants on deform_conv2d (runs on the GPU box), interleaved rounds in one process.

    MV_BUILD_VARIANT=dfnog MV_VARIANT_SOURCES=deform_fused.hip MV_HIPCC_EXTRA=-DMV_DF_ABLATE=1 python cpu-vision_amd/_build.py
    python tools/ab_deform.py [--shape 8,256,256,64] base tuning@MV_DEFORM_UNFUSED=1 dfnog       (name[@ENV=VAL,...])
Variants whose name starts with `df` are ablations (wrong results on purpose); every other one must equal the first bit for bit."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="8,256,256,64", help="n,cin,cout,hw")
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("specs", nargs="*", default=["base"])
a = ap.parse_args()
n, cin, cout, hw = [int(v) for v in a.shape.split(",")]
vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
libs = {}
for spec in a.specs:
    name = spec.split("@")[0]
    if name not in libs:
        p = ROOT / "cpu-vision_amd" / "lib" / ("libmi355vision.so" if name == "base" else f"libmi355vision_{name}.so")
        lib = C.CDLL(str(p))
        lib.mv_deform_conv2d_f32.argtypes = [vp] * 6 + [i64] + [i32] * 15 + [vp, i64, vp]
        lib.mv_last_kernel.restype = C.c_char_p
        libs[name] = lib
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.rand((n, cin, hw, hw), generator=g, device="cuda")
w = torch.randn((cout, cin, 3, 3), generator=g, device="cuda") * 0.05
b = torch.rand(cout, generator=g, device="cuda")
off = torch.randn((n, 18, hw, hw), generator=g, device="cuda") * 1.5
mask = torch.rand((n, 9, hw, hw), generator=g, device="cuda")
y = torch.empty((n, cout, hw, hw), device="cuda")
ws = torch.empty(n * cin * 9 * hw * hw, device="cuda")
s = torch.cuda.current_stream().cuda_stream
flops = 2.0 * n * cout * cin * 9 * hw * hw


def call(spec):
    name, _, env = spec.partition("@")
    kv = dict(e.split("=") for e in env.split(",")) if env else {}
    os.environ.update(kv)
    try:
        rc = libs[name].mv_deform_conv2d_f32(x.data_ptr(), w.data_ptr(), off.data_ptr(), mask.data_ptr(), b.data_ptr(), y.data_ptr(), n, cin,
                                              hw, hw, cout, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, ws.data_ptr(), ws.numel() * 4, s)
        assert rc == 0, rc
    finally:
        for k in kv:
            os.environ.pop(k)
    return libs[name].mv_last_kernel().decode()


first = None
kern = {}
for spec in a.specs:
    kern[spec] = call(spec)
    torch.cuda.synchronize()
    if not spec.startswith("df"):
        if first is None:
            first = y.clone()
        else:
            assert torch.equal(first, y), f"{spec} differs from {a.specs[0]}"
times = {sp: [] for sp in a.specs}
for r in range(a.rounds):
    for spec in a.specs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            call(spec)
        e1.record()
        e1.synchronize()
        times[spec].append(e0.elapsed_time(e1) / 5)
for spec in a.specs:
    t = sorted(times[spec])
    med = t[len(t) // 2]
    print(f"{spec:40s} {kern[spec]:34s} median {med:.3f} ms  min {t[0]:.3f}  {flops / med / 1e9:7.1f} TFLOP/s", flush=True)
