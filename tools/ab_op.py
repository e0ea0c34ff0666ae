#!/usr/bin/env python3
"""Generic A/B of library variants on one operator, interleaved rounds in one process (runs on the GPU box).

    MV_BUILD_VARIANT=c3plain MV_VARIANT_SOURCES=conv3x3_c3.hip MV_HIPCC_EXTRA=-DMV_C3_NT=0 python cpu-vision_amd/_build.py
    python tools/ab_op.py --op conv tuning c3plain tuning@MV_C3_TH=8        (name[@ENV=VAL,...]; 'base' = the product library)
ops: conv (cfg4), sobel5 (cfg3), sep5, blur3 (32 x 4K f32), u8blur3, u8sharp, sharp
Every variant must produce the same bytes (checked on a strided sample)."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--op", default="conv")
ap.add_argument("--rounds", type=int, default=14)
ap.add_argument("--no-check", action="store_true", help="ablation builds produce wrong results on purpose")
ap.add_argument("specs", nargs="*", default=["base"])
a = ap.parse_args()
fp, vp, i32, i64 = C.POINTER(C.c_float), C.c_void_p, C.c_int, C.c_int64
libs = {}
for spec in a.specs:
    n = spec.split("@")[0]
    if n not in libs:
        p = ROOT / "cpu-vision_amd" / "lib" / ("libmi355vision.so" if n == "base" else f"libmi355vision_{n}.so")
        lib = C.CDLL(str(p))
        lib.mv_conv3x3_bias_relu_f32.argtypes = [vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]
        lib.mv_gaussian_sobel_f32.argtypes = [vp, vp, vp, i64, i32, i32, fp, i32, fp, i32, vp]
        lib.mv_separable_blur_f32.argtypes = [vp, vp, i64, i32, i32, fp, i32, fp, i32, vp]
        lib.mv_gaussian_blur_f32.argtypes = [vp, vp, i64, i32, i32, fp, i32, fp, i32, vp]
        lib.mv_gaussian_blur_u8.argtypes = [vp, vp, i64, i32, i32, fp, i32, fp, i32, vp]
        lib.mv_sharpness_u8.argtypes = [vp, vp, i64, i32, i32, C.c_double, i32, vp]
        lib.mv_sharpness_f32.argtypes = [vp, vp, i64, i32, i32, C.c_double, i32, C.c_float, i32, vp]
        libs[n] = lib
g = torch.Generator(device="cuda").manual_seed(0)
s = torch.cuda.current_stream().cuda_stream
k3 = (C.c_float * 3)(0.25, 0.5, 0.25)
k5 = (C.c_float * 5)(0.1, 0.2, 0.4, 0.2, 0.1)
if a.op == "conv":
    x = torch.rand((256, 3, 224, 224), generator=g, device="cuda")
    w = torch.randn((64, 3, 3, 3), generator=g, device="cuda") * 0.06
    b = torch.rand(64, generator=g, device="cuda") - 0.5
    outs = [torch.empty((256, 64, 224, 224), device="cuda")]
    nbytes = (x.numel() + outs[0].numel() + w.numel()) * 4
    call = lambda lib: lib.mv_conv3x3_bias_relu_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), outs[0].data_ptr(), 256, 3, 224, 224, 64, 1, s)  # noqa: E731
else:
    u8 = a.op.startswith("u8")
    shape = (32, 3, 2160, 3840)
    x = torch.randint(0, 256, shape, generator=g, device="cuda", dtype=torch.uint8) if u8 else torch.rand(shape, generator=g, device="cuda")
    outs = [torch.empty_like(x) for _ in range(2 if a.op == "sobel5" else 1)]
    nbytes = x.numel() * x.element_size() * (1 + len(outs))
    P = (x.data_ptr(), *[o.data_ptr() for o in outs])
    call = {
        "sobel5": lambda lib: lib.mv_gaussian_sobel_f32(*P, 96, 2160, 3840, k5, 5, k5, 5, s),
        "sep5": lambda lib: lib.mv_separable_blur_f32(*P, 96, 2160, 3840, k5, 5, k5, 5, s),
        "blur3": lambda lib: lib.mv_gaussian_blur_f32(*P, 96, 2160, 3840, k3, 3, k3, 3, s),
        "u8blur3": lambda lib: lib.mv_gaussian_blur_u8(*P, 96, 2160, 3840, k3, 3, k3, 3, s),
        "u8sharp": lambda lib: lib.mv_sharpness_u8(*P, 96, 2160, 3840, 1.5, 0, s),
        "sharp": lambda lib: lib.mv_sharpness_f32(*P, 96, 2160, 3840, 1.5, 0, 1.0, 0, s),
    }[a.op]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    torch.cuda.synchronize()
    assert rc == 0, rc
    return e0.elapsed_time(e1)


res = {spec: [] for spec in a.specs}
chk = None
for r in range(a.rounds):
    for spec in a.specs:
        n, _, envs = spec.partition("@")
        kv = dict(e.split("=") for e in envs.split(",") if e)
        os.environ.update(kv)
        ms = timed(lambda: call(libs[n]))
        if r == 0:
            c = sum(float(o.view(-1)[::97].double().sum().item()) for o in outs)
            chk = c if chk is None else chk
            assert a.no_check or c == chk, (spec, c, chk)
        if r >= 2:
            res[spec].append(ms)
        for e in kv:
            os.environ.pop(e)
for spec, v in res.items():
    v.sort()
    med = v[len(v) // 2]
    print(f"{a.op:8s} {spec:36s} median {med:7.4f} ms  min {v[0]:7.4f}  {nbytes / med / 1e6:7.0f} GB/s ({nbytes / med / 1e6 / 80:4.1f} % of 8 TB/s)")
