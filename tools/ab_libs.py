#!/usr/bin/env python3
"""A/B whole-library variants (cpu-vision_amd/lib/libmi355vision_<name>.so, built with MV_BUILD_VARIANT) on the uint8 3x3
blur and sharpness, interleaved rounds in one process.   python tools/ab_libs.py base g2 g3 g6"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

names = sys.argv[1:] or ["base"]
libs = {}
for n in names:
    p = ROOT / "cpu-vision_amd" / "lib" / ("libmi355vision.so" if n == "base" else f"libmi355vision_{n}.so")
    lib = C.CDLL(str(p))
    fp = C.POINTER(C.c_float)
    lib.mv_gaussian_blur_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_void_p]
    lib.mv_sharpness_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]
    libs[n] = lib
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randint(0, 256, (32, 3, 2160, 3840), generator=g, device="cuda", dtype=torch.uint8)
y = torch.empty_like(x)
k = (C.c_float * 3)(0.25, 0.5, 0.25)
k5 = (C.c_float * 5)(0.1, 0.2, 0.4, 0.2, 0.1)
k7 = (C.c_float * 7)(0.05, 0.1, 0.2, 0.3, 0.2, 0.1, 0.05)
s = torch.cuda.current_stream().cuda_stream


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


res = {n: ([], [], [], []) for n in names}
for r in range(12):
    for n, lib in libs.items():
        a = timed(lambda: lib.mv_gaussian_blur_u8(x.data_ptr(), y.data_ptr(), 96, 2160, 3840, k, 3, k, 3, s))
        b = timed(lambda: lib.mv_sharpness_u8(x.data_ptr(), y.data_ptr(), 96, 2160, 3840, 1.5, 0, s))
        c5 = timed(lambda: lib.mv_gaussian_blur_u8(x.data_ptr(), y.data_ptr(), 96, 2160, 3840, k5, 5, k5, 5, s))
        c7 = timed(lambda: lib.mv_gaussian_blur_u8(x.data_ptr(), y.data_ptr(), 96, 2160, 3840, k7, 7, k7, 7, s))
        if r >= 2:
            res[n][0].append(a), res[n][1].append(b), res[n][2].append(c5), res[n][3].append(c7)
for n, (a, b, c5, c7) in res.items():
    a.sort(), b.sort(), c5.sort(), c7.sort()
    print(f"{n:8s} blur3 {a[len(a) // 2]:6.3f} ms   sharpness {b[len(b) // 2]:6.3f} ms   blur5 {c5[len(c5) // 2]:6.3f} ms   blur7 {c7[len(c7) // 2]:6.3f} ms")
