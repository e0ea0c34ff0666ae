#!/usr/bin/env python3
"""cfg3 on several independently allocated (gx, gy) pairs, three timed series each: how much of this kernel's run-to-run spread is WHERE its
output buffers land?  (Round 3: six pairs in one process ran 1.555 ... 1.705 ms, each pair reproducibly; storing gy one row behind gx -- a
variant built for this probe, MV_SEPFAST_SKEW, since removed -- changed nothing, so it is not the two write streams colliding.)  GPU box."""
import os
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402
import ctypes as C  # noqa: E402

n, H, W = 32, 2160, 3840
VARIANT = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
if True:
    lib = _lib.load()
    libs = [("product", lib)]
    if VARIANT:  # a variant build of the library on the SAME allocations (MV_BUILD_VARIANT=<name> ... python cpu-vision_amd/_build.py)
        libs.append((VARIANT, C.CDLL(str(Path(__file__).resolve().parent.parent / "cpu-vision_amd" / "lib" / f"libmi355vision_{VARIANT}.so"))))
    hip = C.CDLL("libamdhip64.so")
    s = torch.cuda.current_stream().cuda_stream
    for _, l in libs:
        l.mv_gaussian_sobel_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]
    _unused = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]
    k5 = (C.c_float * 5)(0.1, 0.2, 0.4, 0.2, 0.1)
    x = torch.rand((n, 3, H, W), device="cuda")
    nb = x.numel() * 4
    pairs = []
    for i in range(6):
        a, b = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(a), C.c_size_t(nb + (i % 3) * (1 << 21))) == 0 and hip.hipMalloc(C.byref(b), C.c_size_t(nb)) == 0
        pairs.append((a.value, b.value))
    ref = None
    for i, (gx, gy) in enumerate(pairs):
        line = f"pair {i}: gx {gx:#x} gy {gy:#x}:"
        for series, (lname, lib) in enumerate(libs * (3 if len(libs) == 1 else 2)):
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(3):
                    assert lib.mv_gaussian_sobel_f32(C.c_void_p(x.data_ptr()), C.c_void_p(gx), C.c_void_p(gy), n * 3, H, W, k5, 5, k5, 5, C.c_void_p(s)) == 0
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 3)
            line += f"   {lname if len(libs) > 1 else 'series ' + str(series)}: {statistics.median(ts):.4f} ms"
        print(line, flush=True)


# ---- which kernels show the band?  The same six output pairs under four operators of the same size (one input, 32 x 4K):
#      fused Gaussian -> Sobel (2 outputs), plain Sobel pair (2 outputs, another kernel), separable 5x5 blur and 3x3 blur (1 output: gx only)
if not VARIANT:
    lib.mv_sobel_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.mv_separable_blur_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]
    lib.mv_gaussian_blur_f32.argtypes = lib.mv_separable_blur_f32.argtypes
    k3 = (C.c_float * 3)(0.2, 0.6, 0.2)
    xp = C.c_void_p(x.data_ptr())
    ops = {
        "gaussian -> sobel (2 out)": lambda gx, gy: lib.mv_gaussian_sobel_f32(xp, C.c_void_p(gx), C.c_void_p(gy), n * 3, H, W, k5, 5, k5, 5, C.c_void_p(s)),
        "sobel pair (2 out)": lambda gx, gy: lib.mv_sobel_f32(xp, C.c_void_p(gx), C.c_void_p(gy), n * 3, H, W, 1, C.c_void_p(s)),
        "separable 5x5 (1 out)": lambda gx, gy: lib.mv_separable_blur_f32(xp, C.c_void_p(gx), n * 3, H, W, k5, 5, k5, 5, C.c_void_p(s)),
        "gaussian 3x3 (1 out)": lambda gx, gy: lib.mv_gaussian_blur_f32(xp, C.c_void_p(gx), n * 3, H, W, k3, 3, k3, 3, C.c_void_p(s)),
    }
    for name, op in ops.items():
        line = f"{name:28s}"
        for gx, gy in pairs:
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(3):
                    assert op(gx, gy) == 0
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 3)
            line += f"  {statistics.median(ts):.4f}"
        print(line + "  ms per pair", flush=True)
