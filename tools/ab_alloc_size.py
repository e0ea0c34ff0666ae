#!/usr/bin/env python3
"""Is a kernel's OUTPUT stream slower when its buffer is a slice of a large allocation?  (tools/ab_placement.py: cfg3 writes 15 % slower into a 12.5 GB
pool than into buffers of their own, wherever the input lives.)  3x3 blur (one read stream, one write stream) on 32 and 128 4K frames, output = its own
hipMalloc against a slice of a pool of 1x, 2x, 4x its size, and the pool's first touch ruled out (every buffer is written once before timing).  GPU box."""
import ctypes as C
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib  # noqa: E402

lib = _lib.load()
hip = C.CDLL("libamdhip64.so")
s = torch.cuda.current_stream().cuda_stream
fp = C.POINTER(C.c_float)
lib.mv_gaussian_blur_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_void_p]
k3 = (C.c_float * 3)(0.2, 0.6, 0.2)
H, W = 2160, 3840


def hmalloc(nbytes):
    q = C.c_void_p()
    rc = hip.hipMalloc(C.byref(q), C.c_size_t(nbytes))
    assert rc == 0, rc
    return q.value


def timed(xa, ya, planes):
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            rc = lib.mv_gaussian_blur_f32(C.c_void_p(xa), C.c_void_p(ya), planes, H, W, k3, 3, k3, 3, C.c_void_p(s))
            assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    return statistics.median(ts)


for frames in (32, 128):
    planes = frames * 3
    nb = planes * H * W * 4
    x = hmalloc(nb)
    hip.hipMemset(C.c_void_p(x), 0x3c, C.c_size_t(nb))
    own = hmalloc(nb)
    rows = [("output = its own hipMalloc", own)]
    pools = []
    for mult in (2, 4):
        if mult * nb > 60 << 30:
            continue
        pool = hmalloc(mult * nb)
        hip.hipMemset(C.c_void_p(pool), 0, C.c_size_t(mult * nb))
        pools.append(pool)
        rows.append((f"output = first slice of a pool of {mult}x its size", pool))
        rows.append((f"output = last slice of a pool of {mult}x its size", pool + (mult - 1) * nb))
    torch.cuda.synchronize()
    for label, ya in rows:
        t = timed(x, ya, planes)
        print(f"{frames:3d} frames ({nb / 1e9:5.2f} GB out): {label:52s} {t:7.4f} ms  {2 * nb / t / 1e6:7.1f} GB/s", flush=True)
    for q in [x, own] + pools:
        hip.hipFree(C.c_void_p(q))
