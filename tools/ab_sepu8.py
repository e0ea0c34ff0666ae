#!/usr/bin/env python3
"""A/B of the uint8 separable blur (mv_separable_blur_u8, kernel sides 5 and 7) over library variants and strip heights,
interleaved rounds in one process, 32 x 4K uint8 frames.

    MV_BUILD_VARIANT=pf2 MV_VARIANT_SOURCES=dwk_u8.hip MV_HIPCC_EXTRA=-DMV_DWK_PF=2 python cpu-vision_amd/_build.py
    python tools/ab_sepu8.py tuning pf2 tuning@MV_DWK_U8_ROWS=32        (name[@ENV=VAL,...]; 'base' = the product library)
"""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

specs = sys.argv[1:] or ["base"]
libs = {}
for spec in specs:
    n = spec.split("@")[0]
    if n not in libs:
        p = ROOT / "cpu-vision_amd" / "lib" / ("libmi355vision.so" if n == "base" else f"libmi355vision_{n}.so")
        lib = C.CDLL(str(p))
        fp = C.POINTER(C.c_float)
        lib.mv_separable_blur_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_void_p]
        lib.mv_gaussian_blur_u8.argtypes = lib.mv_separable_blur_u8.argtypes
        lib.mv_last_error.restype = C.c_char_p
        libs[n] = lib
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randint(0, 256, (32, 3, 2160, 3840), generator=g, device="cuda", dtype=torch.uint8)
y = torch.empty_like(x)
taps = {5: (C.c_float * 5)(0.1, 0.2, 0.4, 0.2, 0.1), 7: (C.c_float * 7)(0.05, 0.1, 0.2, 0.3, 0.2, 0.1, 0.05),
        3: (C.c_float * 3)(0.25, 0.5, 0.25)}
s = torch.cuda.current_stream().cuda_stream


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    torch.cuda.synchronize()
    assert rc == 0, rc
    return e0.elapsed_time(e1)


res = {spec: {k: [] for k in taps} for spec in specs}
ref_out = {}
for r in range(12):
    for spec in specs:
        n, _, envs = spec.partition("@")
        kv = dict(e.split("=") for e in envs.split(",") if e)
        os.environ.update(kv)
        for k, t in taps.items():
            ms = timed(lambda: libs[n].mv_separable_blur_u8(x.data_ptr(), y.data_ptr(), 96, 2160, 3840, t, k, t, k, s))
            if r == 0:  # every variant must produce the same bytes
                chk = int(y[::7].to(torch.int64).sum().item())
                assert ("abl" in spec) or ref_out.setdefault(k, chk) == chk, (spec, k)
            if r >= 2:
                res[spec][k].append(ms)
        ms = timed(lambda: libs[n].mv_gaussian_blur_u8(x.data_ptr(), y.data_ptr(), 96, 2160, 3840, taps[3], 3, taps[3], 3, s))
        if r >= 2:
            res[spec].setdefault("2d3", []).append(ms)
        for e in kv:
            os.environ.pop(e)
nbytes = x.numel() * 2
for spec, d in res.items():
    line = f"{spec:40s}"
    for k, v in d.items():
        v.sort()
        med = v[len(v) // 2]
        line += f"  {k if k == '2d3' else f'{k}x{k}'} {med:6.3f} ms ({nbytes / med / 1e6 / 80:4.1f} %)"
    print(line)
