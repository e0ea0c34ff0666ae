#!/usr/bin/env python3
"""A/B harness for the headline 3x3 kernel (runs on the GPU box).

Builds variants of libmi355vision.so with different -D knobs for dw3x3.hip into /tmp, loads them all into ONE
process and times them in interleaved rounds on the bench workload (CDNA guide, methodology rule 24: perf deltas
come from interleaved rounds in one process).  Also times torch's device copy of the same buffers as the
achievable-HBM yardstick of this box.

    python tools/tune_dw3x3.py [--frames 64] [--rounds 7] [variant ...]
    variant syntax:  name:KEY=VAL,KEY=VAL[@ENV=VAL]     e.g.  g8:MV_DW3X3_GROUP=8   r32:@MV_DW3X3_ROWS=32
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

CSRC = ROOT / "cpu-vision_amd" / "csrc"
OBJ = ROOT / "cpu-vision_amd" / "build"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math"]
ALL = ["abi.hip", "dw3x3.hip", "dw3x3_u8.hip", "dwk_u8.hip", "dwtile.hip", "separable.hip", "sepfast.hip", "sepstream.hip", "conv3x3_mfma.hip", "conv3x3_c3.hip", "conv3x3_gen.hip", "cnn_ops.hip", "linear_mfma.hip", "resize.hip", "convnorm.hip", "deform.hip"]
VARIANT_FILES = ["dw3x3.hip", "dwtile.hip"]


def build_variant(name, defs):
    OTHERS = [f for f in ALL if f not in VARIANT_FILES]
    out = Path(f"/tmp/mv_{name}.so")
    objs = []
    for src in VARIANT_FILES:
        obj = Path(f"/tmp/mv_{name}_{src}.o")
        cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *[f"-D{d}" for d in defs], "-c", str(CSRC / src), "-o", str(obj)]
        subprocess.run(cmd, check=True)
        objs.append(str(obj))
    others = []
    for s in OTHERS:
        o = OBJ / (s + ".o")
        if not o.exists():
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", str(CSRC / s), "-o", str(o)], check=True)
        others.append(str(o))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *objs, *others], check=True)
    lib = C.CDLL(str(out))
    fp = C.POINTER(C.c_float)
    lib.mv_gaussian_blur_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_void_p]
    lib.mv_separable_blur_f32.argtypes = lib.mv_gaussian_blur_f32.argtypes
    lib.mv_conv3x3_bias_relu_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.mv_gaussian_sobel_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_void_p]
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--h", type=int, default=2160)
    ap.add_argument("--w", type=int, default=3840)
    ap.add_argument("--op", default="blur3", choices=["blur3", "sep5", "sobel5", "conv"])
    ap.add_argument("--files", default="dw3x3.hip,dwtile.hip", help="sources rebuilt per variant")
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    global VARIANT_FILES
    VARIANT_FILES = a.files.split(",")
    specs = a.variants or ["base:"]
    variants = []
    for spec in specs:
        name, _, rest = spec.partition(":")
        defs_s, _, env_s = rest.partition("@")
        defs = [d for d in defs_s.split(",") if d]
        env = dict(e.split("=") for e in env_s.split(",") if e)
        variants.append((name, build_variant(name, defs), env))
    n, H, W = a.frames, a.h, a.w
    if a.op == "conv":
        n, H, W = 256, 224, 224
    x = torch.rand((n, 3, H, W), device="cuda")
    y = torch.empty_like(x) if a.op != "conv" else torch.empty((n, 64, H, W), device="cuda")
    cw = torch.randn((64, 3, 3, 3), device="cuda") * 0.06
    cb = torch.rand(64, device="cuda") - 0.5
    k = (C.c_float * 3)(0.2, 0.6, 0.2)
    k5 = (C.c_float * 5)(0.1, 0.2, 0.4, 0.2, 0.1)
    y2 = torch.empty_like(x) if a.op == "sobel5" else None
    alg = n * 3 * H * W * (12 if a.op == "sobel5" else 8)
    if a.op == "conv":
        alg = x.numel() * 4 + y.numel() * 4
    s = torch.cuda.current_stream().cuda_stream

    def run(lib, env):
        for kk, v in env.items():
            os.environ[kk] = v
        if a.op == "blur3":
            rc = lib.mv_gaussian_blur_f32(x.data_ptr(), y.data_ptr(), n * 3, H, W, k, 3, k, 3, s)
        elif a.op == "conv":
            rc = lib.mv_conv3x3_bias_relu_f32(x.data_ptr(), cw.data_ptr(), cb.data_ptr(), y.data_ptr(), n, 3, H, W, 64, 1, s)
        elif a.op == "sep5":
            rc = lib.mv_separable_blur_f32(x.data_ptr(), y.data_ptr(), n * 3, H, W, k5, 5, k5, 5, s)
        else:
            rc = lib.mv_gaussian_sobel_f32(x.data_ptr(), y.data_ptr(), y2.data_ptr(), n * 3, H, W, k5, 5, k5, 5, s)
        for kk in env:
            os.environ.pop(kk, None)
        assert rc == 0

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    results = {name: [] for name, _, _ in variants}
    results["torch_copy"] = []
    ref_out = None
    for name, lib, env in variants:  # warmup + cross-variant agreement
        run(lib, env)
        torch.cuda.synchronize()
        if ref_out is None:
            ref_out = y[0].clone()
        elif "ablate" not in name:
            assert torch.equal(ref_out, y[0]), f"variant {name} changes the result"
    for _ in range(a.rounds):
        for name, lib, env in variants:
            results[name].append(timed(lambda: run(lib, env)))
        results["torch_copy"].append(timed(lambda: y.copy_(x) if y.shape == x.shape else y.zero_()))
    print(f"{'variant':28s} {'median ms':>10s} {'min ms':>8s} {'GB/s(med)':>10s} {'%8TB/s':>7s}")
    summary = {}
    for name, ts in results.items():
        ts = sorted(ts)
        med, mn = ts[len(ts) // 2], ts[0]
        gbs = alg / (med * 1e-3) / 1e9
        summary[name] = {"median_ms": med, "min_ms": mn, "gbps": gbs}
        print(f"{name:28s} {med:10.4f} {mn:8.4f} {gbs:10.1f} {gbs / 80:7.2f}")
    print(json.dumps({"frames": n, "h": H, "w": W, "results": summary}))


if __name__ == "__main__":
    main()
