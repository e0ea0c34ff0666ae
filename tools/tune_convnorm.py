#!/usr/bin/env python3
"""A/B + ablation harness for csrc/convnorm.hip (runs on the GPU box): builds -D variants into /tmp, loads them into
ONE process and times one Conv2dNormActivation layer in interleaved rounds.

    python tools/tune_convnorm.py --layer pw:16:96:112 base: nostore:MV_ABLATE_STORE nomfma:MV_ABLATE_MFMA
    layer syntax: pw:CIN:COUT:HW | dw:C:HW:STRIDE | stem:COUT:HW:STRIDE     (batch from --batch)
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import tools.tune_dw3x3 as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", default="pw:16:96:112")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    T.VARIANT_FILES = ["convnorm.hip"]
    libs = []
    for spec in (a.variants or ["base:"]):
        name, _, defs = spec.partition(":")
        lib = T.build_variant("cn_" + name, [d for d in defs.split(",") if d])
        vp, i, i64 = C.c_void_p, C.c_int, C.c_int64
        lib.mv_conv_norm_act_f32.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, i64, i, i, i, i, i, i, i, vp]
        libs.append((name, lib))
    kind, *nums = a.layer.split(":")
    nums = [int(v) for v in nums]
    n = a.batch
    if kind == "pw":
        cin, cout, hw = nums
        code, stride, wshape = 2, 1, (cout, cin, 1, 1)
    elif kind == "dw":
        cin, hw, stride = nums
        cout, code, wshape = cin, 1, (cin, 1, 3, 3)
    else:
        cout, hw, stride = nums
        cin, code, wshape = 3, 0, (cout, 3, 3, 3)
    oh = (hw - 1) // stride + 1
    x = torch.rand((n, cin, hw, hw), device="cuda")
    w = torch.randn(wshape, device="cuda") * 0.1
    al, be = torch.rand(cout, device="cuda") + 0.5, torch.rand(cout, device="cuda")
    y = torch.empty((n, cout, oh, oh), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    alg = (x.numel() + y.numel()) * 4
    flop = 2.0 * y.numel() * (cin if kind != "dw" else 1) * (1 if kind == "pw" else 9)

    def run(lib):
        rc = lib.mv_conv_norm_act_f32(code, x.data_ptr(), w.data_ptr(), None, al.data_ptr(), be.data_ptr(), None, y.data_ptr(), n, cin, hw, hw,
                                      cout, stride, 2, 2, s)
        assert rc == 0

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    res = {name: [] for name, _ in libs}
    ref = None
    for name, lib in libs:
        run(lib)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        elif "ablate" not in name and not name.startswith("no"):
            assert torch.equal(ref, y), f"variant {name} changes the result"
    for _ in range(a.rounds):
        for name, lib in libs:
            res[name].append(timed(lambda: run(lib)))
    print(f"layer {a.layer} batch {n}: algorithmic {alg / 1e6:.1f} MB, {flop / 1e9:.2f} GFLOP")
    for name, ts in res.items():
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"{name:24s} {med * 1e3:9.1f} us  {alg / med / 1e6:8.1f} GB/s  {flop / med / 1e9:7.2f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
