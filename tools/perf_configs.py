#!/usr/bin/env python3
"""Per-kernel throughput of every BASELINE config on one MI355X (runs on the GPU box).

For each workload: median launch time from HIP events over `--rounds` launches (inputs resident in HBM, working
set >> 256 MB Infinity Cache), algorithmic bytes / flops per launch, achieved GB/s | TFLOP/s and
the fraction of the roof that bounds it (HBM 8 TB/s; fp32 MFMA 157.3 TFLOP/s).
"""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import cpu_vision_amd as mv  # noqa: E402
from cpu_vision_amd import functional as F  # noqa: E402

HBM, MFMA = 8000.0, 157.3


def timeit(fn, rounds):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    rows = []

    def rec(name, ms, mn, nbytes, flops=0.0, note=""):
        gbs = nbytes / (ms * 1e-3) / 1e9
        tf = flops / (ms * 1e-3) / 1e12
        rows.append({"workload": name, "median_ms": round(ms, 4), "min_ms": round(mn, 4), "alg_bytes": nbytes,
                     "GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM, 4), "TFLOPs": round(tf, 2),
                     "mfma_frac": round(tf / MFMA, 4), "note": note})
        print(f"{name:58s} {ms:9.4f} ms  {gbs:8.1f} GB/s ({gbs / HBM * 100:5.1f}% HBM)" + (f"  {tf:6.1f} TF ({tf / MFMA * 100:4.1f}% MFMA)" if flops else ""), flush=True)

    def want(tag):
        return not a.only or a.only in tag

    g = torch.Generator(device="cuda").manual_seed(0)
    n4k = 32
    x4k = torch.rand((n4k, 3, 2160, 3840), generator=g, device="cuda")
    el4k = x4k.numel()

    if want("metric"):
        y4k = torch.empty_like(x4k)
        ms, mn = timeit(lambda: y4k.copy_(x4k), a.rounds)
        rec("yardstick: torch device copy of the 32x4K f32 batch (this box)", ms, mn, el4k * 8, note="box-to-box spread of one binary is 10-15 %: read the rows below against this")
        del y4k
        ms, mn = timeit(lambda: F.gaussian_blur(x4k, [3, 3]), a.rounds)
        rec("metric 3x3 gaussian f32, 32x4K batch (LDS halo tile, default)", ms, mn, el4k * 8)
        from cpu_vision_amd import _lib
        with _lib.tuning_library():  # forced kernels exist in the -DMV_TUNING build only
            os.environ["MV_FORCE_REG3X3"] = "1"
            ms, mn = timeit(lambda: F.gaussian_blur(x4k, [3, 3]), a.rounds)
            os.environ.pop("MV_FORCE_REG3X3")
        rec("metric 3x3 gaussian f32, 32x4K batch (register-window variant)", ms, mn, el4k * 8, note="A/B: LDS halo tile vs register window")
        # single-frame launches rotating over 32 distinct frames (launch + tail effects visible)
        outs = torch.empty_like(x4k[0])

        def rot():
            for i in range(n4k):
                F.gaussian_blur(x4k[i], [3, 3])
        ms, mn = timeit(rot, max(3, a.rounds // 3))
        rec("metric 3x3 gaussian f32, 4K frame per launch (x32, rotating)", ms / n4k, mn / n4k, el4k // n4k * 8, note="per launch incl. Python + allocator")
    if want("cfg2"):
        x = torch.rand((96, 3, 1080, 1920), generator=g, device="cuda")
        ms, mn = timeit(lambda: F.gaussian_blur(x, [3, 3]), a.rounds)
        rec("cfg2 3x3 gaussian f32, 96x1080p batch", ms, mn, x.numel() * 8)

        def rot2():
            for i in range(96):
                F.gaussian_blur(x[i], [3, 3])
        ms, mn = timeit(rot2, max(3, a.rounds // 3))
        rec("cfg2 3x3 gaussian f32, 1080p frame per launch (x96)", ms / 96, mn / 96, x.numel() // 96 * 8, note="launch-bound: 6.2 us at peak")
        # the same 96 per-frame launches captured once in a HIP graph (the C ABI allocates nothing and never syncs)
        outs = torch.empty_like(x)
        from cpu_vision_amd import _lib
        lib = mv.load_library()
        tp = F._host_taps(3, 0.8)[1]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                sp = torch.cuda.current_stream().cuda_stream
                for i in range(96):
                    _lib.check(lib.mv_gaussian_blur_f32(x[i].data_ptr(), outs[i].data_ptr(), 3, 1080, 1920, tp, 3, tp, 3, sp))
        torch.cuda.current_stream().wait_stream(side)
        ms, mn = timeit(lambda: graph.replay(), a.rounds)
        assert torch.equal(outs[5], F.gaussian_blur(x[5], [3, 3]))
        rec("cfg2 3x3 gaussian f32, 1080p frame per launch (x96, HIP graph replay)", ms / 96, mn / 96, x.numel() // 96 * 8, note="per kernel node")
        del outs
        del x
    if want("u8"):
        xu = torch.randint(0, 256, (n4k, 3, 2160, 3840), generator=g, device="cuda", dtype=torch.uint8)
        ms, mn = timeit(lambda: F.gaussian_blur(xu, [3, 3]), a.rounds)
        rec("u8 3x3 gaussian, 32x4K uint8 batch (2-D pass)", ms, mn, xu.numel() * 2)
        # default = the reference's integers (the bits of its single 2-D pass; from 25 taps up through pair + tie check + fix-up);
        # opt-in = fp32 separable pair + round
        for ks in ([5, 5], [7, 7], [9, 9]):
            F.INTEGER_BLUR_EXACT_2D = True
            ms, mn = timeit(lambda: F.gaussian_blur(xu, ks), a.rounds)
            rec(f"u8 {ks[0]}x{ks[1]} gaussian, 32x4K uint8: default (exact: the 2-D chain's integers)", ms, mn, xu.numel() * 2, flops=2.0 * ks[0] * ks[1] * xu.numel())
            F.INTEGER_BLUR_EXACT_2D = False
            ms, mn = timeit(lambda: F.gaussian_blur(xu, ks), a.rounds)
            rec(f"u8 {ks[0]}x{ks[1]} gaussian, 32x4K uint8: separable pair + round (opt-in, ~1e-5 of pixels +-1)", ms, mn, xu.numel() * 2, flops=2.0 * sum(ks) * xu.numel())
        ms, mn = timeit(lambda: F.gaussian_blur(xu, [23, 23]), a.rounds)
        rec("u8 23x23 gaussian, 32x4K uint8: separable pair + round (opt-in)", ms, mn, xu.numel() * 2, flops=2.0 * 46 * xu.numel())
        F.INTEGER_BLUR_EXACT_2D = True
        ms, mn = timeit(lambda: F.gaussian_blur(xu[:4], [23, 23]), 3)
        rec("u8 23x23 gaussian, 4x4K uint8: default (exact)", ms, mn, xu[:4].numel() * 2, flops=2.0 * 529 * xu[:4].numel())
        del xu
    if want("cfg3"):
        ms, mn = timeit(lambda: F.gaussian_sobel(x4k, [5, 5], [1.1, 1.1]), a.rounds)
        rec("cfg3 separable 5x5 -> sobel fused, 32x4K", ms, mn, el4k * 12, note="36 B/pixel fused minimum")
        ms, mn = timeit(lambda: F.separable_gaussian_blur(x4k, [5, 5], [1.1, 1.1]), a.rounds)
        rec("separable 5x5 blur only, 32x4K", ms, mn, el4k * 8)
        ms, mn = timeit(lambda: F.gaussian_blur(x4k, [5, 5], [1.1, 1.1]), a.rounds)
        rec("direct 2-D 5x5 gaussian (LDS tile), 32x4K", ms, mn, el4k * 8)
        ms, mn = timeit(lambda: F.sobel(x4k), a.rounds)
        rec("sobel pair (dw3x3, 2 outputs), 32x4K", ms, mn, el4k * 12)
        ms, mn = timeit(lambda: F.gaussian_blur(x4k[:8], [23, 23], [3.0, 3.0]), a.rounds)
        rec("gaussian 23x23 (separable path), 8x4K", ms, mn, el4k // 4 * 8)
    if want("sharp"):
        ms, mn = timeit(lambda: F.adjust_sharpness(x4k, 1.5), a.rounds)
        rec("adjust_sharpness f32, 32x4K", ms, mn, el4k * 8)
        xu = (x4k * 255).to(torch.uint8)
        ms, mn = timeit(lambda: F.adjust_sharpness(xu, 1.5), a.rounds)
        rec("adjust_sharpness u8, 32x4K", ms, mn, el4k * 2)
        ms, mn = timeit(lambda: F.gaussian_blur(xu, [3, 3]), a.rounds)
        rec("gaussian 3x3 u8, 32x4K", ms, mn, el4k * 2)
        ms, mn = timeit(lambda: F.gaussian_blur(xu, [5, 5]), a.rounds)
        rec("gaussian 5x5 u8 (default, exact), 32x4K", ms, mn, el4k * 2)
        del xu
    del x4k
    if want("cfg4"):
        n = 256
        x = torch.rand((n, 3, 224, 224), generator=g, device="cuda")
        w = torch.randn((64, 3, 3, 3), generator=g, device="cuda") * (2.0 / 576) ** 0.5
        b = torch.zeros(64, device="cuda")
        out = torch.empty((n, 64, 224, 224), device="cuda")
        ms, mn = timeit(lambda: F.conv2d_bias_relu(x, w, b, out=out), a.rounds)
        nbytes = x.numel() * 4 + out.numel() * 4 + w.numel() * 4
        rec("cfg4 conv3x3x64+bias+relu, 256x3x224x224 (MFMA)", ms, mn, nbytes, 2.0 * n * 64 * 224 * 224 * 27)
        ms, mn = timeit(lambda: out.copy_(out.roll(1, 0)) if False else out.zero_(), a.rounds)
        rec("  yardstick: memset of the 3.29 GB output", ms, mn, out.numel() * 4)
    if want("preset"):
        n = 256
        xu = torch.randint(0, 256, (n, 3, 224, 224), generator=g, device="cuda", dtype=torch.uint8)
        mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
        ms, mn = timeit(lambda: F.to_float_normalize(xu, mean, std), a.rounds)
        rec("preset tail: uint8 -> float + normalize, 256x3x224x224", ms, mn, xu.numel() * 5)
        w = torch.randn((64, 3, 3, 3), generator=g, device="cuda") * (2.0 / 576) ** 0.5
        b = torch.zeros(64, device="cuda")
        ms2, mn2 = timeit(lambda: F.conv2d_bias_relu(F.to_float_normalize(xu, mean, std), w, b), a.rounds)
        nb2 = xu.numel() * 5 + xu.numel() * 4 + n * 64 * 224 * 224 * 4
        rec("preset tail then conv3x3x64+relu (two launches)", ms2, mn2, nb2)
        ms3, mn3 = timeit(lambda: F.normalized_conv2d_bias_relu(xu, mean, std, w, b), a.rounds)
        rec("preset tail FUSED into conv3x3x64+relu (one launch)", ms3, mn3, xu.numel() + n * 64 * 224 * 224 * 4, 2.0 * n * 64 * 224 * 224 * 27)
    if want("presethead"):
        from cpu_vision_amd.presets import ImageClassification
        pre = ImageClassification(crop_size=224)
        for n, hh, ww in ((64, 375, 500), (24, 1080, 1920), (8, 2160, 3840)):
            xu = torch.randint(0, 256, (n, 3, hh, ww), generator=g, device="cuda", dtype=torch.uint8)
            ms, mn = timeit(lambda: pre(xu), a.rounds)
            # algorithmic bytes: the input rows/columns the crop window needs, read once (u8) + the fp32 output
            oh, ow = (256, int(256 * ww / hh))
            frac = (224 / oh) * (224 / ow)
            rec(f"ImageClassification(256/224) on {n}x3x{hh}x{ww} uint8 (one kernel up to scale 7, else two)", ms, mn, int(xu.numel() * frac) + n * 3 * 224 * 224 * 4,
                note=f"{n / ms * 1e3:.0f} img/s")

            def one():
                for i in range(n):
                    pre(xu[i])
            ms1, mn1 = timeit(one, max(3, a.rounds // 3))
            rec(f"  the same, one {hh}x{ww} image per call", ms1 / n, mn1 / n, int(xu.numel() * frac) // n + 3 * 224 * 224 * 4, note=f"{n / ms1 * 1e3:.0f} img/s")
            del xu
    Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "perf_configs.json").write_text(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
