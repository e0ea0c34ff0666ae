#!/usr/bin/env python3
"""Interleaved A/B of one MV_* environment knob of the tuning build (libmi355vision_tuning.so) on a BASELINE config (GPU box).

    python tools/sweep_env.py --op sobel5 MV_SEPFAST_ROWS 16 32 48 64 96
    python tools/sweep_env.py --op sep5   MV_SEPFAST_ROWS 8 16 32

Every value is run in interleaved rounds in ONE process (CDNA guide, methodology rule 24); the first value's output is the
reference every other value must reproduce bit for bit.
"""
import argparse
import os
import statistics
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cpu_vision_amd import _lib  # noqa: E402
from cpu_vision_amd import functional as F  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="sobel5", choices=["sobel5", "sep5", "blur3", "blur23", "conv"])
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("knob")
    ap.add_argument("values", nargs="+")
    a = ap.parse_args()
    if a.op == "conv":
        x = torch.rand((256, 3, 224, 224), device="cuda")
        wt, b = torch.randn((64, 3, 3, 3), device="cuda") * 0.06, torch.rand(64, device="cuda") - 0.5
        alg = x.numel() * 4 * (1 + 64 / 3)
        fn = lambda: F.conv2d_bias_relu(x, wt, b)  # noqa: E731
    else:
        x = torch.rand((a.frames, 3, 2160, 3840), device="cuda")
        alg = x.numel() * (12 if a.op == "sobel5" else 8)
        if a.op == "sobel5":
            fn = lambda: F.gaussian_sobel(x, [5, 5], [1.1, 1.1])[0]  # noqa: E731
        elif a.op == "sep5":
            fn = lambda: F.separable_blur(x, [5, 5], [1.1, 1.1]) if hasattr(F, "separable_blur") else F.gaussian_blur(x, [5, 5], [1.1, 1.1])  # noqa: E731
        elif a.op == "blur23":
            fn = lambda: F.gaussian_blur(x, [23, 23], [3.5, 3.5])  # noqa: E731
        else:
            fn = lambda: F.gaussian_blur(x, [3, 3])  # noqa: E731
    times = {v: [] for v in a.values}
    ref = None
    with _lib.tuning_library():
        for v in a.values:
            os.environ[a.knob] = v
            out = fn()
            torch.cuda.synchronize()
            print(f"{a.knob}={v}: {_lib.last_kernel()}")
            if ref is None:
                ref = out[:2].clone()
            else:
                assert torch.equal(ref, out[:2]), f"{a.knob}={v} changes the result"
            del out
        for _ in range(a.rounds):
            for v in a.values:
                os.environ[a.knob] = v
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                out = fn()
                e1.record()
                torch.cuda.synchronize()
                del out
                times[v].append(e0.elapsed_time(e1))
    os.environ.pop(a.knob, None)
    for v, ts in times.items():
        med = statistics.median(ts)
        print(f"{a.knob}={v:>6s}  median {med:8.4f} ms  min {min(ts):8.4f}  {alg / med / 1e6:8.1f} GB/s  {alg / med / 1e6 / 80:5.1f} % HBM")


if __name__ == "__main__":
    main()
