#!/usr/bin/env python3
"""Classifier layers (models/vgg.py:42-50, alexnet.py) on one MI355X: single ascending-k chain against the sliced-K pass
(mv_linear_bias_relu_ws_f32), in GB/s of weights streamed (the bound at inference-size batches) and TFLOP/s."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
# knob sweeps run on the -DMV_TUNING build of the same sources (the product library reads no environment variable)
os.environ.setdefault("MI355VISION_LIB", str(Path(__file__).resolve().parent.parent / "cpu-vision_amd" / "lib" / "libmi355vision_tuning.so"))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402
from tools.perf_configs import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for k, m in ((25088, 4096), (9216, 4096), (4096, 4096), (4096, 1000), (1280, 1000)):
    w = (torch.rand((m, k), generator=g, device="cuda") - 0.5) * 0.02
    b = torch.rand((m,), generator=g, device="cuda")
    for n in (1, 4, 8, 64, 256, 1024):
        x = torch.rand((n, k), generator=g, device="cuda")
        one, _ = timeit(lambda: F.linear_bias_relu(x, w, b, relu=True, sliced_k=False), 7)
        two, _ = timeit(lambda: F.linear_bias_relu(x, w, b, relu=True), 7)
        alts = []
        for v in ("0", "1"):  # W staging order forced: float4-of-a-row fastest / row fastest over the lanes
            os.environ["MV_LINEAR_ROWFAST"] = v
            alts.append(timeit(lambda: F.linear_bias_relu(x, w, b, relu=True), 7)[0])
        os.environ.pop("MV_LINEAR_ROWFAST")
        nog = None
        if n <= 4:  # batch <= 4 runs as v_fma chains (GEMV); MV_LINEAR_GEMV=0 forces the MFMA tiles
            os.environ["MV_LINEAR_GEMV"] = "0"
            nog = timeit(lambda: F.linear_bias_relu(x, w, b, relu=True), 7)[0]
            os.environ.pop("MV_LINEAR_GEMV")
        s, sl = F.linear_k_slices(n, k, m)
        wb = w.numel() * 4
        fl = 2.0 * n * k * m
        print(f"linear {k:5d}->{m:4d} batch {n:4d}: single chain {one * 1e3:8.1f} us ({wb / one / 1e6:6.0f} GB/s, {fl / one / 1e9:6.1f} TF)   "
              f"{s:2d} slices of {sl:5d}: {two * 1e3:8.1f} us ({wb / two / 1e6:6.0f} GB/s, {fl / two / 1e9:6.1f} TF)   "
              f"forced W staging order q-fast / row-fast: {alts[0] * 1e3:8.1f} / {alts[1] * 1e3:8.1f} us"
              + (f"   MFMA tiles instead of fma chains: {nog * 1e3:8.1f} us" if nog is not None else ""), flush=True)
