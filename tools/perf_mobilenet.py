#!/usr/bin/env python3
"""Per-kernel time of MobileNetV2 on one MI355X: every fused conv -> norm -> activation launch with its
algorithmic bytes and flops, grouped by kernel kind, and the whole forward in img/s."""
import argparse
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cpu_vision_amd import mobilenet as M  # noqa: E402
from tools.perf_vgg import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    n = a.batch
    torch.manual_seed(0)
    net = M.MobileNetV2(1000).cuda()
    x = torch.rand((n, 3, 224, 224), device="cuda")
    calls = []
    orig = M.fused_conv_block

    def spy(xx, conv, norm, act, cache, residual=None):
        y = orig(xx, conv, norm, act, cache, residual)
        calls.append((xx, conv, norm, act, cache, residual, y))
        return y
    M.fused_conv_block = spy
    net(x)
    M.fused_conv_block = orig
    kinds = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
    total = 0.0
    for xx, conv, norm, act, cache, residual, y in calls:
        ms = timeit(lambda: orig(xx, conv, norm, act, cache, residual), 5)
        k = conv.kernel_size[0]
        kind = "depthwise3x3" if conv.groups > 1 else ("pointwise1x1" if k == 1 else "stem3x3")
        nbytes = (xx.numel() + y.numel() + (0 if residual is None else residual.numel())) * 4
        flop = 2.0 * y.numel() * (conv.in_channels // conv.groups) * k * k
        kinds[kind][0] += ms
        kinds[kind][1] += nbytes
        kinds[kind][2] += flop
        kinds[kind][3] += 1
        total += ms
        if a.verbose:
            print(f"{kind:13s} {conv.in_channels:4d}->{conv.out_channels:4d} s{conv.stride[0]} @{y.shape[-1]:3d}  {ms:7.3f} ms  {nbytes / ms / 1e6:7.1f} GB/s  {flop / ms / 1e9:6.2f} TF", flush=True)
    for kind, (ms, nbytes, flop, cnt) in kinds.items():
        print(f"{kind:13s} x{cnt:2d}: {ms:7.3f} ms  {nbytes / ms / 1e6:7.1f} GB/s ({nbytes / ms / 1e6 / 80:4.1f}% HBM)  {flop / ms / 1e9:6.2f} TFLOP/s", flush=True)
    ms_all = timeit(lambda: net(x), 5)
    print(f"batch {n}: sum of conv kernels {total:.3f} ms; whole forward {ms_all:.3f} ms = {n / ms_all * 1e3:.0f} img/s", flush=True)


if __name__ == "__main__":
    main()
