#!/usr/bin/env python3
"""Per-layer throughput of VGG-11's feature extractor on one MI355X: fp32 MFMA TFLOP/s of every
conv layer (roof 157.3), GB/s of the pooling layers (roof 8 TB/s), and the whole `features` forward."""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402
from cpu_vision_amd.nn import VGGFeatures, Conv3x3ReLU, MaxPool2x2  # noqa: E402


def timeit(fn, rounds=7):
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
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    n = a.batch
    feats = VGGFeatures("A").cuda()
    x = torch.rand((n, 3, 224, 224), device="cuda")
    rows, total_ms, total_flop = [], 0.0, 0.0
    cur = x
    for layer in feats.layers:
        inp = cur
        ms = timeit(lambda: layer(inp))
        cur = layer(inp)
        if isinstance(layer, Conv3x3ReLU):
            flop = 2.0 * n * layer.out_channels * cur.shape[-1] * cur.shape[-2] * layer.in_channels * 9
            tf = flop / (ms * 1e-3) / 1e12
            nbytes = (inp.numel() + cur.numel() + layer.weight.numel()) * 4
            print(f"conv {layer.in_channels:3d}->{layer.out_channels:3d} @{cur.shape[-1]:3d}  {ms:8.3f} ms  {tf:6.1f} TF ({tf / 157.3 * 100:4.1f}% MFMA)  {nbytes / ms / 1e6:7.1f} GB/s", flush=True)
            rows.append({"layer": f"conv{layer.in_channels}-{layer.out_channels}@{cur.shape[-1]}", "ms": ms, "TFLOPs": tf, "mfma_frac": tf / 157.3})
            total_flop += flop
        else:
            nbytes = (inp.numel() + cur.numel()) * 4
            print(f"maxpool        @{cur.shape[-1]:3d}  {ms:8.3f} ms  {nbytes / ms / 1e6:7.1f} GB/s ({nbytes / ms / 1e6 / 80:4.1f}% HBM)", flush=True)
            rows.append({"layer": f"maxpool@{cur.shape[-1]}", "ms": ms, "GBps": nbytes / ms / 1e6})
        total_ms += ms
    from cpu_vision_amd.nn import vgg11
    net = vgg11(1000).cuda().eval()
    f = feats(x).reshape(n, -1)
    from cpu_vision_amd import functional as F
    linears = [m for m in net.classifier if isinstance(m, torch.nn.Linear)]
    for i, layer in enumerate(linears):
        last = i == len(linears) - 1
        run = lambda: F.linear_bias_relu(f, layer.weight, layer.bias, relu=not last)  # noqa: E731 (Linear + ReLU is one launch in the net)
        ms = timeit(run)
        flop = 2.0 * n * layer.in_features * layer.out_features
        wbytes = layer.weight.numel() * 4
        print(f"linear {layer.in_features:5d}->{layer.out_features:4d}      {ms:8.3f} ms  {flop / ms / 1e9:6.1f} TF  weights {wbytes / ms / 1e6:7.1f} GB/s", flush=True)
        rows.append({"layer": f"linear{layer.in_features}-{layer.out_features}", "ms": ms, "TFLOPs": flop / ms / 1e9})
        f = run()
    with torch.no_grad():
        whole_net = timeit(lambda: net(x))
    print(f"vgg11 whole forward, batch {n}: {whole_net:.3f} ms ({n / whole_net * 1e3:.0f} img/s)")
    whole = timeit(lambda: feats(x))
    print(f"features forward, batch {n}: {whole:.3f} ms ({n / whole * 1e3:.0f} img/s), conv flops {total_flop / 1e9:.1f} GFLOP -> {total_flop / (whole * 1e-3) / 1e12:.1f} TF")
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "perf_vgg.json").write_text(json.dumps({"batch": n, "layers": rows, "features_ms": whole}, indent=1))


if __name__ == "__main__":
    main()
