#!/usr/bin/env python3
"""Every InvertedResidual block of MobileNetV2 (width 1.0, 224 x 224) as ONE fused kernel (csrc/invres.hip) against the three
stand-alone launches, interleaved in one process; then the whole forward, eager and as a replayed HIP graph, both ways.

    python tools/perf_invres.py [--batch 64] [--rounds 7]
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F, graphs, mobilenet as M  # noqa: E402


def timeit(fn, rounds, inner=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    ts.sort()
    return ts[len(ts) // 2]


def graph_time(fn, rounds, inner=10):
    """GPU time per call of fn with the host out of the picture: `inner` calls captured into one HIP graph, replayed."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner):
            fn()
    return timeit(g.replay, rounds, inner=1) / inner


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    n = a.batch
    torch.manual_seed(0)
    net = M.MobileNetV2(1000).cuda()
    x = torch.rand((n, 3, 224, 224), device="cuda")
    with torch.no_grad():
        # spin the clocks up (bench.py: launch times follow the shader clock's dip after idle)
        for _ in range(30):
            net(x)
        acts = [x]
        for layer in net.features:
            acts.append(layer(acts[-1]))
        tot_f = tot_u = 0.0
        print(f"batch {n}: per InvertedResidual block, fused kernel vs three launches (GPU time: 10 calls captured in a HIP graph, median of {a.rounds} replays)")
        for i, layer in enumerate(net.features):
            if not isinstance(layer, M.InvertedResidual):
                continue
            inp = acts[i]
            M.FUSE_INVERTED_RESIDUAL = True
            plan = layer._fused_plan(inp)
            M.FUSE_INVERTED_RESIDUAL = False
            t_u = graph_time(lambda: layer(inp), a.rounds)
            t_f = None
            if plan is not None:
                M.FUSE_INVERTED_RESIDUAL = True
                t_f = graph_time(lambda: layer(inp), a.rounds)
                kern = _lib.last_kernel()
            cin, hid, cout = layer.conv[0][0].in_channels, layer.conv[-2].in_channels, layer.out_channels
            flop = 2.0 * n * (inp.shape[-1] ** 2 * cin * hid * (len(layer.conv) == 4) + acts[i + 1].shape[-1] ** 2 * hid * (9 + cout))
            line = f"features[{i:2d}] {cin:4d}->{hid:4d}->{cout:4d} @{inp.shape[-1]:3d} s{layer.stride}  three launches {t_u * 1e3:7.1f} us"
            if t_f is not None:
                line += f"   fused {t_f * 1e3:7.1f} us ({flop / t_f / 1e9:5.1f} TF, plan {plan})  x{t_u / t_f:4.2f}   {kern}"
                tot_f += t_f
                tot_u += t_u
            print(line, flush=True)
        print(f"fused blocks: {tot_u * 1e3:.0f} us as three launches each -> {tot_f * 1e3:.0f} us fused")
        for fuse in (False, True):
            M.FUSE_INVERTED_RESIDUAL = fuse
            eager = timeit(lambda: net(x), a.rounds, inner=3)
            g = graphs.capture(net, x)
            replay = timeit(lambda: g.graph.replay(), a.rounds, inner=3)
            print(f"whole forward, batch {n}, FUSE_INVERTED_RESIDUAL={fuse}: eager {eager:.3f} ms = {n / eager * 1e3:.0f} img/s; "
                  f"HIP graph replay {replay:.3f} ms = {n / replay * 1e3:.0f} img/s", flush=True)


if __name__ == "__main__":
    main()
