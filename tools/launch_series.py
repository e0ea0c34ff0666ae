#!/usr/bin/env python3
"""Why do launch times of one kernel range 606-809 us inside one run?

For cfg4 (k_conv3x3_c3), cfg3 (k_sepfast<5,sobel>), cfg2 and the 32-frame headline kernel: a series of back-to-back
launches after the GPU sat idle (the state every leg of bench.py / every rocprofv3 run starts in: the host was comparing
frames with the oracle or allocating), HIP events per launch, and between launches a 20 us probe kernel that reports the
shader clock (shader cycles per 100 MHz reference tick, tools/micro/clock_probe.hip).  Then the same series again without
the idle gap.  Prints, per phase, time and clock at the start and at the end of the series and their correlation.
"""
import argparse
import ctypes as C
import json
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import cpu_vision_amd as mv  # noqa: E402
from cpu_vision_amd import _lib, functional as F  # noqa: E402


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        card = next(iter(d.values()))
        keep = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "junction", "memory)"))}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"rocm-smi": f"unavailable ({type(e).__name__})"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=120)
    ap.add_argument("--idle", type=float, default=2.0)
    a = ap.parse_args()
    lib = mv.load_library()
    probe = C.CDLL(str(ROOT / "tools" / "micro" / "libclock_probe.so"))
    probe.clock_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    sp = stream.cuda_stream
    g = torch.Generator(device=dev).manual_seed(1)
    print("rocm-smi at start:", smi(), flush=True)

    def series(launch, n, with_probe=True):
        pb = torch.zeros((n, 2), dtype=torch.int64, device=dev)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n)]
        for i in range(n):
            ev[2 * i].record(stream)
            launch()
            ev[2 * i + 1].record(stream)
            if with_probe:
                probe.clock_probe(pb[i].data_ptr(), 2000, sp)  # 20 us at 100 MHz
        stream.synchronize()
        ms = [ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(n)]
        p = pb.cpu().double()
        mhz = (p[:, 0] / p[:, 1].clamp(min=1) * 100.0).tolist() if with_probe else [0.0] * n
        return ms, mhz

    def report(tag, ms, mhz):
        n = len(ms)
        t, m = torch.tensor(ms, dtype=torch.float64), torch.tensor(mhz, dtype=torch.float64)
        inv = 1.0 / m.clamp(min=1)
        corr = float(torch.corrcoef(torch.stack([t, inv]))[0, 1]) if m.max() > 0 and t.std() > 0 and inv.std() > 0 else float("nan")
        k = min(5, n)
        print(f"  {tag:34s} first {k}: {sum(ms[:k]) / k:7.4f} ms @ {sum(mhz[:k]) / k:6.0f} MHz   last {k}: {sum(ms[-k:]) / k:7.4f} ms @ {sum(mhz[-k:]) / k:6.0f} MHz"
              f"   min {min(ms):.4f} max {max(ms):.4f}   corr(ms, 1/MHz) = {corr:+.3f}", flush=True)
        # time x clock: constant if the kernel is bound by the shader clock, ~ time if bound by memory
        step = max(1, n // 12)
        print("     ms      :", " ".join(f"{v:7.4f}" for v in ms[::step]))
        print("     MHz     :", " ".join(f"{v:7.0f}" for v in mhz[::step]))
        return {"tag": tag, "ms": [round(v, 4) for v in ms], "mhz": [round(v) for v in mhz]}

    out = {}

    def study(name, launch, gb):
        print(f"== {name}  ({gb:.2f} GB algorithmic per launch)", flush=True)
        rec = []
        for _ in range(3):
            launch()
        stream.synchronize()
        time.sleep(a.idle)
        ms, mhz = series(launch, a.launches)
        rec.append(report(f"after {a.idle:.0f} s idle", ms, mhz))
        ms, mhz = series(launch, a.launches // 2)
        rec.append(report("immediately again (no idle)", ms, mhz))
        time.sleep(a.idle)
        ms, mhz = series(launch, a.launches // 2, with_probe=False)
        rec.append(report(f"after {a.idle:.0f} s idle, no probe kernel", ms, [0.0] * len(ms)))
        print("  rocm-smi right after:", smi(), flush=True)
        out[name] = rec

    # cfg4
    n, hh, cout = 256, 224, 64
    x = torch.empty((n, 3, hh, hh), device=dev).uniform_(0, 1, generator=g)
    wt = torch.empty((cout, 3, 3, 3), device=dev).normal_(0, (2.0 / (cout * 9)) ** 0.5, generator=g)
    b = torch.empty((cout,), device=dev).uniform_(-0.1, 0.1, generator=g)
    y = torch.empty((n, cout, hh, hh), device=dev)
    study("cfg4 k_conv3x3_c3", lambda: _lib.check(lib.mv_conv3x3_bias_relu_f32(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), n, 3, hh, hh, cout, 1, sp)),
          (x.numel() + y.numel()) * 4 / 1e9)
    study("memset of cfg4's 3.29 GB output (hipMemsetAsync via torch)", lambda: y.zero_(), y.numel() * 4 / 1e9)
    del x, y
    # cfg3 / headline kernel on 32 x 4K
    x = torch.empty((32, 3, 2160, 3840), device=dev).uniform_(0, 1, generator=g)
    gx, gy = torch.empty_like(x), torch.empty_like(x)
    t5 = _lib.taps_from_tensor(F._get_gaussian_kernel1d(5, 1.1))
    study("cfg3 k_sepfast<5,sobel>", lambda: _lib.check(lib.mv_gaussian_sobel_f32(x.data_ptr(), gx.data_ptr(), gy.data_ptr(), 96, 2160, 3840, t5, 5, t5, 5, sp)),
          x.numel() * 12 / 1e9)
    t3 = _lib.taps_from_tensor(F._get_gaussian_kernel1d(3, 0.8))
    study("3x3 gaussian k_dwtile, 32 x 4K", lambda: _lib.check(lib.mv_gaussian_blur_f32(x.data_ptr(), gx.data_ptr(), 96, 2160, 3840, t3, 3, t3, 3, sp)),
          x.numel() * 8 / 1e9)
    study("torch copy_ of the 32 x 4K batch", lambda: gx.copy_(x), x.numel() * 8 / 1e9)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "launch_series.json").write_text(json.dumps(out))


if __name__ == "__main__":
    main()
