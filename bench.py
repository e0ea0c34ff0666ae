#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the 3x3 depthwise conv2d (Gaussian 3x3, sigma 0.8, reflect border -- the
call gaussian_blur_image makes) over 4K fp32 3-channel frames, image-sharded over N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]          # N > 1: spawns its own N ranks (see below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                     # or is launched as one rank of N

Launch convention (the reference's: references/classification/utils.py:245-269 -- RANK / WORLD_SIZE / LOCAL_RANK from
the environment, init_method env://, backend nccl).  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the
environment is the LAUNCHER: before anything touches the GPU (it never imports torch) it starts
`python -m torch.distributed.run ... bench.py <same arguments>` as a child process, which starts the N ranks as fresh
processes; the launcher forwards rank 0's single JSON line and exits with the child's code.  No process that has
initialised HIP is ever re-exec'ed.

Workload (BASELINE.json metric / cfg5): every rank owns 128 distinct frames of 3x2160x3840 fp32 resident in its
HBM (12.7 GB in, 12.7 GB out; 1024 frames at N=8 -- weak scaling).  One step = one pass of the hot path over the
rank's whole shard = ONE kernel launch; there is no collective in the data path (frames are independent), RCCL
is used for the barriers around the timed region, the max-over-ranks of the elapsed time and the (untimed) gather of
per-rank statistics.  25.5 GB of distinct data per step per GPU, so the 256 MB Infinity Cache cannot masquerade as HBM.

The JSON line also carries
  roofline     achieved algorithmic GB/s of the kernel (8 B per element: one fp32 read + one fp32 write) from HIP
               events around every timed launch on the launch stream, against the 8 TB/s HBM3E peak; the kernel name
               comes from the library's launcher (mv_last_kernel), the PMC traffic from profiles/traffic_latest.json
               (stamped with the library build it was measured on);
  per_rank     for every rank: average / minimum launch time, achieved GB/s, the card it ran on (name, PCI bus id) and a
               bit-for-bit check of the first and the last frame of its shard against the C oracle (untimed);
  configs      (N = 1, untimed for `value`) the other single-GPU BASELINE configs -- cfg2 96 x 1080p 3x3 Gaussian, cfg3 32 x 4K
               separable 5x5 Gaussian -> Sobel, cfg4 256 x 3x224x224 Conv2d(3,64,3,p=1)+bias+ReLU -- each one launch per batch:
               kernel name from the launcher, HIP-event avg / min / max / series over --config-launches launches,
               algorithmic bytes, fraction of the HBM peak (cfg4: also of the fp32 MFMA peak), bit-exact flag vs the oracle;
  cpu_baseline (N = 1) the reference's CPU path (pad(reflect) + conv2d(groups=C) through torch CPU ops, the exact call
               sequence of gaussian_blur_image) timed on this box's host cores on a bounded sample; `legs`: the same on
               one thread, and cfg4's Conv2d+ReLU on the fastest pool size and on one thread.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

H, W, C = 2160, 3840, 3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a float4 copy achieves
BYTES_PER_ELEMENT = 8  # algorithmic: 4 B read + 4 B written per element (24 B per 3-channel pixel)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--frames-per-gpu", type=int, default=128, help="cfg5: 1024 frames over 8 GPUs")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-configs", action="store_true", help="skip the cfg2 / cfg3 / cfg4 legs (N = 1 runs them by default)")
    p.add_argument("--config-launches", type=int, default=30, help="timed launches per cfg2 / cfg3 / cfg4 leg")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo + --all-ranks-on-gpu0 rehearses the N>1 control flow on a one-GPU box")
    p.add_argument("--all-ranks-on-gpu0", action="store_true", help="rehearsal only: every rank uses cuda:0")
    p.add_argument("--control-plane-only", action="store_true",
                   help="rehearsal of the launcher and of the distributed control flow on a box WITHOUT a GPU: ranks "
                        "rendezvous, barrier, reduce and gather exactly as in a real run but launch no kernel; the line "
                        "carries value = null (nothing was measured)")
    p.add_argument("--force-dist", action="store_true",
                   help="initialise the process group even for a single rank (exercises the RCCL calls on a one-GPU box)")
    p.add_argument("--launch-timeout", type=float, default=1500.0, help="launcher: seconds before the child is killed")
    return p.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher (no torch)
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _signal_group(pgid: int, sig: int) -> None:
    try:
        os.killpg(pgid, sig)
    except ProcessLookupError:  # the child exited between the timeout and the signal
        pass


def launch_ranks(a, argv) -> int:
    """`python bench.py --gpus N` (N > 1, not under torchrun): start the N ranks as children of a torch.distributed.run
    child.  This process never initialises HIP (it does not even import torch), so nothing that holds a GPU context is
    ever exec'ed or forked.  Forwards rank 0's JSON line; returns the child's exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve()), *argv]
    env = dict(os.environ, MV_BENCH_LAUNCHER="self", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=a.launch_timeout)
    except subprocess.TimeoutExpired:
        _signal_group(proc.pid, 15)  # the exact process group this launcher started
        try:
            out, _ = proc.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            _signal_group(proc.pid, 9)
            out, _ = proc.communicate()
        sys.stderr.write(f"bench.py launcher: ranks did not finish within {a.launch_timeout:.0f} s\n")
        return 124
    line = None
    for ln in out.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line, flush=True)
    if proc.returncode == 0 and line is None:
        sys.stderr.write("bench.py launcher: the ranks exited 0 without printing a result line\n")
        return 1
    return proc.returncode


# ------------------------------------------------------------------------------------------------ one rank
def _cgroup_cpus():
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        return max(1, int(int(quota) / int(period))) if quota != "max" else None
    except Exception:
        return None


def frames_vs_oracle(x, y, idxs):
    """Bit-for-bit comparison of the given frames of the rank's shard with the C oracle (the checker -- never the
    product path).  Returns {frame: bool}."""
    import torch
    from cpu_vision_amd import functional as F
    from oracle import ref

    k = F._get_gaussian_kernel1d(3, 0.8).numpy()
    out = {}
    for i in idxs:
        want = torch.from_numpy(ref.gaussian_blur(x[i].cpu().numpy(), k, k))
        out[int(i)] = bool(torch.equal(y[i].cpu(), want))
    return out


def cpu_baseline(x_frame, y_frame, budget_s: float):
    """Rank 0, N=1 only.  Times the reference's CPU call sequence on ONE 4K frame repeatedly (bounded sample) and
    compares the GPU result for that frame with it (1e-5 relative) -- the oracle comparison is frames_vs_oracle."""
    import torch
    from oracle import ref, ref_torch  # the checker / the baseline -- never the product path
    from cpu_vision_amd import functional as F

    xf = x_frame.cpu()
    ks, sg = [3, 3], [0.8, 0.8]
    # thread count: torch's default is every logical CPU of the host (256 here) although the box's cgroup
    # grants ~16 CPUs; oversubscription makes the reference look slower than it is, so try a few pool sizes
    # briefly and keep the fastest for the timed sample
    default_threads = torch.get_num_threads()
    quota = _cgroup_cpus() or default_threads
    best = None
    for cand in sorted({1, min(8, quota), quota, min(2 * quota, default_threads), default_threads}):
        torch.set_num_threads(cand)
        ref_torch.gaussian_blur_image(xf, ks, sg)
        t0 = time.perf_counter()
        for _ in range(2):
            ref_torch.gaussian_blur_image(xf, ks, sg)
        dt = (time.perf_counter() - t0) / 2
        if best is None or dt < best[1]:
            best = (cand, dt)
    threads = best[0]
    torch.set_num_threads(threads)
    times = []
    t_end = time.perf_counter() + budget_s
    while (time.perf_counter() < t_end and len(times) < 60) or len(times) < 5:
        t0 = time.perf_counter()
        y_cpu = ref_torch.gaussian_blur_image(xf, ks, sg)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    mpix = H * W / 1e6
    yg = y_frame.cpu()
    err = (yg - y_cpu).abs()
    tol = 1e-5 * y_cpu.abs() + 1e-6
    # the plain-C restatement, for reference (OpenMP over planes: 3 planes -> 3 threads busy)
    k = F._get_gaussian_kernel1d(3, 0.8).numpy()
    t0 = time.perf_counter()
    ref.gaussian_blur(xf.numpy(), k, k)
    t_c = time.perf_counter() - t0
    return {
        "value": round(mpix / med, 2),
        "unit": "Mpixels/s",
        "cores": threads,
        "kind": "port",
        "sample": f"one 3x{H}x{W} fp32 frame, pad(reflect)+conv2d(groups=3) via torch {torch.__version__} CPU ops "
                  f"(the reference's call sequence), median of {len(times)} runs, {threads} threads (fastest pool size tried; "
                  f"cgroup quota {_cgroup_cpus()} of {os.cpu_count()} host cpus)",
        "ms_per_frame": round(med * 1e3, 2),
        "c_oracle_ms_per_frame": round(t_c * 1e3, 2),
        "gpu_vs_cpu_path_within_1e-5": bool((err <= tol).all()),
        "max_abs_err_vs_cpu_path": float(err.max()),
    }


MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix peak (MI355X_MICROARCH.md)


SPIN_UP_S = 0.25  # untimed GPU work before a timed series: see _spin_up


def _spin_up(launch, stream, seconds: float = SPIN_UP_S, min_launches: int = 3) -> int:
    """Launch until `seconds` of wall time have passed (at least min_launches), then drain the stream.  Why: after the GPU has
    sat idle (the host was generating data or comparing frames with the oracle) the power controller drops the shader clock
    to ~1.8-1.9 GHz under the first ~10-40 ms of a sudden heavy load and recovers to 2.43 GHz afterwards; launch times of the
    MFMA convolution follow 1/sclk with r = +0.97 (k_conv3x3_c3: 0.72-0.85 ms during the dip, 0.63-0.64 ms after it), a pure
    memset does not move (profiles/r03_launch_series.log, tools/launch_series.py).  A count of 3 warm-up launches of a
    0.6-1.8 ms kernel ends inside that dip -- that was round 2's unexplained 606-809 us range.  What is timed after this is
    the steady state a pipeline that keeps the GPU busy sees."""
    t_end, k = time.perf_counter() + seconds, 0
    while k < min_launches or time.perf_counter() < t_end:
        launch()
        k += 1
        if k % 16 == 0:
            stream.synchronize()  # keep the queue short, so that wall time tracks GPU time
    stream.synchronize()
    return k


def _timed_launches(launch, stream, launches: int):
    """HIP events on the launch stream around each of `launches` back-to-back launches (after _spin_up) -> series in ms."""
    import torch

    _spin_up(launch, stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    ev[0].record(stream)
    for i in range(launches):
        launch()
        ev[i + 1].record(stream)
    stream.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(launches)]


def _leg(series, kernel, alg_bytes, flops=0.0):
    avg = sum(series) / len(series)
    srt = sorted(series)
    gbs = alg_bytes / (avg * 1e-3) / 1e9
    out = {"kernel": kernel, "launches": len(series), "avg_launch_ms": round(avg, 4), "min_launch_ms": round(srt[0], 4),
           "max_launch_ms": round(srt[-1], 4), "median_launch_ms": round(srt[len(srt) // 2], 4),
           "algorithmic_bytes_per_launch": int(alg_bytes),
           "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
           "spin_up_s_before_timing": SPIN_UP_S, "launch_ms_series": [round(t, 4) for t in series]}
    if flops:
        tf = flops / (avg * 1e-3) / 1e12
        out["flops_per_launch"] = int(flops)
        out["mfma"] = {"achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}
    return out


def other_configs(dev, launches: int):
    """The other single-GPU BASELINE configs, each ONE launch over a batch resident in HBM (N = 1 only, after the headline's
    timed region): cfg2 (96 x 1080p, 3x3 Gaussian), cfg3 (32 x 4K, separable 5x5 Gaussian -> Sobel pair, fused) and cfg4
    (256 x 3x224x224 -> Conv2d(3,64,3,p=1)+bias+ReLU).  Per leg: the kernel the library's launcher reports, HIP-event launch
    times on the launch stream, algorithmic bytes, fraction of the 8 TB/s HBM peak (cfg4 also of the fp32 MFMA
    peak) and a bit-for-bit comparison of sampled frames with the C oracle (the checker, untimed)."""
    import numpy as np
    import torch
    import cpu_vision_amd as mv
    from cpu_vision_amd import _lib, functional as F
    from oracle import ref

    lib = mv.load_library()
    stream = torch.cuda.current_stream(dev)
    sp = stream.cuda_stream
    out = {}
    g = torch.Generator(device=dev).manual_seed(2000)

    # ---- cfg2: 96 frames of 3x1080x1920, gaussian 3x3 sigma 0.8 (working set 4.8 GB >> the 256 MB Infinity Cache)
    n, h, w = 96, 1080, 1920
    x = torch.empty((n, C, h, w), dtype=torch.float32, device=dev).uniform_(0.0, 1.0, generator=g)
    y = torch.empty_like(x)
    k1 = F._get_gaussian_kernel1d(3, 0.8)
    t3 = _lib.taps_from_tensor(k1)
    series = _timed_launches(lambda: _lib.check(lib.mv_gaussian_blur_f32(x.data_ptr(), y.data_ptr(), n * C, h, w, t3, 3, t3, 3, sp)), stream, launches)
    leg = _leg(series, _lib.last_kernel(), x.numel() * BYTES_PER_ELEMENT)
    k3 = k1.numpy()
    leg["workload"] = f"cfg2: {n} frames of 3x{h}x{w} fp32, 3x3 Gaussian sigma=0.8, reflect border, one launch"
    leg["bit_exact_vs_oracle"] = all(bool(np.array_equal(y[i].cpu().numpy(), ref.gaussian_blur(x[i].cpu().numpy(), k3, k3))) for i in (0, n - 1))
    leg["frames_checked"] = [0, n - 1]
    out["cfg2"] = leg
    del x, y

    # ---- cfg3: 32 frames of 3x2160x3840, separable 5x5 Gaussian (sigma 1.1) -> Sobel (gx, gy): x read once, two outputs written
    n = 32
    x = torch.empty((n, C, H, W), dtype=torch.float32, device=dev).uniform_(0.0, 1.0, generator=g)
    gx, gy = torch.empty_like(x), torch.empty_like(x)
    k5 = F._get_gaussian_kernel1d(5, 1.1)
    t5 = _lib.taps_from_tensor(k5)
    series = _timed_launches(lambda: _lib.check(lib.mv_gaussian_sobel_f32(x.data_ptr(), gx.data_ptr(), gy.data_ptr(), n * C, H, W, t5, 5, t5, 5, sp)),
                             stream, launches)
    leg = _leg(series, _lib.last_kernel(), x.numel() * 12)
    leg["workload"] = f"cfg3: {n} frames of 3x{H}x{W} fp32, separable 5x5 Gaussian sigma=1.1 then Sobel gx, gy (fused; 36 B/pixel), one launch"
    k5n = k5.numpy()
    ok = True
    for i in (n - 1,):
        wx, wy = ref.gaussian_sobel(x[i].cpu().numpy(), k5n, k5n)
        ok = ok and bool(np.array_equal(gx[i].cpu().numpy(), wx)) and bool(np.array_equal(gy[i].cpu().numpy(), wy))
    leg["bit_exact_vs_oracle"] = ok
    leg["frames_checked"] = [n - 1]
    # The run time of this kernel follows the PHYSICAL placement of its two output buffers (six allocations in
    # one process 1.56-1.79 ms, each reproducible to 0.2 %).  The leg's figure is the first allocation, as a caller would get it;
    # two more output pairs (the earlier ones stay allocated, so these land elsewhere) show the band inside this very run.
    extra, keep = [], [(gx, gy)]
    for _ in range(2):
        gx2, gy2 = torch.empty_like(x), torch.empty_like(x)
        keep.append((gx2, gy2))
        ser = _timed_launches(lambda: _lib.check(lib.mv_gaussian_sobel_f32(x.data_ptr(), gx2.data_ptr(), gy2.data_ptr(), n * C, H, W, t5, 5, t5, 5, sp)),
                              stream, max(launches // 3, 3))
        extra.append(round(sum(ser) / len(ser), 4))
    leg["other_output_allocations_avg_launch_ms"] = extra
    out["cfg3"] = leg
    del x, gx, gy, keep, gx2, gy2

    # ---- cfg4: 256 x 3x224x224 -> Conv2d(3, 64, 3, padding=1) + bias + ReLU (vgg.py:81-85); kaiming fan_out weights (vgg.py:55)
    n, hh, cout = 256, 224, 64
    x = torch.empty((n, 3, hh, hh), dtype=torch.float32, device=dev).uniform_(0.0, 1.0, generator=g)
    wt = torch.empty((cout, 3, 3, 3), dtype=torch.float32, device=dev).normal_(0.0, (2.0 / (cout * 9)) ** 0.5, generator=g)
    b = torch.empty((cout,), dtype=torch.float32, device=dev).uniform_(-0.1, 0.1, generator=g)
    y = torch.empty((n, cout, hh, hh), dtype=torch.float32, device=dev)
    series = _timed_launches(lambda: _lib.check(lib.mv_conv3x3_bias_relu_f32(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y.data_ptr(), n, 3, hh, hh, cout, 1, sp)),
                             stream, launches)
    leg = _leg(series, _lib.last_kernel(), (x.numel() + y.numel() + wt.numel() + b.numel()) * 4, flops=2.0 * n * cout * hh * hh * 27)
    leg["workload"] = f"cfg4: {n}x3x{hh}x{hh} fp32 -> Conv2d(3,{cout},3,padding=1)+bias+ReLU, one launch (write-bound: 3.29 GB out)"
    wn, bn = wt.cpu().numpy(), b.cpu().numpy()
    leg["bit_exact_vs_oracle"] = all(bool(np.array_equal(y[i].cpu().numpy(), ref.conv3x3_bias_relu(x[i:i + 1].cpu().numpy(), wn, bn)[0])) for i in (0, n - 1))
    leg["frames_checked"] = [0, n - 1]
    cpu_in = (x[:64].cpu(), wt.cpu(), b.cpu(), y[:64].cpu())
    extra, keep = [], [y]
    for _ in range(2):  # as for cfg3: the same launch into two more output allocations
        y2 = torch.empty_like(y)
        keep.append(y2)
        ser = _timed_launches(lambda: _lib.check(lib.mv_conv3x3_bias_relu_f32(x.data_ptr(), wt.data_ptr(), b.data_ptr(), y2.data_ptr(), n, 3, hh, hh, cout, 1, sp)),
                              stream, max(launches // 3, 3))
        extra.append(round(sum(ser) / len(ser), 4))
    leg["other_output_allocations_avg_launch_ms"] = extra
    out["cfg4"] = leg
    del x, y, keep, y2
    torch.cuda.empty_cache()
    return out, cpu_in


def cpu_legs(x_frame, threads: int, cfg4_in):
    """Further legs of the CPU baseline: the headline filter on ONE thread, and cfg4's
    Conv2d(3,64,3,p=1)+ReLU through torch CPU ops (the reference's path: nn.Conv2d + nn.ReLU, vgg.py:81-85) with the pool size
    the headline leg found fastest and on one thread.  Bounded samples (a few seconds each)."""
    import torch
    from oracle import ref_torch

    def med(fn, reps, budget):
        fn()
        ts, t_end = [], time.perf_counter() + budget
        while len(ts) < reps and (time.perf_counter() < t_end or len(ts) < 2):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    legs = {}
    xf = x_frame.cpu()
    torch.set_num_threads(1)
    t, reps = med(lambda: ref_torch.gaussian_blur_image(xf, [3, 3], [0.8, 0.8]), 5, 6.0)
    legs["headline_1_thread"] = {"value": round(H * W / 1e6 / t, 2), "unit": "Mpixels/s", "cores": 1, "ms_per_frame": round(t * 1e3, 2),
                                 "sample": f"one 3x{H}x{W} frame, median of {reps} runs"}
    if cfg4_in is not None:
        x, w, b, y_gpu = cfg4_in
        n = x.shape[0]
        torch.set_num_threads(threads)
        t, reps = med(lambda: ref_torch.conv3x3_bias_relu(x, w, b), 7, 6.0)
        y_cpu = ref_torch.conv3x3_bias_relu(x, w, b)
        err = (y_gpu - y_cpu).abs()
        tol = 1e-5 * y_cpu.abs() + 1e-6 * float(w.abs().sum(dim=(1, 2, 3)).max()) * float(x.abs().max())
        legs["cfg4_conv_relu"] = {"value": round(n / t, 1), "unit": "images/s", "cores": threads, "ms_per_256_images": round(t / n * 256 * 1e3, 1),
                                  "GFLOPs": round(2.0 * n * 64 * 224 * 224 * 27 / t / 1e9, 1),
                                  "sample": f"{n} of the 256 images, conv2d(pad=1)+relu_ via torch CPU ops, median of {reps} runs",
                                  "gpu_vs_cpu_path_within_1e-5": bool((err <= tol).all()), "max_abs_err_vs_cpu_path": float(err.max())}
        torch.set_num_threads(1)
        xs = x[:8]
        t, reps = med(lambda: ref_torch.conv3x3_bias_relu(xs, w, b), 5, 6.0)
        legs["cfg4_conv_relu_1_thread"] = {"value": round(8 / t, 1), "unit": "images/s", "cores": 1,
                                           "GFLOPs": round(2.0 * 8 * 64 * 224 * 224 * 27 / t / 1e9, 1),
                                           "sample": f"8 of the 256 images, median of {reps} runs"}
    torch.set_num_threads(threads)
    return legs


def kernel_source_sha() -> str:
    """SHA-256 (16 hex digits) of the sources of the headline kernel: ties profiles/traffic_latest.json to the kernel
    it was measured on even when unrelated files of the library change."""
    h = hashlib.sha256()
    for f in ("cpu-vision_amd/csrc/dwtile.hip", "cpu-vision_amd/csrc/mv_common.h"):
        h.update((ROOT / f).read_bytes())
    return h.hexdigest()[:16]


def _rccl_version(backend):
    if backend != "nccl":
        return None
    try:
        import torch

        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        return None


def run_rank(a) -> int:
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}")
    dry = a.control_plane_only
    if not dry and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if a.all_ranks_on_gpu0:
        local = 0
    backend = "gloo" if dry else a.backend
    dev = torch.device("cpu") if dry else torch.device("cuda", local)
    if not dry:
        torch.cuda.set_device(local)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where scalar reductions live

    from cpu_vision_amd import sharding

    frames_total = a.frames_per_gpu * world
    lo, hi = sharding.shard_range(frames_total, world, rank)
    n = hi - lo

    def barrier():
        if not dry:
            torch.cuda.synchronize(dev)
        if use_dist:
            if backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        if not dry:
            torch.cuda.synchronize(dev)

    kernel_name, build_id, x, y = "", "", None, None
    if dry:
        def step():
            return None
    else:
        import cpu_vision_amd as mv
        from cpu_vision_amd import _lib, functional as F

        lib = mv.load_library()
        build_id = _lib.build_id()
        g = torch.Generator(device=dev).manual_seed(5000 + rank)
        x = torch.empty((n, C, H, W), dtype=torch.float32, device=dev)
        for i in range(0, n, 16):  # fill in slices: no 12 GB temporaries
            x[i:i + 16].uniform_(0.0, 1.0, generator=g)
        y = torch.empty_like(x)
        k1 = F._get_gaussian_kernel1d(3, 0.8)  # sigma = 0.15*3 + 0.35, gaussian_blur_image's default
        tx, ty = _lib.taps_from_tensor(k1), _lib.taps_from_tensor(k1)
        planes = n * C
        stream = torch.cuda.current_stream(dev)

        def step():
            _lib.check(lib.mv_gaussian_blur_f32(x.data_ptr(), y.data_ptr(), planes, H, W, tx, 3, ty, 3, stream.cuda_stream))

    if not dry:
        _spin_up(step, stream)  # setup: clocks to their steady state under load (see _spin_up); then the contract's W warm-up steps
    for _ in range(a.warmup):
        step()
    if not dry:
        kernel_name = _lib.last_kernel()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    if not dry:
        ev[0].record(stream)
    for i in range(a.steps):
        step()
        if not dry:
            ev[i + 1].record(stream)  # HIP events on the stream the kernel is launched on
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, red_dev)

    # ---- untimed: per-rank launch statistics, parity against the oracle, checksum of the WHOLE output
    alg_bytes = n * C * H * W * BYTES_PER_ELEMENT
    if dry:
        stats = [0.0, 0.0, -1.0, -1.0, 0.0]  # -1: nothing ran, nothing was compared
        ident = {"device": "cpu (control-plane rehearsal)", "pci_bus_id": None}
    else:
        launch_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps))
        checked = frames_vs_oracle(x, y, sorted({0, n - 1}))
        csum = 0.0
        for i in range(0, n, 16):
            csum += float(y[i:i + 16].double().sum().item())
        stats = [sum(launch_ms) / len(launch_ms), launch_ms[0], float(checked[0]), float(checked[n - 1]), csum]
        pr = torch.cuda.get_device_properties(dev)
        ident = {"device": pr.name, "gcn_arch": getattr(pr, "gcnArchName", None), "compute_units": pr.multi_processor_count,
                 "hbm_GiB": round(pr.total_memory / 2 ** 30, 1), "local_rank": local,
                 "pci_bus_id": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0))}
    t = torch.tensor(stats, dtype=torch.float64, device=red_dev)
    if use_dist:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)  # RCCL / gloo, outside the timed region
        idents = [None] * world
        dist.all_gather_object(idents, ident)  # which card every rank really ran on
        world_seen, backend_seen = dist.get_world_size(), dist.get_backend()
    else:
        parts, idents = [t], [ident]
        world_seen, backend_seen = 1, None
    per_rank = []
    for r, p in enumerate(parts):
        avg_ms, min_ms, ok0, ok1, cs = [float(v) for v in p.cpu()]
        r_lo, r_hi = sharding.shard_range(frames_total, world, r)
        r_bytes = (r_hi - r_lo) * C * H * W * BYTES_PER_ELEMENT
        per_rank.append({"rank": r, "frames": [r_lo, r_hi], "avg_launch_ms": round(avg_ms, 4), "min_launch_ms": round(min_ms, 4),
                         "achieved_GBps": round(r_bytes / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None,
                         "first_frame_bit_exact_vs_oracle": None if ok0 < 0 else bool(ok0),
                         "last_frame_bit_exact_vs_oracle": None if ok1 < 0 else bool(ok1),
                         "checksum": cs, **(idents[r] or {})})
    avg_launch = [p["avg_launch_ms"] for p in per_rank]
    slowest = max(avg_launch)  # the roofline figure of the job is the slowest rank's kernel
    achieved = alg_bytes / (slowest * 1e-3) / 1e9 if slowest > 0 else None
    parity_ok = None if dry else all(p["first_frame_bit_exact_vs_oracle"] and p["last_frame_bit_exact_vs_oracle"] for p in per_rank)

    total_mpix = frames_total * H * W / 1e6
    out = {
        "metric": "Mpixels/sec on 3x3 conv2d, 4K fp32 frames",
        "value": None if dry else round(total_mpix * a.steps / elapsed, 1),
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(elapsed / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{frames_total} frames of 3x{H}x{W} fp32 ({a.frames_per_gpu}/GPU, BASELINE cfg5 shard), "
                               f"3x3 Gaussian sigma=0.8 depthwise conv2d, reflect border, one launch per step",
                   "frames_per_gpu": a.frames_per_gpu, "setup_spin_up_s": None if dry else SPIN_UP_S, "sharding": f"image-sharded x{world}, no data-path collective",
                   "backend": ("RCCL (torch.distributed nccl)" if backend == "nccl" else "gloo (rehearsal)") if use_dist else "single process",
                   "world_size": world_seen, "world_size_env": world, "dist_backend": backend_seen, "rccl_version": _rccl_version(backend_seen),
                   "distinct_devices": len({(p.get("pci_bus_id"), p.get("local_rank")) for p in per_rank}),
                   "launcher": os.environ.get("MV_BENCH_LAUNCHER", "torchrun" if world > 1 else "direct")},
        "roofline": {
            "bound": "hbm",
            "achieved": None if achieved is None else round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "kernel": kernel_name,
            "library_build_id": build_id,
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": round(slowest, 4),
            "avg_launch_ms_min_over_ranks": round(min(avg_launch), 4),
            "avg_launch_ms_max_over_ranks": round(slowest, 4),
            "min_launch_ms": round(min(p["min_launch_ms"] for p in per_rank), 4),
        },
        "parity": {"gpu_vs_oracle_bit_exact": parity_ok, "frames_checked_per_rank": "first and last frame of each rank's shard",
                   "checksum_all_frames": sum(p["checksum"] for p in per_rank)},
        "per_rank": per_rank,
    }
    if dry:
        out["rehearsal"] = "control-plane-only: launcher, rendezvous, barriers, reductions and gathers ran; NO kernel was launched and nothing was measured"
    traffic_file = ROOT / "profiles" / "traffic_latest.json"
    if traffic_file.exists() and not dry:  # PMC-derived HBM bytes per launch from separate rocprofv3 --pmc passes (tools/profile_pmc.py)
        try:
            tj = json.loads(traffic_file.read_text())
            same_kernel = tj.get("kernel_source_sha") == kernel_source_sha() and tj.get("kernel") == kernel_name
            if tj.get("frames_per_gpu") == a.frames_per_gpu and same_kernel:
                out["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = tj.get("source")
                out["roofline"]["traffic_measured_on_build"] = tj.get("library_build_id")
            else:
                out["roofline"]["traffic_note"] = ("profiles/traffic_latest.json was measured on another kernel / shard size "
                                                   f"({tj.get('kernel')}, sources {tj.get('kernel_source_sha')}): not reported")
        except Exception:
            pass
    cfg4_cpu_in = None
    if world == 1 and not dry:
        x0, y0 = x[0].clone(), y[0].clone()
        del x, y  # the headline's 25 GB: the other configs bring their own batches
        torch.cuda.empty_cache()
        if not a.no_configs:
            out["configs"], cfg4_cpu_in = other_configs(dev, a.config_launches)
            parity_ok = parity_ok and all(c["bit_exact_vs_oracle"] for c in out["configs"].values())
            out["parity"]["configs_bit_exact_vs_oracle"] = {k: c["bit_exact_vs_oracle"] for k, c in out["configs"].items()}
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not dry:
        out["cpu_baseline"] = cpu_baseline(x0, y0, a.cpu_seconds)
        out["cpu_baseline"]["gpu_vs_oracle_bit_exact"] = parity_ok
        out["cpu_baseline"]["legs"] = cpu_legs(x0, out["cpu_baseline"]["cores"], cfg4_cpu_in)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier() if backend != "nccl" else dist.barrier(device_ids=[local])
        dist.destroy_process_group()
    if not dry and not parity_ok:
        return 3  # a fast kernel with wrong results is not a result
    return 0


def main() -> int:
    argv = sys.argv[1:]
    a = parse(argv)
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a, argv)
    return run_rank(a)


if __name__ == "__main__":
    sys.exit(main())
