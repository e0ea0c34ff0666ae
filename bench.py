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
  per_rank     for every rank: average / minimum launch time, achieved GB/s, and a bit-for-bit check of the first and
               the last frame of its shard against the C oracle (untimed);
  cpu_baseline (N = 1) the reference's CPU path (pad(reflect) + conv2d(groups=C) through torch CPU ops, the exact call
               sequence of gaussian_blur_image) timed on this box's host cores on a bounded sample.
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
        os.killpg(proc.pid, 15)  # the exact process group this launcher started
        try:
            out, _ = proc.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, 9)
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


def kernel_source_sha() -> str:
    """SHA-256 (16 hex digits) of the sources of the headline kernel: ties profiles/traffic_latest.json to the kernel
    it was measured on even when unrelated files of the library change."""
    h = hashlib.sha256()
    for f in ("cpu-vision_amd/csrc/dwtile.hip", "cpu-vision_amd/csrc/mv_common.h"):
        h.update((ROOT / f).read_bytes())
    return h.hexdigest()[:16]


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
    else:
        launch_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps))
        checked = frames_vs_oracle(x, y, sorted({0, n - 1}))
        csum = 0.0
        for i in range(0, n, 16):
            csum += float(y[i:i + 16].double().sum().item())
        stats = [sum(launch_ms) / len(launch_ms), launch_ms[0], float(checked[0]), float(checked[n - 1]), csum]
    t = torch.tensor(stats, dtype=torch.float64, device=red_dev)
    if use_dist:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)  # RCCL / gloo, outside the timed region
    else:
        parts = [t]
    per_rank = []
    for r, p in enumerate(parts):
        avg_ms, min_ms, ok0, ok1, cs = [float(v) for v in p.cpu()]
        r_lo, r_hi = sharding.shard_range(frames_total, world, r)
        r_bytes = (r_hi - r_lo) * C * H * W * BYTES_PER_ELEMENT
        per_rank.append({"rank": r, "frames": [r_lo, r_hi], "avg_launch_ms": round(avg_ms, 4), "min_launch_ms": round(min_ms, 4),
                         "achieved_GBps": round(r_bytes / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None,
                         "first_frame_bit_exact_vs_oracle": None if ok0 < 0 else bool(ok0),
                         "last_frame_bit_exact_vs_oracle": None if ok1 < 0 else bool(ok1),
                         "checksum": cs})
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
                   "frames_per_gpu": a.frames_per_gpu, "sharding": f"image-sharded x{world}, no data-path collective",
                   "backend": ("RCCL (torch.distributed nccl)" if backend == "nccl" else "gloo (rehearsal)") if use_dist else "single process",
                   "world_size": world, "launcher": os.environ.get("MV_BENCH_LAUNCHER", "torchrun" if world > 1 else "direct")},
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
    if rank == 0 and world == 1 and not a.no_cpu_baseline and not dry:
        out["cpu_baseline"] = cpu_baseline(x[0], y[0], a.cpu_seconds)
        out["cpu_baseline"]["gpu_vs_oracle_bit_exact"] = parity_ok
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
