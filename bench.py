#!/usr/bin/env python3
"""Headline benchmark: Mpixels/s of the 3x3 depthwise conv2d (Gaussian 3x3, sigma 0.8, reflect border -- the
call gaussian_blur_image makes) over 4K fp32 3-channel frames, image-sharded over N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / cfg5): every rank owns 128 distinct frames of 3x2160x3840 fp32 resident in its
HBM (12.7 GB in, 12.7 GB out; 1024 frames at N=8 -- weak scaling).  One step = one pass of the hot path over the
rank's whole shard = ONE kernel launch; there is no collective in the data path (frames are independent), RCCL
is used for the barriers around the timed region and the max-over-ranks of the elapsed time.  25.5 GB of distinct
data per step per GPU, so the 256 MB Infinity Cache cannot masquerade as HBM.

The JSON line also carries
  roofline     achieved algorithmic GB/s of the kernel (8 B per element: one fp32 read + one fp32 write) from HIP
               events around every timed launch, against the 8 TB/s HBM3E peak;
  cpu_baseline the reference's CPU path (pad(reflect) + conv2d(groups=C) through torch CPU ops, the exact call
               sequence of gaussian_blur_image) timed on this box's host cores on a bounded sample, and a
               bit-for-bit check of the GPU output against the C oracle on the same frame.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, W, C = 2160, 3840, 3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a float4 copy achieves
BYTES_PER_ELEMENT = 8  # algorithmic: 4 B read + 4 B written per element (24 B per 3-channel pixel)


def parse():
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
    return p.parse_args()


def _cgroup_cpus():
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        return max(1, int(int(quota) / int(period))) if quota != "max" else None
    except Exception:
        return None


def cpu_baseline_and_parity(x_frame: torch.Tensor, y_frame: torch.Tensor, budget_s: float):
    """Rank 0, N=1 only.  Times the reference's CPU call sequence on ONE 4K frame repeatedly (bounded sample) and
    checks the GPU result for that frame against the C oracle, bit for bit."""
    from oracle import ref, ref_torch  # the checker / the baseline -- never the product path

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
    # parity in the same run, same input: GPU vs C oracle (bit-exact) and vs the torch CPU path (1e-5 relative)
    from cpu_vision_amd import functional as F
    k = F._get_gaussian_kernel1d(3, 0.8).numpy()
    y_orc = torch.from_numpy(ref.gaussian_blur(xf.numpy(), k, k))
    yg = y_frame.cpu()
    bit_exact = bool(torch.equal(yg, y_orc))
    err = (yg - y_cpu).abs()
    tol = 1e-5 * y_cpu.abs() + 1e-6
    within = bool((err <= tol).all())
    # the plain-C restatement, for reference (OpenMP over planes: 3 planes -> 3 threads busy)
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
        "gpu_vs_oracle_bit_exact": bit_exact,
        "gpu_vs_cpu_path_within_1e-5": within,
        "max_abs_err_vs_cpu_path": float(err.max()),
    }


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with --nproc-per-node {a.gpus}")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if a.all_ranks_on_gpu0:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    red_dev = dev if a.backend == "nccl" else torch.device("cpu")  # where scalar reductions live

    import cpu_vision_amd as mv
    from cpu_vision_amd import _lib, functional as F, sharding

    lib = mv.load_library()
    frames_total = a.frames_per_gpu * world
    lo, hi = sharding.shard_range(frames_total, world, rank)
    n = hi - lo
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

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            if a.backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    barrier()
    t0 = time.perf_counter()
    ev[0].record(stream)
    for i in range(a.steps):
        step()
        ev[i + 1].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, red_dev)

    launch_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps))
    avg_launch_ms = sum(launch_ms) / len(launch_ms)
    alg_bytes = n * C * H * W * BYTES_PER_ELEMENT
    achieved = alg_bytes / (avg_launch_ms * 1e-3) / 1e9
    checksum = sharding.global_checksum(y[:1] if a.backend == "nccl" else y[:1].cpu())  # scalar all-reduce, untimed

    total_mpix = frames_total * H * W / 1e6
    out = {
        "metric": "Mpixels/sec on 3x3 conv2d, 4K fp32 frames",
        "value": round(total_mpix * a.steps / elapsed, 1),
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
                   "frames_per_gpu": a.frames_per_gpu, "sharding": f"image-sharded x{world}, no data-path collective"},
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "kernel": "mv::k_dwtile<float, 3, 3, 4, vec4> (LDS halo tile 256x16)",
            "algorithmic_bytes_per_launch": alg_bytes,
            "avg_launch_ms": round(avg_launch_ms, 4),
            "min_launch_ms": round(launch_ms[0], 4),
        },
        "checksum_frame0": checksum,
    }
    traffic_file = ROOT / "profiles" / "traffic_latest.json"
    if traffic_file.exists():  # PMC-derived HBM bytes per launch from a separate rocprofv3 --pmc pass (tools/profile_pmc.py)
        try:
            t = json.loads(traffic_file.read_text())
            if t.get("frames_per_gpu") == a.frames_per_gpu:
                out["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = t.get("source")
        except Exception:
            pass
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_and_parity(x[0], y[0], a.cpu_seconds)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
