"""bench.py's multi-GPU leg: `python bench.py --gpus N` must start its own N ranks (the driver's SCALE run calls it that
way) and must equally work as one rank under `python -m torch.distributed.run` (the reference's convention,
references/classification/utils.py:245-269: RANK / WORLD_SIZE / LOCAL_RANK from the environment).

CPU part (this file, no marker): the launcher and the distributed control flow with `--control-plane-only` -- ranks
rendezvous over gloo, barrier, reduce and gather exactly as in a real run, but launch no kernel and report value = null.
GPU part (`-m gpu`): the same launcher with real launches, two gloo ranks sharing the box's one GPU."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BENCH = str(ROOT / "bench.py")


def _run(cmd, timeout=300, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e, cwd=str(ROOT))


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def _check_control_plane_line(d, world, launcher):
    assert d["n_gpus"] == world and d["config"]["world_size"] == world and d["config"]["launcher"] == launcher
    assert d["value"] is None and "control-plane-only" in d["rehearsal"]          # nothing measured, and it says so
    assert d["parity"]["gpu_vs_oracle_bit_exact"] is None and d["roofline"]["achieved"] is None
    assert d["scaling"] == "weak" and d["steps"] == 3 and d["warmup"] == 1
    assert [p["rank"] for p in d["per_rank"]] == list(range(world))
    per = d["config"]["frames_per_gpu"]
    assert [p["frames"] for p in d["per_rank"]] == [[r * per, (r + 1) * per] for r in range(world)]


@pytest.mark.parametrize("world", [2, 3])
def test_self_launcher_spawns_its_ranks(world):
    r = _run([sys.executable, BENCH, "--gpus", str(world), "--control-plane-only", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    _check_control_plane_line(_line(r.stdout), world, "self")


def test_runs_as_a_rank_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), BENCH, "--gpus", "2", "--control-plane-only", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    _check_control_plane_line(_line(r.stdout), 2, "torchrun")


def test_single_process_control_plane_and_world_size_mismatch():
    r = _run([sys.executable, BENCH, "--control-plane-only", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    _check_control_plane_line(_line(r.stdout), 1, "direct")
    r = _run([sys.executable, BENCH, "--gpus", "2", "--control-plane-only"], env={"WORLD_SIZE": "4", "RANK": "0"})
    assert r.returncode != 0 and "does not match" in r.stderr


def test_without_a_gpu_the_real_path_fails_loudly():
    """No CPU fallback: on a box without a HIP device the measured path refuses to run, under the launcher too."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("box has a GPU")
    r = _run([sys.executable, BENCH, "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "no HIP device" in r.stderr and not r.stdout.strip()
    r = _run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"])
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_self_launcher_two_ranks_real_launches_one_gpu():
    """Two gloo ranks sharing cuda:0 (RCCL refuses two ranks on one device), 8 frames each: real launches, per-rank launch
    statistics, per-rank oracle parity and the kernel name reported by the library."""
    r = _run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--all-ranks-on-gpu0", "--frames-per-gpu", "8",
              "--steps", "3", "--warmup", "1"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["launcher"] == "self" and d["value"] > 0
    assert d["parity"]["gpu_vs_oracle_bit_exact"] is True
    assert len(d["per_rank"]) == 2 and all(p["first_frame_bit_exact_vs_oracle"] and p["last_frame_bit_exact_vs_oracle"]
                                           and p["avg_launch_ms"] > 0 for p in d["per_rank"])
    assert d["roofline"]["kernel"].startswith("k_dwtile<f32,3x3") and d["roofline"]["library_build_id"]
    assert d["per_rank"][0]["checksum"] != d["per_rank"][1]["checksum"]  # different seeds per rank: distinct shards


@pytest.mark.gpu
def test_rccl_calls_on_a_one_rank_group():
    """The RCCL code path of bench.py (init_process_group("nccl", device_id=...), barrier(device_ids=...), MAX all-reduce and
    all-gather of device tensors, destroy) on the one GPU of the box: a one-rank group under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
              "--master-port", str(port), BENCH, "--gpus", "1", "--force-dist", "--frames-per-gpu", "8", "--steps", "3", "--warmup", "1",
              "--no-cpu-baseline"], timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["config"]["backend"].startswith("RCCL") and d["n_gpus"] == 1 and d["value"] > 0
    assert d["parity"]["gpu_vs_oracle_bit_exact"] is True and len(d["per_rank"]) == 1
