"""Image-batch sharding across the GPUs of one node (BASELINE cfg5).

Every frame is independent, so the batch path shards with NO collective in the data path: rank r owns the
contiguous block of frames [r*N/W, (r+1)*N/W) and runs the same kernel on its own HBM.  torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only to align the timed
region, to reduce a scalar checksum / elapsed time, and -- outside any timed region -- to gather outputs for
verification.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block partition; the first (n_items % world_size) ranks get one extra item."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(n_items: int, world_size: int) -> List[int]:
    return [shard_range(n_items, world_size, r)[1] - shard_range(n_items, world_size, r)[0] for r in range(world_size)]


def _world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def apply_sharded(fn: Callable[[torch.Tensor], torch.Tensor], frames: torch.Tensor, gather: bool = False
                  ) -> torch.Tensor:
    """Run `fn` on this rank's block of `frames` (N, ...).  With gather=True every rank returns the full
    (N, ...) result (all_gather of the per-rank outputs; verification only -- 12.7 GB/GPU of 4K output over
    ~153 GB/s xGMI links would dwarf the kernel, so never inside a timed region)."""
    rank, world = _world()
    lo, hi = shard_range(frames.shape[0], world, rank)
    local = fn(frames[lo:hi])
    if not gather or world == 1:
        return local
    sizes = shard_sizes(frames.shape[0], world)
    pad = max(sizes)
    buf = local.new_zeros((pad,) + tuple(local.shape[1:]))
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def _in_group() -> bool:
    return dist.is_available() and dist.is_initialized()


def global_checksum(local: torch.Tensor) -> float:
    """fp64 sum over all ranks' outputs (one scalar all-reduce; also run for a one-rank group, which costs nothing and keeps
    the collective path exercised)."""
    s = local.double().sum().reshape(1)
    if _in_group():
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return float(s.item())


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    if not _in_group():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
