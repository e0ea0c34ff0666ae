"""Batch sharding (BASELINE cfg5): partition arithmetic, and the N>1 path under world_size-2 gloo on CPU.

The per-rank function in the multi-process test is the CPU oracle (tests may use it): what is under test is the
sharding / gather / reduction logic around the kernel, which is identical on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cpu_vision_amd import sharding


def test_shard_range_partitions_exactly():
    for n in [0, 1, 7, 8, 128, 1024, 1031]:
        for world in [1, 2, 3, 4, 8]:
            spans = [sharding.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a1 >= a0
            sizes = sharding.shard_sizes(n, world)
            assert sum(sizes) == n and max(sizes) - min(sizes) <= 1
    assert sharding.shard_range(1024, 8, 3) == (384, 512)  # cfg5: 128 frames per GPU
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def test_single_process_passthrough():
    x = torch.arange(24.0).reshape(6, 4)
    y = sharding.apply_sharded(lambda t: t * 2, x, gather=True)
    assert torch.equal(y, x * 2)
    assert sharding.global_checksum(x) == float(x.sum())
    assert sharding.max_over_ranks(3.5) == 3.5


def _worker(rank, world, port, n_frames, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref
        from tests._util import philox_f32

        frames = torch.from_numpy(philox_f32(55, (n_frames, 3, 12, 20)))
        k = ref.gaussian_kernel1d(3, 0.8)

        def blur(t):
            return torch.from_numpy(ref.gaussian_blur(t.numpy(), k, k))

        full = sharding.apply_sharded(blur, frames, gather=True)
        local = sharding.apply_sharded(blur, frames)
        lo, hi = sharding.shard_range(n_frames, world, rank)
        ok_local = local.shape[0] == hi - lo and torch.equal(local, full[lo:hi])
        want = blur(frames)
        csum = sharding.global_checksum(local)
        tmax = sharding.max_over_ranks(float(rank + 1))
        q.put((rank, bool(ok_local), bool(torch.equal(full, want)), abs(csum - float(want.double().sum())) < 1e-6, tmax))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [8, 5])
def test_world_size_2_gloo(n_frames):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_local, ok_full, ok_sum, tmax in results:
        assert ok_local and ok_full and ok_sum and tmax == 2.0, (rank, ok_local, ok_full, ok_sum, tmax)
