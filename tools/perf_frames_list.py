#!/usr/bin/env python3
"""Separately allocated frames (a DataLoader's output): one launch per frame against ONE launch for the whole list
(mv_gaussian_blur_f32_v: per-frame base pointers in the kernel arguments).  96 x 1080p fp32 frames = BASELINE cfg2."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import cpu_vision_amd as mv  # noqa: E402
from cpu_vision_amd import _lib, functional as F  # noqa: E402

lib = mv.load_library()
HBM = 8000.0


def timeit(fn, rounds=9):
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


for name, shape, n, dtype in (("1080p f32", (3, 1080, 1920), 96, torch.float32), ("720p f32", (3, 720, 1280), 96, torch.float32),
                              ("4K f32", (3, 2160, 3840), 32, torch.float32), ("1080p u8", (3, 1080, 1920), 96, torch.uint8),
                              ("224 crop f32", (3, 224, 224), 256, torch.float32)):
    keep, frames = [], []
    for i in range(n):  # separate allocations, other allocations in between
        keep.append(torch.empty(4096 * (1 + i % 3), dtype=torch.uint8, device="cuda"))
        frames.append(torch.rand(shape, device="cuda") if dtype == torch.float32 else torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda"))
    outs = [torch.empty_like(f) for f in frames]
    k = F._host_taps(3, 0.8)[1]
    planes, h, w = shape
    sp = torch.cuda.current_stream().cuda_stream
    one = lib.mv_gaussian_blur_f32 if dtype == torch.float32 else lib.mv_gaussian_blur_u8
    many = lib.mv_gaussian_blur_f32_v if dtype == torch.float32 else lib.mv_gaussian_blur_u8_v

    def per_frame():
        for f, o in zip(frames, outs):
            one(f.data_ptr(), o.data_ptr(), planes, h, w, k, 3, k, 3, sp)

    xs, ys = _lib.pointer_table(frames), _lib.pointer_table(outs)

    def one_launch():
        _lib.check(many(xs, ys, n, planes, h, w, k, 3, k, 3, sp))

    nbytes = n * planes * h * w * 2 * frames[0].element_size()
    a, b = timeit(per_frame), timeit(one_launch)
    batch = torch.stack(frames)
    bo = torch.empty_like(batch)
    c = timeit(lambda: one(batch.data_ptr(), bo.data_ptr(), n * planes, h, w, k, 3, k, 3, sp))
    assert all(torch.equal(o, bo[i]) for i, o in enumerate(outs))
    for label, ms in (("one launch per frame (C ABI calls)", a), ("ONE launch, pointer table (mv_*_v)", b), ("contiguous batch, one launch", c)):
        print(f"{n:4d} x {name:13s} 3x3 blur  {label:38s} {ms / n * 1e3:8.2f} us/frame  {nbytes / ms / 1e6:7.0f} GB/s ({nbytes / ms / 1e6 / HBM * 100:4.1f} % HBM)", flush=True)
    del frames, outs, keep, batch, bo
