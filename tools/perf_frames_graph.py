#!/usr/bin/env python3
"""Independent frames, one launch each (BASELINE cfg2: a single 1920x1080 3-channel fp32 frame per call): a HIP graph whose
kernel nodes sit on `streams` parallel branches lets consecutive frames overlap each other's ramp-up and tail on the chip,
with no host work per frame.  Prints microseconds per frame for the plain stream of launches, a linear graph and forked
graphs."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
lib = _lib.load()
k = F._get_gaussian_kernel1d(3, 0.8)
tx = _lib.taps_from_tensor(k)


def time_us(fn, per):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / per)
    return best


for name, (h, w), nfr in (("1080p", (1080, 1920), 32), ("720p", (720, 1280), 64), ("4K", (2160, 3840), 16)):
    frames = torch.rand((nfr, 3, h, w), generator=g, device="cuda")
    outs = torch.empty_like(frames)
    xp = [frames[i].data_ptr() for i in range(nfr)]
    yp = [outs[i].data_ptr() for i in range(nfr)]
    fn = lib.mv_gaussian_blur_f32

    def launches(nstreams, side):
        cur = torch.cuda.current_stream()
        if nstreams == 1:
            for i in range(nfr):
                fn(xp[i], yp[i], 3, h, w, tx, 3, tx, 3, cur.cuda_stream)
            return
        for s in side[:nstreams]:
            s.wait_stream(cur)
        for i in range(nfr):
            fn(xp[i], yp[i], 3, h, w, tx, 3, tx, 3, side[i % nstreams].cuda_stream)
        for s in side[:nstreams]:
            cur.wait_stream(s)

    side = [torch.cuda.Stream() for _ in range(8)]
    line = f"{name}: {nfr} frames of 3x{h}x{w} fp32, us per frame:  stream of launches {time_us(lambda: launches(1, side), nfr):6.2f}"
    want = None
    for ns in (1, 2, 4, 8):
        launches(ns, side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            launches(ns, side)
        t = time_us(gr.replay, nfr)
        if want is None:
            want = outs.clone()
        assert torch.equal(outs, want)
        line += f"  graph x{ns} {t:6.2f} ({3 * h * w * 8 / t / 1e6:4.2f} TB/s)"
    one = torch.empty_like(frames)
    tb = time_us(lambda: fn(frames.data_ptr(), one.data_ptr(), 3 * nfr, h, w, tx, 3, tx, 3, torch.cuda.current_stream().cuda_stream), nfr)
    assert torch.equal(one, want)
    line += f"  one batched launch {tb:6.2f} ({3 * h * w * 8 / tb / 1e6:4.2f} TB/s)"
    print(line, flush=True)
