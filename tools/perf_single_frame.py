#!/usr/bin/env python3
"""One frame per launch (BASELINE cfg2: a single 1920x1080 3-channel fp32 frame, 3x3 Gaussian): microseconds per launch for
the LDS-tile kernel and the register-window kernel at several strip heights, frames rotated through > 256 MB so the
Infinity Cache cannot hold them.  Launches are issued back to back on one stream and timed as a block (HIP events)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
# knob sweeps run on the -DMV_TUNING build of the same sources (the product library reads no environment variable)
os.environ.setdefault("MI355VISION_LIB", str(Path(__file__).resolve().parent.parent / "cpu-vision_amd" / "lib" / "libmi355vision_tuning.so"))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)


def block_us(fn, n):
    fn(0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for name, (h, w), nfr in (("1080p", (1080, 1920), 24), ("4K", (2160, 3840), 8), ("720p", (720, 1280), 48)):
    frames = torch.rand((nfr, 3, h, w), generator=g, device="cuda")
    k = F._get_gaussian_kernel1d(3, 0.8)
    from cpu_vision_amd import _lib
    lib = _lib.load()
    tx = _lib.taps_from_tensor(k)
    y = torch.empty_like(frames[0])
    s = torch.cuda.current_stream().cuda_stream

    ptrs = [frames[i].data_ptr() for i in range(nfr)]
    yp = y.data_ptr()
    fn = lib.mv_gaussian_blur_f32

    def run(i):
        fn(ptrs[i % nfr], yp, 3, h, w, tx, 3, tx, 3, s)

    line = f"{name} 3x{h}x{w} fp32 ({3 * h * w * 8 / 1e6:.1f} MB read+written):"
    os.environ.pop("MV_FORCE_REG3X3", None)
    os.environ.pop("MV_DW3X3_ROWS", None)
    t = block_us(run, 4 * nfr)
    line += f"  tile {t:6.1f} us ({3 * h * w * 8 / t / 1e6:5.2f} TB/s)"
    for rpt, th in ((2, 8), (8, 32)):
        os.environ["MV_TILE_RPT_SIZED"] = str(rpt)
        t = block_us(run, 4 * nfr)
        line += f"  tile/{th} {t:6.1f} us ({3 * h * w * 8 / t / 1e6:5.2f} TB/s)"
    os.environ.pop("MV_TILE_RPT_SIZED", None)
    os.environ["MV_FORCE_REG3X3"] = "1"
    for rows in (4, 8, 16, 32):
        os.environ["MV_DW3X3_ROWS"] = str(rows)
        t = block_us(run, 4 * nfr)
        line += f"  reg/{rows} {t:6.1f} us ({3 * h * w * 8 / t / 1e6:5.2f} TB/s)"
    os.environ.pop("MV_FORCE_REG3X3", None)
    os.environ.pop("MV_DW3X3_ROWS", None)
    print(line, flush=True)
