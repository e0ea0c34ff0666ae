#!/usr/bin/env python3
"""Does the PLACEMENT of the output buffer move cfg4 / cfg3?  (bench.py under rocprofv3 ran cfg4 11 % faster and cfg3 13 % slower than
the same commit without the profiler: profiles/r03_bench.json vs r03_bench_under_rocprofv3.json -- the profiler changes nothing but
where the allocations land.)  One pool, the output carved at different byte offsets, interleaved rounds (GPU box)."""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import functional as F, _lib  # noqa: E402
import ctypes as C  # noqa: E402

lib = _lib.load()
s = torch.cuda.current_stream().cuda_stream
n, H, W = 256, 224, 224
x = torch.rand((n, 3, H, W), device="cuda")
wt, b = torch.randn((64, 3, 3, 3), device="cuda") * 0.06, torch.rand(64, device="cuda") - 0.5
ybytes = n * 64 * H * W * 4
pool = torch.empty(ybytes + (1 << 30), dtype=torch.uint8, device="cuda")
base = pool.data_ptr()
print(f"x at {x.data_ptr():#x}, pool at {base:#x} (mod 2 MiB: x {x.data_ptr() % (1 << 21):#x}, pool {base % (1 << 21):#x})")
offsets = [0, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 16 << 20, (16 << 20) + 65536, 128 << 20, 512 << 20, (512 << 20) + (1 << 20)]
for _ in range(50):
    x.mul_(1.0)
times = {o: [] for o in offsets}
for r in range(7):
    for o in offsets:
        yp = base + o
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            rc = lib.mv_conv3x3_bias_relu_f32(C.c_void_p(x.data_ptr()), C.c_void_p(wt.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(yp), n, 3, H, W, 64, 1, C.c_void_p(s))
            assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        times[o].append(e0.elapsed_time(e1) / 5)
for o in offsets:
    print(f"cfg4 y at pool + {o:>12d} B: median {statistics.median(times[o]):.4f} ms  min {min(times[o]):.4f}")
