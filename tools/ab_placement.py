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

# ---- cfg3: one read stream and two write streams: does their RELATIVE placement matter?
del pool, x
torch.cuda.empty_cache()
n, H, W = 32, 2160, 3840
nb = n * 3 * H * W * 4
pool = torch.empty(3 * nb + (3 << 30), dtype=torch.uint8, device="cuda")
base = (pool.data_ptr() + (1 << 21) - 1) & ~((1 << 21) - 1)
xf = torch.rand((n, 3, H, W), device="cuda")
k5 = (C.c_float * 5)(0.1, 0.2, 0.4, 0.2, 0.1)
import ctypes
lib.mv_gaussian_sobel_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]
combos = [(0, 0), (4096, 8192), (1 << 20, 2 << 20), ((1 << 20) + 4096, (2 << 20) + 8192), (64 << 20, 128 << 20), ((64 << 20) + 65536, (128 << 20) + 131072),
          (512 << 20, 1 << 30), ((512 << 20) + (1 << 19), (1 << 30) + (1 << 20))]
# x copied into the pool at `base`; gx at base + nb + d1; gy at base + 2 nb + d2
xin = base
ctypes.memmove  # (no host copy: device-to-device below)
torch.cuda.synchronize()
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(C.c_void_p(xin), C.c_void_p(xf.data_ptr()), C.c_size_t(nb), 3)
times = {c: [] for c in combos}
for r in range(7):
    for c in combos:
        gx, gy = base + nb + c[0], base + 2 * nb + c[1]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            rc = lib.mv_gaussian_sobel_f32(C.c_void_p(xin), C.c_void_p(gx), C.c_void_p(gy), n * 3, H, W, k5, 5, k5, 5, C.c_void_p(s))
            assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        times[c].append(e0.elapsed_time(e1) / 3)
print(f"cfg3 pool base {base:#x}")
for c in combos:
    print(f"cfg3 gx at +{c[0]:>11d} B, gy at +{c[1]:>11d} B past their slots: median {statistics.median(times[c]):.4f} ms  min {min(times[c]):.4f}")
# and through torch's allocator, as bench.py does it
gxt, gyt = torch.empty_like(xf), torch.empty_like(xf)
ts = []
for r in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        lib.mv_gaussian_sobel_f32(C.c_void_p(xf.data_ptr()), C.c_void_p(gxt.data_ptr()), C.c_void_p(gyt.data_ptr()), n * 3, H, W, k5, 5, k5, 5, C.c_void_p(s))
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 3)
print(f"cfg3 torch allocations x {xf.data_ptr():#x} gx {gxt.data_ptr():#x} gy {gyt.data_ptr():#x}: median {statistics.median(ts):.4f} ms")

# ---- which part of "torch allocations" is it: the order of the three buffers, or one allocation against three?
def run3(xa, gxa, gya, label):
    ts = []
    for r in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            rc = lib.mv_gaussian_sobel_f32(C.c_void_p(xa), C.c_void_p(gxa), C.c_void_p(gya), n * 3, H, W, k5, 5, k5, 5, C.c_void_p(s))
            assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 3)
    print(f"cfg3 {label}: x {xa:#x} gx {gxa:#x} gy {gya:#x}: median {statistics.median(ts):.4f} ms  min {min(ts):.4f}", flush=True)

slot = (nb + (1 << 21) - 1) & ~((1 << 21) - 1)
hip.hipMemcpy(C.c_void_p(base + 2 * slot), C.c_void_p(xf.data_ptr()), C.c_size_t(nb), 3)
run3(base + 2 * slot, base + slot, base, "ONE pool, descending (gy, gx, x)")
hip.hipMemcpy(C.c_void_p(base + slot), C.c_void_p(xf.data_ptr()), C.c_size_t(nb), 3)
run3(base + slot, base, base + 2 * slot, "ONE pool, x in the middle")
run3(xf.data_ptr(), base, base + slot, "x = torch tensor, gx / gy in the pool")
run3(base + slot, gxt.data_ptr(), gyt.data_ptr(), "x in the pool, gx / gy = torch tensors")
ptrs = []
for i in range(3):
    q = C.c_void_p()
    assert hip.hipMalloc(C.byref(q), C.c_size_t(nb)) == 0
    ptrs.append(q.value)
hip.hipMemcpy(C.c_void_p(ptrs[0]), C.c_void_p(xf.data_ptr()), C.c_size_t(nb), 3)
run3(ptrs[0], ptrs[1], ptrs[2], "three hipMalloc buffers")
run3(xf.data_ptr(), gxt.data_ptr(), gyt.data_ptr(), "three torch tensors (again)")

run3(xf.data_ptr(), base, gyt.data_ptr(), "gx in the pool, gy = torch tensor")
run3(xf.data_ptr(), gxt.data_ptr(), base, "gx = torch tensor, gy in the pool")
pa, pb = C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(pa), C.c_size_t(2 * slot)) == 0 and hip.hipMalloc(C.byref(pb), C.c_size_t(2 * slot)) == 0
run3(xf.data_ptr(), pa.value, pb.value, "gx, gy in two hipMalloc buffers of 2x their size")
run3(xf.data_ptr(), pa.value, pa.value + slot, "gx, gy both in ONE hipMalloc buffer of 2x their size")
run3(xf.data_ptr(), ptrs[1], ptrs[2], "gx, gy = two hipMalloc buffers of exactly their size (again)")
