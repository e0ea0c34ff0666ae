"""Subprocess body of tests/test_gpu_shim.py and of the CPU-side load check: loads the C++ dispatcher shim
(cpu-vision_amd/lib/libmi355vision_torch.so) the way the reference loads its extension (torch.ops.load_library) and drives
torch.ops.torchvision.deform_conv2d through it.  A process of its own: the shim and the Python registration of
cpu_vision_amd.ops both claim the operator's CUDA key.

    python tests/_shim_worker.py cpu   -> registration + Meta (fake) kernel + schema checks, no GPU needed
    python tests/_shim_worker.py gpu   -> the reference test's configuration through the C++ CUDA key, autocast, torch.compile
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from cpu_vision_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "cpu"
ops.register_torchvision_op(native=True)
op = torch.ops.torchvision.deform_conv2d
out = {"mode": mode}
keys = torch._C._dispatch_dump("torchvision::deform_conv2d")
out["has_cuda_kernel"] = "CUDA:" in keys or "CUDA " in keys
out["has_meta_kernel"] = "Meta" in keys
out["has_autocast_kernel"] = "AutocastCUDA" in keys
out["kernels_are_native"] = "deform_conv2d_shim.cpp" in keys or "boxed unboxed" in keys

# the reference test's own configuration (test/test_ops.py:TestDeformConv.get_fn_args): 6 -> 2 channels, 2 weight groups,
# 3 offset groups, kernel (3, 2), stride (2, 1), padding (1, 0), dilation (2, 1)
n, cin, cout, h, w, kh, kw = 4, 6, 2, 5, 4, 3, 2
sh, sw, ph, pw, dh, dw, groups, og = 2, 1, 1, 0, 2, 1, 2, 3
oh = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
ow = (w + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1

# ---- Meta / fake kernel: the shape rule of _meta_registrations.py:177-198, no data touched
meta = lambda *s, **k: torch.empty(*s, device="meta", **k)  # noqa: E731
y = op(meta(n, cin, h, w), meta(cout, cin // groups, kh, kw), meta(n, og * 2 * kh * kw, oh, ow), meta(n, og * kh * kw, oh, ow), meta(cout),
       sh, sw, ph, pw, dh, dw, groups, og, True)
out["meta_shape"] = list(y.shape)
out["meta_device"] = str(y.device)
from torch._subclasses.fake_tensor import FakeTensorMode  # noqa: E402

with FakeTensorMode():
    fy = op(torch.empty(n, cin, h, w), torch.empty(cout, cin // groups, kh, kw), torch.empty(n, og * 2 * kh * kw, oh, ow),
            torch.empty(n, og * kh * kw, oh, ow), torch.empty(cout), sh, sw, ph, pw, dh, dw, groups, og, True)
out["fake_shape"] = list(fy.shape)

if mode == "gpu":
    from oracle import ref  # the checker

    rng = np.random.Generator(np.random.Philox(4242))
    x = rng.random((n, cin, h, w), dtype=np.float32) * 2 - 1
    off = (rng.standard_normal((n, og * 2 * kh * kw, oh, ow)) * 1.5).astype(np.float32)
    mask = rng.random((n, og * kh * kw, oh, ow), dtype=np.float32)
    wt = ((rng.random((cout, cin // groups, kh, kw), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    b = rng.random(cout, dtype=np.float32) - 0.5
    want = ref.deform_conv2d(x, off, wt, b, (sh, sw), (ph, pw), (dh, dw), mask)
    d = lambda a: torch.from_numpy(a).cuda()  # noqa: E731
    args = (d(x), d(wt), d(off), d(mask), d(b), sh, sw, ph, pw, dh, dw, groups, og, True)
    got = op(*args)
    out["bit_exact_vs_oracle"] = bool(np.array_equal(got.cpu().numpy(), want))
    # DCNv1 call of the reference's Python (ops/deform_conv.py:70-74): zero-sized placeholder mask, no bias
    got1 = op(d(x), d(wt), d(off), torch.zeros((n, 0), device="cuda"), torch.zeros(0, device="cuda"), sh, sw, ph, pw, dh, dw, groups, og, False)
    out["v1_bit_exact_vs_oracle"] = bool(np.array_equal(got1.cpu().numpy(), ref.deform_conv2d(x, off, wt, None, (sh, sw), (ph, pw), (dh, dw), None)))
    # float64 tensors (the reference's test dtype): computed in float32, returned as float64
    got64 = op(*[t.double() if isinstance(t, torch.Tensor) else t for t in args])
    out["f64_dtype"] = str(got64.dtype)
    out["f64_max_abs_err"] = float(np.abs(got64.cpu().numpy() - want).max())
    # autocast: half inputs are widened by the Autocast kernel, result in the input's dtype
    with torch.autocast("cuda", dtype=torch.float16):
        ha = op(*[t.half() if isinstance(t, torch.Tensor) else t for t in args])
    out["autocast_dtype"] = str(ha.dtype)
    xh = [t.half().float().cpu().numpy() for t in args[:5]]
    want_h = ref.deform_conv2d(xh[0], xh[2], xh[1], xh[4], (sh, sw), (ph, pw), (dh, dw), xh[3])
    out["autocast_max_abs_err_vs_oracle_on_rounded_inputs"] = float(np.abs(ha.float().cpu().numpy() - want_h).max())
    # torch.compile traces through the op (needs the fake kernel), then runs the C++ kernel
    try:
        fn = torch.compile(lambda *a: op(*a) * 2.0, backend="aot_eager", fullgraph=True)
        out["compiled_equals_eager"] = bool(torch.equal(fn(*args), got * 2.0))
    except Exception as e:  # noqa: BLE001
        out["compiled_equals_eager"] = f"{type(e).__name__}: {e}"[:300]
    # errors surface as RuntimeError through TORCH_CHECK, like the reference's kernels
    try:
        op(d(x)[:, :, 0], d(wt), d(off), d(mask), d(b), sh, sw, ph, pw, dh, dw, groups, og, True)
        out["bad_rank_raises"] = False
    except RuntimeError as e:
        out["bad_rank_raises"] = "4-D" in str(e)
    # a larger launch through the fused kernel (no workspace) and one that needs the columns workspace (7 x 7 taps)
    for tag, (cn, ci, co, hh, kk, pad) in {"fused": (2, 16, 32, 20, 3, 1), "columns": (2, 8, 6, 16, 7, 3)}.items():
        xx = rng.random((cn, ci, hh, hh), dtype=np.float32) * 2 - 1
        oo = (rng.standard_normal((cn, 2 * kk * kk, hh, hh)) * 1.5).astype(np.float32)
        ww = ((rng.random((co, ci, kk, kk), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
        gg = op(d(xx), d(ww), d(oo), torch.zeros((cn, 0), device="cuda"), torch.zeros(0, device="cuda"), 1, 1, pad, pad, 1, 1, 1, 1, False)
        out[f"{tag}_bit_exact_vs_oracle"] = bool(np.array_equal(gg.cpu().numpy(), ref.deform_conv2d(xx, oo, ww, None, (1, 1), (pad, pad), (1, 1), None)))
print(json.dumps(out))
