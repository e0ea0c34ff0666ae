"""Subprocess body of test_oracle_deform_vs_reference_native_kernel: loads oracle/_ref/ref_deform_conv2d_cpu.so (the
reference's own CPU kernel, built by oracle/build_ref.py) and compares the C restatement with it.  A separate process,
because the library defines the torchvision::deform_conv2d schema that cpu_vision_amd.ops.register_torchvision_op() would
define too."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from oracle import ref  # noqa: E402
from oracle.build_ref import built_library  # noqa: E402

so = built_library()
torch.ops.load_library(str(so))
op = torch.ops.torchvision.deform_conv2d
rng = np.random.Generator(np.random.Philox(77))
worst = 0.0
cases = [  # n, cin, cout, h, w, kh, kw, stride, pad, dil, groups, offset groups, mask, bias
    (4, 6, 2, 5, 4, 3, 2, (2, 1), (1, 0), (2, 1), 2, 3, True, True),      # TestDeformConv's configuration
    (2, 16, 32, 20, 24, 3, 3, (1, 1), (1, 1), (1, 1), 1, 1, True, True),
    (1, 64, 64, 28, 28, 3, 3, (1, 1), (1, 1), (1, 1), 1, 4, False, False),
    (2, 8, 12, 17, 13, 5, 3, (2, 2), (2, 1), (1, 1), 4, 2, True, False),
    (3, 3, 64, 16, 20, 3, 3, (1, 1), (1, 1), (1, 1), 1, 1, False, True),   # zero offsets below: the dense first layer (cfg4)
]
out = []
for ci, (n, cin, cout, h, w, kh, kw, st, pd, dl, g, og, um, ub) in enumerate(cases):
    oh = (h + 2 * pd[0] - (dl[0] * (kh - 1) + 1)) // st[0] + 1
    ow = (w + 2 * pd[1] - (dl[1] * (kw - 1) + 1)) // st[1] + 1
    x = rng.random((n, cin, h, w), dtype=np.float32) * 2 - 1
    off = (rng.standard_normal((n, og * 2 * kh * kw, oh, ow)) * 1.5).astype(np.float32)
    if ci == len(cases) - 1:
        off[:] = 0
    mask = rng.random((n, og * kh * kw, oh, ow), dtype=np.float32) if um else None
    wt = ((rng.random((cout, cin // g, kh, kw), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    b = (rng.random(cout, dtype=np.float32) - 0.5) if ub else None
    t = torch.from_numpy
    want = op(t(x), t(wt), t(off), t(mask) if um else torch.zeros((n, 0)), t(b) if ub else torch.zeros(cout), st[0], st[1], pd[0], pd[1],
              dl[0], dl[1], g, og, um).numpy()
    got = ref.deform_conv2d(x, off, wt, b, st, pd, dl, mask)
    err = float(np.abs(got - want).max())
    scale = float(np.abs(want).max())
    out.append({"case": ci, "max_abs_err": err, "max_abs": scale, "equal": bool(np.array_equal(got, want))})
    if ci == len(cases) - 1:  # zero offsets: also the oracle's plain conv3x3 + bias (no ReLU)
        conv = ref.conv3x3_bias_relu(x, wt, b, relu=False)
        out[-1]["dense_conv_max_abs_err"] = float(np.abs(conv - want).max())
# Non-finite pixels on the image border and samples EXACTLY on the outside boundary (h == -1 / w == -1: zero offsets with padding 1
# put the first kernel row / column there): bilinear_interpolate returns 0 for them before any pixel is touched
# (deform_conv2d_kernel.cpp:88-90), so the output pixel stays finite -- a restatement that multiplied the border pixel by a
# zero weight instead would produce NaN.  The non-finite PATTERN must match the reference's kernel exactly.
n, cin, cout, h, w = 1, 4, 6, 9, 11
x = rng.random((n, cin, h, w), dtype=np.float32) * 2 - 1
x[0, 0, 0, 0], x[0, 1, 0, 5], x[0, 2, 4, 0], x[0, 3, h - 1, w - 1] = np.inf, -np.inf, np.inf, np.nan
off = np.zeros((n, 18, h, w), np.float32)
off[0, :, 4:, :] = (rng.standard_normal((18, h - 4, w)) * 0.7).astype(np.float32)  # lower half: ordinary fractional samples
wt = ((rng.random((cout, cin, 3, 3), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
want = op(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(off), torch.zeros((n, 0)), torch.zeros(cout), 1, 1, 1, 1, 1, 1, 1, 1,
          False).numpy()
got = ref.deform_conv2d(x, off, wt, None, (1, 1), (1, 1), (1, 1), None)
fin = np.isfinite(want)
out.append({"case": "non-finite border", "same_nan_pattern": bool(np.array_equal(np.isnan(got), np.isnan(want))),
            "same_inf_pattern": bool(np.array_equal(np.isinf(got), np.isinf(want)) and np.array_equal(np.sign(got[np.isinf(got)]), np.sign(want[np.isinf(want)]))),
            "finite_outputs": int(fin.sum()), "non_finite_outputs": int((~fin).sum()),
            "max_abs_err": float(np.abs(got[fin] - want[fin]).max()), "max_abs": float(np.abs(want[fin]).max())})
print(json.dumps(out))
