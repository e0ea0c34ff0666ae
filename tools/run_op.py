#!/usr/bin/env python3
"""Run one operator a few times (target program for rocprofv3 passes)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--op", default="conv")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(0)
if a.op == "conv":
    x = torch.rand((256, 3, 224, 224), generator=g, device="cuda")
    w = torch.randn((64, 3, 3, 3), generator=g, device="cuda") * 0.06
    b = torch.rand(64, generator=g, device="cuda") - 0.5
    out = torch.empty((256, 64, 224, 224), device="cuda")
    fn = lambda: F.conv2d_bias_relu(x, w, b, out=out)  # noqa: E731
elif a.op == "deform":
    # DCNv2 layer 8 x 256 x 64 x 64 -> 256, 3x3, one offset group, mask (tools/perf_deform.py's first shape)
    from cpu_vision_amd import ops
    n, c, m, hw = 8, 256, 256, 64
    x = torch.rand((n, c, hw, hw), generator=g, device="cuda")
    w = torch.randn((m, c, 3, 3), generator=g, device="cuda") * 0.05
    b = torch.rand(m, generator=g, device="cuda")
    off = torch.randn((n, 18, hw, hw), generator=g, device="cuda") * 1.5
    mask = torch.rand((n, 9, hw, hw), generator=g, device="cuda")
    fn = lambda: ops.deform_conv2d(x, off, w, b, padding=(1, 1), mask=mask)  # noqa: E731
elif a.op.startswith("mobilenet"):
    # mobilenetN: the whole MobileNetV2 forward at batch N (default 64)
    from cpu_vision_amd import mobilenet as M
    torch.manual_seed(0)
    net = M.MobileNetV2(1000).cuda()
    x = torch.rand((int(a.op[9:] or 64), 3, 224, 224), generator=g, device="cuda")
    fn = lambda: net(x)  # noqa: E731
elif a.op.startswith("gen"):
    # genN_CIN_COUT_HW: the general-cin 3x3 conv + bias + ReLU (csrc/conv3x3_gen.hip), e.g. gen1_512_512_28
    n, cin, cout, hw = [int(v) for v in a.op[3:].split("_")]
    x = torch.rand((n, cin, hw, hw), generator=g, device="cuda")
    w = torch.randn((cout, cin, 3, 3), generator=g, device="cuda") * 0.02
    b = torch.rand(cout, generator=g, device="cuda")
    fn = lambda: F.conv2d_bias_relu(x, w, b)  # noqa: E731
elif a.op.startswith("pw") or a.op.startswith("dw"):
    # pwCIN_COUT_HW / dwC_HW_STRIDE at batch 64, BatchNorm fold + ReLU6
    parts = [int(v) for v in a.op[2:].split("_")]
    if a.op.startswith("pw"):
        cin, cout, hw = parts
        x = torch.rand((64, cin, hw, hw), generator=g, device="cuda")
        w = torch.randn((cout, cin, 1, 1), generator=g, device="cuda") * 0.1
        al, be = torch.rand(cout, generator=g, device="cuda") + 0.5, torch.rand(cout, generator=g, device="cuda")
        fn = lambda: F.conv_norm_act(x, w, None, al, be, None, affine="fma", activation="relu6")  # noqa: E731
    else:
        c, hw, st = parts
        x = torch.rand((64, c, hw, hw), generator=g, device="cuda")
        w = torch.randn((c, 1, 3, 3), generator=g, device="cuda") * 0.3
        al, be = torch.rand(c, generator=g, device="cuda") + 0.5, torch.rand(c, generator=g, device="cuda")
        fn = lambda: F.conv_norm_act(x, w, None, al, be, None, stride=st, groups=c, affine="fma", activation="relu6")  # noqa: E731
else:
    x = torch.rand((32, 3, 2160, 3840), generator=g, device="cuda")
    fn = {"blur3": lambda: F.gaussian_blur(x, [3, 3]),
          "sobel5": lambda: F.gaussian_sobel(x, [5, 5], [1.1, 1.1]),
          "sep5": lambda: F.separable_gaussian_blur(x, [5, 5], [1.1, 1.1]),
          "sep23": lambda: F.gaussian_blur(x, [23, 23]),
          "sep31": lambda: F.gaussian_blur(x, [31, 31]),
          "sep15": lambda: F.gaussian_blur(x, [15, 15]),
          "sharp": lambda: F.adjust_sharpness(x, 1.5),
          "u8blur3": None, "u8blur5": None, "u8blur7": None, "u8sharp": None}[a.op]
    if fn is None:
        xu = torch.randint(0, 256, (32, 3, 2160, 3840), generator=g, device="cuda", dtype=torch.uint8)
        yu = torch.empty_like(xu)
        fn = {"u8blur3": lambda: F.gaussian_blur(xu, [3, 3]), "u8blur5": lambda: F.gaussian_blur(xu, [5, 5]),
              "u8blur7": lambda: F.gaussian_blur(xu, [7, 7]),
              "u8sharp": lambda: F.adjust_sharpness(xu, 1.5)}[a.op]
for _ in range(a.iters):
    fn()
torch.cuda.synchronize()
