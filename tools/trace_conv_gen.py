#!/usr/bin/env python3
"""Shader-clock stamps inside k_conv3x3_gen's loader / compute specialisation (GPU box): where workgroup 0's compute wave 0 and
loader wave 4 spend a K chunk.

    MV_BUILD_VARIANT=gentrace MV_VARIANT_SOURCES=conv3x3_gen.hip MV_HIPCC_EXTRA=-DMV_GEN_TRACE python cpu-vision_amd/_build.py
    python tools/trace_conv_gen.py gentrace [more variants]
The stamps land behind the output tensor (the unsliced entry point)."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
s = torch.cuda.current_stream().cuda_stream
for variant in sys.argv[1:] or ["gentrace"]:
    lib = C.CDLL(str(ROOT / "cpu-vision_amd" / "lib" / f"libmi355vision_{variant}.so"))
    lib.mv_conv3x3_bias_relu_f32.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_int] * 5 + [C.c_void_p]
    lib.mv_last_kernel.restype = C.c_char_p
    for n, cin, cout, hw in ((1, 512, 512, 28), (1, 256, 256, 56)):
        x = torch.rand((n, cin, hw, hw), generator=g, device="cuda")
        w = torch.randn((cout, cin, 3, 3), generator=g, device="cuda") * 0.02
        b = torch.rand(cout, generator=g, device="cuda")
        ysz = n * cout * hw * hw
        y = torch.zeros(ysz + 2048, device="cuda")
        for _ in range(3):
            y[ysz:].zero_()
            assert lib.mv_conv3x3_bias_relu_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), n, cin, hw, hw, cout, 1, s) == 0
            torch.cuda.synchronize()
        t = y[ysz:ysz + 1024].view(torch.int64).cpu().reshape(8, 64)
        print(f"== {variant}: conv {cin}->{cout} @{hw} batch {n}: {lib.mv_last_kernel().decode()}")
        base = int(t[0, 0])
        cw = [int(v) - base for v in t[0] if int(v) != 0]
        lw = [int(v) - base for v in t[4] if int(v) != 0]
        print(f"  compute wave 0: start {cw[0]}, first chunk at {cw[1]}; per chunk [MFMAs issued | barrier wait]:")
        print("   ", "  ".join(f"{cw[i + 1] - cw[i]:5d}|{cw[i + 2] - cw[i + 1]:5d}" for i in range(1, min(len(cw) - 2, 41), 2)))
        print(f"  loader wave 4: start {lw[0] if lw else None}; per chunk [lstore | gload | barrier wait]:")
        print("   ", "  ".join(f"{lw[i + 1] - lw[i]:5d}|{lw[i + 2] - lw[i + 1]:5d}|{lw[i + 3] - lw[i + 2]:5d}" for i in range(1, min(len(lw) - 3, 58), 3)))
