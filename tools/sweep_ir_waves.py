#!/usr/bin/env python3
"""k_invres with 4 against 8 waves per workgroup on MobileNetV2's 28 / 14 / 7-pixel blocks, interleaved in one process through the
tuning build's MV_IR_WAVES knob; the two must agree bit for bit (GPU box).

    python tools/sweep_ir_waves.py [--batch 64] [--rounds 7]
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from cpu_vision_amd import _lib, mobilenet as M  # noqa: E402
from tools.perf_invres import graph_time  # noqa: E402

BLOCKS = [(32, 32, 28, 1), (32, 64, 28, 2), (64, 64, 14, 1), (64, 96, 14, 1), (96, 96, 14, 1), (96, 160, 14, 2), (160, 160, 7, 1), (160, 320, 7, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=7)
    a = ap.parse_args()
    torch.manual_seed(0)
    with torch.no_grad(), _lib.tuning_library():
        spin = torch.rand((64, 64, 224, 224), device="cuda")
        for _ in range(50):
            spin.mul_(1.0)
        for cin, cout, side, stride in BLOCKS:
            blk = M.InvertedResidual(cin, cout, stride, 6).cuda().eval()
            x = torch.rand((a.batch, cin, side, side), device="cuda") * 2 - 1
            line, ref = f"{cin:3d}->{6 * cin:3d}->{cout:3d} @{side:2d} s{stride} batch {a.batch}:", None
            for waves in ("4", "8"):
                os.environ["MV_IR_WAVES"] = waves
                y = blk(x)
                kern = _lib.last_kernel()
                if ref is None:
                    ref = y.clone()
                assert torch.equal(ref, y), f"{waves} waves change the result ({kern})"
                t = graph_time(lambda: blk(x), a.rounds)
                line += f"   {waves} waves {t * 1e3:6.1f} us"
            os.environ.pop("MV_IR_WAVES", None)
            print(line + f"   {kern}", flush=True)


if __name__ == "__main__":
    main()
