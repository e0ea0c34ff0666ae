#!/usr/bin/env python3
"""The variants of k_invres_wide (csrc/invres.hip: output rows per region x waves per workgroup x register bound) on MobileNetV2's
three wide blocks, interleaved in one process through the tuning build's MV_IRW_VARIANT knob (GPU box).

    python tools/sweep_irw.py [--batch 64] [--rounds 7]
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

BLOCKS = [(32, 16, 112, 1, 4), (16, 24, 112, 2, 5), (24, 24, 56, 1, 6), (24, 32, 56, 2, 7)]  # cin, cout, side, stride, variants


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
        for cin, cout, side, stride, nvar in BLOCKS:
            t = 1 if cin == 32 else 6
            blk = M.InvertedResidual(cin, cout, stride, t).cuda().eval()
            x = torch.rand((a.batch, cin, side, side), device="cuda") * 2 - 1
            M.FUSE_INVERTED_RESIDUAL = False
            t3 = graph_time(lambda: blk(x), a.rounds)
            M.FUSE_INVERTED_RESIDUAL = True
            print(f"{cin}->{t * cin}->{cout} @{side} s{stride} batch {a.batch}: three launches {t3 * 1e3:7.1f} us")
            ref = None
            for v in range(nvar):
                os.environ["MV_IRW_VARIANT"] = str(v)
                y = blk(x)
                kern = _lib.last_kernel()
                if ref is None:
                    ref = y.clone()
                else:
                    same = torch.equal(ref, y)  # variants with another slice plan sum the projection in another order
                    assert same or ("slices1>" not in kern and torch.allclose(ref, y, rtol=1e-5, atol=1e-5)), f"variant {v} changes the result"
                t = graph_time(lambda: blk(x), a.rounds)
                print(f"   variant {v}: {t * 1e3:7.1f} us   {kern}")
            os.environ.pop("MV_IRW_VARIANT", None)


if __name__ == "__main__":
    main()
