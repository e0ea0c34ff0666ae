#!/usr/bin/env python3
"""uint8 Gaussian blur, 5x5 and 7x7 on 32 x 4K: the plain 2-D pass (the product's choice up to 49 taps) against pair + tie check + fix-up
(forced through the tuning build's MV_U8_HYBRID_MIN_TAPS), same bits required.  GPU box."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402
from tools.perf_invres import graph_time  # noqa: E402

x = torch.randint(0, 256, (32, 3, 2160, 3840), dtype=torch.uint8, device="cuda")
with _lib.tuning_library():
    for k, s in ((5, 1.1), (7, 1.4), (9, 1.7)):
        line, ref = f"{k} x {k}:", None
        for taps in ("49", "8"):
            os.environ["MV_U8_HYBRID_MIN_TAPS"] = taps
            y = F.gaussian_blur(x, [k, k], [s, s])
            kern = _lib.last_kernel()
            if ref is None:
                ref = y
            assert torch.equal(ref, y), "the two exact paths disagree"
            t = graph_time(lambda: F.gaussian_blur(x, [k, k], [s, s]), 5, inner=3)
            line += f"   min_taps {taps}: {t:.4f} ms ({kern})"
        print(line, flush=True)
    os.environ.pop("MV_U8_HYBRID_MIN_TAPS", None)
