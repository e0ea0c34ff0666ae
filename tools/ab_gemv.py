#!/usr/bin/env python3
"""k_linear_gemv stage size A/B (64 against 128 k per stage) on the classifier layers at batch 1 and 4, interleaved, through the
tuning build's MV_GEMV_K128 knob (GPU box).  GPU time per call = 10 calls captured in a HIP graph."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import _lib, functional as F  # noqa: E402
from tools.perf_invres import graph_time  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
with _lib.tuning_library():
    for k, m in ((25088, 4096), (9216, 4096), (4096, 4096), (4096, 1000)):
        w = (torch.rand((m, k), generator=g, device="cuda") - 0.5) * 0.02
        b = torch.rand((m,), generator=g, device="cuda")
        for n in (1, 4):
            x = torch.rand((n, k), generator=g, device="cuda")
            ref = None
            line = f"linear {k:5d}->{m:4d} batch {n}: "
            for knob in (None, "1"):
                if knob:
                    os.environ["MV_GEMV_K128"] = knob
                y = F.linear_bias_relu(x, w, b, relu=True)
                if ref is None:
                    ref = y.clone()
                assert torch.equal(ref, y)
                t = graph_time(lambda: F.linear_bias_relu(x, w, b, relu=True), 7)
                os.environ.pop("MV_GEMV_K128", None)
                line += f"  stage {'128' if knob else ' 64'}: {t * 1e3:7.1f} us ({w.numel() * 4 / t / 1e6:6.0f} GB/s)"
            print(line, flush=True)
