#!/usr/bin/env python3
"""uint8 3x3 blur / sharpness: strip height sweep (MV_DW3X3_U8_ROWS) on 32 x 4K uint8 frames."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
# knob sweeps run on the -DMV_TUNING build of the same sources (the product library reads no environment variable)
os.environ.setdefault("MI355VISION_LIB", str(Path(__file__).resolve().parent.parent / "cpu-vision_amd" / "lib" / "libmi355vision_tuning.so"))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402
from tools.perf_configs import timeit  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
xu = torch.randint(0, 256, (32, 3, 2160, 3840), generator=g, device="cuda", dtype=torch.uint8)
for rows in (0, 8, 16, 32, 0, 64, 32, 8):
    os.environ["MV_DW3X3_U8_ROWS"] = str(rows)  # 0 = the library's default
    a, _ = timeit(lambda: F.gaussian_blur(xu, [3, 3]), 9)
    b, _ = timeit(lambda: F.adjust_sharpness(xu, 1.5), 9)
    print(f"rows={rows:4d}  blur3 {a:6.3f} ms   sharpness {b:6.3f} ms", flush=True)
