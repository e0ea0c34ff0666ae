#!/usr/bin/env python3
"""uint8 KxK blur: strip height sweep (MV_DWK_U8_ROWS) on 32 x 4K uint8 frames."""
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
for rows in (16, 32, 64, 128, 256, 540):
    os.environ["MV_DWK_U8_ROWS"] = str(rows)
    line = f"rows={rows:4d}"
    for ks in ([5, 5], [7, 7], [3, 5], [7, 3]):
        ms, _ = timeit(lambda: F.gaussian_blur(xu, ks), 7)
        line += f"  {ks[0]}x{ks[1]}: {ms:6.3f} ms"
    print(line, flush=True)
