#!/usr/bin/env python3
"""Large-kernel separable blur sweep on one MI355X: ms, GB/s and fp32 FMA rate per kernel size (32 x 4K frames)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd import functional as F  # noqa: E402
from tools.perf_configs import timeit  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand((32, 3, 2160, 3840), generator=g, device="cuda")
    xu = torch.randint(0, 256, (32, 3, 2160, 3840), generator=g, device="cuda", dtype=torch.uint8)
    for k in (9, 11, 15, 17, 23, 27, 31, 33, 41, 47, 55, 63):
        for name, t in (("f32", x), ("u8", xu)):
            ms, mn = timeit(lambda: F.gaussian_blur(t, [k, k]), 7)
            el = t.numel()
            by = el * (8 if name == "f32" else 2)
            print(f"K={k:2d} {name:3s} {ms:8.3f} ms  {by / ms / 1e6:8.1f} GB/s  {2 * k * el / ms / 1e9:7.2f} TFMA/s(real taps)", flush=True)


if __name__ == "__main__":
    main()
