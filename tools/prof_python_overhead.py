#!/usr/bin/env python3
"""Host-side cost of a MobileNetV2 forward at batch 1 (53 launches through the Python layer): wall time per forward and cProfile's
top functions (runs on the GPU box)."""
import cProfile, pstats, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from cpu_vision_amd import mobilenet as M, functional as F
torch.manual_seed(0)
net = M.MobileNetV2(1000).cuda().eval()
x = torch.rand((1, 3, 224, 224), device="cuda")
for _ in range(3):
    net(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    net(x)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host time per forward (53 launches): {(t1 - t0) / 20 * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    net(x)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
