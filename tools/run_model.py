#!/usr/bin/env python3
"""Run one small CNN forward a few times (target program for `rocprofv3 --kernel-trace --stats`).

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_alexnet_b1 -- python3 tools/run_model.py --model alexnet --batch 1
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from cpu_vision_amd.mobilenet import MobileNetV2  # noqa: E402
from cpu_vision_amd.nn import alexnet, vgg11  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="alexnet", choices=["alexnet", "vgg11", "mobilenet_v2"])
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
torch.manual_seed(0)
model = {"alexnet": alexnet, "vgg11": vgg11, "mobilenet_v2": MobileNetV2}[a.model](1000).cuda().eval()
x = torch.rand((a.batch, 3, 224, 224), device="cuda")
for _ in range(a.iters):
    y = model(x)
torch.cuda.synchronize()
print(a.model, a.batch, float(y.sum()))
