#!/usr/bin/env python3
"""Writes tests/golden/k_slice_plans.json: the summation-order plans (mv_conv3x3_k_slices, mv_linear_k_slices,
mv_conv1x1_k_slices, mv_inverted_residual_k_slices -- host logic, no GPU) the library states for every VGG-11 / AlexNet / MobileNetV2 layer shape at
batch 1, 4, 8 and 64.

Why a fixture: the GPU parity tests ask the library for its plan and have the oracle restate it, so a change that moves the
plan and the kernel together would still pass them.  tests/test_host_logic.py::test_k_slice_plans_are_pinned compares the
library's answers with this file; changing a plan on purpose means re-running this script in the same commit and saying so
in DESIGN.md (the end-to-end fixtures vgg11_forward.npz / mobilenet_v2.npz bound the numerical effect at 1e-5).
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from cpu_vision_amd import functional as F  # noqa: E402

BATCHES = (1, 4, 8, 64)


def vgg11_convs():  # models/vgg.py:90 cfg "A": (cin, cout, map side)
    return [(3, 64, 224), (64, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28), (512, 512, 28), (512, 512, 14)]


def alexnet_convs3x3():  # models/alexnet.py:28-33
    return [(192, 384, 13), (384, 256, 13), (256, 256, 13)]


def linears():  # vgg.py:42-50, alexnet.py:37-45 (1000 and the tests' 50 classes)
    return [(25088, 4096), (4096, 4096), (4096, 1000), (4096, 50), (9216, 4096)]


def mobilenet_pointwise():  # models/mobilenetv2.py:88-121 (width 1.0, 224 x 224 input)
    setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]
    out, inp, side = [], 32, 112
    for t, c, reps, s in setting:
        for i in range(reps):
            stride = s if i == 0 else 1
            hidden = inp * t
            if t != 1:
                out.append((inp, hidden, side))
            side //= stride
            out.append((hidden, c, side))
            inp = c
    out.append((320, 1280, 7))
    return sorted(set(out))


def mobilenet_blocks():  # InvertedResidual blocks with expansion: (cin, cout, input side, stride)
    setting = [[6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]
    out, inp, side = [], 16, 112
    for t, c, reps, s in setting:
        for i in range(reps):
            stride = s if i == 0 else 1
            out.append((inp, c, side, stride))
            side //= stride
            inp = c
    return sorted(set(out))


def plans():
    p = {"conv3x3": {}, "linear": {}, "conv1x1": {}, "inverted_residual": {}}
    for n in BATCHES:
        for cin, cout, s in vgg11_convs() + alexnet_convs3x3():
            p["conv3x3"][f"{n},{cin},{s},{s},{cout}"] = list(F.conv3x3_k_slices(n, cin, s, s, cout))
        for k, m in linears():
            p["linear"][f"{n},{k},{m}"] = list(F.linear_k_slices(n, k, m))
        for cin, cout, s in mobilenet_pointwise():
            p["conv1x1"][f"{n},{cin},{s},{s},{cout}"] = list(F.conv1x1_k_slices(n, cin, s, s, cout))
        for cin, cout, s, stride in mobilenet_blocks():  # 0 slices = no fused kernel for the shape (three launches)
            p["inverted_residual"][f"{n},{cin},{6 * cin},{cout},{s},{s},{stride}"] = list(F.inverted_residual_k_slices(n, cin, 6 * cin, cout, s, s, stride))
        p["inverted_residual"][f"{n},32,32,16,112,112,1"] = list(F.inverted_residual_k_slices(n, 32, 32, 16, 112, 112, 1))  # the first block: no expansion
    return p


if __name__ == "__main__":
    out = Path(__file__).with_name("k_slice_plans.json")
    out.write_text(json.dumps(plans(), indent=0, sort_keys=True) + "\n")
    print(f"wrote {out}: {sum(len(v) for v in plans().values())} plans")
