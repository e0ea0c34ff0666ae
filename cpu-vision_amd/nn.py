"""The first layer of the reference's small CNNs as a module on the gfx950 implicit-GEMM kernel.

  make_layers: nn.Conv2d(in, v, kernel_size=3, padding=1) + nn.ReLU(inplace=True)   models/vgg.py:73-87
  VGG init: kaiming_normal_(fan_out, relu), bias 0                                   models/vgg.py:52-57
  Conv2dNormActivation: ONE class, `mobilenet.Conv2dNormActivation` (the reference's constructor and defaults,
  ops/misc.py:68-172), re-exported here; with norm_layer=None on a 3x3 / stride 1 conv it runs this module's fused kernel.

Inference only (forward).  VGG / AlexNet keep the reference's module tree (torch containers hold the parameters), so the
reference's state dicts load with plain `load_state_dict`; Conv3x3ReLU / VGGFeatures are the fused single-layer modules.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import functional as F
from ._lib import forward_only as _forward_only
from .mobilenet import Conv2dNormActivation  # noqa: F401  (the one definition; see the module docstring)


class Conv3x3ReLU(nn.Module):
    """nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.ReLU(inplace=True)) as one fused kernel."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, relu: bool = True,
                 init_weights: bool = True) -> None:
        super().__init__()
        self.in_channels, self.out_channels, self.relu = in_channels, out_channels, relu
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, 3, 3))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        if init_weights:  # models/vgg.py:54-57
            nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
            if self.bias is not None:
                nn.init.constant_(self.bias, 0)
        else:  # nn.Conv2d's default reset_parameters
            nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
            if self.bias is not None:
                bound = 1 / (in_channels * 9) ** 0.5
                nn.init.uniform_(self.bias, -bound, bound)

    @classmethod
    def from_conv(cls, conv: nn.Conv2d, relu: bool = True) -> "Conv3x3ReLU":
        if conv.kernel_size != (3, 3) or conv.padding != (1, 1) or conv.stride != (1, 1) or conv.dilation != (1, 1) \
                or conv.groups != 1:
            raise ValueError("only Conv2d(kernel_size=3, stride=1, padding=1, dilation=1, groups=1) maps to this kernel")
        m = cls(conv.in_channels, conv.out_channels, bias=conv.bias is not None, relu=relu, init_weights=False)
        with torch.no_grad():
            m.weight.copy_(conv.weight)
            if conv.bias is not None:
                m.bias.copy_(conv.bias)
        return m.to(conv.weight.device)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.conv2d_bias_relu(x, self.weight, self.bias, relu=self.relu)

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}, kernel_size=(3, 3), padding=(1, 1), relu={self.relu}"


# --------------------------------------------------------------------------------------------- VGG feature extractor (8f.1)
VGG_CFGS = {  # models/vgg.py:90-95
    "A": [64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "B": [64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"],
    "D": [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"],
    "E": [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"],
}


class MaxPool2x2(nn.Module):
    """nn.MaxPool2d(kernel_size=2, stride=2) (models/vgg.py:78-79)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.max_pool2d_2x2(x)


class VGGFeatures(nn.Module):
    """`make_layers(cfg, batch_norm=False)` (models/vgg.py:73-87) on the MI355X kernels: every
    Conv2d(in, v, 3, padding=1) + ReLU(inplace) pair is one fused launch, MaxPool2d(2, 2) one launch.
    `ref_index` keeps the position each layer has in the reference's nn.Sequential, so reference state dicts
    (`features.<idx>.weight`) load unchanged."""

    def __init__(self, cfg: str = "A", in_channels: int = 3) -> None:
        super().__init__()
        self.layers = nn.ModuleList()
        self.ref_index = []  # index of each fused layer's FIRST module in the reference Sequential
        idx = 0
        for v in VGG_CFGS[cfg]:
            if v == "M":
                self.layers.append(MaxPool2x2())
                self.ref_index.append(idx)
                idx += 1
            else:
                self.layers.append(Conv3x3ReLU(in_channels, int(v), bias=True, relu=True, init_weights=True))
                self.ref_index.append(idx)
                idx += 2  # conv + relu
                in_channels = int(v)
        self.ref_len = idx

    def load_reference_state_dict(self, state) -> None:
        with torch.no_grad():
            for layer, idx in zip(self.layers, self.ref_index):
                if isinstance(layer, Conv3x3ReLU):
                    layer.weight.copy_(state[f"features.{idx}.weight"])
                    layer.bias.copy_(state[f"features.{idx}.bias"])

    def run_prefix(self, x: torch.Tensor, ref_stop: int) -> torch.Tensor:
        """features[0:ref_stop] in the reference's indexing (ref_stop must fall on a fused-layer boundary)."""
        for layer, idx in zip(self.layers, self.ref_index):
            if idx >= ref_stop:
                break
            x = layer(x)
        return x

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.run_prefix(x, self.ref_len)


def vgg11_reference_init(num_classes: int = 1000, seed: int = 0, cfg: str = "A"):
    """State dict of `vgg11(num_classes=...)` built with the reference's constructor sequence (models/vgg.py:36-63,
    73-87) on the host RNG, so that torch.manual_seed(seed) yields the reference's weights bit for bit without
    importing it: Conv2d / Linear default inits in construction order, then kaiming_normal_(fan_out, relu) + zero
    bias for every conv and normal_(0, 0.01) + zero bias for every linear, in module order."""
    torch.manual_seed(seed)
    convs, idxs, in_ch, idx = [], [], 3, 0
    for v in VGG_CFGS[cfg]:
        if v == "M":
            idx += 1
        else:
            convs.append(nn.Conv2d(in_ch, int(v), kernel_size=3, padding=1))
            idxs.append(idx)
            idx += 2
            in_ch = int(v)
    linears = [nn.Linear(512 * 7 * 7, 4096), nn.Linear(4096, 4096), nn.Linear(4096, num_classes)]
    for m in convs:
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(m.bias, 0)
    for m in linears:
        nn.init.normal_(m.weight, 0, 0.01)
        nn.init.constant_(m.bias, 0)
    state = {}
    for m, i in zip(convs, idxs):
        state[f"features.{i}.weight"] = m.weight.detach()
        state[f"features.{i}.bias"] = m.bias.detach()
    for m, i in zip(linears, (0, 3, 6)):
        state[f"classifier.{i}.weight"] = m.weight.detach()
        state[f"classifier.{i}.bias"] = m.bias.detach()
    return state


class LinearReLU(nn.Module):
    """nn.Linear(in, out) [+ nn.ReLU(True)] as one fused launch (classifier of models/vgg.py:42-50)."""

    def __init__(self, in_features: int, out_features: int, relu: bool = False) -> None:
        super().__init__()
        self.in_features, self.out_features, self.relu = in_features, out_features, relu
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.zeros(out_features))
        nn.init.normal_(self.weight, 0, 0.01)  # models/vgg.py:61-63

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.linear_bias_relu(x, self.weight, self.bias, relu=self.relu)


def _run_conv_stack(mods, x: torch.Tensor) -> torch.Tensor:
    """Walk a reference-style nn.Sequential of Conv2d / ReLU / MaxPool2d containers on the gfx950 kernels: every
    conv (+ the ReLU behind it) is one fused launch, every pooling layer one launch."""
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            if m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.dilation == (1, 1) and m.groups == 1:
                x = F.conv2d_bias_relu(x, m.weight, m.bias, relu=relu)
            else:
                x = F.conv2d_bias_act(x, m.weight, m.bias, stride=m.stride, padding=m.padding, activation="relu" if relu else None)
            i += 2 if relu else 1
        elif isinstance(m, nn.MaxPool2d):
            ks = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
            st = m.stride if isinstance(m.stride, int) else m.stride[0]
            x = F.max_pool2d_2x2(x) if (ks, st) == (2, 2) else F.max_pool2d(x, ks, st)
            i += 1
        else:
            raise NotImplementedError(type(m).__name__)
    return x


class VGG(nn.Module):
    """models.vgg.VGG (vgg.py:35-70) with the REFERENCE'S MODULE TREE: `features` is make_layers' nn.Sequential of
    Conv2d / ReLU / MaxPool2d (vgg.py:73-87), `avgpool` AdaptiveAvgPool2d((7, 7)), `classifier` the Sequential of
    Linear / ReLU / Dropout (vgg.py:42-50) -- torch containers that hold the parameters, so a reference state dict
    (`features.0.weight`, `classifier.6.bias`, ...) loads with plain `load_state_dict`, and a seeded construction draws
    from the RNG in the reference's order (weights equal bit for bit).  The forward never calls those containers: it walks
    them on the gfx950 kernels (conv + ReLU fused, MaxPool, AdaptiveAvgPool, Linear + ReLU fused).  Inference only."""

    def __init__(self, cfg: str = "A", num_classes: int = 1000, init_weights: bool = True, dropout: float = 0.5) -> None:
        super().__init__()
        layers, in_channels = [], 3
        for v in VGG_CFGS[cfg]:
            if v == "M":
                layers += [nn.MaxPool2d(kernel_size=2, stride=2)]
            else:
                layers += [nn.Conv2d(in_channels, int(v), kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                in_channels = int(v)
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        self.classifier = nn.Sequential(
            nn.Linear(512 * 7 * 7, 4096), nn.ReLU(True), nn.Dropout(p=dropout),
            nn.Linear(4096, 4096), nn.ReLU(True), nn.Dropout(p=dropout),
            nn.Linear(4096, num_classes),
        )
        if init_weights:  # vgg.py:52-63
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)
                elif isinstance(m, nn.Linear):
                    nn.init.normal_(m.weight, 0, 0.01)
                    nn.init.constant_(m.bias, 0)
        self.eval()

    def load_reference_state_dict(self, state) -> None:
        """Kept for callers of round 1: a reference state dict is this module's own state dict."""
        self.load_state_dict(state)

    def run_features(self, x: torch.Tensor, stop: Optional[int] = None) -> torch.Tensor:
        """features[0:stop] in the reference's indexing (stop must not split a conv from its ReLU)."""
        return _run_conv_stack(list(self.features)[:stop], x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise RuntimeError("the MI355X VGG is inference-only (Dropout is the identity): call .eval()")
        inp = x
        with torch.no_grad():  # no autograd bookkeeping per layer; the result is marked once (its backward raises)
            x = self.run_features(x)
            if tuple(x.shape[-2:]) != (7, 7):  # windows of one element average to the element itself (224 x 224 inputs): no launch
                x = F.adaptive_avg_pool2d(x, (7, 7))
            x = torch.flatten(x, 1)
            c = self.classifier
            x = F.linear_bias_relu(x, c[0].weight, c[0].bias, relu=True)
            x = F.linear_bias_relu(x, c[3].weight, c[3].bias, relu=True)
            x = F.linear_bias_relu(x, c[6].weight, c[6].bias, relu=False)
        return _forward_only(x, "VGG.forward", inp, self.classifier[6].weight)


def vgg11(num_classes: int = 1000) -> VGG:
    return VGG("A", num_classes)


# --------------------------------------------------------------------------------------------- AlexNet (models/alexnet.py:17-52)
class AlexNet(nn.Module):
    """models/alexnet.py:17-52 with the reference's module tree (torch containers hold the parameters; state dicts load as
    they are).  conv1 (11x11 stride 4) and conv2 (5x5) run as im2col + fp32 MFMA GEMM, conv3-5 on the K-chunked 3x3 MFMA
    kernel, each with its bias + ReLU fused; MaxPool2d(3, 2), AdaptiveAvgPool2d((6, 6)) and the classifier's Linear+ReLU on
    their own kernels.  Inference only (Dropout is the identity)."""

    def __init__(self, num_classes: int = 1000, dropout: float = 0.5) -> None:
        super().__init__()
        self.features = nn.Sequential(
            nn.Conv2d(3, 64, kernel_size=11, stride=4, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(64, 192, kernel_size=5, padding=2), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(192, 384, kernel_size=3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(384, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(256, 256, kernel_size=3, padding=1), nn.ReLU(inplace=True), nn.MaxPool2d(kernel_size=3, stride=2),
        )
        self.avgpool = nn.AdaptiveAvgPool2d((6, 6))
        self.classifier = nn.Sequential(
            nn.Dropout(p=dropout), nn.Linear(256 * 6 * 6, 4096), nn.ReLU(inplace=True),
            nn.Dropout(p=dropout), nn.Linear(4096, 4096), nn.ReLU(inplace=True),
            nn.Linear(4096, num_classes),
        )
        self.eval()

    def run_features(self, x: torch.Tensor, stop: Optional[int] = None) -> torch.Tensor:
        return _run_conv_stack(list(self.features)[:stop], x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise RuntimeError("the MI355X AlexNet is inference only: call .eval()")
        inp = x
        with torch.no_grad():
            x = self.run_features(x)
            if tuple(x.shape[-2:]) != (6, 6):  # windows of one element average to the element itself (224 x 224 inputs): no launch
                x = F.adaptive_avg_pool2d(x, (6, 6))
            x = torch.flatten(x, 1)
            c = self.classifier
            x = F.linear_bias_relu(x, c[1].weight, c[1].bias, relu=True)
            x = F.linear_bias_relu(x, c[4].weight, c[4].bias, relu=True)
            x = F.linear_bias_relu(x, c[6].weight, c[6].bias, relu=False)
        return _forward_only(x, "AlexNet.forward", inp, self.classifier[6].weight)


def alexnet(num_classes: int = 1000) -> AlexNet:
    return AlexNet(num_classes)
