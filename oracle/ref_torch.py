"""The reference's CPU call sequence for the hot path, restated over torch CPU ops.

TEST INFRASTRUCTURE ONLY (see oracle.c).  The reference has no filter arithmetic
of its own; on CPU it executes ``torch.nn.functional.pad(mode="reflect")`` +
``torch.nn.functional.conv2d(groups=C)`` (third-party ATen / oneDNN).  This file
issues exactly those torch calls in the reference's order, so on the same torch
build it reproduces the reference bit for bit (checked against the fixtures in
tests/golden/, which were produced by importing the reference itself).  It is
what ``bench.py`` times as ``cpu_baseline`` (kind "port": the reference's own
source files do not travel to the GPU box) and a second checker beside oracle.c.

Nothing in cpu-vision_amd/ imports this file.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as TF


def gaussian_kernel1d(k: int, sigma: float, dtype=torch.float32, v1: bool = False) -> torch.Tensor:
    """v2: transforms/v2/functional/_misc.py:86-90; v1: transforms/_functional_tensor.py:727-734."""
    if v1:
        half = (k - 1) * 0.5
        x = torch.linspace(-half, half, steps=k, dtype=dtype)
        pdf = torch.exp(-0.5 * (x / sigma).pow(2))
        return pdf / pdf.sum()
    lim = (k - 1) / (2.0 * math.sqrt(2.0))
    x = torch.linspace(-lim, lim, steps=k, dtype=dtype)
    return torch.softmax(x.div(sigma).pow(2).neg(), dim=0)


def gaussian_kernel2d(kernel_size: Sequence[int], sigma: Sequence[float], dtype=torch.float32, v1: bool = False):
    """_misc.py:93-99 (outer product; v1 :737-743 uses mm -- same values)."""
    kx = gaussian_kernel1d(kernel_size[0], sigma[0], dtype, v1)
    ky = gaussian_kernel1d(kernel_size[1], sigma[1], dtype, v1)
    return torch.mm(ky[:, None], kx[None, :]) if v1 else ky.unsqueeze(-1) * kx


def depthwise(x4: torch.Tensor, k2d: torch.Tensor, border: str) -> torch.Tensor:
    """The reference primitive: [pad] + conv2d(groups=C).  x4 is (B,C,H,W)."""
    c = x4.shape[1]
    ky, kx = k2d.shape
    w = k2d.expand(c, 1, ky, kx)
    if border == "reflect":
        x4 = TF.pad(x4, [kx // 2, kx // 2, ky // 2, ky // 2], mode="reflect")
    elif border == "zero":
        x4 = TF.pad(x4, [kx // 2, kx // 2, ky // 2, ky // 2])
    elif border != "valid":
        raise ValueError(border)
    return TF.conv2d(x4, w, groups=c)


def gaussian_blur_image(image: torch.Tensor, kernel_size: List[int], sigma: Optional[List[float]] = None,
                        v1: bool = False) -> torch.Tensor:
    """Call sequence of gaussian_blur_image (_misc.py:138-163); arguments already normalised."""
    if sigma is None:
        sigma = [k * 0.15 + 0.35 for k in kernel_size]
    if image.numel() == 0:
        return image
    shape, dtype = image.shape, image.dtype
    x = image.reshape((-1,) + tuple(shape[-3:]))
    fp = x.is_floating_point()
    k2d = gaussian_kernel2d(kernel_size, sigma, dtype if fp else torch.float32, v1)
    out = depthwise(x if fp else x.to(torch.float32), k2d, "reflect")
    out = out.reshape(shape)
    if not fp:
        out = out.round_().to(dtype)
    return out


def separable_blur(image: torch.Tensor, kernel_size: List[int], sigma: List[float]) -> torch.Tensor:
    """Row pass then column pass of the same primitive (cfg3's 'separable 5x5')."""
    shape = image.shape
    x = image.reshape((-1,) + tuple(shape[-3:]))
    kx = gaussian_kernel1d(kernel_size[0], sigma[0], x.dtype)
    ky = gaussian_kernel1d(kernel_size[1], sigma[1], x.dtype)
    t = depthwise(x, kx[None, :], "reflect")
    return depthwise(t, ky[:, None], "reflect").reshape(shape)


SOBEL_GX = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
SOBEL_GY = SOBEL_GX.t().contiguous()


def sobel(image: torch.Tensor, border: str = "reflect"):
    shape = image.shape
    x = image.reshape((-1,) + tuple(shape[-3:]))
    gx = depthwise(x, SOBEL_GX.to(x.dtype), border)
    gy = depthwise(x, SOBEL_GY.to(x.dtype), border)
    oshape = tuple(shape[:-2]) + tuple(gx.shape[-2:])
    return gx.reshape(oshape), gy.reshape(oshape)


def adjust_sharpness_image(image: torch.Tensor, sharpness_factor: float) -> torch.Tensor:
    """Call sequence of adjust_sharpness_image (v2, _color.py:242-280)."""
    c, h, w = image.shape[-3:]
    if image.numel() == 0 or h <= 2 or w <= 2:
        return image
    fp = image.is_floating_point()
    bound = 1.0 if fp else 255.0
    shape = image.shape
    x = image.reshape(-1, c, h, w)
    kdt = x.dtype if fp else torch.float32
    a, b = 1.0 / 13.0, 5.0 / 13.0
    k = torch.tensor([[a, a, a], [a, b, a], [a, a, a]], dtype=kdt)
    out = x.to(dtype=kdt, copy=True)
    blurred = depthwise(out, k, "valid")
    if not fp:
        blurred = blurred.round_()
    view = out[..., 1:-1, 1:-1]
    view.add_(blurred.sub_(view), alpha=(1.0 - sharpness_factor))
    out = out.clamp_(0, bound)
    if not fp:
        out = out.to(image.dtype)
    return out.reshape(shape)


def conv3x3_bias_relu(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], relu: bool = True) -> torch.Tensor:
    """nn.Conv2d(Cin,Cout,3,padding=1) + nn.ReLU  (models/vgg.py:81-85)."""
    y = TF.conv2d(x, w, b, padding=1)
    return torch.relu_(y) if relu else y
