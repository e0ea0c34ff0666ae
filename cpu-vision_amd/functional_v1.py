"""v1 functional API of the hot path (the reference's torchvision.transforms.functional /
_functional_tensor), on the same gfx950 kernels.

  gaussian_blur      transforms/functional.py:1318-1384  ->  _functional_tensor.py:746-764
  adjust_sharpness   transforms/functional.py:1451-1470  ->  _functional_tensor.py:809-838, 258-261
  _get_gaussian_kernel1d / 2d                                _functional_tensor.py:727-743

v1 differs from v2 in the tap formula (exp/sum instead of softmax), in accepting only (C,H,W) / (B,C,H,W)
and in the sharpness blend (f*x + (1-f)*blur, two roundings, applied to the border too).
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import functional as F2


def _get_gaussian_kernel1d(kernel_size: int, sigma: float, dtype: torch.dtype = torch.float32, device="cpu") -> torch.Tensor:
    ksize_half = (kernel_size - 1) * 0.5
    x = torch.linspace(-ksize_half, ksize_half, steps=kernel_size, dtype=dtype, device=device)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    return pdf / pdf.sum()


def _get_gaussian_kernel2d(kernel_size: List[int], sigma: List[float], dtype: torch.dtype = torch.float32, device="cpu"):
    kernel1d_x = _get_gaussian_kernel1d(kernel_size[0], sigma[0], dtype, device)
    kernel1d_y = _get_gaussian_kernel1d(kernel_size[1], sigma[1], dtype, device)
    return torch.mm(kernel1d_y[:, None], kernel1d_x[None, :])


def _assert_image_tensor(img: torch.Tensor) -> None:
    if not (isinstance(img, torch.Tensor) and img.ndim >= 2):
        raise TypeError("Tensor is not a torch image.")


def gaussian_blur(img: torch.Tensor, kernel_size: List[int], sigma: Optional[List[float]] = None) -> torch.Tensor:
    if not isinstance(kernel_size, (int, list, tuple)):
        raise TypeError(f"kernel_size should be int or a sequence of integers. Got {type(kernel_size)}")
    if isinstance(kernel_size, int):
        kernel_size = [kernel_size, kernel_size]
    if len(kernel_size) != 2:
        raise ValueError(f"If kernel_size is a sequence its length should be 2. Got {len(kernel_size)}")
    for ksize in kernel_size:
        if ksize % 2 == 0 or ksize < 0:
            raise ValueError(f"kernel_size should have odd and positive integers. Got {kernel_size}")
    if sigma is None:
        sigma = [ksize * 0.15 + 0.35 for ksize in kernel_size]
    if sigma is not None and not isinstance(sigma, (int, float, list, tuple)):
        raise TypeError(f"sigma should be either float or sequence of floats. Got {type(sigma)}")
    if isinstance(sigma, (int, float)):
        sigma = [float(sigma), float(sigma)]
    if isinstance(sigma, (list, tuple)) and len(sigma) == 1:
        sigma = [sigma[0], sigma[0]]
    if len(sigma) != 2:
        raise ValueError(f"If sigma is a sequence, its length should be 2. Got {len(sigma)}")
    for s in sigma:
        if s <= 0.0:
            raise ValueError(f"sigma should have positive values. Got {sigma}")
    if not isinstance(img, torch.Tensor):
        raise TypeError(f"img should be Tensor. Got {type(img)}")
    _assert_image_tensor(img)
    img.shape[-3]  # noqa: B018 -- v1 indexes shape[-3] too (_functional_tensor.py:754)
    if img.ndim > 4:
        raise NotImplementedError("the v1 tensor backend takes (C, H, W) or (B, C, H, W) images")
    if img.numel() == 0:
        return img
    k1d_x = F2._host_taps(kernel_size[0], float(sigma[0]), True)
    k1d_y = F2._host_taps(kernel_size[1], float(sigma[1]), True)
    separable = img.is_floating_point() and kernel_size[0] * kernel_size[1] > F2._DIRECT_2D_MAX_TAPS
    return F2._blur_with_taps(img, k1d_x, k1d_y, separable)


def adjust_sharpness(img: torch.Tensor, sharpness_factor: float) -> torch.Tensor:
    if sharpness_factor < 0:
        raise ValueError(f"sharpness_factor ({sharpness_factor}) is not non-negative.")
    _assert_image_tensor(img)
    c = 1 if img.ndim == 2 else int(img.shape[-3])
    if c not in (1, 3):
        raise TypeError(f"Input image tensor permitted channel values are {[1, 3]}, but found {c}")
    if img.size(-1) <= 2 or img.size(-2) <= 2:
        return img
    if img.numel() == 0:
        return img
    return F2._sharpness(img, sharpness_factor, v1=True)
