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
    k1d_x = F2._host_taps(kernel_size[0], float(sigma[0]), True, F2._taps_dtype(img))
    k1d_y = F2._host_taps(kernel_size[1], float(sigma[1]), True, F2._taps_dtype(img))
    separable = img.is_floating_point() and F2._use_separable(kernel_size[0], kernel_size[1], img)
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


# --------------------------------------------------------------------------------------------- preset head (8f.2)
def _compute_resized_output_size(image_size, size, max_size: Optional[int] = None) -> List[int]:
    """transforms/functional.py:353-384 (same messages)."""
    h, w = image_size
    short, long = (w, h) if w <= h else (h, w)
    if len(size) == 1:  # specified size only for the smallest edge
        requested_new_short = size[0]
        new_short, new_long = requested_new_short, int(requested_new_short * long / short)
        if max_size is not None:
            if max_size <= requested_new_short:
                raise ValueError(
                    f"max_size = {max_size} must be strictly greater than the requested "
                    f"size for the smaller edge size = {size}")
            if new_long > max_size:
                new_short, new_long = int(max_size * new_short / new_long), max_size
        new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    else:  # specified both h and w
        new_w, new_h = size[1], size[0]
    return [new_h, new_w]


def _check_resize_args(img, size, interpolation, max_size, antialias):
    _assert_image_tensor(img)
    interpolation = getattr(interpolation, "value", interpolation)
    if not isinstance(interpolation, str):
        raise TypeError("Argument interpolation should be a InterpolationMode or a corresponding Pillow integer constant")
    if isinstance(size, (list, tuple)):
        if len(size) not in [1, 2]:
            raise ValueError(f"Size must be an int or a 1 or 2 element tuple/list, not a {len(size)} element tuple/list")
        if max_size is not None and len(size) != 1:
            raise ValueError("max_size should only be passed if size specifies the length of the smaller edge, "
                             "i.e. size should be an int or a sequence of length 1 in torchscript mode.")
        size = list(size)
    elif isinstance(size, int):
        size = [size]
    else:
        raise TypeError(f"Size should be int or sequence. Got {type(size)}")
    if interpolation != "bilinear" or not antialias:
        raise NotImplementedError("the MI355X resize is the preset's: interpolation=BILINEAR with antialias=True "
                                  f"(got {interpolation!r}, antialias={antialias})")
    return size


def _resize_window(img: torch.Tensor, oh: int, ow: int, top: int, left: int, ch: int, cw: int, preset=None) -> torch.Tensor:
    """Resize (..., H, W) to (oh, ow) and return the (ch, cw) window at (top, left) of the result, through
    mv_resize_bilinear_aa_* / mv_preset_classification_*."""
    from . import _lib
    _lib.require_device(img)
    if img.dtype not in (torch.uint8, torch.float32):
        raise NotImplementedError(f"resize runs on uint8 and float32 images. Got {img.dtype}")
    lib = _lib.load()
    h, w = int(img.shape[-2]), int(img.shape[-1])
    lead = tuple(img.shape[:-2])
    planes = 1
    for d in lead:
        planes *= int(d)
    u8 = img.dtype == torch.uint8
    with _lib.on_device_of(img):
        x = img.contiguous()
        out_dtype = torch.float32 if preset is not None else img.dtype
        y = torch.empty(lead + (ch, cw), dtype=out_dtype, device=img.device)
        if planes == 0:
            return y
        nbytes = int(lib.mv_resize_workspace_bytes(planes, h, w, oh, ow, top, left, ch, cw))
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=img.device)
        sp = _lib.stream_ptr(x)
        if preset is None:
            fn = lib.mv_resize_bilinear_aa_u8 if u8 else lib.mv_resize_bilinear_aa_f32
            _lib.check(fn(x.data_ptr(), y.data_ptr(), planes, h, w, oh, ow, top, left, ch, cw, ws.data_ptr(), nbytes, sp))
        else:
            mean, std = preset
            c = int(img.shape[-3])
            fn = lib.mv_preset_classification_u8 if u8 else lib.mv_preset_classification_f32
            _lib.check(fn(x.data_ptr(), y.data_ptr(), planes // c, c, h, w, oh, ow, top, left, ch, cw, _lib.taps_from_tensor(mean),
                          _lib.taps_from_tensor(std), ws.data_ptr(), nbytes, sp))
    return y


def resize(img: torch.Tensor, size: List[int], interpolation="bilinear", max_size: Optional[int] = None,
           antialias: Optional[bool] = True) -> torch.Tensor:
    """F.resize on a tensor image (transforms/functional.py:387-478 -> _functional_tensor.py:441-474): bilinear with
    antialias, the configuration every classification preset uses; uint8 results are rounded like the reference's."""
    size = _check_resize_args(img, size, interpolation, max_size, antialias)
    h, w = int(img.shape[-2]), int(img.shape[-1])
    oh, ow = _compute_resized_output_size((h, w), size, max_size)
    if [h, w] == [oh, ow]:
        return img
    return _resize_window(img, oh, ow, 0, 0, oh, ow)


def _center_crop_window(h: int, w: int, output_size):
    """Crop box of F.center_crop (transforms/functional.py:572-594) on an (h, w) image, as (top, left, ch, cw) in the
    coordinates of the UNPADDED image: a negative offset means zero padding on that side."""
    if isinstance(output_size, (int, float)):
        output_size = (int(output_size), int(output_size))
    elif isinstance(output_size, (tuple, list)) and len(output_size) == 1:
        output_size = (output_size[0], output_size[0])
    ch, cw = int(output_size[0]), int(output_size[1])
    pad_l = (cw - w) // 2 if cw > w else 0
    pad_t = (ch - h) // 2 if ch > h else 0
    pad_r = (cw - w + 1) // 2 if cw > w else 0
    pad_b = (ch - h + 1) // 2 if ch > h else 0
    ph, pw = h + pad_t + pad_b, w + pad_l + pad_r
    top = int(round((ph - ch) / 2.0)) - pad_t
    left = int(round((pw - cw) / 2.0)) - pad_l
    return top, left, ch, cw


def center_crop(img: torch.Tensor, output_size: List[int]) -> torch.Tensor:
    """F.center_crop on a tensor image: a view when the box lies inside the image (like the reference's crop), a
    zero-padded copy otherwise."""
    _assert_image_tensor(img)
    h, w = int(img.shape[-2]), int(img.shape[-1])
    top, left, ch, cw = _center_crop_window(h, w, output_size)
    if top >= 0 and left >= 0 and top + ch <= h and left + cw <= w:
        return img[..., top:top + ch, left:left + cw]
    out = img.new_zeros(tuple(img.shape[:-2]) + (ch, cw))
    y0, y1, x0, x1 = max(top, 0), min(top + ch, h), max(left, 0), min(left + cw, w)
    out[..., y0 - top:y1 - top, x0 - left:x1 - left] = img[..., y0:y1, x0:x1]
    return out


def resize_center_crop(img: torch.Tensor, size: List[int], output_size: List[int], max_size: Optional[int] = None) -> torch.Tensor:
    """center_crop(resize(img, size), output_size) computing only the rows and columns the crop keeps."""
    size = _check_resize_args(img, size, "bilinear", max_size, True)
    h, w = int(img.shape[-2]), int(img.shape[-1])
    oh, ow = _compute_resized_output_size((h, w), size, max_size)
    top, left, ch, cw = _center_crop_window(oh, ow, output_size)
    return _resize_window(img, oh, ow, top, left, ch, cw)
