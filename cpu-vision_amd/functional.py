"""Functional API of the hot path -- same names, signatures, defaults, validation and error text as the
reference's torchvision.transforms.v2.functional for this path, computed by libmi355vision.so.

Reference functions mirrored (paths relative to the reference root):
  gaussian_blur / gaussian_blur_image / gaussian_blur_video   transforms/v2/functional/_misc.py:75-181
  _get_gaussian_kernel1d / _get_gaussian_kernel2d             transforms/v2/functional/_misc.py:86-99
  adjust_sharpness / adjust_sharpness_image / _video          transforms/v2/functional/_color.py:218-288
New operators on the same primitive (BASELINE cfg1/cfg3; the reference has no box / separable / Sobel):
  depthwise_conv2d, box_filter, separable_gaussian_blur, sobel, gaussian_sobel
First CNN layer (models/vgg.py:81-85, ops/misc.py:97-119):
  conv2d_bias_relu

Tensors must live on a HIP device; there is no CPU fallback (see _lib.py).  Host-side work here is what
the reference also does on the host: argument checking, the handful of Gaussian taps, reshapes.
"""
from __future__ import annotations

import functools
import math
from typing import List, Optional, Sequence, Tuple, Union

import ctypes as C

import torch

from . import _lib, _pil, tv_tensors
from ._registry import _get_kernel, _register_kernel_internal

# direct 2-D evaluation (the reference's own formulation) up to this many taps; larger float kernels
# run as the fused separable pair (same result to ~1e-7 relative, see DESIGN.md "Numerics")
_DIRECT_2D_MAX_TAPS = 49
# uint8 images with a kernel side above 3.  True (default) = ONE 2-D pass, the reference's own formulation
# (_misc.py:147-163: pad -> conv2d with the outer-product kernel -> round_()): equal to the reference's outputs on every pixel of
# the committed fixtures (93 144 uint8 pixels, tests/golden/gaussian_blur.npz) and on 37.7 M random pixels against the
# reference's call sequence on torch-CPU (3x3 ... 23x23: 0 differing pixels).  False = the fp32 separable pair + one round_():
# kx + ky instead of kx * ky taps (5x5 1.8x, 7x7 2.4x, 23x23 28x faster on 32 x 4K uint8) but another association of the sum --
# on the same data it differs from the reference by 1 LSB in 1 of the 93 144 fixture pixels and in 0.0006-0.0036 % of the
# random ones (exact rounding ties; the reference's own test allows atol = 1).  Integer results are the reference's by
# default; the fast form is an opt-in (DESIGN.md section 4 has the table).
INTEGER_BLUR_EXACT_2D = True
# With INTEGER_BLUR_EXACT_2D: True = get those integers through mv_gaussian_blur_u8_ws -- the separable pair for every pixel
# plus the 2-D chain for the 1-4 % of lane-rows whose value lies within the two forms' (proved) error bound of a rounding tie:
# the same bits as the 2-D pass, 1.5-20x faster (csrc/tiefix_u8.hip).  False = run the plain 2-D pass.
INTEGER_BLUR_EXACT_FAST = True
# The 3x3 convolutions and the Linear layers of the CNNs run K in slices when a launch is too small to fill the chip
# (conv3x3_k_slices / linear_k_slices).  The slice plan depends on the workgroup count and therefore on the BATCH SIZE: the
# same image can differ in its last bits between batch 1 and batch 8 (<= 1e-6 relative; each plan is stated by the library
# and restated by the oracle).  True = every such call keeps the single ascending chain per output, whatever the batch:
# batch-invariant bits, at the small-batch speed of the unsliced kernels.  A fused InvertedResidual block whose plan has several
# slices (mv_inverted_residual_k_slices) then runs as three launches.  (MobileNet's pointwise convs slice K inside the
# workgroup, mv_conv1x1_k_slices: that plan is part of the kernel and is not switched by this flag.)
BATCH_INVARIANT_SUMMATION = False


def _max_value(dtype: torch.dtype) -> int:
    """transforms/_functional_tensor.py:41-57"""
    return {torch.uint8: 255, torch.int8: 127, torch.int16: 32767, torch.uint16: 65535, torch.int32: 2147483647,
            torch.int64: 9223372036854775807}.get(dtype, 1)


# --------------------------------------------------------------------------------------------- taps
def _get_gaussian_kernel1d(kernel_size: int, sigma: float, dtype: torch.dtype = torch.float32,
                           device: Union[str, torch.device] = "cpu") -> torch.Tensor:
    """_misc.py:86-90 -- the same torch ops, so on the host the taps are bit-identical to the reference's."""
    lim = (kernel_size - 1) / (2.0 * math.sqrt(2.0))
    x = torch.linspace(-lim, lim, steps=kernel_size, dtype=dtype, device=device)
    return torch.softmax(x.div(sigma).pow(2).neg(), dim=0)


def _get_gaussian_kernel2d(kernel_size: List[int], sigma: List[float], dtype: torch.dtype = torch.float32,
                           device: Union[str, torch.device] = "cpu") -> torch.Tensor:
    """_misc.py:93-99: kernel_size / sigma are (x, y); the result has shape (ky, kx)."""
    kernel1d_x = _get_gaussian_kernel1d(kernel_size[0], sigma[0], dtype, device)
    kernel1d_y = _get_gaussian_kernel1d(kernel_size[1], sigma[1], dtype, device)
    return kernel1d_y.unsqueeze(-1) * kernel1d_x


@functools.lru_cache(maxsize=256)
def _host_taps(kernel_size: int, sigma: float, v1: bool = False, dtype: torch.dtype = torch.float32):
    """(tensor, ctypes array) of one 1-D Gaussian, cached: a DataLoader calls the same (k, sigma) over and
    over, and building the taps with five torch ops costs more host time than launching the kernel.
    dtype float64: the taps of a float64 image, which the reference builds in the image's dtype (_misc.py:141)."""
    if v1:
        from . import functional_v1
        t = functional_v1._get_gaussian_kernel1d(kernel_size, sigma, dtype)
    else:
        t = _get_gaussian_kernel1d(kernel_size, sigma, dtype)
    return t, (_lib.taps64_from_tensor(t) if dtype == torch.float64 else _lib.taps_from_tensor(t))


def _taps_dtype(image: torch.Tensor) -> torch.dtype:
    return torch.float64 if image.dtype == torch.float64 else torch.float32


# --------------------------------------------------------------------------------------------- plumbing
def _planes(image: torch.Tensor) -> Tuple[int, int, int]:
    h, w = image.shape[-2:]
    planes = 1
    for d in image.shape[:-2]:
        planes *= int(d)
    return planes, int(h), int(w)


def _compute_dtype(image: torch.Tensor) -> str:
    if image.dtype == torch.float32:
        return "f32"
    if image.dtype == torch.uint8:
        return "u8"
    return "float" if image.is_floating_point() else "int"


def _filter_f32_u8(image: torch.Tensor, call_f32, call_u8, out_shape=None) -> torch.Tensor:
    """Run a filter whose kernels exist for fp32 and uint8 storage.

    fp16 / bf16 are computed in fp32 and narrowed back (the reference computes in the input dtype; fp32 accumulation
    is at least as accurate).  float64 reaches this helper only for the operators that have no float64 kernel
    (box / Sobel / the generic primitive with a float32 weight); gaussian_blur and adjust_sharpness compute float64
    images in float64 (mv_gaussian_blur_f64 / mv_sharpness_f64).
    Other integer dtypes follow the reference's integer recipe: .to(float32) -> filter -> round_() -> .to(dtype).
    """
    _lib.require_device(image)
    kind = _compute_dtype(image)
    with _lib.on_device_of(image):
        if kind == "u8" and call_u8 is not None:
            x = image.contiguous()
            y = torch.empty(out_shape or x.shape, dtype=torch.uint8, device=x.device)
            call_u8(x, y)
            return y
        x = image.contiguous() if kind == "f32" else image.to(torch.float32).contiguous()
        y = torch.empty(out_shape or x.shape, dtype=torch.float32, device=x.device)
        call_f32(x, y)
        if kind == "f32":
            return y
        if kind == "float":
            return y.to(image.dtype)
        return y.round_().to(image.dtype)


def _border(border: str) -> int:
    try:
        return _lib.BORDERS[border]
    except KeyError:
        raise ValueError(f"border should be one of {sorted(_lib.BORDERS)}. Got {border!r}") from None


# --------------------------------------------------------------------------------------------- gaussian_blur
def gaussian_blur(inpt: torch.Tensor, kernel_size: List[int], sigma: Optional[List[float]] = None) -> torch.Tensor:
    """Dispatcher, as transforms/v2/functional/_misc.py:75-83."""
    kernel = _get_kernel(gaussian_blur, type(inpt))
    return kernel(inpt, kernel_size=kernel_size, sigma=sigma)


def _check_gaussian_args(kernel_size, sigma):
    """Argument normalisation of gaussian_blur_image, _misc.py:108-133 (same messages)."""
    if isinstance(kernel_size, int):
        kernel_size = [kernel_size, kernel_size]
    elif len(kernel_size) != 2:
        raise ValueError(f"If kernel_size is a sequence its length should be 2. Got {len(kernel_size)}")
    for ksize in kernel_size:
        if ksize % 2 == 0 or ksize < 0:
            raise ValueError(f"kernel_size should have odd and positive integers. Got {kernel_size}")

    if sigma is None:
        sigma = [ksize * 0.15 + 0.35 for ksize in kernel_size]
    else:
        if isinstance(sigma, (list, tuple)):
            length = len(sigma)
            if length == 1:
                s = sigma[0]
                sigma = [s, s]
            elif length != 2:
                raise ValueError(f"If sigma is a sequence, its length should be 2. Got {length}")
        elif isinstance(sigma, (int, float)):
            s = float(sigma)
            sigma = [s, s]
        else:
            raise TypeError(f"sigma should be either float or sequence of floats. Got {type(sigma)}")
    for s in sigma:
        if s <= 0.0:
            raise ValueError(f"sigma should have positive values. Got {sigma}")
    return list(kernel_size), list(sigma)


def _blur_with_taps(image: torch.Tensor, taps_x, taps_y, separable: bool) -> torch.Tensor:
    """pad(reflect) + depthwise conv with the outer-product kernel (or its separable factorisation).
    taps_* are (tensor, ctypes array) pairs from _host_taps."""
    (k1d_x, tx), (k1d_y, ty) = taps_x, taps_y
    kx, ky = k1d_x.numel(), k1d_y.numel()
    h, w = image.shape[-2:]
    if kx // 2 >= w or ky // 2 >= h:
        # ATen's reflection_pad2d message, raised here before any launch
        raise RuntimeError(
            f"Argument #4: Padding size should be less than the corresponding input dimension, but got: padding "
            f"({kx // 2}, {kx // 2}) at dimension 3 of input {list(image.shape)}"
            if kx // 2 >= w else
            f"Argument #6: Padding size should be less than the corresponding input dimension, but got: padding "
            f"({ky // 2}, {ky // 2}) at dimension 2 of input {list(image.shape)}")
    lib = _lib.load()
    planes, h, w = _planes(image)
    if image.dtype == torch.float64:
        # float64 images are computed in float64 like the reference's (taps, outer product and accumulation), one 2-D pass
        assert k1d_x.dtype == torch.float64 and k1d_y.dtype == torch.float64
        _lib.require_device(image)
        with _lib.on_device_of(image):
            x = image.contiguous()
            y = torch.empty_like(x)
            if max(kx, ky) <= _lib.MAX_TAPS_1D:
                _lib.check(lib.mv_gaussian_blur_f64(x.data_ptr(), y.data_ptr(), planes, h, w, tx, kx, ty, ky, _lib.stream_ptr(x)))
            else:  # kernel sides above 63 (ElasticTransform with sigma >= 8): the 2-D kernel as a device array
                k2 = (k1d_y.unsqueeze(-1) * k1d_x).to(x.device).contiguous()
                _lib.check(lib.mv_depthwise_conv2d_f64(x.data_ptr(), y.data_ptr(), k2.data_ptr(), planes, h, w, ky, kx,
                                                       _lib.BORDER_REFLECT, _lib.stream_ptr(x)))
        return y
    huge = max(kx, ky) > _lib.MAX_TAPS_1D

    def f32(x, y):
        if huge:  # beyond the 63-tap separable kernels: the reference's own single 2-D pass with the taps on the device
            k2 = (k1d_y.unsqueeze(-1) * k1d_x).to(x.device).contiguous()
            _lib.check(lib.mv_depthwise_conv2d_f32(x.data_ptr(), y.data_ptr(), k2.data_ptr(), 1, planes, h, w, ky, kx,
                                                   _lib.BORDER_REFLECT, _lib.stream_ptr(x)))
            return
        fn = lib.mv_separable_blur_f32 if separable else lib.mv_gaussian_blur_f32
        _lib.check(fn(x.data_ptr(), y.data_ptr(), planes, h, w, tx, kx, ty, ky, _lib.stream_ptr(x)))

    def u8(x, y):
        if huge:
            k2 = (k1d_y.unsqueeze(-1) * k1d_x).to(x.device).contiguous()
            _lib.check(lib.mv_depthwise_conv2d_u8(x.data_ptr(), y.data_ptr(), k2.data_ptr(), 1, planes, h, w, ky, kx,
                                                  _lib.BORDER_REFLECT, _lib.stream_ptr(x)))
            return
        # large kernels on uint8 storage (SimCLR-style GaussianBlur(23)): the separable pair in fp32, then round_()
        big = separable and max(kx, ky) <= 63 and w >= 8
        if not big and INTEGER_BLUR_EXACT_2D and INTEGER_BLUR_EXACT_FAST:
            # the reference's integers at the separable pair's cost (mv_gaussian_blur_u8_ws): the pair everywhere, the 2-D chain
            # only for the lane-rows within its error bound of a rounding tie
            nbytes = int(lib.mv_gaussian_blur_u8_workspace_bytes(planes, h, w, kx, ky))
            if nbytes:
                ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
                _lib.check(lib.mv_gaussian_blur_u8_ws(x.data_ptr(), y.data_ptr(), planes, h, w, tx, kx, ty, ky, ws.data_ptr(), nbytes,
                                                      _lib.stream_ptr(x)))
                return
        fn = lib.mv_separable_blur_u8 if big else lib.mv_gaussian_blur_u8
        _lib.check(fn(x.data_ptr(), y.data_ptr(), planes, h, w, tx, kx, ty, ky, _lib.stream_ptr(x)))

    if image.dtype in (torch.float16, torch.bfloat16) and not separable and not huge:
        # half-precision storage, 2-D pass: fp32 arithmetic inside the tile kernel, one rounding on store -- the same bits as
        # .to(float32) -> filter -> .to(dtype), without the two conversion passes over the image
        _lib.require_device(image)
        with _lib.on_device_of(image):
            x = image.contiguous()
            y = torch.empty_like(x)
            fn = lib.mv_gaussian_blur_f16 if image.dtype == torch.float16 else lib.mv_gaussian_blur_bf16
            _lib.check(fn(x.data_ptr(), y.data_ptr(), planes, h, w, tx, kx, ty, ky, _lib.stream_ptr(x)))
        return y
    return _filter_f32_u8(image, f32, u8)


def _use_separable(kx: int, ky: int, image: torch.Tensor) -> bool:
    """Which formulation gaussian_blur_image runs.  The reference's is one 2-D pass with the outer-product kernel; the
    separable pair (row pass, then column pass, both fp32) agrees with it to <= 2e-7 relative and costs kx + ky instead of
    kx * ky taps.
      float images   2-D while both sides are <= 5 (the templated tile kernels: 3x3, 5x5, 3x5, 5x3, HBM-bound anyway);
                     separable beyond -- 7x7 runs 1.56 -> 1.15 ms and unequal sizes such as (7, 3) or (5, 9), which only had
                     the run-time-size tile kernel, 2.4-3.8 -> 1.1-1.3 ms on 32 x 4K frames.  Widths with W % 4 != 0 keep
                     the 2-D pass while both sides are <= 7 (templated tile kernel, 3.5 TB/s): k_sepfast needs 16-byte
                     rows and the LDS fallback behind it runs at 1.6 TB/s (k_sepstream, sides above 7, takes any width)
      uint8          the reference's single 2-D pass by default (INTEGER_BLUR_EXACT_2D = True: bit-for-bit the reference's
                     integers).  With INTEGER_BLUR_EXACT_2D = False: 2-D for 3x3 (HBM-bound already) and for images narrower
                     than 16 pixels, the fp32 separable pair + one round_() for every larger kernel -- 5x5 / 7x7 are
                     VALU-bound as one 25- / 49-tap chain (0.54 / 0.96 ms on 32 x 4K uint8) and need 10 / 14 taps as a pair;
                     the pair differs from the 2-D sum only at exact rounding ties (~1e-5 of the pixels by 1 LSB, inside the
                     reference's own atol = 1, test_transforms_v2.py:3309)
      other integers always the 2-D pass."""
    if image.is_floating_point():
        if kx <= 5 and ky <= 5:
            return False
        if image.dtype in (torch.float16, torch.bfloat16) and kx <= 11 and ky <= 11:
            return False  # half-precision storage runs fused in the 2-D tile kernel (no fp32 copies of the image)
        odd_width = image.ndim >= 1 and image.shape[-1] % 4 != 0
        return not (odd_width and kx <= 7 and ky <= 7)
    if image.dtype != torch.uint8 or INTEGER_BLUR_EXACT_2D:
        return False
    wide = image.ndim >= 1 and image.shape[-1] >= 16  # the 16-pixel-per-lane kernels
    return kx > 7 or ky > 7 or ((kx > 3 or ky > 3) and wide)


@_register_kernel_internal(gaussian_blur, torch.Tensor)
@_register_kernel_internal(gaussian_blur, tv_tensors.Image)
def gaussian_blur_image(image: torch.Tensor, kernel_size: List[int], sigma: Optional[List[float]] = None) -> torch.Tensor:
    """gaussian_blur_image (_misc.py:102-165): (..., C, H, W) of any dtype -> same shape and dtype.

    Differences from the reference, all below its own test tolerance: the reflect border is resolved inside
    the kernel (no padded copy), fp16 / bf16 images accumulate in fp32 (float64 images in float64, float32 in float32,
    like the reference), and larger float32 kernels run as the fused separable pair (`_use_separable`).
    """
    kernel_size, sigma = _check_gaussian_args(kernel_size, sigma)
    if image.numel() == 0:
        return image
    image.shape[-3]  # noqa: B018 -- (..., C, H, W) required, IndexError like the reference otherwise
    k1d_x = _host_taps(kernel_size[0], float(sigma[0]), False, _taps_dtype(image))
    k1d_y = _host_taps(kernel_size[1], float(sigma[1]), False, _taps_dtype(image))
    separable = _use_separable(kernel_size[0], kernel_size[1], image)
    return _blur_with_taps(image, k1d_x, k1d_y, separable)


def _gaussian_blur_image_pil(image, kernel_size: List[int], sigma: Optional[List[float]] = None):
    """_gaussian_blur_image_pil (_misc.py:169-174): pil_to_tensor -> gaussian_blur_image -> to_pil_image(mode=image.mode).
    The tensor goes to the current HIP device for the kernel and comes back (PCIe both ways: a DataLoader that decodes to
    PIL should convert once and keep the batch on the device -- see DESIGN.md "Host buffers")."""
    t_img = _pil.pil_to_tensor(image)
    output = gaussian_blur_image(t_img.to(_pil.device_for_host_inputs()), kernel_size=kernel_size, sigma=sigma)
    return _pil.to_pil_image(output.cpu(), mode=image.mode)


if _pil.PIL is not None:
    _register_kernel_internal(gaussian_blur, _pil.PIL.Image.Image)(_gaussian_blur_image_pil)


@_register_kernel_internal(gaussian_blur, tv_tensors.Video)
def gaussian_blur_video(video: torch.Tensor, kernel_size: List[int], sigma: Optional[List[float]] = None) -> torch.Tensor:
    return gaussian_blur_image(video, kernel_size, sigma)


# --------------------------------------------------------------------------------------------- adjust_sharpness
def adjust_sharpness(inpt: torch.Tensor, sharpness_factor: float) -> torch.Tensor:
    """Dispatcher, as transforms/v2/functional/_color.py:218-226."""
    kernel = _get_kernel(adjust_sharpness, type(inpt))
    return kernel(inpt, sharpness_factor=sharpness_factor)


def _sharpness(image: torch.Tensor, sharpness_factor: float, v1: bool) -> torch.Tensor:
    lib = _lib.load()
    _lib.require_device(image)
    planes, h, w = _planes(image)
    f = float(sharpness_factor)
    with _lib.on_device_of(image):
        if image.dtype == torch.uint8:
            x = image.contiguous()
            y = torch.empty_like(x)
            _lib.check(lib.mv_sharpness_u8(x.data_ptr(), y.data_ptr(), planes, h, w, f, int(v1), _lib.stream_ptr(x)))
            return y
        if image.dtype == torch.float64:  # computed in float64 like the reference (kernel dtype = image dtype, _color.py:246)
            x = image.contiguous()
            y = torch.empty_like(x)
            _lib.check(lib.mv_sharpness_f64(x.data_ptr(), y.data_ptr(), planes, h, w, f, int(v1), _lib.stream_ptr(x)))
            return y
        fp = image.is_floating_point()
        x = image.to(torch.float32).contiguous()
        y = torch.empty_like(x)
        _lib.check(lib.mv_sharpness_f32(x.data_ptr(), y.data_ptr(), planes, h, w, f, int(v1),
                                        float(_max_value(image.dtype)), int(not fp), _lib.stream_ptr(x)))
        return y if image.dtype == torch.float32 else y.to(image.dtype)


@_register_kernel_internal(adjust_sharpness, torch.Tensor)
@_register_kernel_internal(adjust_sharpness, tv_tensors.Image)
def adjust_sharpness_image(image: torch.Tensor, sharpness_factor: float) -> torch.Tensor:
    """adjust_sharpness_image (_color.py:229-280): valid 3x3 smoothing, in-place blend, clamp -- one kernel."""
    num_channels, height, width = image.shape[-3:]
    if num_channels not in (1, 3):
        raise TypeError(f"Input image tensor can have 1 or 3 channels, but found {num_channels}")
    if sharpness_factor < 0:
        raise ValueError(f"sharpness_factor ({sharpness_factor}) is not non-negative.")
    if image.numel() == 0 or height <= 2 or width <= 2:
        return image
    return _sharpness(image, sharpness_factor, v1=False)


def _adjust_sharpness_image_pil(image, sharpness_factor: float):
    """The reference registers PIL's own ImageEnhance.Sharpness here (_color.py:283 -> _functional_pil.py:113-121): SMOOTH
    filter, blend with the original, alpha band kept.  The uint8 tensor kernel reproduces PIL's result exactly
    (tests/golden: PIL-exact vectors), so the PIL entry is that kernel around a conversion; an alpha band passes through."""
    if not _pil.is_pil_image(image):
        raise TypeError(f"img should be PIL Image. Got {type(image)}")
    t = _pil.pil_to_tensor(image)
    if image.mode in ("LA", "RGBA"):
        color, alpha = t[:-1], t[-1:]
    else:
        color, alpha = t, None
    if color.shape[0] not in (1, 3) or color.dtype != torch.uint8:
        raise TypeError(f"adjust_sharpness on a PIL image supports modes L, LA, RGB and RGBA, got {image.mode}")
    out = adjust_sharpness_image(color.contiguous().to(_pil.device_for_host_inputs()), sharpness_factor).cpu()
    if alpha is not None:
        out = torch.cat([out, alpha], 0)
    return _pil.to_pil_image(out, mode=image.mode)


if _pil.PIL is not None:
    _register_kernel_internal(adjust_sharpness, _pil.PIL.Image.Image)(_adjust_sharpness_image_pil)


@_register_kernel_internal(adjust_sharpness, tv_tensors.Video)
def adjust_sharpness_video(video: torch.Tensor, sharpness_factor: float) -> torch.Tensor:
    return adjust_sharpness_image(video, sharpness_factor=sharpness_factor)


# --------------------------------------------------------------------------------------------- lists of frames, one launch
def _same_frames(frames: Sequence[torch.Tensor]) -> bool:
    f0 = frames[0]
    return all(isinstance(f, torch.Tensor) and f.is_cuda and f.device == f0.device and f.dtype == f0.dtype
               and f.shape == f0.shape for f in frames)


def _frames_call(frames: Sequence[torch.Tensor], fn_name: str, args_of):
    """Run one mv_*_v launch over separately allocated, equally shaped frames; returns views of one output batch."""
    lib = _lib.load()
    f0 = frames[0]
    planes, h, w = _planes(f0)
    with _lib.on_device_of(f0):
        xs = [f.contiguous() for f in frames]
        out = torch.empty((len(xs),) + tuple(f0.shape), dtype=f0.dtype, device=f0.device)
        ys = [out[i] for i in range(len(xs))]
        _lib.check(getattr(lib, fn_name)(_lib.pointer_table(xs), _lib.pointer_table(ys), len(xs), planes, h, w,
                                         *args_of(), _lib.stream_ptr(f0)))
    return ys


def gaussian_blur_frames(frames: Sequence[torch.Tensor], kernel_size: List[int], sigma: Optional[List[float]] = None
                         ) -> List[torch.Tensor]:
    """gaussian_blur_image over a LIST of separately allocated frames (what a DataLoader hands to a transform,
    transforms/v2/_transform.py:40-55).  Equally shaped float32 / uint8 frames on one device go through ONE launch
    (mv_gaussian_blur_*_v / mv_separable_blur_*_v, whichever formulation gaussian_blur_image runs for that size: per-frame
    base pointers in the kernel arguments, no copy of the frames); anything else is a loop over gaussian_blur_image.
    Results equal the per-frame calls bit for bit."""
    frames = list(frames)
    if not frames:
        return []
    kernel_size, sigma = _check_gaussian_args(kernel_size, sigma)
    f0 = frames[0]
    kx, ky = kernel_size
    batched = (len(frames) > 1 and _same_frames(frames) and f0.dtype in (torch.float32, torch.uint8) and f0.ndim >= 3
               and f0.numel() > 0 and kx // 2 < f0.shape[-1] and ky // 2 < f0.shape[-2] and max(kx, ky) <= _lib.MAX_TAPS_1D)
    separable = batched and _use_separable(kx, ky, f0)
    if separable and f0.dtype == torch.uint8 and f0.shape[-1] < (8 if max(kx, ky) > 7 else 16):
        batched = False  # (gaussian_blur_image takes the 2-D pass there)
    if batched and not separable and max(kx, ky) > 11:
        batched = False
    if not batched:
        return [gaussian_blur_image(f, kernel_size, sigma) for f in frames]
    (_, tx), (_, ty) = _host_taps(kx, float(sigma[0])), _host_taps(ky, float(sigma[1]))
    name = ("mv_separable_blur_" if separable else "mv_gaussian_blur_") + ("f32_v" if f0.dtype == torch.float32 else "u8_v")
    return _frames_call(frames, name, lambda: (tx, kx, ty, ky))


def adjust_sharpness_frames(frames: Sequence[torch.Tensor], sharpness_factor: float) -> List[torch.Tensor]:
    """adjust_sharpness_image over a list of separately allocated frames: one launch for equally shaped float32 / uint8
    frames (mv_sharpness_*_v), a loop otherwise."""
    frames = list(frames)
    if not frames:
        return []
    f0 = frames[0]
    batched = (len(frames) > 1 and _same_frames(frames) and f0.dtype in (torch.float32, torch.uint8) and f0.ndim >= 3
               and f0.shape[-3] in (1, 3) and f0.numel() > 0 and f0.shape[-1] > 2 and f0.shape[-2] > 2 and sharpness_factor >= 0)
    if not batched:
        return [adjust_sharpness_image(f, sharpness_factor) for f in frames]
    f = float(sharpness_factor)
    if f0.dtype == torch.uint8:
        return _frames_call(frames, "mv_sharpness_u8_v", lambda: (f, 0))
    return _frames_call(frames, "mv_sharpness_f32_v", lambda: (f, 0, 1.0, 0))


# --------------------------------------------------------------------------------------------- the primitive, exposed
def depthwise_conv2d(image: torch.Tensor, weight: torch.Tensor, border: str = "reflect") -> torch.Tensor:
    """[pad(border)] + conv2d(image, weight.expand(C,1,ky,kx), groups=C): the primitive every filter of the
    reference is built on (_misc.py:153-155).  `weight` is a (ky, kx) tensor (host or device)."""
    if weight.ndim != 2:
        raise ValueError(f"weight should be a 2-D (ky, kx) tensor. Got shape {tuple(weight.shape)}")
    ky, kx = int(weight.shape[0]), int(weight.shape[1])
    if ky % 2 == 0 or kx % 2 == 0:
        raise ValueError(f"kernel size must be odd and positive, got ({ky}, {kx})")
    b = _border(border)
    if image.numel() == 0:
        return image
    lib = _lib.load()
    planes, h, w = _planes(image)
    if b == _lib.BORDER_VALID:
        if ky > h or kx > w:
            raise ValueError(f"valid conv: kernel ({ky}, {kx}) larger than image ({h}, {w})")
        out_shape = tuple(image.shape[:-2]) + (h - ky + 1, w - kx + 1)
    else:
        out_shape = None
    if image.dtype == torch.float64:
        # a float64 image is filtered in float64, like conv2d of a float64 tensor (the weight is widened exactly if it is float32)
        _lib.require_device(image)
        with _lib.on_device_of(image):
            x = image.contiguous()
            y = torch.empty(out_shape or x.shape, dtype=torch.float64, device=x.device)
            wd = weight.detach().to(x.device, torch.float64).contiguous()
            _lib.check(lib.mv_depthwise_conv2d_f64(x.data_ptr(), y.data_ptr(), wd.data_ptr(), planes, h, w, ky, kx, b, _lib.stream_ptr(x)))
        return y
    on_device = ky * kx > _lib.MAX_HOST_TAPS_2D
    if on_device:
        wt = weight.detach().to(image.device, torch.float32).contiguous()
        wp = wt.data_ptr()
    else:
        wt = _lib.taps_from_tensor(weight)
        wp = wt

    def f32(x, y):
        _lib.check(lib.mv_depthwise_conv2d_f32(x.data_ptr(), y.data_ptr(), wp, int(on_device), planes, h, w, ky, kx, b,
                                               _lib.stream_ptr(x)))

    def u8(x, y):
        _lib.check(lib.mv_depthwise_conv2d_u8(x.data_ptr(), y.data_ptr(), wp, int(on_device), planes, h, w, ky, kx, b,
                                              _lib.stream_ptr(x)))

    return _filter_f32_u8(image, f32, u8, out_shape)


def box_filter(image: torch.Tensor, kernel_size: Union[int, Sequence[int]] = 3, border: str = "reflect") -> torch.Tensor:
    """k x k mean filter (BASELINE cfg1): the primitive with w = 1/(kx*ky)."""
    if isinstance(kernel_size, int):
        kernel_size = [kernel_size, kernel_size]
    kx, ky = kernel_size
    w = torch.full((ky, kx), 1.0 / float(kx * ky), dtype=torch.float32)
    return depthwise_conv2d(image, w, border)


def separable_gaussian_blur(image: torch.Tensor, kernel_size: List[int], sigma: Optional[List[float]] = None) -> torch.Tensor:
    """Gaussian blur as the fused (1 x kx) then (ky x 1) pair of the primitive (BASELINE cfg3); float images."""
    kernel_size, sigma = _check_gaussian_args(kernel_size, sigma)
    if image.numel() == 0:
        return image
    k1d_x = _host_taps(kernel_size[0], float(sigma[0]))
    k1d_y = _host_taps(kernel_size[1], float(sigma[1]))
    if not image.is_floating_point():
        raise TypeError(f"separable_gaussian_blur expects a floating point image. Got {image.dtype}")
    return _blur_with_taps(image, k1d_x, k1d_y, separable=True)


def _pair_f32(image: torch.Tensor, call) -> Tuple[torch.Tensor, torch.Tensor]:
    _lib.require_device(image)
    if not image.is_floating_point():
        raise TypeError(f"expected a floating point image. Got {image.dtype}")
    with _lib.on_device_of(image):
        x = image.to(torch.float32).contiguous()
        shape = call.out_shape(x)
        gx = torch.empty(shape, dtype=torch.float32, device=x.device)
        gy = torch.empty(shape, dtype=torch.float32, device=x.device)
        call(x, gx, gy)
    if image.dtype != torch.float32:
        gx, gy = gx.to(image.dtype), gy.to(image.dtype)
    return gx, gy


def sobel(image: torch.Tensor, border: str = "reflect") -> Tuple[torch.Tensor, torch.Tensor]:
    """(gx, gy): the primitive with taps [[-1,0,1],[-2,0,2],[-1,0,1]] and its transpose (cross-correlation)."""
    b = _border(border)
    lib = _lib.load()
    planes, h, w = _planes(image)
    if image.numel() == 0:
        return image, image

    def call(x, gx, gy):
        _lib.check(lib.mv_sobel_f32(x.data_ptr(), gx.data_ptr(), gy.data_ptr(), planes, h, w, b, _lib.stream_ptr(x)))

    call.out_shape = lambda x: (tuple(x.shape[:-2]) + (h - 2, w - 2)) if b == _lib.BORDER_VALID else tuple(x.shape)
    return _pair_f32(image, call)


def gaussian_sobel(image: torch.Tensor, kernel_size: List[int], sigma: Optional[List[float]] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """BASELINE cfg3 as one kernel: separable Gaussian (reflect) then Sobel (reflect) of the blurred image."""
    kernel_size, sigma = _check_gaussian_args(kernel_size, sigma)
    if image.numel() == 0:
        return image, image
    (k1d_x, tx), (k1d_y, ty) = _host_taps(kernel_size[0], float(sigma[0])), _host_taps(kernel_size[1], float(sigma[1]))
    lib = _lib.load()
    planes, h, w = _planes(image)

    def call(x, gx, gy):
        _lib.check(lib.mv_gaussian_sobel_f32(x.data_ptr(), gx.data_ptr(), gy.data_ptr(), planes, h, w, tx,
                                             k1d_x.numel(), ty, k1d_y.numel(), _lib.stream_ptr(x)))

    call.out_shape = lambda x: tuple(x.shape)
    return _pair_f32(image, call)


# --------------------------------------------------------------------------------------------- first CNN layer
def conv2d_bias_relu(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = True,
                     out: Optional[torch.Tensor] = None, sliced_k: Optional[bool] = None) -> torch.Tensor:
    """relu(conv2d(x, weight, bias, padding=1)) for 3x3 kernels -- nn.Conv2d(cin, cout, 3, padding=1) + nn.ReLU
    (models/vgg.py:81-85) on the fp32 MFMA.  x (N, Cin, H, W) fp32, weight (Cout, Cin, 3, 3).
    sliced_k: None = `not BATCH_INVARIANT_SUMMATION`; True lets small launches sum K in slices across workgroups (the order
    conv3x3_k_slices states: it depends on the batch size); False keeps one chain per output for every batch size."""
    if sliced_k is None:
        sliced_k = not BATCH_INVARIANT_SUMMATION
    if x.ndim != 4:
        raise ValueError(f"Expected 4D (N, C, H, W) input. Got {x.ndim}D")
    if weight.ndim != 4 or weight.shape[2:] != (3, 3):
        raise ValueError(f"weight should have shape (Cout, Cin, 3, 3). Got {tuple(weight.shape)}")
    n, cin, h, w = (int(d) for d in x.shape)
    cout = int(weight.shape[0])
    if int(weight.shape[1]) != cin:
        raise RuntimeError(f"Given groups=1, weight of size {list(weight.shape)}, expected input{list(x.shape)} "
                           f"to have {int(weight.shape[1])} channels, but got {cin} channels instead")
    if bias is not None and tuple(bias.shape) != (cout,):
        raise ValueError(f"bias should have shape ({cout},). Got {tuple(bias.shape)}")
    _lib.require_device(x)
    _lib.require_device(weight, "weight")
    if x.dtype != torch.float32 or weight.dtype != torch.float32:
        raise TypeError(f"conv2d_bias_relu computes in float32. Got input {x.dtype}, weight {weight.dtype}")
    lib = _lib.load()
    with _lib.on_device_of(x):
        xc, wc = x.contiguous(), weight.detach().contiguous()
        bc = None if bias is None else bias.detach().to(x.device, torch.float32).contiguous()
        if out is None:
            out = torch.empty((n, cout, h, w), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (n, cout, h, w) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous float32 tensor of shape (N, Cout, H, W)")
        # launches too small to fill the chip with one chain per output run K in slices across workgroups (conv3x3_k_slices
        # states the order); their raw sums pass through a workspace
        nbytes = int(lib.mv_conv3x3_workspace_bytes(n, cin, h, w, cout)) if sliced_k else 0
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            _lib.check(lib.mv_conv3x3_bias_relu_ws_f32(xc.data_ptr(), wc.data_ptr(), None if bc is None else bc.data_ptr(), out.data_ptr(),
                                                       n, cin, h, w, cout, int(relu), ws.data_ptr(), nbytes, _lib.stream_ptr(xc)))
        else:
            _lib.check(lib.mv_conv3x3_bias_relu_f32(xc.data_ptr(), wc.data_ptr(), None if bc is None else bc.data_ptr(),
                                                    out.data_ptr(), n, cin, h, w, cout, int(relu), _lib.stream_ptr(xc)))
    return _lib.forward_only(out, "conv2d_bias_relu", x, weight, bias)


def conv3x3_k_slices(n: int, cin: int, h: int, w: int, cout: int) -> Tuple[int, int]:
    """(slices, slice_channels) of the summation order conv2d_bias_relu uses for a 3x3 conv of this shape
    (mv_conv3x3_k_slices): 1 slice = one ascending (channel, ky, kx) chain per output; more = chains over slices of
    slice_channels input channels, added in ascending order, then bias and ReLU (host logic, no GPU needed)."""
    sc = C.c_int(0)
    s = int(_lib.load().mv_conv3x3_k_slices(int(n), int(cin), int(h), int(w), int(cout), C.byref(sc)))
    return s, int(sc.value)


# --------------------------------------------------------------------------------------------- rest of the small CNN (8f.1)
def max_pool2d_2x2(x: torch.Tensor) -> torch.Tensor:
    """nn.MaxPool2d(kernel_size=2, stride=2) (models/vgg.py:78-79) on (..., H, W) fp32."""
    _lib.require_device(x)
    if x.dtype != torch.float32:
        raise TypeError(f"max_pool2d_2x2 computes in float32. Got {x.dtype}")
    planes, h, w = _planes(x)
    lib = _lib.load()
    with _lib.on_device_of(x):
        xc = x.contiguous()
        y = torch.empty(tuple(x.shape[:-2]) + (h // 2, w // 2), dtype=torch.float32, device=x.device)
        _lib.check(lib.mv_maxpool2x2_f32(xc.data_ptr(), y.data_ptr(), planes, h, w, _lib.stream_ptr(xc)))
    return y


def max_pool2d(x: torch.Tensor, kernel_size: int, stride: Optional[int] = None) -> torch.Tensor:
    """nn.MaxPool2d(kernel_size, stride) without padding, floor mode (AlexNet: 3, 2; models/alexnet.py:24) on (..., H, W)."""
    stride = kernel_size if stride is None else stride
    if (kernel_size, stride) == (2, 2):
        return max_pool2d_2x2(x)
    _lib.require_device(x)
    if x.dtype != torch.float32:
        raise TypeError(f"max_pool2d computes in float32. Got {x.dtype}")
    planes, h, w = _planes(x)
    if kernel_size > h or kernel_size > w:
        raise RuntimeError(f"max_pool2d: kernel {kernel_size} exceeds the {h}x{w} input")
    lib = _lib.load()
    with _lib.on_device_of(x):
        xc = x.contiguous()
        y = torch.empty(tuple(x.shape[:-2]) + ((h - kernel_size) // stride + 1, (w - kernel_size) // stride + 1), dtype=torch.float32,
                        device=x.device)
        _lib.check(lib.mv_maxpool2d_f32(xc.data_ptr(), y.data_ptr(), planes, h, w, kernel_size, stride, _lib.stream_ptr(xc)))
    return y


_CONV_WORKSPACE_BYTES = 1 << 30


# True: conv2d_bias_act brings the columns workspace even when the library does not ask for it (the tuning build's MV_CONV_COLUMNS
# then runs the im2col + GEMM form: what the tests and tools compare the implicit kernel with)
CONV2D_COLUMNS_WORKSPACE = False
# True: launches of a handful of workgroups (batch 1) bring the optional columns workspace (mv_conv2d_needs_workspace() == 2): the
# columns form cuts smaller tiles there (AlexNet batch 1: 0.37 -> 0.28 ms)
CONV2D_SMALL_LAUNCH_WORKSPACE = True


def conv2d_bias_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, stride=1, padding=0, dilation=1,
                    groups: int = 1, activation: Optional[str] = None) -> torch.Tensor:
    """Any nn.Conv2d [+ ReLU ...] of the small CNNs that the specialised kernels do not cover (AlexNet's 11x11 stride 4 and
    5x5: models/alexnet.py:22-33): implicit GEMM on the fp32 MFMA (mv_conv2d_bias_act_f32) -- the im2col columns are gathered
    from the input chunk by chunk into LDS and never reach HBM; no workspace unless mv_conv2d_needs_workspace() asks for one."""
    if x.ndim != 4 or weight.ndim != 4:
        raise RuntimeError(f"Expected 4D input and weight. Got {tuple(x.shape)} and {tuple(weight.shape)}")
    _lib.require_device(x)
    _lib.require_device(weight, "weight")
    if x.dtype != torch.float32 or weight.dtype != torch.float32:
        raise TypeError(f"conv2d_bias_act computes in float32. Got input {x.dtype}, weight {weight.dtype}")
    pair = lambda v: (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))  # noqa: E731
    (sh, sw), (ph, pw), (dh, dw) = pair(stride), pair(padding), pair(dilation)
    n, cin, h, w = (int(d) for d in x.shape)
    cout, cg, kh, kw = (int(d) for d in weight.shape)
    if cg * groups != cin or cout % groups:
        raise RuntimeError(f"weight {tuple(weight.shape)} does not fit {cin} input channels in {groups} groups")
    if activation not in _ACT_CODES:
        raise ValueError(f"unknown activation {activation!r}")
    oh, ow = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (w + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    if oh <= 0 or ow <= 0:
        raise RuntimeError(f"Calculated output size too small - out_h: {oh} out_w: {ow}")
    lib = _lib.load()
    with _lib.on_device_of(x):
        xc, wc = x.contiguous(), weight.detach().contiguous()
        bc = None if bias is None else bias.detach().to(x.device, torch.float32).contiguous()
        y = torch.empty((n, cout, oh, ow), dtype=torch.float32, device=x.device)
        if n == 0:
            return y
        ws = None
        need = int(lib.mv_conv2d_needs_workspace(n, cin, cout, h, w, kh, kw, sh, sw, ph, pw, dh, dw, groups))
        if need == 1 or (need == 2 and CONV2D_SMALL_LAUNCH_WORKSPACE) or CONV2D_COLUMNS_WORKSPACE:
            per_image = int(lib.mv_deform_conv2d_workspace_bytes(1, cin, h, w, kh, kw, sh, sw, ph, pw, dh, dw))
            images = max(1, min(n, _CONV_WORKSPACE_BYTES // max(per_image, 1)))
            ws = torch.empty(images * per_image, dtype=torch.uint8, device=x.device)
        _lib.check(lib.mv_conv2d_bias_act_f32(xc.data_ptr(), wc.data_ptr(), None if bc is None else bc.data_ptr(), y.data_ptr(), n, cin, h, w,
                                              cout, kh, kw, sh, sw, ph, pw, dh, dw, groups, _ACT_CODES[activation],
                                              None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), _lib.stream_ptr(xc)))
    return _lib.forward_only(y, "conv2d_bias_act", x, weight, bias)


def adaptive_avg_pool2d(x: torch.Tensor, output_size: Sequence[int]) -> torch.Tensor:
    """nn.AdaptiveAvgPool2d(output_size) (models/vgg.py:41) on (..., H, W) fp32."""
    _lib.require_device(x)
    if x.dtype != torch.float32:
        raise TypeError(f"adaptive_avg_pool2d computes in float32. Got {x.dtype}")
    oh, ow = (int(output_size[0]), int(output_size[1]))
    planes, h, w = _planes(x)
    lib = _lib.load()
    with _lib.on_device_of(x):
        xc = x.contiguous()
        y = torch.empty(tuple(x.shape[:-2]) + (oh, ow), dtype=torch.float32, device=x.device)
        _lib.check(lib.mv_adaptive_avgpool_f32(xc.data_ptr(), y.data_ptr(), planes, h, w, oh, ow, _lib.stream_ptr(xc)))
    return y


def linear_k_slices(n: int, k: int, m: int) -> Tuple[int, int]:
    """(slices, slice_len) of the K slicing linear_bias_relu uses for an (n, k) x (m, k) problem: 1 slice = the single
    ascending-k chain; more = partial chains over contiguous slices of slice_len, added in ascending order (host logic, no
    GPU needed)."""
    sl = C.c_int(0)
    s = int(_lib.load().mv_linear_k_slices(int(n), int(k), int(m), C.byref(sl)))
    return s, int(sl.value)


def conv1x1_k_slices(n: int, cin: int, h: int, w: int, cout: int) -> Tuple[int, int]:
    """(slices, slice_len) of the summation order conv_norm_act uses for a pointwise conv of this shape
    (mv_conv1x1_k_slices): 1 slice = one ascending-channel chain per output; more = chains over contiguous channel slices of
    slice_len, added in ascending order (host logic, no GPU needed)."""
    sl = C.c_int(0)
    s = int(_lib.load().mv_conv1x1_k_slices(int(n), int(cin), int(h), int(w), int(cout), C.byref(sl)))
    return s, int(sl.value)


def linear_bias_relu(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, relu: bool = False,
                     sliced_k: Optional[bool] = None) -> torch.Tensor:
    """relu?(x @ weight.T + bias): nn.Linear [+ nn.ReLU] of the classifier (models/vgg.py:42-50); x (N, K) fp32,
    weight (M, K) as nn.Linear stores it.  Inference-size batches run K in slices over the whole chip (see
    linear_k_slices; the plan depends on the batch size); sliced_k=False forces the single ascending-k chain, None =
    `not BATCH_INVARIANT_SUMMATION`."""
    if sliced_k is None:
        sliced_k = not BATCH_INVARIANT_SUMMATION
    if x.ndim != 2 or weight.ndim != 2 or weight.shape[1] != x.shape[1]:
        raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(x.shape)} and {tuple(weight.t().shape)})")
    _lib.require_device(x)
    _lib.require_device(weight, "weight")
    if x.dtype != torch.float32 or weight.dtype != torch.float32:
        raise TypeError(f"linear_bias_relu computes in float32. Got input {x.dtype}, weight {weight.dtype}")
    n, k = (int(d) for d in x.shape)
    m = int(weight.shape[0])
    lib = _lib.load()
    with _lib.on_device_of(x):
        xc, wc = x.contiguous(), weight.detach().contiguous()
        bc = None if bias is None else bias.detach().to(x.device, torch.float32).contiguous()
        y = torch.empty((n, m), dtype=torch.float32, device=x.device)
        bp = None if bc is None else bc.data_ptr()
        ws_bytes = int(lib.mv_linear_workspace_bytes(n, k, m)) if sliced_k else 0
        if ws_bytes:
            ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device)
            _lib.check(lib.mv_linear_bias_relu_ws_f32(xc.data_ptr(), wc.data_ptr(), bp, y.data_ptr(), n, k, m, int(relu),
                                                      ws.data_ptr(), ws_bytes, _lib.stream_ptr(xc)))
        else:
            _lib.check(lib.mv_linear_bias_relu_f32(xc.data_ptr(), wc.data_ptr(), bp, y.data_ptr(), n, k, m, int(relu),
                                                   _lib.stream_ptr(xc)))
    return _lib.forward_only(y, "linear_bias_relu", x, weight, bias)


# --------------------------------------------------------------------------------------------- Conv2dNormActivation (8f.3)
_ACT_CODES = {None: 0, "none": 0, "relu": 1, "relu6": 2, "hardswish": 3, "silu": 4}
_AFFINE_CODES = {None: 0, "none": 0, "mul_add": 1, "fma": 2}


def fold_batchnorm(weight, bias, running_mean, running_var, eps: float = 1e-5):
    """nn.BatchNorm2d in eval mode as per-channel (alpha, beta) with y = fma(x, alpha, beta): ATen batch_norm_cpu's
    own steps (alpha = (1 / sqrt(var + eps)) * weight, beta = fma(-mean, alpha, bias)), evaluated on the host by
    mv_fold_batchnorm.  Returns CPU float32 tensors."""
    lib = _lib.load()
    mean = running_mean.detach().to("cpu", torch.float32).contiguous()
    var = running_var.detach().to("cpu", torch.float32).contiguous()
    c = mean.numel()
    w = None if weight is None else weight.detach().to("cpu", torch.float32).contiguous()
    b = None if bias is None else bias.detach().to("cpu", torch.float32).contiguous()
    alpha, beta = torch.empty(c, dtype=torch.float32), torch.empty(c, dtype=torch.float32)
    fp = C.POINTER(C.c_float)
    ptr = lambda t: None if t is None else C.cast(t.data_ptr(), fp)  # noqa: E731
    lib.mv_fold_batchnorm(ptr(w), ptr(b), ptr(mean), ptr(var), float(eps), c, ptr(alpha), ptr(beta))
    return alpha, beta


def conv_norm_act(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                  alpha: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
                  residual: Optional[torch.Tensor] = None, stride: int = 1, groups: int = 1, affine: Optional[str] = None,
                  activation: Optional[str] = None) -> torch.Tensor:
    """One Conv2dNormActivation block (ops/misc.py:68-128) as one kernel: conv2d(padding=(k-1)//2) -> + bias ->
    folded norm (`affine`: "mul_add" = FrozenBatchNorm2d, "fma" = BatchNorm2d eval) -> + residual -> activation.
    Covers the MobileNet family's three shapes: dense 3x3 with cin <= 4 (stem), depthwise 3x3 (groups = C),
    pointwise 1x1; stride 1 or 2."""
    if x.ndim != 4 or weight.ndim != 4:
        raise RuntimeError(f"Expected 4D input and weight. Got {tuple(x.shape)} and {tuple(weight.shape)}")
    _lib.require_device(x)
    _lib.require_device(weight, "weight")
    if x.dtype != torch.float32 or weight.dtype != torch.float32:
        raise TypeError(f"conv_norm_act computes in float32. Got input {x.dtype}, weight {weight.dtype}")
    n, cin, h, w = (int(d) for d in x.shape)
    cout, cg, kh, kw = (int(d) for d in weight.shape)
    if groups == 1 and (kh, kw) == (3, 3) and cg == cin and cin <= 4:
        kind = 0
    elif groups == cin and cg == 1 and cout == cin and (kh, kw) == (3, 3):
        kind = 1
    elif groups == 1 and (kh, kw) == (1, 1) and cg == cin:
        kind = 2
    else:
        raise NotImplementedError(
            f"conv_norm_act covers dense 3x3 with cin <= 4, depthwise 3x3 and pointwise 1x1; got weight {tuple(weight.shape)}, "
            f"groups={groups} on {cin} input channels")
    if kind == 2 and stride != 1:
        raise NotImplementedError("pointwise convolution with stride != 1")
    if activation not in _ACT_CODES or affine not in _AFFINE_CODES:
        raise ValueError(f"unknown activation {activation!r} / affine {affine!r}")
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    lib = _lib.load()
    with _lib.on_device_of(x):
        dev = lambda t: None if t is None else t.detach().to(x.device, torch.float32).contiguous()  # noqa: E731
        xc, wc, bc, ac, btc, rc = x.contiguous(), weight.detach().contiguous(), dev(bias), dev(alpha), dev(beta), dev(residual)
        if rc is not None and tuple(rc.shape) != (n, cout, oh, ow):
            raise RuntimeError(f"residual of shape {tuple(rc.shape)} does not match the output {(n, cout, oh, ow)}")
        for t, name in ((bc, "bias"), (ac, "alpha"), (btc, "beta")):
            if t is not None and t.numel() != cout:
                raise RuntimeError(f"{name} has {t.numel()} elements for {cout} output channels")
        y = torch.empty((n, cout, oh, ow), dtype=torch.float32, device=x.device)
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        _lib.check(lib.mv_conv_norm_act_f32(kind, xc.data_ptr(), wc.data_ptr(), p(bc), p(ac), p(btc), p(rc), y.data_ptr(), n, cin, h, w,
                                            cout, stride, _AFFINE_CODES[affine], _ACT_CODES[activation], _lib.stream_ptr(xc)))
    return _lib.forward_only(y, "conv_norm_act", x, weight, bias, alpha, beta, residual)


def inverted_residual_k_slices(n: int, cin: int, hidden: int, cout: int, h: int, w: int, stride: int) -> Tuple[int, int]:
    """(slices, slice_len) of the fused InvertedResidual kernel for this shape (mv_inverted_residual_k_slices; host logic, no GPU):
    slices == 0 -> no fused kernel (the block runs as three conv_norm_act launches); otherwise the projection sums `slices`
    chains over `slice_len` hidden channels each, ascending from +0, added in ascending order."""
    sl = C.c_int(0)
    s = int(_lib.load().mv_inverted_residual_k_slices(int(n), int(cin), int(hidden), int(cout), int(h), int(w), int(stride), C.byref(sl)))
    return s, int(sl.value)


@functools.lru_cache(maxsize=512)
def _inverted_residual_workspace_bytes(lib_id: int, n: int, cin: int, hidden: int, cout: int, h: int, w: int, stride: int) -> int:
    """mv_inverted_residual_workspace_bytes, remembered per shape and loaded library (a pure function of both)."""
    return int(_lib.load().mv_inverted_residual_workspace_bytes(n, cin, hidden, cout, h, w, stride))


def inverted_residual(x: torch.Tensor, w_expand: torch.Tensor, a1: torch.Tensor, b1: torch.Tensor, w_dw: torch.Tensor, a2: torch.Tensor,
                      b2: torch.Tensor, w_project: torch.Tensor, a3: torch.Tensor, b3: torch.Tensor, residual: bool, stride: int = 1,
                      affine: str = "fma") -> torch.Tensor:
    """A whole InvertedResidual block (models/mobilenetv2.py:39-63) as one kernel: [1x1 expand + norm + ReLU6 ->] 3x3 depthwise
    (stride) + norm + ReLU6 -> 1x1 project + norm [-> + x]; the hidden tensor stays on the CU (csrc/invres.hip).  Shapes
    without a fused kernel raise Mi355VisionError (ask inverted_residual_k_slices first).  All tensors fp32 on the device;
    a*, b*: the folded norms of the three convolutions (`affine`: "fma" = BatchNorm2d eval, "mul_add" = FrozenBatchNorm2d)."""
    _lib.require_device(x)
    if x.ndim != 4 or x.dtype != torch.float32:
        raise TypeError(f"inverted_residual expects a float32 (N, C, H, W) tensor. Got {x.dtype} {tuple(x.shape)}")
    n, cin, h, w = (int(d) for d in x.shape)
    hidden, cout = int(w_dw.shape[0]), int(w_project.shape[0])
    if w_expand is None:  # a block without expansion (expand_ratio 1): the depthwise conv runs on x itself
        if hidden != cin or a1 is not None or b1 is not None:
            raise RuntimeError(f"a block without expansion has hidden == cin and no a1 / b1 (hidden {hidden}, cin {cin})")
    elif tuple(w_expand.shape[:2]) != (hidden, cin):
        raise RuntimeError(f"weights {tuple(w_expand.shape)}, {tuple(w_dw.shape)}, {tuple(w_project.shape)} do not form a block on {cin} channels")
    if tuple(w_dw.shape) != (hidden, 1, 3, 3) or tuple(w_project.shape[:2]) != (cout, hidden):
        raise RuntimeError(f"weights {tuple(w_dw.shape)}, {tuple(w_project.shape)} do not form a block on {cin} channels")
    if affine not in ("fma", "mul_add"):
        raise ValueError(f"affine should be 'fma' or 'mul_add'. Got {affine!r}")
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    lib = _lib.load()
    with _lib.on_device_of(x):
        def dev(t):  # already a contiguous fp32 tensor on the device (the usual case): no new tensor object per call
            if t is None or (t.device == x.device and t.dtype == torch.float32 and t.is_contiguous()):
                return t
            return t.detach().to(x.device, torch.float32).contiguous()
        xc = x.contiguous()
        ts = [dev(t) for t in (w_expand, a1, b1, w_dw, a2, b2, w_project, a3, b3)]
        for t, c, name in ((ts[1], hidden, "a1"), (ts[2], hidden, "b1"), (ts[4], hidden, "a2"), (ts[5], hidden, "b2"), (ts[7], cout, "a3"), (ts[8], cout, "b3")):
            if t is not None and t.numel() != c:
                raise RuntimeError(f"{name} has {t.numel()} elements for {c} channels")
        y = torch.empty((n, cout, oh, ow), dtype=torch.float32, device=x.device)
        nbytes = _inverted_residual_workspace_bytes(id(lib), n, cin, hidden, cout, h, w, stride)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
        _lib.check(lib.mv_inverted_residual_f32(xc.data_ptr(), *[None if t is None else t.data_ptr() for t in ts], int(bool(residual)), y.data_ptr(), n, cin, hidden,
                                                cout, h, w, stride, _AFFINE_CODES[affine], None if ws is None else ws.data_ptr(), nbytes,
                                                _lib.stream_ptr(xc)))
    return _lib.forward_only(y, "inverted_residual", x, *([] if w_expand is None else [w_expand]), w_dw, w_project)


# --------------------------------------------------------------------------------------------- the step before the path (8f.2)
def _mean_std(mean, std, c: int):
    m = [float(v) for v in (mean if isinstance(mean, (list, tuple)) else [mean])]
    s = [float(v) for v in (std if isinstance(std, (list, tuple)) else [std])]
    if len(m) == 1:
        m = m * c
    if len(s) == 1:
        s = s * c
    if len(m) != c or len(s) != c:
        raise RuntimeError(f"The size of mean/std ({len(m)}/{len(s)}) must match the number of channels ({c})")
    # ATen narrows the Python doubles to the tensor dtype: torch.as_tensor(mean, dtype=float32) (_misc.py:56-57)
    return _lib.taps(m), _lib.taps(s)


# --------------------------------------------------------------------------------------------- resize / center_crop (v2 surface)
def resize(inpt: torch.Tensor, size: Optional[List[int]], interpolation="bilinear", max_size: Optional[int] = None,
           antialias: Optional[bool] = True) -> torch.Tensor:
    """Dispatcher, as transforms/v2/functional/_geometry.py:172-186."""
    kernel = _get_kernel(resize, type(inpt))
    return kernel(inpt, size=size, interpolation=interpolation, max_size=max_size, antialias=antialias)


@_register_kernel_internal(resize, torch.Tensor)
@_register_kernel_internal(resize, tv_tensors.Image)
def resize_image(image: torch.Tensor, size: Optional[List[int]], interpolation="bilinear", max_size: Optional[int] = None,
                 antialias: Optional[bool] = True) -> torch.Tensor:
    """resize_image (_geometry.py:189-262) for what the presets use: bilinear with antialias.  On a device tensor the
    reference casts uint8 to float32, interpolates, rounds and narrows (:222-254) -- the v1 tensor path's arithmetic
    (_functional_tensor.py:441-474), which mv_resize_bilinear_aa_* reproduces bit for bit."""
    from . import functional_v1
    if size is None:
        if not isinstance(max_size, int):
            raise ValueError(f"max_size must be an integer when size is None, but got {max_size} instead.")
        h, w = int(image.shape[-2]), int(image.shape[-1])  # _geometry.py:117-121: the longer edge becomes max_size
        size = [max_size, int(max_size * w / h)] if h >= w else [int(max_size * h / w), max_size]
        max_size = None
    return functional_v1.resize(image, size, interpolation, max_size, antialias)


@_register_kernel_internal(resize, tv_tensors.Video)
def resize_video(video: torch.Tensor, size: Optional[List[int]], interpolation="bilinear", max_size: Optional[int] = None,
                 antialias: Optional[bool] = True) -> torch.Tensor:
    return resize_image(video, size, interpolation=interpolation, max_size=max_size, antialias=antialias)


def center_crop(inpt: torch.Tensor, output_size: List[int]) -> torch.Tensor:
    """Dispatcher, as transforms/v2/functional/_geometry.py:1811-1824."""
    kernel = _get_kernel(center_crop, type(inpt))
    return kernel(inpt, output_size=output_size)


@_register_kernel_internal(center_crop, torch.Tensor)
@_register_kernel_internal(center_crop, tv_tensors.Image)
@_register_kernel_internal(center_crop, tv_tensors.Video)
def center_crop_image(image: torch.Tensor, output_size: List[int]) -> torch.Tensor:
    """center_crop_image (_geometry.py:1860-1880): a view when the box lies inside the image, zero padding otherwise."""
    from . import functional_v1
    return functional_v1.center_crop(image, output_size)


def to_dtype(inpt: torch.Tensor, dtype: torch.dtype = torch.float, scale: bool = False) -> torch.Tensor:
    """Dispatcher, as transforms/v2/functional/_misc.py:222-231."""
    kernel = _get_kernel(to_dtype, type(inpt))
    return kernel(inpt, dtype=dtype, scale=scale)


@_register_kernel_internal(to_dtype, torch.Tensor)
@_register_kernel_internal(to_dtype, tv_tensors.Image)
def to_dtype_image(image: torch.Tensor, dtype: torch.dtype = torch.float, scale: bool = False) -> torch.Tensor:
    """to_dtype_image (_misc.py:250-309) for the conversion the inference preset makes: uint8 -> float32 with
    scale=True (image.to(float32).mul_(1/255)) as one kernel.  Same-dtype and scale=False requests are the plain
    casts the reference does; other scaled conversions are outside SURVEY.md section 8 and raise."""
    if image.dtype == dtype:
        return image
    if not scale:
        return image.to(dtype)
    if image.dtype == torch.uint8 and dtype == torch.float32:
        if image.numel() == 0:
            return image.to(dtype)
        _lib.require_device(image)
        lib = _lib.load()
        with _lib.on_device_of(image):
            x = image.contiguous()
            y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
            _lib.check(lib.mv_to_float_normalize_u8(x.data_ptr(), y.data_ptr(), 1, 1, x.numel(), None, None, _lib.stream_ptr(x)))
        return y
    raise NotImplementedError(f"scaled conversion {image.dtype} -> {dtype} is not on the MI355X hot path (uint8 -> float32 is)")


@_register_kernel_internal(to_dtype, tv_tensors.BoundingBoxes, tv_tensor_wrapper=False)
@_register_kernel_internal(to_dtype, tv_tensors.Mask, tv_tensor_wrapper=False)
def _to_dtype_tensor_dispatch(inpt: torch.Tensor, dtype: torch.dtype, scale: bool = False) -> torch.Tensor:
    """_misc.py:326-330: masks and boxes are only cast (values are never rescaled)."""
    return inpt.to(dtype)


def normalize(inpt: torch.Tensor, mean: List[float], std: List[float], inplace: bool = False) -> torch.Tensor:
    """Dispatcher, as transforms/v2/functional/_misc.py:19-32."""
    kernel = _get_kernel(normalize, type(inpt))
    return kernel(inpt, mean=mean, std=std, inplace=inplace)


@_register_kernel_internal(normalize, torch.Tensor)
@_register_kernel_internal(normalize, tv_tensors.Image)
def normalize_image(image: torch.Tensor, mean: List[float], std: List[float], inplace: bool = False) -> torch.Tensor:
    """normalize_image (_misc.py:35-67): (image - mean[c]) / std[c] on (..., C, H, W) float images."""
    if not image.is_floating_point():
        raise TypeError(f"Input tensor should be a float tensor. Got {image.dtype}.")
    if image.ndim < 3:
        raise ValueError(f"Expected tensor to be a tensor image of size (..., C, H, W). Got {image.shape}.")
    if isinstance(std, (tuple, list)):
        divzero = not all(std)
    elif isinstance(std, (int, float)):
        divzero = std == 0
    else:
        divzero = False
    if divzero:
        raise ValueError("std evaluated to zero, leading to division by zero.")
    if image.numel() == 0:
        return image
    _lib.require_device(image)
    c = int(image.shape[-3])
    m, s = _mean_std(mean, std, c)
    n = int(image.numel() // (c * image.shape[-1] * image.shape[-2]))
    hw = int(image.shape[-1] * image.shape[-2])
    lib = _lib.load()
    with _lib.on_device_of(image):
        x = image.to(torch.float32).contiguous()
        y = torch.empty_like(x)
        _lib.check(lib.mv_normalize_f32(x.data_ptr(), y.data_ptr(), n, c, hw, m, s, _lib.stream_ptr(x)))
        y = y if image.dtype == torch.float32 else y.to(image.dtype)
        if inplace:
            image.copy_(y)
            return image
    return y


def to_float_normalize(image: torch.Tensor, mean: List[float], std: List[float]) -> torch.Tensor:
    """normalize(to_dtype(image, float32, scale=True), mean, std) -- the tail of the ImageClassification preset
    (transforms/_presets.py:58-60) -- as ONE pass over a uint8 (..., C, H, W) image: 1 B read + 4 B written."""
    if image.dtype != torch.uint8:
        raise TypeError(f"to_float_normalize expects a uint8 image. Got {image.dtype}")
    if image.ndim < 3:
        raise ValueError(f"Expected tensor to be a tensor image of size (..., C, H, W). Got {image.shape}.")
    if isinstance(std, (tuple, list)) and not all(std):
        raise ValueError("std evaluated to zero, leading to division by zero.")
    _lib.require_device(image)
    c = int(image.shape[-3])
    m, s = _mean_std(mean, std, c)
    hw = int(image.shape[-1] * image.shape[-2])
    n = int(image.numel() // max(c * hw, 1))
    lib = _lib.load()
    with _lib.on_device_of(image):
        x = image.contiguous()
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        _lib.check(lib.mv_to_float_normalize_u8(x.data_ptr(), y.data_ptr(), n, c, hw, m, s, _lib.stream_ptr(x)))
    return y


def normalized_conv2d_bias_relu(image_u8: torch.Tensor, mean: List[float], std: List[float], weight: torch.Tensor,
                                bias: Optional[torch.Tensor] = None, relu: bool = True) -> torch.Tensor:
    """relu(conv2d(normalize(to_dtype(image_u8, float32, scale=True), mean, std), weight, bias, padding=1)) with the
    conversion and normalisation fused into the conv's load: the uint8 (N, 3, H, W) batch is read once (1 B/element)
    and no fp32 copy of it is ever written."""
    if image_u8.dtype != torch.uint8 or image_u8.ndim != 4 or image_u8.shape[1] != 3:
        raise TypeError(f"expected a uint8 (N, 3, H, W) batch. Got {image_u8.dtype} {tuple(image_u8.shape)}")
    if weight.ndim != 4 or tuple(weight.shape[1:]) != (3, 3, 3):
        raise ValueError(f"weight should have shape (Cout, 3, 3, 3). Got {tuple(weight.shape)}")
    _lib.require_device(image_u8)
    _lib.require_device(weight, "weight")
    n, _, h, w = (int(d) for d in image_u8.shape)
    cout = int(weight.shape[0])
    m, s = _mean_std(mean, std, 3)
    lib = _lib.load()
    with _lib.on_device_of(image_u8):
        xc, wc = image_u8.contiguous(), weight.detach().to(torch.float32).contiguous()
        bc = None if bias is None else bias.detach().to(image_u8.device, torch.float32).contiguous()
        y = torch.empty((n, cout, h, w), dtype=torch.float32, device=image_u8.device)
        _lib.check(lib.mv_conv3x3_bias_relu_u8norm_f32(xc.data_ptr(), m, s, wc.data_ptr(), None if bc is None else bc.data_ptr(),
                                                       y.data_ptr(), n, h, w, cout, int(relu), _lib.stream_ptr(xc)))
    return _lib.forward_only(y, "normalized_conv2d_bias_relu", weight, bias)
