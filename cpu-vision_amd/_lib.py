"""ctypes binding of libmi355vision.so (include/mi355vision.h) -- the only way the Python layer computes.

There is deliberately no CPU or PyTorch fallback: if the library is missing, cannot be loaded, or is
handed a tensor that does not live on a HIP device, the call raises.  PyTorch is used for device
memory, streams and dtype plumbing only.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional, Sequence

import torch

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MI355VISION_LIB", _HERE / "lib" / "libmi355vision.so"))

BORDER_VALID, BORDER_REFLECT, BORDER_ZERO = 0, 1, 2
BORDERS = {"valid": BORDER_VALID, "reflect": BORDER_REFLECT, "zero": BORDER_ZERO, "zeros": BORDER_ZERO}
MAX_HOST_TAPS_2D = 121
MAX_TAPS_1D = 63

# every symbol include/mi355vision.h declares: (name, restype, argtypes)
_vp, _i, _i64, _d = C.c_void_p, C.c_int, C.c_int64, C.c_double
_fp = C.POINTER(C.c_float)
SYMBOLS = {
    "mv_abi_version": (_i, []),
    "mv_last_error": (C.c_char_p, []),
    "mv_last_kernel": (C.c_char_p, []),
    "mv_build_id": (C.c_char_p, []),
    "mv_device_count": (_i, []),
    "mv_depthwise_conv2d_f32": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _vp]),
    "mv_depthwise_conv2d_u8": (_i, [_vp, _vp, _vp, _i, _i64, _i, _i, _i, _i, _i, _vp]),
    "mv_gaussian_blur_f32": (_i, [_vp, _vp, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_gaussian_blur_u8": (_i, [_vp, _vp, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_gaussian_blur_u8_workspace_bytes": (_i64, [_i64, _i, _i, _i, _i]),
    "mv_gaussian_blur_u8_ws": (_i, [_vp, _vp, _i64, _i, _i, _fp, _i, _fp, _i, _vp, _i64, _vp]),
    "mv_separable_blur_f32": (_i, [_vp, _vp, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_gaussian_blur_f16": (_i, [_vp, _vp, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_gaussian_blur_bf16": (_i, [_vp, _vp, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_gaussian_blur_f64": (_i, [_vp, _vp, _i64, _i, _i, C.POINTER(C.c_double), _i, C.POINTER(C.c_double), _i, _vp]),
    "mv_depthwise_conv2d_f64": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    "mv_sharpness_f64": (_i, [_vp, _vp, _i64, _i, _i, _d, _i, _vp]),
    "mv_gaussian_blur_f32_v": (_i, [_vp, _vp, _i, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_gaussian_blur_u8_v": (_i, [_vp, _vp, _i, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_separable_blur_f32_v": (_i, [_vp, _vp, _i, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_separable_blur_u8_v": (_i, [_vp, _vp, _i, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_sharpness_f32_v": (_i, [_vp, _vp, _i, _i64, _i, _i, _d, _i, C.c_float, _i, _vp]),
    "mv_sharpness_u8_v": (_i, [_vp, _vp, _i, _i64, _i, _i, _d, _i, _vp]),
    "mv_conv1x1_k_slices": (_i, [_i64, _i, _i, _i, _i, C.POINTER(C.c_int)]),
    "mv_separable_blur_u8": (_i, [_vp, _vp, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_sobel_f32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "mv_gaussian_sobel_f32": (_i, [_vp, _vp, _vp, _i64, _i, _i, _fp, _i, _fp, _i, _vp]),
    "mv_sharpness_f32": (_i, [_vp, _vp, _i64, _i, _i, _d, _i, C.c_float, _i, _vp]),
    "mv_sharpness_u8": (_i, [_vp, _vp, _i64, _i, _i, _d, _i, _vp]),
    "mv_conv3x3_bias_relu_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    "mv_to_float_normalize_u8": (_i, [_vp, _vp, _i64, _i, _i64, _fp, _fp, _vp]),
    "mv_normalize_f32": (_i, [_vp, _vp, _i64, _i, _i64, _fp, _fp, _vp]),
    "mv_conv3x3_bias_relu_u8norm_f32": (_i, [_vp, _fp, _fp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "mv_linear_bias_relu_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp]),
    "mv_linear_k_slices": (_i, [_i64, _i, _i, C.POINTER(C.c_int)]),
    "mv_linear_workspace_bytes": (_i64, [_i64, _i, _i]),
    "mv_linear_bias_relu_ws_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp, _i64, _vp]),
    "mv_conv_norm_act_f32": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mv_inverted_residual_k_slices": (_i, [_i64, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_int)]),
    "mv_inverted_residual_workspace_bytes": (_i64, [_i64, _i, _i, _i, _i, _i, _i]),
    "mv_inverted_residual_f32": (_i, [_vp] * 10 + [_i, _vp, _i64] + [_i] * 7 + [_vp, _i64, _vp]),
    "mv_fold_batchnorm": (None, [_fp, _fp, _fp, _fp, C.c_double, _i, _fp, _fp]),
    "mv_deform_conv2d_workspace_bytes": (_i64, [_i64] + [_i] * 11),
    "mv_conv3x3_k_slices": (_i, [_i64, _i, _i, _i, _i, _vp]),
    "mv_conv3x3_workspace_bytes": (_i64, [_i64, _i, _i, _i, _i]),
    "mv_conv3x3_bias_relu_ws_f32": (_i, [_vp] * 4 + [_i64] + [_i] * 5 + [_vp, _i64, _vp]),
    "mv_deform_conv2d_needs_workspace": (_i, [_i64] + [_i] * 14),
    "mv_deform_conv2d_f32": (_i, [_vp] * 6 + [_i64] + [_i] * 15 + [_vp, _i64, _vp]),
    "mv_conv2d_needs_workspace": (_i, [_i64] + [_i] * 13),
    "mv_conv2d_bias_act_f32": (_i, [_vp] * 4 + [_i64] + [_i] * 14 + [_vp, _i64, _vp]),
    "mv_maxpool2d_f32": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "mv_resize_workspace_bytes": (_i64, [_i64, _i, _i, _i, _i, _i, _i, _i, _i]),
    "mv_resize_bilinear_aa_u8": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "mv_resize_bilinear_aa_f32": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i64, _vp]),
    "mv_preset_classification_u8": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _vp, _i64, _vp]),
    "mv_preset_classification_f32": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _vp, _i64, _vp]),
    "mv_maxpool2x2_f32": (_i, [_vp, _vp, _i64, _i, _i, _vp]),
    "mv_adaptive_avgpool_f32": (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp]),
}

_lib: Optional[C.CDLL] = None


class Mi355VisionError(RuntimeError):
    """The native library is missing / failed, or was asked to run off-device."""


def _open(path: Path) -> C.CDLL:
    if not path.exists():
        raise Mi355VisionError(
            f"{path} not found: build it with `python cpu-vision_amd/_build.py` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(str(path))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.mv_abi_version() != 1:
        raise Mi355VisionError(f"ABI version mismatch: library reports {lib.mv_abi_version()}, binding expects 1")
    return lib


def load() -> C.CDLL:
    """Load libmi355vision.so (once).  Raises if it is not built: there is no fallback path."""
    global _lib
    if _lib is None:
        _lib = _open(LIB_PATH)
    return _lib


TUNING_LIB_PATH = _HERE / "lib" / "libmi355vision_tuning.so"


class tuning_library:
    """Context manager for tools/ and for the tests that force an alternative kernel: inside it every call of the
    package goes to the -DMV_TUNING build of the same sources, the only build that reads MV_* environment knobs
    (the product library has no environment lookup at all)."""

    def __init__(self, path: Optional[Path] = None):
        self._path = Path(path) if path else TUNING_LIB_PATH
        self._saved = None

    def __enter__(self) -> C.CDLL:
        global _lib
        self._saved = _lib
        _lib = _open(self._path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._saved
        return False


def last_kernel() -> str:
    """The kernel instantiation the calling thread's last entry point launched (mv_last_kernel)."""
    return load().mv_last_kernel().decode()


def build_id() -> str:
    return load().mv_build_id().decode()


def _raise(rc: int):
    msg = load().mv_last_error().decode("utf-8", "replace")
    if rc == -1:
        raise ValueError(msg)
    raise Mi355VisionError(f"libmi355vision error {rc}: {msg}")


def check(rc: int):
    if rc != 0:
        _raise(rc)


def require_device(t: torch.Tensor, what: str = "input") -> None:
    if not t.is_cuda:
        raise Mi355VisionError(
            f"{what} lives on '{t.device}': mi355vision kernels run on the MI355X only and there is no CPU "
            "fallback -- move the tensor to a HIP device (tensor.cuda()).")


class _ForwardOnly(torch.autograd.Function):
    """Marks the output of a forward-only kernel: it takes part in autograd (so that nothing downstream silently trains on
    missing gradients) and raises as soon as a backward pass reaches it."""

    @staticmethod
    def forward(ctx, out, what, *deps):
        ctx.what = what
        return out.view_as(out)

    @staticmethod
    def backward(ctx, *grads):
        raise RuntimeError(
            f"{ctx.what} is forward-only on the MI355X (inference / data-augmentation path): there is no backward kernel. "
            "Run it under torch.no_grad() or on tensors that do not require grad.")


def forward_only(out: torch.Tensor, what: str, *deps) -> torch.Tensor:
    """`out` was computed by a forward-only kernel from `deps`.  When autograd is recording and any of them requires grad,
    return a result whose backward raises (the reference would compute gradients here); otherwise `out` itself."""
    if torch.is_grad_enabled() and any(isinstance(d, torch.Tensor) and d.requires_grad for d in deps):
        return _ForwardOnly.apply(out, what, *[d for d in deps if isinstance(d, torch.Tensor)])
    return out


class _DeviceOf:
    """`with torch.cuda.device(t.device)` only when t is not on the current device (the context manager costs
    ~10 us of host time per call, which matters for launch-bound per-frame use)."""

    __slots__ = ("_ctx",)

    def __init__(self, t: torch.Tensor):
        idx = t.device.index
        self._ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(t.device)

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def on_device_of(t: torch.Tensor) -> _DeviceOf:
    return _DeviceOf(t)


def stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def taps(values: Sequence[float]):
    arr = (C.c_float * len(values))(*[float(v) for v in values])
    return arr


def pointer_table(tensors):
    """HOST array of device pointers (void*[n]) for the mv_*_v entry points."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def taps64_from_tensor(t: torch.Tensor):
    """Host double array from a (CPU, fp64) tensor of taps -- bit-preserving."""
    flat = t.detach().to("cpu", torch.float64).contiguous().reshape(-1)
    arr = (C.c_double * flat.numel())()
    C.memmove(arr, flat.data_ptr(), flat.numel() * 8)
    return arr


def taps_from_tensor(t: torch.Tensor):
    """Host float array from a (CPU, fp32) tensor of taps -- bit-preserving."""
    flat = t.detach().to("cpu", torch.float32).contiguous().reshape(-1)
    arr = (C.c_float * flat.numel())()
    C.memmove(arr, flat.data_ptr(), flat.numel() * 4)
    return arr
