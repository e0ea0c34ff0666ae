"""numpy front-end of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of oracle.c.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; the product package (``cpu-vision_amd/``) never does.

Every function takes/returns contiguous numpy arrays in planar ``(..., H, W)``
layout and mirrors one reference function (cited in oracle.c).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "liboracle.so"

BORDER_VALID, BORDER_REFLECT, BORDER_ZERO = 0, 1, 2


_ARCH_NEEDS = {"x86-64-v3": {"avx2", "fma", "bmi2", "f16c", "movbe"}, "x86-64": set()}  # Makefile ARCH -> cpu flags it needs


def _host_flags() -> set:
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("flags"):
                return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def build(force: bool = False) -> Path:
    """Compile oracle.c with gcc (a few hundred ms).  Building the checker is not using it.  Rebuilds when the source is
    newer than the library, or when the library was built for an instruction-set level this host lacks (a .so that came
    with the snapshot from another machine must not SIGILL here: that would read as a parity failure)."""
    src = _HERE / "oracle.c"
    stamp = _SO.parent / "arch"
    flags = _host_flags()
    built_for = stamp.read_text().strip() if stamp.exists() else None
    runnable = built_for in _ARCH_NEEDS and _ARCH_NEEDS[built_for] <= flags
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime or not runnable:
        arch = "x86-64-v3" if _ARCH_NEEDS["x86-64-v3"] <= flags else "x86-64"
        subprocess.run(["make", "-C", str(_HERE), "-B", "-s", f"ARCH={arch}"], check=True, stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(_SO))
        fp, u8p, i, l, d = C.c_void_p, C.c_void_p, C.c_int, C.c_long, C.c_double
        _lib.orc_gaussian_kernel1d_v2.argtypes = [i, d, fp]
        _lib.orc_gaussian_kernel1d_v1.argtypes = [i, d, fp]
        _lib.orc_gaussian_kernel2d.argtypes = [fp, i, fp, i, fp]
        _lib.orc_depthwise_conv2d_f32.argtypes = [fp, fp, fp, l, i, i, i, i, i]
        _lib.orc_depthwise_conv2d_pc_f32.argtypes = [fp, fp, fp, l, i, i, i, i, i, i]
        _lib.orc_gaussian_blur_f32.argtypes = [fp, fp, l, i, i, fp, i, fp, i]
        _lib.orc_gaussian_blur_u8.argtypes = [u8p, u8p, l, i, i, fp, i, fp, i]
        _lib.orc_separable_blur_f32.argtypes = [fp, fp, l, i, i, fp, i, fp, i]
        _lib.orc_separable_blur_u8.argtypes = [u8p, u8p, l, i, i, fp, i, fp, i]
        _lib.orc_sobel_f32.argtypes = [fp, fp, fp, l, i, i, i]
        _lib.orc_gaussian_sobel_f32.argtypes = [fp, fp, fp, l, i, i, fp, i, fp, i]
        _lib.orc_sharpness_f32.argtypes = [fp, fp, l, i, i, d, i]
        _lib.orc_sharpness_u8.argtypes = [u8p, u8p, l, i, i, d, i]
        dp = C.c_void_p
        _lib.orc_depthwise_conv2d_f64.argtypes = [dp, dp, dp, l, i, i, i, i, i]
        _lib.orc_gaussian_blur_f64.argtypes = [dp, dp, l, i, i, dp, i, dp, i]
        _lib.orc_sharpness_f64.argtypes = [dp, dp, l, i, i, d, i]
        _lib.orc_conv3x3_bias_relu_f32.argtypes = [fp, fp, fp, fp, l, i, i, i, i, i]
        _lib.orc_conv3x3_sliced_bias_relu_f32.argtypes = [fp, fp, fp, fp, l, i, i, i, i, i, i]
        _lib.orc_maxpool2x2_f32.argtypes = [fp, fp, l, i, i]
        _lib.orc_adaptive_avgpool_f32.argtypes = [fp, fp, l, i, i, i, i]
        _lib.orc_linear_bias_relu_f32.argtypes = [fp, fp, fp, fp, l, i, i, i]
        _lib.orc_linear_sliced_bias_relu_f32.argtypes = [fp, fp, fp, fp, l, i, i, i, i]
        _lib.orc_to_float_normalize_u8.argtypes = [u8p, fp, l, i, l, fp, fp, i]
        _lib.orc_normalize_f32.argtypes = [fp, fp, l, i, l, fp, fp]
        _lib.orc_fold_batchnorm.argtypes = [fp, fp, fp, fp, d, i, fp, fp]
        _lib.orc_fold_batchnorm.restype = None
        _lib.orc_conv2d_affine_act_f32.argtypes = [fp, fp, fp, fp, fp, fp, fp, l, i, i, i, i, i, i, i, i, i, i, i]
        _lib.orc_pointwise_sliced_affine_act_f32.argtypes = [fp, fp, fp, fp, fp, fp, fp, l, i, l, i, i, i, i]
        _lib.orc_maxpool2d_f32.argtypes = [fp, fp, l, i, i, i, i]
        _lib.orc_deform_conv2d_f32.argtypes = [fp, fp, fp, fp, fp, fp, l] + [i] * 15
        _lib.orc_resize_bilinear_aa_f32.argtypes = [fp, fp, l, i, i, i, i]
        _lib.orc_resize_bilinear_aa_u8.argtypes = [u8p, u8p, l, i, i, i, i]
        _lib.orc_set_num_threads.argtypes = [i]
        _lib.orc_num_threads.restype = i
    return _lib


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(int(n))


def num_threads() -> int:
    return int(lib().orc_num_threads())


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _planes(a: np.ndarray):
    if a.ndim < 2:
        raise ValueError("need (..., H, W)")
    h, w = a.shape[-2:]
    planes = int(np.prod(a.shape[:-2], dtype=np.int64))
    return planes, int(h), int(w)


def _check(rc: int, what: str):
    if rc != 0:
        raise ValueError(f"oracle {what} rejected its arguments (rc={rc})")


# --------------------------------------------------------------------------- kernels
def gaussian_kernel1d(k: int, sigma: float, v1: bool = False) -> np.ndarray:
    out = np.empty(k, np.float32)
    (lib().orc_gaussian_kernel1d_v1 if v1 else lib().orc_gaussian_kernel1d_v2)(k, float(sigma), _p(out))
    return out


def gaussian_kernel2d(k1d_x: np.ndarray, k1d_y: np.ndarray) -> np.ndarray:
    k1d_x, k1d_y = _f32(k1d_x), _f32(k1d_y)
    out = np.empty((len(k1d_y), len(k1d_x)), np.float32)
    lib().orc_gaussian_kernel2d(_p(k1d_x), len(k1d_x), _p(k1d_y), len(k1d_y), _p(out))
    return out


def default_sigma(k: int) -> float:
    """_misc.py:116-117: sigma = ksize * 0.15 + 0.35"""
    return k * 0.15 + 0.35


# --------------------------------------------------------------------------- filters
def depthwise_conv2d(x: np.ndarray, w: np.ndarray, border: int = BORDER_REFLECT) -> np.ndarray:
    """pad(border) + conv2d(groups=C) with one shared (ky,kx) kernel."""
    x, w = _f32(x), _f32(w)
    ky, kx = w.shape
    planes, h, wd = _planes(x)
    if border == BORDER_VALID:
        y = np.empty(x.shape[:-2] + (h - ky + 1, wd - kx + 1), np.float32)
    else:
        y = np.empty_like(x)
    if x.size:
        _check(lib().orc_depthwise_conv2d_f32(_p(x), _p(y), _p(w), planes, h, wd, ky, kx, border), "depthwise_conv2d")
    return y


def depthwise_conv2d_per_channel(x: np.ndarray, w: np.ndarray, border: int = BORDER_ZERO) -> np.ndarray:
    """x (N,C,H,W), w (C,ky,kx): conv2d(groups=C) with a distinct kernel per channel."""
    x, w = _f32(x), _f32(w)
    n, c, h, wd = x.shape
    _, ky, kx = w.shape
    if border == BORDER_VALID:
        y = np.empty((n, c, h - ky + 1, wd - kx + 1), np.float32)
    else:
        y = np.empty_like(x)
    if x.size:
        _check(lib().orc_depthwise_conv2d_pc_f32(_p(x), _p(y), _p(w), n, c, h, wd, ky, kx, border), "depthwise_conv2d_pc")
    return y


def gaussian_blur(x: np.ndarray, k1d_x: np.ndarray, k1d_y: np.ndarray) -> np.ndarray:
    """gaussian_blur_image core (2-D outer-product kernel, reflect); uint8 or float32."""
    k1d_x, k1d_y = _f32(k1d_x), _f32(k1d_y)
    planes, h, wd = _planes(x)
    if x.dtype == np.uint8:
        x = np.ascontiguousarray(x)
        y = np.empty_like(x)
        if x.size:
            _check(lib().orc_gaussian_blur_u8(_p(x), _p(y), planes, h, wd, _p(k1d_x), len(k1d_x), _p(k1d_y), len(k1d_y)), "gaussian_blur_u8")
        return y
    x = _f32(x)
    y = np.empty_like(x)
    if x.size:
        w2 = gaussian_kernel2d(k1d_x, k1d_y)
        _check(lib().orc_depthwise_conv2d_f32(_p(x), _p(y), _p(w2), planes, h, wd, len(k1d_y), len(k1d_x), BORDER_REFLECT), "gaussian_blur_f32")
    return y


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def gaussian_blur_f64(x: np.ndarray, k1d_x: np.ndarray, k1d_y: np.ndarray) -> np.ndarray:
    """gaussian_blur_image on a float64 image: float64 taps (built by the caller as the reference builds them), float64
    outer product, float64 accumulation."""
    x, k1d_x, k1d_y = _f64(x), _f64(k1d_x), _f64(k1d_y)
    planes, h, wd = _planes(x)
    y = np.empty_like(x)
    if x.size:
        _check(lib().orc_gaussian_blur_f64(_p(x), _p(y), planes, h, wd, _p(k1d_x), len(k1d_x), _p(k1d_y), len(k1d_y)), "gaussian_blur_f64")
    return y


def depthwise_conv2d_f64(x: np.ndarray, w: np.ndarray, border: int = BORDER_REFLECT) -> np.ndarray:
    x, w = _f64(x), _f64(w)
    ky, kx = w.shape
    planes, h, wd = _planes(x)
    y = np.empty(x.shape[:-2] + (h - ky + 1, wd - kx + 1), np.float64) if border == BORDER_VALID else np.empty_like(x)
    if x.size:
        _check(lib().orc_depthwise_conv2d_f64(_p(x), _p(y), _p(w), planes, h, wd, ky, kx, border), "depthwise_conv2d_f64")
    return y


def adjust_sharpness_f64(x: np.ndarray, factor: float, v1: bool = False) -> np.ndarray:
    x = _f64(x)
    planes, h, wd = _planes(x)
    y = np.empty_like(x)
    if x.size:
        _check(lib().orc_sharpness_f64(_p(x), _p(y), planes, h, wd, float(factor), int(v1)), "sharpness_f64")
    return y


def separable_blur(x: np.ndarray, k1d_x: np.ndarray, k1d_y: np.ndarray) -> np.ndarray:
    x, k1d_x, k1d_y = _f32(x), _f32(k1d_x), _f32(k1d_y)
    planes, h, wd = _planes(x)
    y = np.empty_like(x)
    if x.size:
        _check(lib().orc_separable_blur_f32(_p(x), _p(y), planes, h, wd, _p(k1d_x), len(k1d_x), _p(k1d_y), len(k1d_y)), "separable_blur")
    return y


def separable_blur_u8(x: np.ndarray, k1d_x: np.ndarray, k1d_y: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.uint8)
    k1d_x, k1d_y = _f32(k1d_x), _f32(k1d_y)
    planes, h, wd = _planes(x)
    y = np.empty_like(x)
    if x.size:
        _check(lib().orc_separable_blur_u8(_p(x), _p(y), planes, h, wd, _p(k1d_x), len(k1d_x), _p(k1d_y), len(k1d_y)), "separable_blur_u8")
    return y


def sobel(x: np.ndarray, border: int = BORDER_REFLECT):
    x = _f32(x)
    planes, h, wd = _planes(x)
    if border == BORDER_VALID:
        shp = x.shape[:-2] + (h - 2, wd - 2)
    else:
        shp = x.shape
    gx, gy = np.empty(shp, np.float32), np.empty(shp, np.float32)
    if x.size:
        _check(lib().orc_sobel_f32(_p(x), _p(gx), _p(gy), planes, h, wd, border), "sobel")
    return gx, gy


def gaussian_sobel(x: np.ndarray, k1d_x: np.ndarray, k1d_y: np.ndarray):
    x, k1d_x, k1d_y = _f32(x), _f32(k1d_x), _f32(k1d_y)
    planes, h, wd = _planes(x)
    gx, gy = np.empty_like(x), np.empty_like(x)
    if x.size:
        _check(lib().orc_gaussian_sobel_f32(_p(x), _p(gx), _p(gy), planes, h, wd, _p(k1d_x), len(k1d_x), _p(k1d_y), len(k1d_y)), "gaussian_sobel")
    return gx, gy


def box_filter(x: np.ndarray, k: int = 3, border: int = BORDER_REFLECT) -> np.ndarray:
    w = np.full((k, k), np.float32(1.0) / np.float32(k * k), np.float32)
    return depthwise_conv2d(x, w, border)


def adjust_sharpness(x: np.ndarray, factor: float, v1: bool = False) -> np.ndarray:
    planes, h, wd = _planes(x)
    if x.dtype == np.uint8:
        x = np.ascontiguousarray(x)
        y = np.empty_like(x)
        if x.size:
            _check(lib().orc_sharpness_u8(_p(x), _p(y), planes, h, wd, float(factor), int(v1)), "sharpness_u8")
        return y
    x = _f32(x)
    y = np.empty_like(x)
    if x.size:
        _check(lib().orc_sharpness_f32(_p(x), _p(y), planes, h, wd, float(factor), int(v1)), "sharpness_f32")
    return y


def conv3x3_bias_relu(x: np.ndarray, w: np.ndarray, b, relu: bool = True, slice_channels: int = 0) -> np.ndarray:
    """Conv2d(Cin,Cout,3,padding=1) [+bias] [+ReLU]; x (N,Cin,H,W), w (Cout,Cin,3,3).
    slice_channels > 0: the sliced summation order the library states for a small launch (mv_conv3x3_k_slices): chains over
    slices of that many input channels, added in ascending order, then bias and ReLU."""
    x, w = _f32(x), _f32(w)
    n, cin, h, wd = x.shape
    cout = w.shape[0]
    assert w.shape == (cout, cin, 3, 3)
    bb = None if b is None else _f32(b)
    y = np.empty((n, cout, h, wd), np.float32)
    if y.size:
        if slice_channels and slice_channels < cin:
            _check(lib().orc_conv3x3_sliced_bias_relu_f32(_p(x), _p(w), None if bb is None else _p(bb), _p(y), n, cin, h, wd, cout, int(relu),
                                                          int(slice_channels)), "conv3x3_sliced_bias_relu")
        else:
            _check(lib().orc_conv3x3_bias_relu_f32(_p(x), _p(w), None if bb is None else _p(bb), _p(y), n, cin, h, wd, cout, int(relu)), "conv3x3_bias_relu")
    return y


def maxpool2x2(x: np.ndarray) -> np.ndarray:
    """nn.MaxPool2d(kernel_size=2, stride=2) over the last two dims."""
    x = _f32(x)
    planes, h, wd = _planes(x)
    y = np.empty(x.shape[:-2] + (h // 2, wd // 2), np.float32)
    if y.size:
        _check(lib().orc_maxpool2x2_f32(_p(x), _p(y), planes, h, wd), "maxpool2x2")
    return y


def adaptive_avgpool(x: np.ndarray, oh: int, ow: int) -> np.ndarray:
    x = _f32(x)
    planes, h, wd = _planes(x)
    y = np.empty(x.shape[:-2] + (oh, ow), np.float32)
    if y.size:
        _check(lib().orc_adaptive_avgpool_f32(_p(x), _p(y), planes, h, wd, oh, ow), "adaptive_avgpool")
    return y


def linear_bias_relu(x: np.ndarray, w: np.ndarray, b, relu: bool = False, slice_len: int = 0) -> np.ndarray:
    """nn.Linear [+ ReLU]: x (N, K), w (M, K), b (M) or None.  slice_len > 0: the sliced-K summation order (partial chains
    over contiguous K slices, added in ascending order)."""
    x, w = _f32(x), _f32(w)
    n, k = x.shape
    m = w.shape[0]
    assert w.shape == (m, k)
    bb = None if b is None else _f32(b)
    y = np.empty((n, m), np.float32)
    if y.size and slice_len > 0:
        _check(lib().orc_linear_sliced_bias_relu_f32(_p(x), _p(w), None if bb is None else _p(bb), _p(y), n, k, m, int(relu),
                                                     int(slice_len)), "linear_sliced")
    elif y.size:
        _check(lib().orc_linear_bias_relu_f32(_p(x), _p(w), None if bb is None else _p(bb), _p(y), n, k, m, int(relu)), "linear")
    return y


def to_float_normalize(x: np.ndarray, mean=None, std=None) -> np.ndarray:
    """ToDtype(float32, scale=True) [+ Normalize(mean, std)] on (N, C, H, W) uint8; Normalize alone on float32."""
    n, c = int(np.prod(x.shape[:-3], dtype=np.int64)), int(x.shape[-3])
    hw = int(x.shape[-1] * x.shape[-2])
    y = np.empty(x.shape, np.float32)
    norm = mean is not None
    m = _f32(mean if norm else np.zeros(c))
    s = _f32(std if norm else np.ones(c))
    if x.dtype == np.uint8:
        x = np.ascontiguousarray(x)
        _check(lib().orc_to_float_normalize_u8(_p(x), _p(y), n, c, hw, _p(m), _p(s), int(norm)), "to_float_normalize")
    else:
        x = _f32(x)
        _check(lib().orc_normalize_f32(_p(x), _p(y), n, c, hw, _p(m), _p(s)), "normalize")
    return y


# --------------------------------------------------------------------------- tiny pure-python cross-check
def depthwise_conv2d_py(x: np.ndarray, w: np.ndarray, border: int) -> np.ndarray:
    """Loop-level restatement for very small cases (validates oracle.c itself)."""
    x = np.asarray(x, np.float32)
    ky, kx = w.shape
    ry, rx = ky // 2, kx // 2
    h, wd = x.shape[-2:]
    flat = x.reshape(-1, h, wd)

    def refl(i, n):
        i = -i if i < 0 else i
        return 2 * (n - 1) - i if i >= n else i

    if border == BORDER_VALID:
        out = np.zeros((flat.shape[0], h - ky + 1, wd - kx + 1), np.float64)
        for p in range(flat.shape[0]):
            for oy in range(out.shape[1]):
                for ox in range(out.shape[2]):
                    out[p, oy, ox] = sum(float(w[dy, dx]) * float(flat[p, oy + dy, ox + dx]) for dy in range(ky) for dx in range(kx))
        return out.reshape(x.shape[:-2] + out.shape[1:])
    out = np.zeros(flat.shape, np.float64)
    for p in range(flat.shape[0]):
        for oy in range(h):
            for ox in range(wd):
                acc = 0.0
                for dy in range(ky):
                    for dx in range(kx):
                        sy, sx = oy + dy - ry, ox + dx - rx
                        if border == BORDER_REFLECT:
                            v = flat[p, refl(sy, h), refl(sx, wd)]
                        else:
                            v = flat[p, sy, sx] if (0 <= sy < h and 0 <= sx < wd) else 0.0
                        acc += float(w[dy, dx]) * float(v)
                out[p, oy, ox] = acc
    return out.reshape(x.shape)


# ------------------------------------------------------------------------------------ preset head
def resized_output_size(h: int, w: int, size, max_size=None):
    """_compute_resized_output_size (transforms/functional.py:353-384)."""
    size = [size] if isinstance(size, int) else list(size)
    if len(size) == 2:
        return int(size[0]), int(size[1])
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size[0], int(size[0] * long_ / short)
    if max_size is not None and new_long > max_size:
        new_short, new_long = int(max_size * new_short / new_long), max_size
    return (new_long, new_short) if w <= h else (new_short, new_long)


def resize(x: np.ndarray, size, max_size=None) -> np.ndarray:
    """F.resize(img, size, BILINEAR, antialias=True) on a tensor image (..., H, W): uint8 or float32."""
    planes, h, wd = _planes(x)
    oh, ow = resized_output_size(h, wd, size, max_size)
    if x.dtype == np.uint8:
        x = np.ascontiguousarray(x)
        y = np.empty(x.shape[:-2] + (oh, ow), np.uint8)
        _check(lib().orc_resize_bilinear_aa_u8(_p(x), _p(y), planes, h, wd, oh, ow), "resize_u8")
        return y
    x = _f32(x)
    y = np.empty(x.shape[:-2] + (oh, ow), np.float32)
    _check(lib().orc_resize_bilinear_aa_f32(_p(x), _p(y), planes, h, wd, oh, ow), "resize_f32")
    return y


def center_crop(x: np.ndarray, output_size) -> np.ndarray:
    """F.center_crop (transforms/functional.py:556-594): zero padding when the crop exceeds the image."""
    if isinstance(output_size, int):
        output_size = (output_size, output_size)
    elif len(output_size) == 1:
        output_size = (output_size[0], output_size[0])
    ch, cw = int(output_size[0]), int(output_size[1])
    h, w = x.shape[-2:]
    if cw > w or ch > h:
        pl = (cw - w) // 2 if cw > w else 0
        pt = (ch - h) // 2 if ch > h else 0
        pr = (cw - w + 1) // 2 if cw > w else 0
        pb = (ch - h + 1) // 2 if ch > h else 0
        x = np.pad(x, [(0, 0)] * (x.ndim - 2) + [(pt, pb), (pl, pr)])
        h, w = x.shape[-2:]
        if cw == w and ch == h:
            return x
    top = int(round((h - ch) / 2.0))
    left = int(round((w - cw) / 2.0))
    return np.ascontiguousarray(x[..., top:top + ch, left:left + cw])


def image_classification_preset(x: np.ndarray, crop_size: int, resize_size: int = 256, mean=(0.485, 0.456, 0.406),
                                std=(0.229, 0.224, 0.225)) -> np.ndarray:
    """ImageClassification.forward on a tensor image (transforms/_presets.py:56-63): resize -> center_crop ->
    convert_image_dtype(float) (v1: uint8 `/ 255.0`, _functional_tensor.py:93-99) -> normalize (sub, div)."""
    y = center_crop(resize(x, [resize_size]), [crop_size])
    if y.dtype == np.uint8:
        y = y.astype(np.float32) / np.float32(255.0)
    m = np.asarray(mean, np.float32).reshape(-1, 1, 1)
    s = np.asarray(std, np.float32).reshape(-1, 1, 1)
    return ((y - m) / s).astype(np.float32)


# ------------------------------------------------------------------------------------ Conv2dNormActivation (8f.3)
ACT = {None: 0, "none": 0, "relu": 1, "relu6": 2, "hardswish": 3, "silu": 4}


def fold_batchnorm(weight, bias, mean, var, eps: float = 1e-5):
    """nn.BatchNorm2d(eval) as (alpha, beta) of ATen's batch_norm_cpu: y = fma(x, alpha, beta)."""
    mean, var = _f32(mean), _f32(var)
    c = mean.shape[0]
    w = None if weight is None else _f32(weight)
    b = None if bias is None else _f32(bias)
    alpha, beta = np.empty(c, np.float32), np.empty(c, np.float32)
    lib().orc_fold_batchnorm(None if w is None else _p(w), None if b is None else _p(b), _p(mean), _p(var), float(eps), c, _p(alpha), _p(beta))
    return alpha, beta


def conv2d_affine_act(x, w, bias=None, alpha=None, beta=None, res=None, stride=1, padding=0, groups=1, affine=0, act=None,
                      slice_len=0):
    """conv2d(zero padding) -> folded norm (affine 1: x*a then +b; 2: fma(x, a, b)) -> + res -> activation.
    slice_len > 0 (pointwise convs only): the sliced summation order the library states for that shape
    (mv_conv1x1_k_slices): chains over channel slices of slice_len, added in ascending order."""
    x, w = _f32(x), _f32(w)
    n, cin, h, wd = x.shape
    cout, cg, kh, kw = w.shape
    if slice_len and 0 < slice_len < cin:
        assert (kh, kw, stride, padding, groups) == (1, 1, 1, 0, 1), "sliced order: pointwise convs only"
        y = np.empty((n, cout, h, wd), np.float32)
        keep = [None if a is None else _f32(a) for a in (bias, alpha, beta, res)]
        ptr = [None if a is None else _p(a) for a in keep]
        if y.size:
            _check(lib().orc_pointwise_sliced_affine_act_f32(_p(x), _p(w), ptr[0], ptr[1], ptr[2], ptr[3], _p(y), n, cin, h * wd, cout,
                                                             int(slice_len), affine, ACT[act]), "pointwise_sliced")
        return y
    assert cg == (cin if groups == 1 else 1)
    oh, ow = (h + 2 * padding - kh) // stride + 1, (wd + 2 * padding - kw) // stride + 1
    y = np.empty((n, cout, oh, ow), np.float32)
    opt = lambda a: None if a is None else _p(_f32(a))  # noqa: E731
    keep = [None if a is None else _f32(a) for a in (bias, alpha, beta, res)]
    ptr = [None if a is None else _p(a) for a in keep]
    if y.size:
        _check(lib().orc_conv2d_affine_act_f32(_p(x), _p(w), ptr[0], ptr[1], ptr[2], ptr[3], _p(y), n, cin, h, wd, cout, kh, kw,
                                               stride, padding, groups, affine, ACT[act]), "conv2d_affine_act")
    return y


# ------------------------------------------------------------------------------------ deform_conv2d
def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def deform_conv2d(x, offset, weight, bias=None, stride=(1, 1), padding=(0, 0), dilation=(1, 1), mask=None):
    """torchvision.ops.deform_conv2d forward (ops/deform_conv.py:14-107), fp32."""
    x, offset, weight = _f32(x), _f32(offset), _f32(weight)
    n, cin, h, wd = x.shape
    cout, cg, kh, kw = weight.shape
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    groups = cin // cg
    offset_groups = offset.shape[1] // (2 * kh * kw)
    oh = (h + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    ow = (wd + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    assert offset.shape == (n, 2 * offset_groups * kh * kw, oh, ow), (offset.shape, (n, 2 * offset_groups * kh * kw, oh, ow))
    m = None if mask is None else _f32(mask)
    b = None if bias is None else _f32(bias)
    y = np.empty((n, cout, oh, ow), np.float32)
    if y.size:
        _check(lib().orc_deform_conv2d_f32(_p(x), _p(weight), _p(offset), None if m is None else _p(m), None if b is None else _p(b),
                                           _p(y), n, cin, h, wd, cout, kh, kw, sh, sw, ph, pw, dh, dw, groups, offset_groups,
                                           int(m is not None)), "deform_conv2d")
    return y


def maxpool2d(x: np.ndarray, k: int, stride: int) -> np.ndarray:
    """nn.MaxPool2d(kernel_size=k, stride=stride) over the last two dims (no padding, floor mode)."""
    x = _f32(x)
    planes, h, wd = _planes(x)
    y = np.empty(x.shape[:-2] + ((h - k) // stride + 1, (wd - k) // stride + 1), np.float32)
    if y.size:
        _check(lib().orc_maxpool2d_f32(_p(x), _p(y), planes, h, wd, k, stride), "maxpool2d")
    return y
