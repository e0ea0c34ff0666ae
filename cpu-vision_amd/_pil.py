"""PIL <-> tensor conversion around the kernels.  The reference's PIL entry of the blur is
`pil_to_tensor -> gaussian_blur_image -> to_pil_image(mode=image.mode)` (transforms/v2/functional/_misc.py:169-174);
here the tensor additionally travels to the MI355X and back.

What this path needs is narrow: a PIL image becomes a CHW tensor of the mode's storage type, the kernels return a tensor of
the same type and channel count, and it goes back into the mode it came from.  The conversion is therefore a lookup in
one table -- (bands, storage dtype) -> the modes PIL can build from such an array -- not a general image encoder; the
user-facing rules it keeps from the reference's `to_pil_image` (transforms/functional.py:246-324) are its argument errors
(wrong type, rank, more than four bands, a mode that does not fit the band count) and that float data is scaled by 255 to
uint8 unless mode "F" is asked for.
"""
from __future__ import annotations

import sys

import numpy as np
import torch

try:
    import PIL.Image
except ImportError:  # pragma: no cover -- PIL is optional, exactly as in the reference
    PIL = None


def is_pil_image(obj) -> bool:
    return PIL is not None and isinstance(obj, PIL.Image.Image)


def pil_to_tensor(pic) -> torch.Tensor:
    """CHW tensor sharing nothing with `pic` (transforms/functional.py:181-213: array copy, dtype of the mode)."""
    if not is_pil_image(pic):
        raise TypeError(f"pic should be PIL Image. Got {type(pic)}")
    width, height = pic.size
    hwc = torch.from_numpy(np.array(pic, copy=True)).reshape(height, width, len(pic.getbands()))
    return hwc.movedim(-1, 0)


# bands -> modes accepted for that many bands; the first one is what uint8 data becomes when no mode is given
_MULTIBAND_MODES = {2: ("LA",), 3: ("RGB", "YCbCr", "HSV"), 4: ("RGBA", "CMYK", "RGBX")}
# single band: the one mode each storage type maps to
_SINGLE_BAND_MODE = {np.dtype(np.uint8): "L", np.dtype(np.int16): "I;16" if sys.byteorder == "little" else "I;16B",
                     np.dtype(np.int32): "I", np.dtype(np.float32): "F"}


def _as_hwc(pic) -> np.ndarray:
    if isinstance(pic, torch.Tensor):
        arr = pic.detach().cpu().numpy()
        if arr.ndim == 3:
            arr = np.moveaxis(arr, 0, -1)  # CHW -> HWC
    elif isinstance(pic, np.ndarray):
        arr = pic
    else:
        raise TypeError(f"pic should be Tensor or ndarray. Got {type(pic)}.")
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if arr.ndim != 3:
        raise ValueError(f"pic should be 2/3 dimensional. Got {arr.ndim} dimensions.")
    if arr.shape[2] > 4:
        raise ValueError(f"pic should not have > 4 channels. Got {arr.shape[2]} channels.")
    return arr


def to_pil_image(pic, mode=None):
    """Tensor (CHW or HW) / ndarray (HWC or HW) -> PIL image of `mode` (None: the natural mode of the data)."""
    arr = _as_hwc(pic)
    if arr.dtype.kind == "f" and mode != "F":
        arr = (arr * 255).astype(np.uint8)
    bands = arr.shape[2]
    if bands == 1:
        natural = _SINGLE_BAND_MODE.get(arr.dtype)
        if mode not in (None, natural):
            raise ValueError(f"Incorrect mode ({mode}) supplied for input type {arr.dtype}. Should be {natural}")
        mode, arr = natural, arr[:, :, 0]
    else:
        allowed = _MULTIBAND_MODES[bands]
        if mode is None:
            mode = allowed[0] if arr.dtype == np.uint8 else None
        elif mode not in allowed:
            raise ValueError(f"Only modes {list(allowed)} are supported for {bands}D inputs")
    if mode is None:
        raise TypeError(f"Input type {arr.dtype} is not supported")
    return PIL.Image.fromarray(np.ascontiguousarray(arr), mode=mode)


def device_for_host_inputs() -> torch.device:
    """Where a host-resident input (a PIL image) is computed: the current HIP device.  There is no CPU path."""
    from . import _lib
    if not torch.cuda.is_available():
        raise _lib.Mi355VisionError("a PIL image was passed but no HIP device is visible: the kernels run on the MI355X only "
                                    "and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())
