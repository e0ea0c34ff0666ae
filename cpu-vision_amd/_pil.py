"""PIL <-> tensor conversion around the kernels: the reference's PIL entry of the blur is
`pil_to_tensor -> gaussian_blur_image -> to_pil_image(mode=image.mode)` (transforms/v2/functional/_misc.py:169-174);
here the tensor additionally travels to the MI355X and back.

  pil_to_tensor   transforms/functional.py:181-213  (np.array copy, HWC -> CHW, dtype of the PIL mode)
  to_pil_image    transforms/functional.py:246-324  (mode inference / validation, same messages)
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
    if not is_pil_image(pic):
        raise TypeError(f"pic should be PIL Image. Got {type(pic)}")
    img = torch.as_tensor(np.array(pic, copy=True))
    img = img.view(pic.size[1], pic.size[0], len(pic.getbands()))
    return img.permute((2, 0, 1))  # HWC -> CHW


def to_pil_image(pic, mode=None):
    if isinstance(pic, torch.Tensor):
        if pic.ndim == 3:
            pic = pic.permute((1, 2, 0))
        pic = pic.numpy(force=True)
    elif not isinstance(pic, np.ndarray):
        raise TypeError(f"pic should be Tensor or ndarray. Got {type(pic)}.")
    if pic.ndim == 2:
        pic = np.expand_dims(pic, 2)
    if pic.ndim != 3:
        raise ValueError(f"pic should be 2/3 dimensional. Got {pic.ndim} dimensions.")
    if pic.shape[-1] > 4:
        raise ValueError(f"pic should not have > 4 channels. Got {pic.shape[-1]} channels.")
    npimg = pic
    if np.issubdtype(npimg.dtype, np.floating) and mode != "F":
        npimg = (npimg * 255).astype(np.uint8)
    if npimg.shape[2] == 1:
        expected_mode = None
        npimg = npimg[:, :, 0]
        if npimg.dtype == np.uint8:
            expected_mode = "L"
        elif npimg.dtype == np.int16:
            expected_mode = "I;16" if sys.byteorder == "little" else "I;16B"
        elif npimg.dtype == np.int32:
            expected_mode = "I"
        elif npimg.dtype == np.float32:
            expected_mode = "F"
        if mode is not None and mode != expected_mode:
            raise ValueError(f"Incorrect mode ({mode}) supplied for input type {np.dtype}. Should be {expected_mode}")
        mode = expected_mode
    elif npimg.shape[2] == 2:
        if mode is not None and mode not in ["LA"]:
            raise ValueError("Only modes ['LA'] are supported for 2D inputs")
        if mode is None and npimg.dtype == np.uint8:
            mode = "LA"
    elif npimg.shape[2] == 4:
        if mode is not None and mode not in ["RGBA", "CMYK", "RGBX"]:
            raise ValueError("Only modes ['RGBA', 'CMYK', 'RGBX'] are supported for 4D inputs")
        if mode is None and npimg.dtype == np.uint8:
            mode = "RGBA"
    else:
        if mode is not None and mode not in ["RGB", "YCbCr", "HSV"]:
            raise ValueError("Only modes ['RGB', 'YCbCr', 'HSV'] are supported for 3D inputs")
        if mode is None and npimg.dtype == np.uint8:
            mode = "RGB"
    if mode is None:
        raise TypeError(f"Input type {npimg.dtype} is not supported")
    return PIL.Image.fromarray(np.ascontiguousarray(npimg), mode=mode)


def device_for_host_inputs() -> torch.device:
    """Where a host-resident input (a PIL image) is computed: the current HIP device.  There is no CPU path."""
    from . import _lib
    if not torch.cuda.is_available():
        raise _lib.Mi355VisionError("a PIL image was passed but no HIP device is visible: the kernels run on the MI355X only "
                                    "and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())
