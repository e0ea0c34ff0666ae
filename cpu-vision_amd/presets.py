"""The reference's inference preset in front of the small CNNs (transforms/_presets.py:38-84), on the MI355X kernels.

ImageClassification.forward = resize(bilinear, antialias) -> center_crop -> convert_image_dtype(float) -> normalize.
Here the four steps are ONE call (two kernel launches): only the rows / columns the crop keeps are resized, the
rounded uint8 intermediate of the reference's resize is reproduced in registers, and the result is the fp32
normalised (…, C, crop, crop) tensor the first conv reads.  Bit-identical to oracle.ref.image_classification_preset,
which is pinned to the reference's own outputs (tests/golden/resize_preset.npz).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn

from . import _pil, functional_v1 as F1


class ImageClassification(nn.Module):
    def __init__(self, *, crop_size: int, resize_size: int = 256, mean: Tuple[float, ...] = (0.485, 0.456, 0.406),
                 std: Tuple[float, ...] = (0.229, 0.224, 0.225), interpolation="bilinear",
                 antialias: Optional[bool] = True) -> None:
        super().__init__()
        self.crop_size = [crop_size]
        self.resize_size = [resize_size]
        self.mean = list(mean)
        self.std = list(std)
        self.interpolation = getattr(interpolation, "value", interpolation)
        self.antialias = antialias

    def _forward_pil(self, img) -> torch.Tensor:
        """A PIL image (transforms/_presets.py:54-61): the reference resizes and crops it WITH PIL (F.resize / F.center_crop
        dispatch to Image.resize / Image.crop: Pillow's own fixed-point resampling, not the tensor path's), then
        pil_to_tensor -> convert_image_dtype -> normalize.  The two PIL calls stay PIL's (on the host, exactly as in the
        reference: the same library gives the same bytes); the tensor half -- uint8 -> /255 -> (v - mean) / std -- runs on the
        MI355X through the preset kernel at scale 1 (the resize weights are then exactly 1, the pixel passes through)."""
        import PIL.Image
        if self.interpolation != "bilinear":
            raise NotImplementedError(f"interpolation {self.interpolation!r}: the MI355X preset covers bilinear")
        w, h = img.size
        oh, ow = F1._compute_resized_output_size((h, w), self.resize_size)
        resized = img.resize((ow, oh), PIL.Image.BILINEAR)
        top, left, ch, cw = F1._center_crop_window(oh, ow, self.crop_size)
        cropped = resized.crop((left, top, left + cw, top + ch))  # regions outside the image are zeros = center_crop's padding
        t = _pil.pil_to_tensor(cropped).contiguous().to(_pil.device_for_host_inputs())
        c = int(t.shape[-3])
        mean, std = self._mean_std(c)
        return F1._resize_window(t, ch, cw, 0, 0, ch, cw, preset=(mean, std))

    def _mean_std(self, c: int):
        if len(self.mean) not in (1, c) or len(self.std) not in (1, c):
            raise RuntimeError(f"mean / std of {len(self.mean)} / {len(self.std)} values do not broadcast over {c} channels")
        mean = torch.tensor(self.mean if len(self.mean) == c else self.mean * c, dtype=torch.float32)
        std = torch.tensor(self.std if len(self.std) == c else self.std * c, dtype=torch.float32)
        if (std == 0).any():
            raise ValueError("std evaluated to zero after conversion to torch.float32, leading to division by zero.")
        return mean, std

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        if _pil.is_pil_image(img):
            return self._forward_pil(img)
        if not isinstance(img, torch.Tensor):
            raise TypeError(f"img should be a Tensor or a PIL Image. Got {type(img)}")
        size = F1._check_resize_args(img, self.resize_size, self.interpolation, None, self.antialias)
        if img.ndim < 3:
            raise ValueError(f"Expected tensor to be a tensor image of size (..., C, H, W). Got tensor.size() = {img.size()}")
        c = int(img.shape[-3])
        mean, std = self._mean_std(c)
        h, w = int(img.shape[-2]), int(img.shape[-1])
        oh, ow = F1._compute_resized_output_size((h, w), size)
        top, left, ch, cw = F1._center_crop_window(oh, ow, self.crop_size)
        return F1._resize_window(img, oh, ow, top, left, ch, cw, preset=(mean, std))

    def __repr__(self) -> str:
        return (f"{self.__class__.__name__}(\n    crop_size={self.crop_size}\n    resize_size={self.resize_size}"
                f"\n    mean={self.mean}\n    std={self.std}\n    interpolation={self.interpolation}\n)")

    def describe(self) -> str:
        return ("Accepts batched ``(B, C, H, W)`` and single ``(C, H, W)`` image ``torch.Tensor`` objects on an MI355X. "
                f"The images are resized to ``resize_size={self.resize_size}`` using ``interpolation={self.interpolation}``, "
                f"followed by a central crop of ``crop_size={self.crop_size}``. Finally the values are first rescaled to "
                f"``[0.0, 1.0]`` and then normalized using ``mean={self.mean}`` and ``std={self.std}``.")
