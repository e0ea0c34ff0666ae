"""Transform classes that call the hot path -- mirrors of the reference's callers:

  Transform.forward / _call_kernel        transforms/v2/_transform.py:17-87
  GaussianBlur (v2)                       transforms/v2/_misc.py:168-205
  RandomAdjustSharpness (v2)              transforms/v2/_color.py:356-376 (+ _RandomApplyTransform, _transform.py:~150-190)
  GaussianBlurV1                          transforms/transforms.py:1753-1812
  ElasticTransform._get_params            transforms/v2/_geometry.py:1054-1075 (noise field -> gaussian_blur -> displacement)
  Resize / CenterCrop / ToDtype / Normalize / Compose   v2/_geometry.py:76-193, v2/_misc.py:134-304, v2/_container.py:10-64

Parameter sampling stays on the host RNG exactly like the reference (torch.empty(1).uniform_ / torch.rand(1)).
"""
from __future__ import annotations

import numbers
from typing import Any, Callable, Dict, List, Sequence, Union

import torch
from torch import nn
from torch.utils._pytree import tree_flatten, tree_unflatten

from . import _pil
from . import functional as F
from . import functional_v1 as F1
from . import tv_tensors
from ._registry import _get_kernel, is_pure_tensor

_PIL_TYPES = (_pil.PIL.Image.Image,) if _pil.PIL is not None else ()


import enum as _enum


class InterpolationMode(_enum.Enum):
    """transforms/functional.py:21-35."""

    NEAREST = "nearest"
    NEAREST_EXACT = "nearest-exact"
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"
    BOX = "box"
    HAMMING = "hamming"
    LANCZOS = "lanczos"


def _setup_size(size, error_msg):
    """transforms/transforms.py `_setup_size`"""
    if isinstance(size, numbers.Number):
        return int(size), int(size)
    if isinstance(size, Sequence) and len(size) == 1:
        return size[0], size[0]
    if len(size) != 2:
        raise ValueError(error_msg)
    return size


def _setup_number_or_seq(arg, name: str) -> Sequence[float]:
    """transforms/v2/_utils.py:21-38"""
    if not isinstance(arg, (int, float, Sequence)):
        raise TypeError(f"{name} should be a number or a sequence of numbers. Got {type(arg)}")
    if isinstance(arg, Sequence) and len(arg) not in (1, 2):
        raise ValueError(f"If {name} is a sequence its length should be 1 or 2. Got {len(arg)}")
    if isinstance(arg, Sequence):
        for element in arg:
            if not isinstance(element, (int, float)):
                raise ValueError(f"{name} should be a sequence of numbers. Got {type(element)}")
    if isinstance(arg, (int, float)):
        arg = [float(arg), float(arg)]
    elif isinstance(arg, Sequence):
        arg = [float(arg[0]), float(arg[0])] if len(arg) == 1 else [float(arg[0]), float(arg[1])]
    return arg


def query_size(flat_inputs: List[Any]):
    """transforms/v2/_utils.py:175-196: the one (H, W) of the images / videos / masks / boxes of a sample."""
    sizes = set()
    for inpt in flat_inputs:
        if isinstance(inpt, tv_tensors.BoundingBoxes) and hasattr(inpt, "canvas_size"):
            sizes.add(tuple(inpt.canvas_size))
        elif isinstance(inpt, torch.Tensor):
            if inpt.ndim < 2:
                raise TypeError(f"Input tensor should have at least two dimensions, but got {inpt.ndim}")
            sizes.add(tuple(inpt.shape[-2:]))
        elif isinstance(inpt, _PIL_TYPES):
            sizes.add((inpt.size[1], inpt.size[0]))
    if not sizes:
        raise TypeError("No image, video, mask or bounding box was found in the sample")
    if len(sizes) > 1:
        raise ValueError(f"Found multiple HxW dimensions in the sample: {sorted(sizes)}")
    h, w = sizes.pop()
    return int(h), int(w)


class Transform(nn.Module):
    _transformed_types = (torch.Tensor,) + _PIL_TYPES

    def _check_inputs(self, flat_inputs: List[Any]) -> None:
        pass

    def _get_params(self, flat_inputs: List[Any]) -> Dict[str, Any]:
        return dict()

    def _call_kernel(self, functional: Callable, inpt: Any, *args: Any, **kwargs: Any) -> Any:
        kernel = _get_kernel(functional, type(inpt), allow_passthrough=True)
        return kernel(inpt, *args, **kwargs)

    def _transform(self, inpt: Any, params: Dict[str, Any]) -> Any:
        raise NotImplementedError

    def forward(self, *inputs: Any) -> Any:
        flat_inputs, spec = tree_flatten(inputs if len(inputs) > 1 else inputs[0])
        self._check_inputs(flat_inputs)
        needs = self._needs_transform_list(flat_inputs)
        params = self._get_params([i for i, n in zip(flat_inputs, needs) if n])
        flat_outputs = self._transform_all(flat_inputs, needs, params)
        return tree_unflatten(flat_outputs, spec)

    def _transform_many(self, inpts: List[Any], params: Dict[str, Any]):
        """Hook: transform several plain image tensors of one sample with the same params in one call (None = no batched
        form; GaussianBlur / RandomAdjustSharpness route a list of frames through one mv_*_v launch)."""
        return None

    def _transform_all(self, flat_inputs: List[Any], needs: List[bool], params: Dict[str, Any]) -> List[Any]:
        idx = [k for k, (i, n) in enumerate(zip(flat_inputs, needs))
               if n and type(i) in (torch.Tensor, tv_tensors.Image) and i.is_cuda]
        outs = {}
        if len(idx) > 1:
            many = self._transform_many([flat_inputs[k].as_subclass(torch.Tensor) for k in idx], params)
            if many is not None:
                for k, o in zip(idx, many):
                    like = flat_inputs[k]
                    outs[k] = tv_tensors.wrap(o, like=like) if isinstance(like, tv_tensors.TVTensor) else o
        return [outs[k] if k in outs else (self._transform(i, params) if n else i)
                for k, (i, n) in enumerate(zip(flat_inputs, needs))]

    def extra_repr(self) -> str:
        """v2/_transform.py:89-99: the public, plainly typed attributes."""
        import enum
        extra = []
        for name, value in self.__dict__.items():
            if name.startswith("_") or name == "training":
                continue
            if not isinstance(value, (bool, int, float, str, tuple, list, enum.Enum)):
                continue
            extra.append(f"{name}={value}")
        return ", ".join(extra)

    def _needs_transform_list(self, flat_inputs: List[Any]) -> List[bool]:
        # the reference's pure-tensor heuristic (_transform.py:57-87): with an explicit Image/Video in the
        # sample pure tensors pass through; otherwise only the first pure tensor is treated as the image
        has_explicit = any(isinstance(i, (tv_tensors.Image, tv_tensors.Video) + _PIL_TYPES) for i in flat_inputs)
        transform_pure_tensor = not has_explicit
        out = []
        for inpt in flat_inputs:
            needs = isinstance(inpt, self._transformed_types)
            if needs and is_pure_tensor(inpt):
                if transform_pure_tensor:
                    transform_pure_tensor = False
                else:
                    needs = False
            out.append(needs)
        return out


class _RandomApplyTransform(Transform):
    def __init__(self, p: float = 0.5) -> None:
        if not (0.0 <= p <= 1.0):
            raise ValueError("`p` should be a floating point value in the interval [0.0, 1.0].")
        super().__init__()
        self.p = p

    def forward(self, *inputs: Any) -> Any:
        inputs = inputs if len(inputs) > 1 else inputs[0]
        flat_inputs, spec = tree_flatten(inputs)
        self._check_inputs(flat_inputs)
        if torch.rand(1) >= self.p:
            return inputs
        needs = self._needs_transform_list(flat_inputs)
        params = self._get_params([i for i, n in zip(flat_inputs, needs) if n])
        flat_outputs = self._transform_all(flat_inputs, needs, params)
        return tree_unflatten(flat_outputs, spec)


class GaussianBlur(Transform):
    """v2.GaussianBlur(kernel_size, sigma=(0.1, 2.0)) -- transforms/v2/_misc.py:168-205."""

    def __init__(self, kernel_size: Union[int, Sequence[int]], sigma: Union[int, float, Sequence[float]] = (0.1, 2.0)) -> None:
        super().__init__()
        self.kernel_size = _setup_size(kernel_size, "Kernel size should be a tuple/list of two integers")
        for ks in self.kernel_size:
            if ks <= 0 or ks % 2 == 0:
                raise ValueError("Kernel size value should be an odd and positive number.")
        self.sigma = _setup_number_or_seq(sigma, "sigma")
        if not 0.0 < self.sigma[0] <= self.sigma[1]:
            raise ValueError(f"sigma values should be positive and of the form (min, max). Got {self.sigma}")

    def _get_params(self, flat_inputs: List[Any]) -> Dict[str, Any]:
        sigma = torch.empty(1).uniform_(self.sigma[0], self.sigma[1]).item()
        return dict(sigma=[sigma, sigma])

    def _transform(self, inpt: Any, params: Dict[str, Any]) -> Any:
        return self._call_kernel(F.gaussian_blur, inpt, self.kernel_size, **params)

    def _transform_many(self, inpts: List[Any], params: Dict[str, Any]):
        return F.gaussian_blur_frames(inpts, list(self.kernel_size), **params)


class RandomAdjustSharpness(_RandomApplyTransform):
    """v2.RandomAdjustSharpness(sharpness_factor, p=0.5) -- transforms/v2/_color.py:356-376."""

    def __init__(self, sharpness_factor: float, p: float = 0.5) -> None:
        super().__init__(p=p)
        self.sharpness_factor = sharpness_factor

    def _transform(self, inpt: Any, params: Dict[str, Any]) -> Any:
        return self._call_kernel(F.adjust_sharpness, inpt, sharpness_factor=self.sharpness_factor)

    def _transform_many(self, inpts: List[Any], params: Dict[str, Any]):
        if any(i.shape[-3] not in (1, 3) for i in inpts):
            return None  # the per-image kernel raises the reference's TypeError
        return F.adjust_sharpness_frames(inpts, self.sharpness_factor)


# ---- the classification preset's steps as v2 transforms (row f2): Resize -> CenterCrop -> ToDtype(float32, scale=True) ->
#      Normalize, composable with the blur / sharpness transforms above ---------------------------------------------------
class Resize(Transform):
    """v2.Resize(size, interpolation=BILINEAR, max_size=None, antialias=True) -- transforms/v2/_geometry.py:76-168.
    Bilinear with antialias (what every classification preset uses) runs on the MI355X resize kernels."""

    def __init__(self, size, interpolation=InterpolationMode.BILINEAR, max_size=None, antialias=True) -> None:
        super().__init__()
        if isinstance(size, int):
            size = [size]
        elif isinstance(size, Sequence) and len(size) in {1, 2}:
            size = list(size)
        elif size is None:
            if not isinstance(max_size, int):
                raise ValueError(f"max_size must be an integer when size is None, but got {max_size} instead.")
        else:
            raise ValueError(f"size can be an integer, a sequence of one or two integers, or None, but got {size} instead.")
        self.size = size
        self.interpolation = interpolation
        self.max_size = max_size
        self.antialias = antialias

    def _transform(self, inpt: Any, params: Dict[str, Any]) -> Any:
        return self._call_kernel(F.resize, inpt, self.size, interpolation=self.interpolation, max_size=self.max_size,
                                 antialias=self.antialias)


class CenterCrop(Transform):
    """v2.CenterCrop(size) -- transforms/v2/_geometry.py:171-193."""

    def __init__(self, size) -> None:
        super().__init__()
        self.size = _setup_size(size, "Please provide only two dimensions (h, w) for size.")

    def _transform(self, inpt: Any, params: Dict[str, Any]) -> Any:
        return self._call_kernel(F.center_crop, inpt, output_size=list(self.size))


class ToDtype(Transform):
    """v2.ToDtype(dtype, scale=False) -- transforms/v2/_misc.py:208-304 (a torch.dtype or a {type: dtype, "others": ...} dict)."""

    def __init__(self, dtype, scale: bool = False) -> None:
        super().__init__()
        if not isinstance(dtype, (dict, torch.dtype)):
            raise ValueError(f"dtype must be a dict or a torch.dtype, got {type(dtype)} instead")
        self.dtype = dtype
        self.scale = scale

    def _transform(self, inpt: Any, params: Dict[str, Any]) -> Any:
        if isinstance(self.dtype, torch.dtype):
            if not is_pure_tensor(inpt) and not isinstance(inpt, (tv_tensors.Image, tv_tensors.Video)):
                return inpt
            dtype = self.dtype
        elif type(inpt) in self.dtype:
            dtype = self.dtype[type(inpt)]
        elif "others" in self.dtype:
            dtype = self.dtype["others"]
        else:
            raise ValueError(f"No dtype was specified for type {type(inpt)}. "
                             "If you only need to convert the dtype of images or videos, you can just pass e.g. dtype=torch.float32. "
                             "If you're passing a dict as dtype, "
                             'you can use "others" as a catch-all key '
                             'e.g. dtype={tv_tensors.Mask: torch.int64, "others": None} to pass-through the rest of the inputs.')
        if dtype is None:
            return inpt
        return self._call_kernel(F.to_dtype, inpt, dtype=dtype, scale=self.scale)


class Normalize(Transform):
    """v2.Normalize(mean, std, inplace=False) -- transforms/v2/_misc.py:134-165."""

    def __init__(self, mean: Sequence[float], std: Sequence[float], inplace: bool = False) -> None:
        super().__init__()
        self.mean = list(mean)
        self.std = list(std)
        self.inplace = inplace

    def _check_inputs(self, flat_inputs: List[Any]) -> None:
        if any(isinstance(i, _PIL_TYPES) for i in flat_inputs):
            raise TypeError(f"{type(self).__name__}() does not support PIL images.")

    def _transform(self, inpt: Any, params: Dict[str, Any]) -> Any:
        return self._call_kernel(F.normalize, inpt, mean=self.mean, std=self.std, inplace=self.inplace)


class Compose(Transform):
    """v2.Compose(transforms) -- transforms/v2/_container.py:10-64."""

    def __init__(self, transforms: Sequence[Callable]) -> None:
        super().__init__()
        if not isinstance(transforms, Sequence):
            raise TypeError("Argument transforms should be a sequence of callables")
        elif not transforms:
            raise ValueError("Pass at least one transform")
        self.transforms = transforms

    def forward(self, *inputs: Any) -> Any:
        needs_unpacking = len(inputs) > 1
        for transform in self.transforms:
            outputs = transform(*inputs)
            inputs = outputs if needs_unpacking else (outputs,)
        return outputs

    def extra_repr(self) -> str:
        return "\n".join(f"    {t}" for t in self.transforms)


class ElasticTransform(Transform):
    """v2.ElasticTransform(alpha=50.0, sigma=5.0, ...) -- transforms/v2/_geometry.py:1003-1085.

    `_get_params` is the caller of the hot path mirrored here (row a9): a U(-1, 1) noise field per axis, blurred with a
    Gaussian of `k = int(8 * sigma + 1) | 1` taps per side, scaled by `alpha / size`.  The noise is drawn from the HOST
    generator with the reference's calls in the reference's order (`torch.rand([1, 1, H, W]) * 2 - 1`, dx then dy), so a
    seeded pipeline samples the same displacement field; the blur -- the expensive part: 41 x 41 taps at the default
    sigma = 5 -- runs on the MI355X and the field stays there (`displacement.device` is the HIP device).

    Applying the field (`F.elastic` = grid_sample, _geometry.py:1077-1085) is a geometric resampling outside this
    library's path: `_transform` says so; pass `params["displacement"]` to the resampler you use."""

    def __init__(self, alpha: Union[float, Sequence[float]] = 50.0, sigma: Union[float, Sequence[float]] = 5.0,
                 interpolation="bilinear", fill=0) -> None:
        super().__init__()
        self.alpha = _setup_number_or_seq(alpha, "alpha")
        self.sigma = _setup_number_or_seq(sigma, "sigma")
        self.interpolation = interpolation
        self.fill = fill

    def _get_params(self, flat_inputs: List[Any]) -> Dict[str, Any]:
        size = list(query_size(flat_inputs))
        device = _pil.device_for_host_inputs()

        dx = torch.rand([1, 1] + size) * 2 - 1
        if self.sigma[0] > 0.0:
            kx = int(8 * self.sigma[0] + 1)
            # if kernel size is even we have to make it odd
            if kx % 2 == 0:
                kx += 1
            dx = self._call_kernel(F.gaussian_blur, dx.to(device), [kx, kx], list(self.sigma))
        dx = dx.to(device) * self.alpha[0] / size[0]

        dy = torch.rand([1, 1] + size) * 2 - 1
        if self.sigma[1] > 0.0:
            ky = int(8 * self.sigma[1] + 1)
            if ky % 2 == 0:
                ky += 1
            dy = self._call_kernel(F.gaussian_blur, dy.to(device), [ky, ky], list(self.sigma))
        dy = dy.to(device) * self.alpha[1] / size[1]
        displacement = torch.concat([dx, dy], 1).permute([0, 2, 3, 1])  # 1 x H x W x 2
        return dict(displacement=displacement)

    def get_params(self, *inputs: Any) -> Dict[str, Any]:
        """Sample the displacement field for a sample (public form of `_get_params`)."""
        flat_inputs, _ = tree_flatten(inputs if len(inputs) > 1 else inputs[0])
        return self._get_params(flat_inputs)

    def _transform(self, inpt: Any, params: Dict[str, Any]) -> Any:
        raise NotImplementedError(
            "ElasticTransform: applying the displacement field (F.elastic = grid_sample) is a geometric resampling outside "
            "this library's filtering path; use get_params(sample)['displacement'] (sampled and blurred on the MI355X) with "
            "your resampler.")


class GaussianBlurV1(nn.Module):
    """transforms.GaussianBlur (v1) -- transforms/transforms.py:1753-1812."""

    def __init__(self, kernel_size, sigma=(0.1, 2.0)):
        super().__init__()
        self.kernel_size = _setup_size(kernel_size, "Kernel size should be a tuple/list of two integers")
        for ks in self.kernel_size:
            if ks <= 0 or ks % 2 == 0:
                raise ValueError("Kernel size value should be an odd and positive number.")
        if isinstance(sigma, numbers.Number):
            if sigma <= 0:
                raise ValueError("If sigma is a single number, it must be positive.")
            sigma = (sigma, sigma)
        elif isinstance(sigma, Sequence) and len(sigma) == 2:
            if not 0.0 < sigma[0] <= sigma[1]:
                raise ValueError("sigma values should be positive and of the form (min, max).")
        else:
            raise ValueError("sigma should be a single number or a list/tuple with length 2.")
        self.sigma = sigma

    @staticmethod
    def get_params(sigma_min: float, sigma_max: float) -> float:
        return torch.empty(1).uniform_(sigma_min, sigma_max).item()

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        sigma = self.get_params(self.sigma[0], self.sigma[1])
        return F1.gaussian_blur(img, list(self.kernel_size), [sigma, sigma])

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}(kernel_size={self.kernel_size}, sigma={self.sigma})"
