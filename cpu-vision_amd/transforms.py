"""Transform classes that call the hot path -- mirrors of the reference's callers:

  Transform.forward / _call_kernel        transforms/v2/_transform.py:17-87
  GaussianBlur (v2)                       transforms/v2/_misc.py:168-205
  RandomAdjustSharpness (v2)              transforms/v2/_color.py:356-376 (+ _RandomApplyTransform, _transform.py:~150-190)
  GaussianBlurV1                          transforms/transforms.py:1753-1812

Parameter sampling stays on the host RNG exactly like the reference (torch.empty(1).uniform_ / torch.rand(1)).
"""
from __future__ import annotations

import numbers
from typing import Any, Callable, Dict, List, Sequence, Union

import torch
from torch import nn
from torch.utils._pytree import tree_flatten, tree_unflatten

from . import functional as F
from . import functional_v1 as F1
from . import tv_tensors
from ._registry import _get_kernel, is_pure_tensor


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


class Transform(nn.Module):
    _transformed_types = (torch.Tensor,)

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
        flat_outputs = [self._transform(i, params) if n else i for i, n in zip(flat_inputs, needs)]
        return tree_unflatten(flat_outputs, spec)

    def _needs_transform_list(self, flat_inputs: List[Any]) -> List[bool]:
        # the reference's pure-tensor heuristic (_transform.py:57-87): with an explicit Image/Video in the
        # sample pure tensors pass through; otherwise only the first pure tensor is treated as the image
        has_explicit = any(isinstance(i, (tv_tensors.Image, tv_tensors.Video)) for i in flat_inputs)
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
        flat_outputs = [self._transform(i, params) if n else i for i, n in zip(flat_inputs, needs)]
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


class RandomAdjustSharpness(_RandomApplyTransform):
    """v2.RandomAdjustSharpness(sharpness_factor, p=0.5) -- transforms/v2/_color.py:356-376."""

    def __init__(self, sharpness_factor: float, p: float = 0.5) -> None:
        super().__init__(p=p)
        self.sharpness_factor = sharpness_factor

    def _transform(self, inpt: Any, params: Dict[str, Any]) -> Any:
        return self._call_kernel(F.adjust_sharpness, inpt, sharpness_factor=self.sharpness_factor)


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
