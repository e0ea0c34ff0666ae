"""Kernel registry = the reference's Python plugin boundary (B1).

What the boundary fixes (torchvision/transforms/v2/functional/_utils.py:16-118) is behaviour: per functional a map
input type -> kernel; kernels registered for a TVTensor type receive the plain tensor and their result is re-wrapped like
the input; lookup prefers the most derived registered class and never lets a TVTensor subclass fall through to the
pure-tensor kernel; `allow_passthrough` hands unsupported inputs back untouched; the public `register_kernel` accepts custom
TVTensor subclasses only.  Names and error texts are the reference's, so its own registry tests read the same here.

How it is kept here: one `_Dispatch` table per functional that resolves a type ONCE (walking its MRO up to the TVTensor
barrier) and remembers the answer -- transforms call `_get_kernel` for every leaf of every sample, and after the first
call per type that is a single dict hit.  A registration drops the cached answers of that functional.
"""
from __future__ import annotations

import functools
from typing import Any, Callable, Dict, Optional, Type

import torch

from . import tv_tensors

_BUILTIN_TV_TENSOR_TYPES = frozenset({tv_tensors.Image, tv_tensors.Video, tv_tensors.Mask, tv_tensors.BoundingBoxes})


def is_pure_tensor(inpt: Any) -> bool:
    return isinstance(inpt, torch.Tensor) and not isinstance(inpt, tv_tensors.TVTensor)


def _passthrough(inpt, *args, **kwargs):
    return inpt


def _kernel_tv_tensor_wrapper(kernel):
    """kernel(plain tensor, ...) -> kernel(tv_tensor, ...) whose result carries the input's type and metadata."""

    @functools.wraps(kernel)
    def on_tv_tensor(inpt, *args, **kwargs):
        return tv_tensors.wrap(kernel(inpt.as_subclass(torch.Tensor), *args, **kwargs), like=inpt)

    return on_tv_tensor


class _Dispatch:
    """input type -> kernel for ONE functional."""

    __slots__ = ("functional", "kernels", "_resolved")

    def __init__(self, functional: Callable):
        self.functional = functional
        self.kernels: Dict[Type, Callable] = {}
        self._resolved: Dict[Type, Optional[Callable]] = {}

    def add(self, input_type: Type, kernel: Callable) -> None:
        if input_type in self.kernels:
            raise ValueError(f"Functional {self.functional} already has a kernel registered for type {input_type}.")
        self.kernels[input_type] = kernel
        self._resolved.clear()

    def resolve(self, input_type: Type) -> Optional[Callable]:
        try:
            return self._resolved[input_type]
        except KeyError:
            pass
        found = None
        for cls in input_type.__mro__:
            found = self.kernels.get(cls)
            if found is not None or cls is tv_tensors.TVTensor:
                break  # the barrier: a TVTensor subclass never reaches the kernel registered for torch.Tensor
        self._resolved[input_type] = found
        return found


_KERNEL_REGISTRY: Dict[Callable, _Dispatch] = {}


def _register_kernel_internal(functional, input_type, *, tv_tensor_wrapper=True):
    table = _KERNEL_REGISTRY.get(functional)
    if table is None:
        table = _KERNEL_REGISTRY[functional] = _Dispatch(functional)
    if input_type in table.kernels:  # fail at decoration time, like the reference
        raise ValueError(f"Functional {functional} already has a kernel registered for type {input_type}.")
    rewrap = tv_tensor_wrapper and issubclass(input_type, tv_tensors.TVTensor)

    def decorator(kernel):
        table.add(input_type, _kernel_tv_tensor_wrapper(kernel) if rewrap else kernel)
        return kernel

    return decorator


def _functional_by_name(name: str):
    from . import functional as F

    found = getattr(F, name, None)
    if found is None:
        raise ValueError(f"Could not find functional with name '{name}' in cpu_vision_amd.functional.")
    return found


def register_kernel(functional, tv_tensor_cls):
    """Public registration for CUSTOM TVTensor subclasses (reference `register_kernel`, _utils.py:69-95)."""
    if isinstance(functional, str):
        functional = _functional_by_name(functional)
    elif not callable(functional) or functional not in _KERNEL_REGISTRY:
        raise ValueError(
            f"Kernels can only be registered on functionals from the cpu_vision_amd.functional namespace, "
            f"but got {functional}.")
    is_tv_subclass = isinstance(tv_tensor_cls, type) and issubclass(tv_tensor_cls, tv_tensors.TVTensor)
    if not is_tv_subclass:
        raise ValueError(
            f"Kernels can only be registered for subclasses of tv_tensors.TVTensor, but got {tv_tensor_cls}.")
    if tv_tensor_cls in _BUILTIN_TV_TENSOR_TYPES:
        raise ValueError(f"Kernels cannot be registered for the builtin tv_tensor classes, but got {tv_tensor_cls}")
    return _register_kernel_internal(functional, tv_tensor_cls, tv_tensor_wrapper=False)


def _get_kernel(functional, input_type, *, allow_passthrough=False):
    table = _KERNEL_REGISTRY.get(functional)
    if table is None or not table.kernels:
        raise ValueError(f"No kernel registered for functional {functional.__name__}.")
    kernel = table.resolve(input_type)
    if kernel is not None:
        return kernel
    if allow_passthrough:
        return _passthrough
    raise TypeError(
        f"Functional F.{functional.__name__} supports inputs of type {table.kernels.keys()}, "
        f"but got {input_type} instead.")
