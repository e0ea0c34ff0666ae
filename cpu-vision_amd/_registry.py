"""Kernel registry = the reference's Python plugin boundary (B1).

Mirrors torchvision/transforms/v2/functional/_utils.py:16-118: `{functional: {input_type: kernel}}`,
registration wraps TVTensor kernels with unwrap -> call -> wrap(like=), lookup walks the MRO and stops
at TVTensor, `allow_passthrough` hands unsupported types back untouched.
"""
from __future__ import annotations

import functools
from typing import Any, Callable, Dict, Type

import torch

from . import tv_tensors

_KERNEL_REGISTRY: Dict[Callable, Dict[Type, Callable]] = {}

_BUILTIN_TV_TENSOR_TYPES = {tv_tensors.Image, tv_tensors.Video, tv_tensors.Mask, tv_tensors.BoundingBoxes}


def is_pure_tensor(inpt: Any) -> bool:
    return isinstance(inpt, torch.Tensor) and not isinstance(inpt, tv_tensors.TVTensor)


def _kernel_tv_tensor_wrapper(kernel):
    @functools.wraps(kernel)
    def wrapper(inpt, *args, **kwargs):
        output = kernel(inpt.as_subclass(torch.Tensor), *args, **kwargs)
        return tv_tensors.wrap(output, like=inpt)

    return wrapper


def _register_kernel_internal(functional, input_type, *, tv_tensor_wrapper=True):
    registry = _KERNEL_REGISTRY.setdefault(functional, {})
    if input_type in registry:
        raise ValueError(f"Functional {functional} already has a kernel registered for type {input_type}.")

    def decorator(kernel):
        wrap_it = issubclass(input_type, tv_tensors.TVTensor) and tv_tensor_wrapper
        registry[input_type] = _kernel_tv_tensor_wrapper(kernel) if wrap_it else kernel
        return kernel

    return decorator


def register_kernel(functional, tv_tensor_cls):
    """Public registration for CUSTOM TVTensor subclasses (reference `register_kernel`, _utils.py:69-95)."""
    if isinstance(functional, str):
        from . import functional as F
        try:
            functional = getattr(F, functional)
        except AttributeError:
            raise ValueError(f"Could not find functional with name '{functional}' in cpu_vision_amd.functional.") from None
    elif not (callable(functional) and functional in _KERNEL_REGISTRY):
        raise ValueError(
            f"Kernels can only be registered on functionals from the cpu_vision_amd.functional namespace, "
            f"but got {functional}.")
    if not (isinstance(tv_tensor_cls, type) and issubclass(tv_tensor_cls, tv_tensors.TVTensor)):
        raise ValueError(
            f"Kernels can only be registered for subclasses of tv_tensors.TVTensor, but got {tv_tensor_cls}.")
    if tv_tensor_cls in _BUILTIN_TV_TENSOR_TYPES:
        raise ValueError(f"Kernels cannot be registered for the builtin tv_tensor classes, but got {tv_tensor_cls}")
    return _register_kernel_internal(functional, tv_tensor_cls, tv_tensor_wrapper=False)


def _get_kernel(functional, input_type, *, allow_passthrough=False):
    registry = _KERNEL_REGISTRY.get(functional)
    if not registry:
        raise ValueError(f"No kernel registered for functional {functional.__name__}.")
    for cls in input_type.__mro__:
        if cls in registry:
            return registry[cls]
        if cls is tv_tensors.TVTensor:
            break  # user-defined tv_tensors never fall through to the pure-tensor kernel
    if allow_passthrough:
        return lambda inpt, *args, **kwargs: inpt
    raise TypeError(
        f"Functional F.{functional.__name__} supports inputs of type {registry.keys()}, "
        f"but got {input_type} instead.")
