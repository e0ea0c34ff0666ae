"""Type tags the kernel registry is keyed on -- a minimal mirror of the reference's tv_tensors
(torchvision/tv_tensors/_tv_tensor.py:15-95, tv_tensors/__init__.py:14-35).

Only what the hot path's dispatch needs: tensor subclasses that survive `as_subclass` round trips and a
`wrap(like=)` helper.  Image / Video carry pixels (filtered); Mask / BoundingBoxes are passed through by
the transforms, exactly as in the reference (transforms/v2/_transform.py:33-35).
"""
from __future__ import annotations

import torch


class TVTensor(torch.Tensor):
    @staticmethod
    def _to_tensor(data, dtype=None, device=None) -> torch.Tensor:
        return torch.as_tensor(data, dtype=dtype, device=device)

    def __new__(cls, data, *, dtype=None, device=None):
        return cls._to_tensor(data, dtype=dtype, device=device).as_subclass(cls)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        # like the reference: results of torch ops are plain tensors; only wrap() re-tags
        with torch._C.DisableTorchFunctionSubclass():
            out = func(*args, **(kwargs or {}))
        return out


class Image(TVTensor):
    def __new__(cls, data, *, dtype=None, device=None):
        t = cls._to_tensor(data, dtype=dtype, device=device)
        if t.ndim < 2:
            raise ValueError
        if t.ndim == 2:
            t = t.unsqueeze(0)
        return t.as_subclass(cls)


class Video(TVTensor):
    def __new__(cls, data, *, dtype=None, device=None):
        t = cls._to_tensor(data, dtype=dtype, device=device)
        if t.ndim < 4:
            raise ValueError
        return t.as_subclass(cls)


class Mask(TVTensor):
    pass


class BoundingBoxes(TVTensor):
    pass


def wrap(wrappee: torch.Tensor, *, like: TVTensor) -> TVTensor:
    """tv_tensors.wrap (tv_tensors/__init__.py:14-35): re-tag a plain result with the type of `like`."""
    return wrappee.as_subclass(type(like))
