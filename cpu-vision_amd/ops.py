"""torchvision.ops.deform_conv2d / DeformConv2d on the MI355X kernels.

Mirrors ops/deform_conv.py:14-190 (function, module, argument handling and messages) and the shape checks of the
reference's kernel (csrc/ops/cpu/deform_conv2d_kernel.cpp:907-1005, same messages); the arithmetic runs in
mv_deform_conv2d_f32 (deformable im2col + fp32 MFMA GEMM).  Forward only.

`register_torchvision_op()` additionally plugs the kernel into the reference's own operator registry (boundary B2):
it defines the `torchvision::deform_conv2d` schema if no extension has (csrc/ops/deform_conv2d.cpp:164-169) and
registers this implementation for the CUDA (= HIP) dispatch key, so the reference's unmodified
`torchvision.ops.deform_conv2d` / `DeformConv2d` run on MI355X tensors.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import nn
from torch.nn import init
from torch.nn.modules.utils import _pair
from torch.nn.parameter import Parameter

from . import _lib

# columns workspace per pass: the batch is split so that one pass stays below this (the reference uses <= 32 images)
MAX_WORKSPACE_BYTES = 1 << 30


# True: never allocate the optional workspace of a small launch (tests run the fused kernel on small shapes this way)
FUSED_WHENEVER_POSSIBLE = False


def _deform_conv2d_impl(input, weight, offset, mask, bias, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, n_weight_grps, n_offset_grps,
                        use_mask) -> torch.Tensor:
    """Same signature as the registered operator (csrc/ops/deform_conv2d.cpp:164-169)."""
    if input.ndim != 4 or offset.ndim != 4 or weight.ndim != 4 or (use_mask and mask.ndim != 4):
        raise RuntimeError("deform_conv2d expects 4-D input, offset, weight (and mask)")
    _lib.require_device(input)
    for t, name in ((weight, "weight"), (offset, "offset")):
        _lib.require_device(t, name)
    if not input.is_floating_point():
        raise RuntimeError(f"deform_conv2d expects a floating-point input. Got {input.dtype}")
    # float64 / float16 / bfloat16 tensors (the reference's TestDeformConv runs in float64, test/test_ops.py:931) are computed
    # in float32 and returned in the input's dtype: against the reference's own float64 oracle (expected_fn) that is 2e-6
    # absolute on its test configuration, inside the rtol = atol = 1e-5 the reference holds its kernels to
    out_dtype = input.dtype
    input = input.detach().to(torch.float32)
    n, cin, h, w = (int(d) for d in input.shape)
    cout, cg, kh, kw = (int(d) for d in weight.shape)
    ker_h, ker_w = dil_h * (kh - 1) + 1, dil_w * (kw - 1) + 1
    if stride_h <= 0 or stride_w <= 0:
        raise RuntimeError(f"stride_h: {stride_h} stride_w: {stride_w}")
    if pad_h < 0 or pad_w < 0:
        raise RuntimeError(f"pad_h: {pad_h} pad_w: {pad_w}")
    if dil_h <= 0 or dil_w <= 0:
        raise RuntimeError(f"dilation_h: {dil_h} dilation_w: {dil_w}")
    out_h, out_w = (h + 2 * pad_h - ker_h) // stride_h + 1, (w + 2 * pad_w - ker_w) // stride_w + 1
    if cg * n_weight_grps != cin or cout % n_weight_grps != 0 or cin % n_offset_grps != 0:
        raise RuntimeError(f"channels ({cin} -> {cout}) do not divide into {n_weight_grps} weight groups / {n_offset_grps} offset groups")
    if offset.shape[1] != n_offset_grps * 2 * kh * kw:
        raise RuntimeError(f"offset.shape[1] is not valid: got: {offset.shape[1]} expected: {n_offset_grps * 2 * kh * kw}")
    if use_mask and mask.shape[1] != n_offset_grps * kh * kw:
        raise RuntimeError(f"mask.shape[1] is not valid: got: {mask.shape[1]} expected: {n_offset_grps * kh * kw}")
    if offset.shape[0] != n:
        raise RuntimeError("invalid batch size of offset")
    if tuple(offset.shape[2:]) != (out_h, out_w):
        raise RuntimeError(f"offset output dims: ({offset.shape[2]}, {offset.shape[3]}) - computed output dims: ({out_h}, {out_w})")
    if use_mask and mask.shape[0] != n:
        raise RuntimeError("invalid batch size of mask")
    if use_mask and tuple(mask.shape[2:]) != (out_h, out_w):
        raise RuntimeError(f"mask output dims: ({mask.shape[2]}, {mask.shape[3]}) - computed output dims: ({out_h}, {out_w})")
    if out_h <= 0 or out_w <= 0:
        raise RuntimeError(f"Calculated output size too small - out_h: {out_h} out_w: {out_w}")
    lib = _lib.load()
    with _lib.on_device_of(input):
        f32 = lambda t: t.detach().to(input.device, torch.float32).contiguous()  # noqa: E731
        y = torch.empty((n, cout, out_h, out_w), dtype=torch.float32, device=input.device)
        if n == 0:
            return y.to(out_dtype)
        xc, wc, oc = input.contiguous(), f32(weight), f32(offset)
        mc = f32(mask) if use_mask else None
        bc = None if bias is None else f32(bias)
        ws = None  # the fused kernel keeps the deformable columns in LDS; exotic geometries and small launches use the columns workspace
        need = lib.mv_deform_conv2d_needs_workspace(n, cin, cout, h, w, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, n_weight_grps,
                                                    n_offset_grps)  # 0 unused, 1 required, 2 optional (faster for a small launch)
        if need == 1 or (need == 2 and not FUSED_WHENEVER_POSSIBLE):
            per_image = int(lib.mv_deform_conv2d_workspace_bytes(1, cin, h, w, kh, kw, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w))
            images = max(1, min(n, MAX_WORKSPACE_BYTES // max(per_image, 1)))
            ws = torch.empty(images * per_image, dtype=torch.uint8, device=input.device)
        _lib.check(lib.mv_deform_conv2d_f32(xc.data_ptr(), wc.data_ptr(), oc.data_ptr(), None if mc is None else mc.data_ptr(),
                                            None if bc is None else bc.data_ptr(), y.data_ptr(), n, cin, h, w, cout, kh, kw, stride_h,
                                            stride_w, pad_h, pad_w, dil_h, dil_w, n_weight_grps, n_offset_grps, int(use_mask),
                                            None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(),
                                            _lib.stream_ptr(xc)))
    return y if out_dtype == torch.float32 else y.to(out_dtype)


def deform_conv2d(input: torch.Tensor, offset: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None,
                  stride: Tuple[int, int] = (1, 1), padding: Tuple[int, int] = (0, 0), dilation: Tuple[int, int] = (1, 1),
                  mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ops/deform_conv.py:14-107: Deformable Convolution v2 when `mask` is given, v1 otherwise."""
    use_mask = mask is not None
    stride_h, stride_w = _pair(stride)
    pad_h, pad_w = _pair(padding)
    dil_h, dil_w = _pair(dilation)
    weights_h, weights_w = weight.shape[-2:]
    _, n_in_channels, _, _ = input.shape
    n_offset_grps = offset.shape[1] // (2 * weights_h * weights_w)
    n_weight_grps = n_in_channels // weight.shape[1]
    if n_offset_grps == 0:
        raise RuntimeError(
            "the shape of the offset tensor at dimension 1 is not valid. It should "
            "be a multiple of 2 * weight.size[2] * weight.size[3].\n"
            f"Got offset.shape[1]={offset.shape[1]}, while 2 * weight.size[2] * weight.size[3]={2 * weights_h * weights_w}")
    out = _deform_conv2d_impl(input, weight, offset, mask, bias, stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, n_weight_grps,
                              n_offset_grps, use_mask)
    return _lib.forward_only(out, "deform_conv2d", input, offset, weight, bias, mask)


class DeformConv2d(nn.Module):
    """ops/deform_conv.py:110-190 (same parameters, initialisation and repr)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0, dilation: int = 1,
                 groups: int = 1, bias: bool = True):
        super().__init__()
        if in_channels % groups != 0:
            raise ValueError("in_channels must be divisible by groups")
        if out_channels % groups != 0:
            raise ValueError("out_channels must be divisible by groups")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        self.weight = Parameter(torch.empty(out_channels, in_channels // groups, self.kernel_size[0], self.kernel_size[1]))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = init._calculate_fan_in_and_fan_out(self.weight)
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def forward(self, input: torch.Tensor, offset: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        return deform_conv2d(input, offset, self.weight, self.bias, stride=self.stride, padding=self.padding, dilation=self.dilation,
                             mask=mask)

    def __repr__(self) -> str:
        s = f"{self.__class__.__name__}({self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}"
        s += f", padding={self.padding}" if self.padding != (0, 0) else ""
        s += f", dilation={self.dilation}" if self.dilation != (1, 1) else ""
        s += f", groups={self.groups}" if self.groups != 1 else ""
        s += ", bias=False" if self.bias is None else ""
        return s + ")"


_SCHEMA = ("deform_conv2d(Tensor input, Tensor weight, Tensor offset, Tensor mask, Tensor bias, SymInt stride_h, SymInt stride_w, "
           "SymInt pad_h, SymInt pad_w, SymInt dilation_h, SymInt dilation_w, SymInt groups, SymInt offset_groups, bool use_mask) -> Tensor")
_registered = []


NATIVE_SHIM_PATH = _lib.LIB_PATH.parent / "libmi355vision_torch.so"


def register_torchvision_op(native: bool = False) -> None:
    """Boundary B2: make `torch.ops.torchvision.deform_conv2d` dispatch to the MI355X kernel for device tensors.  The
    reference's ops/deform_conv.py then works unchanged (it passes zero-sized placeholder mask / bias tensors when they
    are absent: ops/deform_conv.py:70-74).

    native=False: the kernels are attached from Python (`torch.library`, CUDA and AutogradCUDA keys).
    native=True:  the C++ shim `libmi355vision_torch.so` (cpu-vision_amd/torch_shim/, built by torch_shim/build.py) is loaded
                  with torch.ops.load_library, exactly as the reference loads its own extension (extension.py:33-35): it
                  registers the CUDA, Meta (fake kernel: torch.compile / FakeTensor trace through the op) and Autocast keys
                  with TORCH_LIBRARY_IMPL over the C ABI; no Python runs per call.  One process takes one of the two forms."""
    if _registered:
        if native != (_registered[0] == "native"):
            raise RuntimeError("torchvision::deform_conv2d is already registered in the other form in this process")
        return
    lib = torch.library.Library("torchvision", "FRAGMENT")
    try:
        lib.define(_SCHEMA)
    except RuntimeError:
        pass  # an installed torchvision extension already defined the schema
    if native:
        if not NATIVE_SHIM_PATH.exists():
            raise _lib.Mi355VisionError(f"{NATIVE_SHIM_PATH} not found: build it with `python cpu-vision_amd/torch_shim/build.py`")
        _lib.load()  # libmi355vision.so first: the shim links against it
        torch.ops.load_library(str(NATIVE_SHIM_PATH))
        _registered.extend(["native", lib])
        return

    def impl(input, weight, offset, mask, bias, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, groups, offset_groups, use_mask):
        return _deform_conv2d_impl(input, weight, offset, mask if use_mask else None, bias, int(stride_h), int(stride_w), int(pad_h),
                                   int(pad_w), int(dilation_h), int(dilation_w), int(groups), int(offset_groups), bool(use_mask))
    lib.impl("deform_conv2d", impl, "CUDA")

    def autograd_impl(input, weight, offset, mask, bias, *rest):
        # the reference registers a backward for this operator (csrc/ops/autograd/deform_conv2d_kernel.cpp); this library is
        # forward-only, so a recorded call returns a result whose backward raises instead of silently dropping gradients
        with torch._C._AutoDispatchBelowAutograd():
            out = torch.ops.torchvision.deform_conv2d(input, weight, offset, mask, bias, *rest)
        return _lib.forward_only(out, "torchvision::deform_conv2d", input, weight, offset, mask, bias)
    try:
        lib.impl("deform_conv2d", autograd_impl, "AutogradCUDA")
    except RuntimeError:
        pass  # an installed extension already owns the autograd key
    _registered.extend(["python", lib])
