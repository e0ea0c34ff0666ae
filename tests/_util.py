"""Shared helpers for the test-suite: seeded inputs, golden loading, the parity tolerance."""
from __future__ import annotations

from functools import lru_cache
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def philox_f32(seed: int, shape) -> np.ndarray:
    """U[0,1) float32 from numpy's Philox generator (same generator as tests/golden/make_golden.py)."""
    return np.random.Generator(np.random.Philox(seed)).random(shape, dtype=np.float32)


def philox_u8(seed: int, shape) -> np.ndarray:
    return np.random.Generator(np.random.Philox(seed)).integers(0, 256, shape, dtype=np.uint8)


@lru_cache(maxsize=None)
def golden(name: str):
    return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)


def conv_tol(ref: np.ndarray, w_abs_sum: float, x_abs_max: float, rel: float = 1e-5, floor: float = 1e-6):
    """SURVEY.md 8(d) / BASELINE.md: |a-b| <= 1e-5*|b| + 1e-6 * sum|w| * max|x|.

    A pure 1e-5 relative bound is ill-posed at ReLU zeros and Sobel zero crossings, hence the
    absolute floor scaled by the filter's gain."""
    return rel * np.abs(ref.astype(np.float64)) + floor * w_abs_sum * x_abs_max


def assert_conv_close(actual, ref, w_abs_sum=1.0, x_abs_max=1.0, rel=1e-5, floor=1e-6, what=""):
    actual = np.asarray(actual)
    ref = np.asarray(ref)
    assert actual.shape == ref.shape, f"{what}: shape {actual.shape} vs {ref.shape}"
    err = np.abs(actual.astype(np.float64) - ref.astype(np.float64))
    tol = conv_tol(ref, w_abs_sum, x_abs_max, rel, floor)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {int(bad.sum())}/{err.size} elements out of tolerance; worst at {i}: "
                             f"got {actual[i]!r} want {ref[i]!r} (err {err[i]:.3e} > tol {tol[i]:.3e})")


# ----------------------------------------------------------------------------- MobileNet helpers
def randomize_norms(model, seed: int) -> None:
    """The seeded perturbation tests/golden/make_golden.py applied to the reference's norm layers, in module order."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if getattr(m, "running_var", None) is not None:
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.rand(m.bias.shape, generator=g) - 0.5)
                m.running_mean.copy_(torch.rand(m.running_mean.shape, generator=g) * 0.4 - 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 1.5 + 0.4)


def oracle_conv_block(ref, x, conv, norm, act_name, residual=None, slice_len=None):
    """One Conv2dNormActivation block through the CPU oracle, parameters taken from torch container modules.
    slice_len: None = the order the library states for a stand-alone pointwise launch of this shape (mv_conv1x1_k_slices);
    0 = the single chain; > 0 = chains over slices of that many input channels (the fused InvertedResidual's projection)."""
    import torch
    from cpu_vision_amd.mobilenet import FrozenBatchNorm2d
    alpha = beta = None
    affine = 0
    if isinstance(norm, FrozenBatchNorm2d):
        a, b = norm.folded()
        alpha, beta, affine = a.numpy(), b.numpy(), 1
    elif isinstance(norm, torch.nn.BatchNorm2d):
        alpha, beta = ref.fold_batchnorm(norm.weight.detach().numpy(), norm.bias.detach().numpy(), norm.running_mean.numpy(),
                                         norm.running_var.numpy(), norm.eps)
        affine = 2
    bias = None if conv.bias is None else conv.bias.detach().numpy()
    if slice_len is not None:
        pass
    elif conv.kernel_size == (1, 1) and conv.groups == 1:
        # pointwise convs: the summation order the library states for this shape (one chain, or K slices inside the workgroup)
        from cpu_vision_amd import functional as F
        slices, sl = F.conv1x1_k_slices(x.shape[0], x.shape[1], x.shape[2], x.shape[3], conv.out_channels)
        slice_len = sl if slices > 1 else 0
    else:
        slice_len = 0
    return ref.conv2d_affine_act(x, conv.weight.detach().numpy(), bias, alpha, beta, residual, conv.stride[0], conv.padding[0],
                                 conv.groups, affine, act_name, slice_len=slice_len)


def oracle_inverted_residual(ref, layer, x):
    """One InvertedResidual block through the oracle in the summation order the library states for it: as ONE fused kernel
    (mobilenet.FUSE_INVERTED_RESIDUAL and mv_inverted_residual_k_slices() > 0) the expansion and the depthwise conv are single
    chains and the projection runs in the stated hidden-channel slices; otherwise three stand-alone launches, each in the order
    mv_conv1x1_k_slices states."""
    from cpu_vision_amd import functional as F, mobilenet
    blocks = list(layer.conv)
    fused = None
    if mobilenet.FUSE_INVERTED_RESIDUAL and len(blocks) in (3, 4):
        slices, sl = F.inverted_residual_k_slices(x.shape[0], x.shape[1], blocks[-3][0].out_channels, blocks[-2].out_channels, x.shape[2],
                                                  x.shape[3], layer.stride)
        fused = (slices, sl) if slices else None
    a = x
    for blk in blocks[:-2]:
        a = oracle_conv_block(ref, a, blk[0], blk[1], "relu6", slice_len=0 if fused else None)
    proj_slices = None if fused is None else (fused[1] if fused[0] > 1 else 0)
    return oracle_conv_block(ref, a, blocks[-2], blocks[-1], None, x if layer.use_res_connect else None, slice_len=proj_slices)


def oracle_mobilenet_features(ref, model, x, upto=None):
    """MobileNetV2.features of a (CPU-resident) cpu_vision_amd.mobilenet.MobileNetV2 through the oracle; returns the
    list of per-layer activations."""
    from cpu_vision_amd.mobilenet import Conv2dNormActivation, InvertedResidual
    acts, a = [], x
    for i, layer in enumerate(model.features):
        if upto is not None and i > upto:
            break
        if isinstance(layer, Conv2dNormActivation):
            a = oracle_conv_block(ref, a, layer[0], layer[1], "relu6")
        else:
            assert isinstance(layer, InvertedResidual)
            inp = a
            a = oracle_inverted_residual(ref, layer, a)
        acts.append(a)
    return acts


def oracle_conv3x3(ref, x, w, b, relu=True):
    """ref.conv3x3_bias_relu in the summation order the library states for this shape (F.conv3x3_k_slices): launches too small to
    fill the chip run K in slices across workgroups."""
    from cpu_vision_amd import functional as F
    n, cin, h, wd = x.shape
    slices, sc = F.conv3x3_k_slices(n, cin, h, wd, w.shape[0])
    return ref.conv3x3_bias_relu(x, w, b, relu=relu, slice_channels=sc if slices > 1 else 0)
