"""Conv2dNormActivation with a folded norm, InvertedResidual and MobileNetV2 on the MI355X kernels.

Mirrors ops/misc.py:13-128 (FrozenBatchNorm2d, Conv2dNormActivation) and models/mobilenetv2.py:18-172 of the reference:
same constructor arguments, same module tree (torch's own nn.Conv2d / nn.BatchNorm2d / nn.ReLU6 / nn.Linear serve as
PARAMETER CONTAINERS, built in the reference's order, so the reference's state dicts load as they are and a seeded
construction consumes torch's RNG exactly like the reference's).  Only `forward` differs: every
conv -> norm -> activation block (and the block's residual add) is ONE kernel launch through
`functional.conv_norm_act`; the containers' own forward is never called.  Inference only (norms in eval mode).
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
from torch import nn

from . import _lib
from . import functional as F


class FrozenBatchNorm2d(nn.Module):
    """ops/misc.py:13-65: BatchNorm2d with fixed statistics and affine parameters (buffers)."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        key = prefix + "num_batches_tracked"
        if key in state_dict:
            del state_dict[key]
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    def folded(self):
        """(scale, bias) exactly as forward computes them (ops/misc.py:55-60), on the host with the same tensor ops."""
        w, b = self.weight.detach().to("cpu", torch.float32), self.bias.detach().to("cpu", torch.float32)
        rv, rm = self.running_var.detach().to("cpu", torch.float32), self.running_mean.detach().to("cpu", torch.float32)
        scale = w * (rv + self.eps).rsqrt()
        bias = b - rm * scale
        return scale, bias

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise RuntimeError("FrozenBatchNorm2d is folded into the preceding convolution (Conv2dNormActivation.forward)")

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({self.weight.shape[0]}, eps={self.eps})"


# True: InvertedResidual blocks whose shape has a fused kernel (every block of MobileNetV2 at 224 x 224: csrc/invres.hip) run as ONE
# kernel -- the hidden tensor never reaches HBM; False: every block is one launch per convolution (what the tests and tools compare the fused kernel with)
FUSE_INVERTED_RESIDUAL = True

_ACTIVATIONS = {nn.ReLU: "relu", nn.ReLU6: "relu6", nn.Hardswish: "hardswish", nn.SiLU: "silu"}


def _tensor_versions(*tensors):
    return tuple((t.data_ptr(), t._version, str(t.device)) for t in tensors if t is not None)


class _FoldedNorm:
    """Device-resident (alpha, beta) of a norm layer, recomputed when its tensors change."""

    def __init__(self):
        self.key, self.value = None, None

    def get(self, norm: Optional[nn.Module], device):
        if norm is None:
            return None, None, None
        if isinstance(norm, FrozenBatchNorm2d):
            tensors, mode = (norm.weight, norm.bias, norm.running_mean, norm.running_var), "mul_add"
        elif isinstance(norm, nn.BatchNorm2d):
            if norm.training or not norm.track_running_stats:
                raise RuntimeError("the MI355X path is inference only: call .eval() (BatchNorm2d folds its running statistics)")
            tensors, mode = (norm.weight, norm.bias, norm.running_mean, norm.running_var), "fma"
        else:
            raise NotImplementedError(f"norm layer {type(norm).__name__}: BatchNorm2d (eval) and FrozenBatchNorm2d fold into the conv")
        key = _tensor_versions(*tensors) + (str(device), norm.eps)
        if key != self.key:
            if mode == "mul_add":
                a, b = norm.folded()
            else:
                a, b = F.fold_batchnorm(norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps)
            self.key, self.value = key, (a.to(device), b.to(device), mode)
        return self.value


def fused_conv_block(x: torch.Tensor, conv: nn.Conv2d, norm: Optional[nn.Module], act: Optional[nn.Module], cache: _FoldedNorm,
                     residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv -> norm -> (+ residual) -> activation as one launch; the modules only hold the parameters."""
    k = conv.kernel_size
    if conv.dilation != (1, 1) or conv.padding != ((k[0] - 1) // 2, (k[1] - 1) // 2) or conv.stride[0] != conv.stride[1] or \
            conv.padding_mode != "zeros":
        raise NotImplementedError(f"{conv}: the MI355X blocks cover dilation 1, 'same'-style zero padding and square strides")
    if act is not None and type(act) not in _ACTIVATIONS:
        raise NotImplementedError(f"activation {type(act).__name__}")
    if norm is None and residual is None and type(act) in (nn.ReLU, type(None)) and k == (3, 3) and conv.stride == (1, 1) and conv.groups == 1:
        # Conv2dNormActivation(norm_layer=None) on a 3x3 / stride-1 conv is the CNNs' first-layer pattern (vgg.py:81-85):
        # the implicit-GEMM kernels of section 3.4 / 3.5, the same (channel, ky, kx) chain + bias
        return F.conv2d_bias_relu(x, conv.weight, conv.bias, relu=act is not None)
    alpha, beta, mode = cache.get(norm, x.device)
    return F.conv_norm_act(x, conv.weight, conv.bias, alpha, beta, residual, stride=conv.stride[0], groups=conv.groups,
                           affine=mode, activation=None if act is None else _ACTIVATIONS[type(act)])


class Conv2dNormActivation(nn.Sequential):
    """ops/misc.py:68-172 with the constructor of the reference (norm_layer defaults to BatchNorm2d, activation to ReLU,
    bias defaults to `norm_layer is None`).  Children: [Conv2d, norm?, activation?] as in the reference."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1, padding: Optional[int] = None,
                 groups: int = 1, norm_layer: Optional[Callable[..., nn.Module]] = nn.BatchNorm2d,
                 activation_layer: Optional[Callable[..., nn.Module]] = nn.ReLU, dilation: int = 1, inplace: Optional[bool] = True,
                 bias: Optional[bool] = None) -> None:
        if padding is None:
            padding = (kernel_size - 1) // 2 * dilation
        if bias is None:
            bias = norm_layer is None
        layers: List[nn.Module] = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation=dilation, groups=groups,
                                             bias=bias)]
        if norm_layer is not None:
            layers.append(norm_layer(out_channels))
        if activation_layer is not None:
            params = {} if inplace is None else {"inplace": inplace}
            layers.append(activation_layer(**params))
        super().__init__(*layers)
        self.out_channels = out_channels
        self._has_norm, self._has_act = norm_layer is not None, activation_layer is not None
        self._fold = _FoldedNorm()

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        norm = self[1] if self._has_norm else None
        act = self[-1] if self._has_act else None
        return fused_conv_block(x, self[0], norm, act, self._fold, residual)


class InvertedResidual(nn.Module):
    """models/mobilenetv2.py:18-63: [1x1 expand + norm + ReLU6] -> 3x3 depthwise + norm + ReLU6 -> 1x1 project + norm
    [+ x]: ONE kernel where mv_inverted_residual_k_slices() names one (_fused_plan), else three launches (two when expand_ratio
    == 1) with the residual add in the last one's epilogue."""

    def __init__(self, inp: int, oup: int, stride: int, expand_ratio: int, norm_layer: Optional[Callable[..., nn.Module]] = None) -> None:
        super().__init__()
        self.stride = stride
        if stride not in [1, 2]:
            raise ValueError(f"stride should be 1 or 2 instead of {stride}")
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        hidden_dim = int(round(inp * expand_ratio))
        self.use_res_connect = self.stride == 1 and inp == oup
        layers: List[nn.Module] = []
        if expand_ratio != 1:
            layers.append(Conv2dNormActivation(inp, hidden_dim, kernel_size=1, norm_layer=norm_layer, activation_layer=nn.ReLU6))
        layers.extend([
            Conv2dNormActivation(hidden_dim, hidden_dim, stride=stride, groups=hidden_dim, norm_layer=norm_layer,
                                 activation_layer=nn.ReLU6),
            nn.Conv2d(hidden_dim, oup, 1, 1, 0, bias=False),
            norm_layer(oup),
        ])
        self.conv = nn.Sequential(*layers)
        self.out_channels = oup
        self._is_cn = stride > 1
        self._fold = _FoldedNorm()
        self._plans = {}

    def _fused_plan(self, x: torch.Tensor):
        """(slices, slice_len) when ONE kernel runs the whole block on this input (csrc/invres.hip: k_invres on the 28 / 14 / 7-pixel
        stages, k_invres_wide on the 112 / 56-pixel ones, the first block -- expand_ratio 1, no expansion conv -- included), else
        None -> one launch per convolution."""
        if not FUSE_INVERTED_RESIDUAL or len(self.conv) not in (3, 4) or x.ndim != 4 or x.dtype != torch.float32:
            return None
        expand = self.conv[0] if len(self.conv) == 4 else None
        dw, project, norm3 = self.conv[-3:]
        kinds = {type(dw[1]), type(norm3)} | ({type(expand[1])} if expand is not None else set())
        if len(kinds) != 1 or not (kinds <= {nn.BatchNorm2d, FrozenBatchNorm2d}) or type(dw[-1]) is not nn.ReLU6:
            return None
        if expand is not None and type(expand[-1]) is not nn.ReLU6:
            return None
        key = (tuple(x.shape), id(_lib.load()))  # the plan is a pure function of the shape and the loaded library
        plan = self._plans.get(key)
        if plan is None:
            n, cin, h, w = (int(d) for d in x.shape)
            slices, sl = F.inverted_residual_k_slices(n, cin, dw[0].out_channels, project.out_channels, h, w, self.stride)
            plan = self._plans[key] = (slices, sl)
        if plan[0] > 1 and F.BATCH_INVARIANT_SUMMATION:
            return None  # several slices = a batch-dependent order of the projection's sum: the caller asked for single chains
        return plan if plan[0] else None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._fused_plan(x) is not None:
            expand = self.conv[0] if len(self.conv) == 4 else None
            dw, project, norm3 = self.conv[-3:]
            a2, b2, mode = dw._fold.get(dw[1], x.device)
            a3, b3, _ = self._fold.get(norm3, x.device)
            if expand is None:  # hidden == cin: no expansion conv in the block (mobilenetv2.py:37-38)
                return F.inverted_residual(x, None, None, None, dw[0].weight, a2, b2, project.weight, a3, b3, self.use_res_connect,
                                           stride=self.stride, affine=mode)
            a1, b1, _ = expand._fold.get(expand[1], x.device)
            return F.inverted_residual(x, expand[0].weight, a1, b1, dw[0].weight, a2, b2, project.weight, a3, b3, self.use_res_connect,
                                       stride=self.stride, affine=mode)
        y = x
        for block in list(self.conv)[:-2]:
            y = block(y)
        return fused_conv_block(y, self.conv[-2], self.conv[-1], None, self._fold, residual=x if self.use_res_connect else None)


def _make_divisible(v: float, divisor: int, min_value: Optional[int] = None) -> int:
    """models/_utils.py:76-90."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class MobileNetV2(nn.Module):
    """models/mobilenetv2.py:66-172."""

    def __init__(self, num_classes: int = 1000, width_mult: float = 1.0, inverted_residual_setting: Optional[List[List[int]]] = None,
                 round_nearest: int = 8, block: Optional[Callable[..., nn.Module]] = None,
                 norm_layer: Optional[Callable[..., nn.Module]] = None, dropout: float = 0.2) -> None:
        super().__init__()
        if block is None:
            block = InvertedResidual
        if norm_layer is None:
            norm_layer = nn.BatchNorm2d
        input_channel, last_channel = 32, 1280
        if inverted_residual_setting is None:
            inverted_residual_setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2],
                                         [6, 320, 1, 1]]
        if len(inverted_residual_setting) == 0 or len(inverted_residual_setting[0]) != 4:
            raise ValueError(f"inverted_residual_setting should be non-empty or a 4-element list, got {inverted_residual_setting}")
        input_channel = _make_divisible(input_channel * width_mult, round_nearest)
        self.last_channel = _make_divisible(last_channel * max(1.0, width_mult), round_nearest)
        features: List[nn.Module] = [Conv2dNormActivation(3, input_channel, stride=2, norm_layer=norm_layer, activation_layer=nn.ReLU6)]
        for t, c, n, s in inverted_residual_setting:
            output_channel = _make_divisible(c * width_mult, round_nearest)
            for i in range(n):
                stride = s if i == 0 else 1
                features.append(block(input_channel, output_channel, stride, expand_ratio=t, norm_layer=norm_layer))
                input_channel = output_channel
        features.append(Conv2dNormActivation(input_channel, self.last_channel, kernel_size=1, norm_layer=norm_layer,
                                             activation_layer=nn.ReLU6))
        self.features = nn.Sequential(*features)
        self.classifier = nn.Sequential(nn.Dropout(p=dropout), nn.Linear(self.last_channel, num_classes))
        for m in self.modules():  # weight initialization, mobilenetv2.py:143-154
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.zeros_(m.bias)
        self.eval()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise RuntimeError("the MI355X MobileNetV2 is inference only: call .eval()")
        inp = x
        with torch.no_grad():  # no autograd bookkeeping per layer; the result is marked once (its backward raises)
            x = self.features(x)
            x = F.adaptive_avg_pool2d(x, (1, 1))  # nn.functional.adaptive_avg_pool2d(x, (1, 1)), mobilenetv2.py:160
            x = torch.flatten(x, 1)
            fc = self.classifier[1]  # Dropout is the identity in eval mode
            x = F.linear_bias_relu(x, fc.weight, fc.bias, relu=False)
        from ._lib import forward_only
        return forward_only(x, "MobileNetV2.forward", inp, fc.weight)


def mobilenet_v2(num_classes: int = 1000, **kwargs) -> MobileNetV2:
    return MobileNetV2(num_classes=num_classes, **kwargs)
