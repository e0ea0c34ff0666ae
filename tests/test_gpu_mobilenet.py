"""SURVEY.md 8(f).3 -- Conv2dNormActivation with a folded norm (BatchNorm2d eval / FrozenBatchNorm2d) + ReLU6 /
Hardswish / ReLU / SiLU, InvertedResidual and MobileNetV2 on the MI355X, against the reference's own outputs (golden
fixtures) and the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cpu_vision_amd import functional as F  # noqa: E402
from cpu_vision_amd.mobilenet import Conv2dNormActivation, FrozenBatchNorm2d, InvertedResidual, MobileNetV2  # noqa: E402
from oracle import ref  # noqa: E402
from tests._util import (assert_conv_close, golden, oracle_conv_block, oracle_inverted_residual, oracle_mobilenet_features,  # noqa: E402
                         philox_f32, philox_u8, randomize_norms)
from tests.test_oracle_golden import _mb_block  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def test_blocks_vs_reference_fixtures_and_oracle():
    g = golden("mobilenet_v2")
    for name in map(str, g["index"]):
        x, conv, norm, act = _mb_block(g, name)
        want_ref, want_orc = g[f"{name}__y"], oracle_conv_block(ref, x, conv, norm, act)
        if isinstance(norm, FrozenBatchNorm2d):
            a, b = norm.folded()
            mode = "mul_add"
        else:
            a, b = F.fold_batchnorm(norm.weight, norm.bias, norm.running_mean, norm.running_var, norm.eps)
            oa, ob = ref.fold_batchnorm(norm.weight.detach().numpy(), norm.bias.detach().numpy(), norm.running_mean.numpy(),
                                        norm.running_var.numpy(), norm.eps)
            np.testing.assert_array_equal(a.numpy(), oa)
            np.testing.assert_array_equal(b.numpy(), ob)
            mode = "fma"
        got = host(F.conv_norm_act(dev(x), conv.weight.detach().cuda(), None, a, b, None, stride=conv.stride[0], groups=conv.groups,
                                   affine=mode, activation=act))
        if act == "silu":
            np.testing.assert_allclose(got, want_orc, rtol=2e-6, atol=1e-7, err_msg=name)
        else:
            np.testing.assert_array_equal(got, want_orc, err_msg=f"{name} vs oracle")
        gain = float(np.abs(conv.weight.detach().numpy()).reshape(conv.out_channels, -1).sum(1).max()) * 3.0
        assert_conv_close(got, want_ref, gain, float(np.abs(x).max()), what=f"{name} vs reference")


def _rand_affine(seed, c):
    a = philox_f32(seed, (c,)) + 0.5
    b = philox_f32(seed + 1, (c,)) - 0.5
    return a, b


@pytest.mark.parametrize("cin,cout,h,w,stride", [(3, 32, 224, 224, 2), (3, 16, 33, 35, 2), (1, 8, 17, 9, 1), (4, 33, 8, 8, 2),
                                                 (2, 7, 1, 1, 1), (3, 64, 5, 300, 1), (3, 9, 2, 2, 2)])
def test_stem_conv_bit_exact_vs_oracle(cin, cout, h, w, stride):
    x = philox_f32(9000 + h, (2, cin, h, w)) * 2 - 1
    wt = (philox_f32(9001 + cout, (cout, cin, 3, 3)) - 0.5) * 0.6
    a, b = _rand_affine(9002, cout)
    bias = philox_f32(9004, (cout,)) - 0.5
    for affine, code in (("fma", 2), ("mul_add", 1), (None, 0)):
        for act in ("relu6", "hardswish", None):
            got = host(F.conv_norm_act(dev(x), dev(wt), dev(bias) if affine is None else None, None if affine is None else dev(a),
                                       None if affine is None else dev(b), None, stride=stride, affine=affine, activation=act))
            want = ref.conv2d_affine_act(x, wt, bias if affine is None else None, None if affine is None else a,
                                         None if affine is None else b, None, stride, 1, 1, code, act)
            np.testing.assert_array_equal(got, want, err_msg=f"{affine} {act}")


@pytest.mark.parametrize("c,h,w,stride", [(32, 112, 112, 1), (96, 112, 112, 2), (144, 56, 56, 1), (192, 28, 28, 2), (384, 14, 14, 1),
                                          (576, 14, 14, 2), (960, 7, 7, 1), (5, 1, 1, 1), (3, 2, 9, 2), (7, 13, 1, 2), (16, 15, 17, 2),
                                          # k_dwpc3x3_small (planes through LDS): every width and stride, plane counts that do not
                                          # fill the last workgroup, a non-square map, a single row
                                          (192, 28, 28, 1), (384, 14, 14, 2), (960, 7, 7, 2), (7, 7, 7, 1), (5, 20, 14, 1), (3, 1, 28, 2),
                                          (9, 3, 7, 2), (11, 28, 14, 2)])
def test_depthwise_per_channel_bit_exact_vs_oracle(c, h, w, stride):
    n = 2 if c * h * w < 500_000 else 1
    x = philox_f32(9100 + c, (n, c, h, w)) * 6 - 3
    wt = (philox_f32(9101 + h, (c, 1, 3, 3)) - 0.5)
    a, b = _rand_affine(9102 + c, c)
    got = host(F.conv_norm_act(dev(x), dev(wt), None, dev(a), dev(b), None, stride=stride, groups=c, affine="fma", activation="relu6"))
    np.testing.assert_array_equal(got, ref.conv2d_affine_act(x, wt, None, a, b, None, stride, 1, c, 2, "relu6"))
    got = host(F.conv_norm_act(dev(x), dev(wt), None, dev(a), dev(b), None, stride=stride, groups=c, affine="mul_add", activation="relu"))
    np.testing.assert_array_equal(got, ref.conv2d_affine_act(x, wt, None, a, b, None, stride, 1, c, 1, "relu"))
    if w in (7, 14, 28) and h <= 28:
        from cpu_vision_amd import _lib
        assert _lib.last_kernel().startswith("k_dwpc3x3_small")
        if stride == 1:  # + residual and Hardswish, and a batch large enough for the full-size workgroups
            xb = philox_f32(9150 + c, (40, c, h, w)) * 6 - 3
            res = philox_f32(9151 + c, (40, c, h, w)) - 0.5
            got = host(F.conv_norm_act(dev(xb), dev(wt), None, dev(a), dev(b), dev(res), groups=c, affine="fma", activation="hardswish"))
            np.testing.assert_array_equal(got, ref.conv2d_affine_act(xb, wt, None, a, b, res, 1, 1, c, 2, "hardswish"))


@pytest.mark.parametrize("cin,cout,h,w", [(32, 16, 112, 112), (16, 96, 56, 56), (144, 24, 56, 56), (192, 64, 14, 14), (384, 96, 14, 14),
                                          (960, 160, 7, 7), (320, 1280, 7, 7), (5, 3, 1, 1), (33, 130, 3, 5), (7, 200, 9, 9),
                                          (130, 7, 6, 6), (64, 64, 2, 50)])
@pytest.mark.parametrize("n", [2, 64])
def test_pointwise_mfma_bit_exact_vs_oracle(cin, cout, h, w, n):
    """Both summation orders of the pointwise kernel: one ascending-channel chain (large launches), and K slices inside the
    workgroup (launches of few workgroups with long K); the library states which one a shape takes (mv_conv1x1_k_slices)
    and the oracle restates it.  The two orders agree to 1e-6 relative of sum |w x|."""
    if n == 64 and h * w > 200:
        pytest.skip("batch 64 only for the small maps (where the K slices are used)")
    x = philox_f32(9200 + cin, (n, cin, h, w)) * 2 - 1
    wt = (philox_f32(9201 + cout, (cout, cin, 1, 1)) - 0.5) * (2.0 / cin) ** 0.5 * 2
    a, b = _rand_affine(9202 + cout, cout)
    res = philox_f32(9204, (n, cout, h, w)) - 0.5
    slices, slice_len = F.conv1x1_k_slices(n, cin, h, w, cout)
    assert slices >= 1 and (slices == 1 or (slice_len % 32 == 0 and (slices - 1) * slice_len < cin <= slices * slice_len))
    sl = slice_len if slices > 1 else 0
    got = host(F.conv_norm_act(dev(x), dev(wt), None, dev(a), dev(b), None, affine="fma", activation="relu6"))
    from cpu_vision_amd import _lib
    assert ("ks" in _lib.last_kernel()) == (slices > 1), (_lib.last_kernel(), slices)
    np.testing.assert_array_equal(got, ref.conv2d_affine_act(x, wt, None, a, b, None, 1, 0, 1, 2, "relu6", slice_len=sl))
    got = host(F.conv_norm_act(dev(x), dev(wt), None, dev(a), dev(b), dev(res), affine="fma", activation=None))
    np.testing.assert_array_equal(got, ref.conv2d_affine_act(x, wt, None, a, b, res, 1, 0, 1, 2, None, slice_len=sl), err_msg="linear bottleneck + residual")
    bias = philox_f32(9205, (cout,)) - 0.5
    got = host(F.conv_norm_act(dev(x), dev(wt), dev(bias), None, None, None, activation="relu"))
    np.testing.assert_array_equal(got, ref.conv2d_affine_act(x, wt, bias, None, None, None, 1, 0, 1, 0, "relu", slice_len=sl), err_msg="bias + relu")
    if slices > 1:  # against the single chain: another association of the same sum
        one = ref.conv2d_affine_act(x, wt, bias, None, None, None, 1, 0, 1, 0, None)
        two = ref.conv2d_affine_act(x, wt, bias, None, None, None, 1, 0, 1, 0, None, slice_len=sl)
        mag = np.einsum("nchw,mc->nmhw", np.abs(x), np.abs(wt[:, :, 0, 0])) + np.abs(bias)[None, :, None, None]
        assert np.all(np.abs(one - two) <= 1e-6 * mag + 1e-30)


def test_modules_mirror_the_reference_tree_and_fold_lazily():
    blk = Conv2dNormActivation(8, 8, stride=2, groups=8, norm_layer=FrozenBatchNorm2d, activation_layer=torch.nn.Hardswish).cuda()
    assert [type(m).__name__ for m in blk] == ["Conv2d", "FrozenBatchNorm2d", "Hardswish"] and blk[0].bias is None
    assert set(blk.state_dict()) == {"0.weight", "1.weight", "1.bias", "1.running_mean", "1.running_var"}
    randomize_norms(blk, 3)
    x = philox_f32(9300, (1, 8, 9, 11))
    y1 = host(blk(dev(x)))
    cpu = Conv2dNormActivation(8, 8, stride=2, groups=8, norm_layer=FrozenBatchNorm2d, activation_layer=torch.nn.Hardswish)
    cpu.load_state_dict(blk.state_dict())
    np.testing.assert_array_equal(y1, oracle_conv_block(ref, x, cpu[0], cpu[1], "hardswish"))
    with torch.no_grad():
        blk[1].running_mean.add_(0.25)  # the cached fold must notice
    cpu.load_state_dict(blk.state_dict())
    np.testing.assert_array_equal(host(blk(dev(x))), oracle_conv_block(ref, x, cpu[0], cpu[1], "hardswish"))
    ir = InvertedResidual(16, 16, 1, 6).cuda().eval()
    assert ir.use_res_connect and [type(m).__name__ for m in ir.conv] == ["Conv2dNormActivation", "Conv2dNormActivation", "Conv2d", "BatchNorm2d"]
    bn = Conv2dNormActivation(3, 8, stride=2).cuda()  # training-mode BatchNorm2d cannot fold
    with pytest.raises(RuntimeError, match="inference only"):
        bn(dev(philox_f32(1, (1, 3, 8, 8))))
    with pytest.raises(NotImplementedError):
        F.conv_norm_act(dev(philox_f32(1, (1, 8, 8, 8))), dev(philox_f32(2, (8, 8, 3, 3))))  # dense 3x3 with cin > 4


# every InvertedResidual of MobileNetV2 (width 1.0, 224 x 224 input) that has a fused kernel: (cin, cout, map side, stride)
_FUSED_BLOCKS = [(16, 24, 112, 2), (24, 24, 56, 1), (24, 32, 56, 2), (32, 32, 28, 1), (32, 64, 28, 2), (64, 64, 14, 1), (64, 96, 14, 1), (96, 96, 14, 1), (96, 160, 14, 2), (160, 160, 7, 1),
                 (160, 320, 7, 1)]


@pytest.mark.parametrize("norm", ["bn", "frozen"])
@pytest.mark.parametrize("cin,cout,side,stride", _FUSED_BLOCKS)
def test_fused_inverted_residual_bit_exact_vs_oracle(cin, cout, side, stride, norm):
    """ONE kernel per block (csrc/invres.hip): expansion and depthwise conv are the stand-alone kernels'
    chains, the projection sums the hidden channels in the slices mv_inverted_residual_k_slices states -- the oracle restates
    exactly that (tests/_util.oracle_inverted_residual), so the block is bit-exact for every batch size: 1 and 3 (one slice per
    chunk; an odd batch leaves a 7 x 7 region with one image), 9, and 64 where the oracle finishes quickly."""
    from cpu_vision_amd import _lib, mobilenet
    layer_kw = {} if norm == "bn" else {"norm_layer": FrozenBatchNorm2d}
    torch.manual_seed(cin * 7 + cout)
    cpu = InvertedResidual(cin, cout, stride, 6, **layer_kw).eval()
    randomize_norms(cpu, cin + cout + side)
    gpu = InvertedResidual(cin, cout, stride, 6, **layer_kw).eval()
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.cuda()
    batches = (1, 3, 9) + ((64,) if (side, cin) in ((14, 64), (7, 160)) and norm == "bn" and cout != 96 else ())
    for n in batches:
        x = philox_f32(9600 + n + side, (n, cin, side, side)) * 2 - 1
        slices, sl = F.inverted_residual_k_slices(n, cin, 6 * cin, cout, side, side, stride)
        assert slices >= 1 and (slices - 1) * sl < 6 * cin <= slices * sl
        got = host(gpu(dev(x)))
        if side >= 56:  # csrc/invres.hip k_invres_wide: regions of a few output rows, hidden / cout padded to whole tiles
            assert _lib.last_kernel().startswith(f"k_invres_wide<{side},s{stride},") and f"cin{cin}" in _lib.last_kernel(), _lib.last_kernel()
            assert f"slices{slices}>" in _lib.last_kernel(), _lib.last_kernel()
        else:
            assert _lib.last_kernel().startswith(f"k_invres<{side},s{stride},cin{cin},cout{cout},slices{slices}>"), _lib.last_kernel()
        want = oracle_inverted_residual(ref, cpu, x)
        np.testing.assert_array_equal(got, want, err_msg=f"batch {n}: fused block vs oracle in the stated order ({slices} slices of {sl})")
        # the three stand-alone launches: the same block in another association of the projection's sum
        old, mobilenet.FUSE_INVERTED_RESIDUAL = mobilenet.FUSE_INVERTED_RESIDUAL, False
        try:
            three = host(gpu(dev(x)))
            assert not _lib.last_kernel().startswith("k_invres")
            np.testing.assert_array_equal(three, oracle_inverted_residual(ref, cpu, x))
        finally:
            mobilenet.FUSE_INVERTED_RESIDUAL = old
        assert np.abs(got - three).max() <= 2e-6 * max(1.0, float(np.abs(three).max()))


@pytest.mark.parametrize("norm", ["bn", "frozen"])
def test_fused_block_without_expansion_bit_exact_vs_oracle(norm):
    """MobileNetV2's first block (expand_ratio 1: depthwise -> project, mobilenetv2.py:37-38) through the same fused kernel with the
    input rows copied into the hidden tile; single chains throughout, so it equals the two stand-alone launches bit for bit."""
    from cpu_vision_amd import _lib, mobilenet
    layer_kw = {} if norm == "bn" else {"norm_layer": FrozenBatchNorm2d}
    torch.manual_seed(77)
    cpu = InvertedResidual(32, 16, 1, 1, **layer_kw).eval()
    randomize_norms(cpu, 78)
    gpu = InvertedResidual(32, 16, 1, 1, **layer_kw).eval()
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.cuda()
    for n in (1, 3, 5):
        x = philox_f32(9700 + n, (n, 32, 112, 112)) * 2 - 1
        assert F.inverted_residual_k_slices(n, 32, 32, 16, 112, 112, 1) == (1, 32)
        got = host(gpu(dev(x)))
        assert _lib.last_kernel().startswith("k_invres_wide<112,s1,") and "no expansion" in _lib.last_kernel(), _lib.last_kernel()
        np.testing.assert_array_equal(got, oracle_inverted_residual(ref, cpu, x))
        old, mobilenet.FUSE_INVERTED_RESIDUAL = mobilenet.FUSE_INVERTED_RESIDUAL, False
        try:
            np.testing.assert_array_equal(host(gpu(dev(x))), got)
            assert not _lib.last_kernel().startswith("k_invres")
        finally:
            mobilenet.FUSE_INVERTED_RESIDUAL = old


@pytest.mark.parametrize("cin,cout,side,stride,t", [(24, 20, 56, 1, 3), (24, 32, 56, 2, 5), (16, 8, 112, 2, 4), (16, 24, 112, 2, 9), (24, 24, 56, 1, 2)])
def test_fused_wide_block_ragged_channels_bit_exact_vs_oracle(cin, cout, side, stride, t):
    """k_invres_wide pads the hidden channels and cout to whole 32-wide tiles with zero weights (exact no-ops of the chains): hidden
    sizes that end inside a tile (72, 120, 48), a cout below 32 that is not MobileNetV2's, one and many chunks -- bit-exact."""
    from cpu_vision_amd import _lib
    torch.manual_seed(cin + cout + t)
    cpu = InvertedResidual(cin, cout, stride, t).eval()
    randomize_norms(cpu, cin * t + cout)
    gpu = InvertedResidual(cin, cout, stride, t).eval()
    gpu.load_state_dict(cpu.state_dict())
    gpu = gpu.cuda()
    hidden = cpu.conv[0][0].out_channels
    for n in (1, 4, 11):
        x = philox_f32(9800 + n + t, (n, cin, side, side)) * 2 - 1
        slices, sl = F.inverted_residual_k_slices(n, cin, hidden, cout, side, side, stride)
        assert slices >= 1 and (slices - 1) * sl < hidden <= slices * sl
        got = host(gpu(dev(x)))
        assert _lib.last_kernel().startswith(f"k_invres_wide<{side},s{stride},"), _lib.last_kernel()
        np.testing.assert_array_equal(got, oracle_inverted_residual(ref, cpu, x), err_msg=f"batch {n}: {slices} slices of {sl}")


def test_fused_inverted_residual_abi_checks():
    """Shapes without a fused kernel say so (the Python layer then runs three launches); several slices need the workspace."""
    from cpu_vision_amd import _lib
    lib = _lib.load()
    assert F.inverted_residual_k_slices(8, 32, 192, 32, 56, 56, 1) == (0, 192)      # 56-pixel maps: only MobileNetV2's own 24 -> 144 blocks
    assert F.inverted_residual_k_slices(64, 24, 144, 24, 56, 56, 1) == (1, 144)
    assert F.inverted_residual_k_slices(8, 64, 384, 64, 14, 12, 1)[0] == 0          # not square
    assert F.inverted_residual_k_slices(8, 48, 288, 48, 14, 14, 1)[0] == 0          # a width multiplier the kernel is not built for
    assert F.inverted_residual_k_slices(256, 64, 384, 64, 14, 14, 1) == (1, 384)    # one region per CU: single chain, no workspace
    assert lib.mv_inverted_residual_workspace_bytes(256, 64, 384, 64, 14, 14, 1) == 0
    assert lib.mv_inverted_residual_workspace_bytes(2, 64, 384, 64, 14, 14, 1) == 12 * 2 * 64 * 196 * 4
    x = dev(philox_f32(1, (2, 64, 14, 14)))
    w1, wd, w2 = dev(philox_f32(2, (384, 64, 1, 1))), dev(philox_f32(3, (384, 1, 3, 3))), dev(philox_f32(4, (64, 384, 1, 1)))
    t = dev(philox_f32(5, (384,)))
    y = torch.empty((2, 64, 14, 14), device="cuda")
    rc = lib.mv_inverted_residual_f32(x.data_ptr(), w1.data_ptr(), t.data_ptr(), t.data_ptr(), wd.data_ptr(), t.data_ptr(), t.data_ptr(),
                                      w2.data_ptr(), t.data_ptr(), t.data_ptr(), 1, y.data_ptr(), 2, 64, 384, 64, 14, 14, 1, 2, None, 0, None)
    assert rc != 0 and b"workspace" in lib.mv_last_error()
    rc = lib.mv_inverted_residual_f32(x.data_ptr(), w1.data_ptr(), t.data_ptr(), t.data_ptr(), wd.data_ptr(), t.data_ptr(), t.data_ptr(),
                                      w2.data_ptr(), t.data_ptr(), t.data_ptr(), 0, y.data_ptr(), 2, 32, 192, 32, 56, 56, 1, 2, None, 0, None)
    assert rc != 0 and b"no fused kernel" in lib.mv_last_error()
    with pytest.raises(ValueError):  # `+ x` needs stride 1 and cin == cout
        _lib.check(lib.mv_inverted_residual_f32(x.data_ptr(), w1.data_ptr(), t.data_ptr(), t.data_ptr(), wd.data_ptr(), t.data_ptr(), t.data_ptr(),
                                                w2.data_ptr(), t.data_ptr(), t.data_ptr(), 1, y.data_ptr(), 256, 64, 384, 96, 14, 14, 1, 2, None, 0, None))


def test_mobilenet_v2_vs_reference_fixture_and_oracle():
    g = golden("mobilenet_v2")
    torch.manual_seed(0)
    cpu = MobileNetV2(num_classes=10)
    randomize_norms(cpu, 7)
    assert abs(float(sum(p.detach().double().sum() for p in cpu.parameters())) - float(g["net__checksum"][0])) < 1e-9
    model = MobileNetV2(num_classes=10)
    model.load_state_dict(cpu.state_dict())
    model = model.cuda()
    x = g["net__x"]
    acts = oracle_mobilenet_features(ref, cpu, x)
    a = dev(x)
    for i, layer in enumerate(model.features):
        a = layer(a)
        np.testing.assert_array_equal(host(a), acts[i], err_msg=f"features[{i}] vs oracle (bit-exact)")
    want = g["net__features"]
    assert np.abs(host(a) - want).max() <= 2e-5 * float(np.abs(want).max())
    logits = host(model(dev(x)))
    np.testing.assert_allclose(logits, g["net__logits"], rtol=1e-4, atol=1e-5)
    pooled = ref.adaptive_avgpool(acts[-1], 1, 1).reshape(2, -1)
    fc = cpu.classifier[1]
    slices, slice_len = F.linear_k_slices(2, fc.in_features, fc.out_features)  # batch 2: K runs in slices over the chip
    np.testing.assert_array_equal(logits, ref.linear_bias_relu(pooled, fc.weight.detach().numpy(), fc.bias.detach().numpy(), relu=False,
                                                               slice_len=slice_len if slices > 1 else 0))


def test_mobilenet_v2_224_batch_properties():
    """ImageNet-size input: two images against the oracle bit for bit; the rows of a batch are independent of each other (bit for
    bit at a fixed batch size; across batch sizes the fused blocks' slice plans differ -- include/mi355vision.h, BATCH DEPENDENCE --
    and the logits agree to 1e-5); eval-only."""
    torch.manual_seed(1)
    cpu = MobileNetV2(num_classes=100)
    randomize_norms(cpu, 11)
    model = MobileNetV2(num_classes=100)
    model.load_state_dict(cpu.state_dict())
    model = model.cuda()
    x = philox_f32(9400, (6, 3, 224, 224)) * 2 - 1
    y = model(dev(x))
    assert y.shape == (6, 100)
    acts = oracle_mobilenet_features(ref, cpu, x[:2])
    np.testing.assert_array_equal(host(model.features(dev(x[:2]))), acts[-1])
    x2 = x.copy()
    x2[[0, 1, 2, 5]] = philox_f32(9401, (4, 3, 224, 224)) * 2 - 1
    assert torch.equal(model(dev(x2))[3:5], y[3:5])
    np.testing.assert_allclose(host(model(dev(x[3:5]))), host(y[3:5]), rtol=1e-5, atol=1e-5)
    model.train()
    with pytest.raises(RuntimeError, match="inference only"):
        model(dev(x[:1]))


def test_hip_graph_capture_replays_the_same_bits():
    """The C ABI allocates nothing and never synchronises: a whole forward captures into one HIP graph; replays equal the
    eager result bit for bit, for new inputs too."""
    from cpu_vision_amd import graphs
    torch.manual_seed(2)
    model = MobileNetV2(num_classes=20).cuda()
    x1 = dev(philox_f32(9500, (2, 3, 96, 96)))
    x2 = dev(philox_f32(9501, (2, 3, 96, 96)))
    want1, want2 = model(x1).clone(), model(x2).clone()
    cap = graphs.capture(model, x1)
    assert torch.equal(cap(x1), want1)
    assert torch.equal(cap(x2), want2)
    assert torch.equal(cap(x1), want1)
    with pytest.raises(ValueError):
        cap(x1[:1])


def test_parallel_branch_graph_of_independent_frames():
    """Frames that arrive as separate tensors: one call each, captured on parallel graph branches; replays read the frames'
    current contents and equal the eager calls bit for bit."""
    from cpu_vision_amd import graphs
    frames = [dev(philox_f32(9600 + i, (3, 40 + i, 64))) for i in range(5)]
    xu = dev(philox_u8(9610, (3, 33, 48)))
    calls = [(lambda f=f: F.gaussian_blur(f, [3, 3])) for f in frames] + [lambda: F.adjust_sharpness(xu, 1.6)]
    cap = graphs.capture_parallel(calls, streams=3)
    outs = cap.replay()
    torch.cuda.synchronize()
    for f, o in zip(frames, outs):
        assert torch.equal(o, F.gaussian_blur(f, [3, 3]))
    assert torch.equal(outs[-1], F.adjust_sharpness(xu, 1.6))
    for f in frames:  # new frame contents, same buffers
        f.mul_(0.5)
    outs = cap.replay()
    torch.cuda.synchronize()
    for f, o in zip(frames, outs):
        assert torch.equal(o, F.gaussian_blur(f, [3, 3]))
    with pytest.raises(ValueError):
        graphs.capture_parallel([])
