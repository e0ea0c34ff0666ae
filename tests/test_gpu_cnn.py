"""SURVEY.md 8(f).1 -- the rest of the small CNNs' feature extractor on the MI355X: general-cin conv3x3 (K-chunked
MFMA), MaxPool2d(2,2), AdaptiveAvgPool2d, and VGG-11 `features` end to end against the reference's own outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cpu_vision_amd as mv  # noqa: E402
from cpu_vision_amd import functional as F  # noqa: E402
from oracle import ref  # noqa: E402
from tests._util import oracle_conv3x3, assert_conv_close, golden, philox_f32  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 128, 12, 20), (1, 8, 32, 14, 14), (3, 16, 130, 9, 33), (1, 128, 256, 7, 7),
                                            (2, 5, 70, 17, 64), (1, 24, 8, 30, 302), (1, 64, 64, 40, 56), (2, 12, 40, 5, 3)])
def test_general_conv_bit_exact_vs_oracle(n, cin, cout, h, w):
    x = philox_f32(7000 + cin + h, (n, cin, h, w)) * 2 - 1
    wt = (philox_f32(7001 + cout, (cout, cin, 3, 3)) - 0.5) * (2.0 / (cin * 9)) ** 0.5 * 2
    b = philox_f32(7002 + w, (cout,)) - 0.5
    got = host(F.conv2d_bias_relu(dev(x), dev(wt), dev(b)))
    np.testing.assert_array_equal(got, oracle_conv3x3(ref, x, wt, b))
    got = host(F.conv2d_bias_relu(dev(x), dev(wt), None, relu=False))
    np.testing.assert_array_equal(got, oracle_conv3x3(ref, x, wt, None, relu=False))


@pytest.mark.parametrize("n,cin,cout,h,w", [(1, 512, 512, 14, 14), (1, 256, 512, 28, 28), (2, 512, 64, 14, 14), (1, 130, 40, 9, 11), (1, 512, 512, 7, 7)])
def test_small_launches_run_k_in_slices_across_workgroups(n, cin, cout, h, w):
    """VGG's late layers at batch 1 are a few dozen workgroups walking 64-128 K chunks each: conv2d_bias_relu cuts K into slices
    across workgroups (mv_conv3x3_bias_relu_ws_f32), the library states the order (mv_conv3x3_k_slices), the oracle restates it --
    bit-exact; the single chain stays available and the two agree to 1e-5 relative of sum |w x|."""
    from cpu_vision_amd import _lib
    x = philox_f32(7500 + cin + h, (n, cin, h, w)) * 2 - 1
    wt = (philox_f32(7501 + cout, (cout, cin, 3, 3)) - 0.5) * (2.0 / (cin * 9)) ** 0.5 * 2
    b = philox_f32(7502 + w, (cout,)) - 0.5
    slices, sc = F.conv3x3_k_slices(n, cin, h, w, cout)
    assert slices >= 1 and sc % 4 == 0 and slices * sc >= cin and (slices - 1) * sc < cin
    if cin >= 256:
        assert slices > 1
    got = host(F.conv2d_bias_relu(dev(x), dev(wt), dev(b)))
    if slices > 1:
        assert "k_conv3x3_reduce" in _lib.last_kernel()
    np.testing.assert_array_equal(got, ref.conv3x3_bias_relu(x, wt, b, slice_channels=sc if slices > 1 else 0))
    one = host(F.conv2d_bias_relu(dev(x), dev(wt), dev(b), relu=False, sliced_k=False))
    np.testing.assert_array_equal(one, ref.conv3x3_bias_relu(x, wt, b, relu=False))
    two = host(F.conv2d_bias_relu(dev(x), dev(wt), dev(b), relu=False))
    xp = np.pad(np.abs(x), ((0, 0), (0, 0), (1, 1), (1, 1)))
    mag = sum(np.einsum("nchw,oc->nohw", xp[:, :, dy:dy + h, dx:dx + w], np.abs(wt[:, :, dy, dx])) for dy in range(3) for dx in range(3))
    assert np.all(np.abs(one - two) <= 1e-5 * (mag + np.abs(b)[None, :, None, None]) + 1e-30)
    # the C entry point: workspace size, refusal without it
    lib = mv.load_library()
    if slices > 1:
        assert lib.mv_conv3x3_workspace_bytes(n, cin, h, w, cout) == 4 * slices * n * cout * h * w
        xd, wd_, yd = dev(x), dev(wt), torch.empty((n, cout, h, w), device="cuda")
        assert lib.mv_conv3x3_bias_relu_ws_f32(xd.data_ptr(), wd_.data_ptr(), None, yd.data_ptr(), n, cin, h, w, cout, 1, None, 0, None) == -1
        assert b"workspace" in lib.mv_last_error()


def test_wide_maps_take_the_first_generation_kernel():
    """Maps wider than 510 pixels do not fit k_conv3x3_gen's row tile: k_conv3x3 (conv3x3_mfma.hip) serves them -- the only
    shapes that still reach it -- with the same bits."""
    from cpu_vision_amd import _lib
    for (n, cin, cout, h, w) in [(1, 8, 16, 6, 600), (2, 3, 5, 4, 514), (1, 4, 70, 3, 1030)]:
        x = philox_f32(7400 + w, (n, cin, h, w)) - 0.5
        wt = (philox_f32(7401 + cout, (cout, cin, 3, 3)) - 0.5) * 0.5
        b = philox_f32(7402, (cout,)) - 0.5
        got = host(F.conv2d_bias_relu(dev(x), dev(wt), dev(b)))
        assert _lib.last_kernel() == "k_conv3x3", _lib.last_kernel()
        np.testing.assert_array_equal(got, oracle_conv3x3(ref, x, wt, b))


@pytest.mark.parametrize("group", ["1", "3", "64"])
def test_general_conv_image_stacking(group, monkeypatch, tuning_library):
    """Small maps are stacked into super-images (one zero separator row between images); any grouping, including one
    that does not divide the batch, must give the same bits."""
    monkeypatch.setenv("MV_CONV_GROUP", group)
    for (n, cin, cout, h, w) in [(5, 8, 40, 14, 14), (7, 16, 32, 7, 12), (4, 4, 130, 28, 28)]:
        x = philox_f32(7300 + h, (n, cin, h, w)) - 0.5
        wt = (philox_f32(7301 + w, (cout, cin, 3, 3)) - 0.5) * 0.4
        b = philox_f32(7302, (cout,)) - 0.5
        np.testing.assert_array_equal(host(F.conv2d_bias_relu(dev(x), dev(wt), dev(b))), oracle_conv3x3(ref, x, wt, b))


@pytest.mark.parametrize("spec", ["0", "1"])
def test_general_conv_largest_wave_tile_with_and_without_loader_waves(spec, monkeypatch, tuning_library):
    """The 128-channel x 256-pixel workgroup tile runs with loader / compute wave specialisation on grids of at most one
    workgroup per CU (k_conv3x3_gen<.., 4, 2, true, ..>) and without it on larger ones: both forced here on small problems
    (ragged widths, channel counts off the tile and off the 4-channel chunk), same bits as the oracle."""
    from cpu_vision_amd import _lib
    monkeypatch.setenv("MV_CONV_SHAPE", "0")
    monkeypatch.setenv("MV_CONV_SPEC42", spec)
    for (n, cin, cout, h, w) in [(3, 8, 40, 14, 14), (2, 13, 130, 7, 12), (1, 4, 200, 28, 28), (2, 16, 64, 20, 33)]:
        x = philox_f32(7500 + h, (n, cin, h, w)) - 0.5
        wt = (philox_f32(7501 + w, (cout, cin, 3, 3)) - 0.5) * 0.4
        b = philox_f32(7502, (cout,)) - 0.5
        np.testing.assert_array_equal(host(F.conv2d_bias_relu(dev(x), dev(wt), dev(b))), oracle_conv3x3(ref, x, wt, b))
        assert "k_conv3x3_gen" in _lib.last_kernel(), _lib.last_kernel()


@pytest.mark.parametrize("dense", ["0", "1"])
def test_general_conv_small_launch_kernel_with_and_without_the_dense_loop(dense, monkeypatch, tuning_library):
    """Small launches take the loader / compute kernel's dense form (buffer loads, an LDS address register set per buffer, no VALU
    instruction in the K loop; a ragged right edge is staged anchored at column w - 4, over its neighbour's cells); MV_CONV_DENSE=0
    forces the first form.  Same bits as the oracle either way, with widths off the 4-column groups, cout off the 32-channel tile
    and a K that is not a multiple of the loader's ring."""
    from cpu_vision_amd import _lib
    monkeypatch.setenv("MV_CONV_DENSE", dense)
    monkeypatch.setenv("MV_CONV_SHAPE", "2")
    for (n, cin, cout, h, w) in [(1, 8, 40, 28, 28), (2, 20, 64, 8, 16), (1, 4, 33, 5, 12), (3, 36, 32, 14, 16), (2, 12, 40, 14, 14),
                                 (1, 8, 32, 7, 13), (2, 4, 64, 9, 5), (1, 16, 33, 6, 27)]:
        x = philox_f32(7600 + h, (n, cin, h, w)) - 0.5
        wt = (philox_f32(7601 + w, (cout, cin, 3, 3)) - 0.5) * 0.4
        b = philox_f32(7602, (cout,)) - 0.5
        np.testing.assert_array_equal(host(F.conv2d_bias_relu(dev(x), dev(wt), dev(b))), oracle_conv3x3(ref, x, wt, b))
        assert "k_conv3x3_gen" in _lib.last_kernel(), _lib.last_kernel()


def test_cnn_layers_vs_reference_fixtures():
    g = golden("cnn_layers")
    w, b = g["c64_128__w"], g["c64_128__b"]
    y = F.conv2d_bias_relu(dev(g["c64_128__x"]), dev(w), dev(b))
    assert_conv_close(host(y), g["c64_128__y"], float(np.abs(w).reshape(128, -1).sum(1).max()), 0.7, what="conv 64->128 vs reference")
    np.testing.assert_array_equal(host(F.max_pool2d_2x2(dev(g["c64_128__y"]))), g["pool__y"])
    np.testing.assert_allclose(host(F.adaptive_avg_pool2d(dev(g["avg__x"]), (7, 7))), g["avg__y"], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(host(F.adaptive_avg_pool2d(dev(g["avg__x"]), (7, 7))), ref.adaptive_avgpool(g["avg__x"], 7, 7))
    # global pooling of small planes (MobileNetV2's 7 x 7 head) has its own kernel: loads first, then the oracle's row-major sum
    from cpu_vision_amd import _lib
    for shape in ((3, 5, 7, 7), (2, 300, 8, 8), (1, 7, 1, 1), (2, 3, 5, 9)):
        xs = philox_f32(7100 + shape[2], shape) * 2 - 1
        np.testing.assert_array_equal(host(F.adaptive_avg_pool2d(dev(xs), (1, 1))), ref.adaptive_avgpool(xs, 1, 1))
        assert _lib.last_kernel() == "k_global_avgpool_small"
    xs = philox_f32(7200, (2, 3, 9, 9))  # 81 elements: the generic kernel
    np.testing.assert_array_equal(host(F.adaptive_avg_pool2d(dev(xs), (1, 1))), ref.adaptive_avgpool(xs, 1, 1))
    assert _lib.last_kernel() == "k_adaptive_avgpool"


@pytest.mark.parametrize("shape", [(2, 3, 8, 16), (1, 5, 7, 9), (3, 1, 224, 224), (1, 2, 2, 2), (1, 1, 3, 30)])
def test_maxpool_bit_exact(shape):
    x = philox_f32(7100 + shape[-1], shape) - 0.5
    np.testing.assert_array_equal(host(F.max_pool2d_2x2(dev(x))), ref.maxpool2x2(x))
    xn = x.copy()
    xn[..., 0, 1] = np.nan
    got = host(F.max_pool2d_2x2(dev(xn)))
    assert np.isnan(got[..., 0, 0]).all()  # NaN propagates like ATen's max_pool2d


def test_vgg11_features_vs_reference():
    """vgg11(num_classes=50), seed 0 -- the reference's test_classification_model recipe (test_models.py:674-693);
    weights are rebuilt with the same seeded constructor sequence and checked against stored checksums."""
    from cpu_vision_amd.nn import VGGFeatures, vgg11_reference_init
    g = golden("vgg11_forward")
    state = vgg11_reference_init(num_classes=50, seed=0)
    for name, s_sum, s_abs in zip(g["param_names"], g["param_sum"], g["param_abs_sum"]):
        p = state[str(name)].double()
        assert abs(float(p.sum()) - s_sum) <= 1e-9 * max(1.0, s_abs) and abs(float(p.abs().sum()) - s_abs) <= 1e-9 * s_abs, name
    feats = VGGFeatures("A").cuda()
    feats.load_reference_state_dict(state)
    x = dev(g["x"])
    out = feats(x)
    assert out.shape == (1, 512, 7, 7)
    want = g["features"]
    err = np.abs(host(out).astype(np.float64) - want)
    assert err.max() <= 1e-5 * np.abs(want).max() + 1e-6, err.max()
    part = feats.run_prefix(x, 6)  # conv, relu, pool, conv, relu, pool
    assert_conv_close(host(part)[:, :8], g["features_0_6"], 1.0, float(np.abs(g["features_0_6"]).max()), rel=2e-5, what="features[0:6]")


@pytest.mark.parametrize("n,k,m", [(5, 300, 70), (1, 64, 32), (130, 257, 33), (64, 1024, 512), (3, 31, 5), (256, 96, 200),
                                   (1, 25088, 48), (2, 4100, 300), (33, 2049, 129), (600, 512, 4096),
                                   # batch <= 4: k_linear_gemv (64-k stages, ragged slice ends, m past the last 128-row block)
                                   (4, 9216, 1000), (3, 100, 129), (2, 68, 260), (4, 25088, 130), (1, 4096, 1000)])
def test_linear_bit_exact_vs_oracle(n, k, m):
    x = philox_f32(7200 + k, (n, k)) - 0.5
    w = (philox_f32(7201 + m, (m, k)) - 0.5) * 0.2
    b = philox_f32(7202 + n, (m,)) - 0.5
    # inference-size batches run K in slices (include/mi355vision.h: mv_linear_bias_relu_ws_f32); the library states the
    # slicing, the oracle restates that summation order
    slices, slice_len = F.linear_k_slices(n, k, m)
    assert slices >= 1 and slice_len % 32 == 0 and slices * slice_len >= k and (slices - 1) * slice_len < k
    sl = slice_len if slices > 1 else 0
    got = host(F.linear_bias_relu(dev(x), dev(w), dev(b), relu=True))
    np.testing.assert_array_equal(got, ref.linear_bias_relu(x, w, b, relu=True, slice_len=sl))
    got = host(F.linear_bias_relu(dev(x), dev(w), None, relu=False))
    np.testing.assert_array_equal(got, ref.linear_bias_relu(x, w, None, relu=False, slice_len=sl))
    # the single ascending-k chain stays available, and the two orders agree to 1e-5 relative of sum |w x|
    one = host(F.linear_bias_relu(dev(x), dev(w), dev(b), relu=False, sliced_k=False))
    np.testing.assert_array_equal(one, ref.linear_bias_relu(x, w, b, relu=False))
    two = host(F.linear_bias_relu(dev(x), dev(w), dev(b), relu=False))
    mag = np.abs(x) @ np.abs(w).T + np.abs(b)
    assert np.all(np.abs(one - two) <= 1e-5 * mag + 1e-30)
    g = golden("cnn_layers")
    yl = host(F.linear_bias_relu(dev(g["lin__x"]), dev(g["lin__w"]), dev(g["lin__b"]), relu=True))
    assert_conv_close(yl, g["lin__y"], float(np.abs(g["lin__w"]).sum(1).max()), 0.5, what="linear+relu vs reference")


def test_vgg11_whole_forward_vs_reference():
    """The reference's own model test (test_models.py:674-693): vgg11(num_classes=50), seed 0, torch.rand(1,3,224,224),
    compared with the reference's output (our fixture) AND with its committed expect file (prec 0.1 there; 1e-5 here)."""
    from cpu_vision_amd.nn import vgg11, vgg11_reference_init
    g = golden("vgg11_forward")
    model = vgg11(num_classes=50).cuda().eval()
    model.load_reference_state_dict(vgg11_reference_init(num_classes=50, seed=0))
    with torch.no_grad():
        y = host(model(dev(g["x"])))
    assert y.shape == (1, 50)
    scale = float(np.abs(g["y"]).max())
    assert np.abs(y - g["y"]).max() <= 1e-5 * scale + 1e-6, np.abs(y - g["y"]).max()
    assert np.abs(y - g["reference_expect_pkl"]).max() <= 1e-5 * scale + 1e-6
    with pytest.raises(RuntimeError, match="inference-only"):
        model.train()(dev(g["x"]))


def test_preset_tail_vs_reference_and_oracle():
    """convert_image_dtype(float) + normalize (transforms/_presets.py:58-60), standalone and fused into
    the first conv's load."""
    g = golden("preset_tail")
    mean, std = [float(v) for v in g["mean"]], [float(v) for v in g["std"]]
    xu = dev(g["x_u8"])
    np.testing.assert_array_equal(host(F.to_dtype(xu, torch.float32, scale=True)), g["to_float"])
    np.testing.assert_array_equal(host(F.normalize(dev(g["to_float"]), mean, std)), g["normalized"])
    np.testing.assert_array_equal(host(F.to_float_normalize(xu, mean, std)), g["normalized"])
    np.testing.assert_array_equal(host(F.to_float_normalize(dev(g["gray_u8"]), [0.5], [0.25])), g["gray_normalized"])
    assert F.to_dtype(xu, torch.uint8) is xu and F.to_dtype(xu, torch.float32).dtype == torch.float32
    with pytest.raises(ValueError, match="std evaluated to zero"):
        F.normalize(dev(g["to_float"]), mean, [0.2, 0.0, 0.1])
    with pytest.raises(TypeError, match="should be a float tensor"):
        F.normalize(xu, mean, std)
    # fused into the first layer: same bits as normalising first and convolving after
    from cpu_vision_amd.nn import vgg11_reference_init
    st = vgg11_reference_init(50, 0)
    w, b = st["features.0.weight"].cuda(), st["features.0.bias"].cuda()
    fused = F.normalized_conv2d_bias_relu(xu, mean, std, w, b)
    two_step = F.conv2d_bias_relu(F.to_float_normalize(xu, mean, std), w, b)
    assert torch.equal(fused, two_step)
    gain = float(w.abs().reshape(64, -1).sum(1).max())
    assert_conv_close(host(fused), g["vgg_first_layer"], gain, float(np.abs(g["normalized"]).max()), what="preset tail -> vgg features[0:2]")
    # ragged / large shapes against the oracle
    for shape in [(1, 3, 7, 20), (2, 3, 224, 224), (3, 1, 33, 17), (1, 4, 64, 64)]:
        x = np.random.default_rng(shape[-1]).integers(0, 256, shape, dtype=np.uint8)
        mm = [0.4, 0.5, 0.3, 0.2][: shape[1]]
        ss = [0.2, 0.25, 0.3, 0.5][: shape[1]]
        np.testing.assert_array_equal(host(F.to_float_normalize(dev(x), mm, ss)), ref.to_float_normalize(x, mm, ss))
        if shape[1] == 3 and shape[-1] % 4 == 0:
            wt = (np.random.default_rng(5).random((40, 3, 3, 3), dtype=np.float32) - 0.5)
            got = host(F.normalized_conv2d_bias_relu(dev(x), mm, ss, dev(wt), None))
            np.testing.assert_array_equal(got, ref.conv3x3_bias_relu(ref.to_float_normalize(x, mm, ss), wt, None))


# ----------------------------------------------------------------------------- AlexNet + the generic conv / pooling kernels
@pytest.mark.parametrize("n,cin,cout,h,w,k,stride,pad,dil,groups", [
    (2, 3, 64, 67, 63, (11, 11), (4, 4), (2, 2), (1, 1), 1),    # AlexNet conv1
    (2, 64, 192, 27, 27, (5, 5), (1, 1), (2, 2), (1, 1), 1),    # AlexNet conv2
    (1, 8, 12, 15, 17, (3, 5), (2, 1), (1, 2), (2, 1), 4),      # groups, dilation, anisotropic everything
    (3, 5, 7, 9, 9, (1, 1), (1, 1), (0, 0), (1, 1), 1),
    (1, 4, 6, 6, 6, (6, 6), (1, 1), (0, 0), (1, 1), 2),         # kernel = image
    (2, 16, 130, 40, 36, (3, 3), (2, 2), (1, 1), (1, 1), 1),    # two channel blocks, 342 output pixels: several pixel tiles per image
    (1, 6, 10, 3, 50, (3, 7), (1, 3), (1, 3), (1, 2), 2),       # a 3-row image: output rows shorter than a tile, wide dilated kernel
    (5, 3, 8, 12, 12, (2, 2), (1, 1), (0, 0), (1, 1), 1),       # even kernel, no padding
])
def test_generic_conv2d_bit_exact_vs_oracle(n, cin, cout, h, w, k, stride, pad, dil, groups, monkeypatch):
    """mv_conv2d_bias_act_f32 is an implicit GEMM (round 3): the K chunk's im2col columns are gathered from the input while the
    GEMM stages them -- no columns in HBM, no workspace.  Bit-exact against the oracle, and against the columns form (plain im2col
    into a workspace + the same GEMM: tuning build, MV_CONV_COLUMNS)."""
    from cpu_vision_amd import _lib
    x = philox_f32(7600 + cin + h, (n, cin, h, w)) * 2 - 1
    wt = (philox_f32(7601 + cout, (cout, cin // groups, k[0], k[1])) - 0.5) * (2.0 / (cin // groups * k[0] * k[1])) ** 0.5 * 2
    b = philox_f32(7602, (cout,)) - 0.5
    lib = _lib.load()
    # 0 = the implicit kernel; 2 = optional workspace (these test shapes are a handful of workgroups): run WITHOUT it here
    assert lib.mv_conv2d_needs_workspace(n, cin, cout, h, w, k[0], k[1], stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], groups) in (0, 2)
    monkeypatch.setattr(F, "CONV2D_SMALL_LAUNCH_WORKSPACE", False)
    got = host(F.conv2d_bias_act(dev(x), dev(wt), dev(b), stride=stride, padding=pad, dilation=dil, groups=groups, activation="relu"))
    assert "implicit" in _lib.last_kernel(), _lib.last_kernel()
    monkeypatch.setattr(F, "CONV2D_SMALL_LAUNCH_WORKSPACE", True)
    small = host(F.conv2d_bias_act(dev(x), dev(wt), dev(b), stride=stride, padding=pad, dilation=dil, groups=groups, activation="relu"))
    np.testing.assert_array_equal(small, got, err_msg="library's choice for a small launch vs the implicit kernel")
    import os
    with _lib.tuning_library():
        os.environ["MV_CONV_COLUMNS"] = "1"
        F.CONV2D_COLUMNS_WORKSPACE = True
        try:
            cols = host(F.conv2d_bias_act(dev(x), dev(wt), dev(b), stride=stride, padding=pad, dilation=dil, groups=groups, activation="relu"))
            assert "implicit" not in _lib.last_kernel()
        finally:
            F.CONV2D_COLUMNS_WORKSPACE = False
            os.environ.pop("MV_CONV_COLUMNS")
    np.testing.assert_array_equal(got, cols, err_msg="implicit GEMM vs im2col + GEMM")
    oh, ow = got.shape[-2:]
    zero_off = np.zeros((n, 2 * k[0] * k[1], oh, ow), np.float32)
    want = np.maximum(ref.deform_conv2d(x, zero_off, wt, b, stride, pad, dil, None), 0)  # zero offsets = conv2d, any geometry
    np.testing.assert_array_equal(got, want)
    if dil == (1, 1) and groups == 1 and stride[0] == stride[1] and pad[0] == pad[1]:
        np.testing.assert_array_equal(got, ref.conv2d_affine_act(x, wt, b, None, None, None, stride[0], pad[0], 1, 0, "relu"))


@pytest.mark.parametrize("shape,k,s", [((2, 5, 55, 55), 3, 2), ((1, 3, 27, 27), 3, 2), ((2, 2, 13, 13), 3, 2), ((1, 1, 3, 3), 3, 2),
                                       ((2, 3, 10, 12), 2, 2), ((1, 2, 9, 7), 4, 1)])
def test_max_pool2d_bit_exact_vs_oracle(shape, k, s):
    x = philox_f32(7700 + shape[-1], shape) - 0.5
    x[0, 0, 0, 0] = np.nan
    np.testing.assert_array_equal(host(F.max_pool2d(dev(x), k, s)), ref.maxpool2d(x, k, s))


def _oracle_alexnet_features(model, x, stop=None):
    a = x
    mods = list(model.features)[:stop]
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Conv2d):
            if m.kernel_size == (3, 3) and m.padding == (1, 1) and m.stride == (1, 1):  # conv2d_bias_relu: K slices at this batch size
                a = oracle_conv3x3(ref, a, m.weight.detach().numpy(), m.bias.detach().numpy())
            else:
                a = ref.conv2d_affine_act(a, m.weight.detach().numpy(), m.bias.detach().numpy(), None, None, None, m.stride[0], m.padding[0],
                                          1, 0, "relu")
            i += 2
        else:
            a = ref.maxpool2d(a, m.kernel_size, m.stride)
            i += 1
    return a


def test_alexnet_vs_reference_fixture_expect_file_and_oracle():
    """models/alexnet.py end to end: alexnet(num_classes=50), torch.manual_seed(0) -- the reference's own model test
    (test_models.py:674-693) and its committed expect file -- and every stage against the oracle bit for bit."""
    from cpu_vision_amd.nn import AlexNet
    g = golden("alexnet_forward")
    torch.manual_seed(0)
    cpu = AlexNet(num_classes=50)
    x = torch.rand(1, 3, 224, 224)
    assert abs(float(sum(p.detach().double().sum() for p in cpu.parameters())) - float(g["checksum"][0])) < 1e-9
    assert abs(float(x.double().sum()) - float(g["x_checksum"][0])) < 1e-9
    model = AlexNet(num_classes=50)
    model.load_state_dict(cpu.state_dict())
    model = model.cuda()
    xd = x.cuda()
    xn = x.numpy()
    for stop, key in ((3, "features_0_3"), (6, "features_0_6")):
        got = host(model.run_features(xd, stop))
        np.testing.assert_array_equal(got, _oracle_alexnet_features(cpu, xn, stop), err_msg=f"features[0:{stop}] vs oracle")
        want = g[key]
        assert np.abs(got[:, :8] - want).max() <= 2e-5 * max(1.0, float(np.abs(want).max())), key
    feats = host(model.run_features(xd))
    np.testing.assert_array_equal(feats, _oracle_alexnet_features(cpu, xn))
    assert np.abs(feats - g["features"]).max() <= 2e-5 * float(np.abs(g["features"]).max())
    y = host(model(xd))
    np.testing.assert_allclose(y, g["y"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(y, g["reference_expect_pkl"], rtol=1e-4, atol=1e-6)
