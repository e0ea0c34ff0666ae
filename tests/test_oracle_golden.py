"""Pins the CPU oracle (oracle/oracle.c, oracle/ref_torch.py) to the reference.

Three anchors, none of which needs /root/reference at run time:
  1. the reference's own OpenCV golden vectors (test/assets/gaussian_blur_opencv_results.pt,
     re-encoded), with the reference's own tolerance (atol=1, test_transforms_v2.py:3273-3309);
  2. fixtures produced by importing and running the reference (tests/golden/make_golden.py);
  3. PIL-exact uint8 sharpness (test_transforms_v2.py:4721-4731).
"""
import numpy as np
import pytest
import torch

from oracle import ref, ref_torch
from tests._util import assert_conv_close, golden

BORD = {"reflect": ref.BORDER_REFLECT, "zero": ref.BORDER_ZERO, "valid": ref.BORDER_VALID}


def _k1d_pair(ks, sg, v1=False):
    if sg is None:
        sg = [k * 0.15 + 0.35 for k in ks]
    kx = ref_torch.gaussian_kernel1d(ks[0], sg[0], v1=v1).numpy()
    ky = ref_torch.gaussian_kernel1d(ks[1], sg[1], v1=v1).numpy()
    return kx, ky


def _parse_blur_name(name):
    # k{kx}x{ky}_s{d|sx_sy}_{dt}_{shape}
    parts = name.split("_")
    kx, ky = map(int, parts[0][1:].split("x"))
    if parts[1] == "sd":
        sg, rest = None, parts[2:]
    else:
        sg, rest = [float(parts[1][1:]), float(parts[2])], parts[3:]
    return [kx, ky], sg, rest[0]


# ----------------------------------------------------------------------------- kernels
def test_gaussian_kernel1d_matches_reference():
    g = golden("gaussian_kernels")
    for k, s in g["cases"]:
        k = int(k)
        for ver in ("v2", "v1"):
            want = g[f"{ver}_{k}_{s}"]
            got_c = ref.gaussian_kernel1d(k, s, v1=(ver == "v1"))
            got_t = ref_torch.gaussian_kernel1d(k, s, v1=(ver == "v1")).numpy()
            np.testing.assert_array_equal(got_t, want)  # same torch ops -> bit-identical
            np.testing.assert_allclose(got_c, want, rtol=2e-6, atol=1e-9)  # libm expf vs sleef: few ulp
    np.testing.assert_array_equal(ref_torch.gaussian_kernel2d([3, 5], [0.8, 0.5]).numpy(), g["v2_2d_3x5"])
    kx, ky = _k1d_pair([3, 5], [0.8, 0.5])
    np.testing.assert_array_equal(ref.gaussian_kernel2d(kx, ky), g["v2_2d_3x5"])


# ----------------------------------------------------------------------------- OpenCV vectors
@pytest.mark.parametrize("dims,ks,sigma", [((3, 10, 12), (3, 3), 0.8), ((3, 10, 12), (3, 3), 0.5),
                                           ((3, 10, 12), (3, 5), 0.8), ((3, 10, 12), (3, 5), 0.5),
                                           ((1, 26, 28), (23, 23), 1.7)])
@pytest.mark.parametrize("dtype", ["f32", "u8"])
def test_oracle_vs_opencv_vectors(dims, ks, sigma, dtype):
    c, h, w = dims
    want = golden("opencv_gaussian_blur")[f"{h}_{w}_{c}__{ks[0]}_{ks[1]}_{sigma}"].reshape(h, w, c).transpose(2, 0, 1)
    x = np.arange(c * h * w, dtype=np.uint8).reshape(h, w, c).transpose(2, 0, 1)
    kx, ky = _k1d_pair(list(ks), [sigma, sigma])
    got = ref.gaussian_blur(np.ascontiguousarray(x) if dtype == "u8" else x.astype(np.float32), kx, ky)
    assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= 1.0


# ----------------------------------------------------------------------------- gaussian blur fixtures
def test_oracle_gaussian_blur_vs_reference_fixtures():
    g = golden("gaussian_blur")
    n_u8 = n_u8_diff = 0
    for name in g["index"]:
        ks, sg, dt = _parse_blur_name(str(name))
        x, want = g[f"{name}__x"], g[f"{name}__y_v2"]
        kx, ky = _k1d_pair(ks, sg)
        got = ref.gaussian_blur(x, kx, ky)
        got_t = ref_torch.gaussian_blur_image(torch.from_numpy(x), ks, sg).numpy()
        assert got.dtype == want.dtype and got.shape == want.shape
        # the torch call-sequence port runs torch's own CPU kernels: bit-identical to the fixtures on the host that generated
        # them, within an ulp on hosts whose oneDNN picks another summation order (the GPU box's CPU does)
        if dt == "u8":
            assert np.abs(got_t.astype(np.int32) - want.astype(np.int32)).max() <= 1, f"torch call-sequence port {name}"
        else:
            assert_conv_close(got_t, want, 1.0, 1.0, what=f"torch call-sequence port {name}")
        if dt == "u8":
            # fp32 sum order differs from oneDNN's, so a value within ~1e-5 of x.5 may round the other way:
            # never more than 1 LSB, and rare
            d = np.abs(got.astype(np.int32) - want.astype(np.int32))
            assert d.max() <= 1, name
            n_u8 += d.size
            n_u8_diff += int((d != 0).sum())
        else:
            assert_conv_close(got, want, 1.0, 1.0, what=str(name))
        if f"{name}__y_v1" in g.files:
            want1 = g[f"{name}__y_v1"]
            kx1, ky1 = _k1d_pair(ks, sg, v1=True)
            got1 = ref.gaussian_blur(x, kx1, ky1)
            if dt == "u8":
                assert np.abs(got1.astype(np.int32) - want1.astype(np.int32)).max() <= 1
            else:
                assert_conv_close(got1, want1, 1.0, 1.0, what=f"{name} v1")
    assert n_u8 > 0 and n_u8_diff <= 1e-3 * n_u8, (n_u8_diff, n_u8)


# ----------------------------------------------------------------------------- sharpness fixtures
def _sharp_cases():
    g = golden("adjust_sharpness")
    for name in g["index"]:
        name = str(name)
        f = float(name.split("_")[0][1:])
        xkey = name.split("_", 1)[1] + "__x"
        yield name, f, g[xkey], g[f"{name}__y_v2"], (g[f"{name}__y_v1"] if f"{name}__y_v1" in g.files else None)


def test_oracle_sharpness_vs_reference_fixtures():
    for name, f, x, want2, want1 in _sharp_cases():
        got = ref.adjust_sharpness(x, f)
        got_t = ref_torch.adjust_sharpness_image(torch.from_numpy(x), f).numpy()
        if x.dtype == np.uint8:
            np.testing.assert_array_equal(got_t, want2, err_msg=f"torch port {name}")
        else:  # float results of torch's CPU conv may differ by an ulp between hosts
            assert_conv_close(got_t, want2, 1.0 + abs(1 - f) * 2, 1.0, what=f"torch port {name}")
        if x.dtype == np.uint8:
            np.testing.assert_array_equal(got, want2, err_msg=name)  # integer contract: bit-exact
        else:
            assert_conv_close(got, want2, 1.0 + abs(1 - f) * 2, 1.0, what=name)
        if want1 is not None:
            got1 = ref.adjust_sharpness(x, f, v1=True)
            if x.dtype == np.uint8:
                np.testing.assert_array_equal(got1, want1, err_msg=f"{name} v1")
            else:
                assert_conv_close(got1, want1, 1.0 + abs(1 - f) * 2, 1.0, what=f"{name} v1")


def test_oracle_sharpness_pil_exact():
    g = golden("adjust_sharpness")
    x = g["pil__x"]
    for f in (0.1, 0.5, 1.0):
        np.testing.assert_array_equal(ref.adjust_sharpness(x, f), g[f"pil__y_{f}"])


# ----------------------------------------------------------------------------- primitive: generic taps, box, separable, sobel
def test_oracle_primitive_vs_reference_fixtures():
    g = golden("primitive_filters")
    assert_conv_close(ref.box_filter(g["box__x"], 3), g["box__y"], what="box")
    x = g["gen__x"]
    for ky, kx in [(3, 3), (5, 3), (1, 5), (7, 7)]:
        w = g[f"gen_{ky}x{kx}__w"]
        for b in ("reflect", "zero", "valid"):
            got = ref.depthwise_conv2d(x, w, BORD[b])
            assert_conv_close(got, g[f"gen_{ky}x{kx}_{b}__y"], float(np.abs(w).sum()), 1.0, what=f"gen {ky}x{kx} {b}")
    xs, k1 = g["sep__x"], g["sep__k1d"]
    blur = ref.separable_blur(xs, k1, k1)
    assert_conv_close(blur, g["sep__blur"], what="separable blur")
    gx, gy = ref.gaussian_sobel(xs, k1, k1)
    assert_conv_close(gx, g["sep__gx"], 8.0, 1.0, what="blur->sobel gx")
    assert_conv_close(gy, g["sep__gy"], 8.0, 1.0, what="blur->sobel gy")
    for b in ("reflect", "zero", "valid"):
        gx, gy = ref.sobel(xs, BORD[b])
        assert_conv_close(gx, g[f"sobel_{b}__gx"], 8.0, 1.0, what=f"sobel {b} gx")
        assert_conv_close(gy, g[f"sobel_{b}__gy"], 8.0, 1.0, what=f"sobel {b} gy")


def test_oracle_c_vs_python_loops_small():
    rng = np.random.default_rng(7)
    x = rng.random((2, 6, 7), dtype=np.float32)
    for ky, kx in [(3, 3), (1, 3), (5, 1)]:
        w = rng.random((ky, kx), dtype=np.float32) - 0.5
        for b in (ref.BORDER_REFLECT, ref.BORDER_ZERO, ref.BORDER_VALID):
            np.testing.assert_allclose(ref.depthwise_conv2d(x, w, b), ref.depthwise_conv2d_py(x, w, b), rtol=1e-5, atol=1e-6)


# ----------------------------------------------------------------------------- first CNN layer
def test_oracle_conv_relu_vs_reference_fixtures():
    g = golden("conv_relu")
    w, b = g["vgg11__w"], g["vgg11__b"]
    gain = float(np.abs(w).reshape(64, -1).sum(1).max())
    assert_conv_close(ref.conv3x3_bias_relu(g["vgg11__x"], w, b), g["vgg11__y"], gain, 1.0, what="vgg11 features[0:2]")
    assert_conv_close(ref.conv3x3_bias_relu(g["bias__x"], w, g["bias__b"], relu=False), g["bias__y_norelu"], gain, 1.0, what="bias no relu")
    assert_conv_close(ref.conv3x3_bias_relu(g["bias__x"], w, g["bias__b"]), g["bias__y"], gain, 1.0, what="bias relu")
    assert_conv_close(ref.conv3x3_bias_relu(g["vgg11__x"], g["cna__w"], g["cna__b"]), g["cna__y"],
                      float(np.abs(g["cna__w"]).reshape(64, -1).sum(1).max()), 1.0, what="Conv2dNormActivation")
    assert_conv_close(ref.conv3x3_bias_relu(g["c16__x"], g["c16__w"], g["c16__b"]), g["c16__y"],
                      float(np.abs(g["c16__w"]).reshape(32, -1).sum(1).max()), 0.5, what="Cin=16")
    yt = ref_torch.conv3x3_bias_relu(torch.from_numpy(g["vgg11__x"]), torch.from_numpy(w), torch.from_numpy(b)).numpy()
    assert_conv_close(yt, g["vgg11__y"], gain, 1.0, what="torch port")


# ----------------------------------------------------------------------------- rest of the small CNN (8f.1)
def test_oracle_cnn_layers_vs_reference_fixtures():
    g = golden("cnn_layers")
    w, b = g["c64_128__w"], g["c64_128__b"]
    y = ref.conv3x3_bias_relu(g["c64_128__x"], w, b)
    assert_conv_close(y, g["c64_128__y"], float(np.abs(w).reshape(128, -1).sum(1).max()), 0.7, what="conv 64->128")
    np.testing.assert_array_equal(ref.maxpool2x2(g["c64_128__y"]), g["pool__y"])
    yl = ref.linear_bias_relu(g["lin__x"], g["lin__w"], g["lin__b"], relu=True)
    assert_conv_close(yl, g["lin__y"], float(np.abs(g["lin__w"]).sum(1).max()), 0.5, what="linear+relu")
    ya = ref.adaptive_avgpool(g["avg__x"], 7, 7)
    np.testing.assert_allclose(ya, g["avg__y"], rtol=1e-6, atol=1e-7)


def test_oracle_preset_tail_vs_reference_fixtures():
    g = golden("preset_tail")
    np.testing.assert_array_equal(ref.to_float_normalize(g["x_u8"]), g["to_float"])
    np.testing.assert_array_equal(ref.to_float_normalize(g["x_u8"], g["mean"], g["std"]), g["normalized"])
    np.testing.assert_array_equal(ref.to_float_normalize(g["to_float"], g["mean"], g["std"]), g["normalized"])
    np.testing.assert_array_equal(ref.to_float_normalize(g["gray_u8"], [0.5], [0.25]), g["gray_normalized"])


def _resize_case(g, name):
    a = g[f"{name}__args"]
    return g[f"{name}__x"], [int(v) for v in a[:-2]], (None if a[-2] < 0 else int(a[-2])), int(a[-1])


def test_oracle_resize_preset_vs_reference_fixtures():
    """8f.2 head: the oracle's restatement of F.resize(bilinear, antialias) / center_crop / the whole
    ImageClassification.forward is BIT-EXACT against the reference's own outputs (tests/golden/resize_preset.npz)."""
    g = golden("resize_preset")
    n_preset = 0
    for name in map(str, g["index"]):
        x, size, max_size, crop = _resize_case(g, name)
        r = ref.resize(x, size, max_size)
        np.testing.assert_array_equal(r, g[f"{name}__resized"], err_msg=name)
        np.testing.assert_array_equal(ref.center_crop(r, [crop]), g[f"{name}__cropped"], err_msg=name)
        if f"{name}__preset" in g.files:
            mean, std = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)) if x.shape[-3] == 3 else ((0.45,), (0.25,))
            got = ref.image_classification_preset(x, crop, size[0], mean, std)
            np.testing.assert_array_equal(got, g[f"{name}__preset"], err_msg=name)
            n_preset += 1
    assert n_preset >= 7


def test_oracle_resize_properties():
    """Size-independent properties of the antialiased resize: constants are preserved, identity size is the identity,
    weights are a partition of unity (a linear ramp's interior is reproduced), uint8 output stays in range."""
    c = np.full((2, 37, 91), 0.37, np.float32)
    np.testing.assert_allclose(ref.resize(c, [13, 29]), 0.37, rtol=3e-7)
    x = np.random.Generator(np.random.Philox(3)).random((1, 20, 30), dtype=np.float32)
    np.testing.assert_array_equal(ref.resize(x, [20, 30]), x)
    ramp = np.tile(np.arange(64, dtype=np.float32), (1, 8, 1))
    r = ref.resize(ramp, [8, 16])  # 4x downscale: centres at 4i + 1.5
    np.testing.assert_allclose(r[0, 0, 2:-2], 4 * np.arange(16)[2:-2] + 1.5, rtol=1e-6)
    u = np.random.Generator(np.random.Philox(4)).integers(0, 256, (3, 50, 70), dtype=np.uint8)
    up = ref.resize(u, [100])
    assert up.shape == (3, 100, 140) and up.dtype == np.uint8
    assert ref.resized_output_size(375, 500, [256]) == (256, 341) and ref.resized_output_size(500, 375, [256]) == (341, 256)


_MB_ACT = {"relu6": "relu6", "hswish": "hardswish", "relu": "relu", "linear": None, "silu": "silu"}


def _mb_block(g, name):
    """Rebuild a fixture block with cpu_vision_amd.mobilenet's containers (parameters from the fixture)."""
    import torch
    from cpu_vision_amd.mobilenet import FrozenBatchNorm2d
    x, w, nrm = g[f"{name}__x"], g[f"{name}__w"], g[f"{name}__norm"]
    kind, stride_tag = name.split("_")[0], name.split("_")[1]
    stride = 2 if (kind == "stem" or stride_tag == "s2") else 1
    cout, cg, k, _ = w.shape
    groups = cout if kind == "dw" else 1
    conv = torch.nn.Conv2d(x.shape[1], cout, k, stride, (k - 1) // 2, groups=groups, bias=False)
    norm = FrozenBatchNorm2d(cout) if "frozen" in name else torch.nn.BatchNorm2d(cout).eval()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
        for t, v in zip((norm.weight, norm.bias, norm.running_mean, norm.running_var), nrm):
            t.copy_(torch.from_numpy(v))
    return x, conv, norm, _MB_ACT[name.split("_")[-1]]


def test_oracle_conv_norm_act_vs_reference_fixtures():
    """8f.3: conv -> folded norm -> activation.  (1) The norm + activation arithmetic is pinned BIT-EXACTLY: applied
    to torch's own conv output it reproduces the reference's block output.  (2) The whole block through the oracle's
    conv (fixed fmaf order; torch's conv fixes none) agrees to 1e-5."""
    import torch
    from tests._util import oracle_conv_block
    g = golden("mobilenet_v2")
    for name in map(str, g["index"]):
        x, conv, norm, act = _mb_block(g, name)
        want = g[f"{name}__y"]
        with torch.no_grad():
            c = conv(torch.from_numpy(x)).numpy()
        ident = torch.nn.Conv2d(c.shape[1], c.shape[1], 1, groups=c.shape[1], bias=False)  # 1x1 depthwise with w = 1: exact copy
        with torch.no_grad():
            ident.weight.fill_(1.0)
        pinned = oracle_conv_block(ref, c, ident, norm, act)
        if act == "silu":
            np.testing.assert_allclose(pinned, want, rtol=2e-6, atol=1e-7, err_msg=name)
        else:
            np.testing.assert_array_equal(pinned, want, err_msg=f"{name}: norm/activation arithmetic")
        got = oracle_conv_block(ref, x, conv, norm, act)
        gain = float(np.abs(conv.weight.detach().numpy()).reshape(conv.out_channels, -1).sum(1).max()) * 3.0
        assert_conv_close(got, want, gain, float(np.abs(x).max()), what=name)


def test_oracle_mobilenet_v2_vs_reference_fixture():
    """The seeded MobileNetV2 (torch.manual_seed(0) + the mirrored constructor = the reference's parameters, checked by
    checksum) through the oracle reproduces the reference's activations and logits to 1e-5 of the layer scale."""
    import torch
    from cpu_vision_amd.mobilenet import MobileNetV2
    from tests._util import oracle_mobilenet_features, randomize_norms
    g = golden("mobilenet_v2")
    torch.manual_seed(0)
    model = MobileNetV2(num_classes=10)
    randomize_norms(model, 7)
    assert abs(float(sum(p.detach().double().sum() for p in model.parameters())) - float(g["net__checksum"][0])) < 1e-9
    acts = oracle_mobilenet_features(ref, model, g["net__x"])
    for i in (0, 1, 3, 7, 14):
        a = acts[i]
        sample = a[:, ::max(1, a.shape[1] // 8), ::2, ::2]
        want = g[f"net__features{i}"]
        assert np.abs(sample - want).max() <= 2e-5 * max(1.0, float(np.abs(want).max())), f"features[{i}]"
    want = g["net__features"]
    assert np.abs(acts[-1] - want).max() <= 2e-5 * float(np.abs(want).max())
    pooled = acts[-1].mean(axis=(2, 3), dtype=np.float64).astype(np.float32)
    fc = model.classifier[1]
    logits = pooled.astype(np.float64) @ fc.weight.detach().numpy().astype(np.float64).T + fc.bias.detach().numpy()
    np.testing.assert_allclose(logits, g["net__logits"], rtol=1e-4, atol=1e-5)


def _deform_case(g, name):
    a = g[f"{name}__args"]
    mask = g[f"{name}__mask"] if f"{name}__mask" in g.files else None
    bias = g[f"{name}__bias"] if int(a[6]) else None
    return (g[f"{name}__x"], g[f"{name}__offset"], g[f"{name}__weight"], bias, tuple(int(v) for v in a[0:2]),
            tuple(int(v) for v in a[2:4]), tuple(int(v) for v in a[4:6]), mask)


def test_oracle_deform_conv2d_vs_reference_expected_fn():
    """8f.4: the oracle against the reference's own test oracle TestDeformConv.expected_fn (float64, test/test_ops.py:
    933-980) at the reference's own tolerance rtol = atol = 1e-5 (test_ops.py:1051-1066), incl. the test's configuration."""
    g = golden("deform_conv2d")
    for name in map(str, g["index"]):
        x, off, w, b, st, pd, dl, mask = _deform_case(g, name)
        got = ref.deform_conv2d(x, off, w, b, st, pd, dl, mask)
        np.testing.assert_allclose(got, g[f"{name}__expected_f64"], rtol=1e-5, atol=1e-5, err_msg=name)


def test_oracle_deform_conv2d_zero_offsets_is_conv2d():
    """Property: zero offsets and no mask reduce deform_conv2d to conv2d -- bit-exactly the oracle's own conv
    (integer sampling positions make every bilinear weight 0 or 1)."""
    rng = np.random.Generator(np.random.Philox(77))
    x = rng.random((2, 4, 9, 8), dtype=np.float32) * 2 - 1
    w = (rng.random((6, 4, 3, 3), dtype=np.float32) - 0.5)
    b = rng.random(6, dtype=np.float32)
    for stride in (1, 2):
        oh, ow = (9 + 2 - 3) // stride + 1, (8 + 2 - 3) // stride + 1
        off = np.zeros((2, 18, oh, ow), np.float32)
        got = ref.deform_conv2d(x, off, w, b, stride, 1, 1, None)
        np.testing.assert_array_equal(got, ref.conv2d_affine_act(x, w, b, None, None, None, stride, 1, 1, 0, None))


# ----------------------------------------------------------------------------- round 2: float64 images
def test_oracle_float64_blur_and_sharpness_vs_reference_fixtures():
    """The reference computes float64 images in float64 (_misc.py:139-155, _color.py:246-275); the fp64 restatement is
    compared with outputs of the reference itself (tests/golden/round2_api.npz) -- to 1e-13 relative, in practice 0."""
    import torch
    from cpu_vision_amd import functional as F, functional_v1 as F1
    g = golden("round2_api")
    for name in g["f64_blur_index"]:
        x = g[name + "__x"]
        ks = [int(v) for v in g[name + "__ks"]]
        sg = [float(v) for v in g[name + "__sigma"]]
        for mod, key in ((F, "__y_v2"), (F1, "__y_v1")):
            if name + key not in g.files:
                continue
            kx = mod._get_gaussian_kernel1d(ks[0], sg[0], torch.float64).numpy()
            ky = mod._get_gaussian_kernel1d(ks[1], sg[1], torch.float64).numpy()
            np.testing.assert_allclose(ref.gaussian_blur_f64(x, kx, ky), g[name + key], rtol=1e-13, atol=1e-15, err_msg=name + key)
    x = g["f64_sharp__x"]
    for f in (0.0, 0.4, 1.0, 2.3):
        np.testing.assert_allclose(ref.adjust_sharpness_f64(x, f), g[f"f64_sharp__y_v2_{f}"], rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(ref.adjust_sharpness_f64(x, f, v1=True), g[f"f64_sharp__y_v1_{f}"], rtol=1e-13, atol=1e-15)


def test_oracle_kernel_side_65_vs_reference_fixture():
    """Kernel sides above 63 (ElasticTransform with sigma >= 8): the single 2-D pass of the reference."""
    from cpu_vision_amd import functional as F
    g = golden("round2_api")
    k = F._get_gaussian_kernel1d(65, 9.0).numpy()
    assert_conv_close(ref.gaussian_blur(g["k65__x"], k, k), g["k65__y"], 1.0, 1.0, what="65x65 f32")
    d = np.abs(ref.gaussian_blur(g["k65u8__x"], k, k).astype(int) - g["k65u8__y"].astype(int))
    assert d.max() <= 1 and (d != 0).mean() <= 2e-3


def test_oracle_deform_vs_reference_native_kernel():
    """Second, reference-OWNED pin of the deform_conv2d restatement (and, with zero offsets, of the dense 3x3 conv of cfg4):
    the reference's CPU kernel csrc/ops/cpu/deform_conv2d_kernel.cpp compiled from its own sources into oracle/_ref/ by
    oracle/build_ref.py (the GPU box uses the prebuilt file).  The restatement reproduces its im2col arithmetic operation
    for operation; the GEMM that follows is ATen's addmm (blocked summation), hence 1e-5 and not bit-equality."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    from oracle.build_ref import build
    so = build()
    if so is None:
        pytest.skip("oracle/_ref not built and /root/reference absent")
    worker = Path(__file__).resolve().parent / "_ref_deform_worker.py"
    r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(res) == 6
    for c in res:
        assert c["max_abs_err"] <= 1e-5 * max(1.0, c["max_abs"]), c
    assert res[4]["dense_conv_max_abs_err"] <= 1e-5 * max(1.0, res[4]["max_abs"]), res[4]
    # non-finite border pixels + samples exactly on the outside boundary: the same outputs are NaN / inf as in the reference's kernel
    nf = res[5]
    assert nf["same_nan_pattern"] and nf["same_inf_pattern"] and nf["finite_outputs"] > 0 and nf["non_finite_outputs"] > 0, nf


def test_uint8_blur_formulation_table_vs_reference_fixtures():
    """The differing-pixel rate of BOTH uint8 formulations against the reference's own outputs, from the
    oracle's statement of each: the single 2-D chain (the library's default) equals the reference on every uint8 fixture pixel;
    the fp32 separable pair + round_() (functional.INTEGER_BLUR_EXACT_2D = False) differs by 1 LSB on at most 1e-4 of them."""
    g = golden("gaussian_blur")
    n = d2 = ds = 0
    for name in map(str, g["index"]):
        ks, sg, dt = _parse_blur_name(name)
        if dt != "u8":
            continue
        x, want = g[f"{name}__x"], g[f"{name}__y_v2"]
        kx, ky = _k1d_pair(ks, sg)
        two, sep = ref.gaussian_blur(x, kx, ky), ref.separable_blur_u8(x, kx, ky)
        assert np.abs(sep.astype(int) - want.astype(int)).max() <= 1, name
        n += want.size
        d2 += int((two != want).sum())
        ds += int((sep != want).sum())
    print(f"uint8 blur fixtures: {n} pixels; 2-D chain differs on {d2}; separable pair differs on {ds}")
    assert n > 90000 and d2 == 0 and 0 < ds <= 1e-4 * n
