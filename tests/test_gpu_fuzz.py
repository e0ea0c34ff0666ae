"""Seeded random-shape sweep of every entry point against the oracle (`-m gpu`): ragged widths, tiny images, odd batch
dims, every border mode -- the shapes nobody writes by hand.  Everything must be bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cpu_vision_amd import functional as F, functional_v1 as F1  # noqa: E402
from oracle import ref  # noqa: E402
from tests._util import oracle_conv3x3, philox_f32, philox_u8  # noqa: E402

BORD = {"reflect": ref.BORDER_REFLECT, "zero": ref.BORDER_ZERO, "valid": ref.BORDER_VALID}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def _cases(seed, n):
    rng = np.random.default_rng(seed)
    widths = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 100, 255, 256, 257, 260, 272, 511, 512, 1000, 1024, 1040]
    for i in range(n):
        w = int(rng.choice(widths))
        h = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 17, 31, 33, 64, 70]))
        lead = [(), (1,), (3,), (2, 1), (2, 3)][int(rng.integers(0, 5))]
        yield i, tuple(lead) + (h, w), rng


@pytest.mark.parametrize("chunk", range(4))
def test_fuzz_depthwise_and_gaussian(chunk):
    for i, shape, rng in _cases(100 + chunk, 25):
        h, w = shape[-2:]
        ky = int(rng.choice([1, 3, 5, 7, 9, 11]))
        kx = int(rng.choice([1, 3, 5, 7, 9, 11]))
        border = ["reflect", "zero", "valid"][int(rng.integers(0, 3))]
        if border == "reflect" and (ky // 2 >= h or kx // 2 >= w):
            border = "zero"
        if border == "valid" and (ky > h or kx > w):
            border = "zero"
        wt = (rng.random((ky, kx), dtype=np.float32) - 0.5)
        x = philox_f32(9000 + chunk * 100 + i, shape) * 2 - 1
        got = host(F.depthwise_conv2d(dev(x), torch.from_numpy(wt), border))
        np.testing.assert_array_equal(got, ref.depthwise_conv2d(x, wt, BORD[border]), err_msg=f"f32 {shape} {ky}x{kx} {border}")
        xu = philox_u8(9500 + chunk * 100 + i, shape)
        wn = np.abs(wt) / max(np.abs(wt).sum(), 1e-6)
        gotu = host(F.depthwise_conv2d(dev(xu), torch.from_numpy(wn), border))
        want = np.rint(ref.depthwise_conv2d(xu.astype(np.float32), wn, BORD[border])).astype(np.uint8)
        np.testing.assert_array_equal(gotu, want, err_msg=f"u8 {shape} {ky}x{kx} {border}")


@pytest.mark.parametrize("chunk", range(3))
def test_fuzz_separable_sobel_sharpness(chunk):
    for i, shape, rng in _cases(200 + chunk, 20):
        h, w = shape[-2:]
        k = int(rng.choice([1, 3, 5, 7, 9, 13, 23, 41]))
        x = philox_f32(9900 + chunk * 100 + i, shape)
        if k // 2 < h and k // 2 < w:
            sg = float(rng.uniform(0.3, 4.0))
            t = F._get_gaussian_kernel1d(k, sg).numpy()
            np.testing.assert_array_equal(host(F.separable_gaussian_blur(dev(x), [k, k], [sg, sg])), ref.separable_blur(x, t, t),
                                          err_msg=f"separable {shape} k={k}")
            if h >= 2 and w >= 2:
                gx, gy = F.gaussian_sobel(dev(x), [k, k], [sg, sg])
                ogx, ogy = ref.gaussian_sobel(x, t, t)
                np.testing.assert_array_equal(host(gx), ogx, err_msg=f"gaussian_sobel gx {shape} k={k}")
                np.testing.assert_array_equal(host(gy), ogy, err_msg=f"gaussian_sobel gy {shape} k={k}")
        if h >= 2 and w >= 2:
            gx, gy = F.sobel(dev(x), "reflect")
            ogx, ogy = ref.sobel(x, ref.BORDER_REFLECT)
            np.testing.assert_array_equal(host(gx), ogx)
            np.testing.assert_array_equal(host(gy), ogy)
        # sharpness needs (..., C in {1,3}, H, W)
        c = int(rng.choice([1, 3]))
        shp = (2, c, h, w)
        f = float(rng.choice([0.0, 0.3, 0.77, 1.0, 1.9, 4.0]))
        xu = philox_u8(9950 + chunk * 100 + i, shp)
        np.testing.assert_array_equal(host(F.adjust_sharpness(dev(xu), f)), ref.adjust_sharpness(xu, f), err_msg=f"sharp u8 {shp} f={f}")
        np.testing.assert_array_equal(host(F1.adjust_sharpness(dev(xu), f)), ref.adjust_sharpness(xu, f, v1=True))
        xf = philox_f32(9960 + chunk * 100 + i, shp)
        np.testing.assert_array_equal(host(F.adjust_sharpness(dev(xf), f)), ref.adjust_sharpness(xf, f), err_msg=f"sharp f32 {shp} f={f}")


def test_fuzz_conv_linear_pool():
    rng = np.random.default_rng(300)
    for i in range(16):
        n = int(rng.integers(1, 4))
        cin = int(rng.choice([1, 2, 3, 4, 5, 8, 13, 16, 24]))
        cout = int(rng.choice([1, 7, 32, 33, 64, 65, 130]))
        h = int(rng.choice([1, 2, 5, 14, 17, 28]))
        w = int(rng.choice([1, 3, 4, 14, 30, 56, 100]))
        x = philox_f32(9700 + i, (n, cin, h, w)) - 0.5
        wt = (philox_f32(9720 + i, (cout, cin, 3, 3)) - 0.5) * 0.5
        b = philox_f32(9740 + i, (cout,)) - 0.5
        relu = bool(i & 1)
        got = F.conv2d_bias_relu(dev(x), dev(wt), dev(b), relu=relu)
        want = oracle_conv3x3(ref, x, wt, b, relu=relu)
        np.testing.assert_array_equal(host(got), want, err_msg=f"conv n={n} cin={cin} cout={cout} {h}x{w}")
        if h >= 2 and w >= 2:
            np.testing.assert_array_equal(host(F.max_pool2d_2x2(got)), ref.maxpool2x2(want))
        k, m = int(rng.choice([1, 3, 31, 32, 33, 100, 257])), int(rng.choice([1, 4, 5, 31, 128, 129]))
        xl = philox_f32(9760 + i, (n * 7, k)) - 0.5
        wl = (philox_f32(9780 + i, (m, k)) - 0.5) * 0.3
        np.testing.assert_array_equal(host(F.linear_bias_relu(dev(xl), dev(wl), dev(b[:m] if m <= cout else np.resize(b, m)), relu=relu)),
                                      ref.linear_bias_relu(xl, wl, b[:m] if m <= cout else np.resize(b, m), relu=relu))


@pytest.mark.parametrize("chunk", range(3))
def test_fuzz_resize_and_preset(chunk):
    """Random sizes through resize / resize+center_crop / the whole preset (uint8 and float32), bit-exact vs the oracle
    (the one torch corner the oracle does not share -- 1-pixel-wide output with a height change -- is a torch bug)."""
    from cpu_vision_amd.presets import ImageClassification
    rng = np.random.default_rng(300 + chunk)
    for i in range(14):
        c = int(rng.choice([1, 3]))
        h, w = int(rng.integers(1, 260)), int(rng.integers(1, 260))
        lead = [(c,), (2, c)][int(rng.integers(0, 2))]
        shape = tuple(lead) + (h, w)
        xu = philox_u8(30000 + chunk * 100 + i, shape)
        xf = philox_f32(30050 + chunk * 100 + i, shape) * 255
        size = [int(rng.integers(1, 200))] if rng.integers(0, 2) else [int(rng.integers(1, 200)), int(rng.integers(1, 200))]
        for x in (xu, xf):
            np.testing.assert_array_equal(host(F1.resize(dev(x), size)), ref.resize(x, size), err_msg=f"resize {shape} -> {size}")
        crop = int(rng.integers(1, 120))
        got = host(F1.resize_center_crop(dev(xu), size, [crop]))
        np.testing.assert_array_equal(got, ref.center_crop(ref.resize(xu, size), [crop]), err_msg=f"resize+crop {shape} {size} {crop}")
        if len(size) == 1:
            mean, std = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)) if c == 3 else ((0.4,), (0.3,))
            pre = ImageClassification(crop_size=crop, resize_size=size[0], mean=mean, std=std)
            np.testing.assert_array_equal(host(pre(dev(xu))), ref.image_classification_preset(xu, crop, size[0], mean, std),
                                          err_msg=f"preset {shape} {size} {crop}")


@pytest.mark.parametrize("chunk", range(3))
def test_fuzz_conv_norm_act_and_deform(chunk):
    """Random Conv2dNormActivation blocks (all three kernels, both norm folds, every exact activation, residual) and random
    deform_conv2d / generic conv2d geometries, bit-exact vs the oracle."""
    from cpu_vision_amd import ops
    rng = np.random.default_rng(400 + chunk)
    acts = [None, "relu", "relu6", "hardswish"]
    for i in range(12):
        seed = 40000 + chunk * 100 + i
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        kind = int(rng.integers(0, 3))
        stride = 1 if kind == 2 else int(rng.integers(1, 3))
        affine = [None, "mul_add", "fma"][int(rng.integers(0, 3))]
        act = acts[int(rng.integers(0, 4))]
        if kind == 0:
            cin, cout, groups, wshape = int(rng.integers(1, 5)), int(rng.integers(1, 40)), 1, None
            wshape = (cout, cin, 3, 3)
        elif kind == 1:
            cin = cout = int(rng.integers(1, 70))
            groups, wshape = cin, (cin, 1, 3, 3)
        else:
            cin, cout, groups = int(rng.integers(1, 150)), int(rng.integers(1, 150)), 1
            wshape = (cout, cin, 1, 1)
        x = philox_f32(seed, (n, cin, h, w)) * 4 - 2
        wt = (philox_f32(seed + 1, wshape) - 0.5) * 0.8
        a, b = philox_f32(seed + 2, (cout,)) + 0.5, philox_f32(seed + 3, (cout,)) - 0.5
        bias = philox_f32(seed + 4, (cout,)) - 0.5 if rng.integers(0, 2) else None
        oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
        res = philox_f32(seed + 5, (n, cout, oh, ow)) - 0.5 if rng.integers(0, 2) else None
        code = {None: 0, "mul_add": 1, "fma": 2}[affine]
        got = host(F.conv_norm_act(dev(x), dev(wt), None if bias is None else dev(bias), None if affine is None else dev(a),
                                   None if affine is None else dev(b), None if res is None else dev(res), stride=stride, groups=groups,
                                   affine=affine, activation=act))
        want = ref.conv2d_affine_act(x, wt, bias, a if affine else None, b if affine else None, res, stride, 0 if kind == 2 else 1, groups,
                                     code, act)
        np.testing.assert_array_equal(got, want, err_msg=f"kind {kind} {x.shape} -> {cout} s{stride} {affine} {act}")
    for i in range(8):
        seed = 41000 + chunk * 100 + i
        groups = int(rng.choice([1, 2]))
        og = int(rng.choice([1, 2]))
        cin, cout = groups * og * int(rng.integers(1, 5)), groups * int(rng.integers(1, 6))
        kh, kw = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        st, pd, dl = (int(rng.integers(1, 3)), int(rng.integers(1, 3))), (int(rng.integers(0, 3)), int(rng.integers(0, 3))), \
            (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
        h, w = int(rng.integers(6, 20)), int(rng.integers(6, 20))
        oh = (h + 2 * pd[0] - (dl[0] * (kh - 1) + 1)) // st[0] + 1
        ow = (w + 2 * pd[1] - (dl[1] * (kw - 1) + 1)) // st[1] + 1
        if oh < 1 or ow < 1:
            continue
        n = int(rng.integers(1, 4))
        x = philox_f32(seed, (n, cin, h, w)) * 2 - 1
        off = ((philox_f32(seed + 1, (n, og * 2 * kh * kw, oh, ow)) - 0.5) * 6).astype(np.float32)
        mask = philox_f32(seed + 2, (n, og * kh * kw, oh, ow)) if rng.integers(0, 2) else None
        wt = philox_f32(seed + 3, (cout, cin // groups, kh, kw)) - 0.5
        bias = philox_f32(seed + 4, (cout,)) - 0.5 if rng.integers(0, 2) else None
        got = host(ops.deform_conv2d(dev(x), dev(off), dev(wt), None if bias is None else dev(bias), stride=st, padding=pd, dilation=dl,
                                     mask=None if mask is None else dev(mask)))
        np.testing.assert_array_equal(got, ref.deform_conv2d(x, off, wt, bias, st, pd, dl, mask), err_msg=f"deform {x.shape} k{kh}x{kw} {st} {pd} {dl}")
        got = host(F.conv2d_bias_act(dev(x), dev(wt), None if bias is None else dev(bias), stride=st, padding=pd, dilation=dl, groups=groups))
        np.testing.assert_array_equal(got, ref.deform_conv2d(x, np.zeros_like(off[:, :2 * kh * kw]), wt, bias, st, pd, dl, None),
                                      err_msg=f"conv2d {x.shape} k{kh}x{kw}")
