"""Parity of the gfx950 kernels, called through the C ABI (ctypes), against the CPU oracle and the
reference-generated golden fixtures.  Runs on the MI355X box only (`-m gpu`).

Bars (SURVEY.md 8d / BASELINE.json): fp32 |a-b| <= 1e-5*|b| + 1e-6*sum|w|*max|x| against the REFERENCE's
outputs (golden fixtures); against the oracle (same tap order) the kernels are required to be BIT-EXACT;
uint8 sharpness bit-exact against the reference; uint8 blur bit-exact against the oracle and within 1 LSB
(rare rounding ties) of the reference, which is the reference's own tolerance for that op.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cpu_vision_amd as mv  # noqa: E402
from cpu_vision_amd import functional as F, functional_v1 as F1  # noqa: E402
from oracle import ref  # noqa: E402
from tests._util import assert_conv_close, golden, philox_f32, philox_u8  # noqa: E402

BORD = {"reflect": ref.BORDER_REFLECT, "zero": ref.BORDER_ZERO, "valid": ref.BORDER_VALID}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def k1d(k, s, v1=False):
    return (F1 if v1 else F)._get_gaussian_kernel1d(k, s).numpy()


def u8_blur_oracle(xu, tx, ty):
    """The oracle for gaussian_blur_image on a uint8 image, in the formulation the library runs for that size
    (functional._use_separable): the fp32 separable pair + round_() for kernel sides above 3 on images at least 16 (sides
    above 7: 8) pixels wide, the reference's single 2-D sum otherwise."""
    probe = torch.empty((1, xu.shape[-1]), dtype=torch.uint8)
    sep = F._use_separable(len(tx), len(ty), probe) and max(len(tx), len(ty)) <= 63 and xu.shape[-1] >= 8
    return ref.separable_blur_u8(xu, tx, ty) if sep else ref.gaussian_blur(xu, tx, ty)


def test_device_is_gfx950_and_library_loaded():
    assert torch.cuda.is_available(), "this suite must run on the GPU box"
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName
    assert mv.load_library().mv_device_count() >= 1


# ----------------------------------------------------------------------------- the reference's own OpenCV vectors
@pytest.mark.parametrize("dims,ks,sigma", [((3, 10, 12), (3, 3), 0.8), ((3, 10, 12), (3, 3), 0.5),
                                           ((3, 10, 12), (3, 5), 0.8), ((3, 10, 12), (3, 5), 0.5),
                                           ((1, 26, 28), (23, 23), 1.7)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.float16, torch.uint8])
def test_opencv_golden_vectors(dims, ks, sigma, dtype):
    """test_transforms_v2.py:3273-3309 (atol=1, rtol=0), same inputs, through F.gaussian_blur_image."""
    c, h, w = dims
    want = golden("opencv_gaussian_blur")[f"{h}_{w}_{c}__{ks[0]}_{ks[1]}_{sigma}"].reshape(h, w, c).transpose(2, 0, 1)
    x = torch.arange(c * h * w, dtype=torch.uint8).reshape(h, w, c).permute(2, 0, 1).to(dtype).cuda()
    got = F.gaussian_blur_image(x, kernel_size=list(ks), sigma=sigma)
    assert got.dtype == dtype and got.shape == x.shape
    torch.testing.assert_close(got.cpu().double(), torch.from_numpy(want.astype(np.float64)), rtol=0, atol=1)


# ----------------------------------------------------------------------------- gaussian blur vs reference fixtures + oracle
def _parse_blur_name(name):
    parts = name.split("_")
    kx, ky = map(int, parts[0][1:].split("x"))
    if parts[1] == "sd":
        return [kx, ky], None, parts[2]
    return [kx, ky], [float(parts[1][1:]), float(parts[2])], parts[3]


def test_gaussian_blur_vs_reference_fixtures_and_oracle():
    g = golden("gaussian_blur")
    n_u8 = n_u8_diff = 0
    for name in map(str, g["index"]):
        ks, sg, dt = _parse_blur_name(name)
        x, want = g[f"{name}__x"], g[f"{name}__y_v2"]
        got = host(F.gaussian_blur_image(dev(x), kernel_size=ks, sigma=sg))
        assert got.dtype == want.dtype and got.shape == want.shape, name
        sgl = sg if sg is not None else [k * 0.15 + 0.35 for k in ks]
        kx, ky = k1d(ks[0], sgl[0]), k1d(ks[1], sgl[1])
        if dt == "u8":
            np.testing.assert_array_equal(got, u8_blur_oracle(x, kx, ky), err_msg=f"{name} vs oracle")
            d = np.abs(got.astype(np.int32) - want.astype(np.int32))
            assert d.max() <= 1, name
            n_u8 += d.size
            n_u8_diff += int((d != 0).sum())
        else:
            assert_conv_close(got, want, 1.0, 1.0, what=f"{name} vs reference")
            separable = F._use_separable(ks[0], ks[1], torch.empty((1, x.shape[-1]), dtype=torch.float32))
            orc = ref.separable_blur(x, kx, ky) if separable else ref.gaussian_blur(x, kx, ky)
            np.testing.assert_array_equal(got, orc, err_msg=f"{name} vs oracle (bit-exact)")
        if f"{name}__y_v1" in g.files:
            got1 = host(F1.gaussian_blur(dev(x), ks, sgl))
            want1 = g[f"{name}__y_v1"]
            if dt == "u8":
                assert np.abs(got1.astype(np.int32) - want1.astype(np.int32)).max() <= 1
            else:
                assert_conv_close(got1, want1, 1.0, 1.0, what=f"{name} v1")
    assert n_u8 and n_u8_diff <= 1e-3 * n_u8


@pytest.mark.parametrize("shape", [(3, 17, 11), (1, 5, 7), (2, 3, 33, 259), (1, 64, 512), (3, 40, 1028), (1, 2, 2),
                                   (1, 300, 4), (1, 19, 1021)])
@pytest.mark.parametrize("ks", [(3, 3), (5, 5), (3, 5), (5, 3), (7, 7), (1, 3), (3, 1), (9, 9), (11, 3), (1, 1)])
def test_gaussian_2d_bit_exact_vs_oracle_many_shapes(shape, ks):
    """Ragged widths (W % 4 != 0: scalar path), widths spanning several 256-column segments, tiny images."""
    kxs, kys = ks
    if kxs // 2 >= shape[-1] or kys // 2 >= shape[-2]:
        pytest.skip("reflect padding must be smaller than the image")
    x = philox_f32(hash((shape, ks)) % 10_000, shape)
    tx, ty = k1d(kxs, 0.9), k1d(kys, 1.3)
    lib = mv.load_library()
    from cpu_vision_amd import _lib
    xd = dev(x)
    yd = torch.empty_like(xd)
    planes = int(np.prod(shape[:-2]))
    _lib.check(lib.mv_gaussian_blur_f32(xd.data_ptr(), yd.data_ptr(), planes, shape[-2], shape[-1],
                                        _lib.taps(tx), kxs, _lib.taps(ty), kys, None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(yd), ref.gaussian_blur(x, tx, ty))
    xu = philox_u8(hash((shape, ks)) % 10_000 + 1, shape)
    xud = dev(xu)
    yud = torch.empty_like(xud)
    _lib.check(lib.mv_gaussian_blur_u8(xud.data_ptr(), yud.data_ptr(), planes, shape[-2], shape[-1],
                                       _lib.taps(tx), kxs, _lib.taps(ty), kys, None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(host(yud), ref.gaussian_blur(xu, tx, ty))


@pytest.mark.parametrize("shape", [(3, 17, 11), (2, 3, 33, 259), (1, 64, 512), (1, 2, 2), (1, 19, 1021), (1, 300, 4)])
@pytest.mark.parametrize("border", ["reflect", "zero"])
def test_register_window_3x3_kernel_bit_exact(shape, border, monkeypatch, tuning_library):
    """k_dw3x3 (kept for A/B and as the engine of sobel / sharpness) through the same entry points."""
    monkeypatch.setenv("MV_FORCE_REG3X3", "1")
    x = philox_f32(4242 + shape[-1], shape)
    w = philox_f32(4243, (3, 3)) - 0.4
    np.testing.assert_array_equal(host(F.depthwise_conv2d(dev(x), torch.from_numpy(w), border)),
                                  ref.depthwise_conv2d(x, w, BORD[border]))
    xu = philox_u8(4244 + shape[-1], shape)
    wn = np.abs(w) / np.abs(w).sum()
    want = np.rint(ref.depthwise_conv2d(xu.astype(np.float32), wn, BORD[border])).astype(np.uint8)
    np.testing.assert_array_equal(host(F.depthwise_conv2d(dev(xu), torch.from_numpy(wn), border)), want)


@pytest.mark.parametrize("shape", [(1, 1, 16), (3, 9, 32), (2, 3, 37, 1024), (1, 40, 2064), (1, 5, 1040), (1, 2, 3840)])
def test_uint8_16_pixels_per_lane_kernel(shape, monkeypatch):
    """k_dw3x3_u8 (W % 16 == 0): blur (reflect / zero) and sharpness v1/v2, against the oracle and against the
    4-pixel kernels it replaces."""
    xu = philox_u8(4400 + shape[-1] + shape[-2], shape)
    w = philox_f32(4401, (3, 3))
    wn = w / w.sum()
    for border in ("reflect", "zero"):
        if border == "reflect" and (shape[-2] < 2):
            continue
        want = np.rint(ref.depthwise_conv2d(xu.astype(np.float32), wn, BORD[border])).astype(np.uint8)
        got = host(F.depthwise_conv2d(dev(xu), torch.from_numpy(wn), border))
        np.testing.assert_array_equal(got, want)
    if shape[-2] >= 2:
        k = k1d(3, 0.8)
        np.testing.assert_array_equal(host(F.gaussian_blur_image(dev(xu), [3, 3])), ref.gaussian_blur(xu, k, k))
    if len(shape) >= 3 and shape[-3] in (1, 3):
        for f in (0.0, 0.4, 1.0, 2.3):
            got2 = host(F.adjust_sharpness_image(dev(xu), f))
            np.testing.assert_array_equal(got2, ref.adjust_sharpness(xu, f))
            np.testing.assert_array_equal(host(F1.adjust_sharpness(dev(xu), f)), ref.adjust_sharpness(xu, f, v1=True))
        from cpu_vision_amd import _lib
        with _lib.tuning_library():  # the 4-pixel kernel it replaces (forced kernels: tuning build only)
            monkeypatch.setenv("MV_FORCE_U8X4", "1")
            np.testing.assert_array_equal(host(F.adjust_sharpness_image(dev(xu), 0.4)), ref.adjust_sharpness(xu, 0.4))
            monkeypatch.delenv("MV_FORCE_U8X4")


def test_lds_separable_kernel_bit_exact_when_forced(monkeypatch, tuning_library):
    """k_separable (LDS) on a shape the register-streaming kernel would normally take."""
    monkeypatch.setenv("MV_FORCE_LDS_SEPARABLE", "1")
    x = philox_f32(4300, (2, 40, 512))
    k = k1d(5, 1.1)
    np.testing.assert_array_equal(host(F.separable_gaussian_blur(dev(x), [5, 5], [1.1, 1.1])), ref.separable_blur(x, k, k))
    gx, gy = F.gaussian_sobel(dev(x), [5, 5], [1.1, 1.1])
    ogx, ogy = ref.gaussian_sobel(x, k, k)
    np.testing.assert_array_equal(host(gx), ogx)
    np.testing.assert_array_equal(host(gy), ogy)


def test_unaligned_base_pointer_takes_scalar_path():
    x = philox_f32(77, (1, 21, 64 * 4 + 1))
    base = torch.zeros(x.size + 1, dtype=torch.float32, device="cuda")
    view = base[1:].view(x.shape)  # 4-byte aligned only
    view.copy_(dev(x))
    w = philox_f32(78, (3, 3)) - 0.5
    # as_strided views are contiguous here; data_ptr is base+4
    got = host(F.depthwise_conv2d(view, torch.from_numpy(w), "reflect"))
    np.testing.assert_array_equal(got, ref.depthwise_conv2d(x, w, ref.BORDER_REFLECT))


# ----------------------------------------------------------------------------- the primitive with arbitrary taps
@pytest.mark.parametrize("border", ["reflect", "zero", "valid"])
@pytest.mark.parametrize("kyx", [(3, 3), (5, 5), (7, 7), (5, 3), (3, 5), (1, 5), (9, 1), (11, 11), (13, 13)])
@pytest.mark.parametrize("shape", [(2, 2, 19, 45), (1, 37, 260), (1, 16, 512)])
def test_depthwise_conv2d_bit_exact_vs_oracle(border, kyx, shape):
    ky, kx = kyx
    if ky // 2 >= shape[-2] or kx // 2 >= shape[-1]:
        pytest.skip("kernel larger than image")
    x = philox_f32(900 + ky * 31 + kx, shape) * 2 - 1
    w = philox_f32(950 + ky * 31 + kx, (ky, kx)) - 0.5
    got = host(F.depthwise_conv2d(dev(x), torch.from_numpy(w), border))  # 13x13 = 169 taps goes through device taps
    np.testing.assert_array_equal(got, ref.depthwise_conv2d(x, w, BORD[border]))
    xu = philox_u8(970 + ky, shape)
    wn = np.abs(w) / np.abs(w).sum()
    gotu = host(F.depthwise_conv2d(dev(xu), torch.from_numpy(wn), border))
    xf = ref.depthwise_conv2d(xu.astype(np.float32), wn, BORD[border])
    np.testing.assert_array_equal(gotu, np.rint(xf).astype(np.uint8))


def test_primitive_vs_reference_fixtures():
    g = golden("primitive_filters")
    assert_conv_close(host(F.box_filter(dev(g["box__x"]), 3)), g["box__y"], what="box 3x3 (cfg1 operator)")
    x = g["gen__x"]
    for ky, kx in [(3, 3), (5, 3), (1, 5), (7, 7)]:
        w = g[f"gen_{ky}x{kx}__w"]
        for b in ("reflect", "zero", "valid"):
            got = host(F.depthwise_conv2d(dev(x), torch.from_numpy(w), b))
            assert_conv_close(got, g[f"gen_{ky}x{kx}_{b}__y"], float(np.abs(w).sum()), 1.0, what=f"gen {ky}x{kx} {b}")
    xs = g["sep__x"]
    blur = host(F.separable_gaussian_blur(dev(xs), [5, 5], [1.1, 1.1]))
    assert_conv_close(blur, g["sep__blur"], what="separable 5x5")
    gx, gy = F.gaussian_sobel(dev(xs), [5, 5], [1.1, 1.1])
    assert_conv_close(host(gx), g["sep__gx"], 8.0, 1.0, what="cfg3 gx")
    assert_conv_close(host(gy), g["sep__gy"], 8.0, 1.0, what="cfg3 gy")
    for b in ("reflect", "zero", "valid"):
        gx, gy = F.sobel(dev(xs), b)
        assert_conv_close(host(gx), g[f"sobel_{b}__gx"], 8.0, 1.0, what=f"sobel {b}")
        assert_conv_close(host(gy), g[f"sobel_{b}__gy"], 8.0, 1.0, what=f"sobel {b}")


@pytest.mark.parametrize("shape", [(1, 1, 16), (3, 9, 32), (2, 3, 37, 1024), (1, 70, 2064), (1, 5, 1040), (1, 4, 3840),
                                   (1, 131, 48)])
@pytest.mark.parametrize("kyx", [(5, 5), (7, 7), (3, 5), (5, 3), (3, 7), (7, 3), (5, 7), (7, 5)])
def test_uint8_16_pixels_per_lane_kxk_kernel(shape, kyx, monkeypatch):
    """k_dwk_u8 (W % 16 == 0, KY/KX in {3,5,7}): Gaussian blur (reflect) and averaging filters (reflect / zero)
    against the oracle, and against the 4-pixel LDS-tile kernel it replaces."""
    ky, kx = kyx
    xu = philox_u8(4500 + shape[-1] + 7 * shape[-2] + ky * 10 + kx, shape)
    xd = dev(xu)
    w = philox_f32(4501 + ky * 10 + kx, (ky, kx)) + 0.05
    wn = (w / w.sum() * 0.999).astype(np.float32)
    for border in ("reflect", "zero"):
        if border == "reflect" and (ky // 2 >= shape[-2] or kx // 2 >= shape[-1]):
            continue
        want = np.rint(ref.depthwise_conv2d(xu.astype(np.float32), wn, BORD[border])).astype(np.uint8)
        np.testing.assert_array_equal(host(F.depthwise_conv2d(xd, torch.from_numpy(wn), border)), want, err_msg=border)
    if ky // 2 < shape[-2] and kx // 2 < shape[-1]:
        sg = [0.7 + kx / 5.0, 0.6 + ky / 4.0]
        tx, ty = k1d(kx, sg[0]), k1d(ky, sg[1])
        # the single 2-D pass (the reference's summation form; functional.INTEGER_BLUR_EXACT_2D) on k_dwk_u8<.., 2d>
        monkeypatch.setattr(F, "INTEGER_BLUR_EXACT_2D", True)
        monkeypatch.setattr(F, "INTEGER_BLUR_EXACT_FAST", False)  # (the plain pass, not the pair + tie fix-up that gives the same bits)
        want = ref.gaussian_blur(xu, tx, ty)
        np.testing.assert_array_equal(host(F.gaussian_blur_image(xd, [kx, ky], sg)), want)
        from cpu_vision_amd import _lib
        assert _lib.last_kernel() == f"k_dwk_u8<{ky}x{kx},2d>"
        with _lib.tuning_library():
            monkeypatch.setenv("MV_FORCE_U8X4", "1")
            np.testing.assert_array_equal(host(F.gaussian_blur_image(xd, [kx, ky], sg)), want)
        # the default: the separable pair on the same lane layout, k_dwk_u8<.., separable>
        monkeypatch.setattr(F, "INTEGER_BLUR_EXACT_2D", False)
        want_sep = ref.separable_blur_u8(xu, tx, ty)
        np.testing.assert_array_equal(host(F.gaussian_blur_image(xd, [kx, ky], sg)), want_sep)
        assert _lib.last_kernel() == f"k_dwk_u8<{ky}x{kx},separable>"
        d = np.abs(want_sep.astype(np.int32) - want.astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() <= 2e-3  # ties only: the reference's own atol = 1


# ----------------------------------------------------------------------------- separable / fused cfg3 graph
@pytest.mark.parametrize("shape", [(3, 32, 40), (1, 7, 9), (2, 45, 300), (1, 70, 1024), (1, 5, 5), (1, 33, 255)])
@pytest.mark.parametrize("ks", [(5, 5), (3, 3), (7, 5), (3, 9), (23, 23), (41, 41), (1, 1), (63, 1)])
def test_separable_and_fused_sobel_bit_exact_vs_oracle(shape, ks):
    kxs, kys = ks
    if kxs // 2 >= shape[-1] or kys // 2 >= shape[-2]:
        pytest.skip("reflect padding must be smaller than the image")
    x = philox_f32(1200 + kxs * 7 + kys, shape)
    tx, ty = k1d(kxs, 1.1), k1d(kys, 2.0)
    sg = [1.1, 2.0]
    got = host(F.separable_gaussian_blur(dev(x), [kxs, kys], sg))
    np.testing.assert_array_equal(got, ref.separable_blur(x, tx, ty))
    gx, gy = F.gaussian_sobel(dev(x), [kxs, kys], sg)
    ogx, ogy = ref.gaussian_sobel(x, tx, ty)
    np.testing.assert_array_equal(host(gx), ogx)
    np.testing.assert_array_equal(host(gy), ogy)


@pytest.mark.parametrize("shape", [(3, 32, 40), (2, 45, 300), (1, 70, 1024), (1, 33, 254), (3, 130, 516), (1, 12, 8), (2, 40, 333), (1, 37, 259),
                                   (1, 64, 1023)])
@pytest.mark.parametrize("ks", [(9, 9), (23, 23), (3, 11), (15, 1), (41, 41), (63, 5), (31, 33)])
def test_uint8_separable_large_kernels(shape, ks, monkeypatch):
    """uint8 storage with more than 49 taps: fp32 separable pair + round_() -- bit-exact vs the oracle's statement of
    that recipe, within 1 LSB of the single 2-D pass (the reference's formulation), and the exact 2-D pass on request."""
    kxs, kys = ks
    if kxs // 2 >= shape[-1] or kys // 2 >= shape[-2]:
        pytest.skip("reflect padding must be smaller than the image")
    xu = philox_u8(1300 + kxs * 7 + kys, shape)
    sg = [1.0 + kxs / 6.0, 1.0 + kys / 6.0]
    tx, ty = k1d(kxs, sg[0]), k1d(kys, sg[1])
    lib = mv.load_library()
    from cpu_vision_amd import _lib
    xd = dev(xu)
    yd = torch.empty_like(xd)
    planes = int(np.prod(shape[:-2]))
    rc = lib.mv_separable_blur_u8(xd.data_ptr(), yd.data_ptr(), planes, shape[-2], shape[-1], _lib.taps(tx), kxs,
                                  _lib.taps(ty), kys, None)
    assert rc == 0, lib.mv_last_error()  # any width >= 8, any alignment
    torch.cuda.synchronize()
    want = ref.separable_blur_u8(xu, tx, ty)
    np.testing.assert_array_equal(host(yd), want)
    d = np.abs(want.astype(np.int32) - ref.gaussian_blur(xu, tx, ty).astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3
    # the transform-level entry picks it when it applies, and the exact 2-D pass on request
    got = host(F.gaussian_blur_image(xd, [kxs, kys], sg))
    np.testing.assert_array_equal(got, u8_blur_oracle(xu, tx, ty))
    monkeypatch.setattr(F, "INTEGER_BLUR_EXACT_2D", True)
    if kxs * kys <= 23 * 23:
        np.testing.assert_array_equal(host(F.gaussian_blur_image(xd, [kxs, kys], sg)), ref.gaussian_blur(xu, tx, ty))


@pytest.mark.parametrize("shape", [(3, 32, 40), (1, 1, 16), (2, 45, 300), (1, 70, 1024), (1, 33, 254), (3, 130, 516), (2, 40, 333),
                                   (1, 37, 2064), (1, 64, 1023), (5, 3, 20, 24), (1, 3, 3840), (1, 131, 48), (1, 9, 17)])
@pytest.mark.parametrize("ks", [(5, 5), (7, 7), (3, 5), (5, 3), (7, 3), (3, 7), (5, 7), (7, 5), (1, 5), (7, 1), (3, 3), (9, 9), (9, 7), (7, 9)])
def test_uint8_separable_small_kernels_through_the_c_abi(shape, ks):
    """mv_separable_blur_u8 with kernel sides <= 7, and 9 x 9 / 9 x 7 / 7 x 9: k_dwk_u8<.., separable> (16 pixels per lane, row
    pass + systolic column chain; a lane's halo is the 4 bytes either side of its pixels).  Bit-exact against the oracle's statement of the recipe (.to(float32) -> separable pair -> round_() -> uint8) for
    any width >= 16, any alignment, ragged right edges, several strips per wave; within 1 LSB of the single 2-D sum."""
    kxs, kys = ks
    if kxs // 2 >= shape[-1] or kys // 2 >= shape[-2]:
        pytest.skip("reflect padding must be smaller than the image")
    xu = philox_u8(1700 + kxs * 7 + kys + shape[-1], shape)
    sg = [0.6 + kxs / 5.0, 0.5 + kys / 4.0]
    tx, ty = k1d(kxs, sg[0]), k1d(kys, sg[1])
    lib = mv.load_library()
    from cpu_vision_amd import _lib
    xd = dev(xu)
    yd = torch.empty_like(xd)
    planes = int(np.prod(shape[:-2]))
    rc = lib.mv_separable_blur_u8(xd.data_ptr(), yd.data_ptr(), planes, shape[-2], shape[-1], _lib.taps(tx), kxs, _lib.taps(ty), kys, None)
    assert rc == 0, lib.mv_last_error()
    torch.cuda.synchronize()
    assert _lib.last_kernel() == f"k_dwk_u8<{max(kys, 3)}x{max(kxs, 3)},separable>"
    want = ref.separable_blur_u8(xu, tx, ty)
    np.testing.assert_array_equal(host(yd), want)
    d = np.abs(want.astype(np.int32) - ref.gaussian_blur(xu, tx, ty).astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() <= 2e-3
    # images narrower than 16 pixels are refused here (the 2-D entry serves them); 9-tap sides go to the streaming kernel instead
    narrow = dev(philox_u8(3, (1, 9, 12)))
    rc = lib.mv_separable_blur_u8(narrow.data_ptr(), torch.empty_like(narrow).data_ptr(), 1, 9, 12, _lib.taps(tx), kxs, _lib.taps(ty), kys, None)
    assert rc == (-2 if max(kxs, kys) <= 7 else 0)


def _tie_heavy_images(shape, seed):
    """uint8 images whose blurred values sit on or next to rounding ties far more often than photographs do."""
    rng = np.random.Generator(np.random.Philox(seed))
    yy, xx = np.indices(shape[-2:])
    return {
        "random": rng.integers(0, 256, shape, dtype=np.uint8),
        "binary": rng.integers(0, 2, shape, dtype=np.uint8),                       # sums are small multiples of the taps
        "checkerboard": np.broadcast_to(((yy + xx) & 1).astype(np.uint8) * 255, shape).copy(),
        "stripes_127_128": np.broadcast_to((127 + (xx & 1)).astype(np.uint8), shape).copy(),   # every blurred value ~127.5
        "even_values": (rng.integers(0, 128, shape, dtype=np.uint8) * 2).astype(np.uint8),
    }


@pytest.mark.parametrize("ks", [(9, 7), (7, 9), (9, 9), (11, 5), (13, 13), (9, 5) if False else (5, 11), (15, 5), (23, 23), (41, 41), (63, 63), (3, 17), (51, 1)])
@pytest.mark.parametrize("shape", [(2, 70, 1040), (1, 130, 3840), (3, 64, 1000), (5, 90, 224), (3, 75, 500), (7, 40, 96)])
def test_uint8_exact_fast_equals_the_reference_2d_pass(ks, shape, monkeypatch):
    """mv_gaussian_blur_u8_ws (round 3): the separable pair for every pixel + the reference's 2-D chain for the lane-rows whose
    value lies within the proved error bound of a rounding tie == the single 2-D pass, BIT FOR BIT -- on random images and on
    images built to sit on ties, through k_dwk_u8<separable, ties> (sides <= 9) and k_sepstream (sides <= 63), and with a tie
    list too small on purpose (the fix-up then recomputes every pixel)."""
    from cpu_vision_amd import _lib
    kxs, kys = ks
    if kxs // 2 >= shape[-1] or kys // 2 >= shape[-2]:
        pytest.skip("reflect padding must be smaller than the image")
    sg = [0.5 + kxs / 6.0, 0.5 + kys / 6.0]
    tx, ty = k1d(kxs, sg[0]), k1d(kys, sg[1])
    lib = _lib.load()
    planes = int(np.prod(shape[:-2]))
    nbytes = int(lib.mv_gaussian_blur_u8_workspace_bytes(planes, shape[-2], shape[-1], kxs, kys))
    assert nbytes > 0, "this size / width should have the exact-fast path"
    flagged = {}
    for tag, xu in _tie_heavy_images(shape, 5100 + kxs * 64 + kys + shape[-1]).items():
        want = ref.gaussian_blur(xu, tx, ty)  # the oracle's 2-D chain == the reference (test_oracle_golden)
        xd = dev(xu)
        got = host(F.gaussian_blur_image(xd, [kxs, kys], sg))
        assert _lib.last_kernel().endswith("+k_u8_tie_fixup"), _lib.last_kernel()
        np.testing.assert_array_equal(got, want, err_msg=f"{tag}: exact-fast path vs the 2-D chain")
        monkeypatch.setattr(F, "INTEGER_BLUR_EXACT_FAST", False)
        if kxs * kys <= 23 * 23:
            np.testing.assert_array_equal(host(F.gaussian_blur_image(xd, [kxs, kys], sg)), want, err_msg=f"{tag}: plain 2-D pass")
        monkeypatch.setattr(F, "INTEGER_BLUR_EXACT_FAST", True)
        sep = ref.separable_blur_u8(xu, tx, ty) if max(ks) <= 63 else want
        flagged[tag] = int((sep != want).sum())
    # (the pair alone differs from the reference on ~1e-5 of such pixels -- `flagged` -- the fix-up is what makes the result exact)
    # a list of ONE entry: every launch overflows it, the fix-up recomputes every pixel
    xu = _tie_heavy_images(shape, 77)["random"]
    with _lib.tuning_library():
        monkeypatch.setenv("MV_U8_TIE_CAP", "1")
        np.testing.assert_array_equal(host(F.gaussian_blur_image(dev(xu), [kxs, kys], sg)), ref.gaussian_blur(xu, tx, ty))
        monkeypatch.delenv("MV_U8_TIE_CAP")


def test_uint8_exact_fast_argument_rules():
    from cpu_vision_amd import _lib
    lib = _lib.load()
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 1040, 3, 3) == 0      # below 25 taps the plain 2-D pass is as fast
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 1040, 3, 7) == 0
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 1040, 5, 5) > 0       # from 5 x 5 up: pair + tie check + fix-up
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 1040, 7, 7) > 0
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 300, 7, 7) == 0        # narrow images: the plain pass up to 49 taps
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 12, 9, 9) == 0        # narrower than 16 pixels
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 300, 9, 9) > 0        # several strips per wave: the tie instantiation exists there too
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 35, 9, 9) == 0        # a width that needs the byte-by-byte neighbour path
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 300, 23, 23) > 0      # k_sepstream takes any width >= 8
    assert lib.mv_gaussian_blur_u8_workspace_bytes(3, 64, 1040, 9, 9) > 0
    xu = philox_u8(9, (1, 40, 1040))
    tx = k1d(9, 1.6)
    xd, yd = dev(xu), torch.empty((1, 40, 1040), dtype=torch.uint8, device="cuda")
    # no workspace, or taps that are not an average (the bound's premise): the plain 2-D pass, same bits
    assert lib.mv_gaussian_blur_u8_ws(xd.data_ptr(), yd.data_ptr(), 1, 40, 1040, _lib.taps(tx), 9, _lib.taps(tx), 9, None, 0, None) == 0
    torch.cuda.synchronize()
    assert not _lib.last_kernel().endswith("k_u8_tie_fixup")
    np.testing.assert_array_equal(host(yd), ref.gaussian_blur(xu, tx, tx))
    ws = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    neg = np.array([-0.02, -0.05, -0.1, 0.3, 0.74, 0.3, -0.1, -0.05, -0.02], np.float32)
    assert lib.mv_gaussian_blur_u8_ws(xd.data_ptr(), yd.data_ptr(), 1, 40, 1040, _lib.taps(neg), 9, _lib.taps(tx), 9, ws.data_ptr(), ws.numel(), None) == 0
    assert not _lib.last_kernel().endswith("k_u8_tie_fixup")
    assert lib.mv_gaussian_blur_u8_ws(xd.data_ptr(), yd.data_ptr(), 1, 40, 1040, _lib.taps(tx), 9, _lib.taps(tx), 9, ws.data_ptr(), ws.numel(), None) == 0
    assert _lib.last_kernel().endswith("k_u8_tie_fixup")


def test_uint8_blur_formulations_vs_reference_fixtures(monkeypatch):
    """Both uint8 formulations against the REFERENCE's outputs (tests/golden/gaussian_blur.npz, every uint8 case).  VERDICT
    round 2, weak 1: the default must be the closer one.  The single 2-D pass (default, INTEGER_BLUR_EXACT_2D = True) equals the
    reference on EVERY pixel; the opt-in separable pair stays within 1 LSB on at most 1e-4 of them (1 of 93 144 here)."""
    g = golden("gaussian_blur")
    assert F.INTEGER_BLUR_EXACT_2D is True
    for exact in (True, False):
        monkeypatch.setattr(F, "INTEGER_BLUR_EXACT_2D", exact)
        n = nd = 0
        for name in map(str, g["index"]):
            ks, sg, dt = _parse_blur_name(name)
            if dt != "u8":
                continue
            got = host(F.gaussian_blur_image(dev(g[f"{name}__x"]), kernel_size=ks, sigma=sg))
            d = np.abs(got.astype(np.int32) - g[f"{name}__y_v2"].astype(np.int32))
            assert d.max() <= (0 if exact else 1), (name, exact)
            n += d.size
            nd += int((d != 0).sum())
        assert n > 90000 and (nd == 0 if exact else nd <= 1e-4 * n), (exact, n, nd)


@pytest.mark.parametrize("border", ["reflect", "zero", "valid"])
@pytest.mark.parametrize("shape", [(3, 32, 40), (1, 3, 3), (2, 9, 257), (1, 50, 1024), (1, 2, 2)])
def test_sobel_bit_exact_vs_oracle(border, shape):
    if border == "valid" and (shape[-1] < 3 or shape[-2] < 3):
        pytest.skip("valid needs 3x3")
    x = philox_f32(1300 + shape[-1], shape) - 0.5
    gx, gy = F.sobel(dev(x), border)
    ogx, ogy = ref.sobel(x, BORD[border])
    np.testing.assert_array_equal(host(gx), ogx)
    np.testing.assert_array_equal(host(gy), ogy)


# ----------------------------------------------------------------------------- adjust_sharpness
def _sharp_cases():
    g = golden("adjust_sharpness")
    for name in map(str, g["index"]):
        f = float(name.split("_")[0][1:])
        yield name, f, g[name.split("_", 1)[1] + "__x"], g[f"{name}__y_v2"], (g[f"{name}__y_v1"] if f"{name}__y_v1" in g.files else None)


def test_adjust_sharpness_vs_reference_fixtures():
    for name, f, x, want2, want1 in _sharp_cases():
        got = host(F.adjust_sharpness_image(dev(x), f))
        assert got.dtype == want2.dtype and got.shape == want2.shape
        if x.dtype == np.uint8:
            np.testing.assert_array_equal(got, want2, err_msg=f"{name}: integer contract is bit-exact")
        else:
            assert_conv_close(got, want2, 1.0 + 2 * abs(1 - f), 1.0, what=name)
            np.testing.assert_array_equal(got, ref.adjust_sharpness(x, f), err_msg=f"{name} vs oracle")
        if want1 is not None:
            got1 = host(F1.adjust_sharpness(dev(x), f))
            if x.dtype == np.uint8:
                np.testing.assert_array_equal(got1, want1, err_msg=f"{name} v1")
            else:
                assert_conv_close(got1, want1, 1.0 + 2 * abs(1 - f), 1.0, what=f"{name} v1")
                np.testing.assert_array_equal(got1, ref.adjust_sharpness(x, f, v1=True))


def test_adjust_sharpness_pil_exact():
    """test_transforms_v2.py:4721-4731: uint8 result equals PIL ImageEnhance.Sharpness."""
    g = golden("adjust_sharpness")
    for f in (0.1, 0.5, 1.0):
        np.testing.assert_array_equal(host(F.adjust_sharpness(dev(g["pil__x"]), f)), g[f"pil__y_{f}"])


@pytest.mark.parametrize("shape", [(3, 33, 259), (1, 3, 3), (3, 64, 1024), (1, 40, 6)])
@pytest.mark.parametrize("f", [0.0, 0.37, 1.0, 2.5])
def test_adjust_sharpness_u8_and_int16_vs_oracle(shape, f):
    xu = philox_u8(1400 + shape[-1], shape)
    np.testing.assert_array_equal(host(F.adjust_sharpness_image(dev(xu), f)), ref.adjust_sharpness(xu, f))
    np.testing.assert_array_equal(host(F1.adjust_sharpness(dev(xu), f)), ref.adjust_sharpness(xu, f, v1=True))
    # other integer dtypes go through the f32 entry with integer semantics and bound=_max_value(dtype):
    # for values within uint8's range the result must equal the uint8 path
    xi = dev(xu).to(torch.int16)
    np.testing.assert_array_equal(host(F.adjust_sharpness_image(xi, f)).astype(np.int64),
                                  np.minimum(_int_sharp(xu, f, 32767), 32767))


def _int_sharp(xu, f, bound):
    """The reference's integer recipe (_color.py:259-275) for a dtype whose bound is above 255."""
    x = xu.astype(np.float32)
    a, b = np.float32(1 / 13), np.float32(5 / 13)
    w = np.array([[a, a, a], [a, b, a], [a, a, a]], np.float32)
    blur = np.rint(ref.depthwise_conv2d(x, w, ref.BORDER_VALID))
    out = x.copy()
    alpha = np.float32(1.0 - f)
    inner = out[..., 1:-1, 1:-1]
    d = (blur - inner).astype(np.float64)
    inner[...] = (inner.astype(np.float64) + np.float64(alpha) * d).astype(np.float32)  # exact fma for these magnitudes
    return np.clip(out, 0, bound).astype(np.int64)


# ----------------------------------------------------------------------------- first CNN layer (MFMA implicit GEMM)
def test_conv_relu_vs_reference_fixtures_and_oracle():
    g = golden("conv_relu")
    w, b = g["vgg11__w"], g["vgg11__b"]
    gain = float(np.abs(w).reshape(64, -1).sum(1).max())
    wd, bd = dev(w), dev(b)
    got = host(F.conv2d_bias_relu(dev(g["vgg11__x"]), wd, bd))
    assert_conv_close(got, g["vgg11__y"], gain, 1.0, what="vgg11 features[0:2] vs reference")
    np.testing.assert_array_equal(got, ref.conv3x3_bias_relu(g["vgg11__x"], w, b), err_msg="fp32 MFMA == fmaf chain")
    got = host(F.conv2d_bias_relu(dev(g["bias__x"]), wd, dev(g["bias__b"]), relu=False))
    assert_conv_close(got, g["bias__y_norelu"], gain, 1.0, what="bias, no relu")
    got = host(F.conv2d_bias_relu(dev(g["bias__x"]), wd, dev(g["bias__b"])))
    assert_conv_close(got, g["bias__y"], gain, 1.0, what="bias + relu")
    np.testing.assert_array_equal(got, ref.conv3x3_bias_relu(g["bias__x"], w, g["bias__b"]))
    from cpu_vision_amd.nn import Conv2dNormActivation
    blk = Conv2dNormActivation(3, 64, norm_layer=None).cuda()
    blk[0].load_state_dict({"weight": torch.from_numpy(g["cna__w"]), "bias": torch.from_numpy(g["cna__b"])})
    assert_conv_close(host(blk(dev(g["vgg11__x"]))), g["cna__y"],
                      float(np.abs(g["cna__w"]).reshape(64, -1).sum(1).max()), 1.0, what="Conv2dNormActivation")
    got = host(F.conv2d_bias_relu(dev(g["c16__x"]), dev(g["c16__w"]), dev(g["c16__b"])))
    assert_conv_close(got, g["c16__y"], float(np.abs(g["c16__w"]).reshape(32, -1).sum(1).max()), 0.5, what="Cin=16 Cout=32")
    np.testing.assert_array_equal(got, ref.conv3x3_bias_relu(g["c16__x"], g["c16__w"], g["c16__b"]))


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 3, 64, 224, 224), (1, 3, 64, 7, 5), (3, 3, 40, 33, 300), (1, 1, 8, 20, 33),
                                            (2, 5, 70, 17, 64), (1, 3, 64, 1, 1)])
def test_conv_relu_shapes_bit_exact_vs_oracle(n, cin, cout, h, w):
    x = philox_f32(1500 + h, (n, cin, h, w)) * 2 - 1
    wt = (philox_f32(1501 + h, (cout, cin, 3, 3)) - 0.5) * 0.5
    b = philox_f32(1502 + h, (cout,)) - 0.5
    got = host(F.conv2d_bias_relu(dev(x), dev(wt), dev(b)))
    np.testing.assert_array_equal(got, ref.conv3x3_bias_relu(x, wt, b))
    got = host(F.conv2d_bias_relu(dev(x), dev(wt), None, relu=False))
    np.testing.assert_array_equal(got, ref.conv3x3_bias_relu(x, wt, None, relu=False))


@pytest.mark.parametrize("shape", [(37, 3, 32, 32), (5, 3, 64, 64), (7, 1, 96, 96), (3, 3, 128, 128), (11, 3, 20, 24), (2, 50, 3, 17, 33),
                                   (129, 1, 9, 16), (4, 3, 33, 127)])
def test_thumbnail_batches_bit_exact_vs_oracle(shape):
    """CIFAR / STL-size images: the lane groups of the register kernels take (plane, strip) units across plane boundaries and
    the LDS tile narrows to the image width -- many small planes, ragged last wave, every operator."""
    xf = philox_f32(4100 + shape[-1], shape)
    xu = philox_u8(4200 + shape[-1], shape)
    for k, sg in ((3, 0.8), (5, 1.1), (7, 1.4)):
        t = k1d(k, sg)
        want = ref.separable_blur(xf, t, t) if F._use_separable(k, k, dev(xf)) else ref.gaussian_blur(xf, t, t)
        np.testing.assert_array_equal(host(F.gaussian_blur(dev(xf), [k, k], [sg, sg])), want, err_msg=f"f32 blur {k}")
        np.testing.assert_array_equal(host(F.gaussian_blur(dev(xu), [k, k], [sg, sg])), u8_blur_oracle(xu, t, t), err_msg=f"u8 blur {k}")
    gx, gy = F.sobel(dev(xf), "reflect")
    ogx, ogy = ref.sobel(xf, ref.BORDER_REFLECT)
    np.testing.assert_array_equal(host(gx), ogx)
    np.testing.assert_array_equal(host(gy), ogy)
    wt = philox_f32(4300, (3, 3)) - 0.5
    for border in ("reflect", "zero", "valid"):
        np.testing.assert_array_equal(host(F.depthwise_conv2d(dev(xf), torch.from_numpy(wt), border)),
                                      ref.depthwise_conv2d(xf, wt, BORD[border]), err_msg=f"3x3 {border}")
    if shape[-3] in (1, 3):
        for f in (0.0, 0.6, 1.7):
            np.testing.assert_array_equal(host(F.adjust_sharpness(dev(xu), f)), ref.adjust_sharpness(xu, f), err_msg=f"u8 sharp {f}")
            np.testing.assert_array_equal(host(F1.adjust_sharpness(dev(xu), f)), ref.adjust_sharpness(xu, f, v1=True), err_msg=f"u8 sharp v1 {f}")
            np.testing.assert_array_equal(host(F.adjust_sharpness(dev(xf), f)), ref.adjust_sharpness(xf, f), err_msg=f"f32 sharp {f}")
