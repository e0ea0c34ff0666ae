"""Separately allocated frames in ONE launch (mv_*_v, `-m gpu`): what a DataLoader hands to Transform.forward
(transforms/v2/_transform.py:40-55) is a list of frames allocated one by one.  Every result must equal the single-frame
entry point's -- and through it the oracle -- bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cpu_vision_amd as mv  # noqa: E402
from cpu_vision_amd import _lib, functional as F, transforms, tv_tensors  # noqa: E402
from oracle import ref  # noqa: E402
from tests._util import philox_f32, philox_u8  # noqa: E402


def scattered(arrays):
    """Device copies of `arrays`, each in its own allocation with other allocations in between."""
    out, keep = [], []
    for i, a in enumerate(arrays):
        keep.append(torch.empty(1000 + 512 * (i % 5), dtype=torch.uint8, device="cuda"))
        out.append(torch.from_numpy(np.ascontiguousarray(a)).cuda())
    return out, keep


def k1d(k, s):
    return F._get_gaussian_kernel1d(k, s).numpy()


@pytest.mark.parametrize("shape,n", [((3, 70, 130), 7), ((1, 33, 1024), 3), ((3, 32, 32), 200), ((3, 16, 48), 230), ((2, 3, 40, 64), 5)])
@pytest.mark.parametrize("ks", [(3, 3), (5, 5), (7, 3), (9, 9)])
def test_gaussian_blur_frame_list_through_the_c_abi(shape, n, ks):
    lib = mv.load_library()
    planes = int(np.prod(shape[:-2]))
    h, w = shape[-2:]
    sg = [0.5 + ks[0] / 5.0, 0.4 + ks[1] / 4.0]
    tx, ty = k1d(ks[0], sg[0]), k1d(ks[1], sg[1])
    xf = [philox_f32(6000 + i, shape) for i in range(n)]
    xs, keep = scattered(xf)
    ys = [torch.empty_like(x) for x in xs]
    _lib.check(lib.mv_gaussian_blur_f32_v(_lib.pointer_table(xs), _lib.pointer_table(ys), n, planes, h, w, _lib.taps(tx), ks[0],
                                          _lib.taps(ty), ks[1], None))
    torch.cuda.synchronize()
    for i in sorted({0, 1, n // 2, n - 1}):  # > 112 frames: several launches
        np.testing.assert_array_equal(ys[i].cpu().numpy(), ref.gaussian_blur(xf[i], tx, ty), err_msg=f"f32 frame {i}")
    single = torch.empty_like(xs[-1])
    _lib.check(lib.mv_gaussian_blur_f32(xs[-1].data_ptr(), single.data_ptr(), planes, h, w, _lib.taps(tx), ks[0], _lib.taps(ty), ks[1], None))
    assert torch.equal(single, ys[-1])
    xu = [philox_u8(6500 + i, shape) for i in range(n)]
    xus, keep2 = scattered(xu)
    yus = [torch.empty_like(x) for x in xus]
    _lib.check(lib.mv_gaussian_blur_u8_v(_lib.pointer_table(xus), _lib.pointer_table(yus), n, planes, h, w, _lib.taps(tx), ks[0],
                                         _lib.taps(ty), ks[1], None))
    torch.cuda.synchronize()
    for i in sorted({0, n // 3, n - 1}):
        np.testing.assert_array_equal(yus[i].cpu().numpy(), ref.gaussian_blur(xu[i], tx, ty), err_msg=f"u8 frame {i}")


@pytest.mark.parametrize("shape,n", [((3, 70, 132), 7), ((1, 40, 1024), 3), ((3, 32, 32), 130), ((2, 3, 41, 64), 5), ((3, 30, 45), 4)])
@pytest.mark.parametrize("ks", [(5, 5), (7, 7), (7, 3), (9, 9), (23, 23), (3, 15)])
def test_separable_blur_frame_list_through_the_c_abi(shape, n, ks):
    """mv_separable_blur_f32_v / _u8_v: the register-streaming (k_sepfast, k_dwk_u8<separable>), row-streaming (k_sepstream)
    and LDS (k_separable) kernels with per-frame base pointers -- equal to the contiguous entry points and the oracle."""
    lib = mv.load_library()
    planes = int(np.prod(shape[:-2]))
    h, w = shape[-2:]
    if ks[0] // 2 >= w or ks[1] // 2 >= h:
        pytest.skip("reflect padding must be smaller than the image")
    sg = [0.5 + ks[0] / 5.0, 0.4 + ks[1] / 4.0]
    tx, ty = k1d(ks[0], sg[0]), k1d(ks[1], sg[1])
    xf = [philox_f32(7600 + i, shape) for i in range(n)]
    xs, keep = scattered(xf)
    ys = [torch.empty_like(x) for x in xs]
    _lib.check(lib.mv_separable_blur_f32_v(_lib.pointer_table(xs), _lib.pointer_table(ys), n, planes, h, w, _lib.taps(tx), ks[0],
                                           _lib.taps(ty), ks[1], None))
    torch.cuda.synchronize()
    for i in sorted({0, n // 2, n - 1}):
        np.testing.assert_array_equal(ys[i].cpu().numpy(), ref.separable_blur(xf[i], tx, ty), err_msg=f"f32 frame {i}")
    u8_ok = w >= (8 if max(ks) > 7 else 16)
    xu = [philox_u8(7700 + i, shape) for i in range(n)]
    xus, keep2 = scattered(xu)
    yus = [torch.empty_like(x) for x in xus]
    rc = lib.mv_separable_blur_u8_v(_lib.pointer_table(xus), _lib.pointer_table(yus), n, planes, h, w, _lib.taps(tx), ks[0],
                                    _lib.taps(ty), ks[1], None)
    assert (rc == 0) == u8_ok, lib.mv_last_error()
    if u8_ok:
        torch.cuda.synchronize()
        for i in sorted({0, n // 3, n - 1}):
            np.testing.assert_array_equal(yus[i].cpu().numpy(), ref.separable_blur_u8(xu[i], tx, ty), err_msg=f"u8 frame {i}")
    # the functional entry picks the formulation gaussian_blur_image runs for this size: same bits as the per-frame calls
    for frames in (xs, xus):
        outs = F.gaussian_blur_frames(frames, list(ks), sg)
        for f, o in zip(frames[:3], outs[:3]):
            assert torch.equal(o, F.gaussian_blur_image(f, list(ks), sg))


@pytest.mark.parametrize("shape,n", [((3, 37, 260), 6), ((1, 20, 48), 150), ((3, 2, 9), 3)])
@pytest.mark.parametrize("v1", [0, 1])
def test_sharpness_frame_list_through_the_c_abi(shape, n, v1):
    lib = mv.load_library()
    planes, h, w = shape[0], shape[1], shape[2]
    xf = [philox_f32(6800 + i, shape) for i in range(n)]
    xu = [philox_u8(6900 + i, shape) for i in range(n)]
    for f in (0.0, 0.4, 1.8):
        xs, keep = scattered(xf)
        ys = [torch.empty_like(x) for x in xs]
        _lib.check(lib.mv_sharpness_f32_v(_lib.pointer_table(xs), _lib.pointer_table(ys), n, planes, h, w, f, v1, 1.0, 0, None))
        xus, keep2 = scattered(xu)
        yus = [torch.empty_like(x) for x in xus]
        _lib.check(lib.mv_sharpness_u8_v(_lib.pointer_table(xus), _lib.pointer_table(yus), n, planes, h, w, f, v1, None))
        torch.cuda.synchronize()
        for i in sorted({0, n // 2, n - 1}):
            np.testing.assert_array_equal(ys[i].cpu().numpy(), ref.adjust_sharpness(xf[i], f, v1=bool(v1)), err_msg=f"f32 frame {i} f={f}")
            np.testing.assert_array_equal(yus[i].cpu().numpy(), ref.adjust_sharpness(xu[i], f, v1=bool(v1)), err_msg=f"u8 frame {i} f={f}")


def test_frame_list_argument_checks_and_unaligned_frames():
    lib = mv.load_library()
    t = _lib.taps(k1d(3, 0.8))
    x = torch.rand(3, 20, 36, device="cuda")
    y = torch.empty_like(x)
    assert lib.mv_gaussian_blur_f32_v(None, None, 0, 3, 20, 36, t, 3, t, 3, None) == 0          # empty list
    assert lib.mv_gaussian_blur_f32_v(None, None, 2, 3, 20, 36, t, 3, t, 3, None) == -1         # null table
    assert lib.mv_gaussian_blur_f32_v(_lib.pointer_table([x, x]), _lib.pointer_table([y, x]), 2, 3, 20, 36, t, 3, t, 3, None) == -1
    assert b"alias" in lib.mv_last_error()
    null = (C.c_void_p * 2)(x.data_ptr(), None)
    assert lib.mv_gaussian_blur_f32_v(null, _lib.pointer_table([y, y]), 2, 3, 20, 36, t, 3, t, 3, None) == -1
    # frames at addresses that are not 16-byte aligned: served frame by frame, same bits
    big = torch.rand(3 * 20 * 36 * 2 + 8, device="cuda")
    xs = [big[1:1 + 3 * 20 * 36].view(3, 20, 36), big[3 * 20 * 36 + 3:2 * 3 * 20 * 36 + 3].view(3, 20, 36)]
    assert xs[0].data_ptr() % 16 != 0
    ys = [torch.empty(3, 20, 36, device="cuda") for _ in xs]
    _lib.check(lib.mv_gaussian_blur_f32_v(_lib.pointer_table(xs), _lib.pointer_table(ys), 2, 3, 20, 36, t, 3, t, 3, None))
    for xi, yi in zip(xs, ys):
        assert torch.equal(yi, F.gaussian_blur_image(xi.contiguous(), [3, 3]))


def test_functional_and_transform_route_lists_through_one_launch():
    frames = [torch.rand(3, 90, 160, device="cuda") for _ in range(9)]
    outs = F.gaussian_blur_frames(frames, [3, 3], [0.8, 0.8])
    assert _lib.last_kernel().startswith("k_dwtile<f32,3x3") and len(outs) == 9
    for f, o in zip(frames, outs):
        assert torch.equal(o, F.gaussian_blur_image(f, [3, 3], [0.8, 0.8]))
    u8 = [(f * 255).to(torch.uint8) for f in frames]
    for f, o in zip(u8, F.adjust_sharpness_frames(u8, 1.7)):
        assert torch.equal(o, F.adjust_sharpness_image(f, 1.7))
    for f, o in zip(u8, F.gaussian_blur_frames(u8, [5, 5])):  # uint8 5x5: the reference's single 2-D pass by default, one launch as well
        assert torch.equal(o, F.gaussian_blur_image(f, [5, 5]))
    # (the last call is gaussian_blur_image on one 90 x 160 frame: pair + tie check + fix-up, the same integers as the 2-D pass of the
    # frames entry point)
    assert _lib.last_kernel() in ("k_dwk_u8<5x5,2d>", "k_dwk_u8<separable,ties>+k_u8_tie_fixup")
    F.INTEGER_BLUR_EXACT_2D = False  # the opt-in separable pair takes the same route
    try:
        for f, o in zip(u8, F.gaussian_blur_frames(u8, [5, 5])):
            assert torch.equal(o, F.gaussian_blur_image(f, [5, 5]))
        assert _lib.last_kernel() == "k_dwk_u8<5x5,separable>"
    finally:
        F.INTEGER_BLUR_EXACT_2D = True
    mixed = frames[:2] + [torch.rand(3, 50, 60, device="cuda")]
    for f, o in zip(mixed, F.gaussian_blur_frames(mixed, [3, 3])):
        assert torch.equal(o, F.gaussian_blur_image(f, [3, 3]))
    # Transform.forward on a sample holding several images: same parameters for all, one launch
    torch.manual_seed(11)
    sample = {"views": [tv_tensors.Image(f) for f in frames[:4]], "label": 3, "mask": tv_tensors.Mask(torch.zeros(90, 160, device="cuda"))}
    out = transforms.GaussianBlur(3, sigma=(0.3, 1.5))(sample)
    torch.manual_seed(11)
    sigma = torch.empty(1).uniform_(0.3, 1.5).item()
    assert out["label"] == 3 and out["mask"] is sample["mask"]
    for f, o in zip(frames[:4], out["views"]):
        assert type(o) is tv_tensors.Image
        assert torch.equal(o.as_subclass(torch.Tensor), F.gaussian_blur_image(f, [3, 3], [sigma, sigma]))
    sharp = transforms.RandomAdjustSharpness(2.0, p=1.0)([tv_tensors.Image(f) for f in u8[:3]])
    for f, o in zip(u8[:3], sharp):
        assert torch.equal(o.as_subclass(torch.Tensor), F.adjust_sharpness_image(f, 2.0))
