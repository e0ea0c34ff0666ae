"""SURVEY.md 8(f).2 -- the ImageClassification preset (transforms/_presets.py:38-64) on the MI355X: resize(bilinear,
antialias) -> center_crop -> convert_image_dtype(float) -> normalize, against the reference's own outputs (golden
fixtures) and the CPU oracle, bit for bit."""
import contextlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cpu_vision_amd as mv  # noqa: E402
from cpu_vision_amd import functional_v1 as F1  # noqa: E402
from cpu_vision_amd.presets import ImageClassification  # noqa: E402
from oracle import ref  # noqa: E402
from tests._util import golden, philox_f32, philox_u8  # noqa: E402

MEAN3, STD3 = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@contextlib.contextmanager
def _two_kernels():
    """The tuning build with MV_RESIZE_TWO_KERNELS: width pass to a workspace, height pass from it."""
    from cpu_vision_amd import _lib
    with _lib.tuning_library():
        os.environ["MV_RESIZE_TWO_KERNELS"] = "1"
        try:
            yield
        finally:
            os.environ.pop("MV_RESIZE_TWO_KERNELS")


def test_resize_crop_preset_vs_reference_fixtures():
    g = golden("resize_preset")
    for name in map(str, g["index"]):
        a = g[f"{name}__args"]
        x, size, max_size, crop = g[f"{name}__x"], [int(v) for v in a[:-2]], (None if a[-2] < 0 else int(a[-2])), int(a[-1])
        xd = dev(x)
        r = F1.resize(xd, size, max_size=max_size)
        assert r.dtype == xd.dtype
        np.testing.assert_array_equal(host(r), g[f"{name}__resized"], err_msg=f"{name} resize")
        np.testing.assert_array_equal(host(F1.center_crop(r, [crop])), g[f"{name}__cropped"], err_msg=f"{name} crop")
        np.testing.assert_array_equal(host(F1.resize_center_crop(xd, size, [crop], max_size=max_size)), g[f"{name}__cropped"],
                                      err_msg=f"{name} fused resize+crop")
        if f"{name}__preset" in g.files:
            mean, std = (MEAN3, STD3) if x.shape[-3] == 3 else ((0.45,), (0.25,))
            got = ImageClassification(crop_size=crop, resize_size=size[0], mean=mean, std=std)(xd)
            assert got.dtype == torch.float32
            np.testing.assert_array_equal(host(got), g[f"{name}__preset"], err_msg=f"{name} preset")


@pytest.mark.parametrize("shape,size", [((3, 37, 53), [24]), ((1, 64, 48), [17, 20]), ((2, 3, 100, 75), [64]), ((3, 33, 90), [50]),
                                        ((1, 1, 1), [5]), ((3, 2, 300), [2]), ((1, 300, 2), [3, 1]), ((3, 19, 23), [19, 40]),
                                        ((3, 19, 23), [40, 23]), ((3, 7, 1200), [7, 100]), ((1, 513, 9), [20, 9]),
                                        # 9 ... 15 taps per window: the fused kernel's batched width pass (csrc/resize.hip, MAXT = 16)
                                        ((3, 200, 330), [40, 55]), ((1, 120, 448), [30, 64]), ((2, 3, 90, 301), [18, 47])])
@pytest.mark.parametrize("dtype", ["u8", "f32"])
def test_resize_bit_exact_vs_oracle(shape, size, dtype):
    """Down- and up-scaling, one axis unchanged (ATen skips that pass), extreme aspect ratios, 1-pixel images."""
    x = philox_u8(8100 + shape[-1], shape) if dtype == "u8" else philox_f32(8101 + shape[-1], shape) * 255
    want = ref.resize(x, size)
    got = host(F1.resize(dev(x), size))
    assert got.shape == want.shape and got.dtype == want.dtype
    np.testing.assert_array_equal(got, want)
    with _two_kernels():  # the width-pass-to-workspace form (tuning build), same bits
        np.testing.assert_array_equal(host(F1.resize(dev(x), size)), want)


@pytest.mark.parametrize("shape,resize,crop", [((3, 375, 500), 256, 224), ((3, 500, 375), 256, 224), ((3, 30, 90), 16, 24),
                                               ((1, 3, 64, 64), 8, 20), ((3, 480, 640), 232, 224), ((3, 100, 100), 342, 299),
                                               ((2, 3, 90, 70), 33, 33)])
def test_preset_bit_exact_vs_oracle(shape, resize, crop):
    """The torchvision presets' configurations (256/224, 232/224, 342/299), crops larger than the resized image
    (zero padding, normalised like the reference), batches, uint8 and float32 inputs."""
    xu = philox_u8(8200 + resize, shape)
    pre = ImageClassification(crop_size=crop, resize_size=resize)
    np.testing.assert_array_equal(host(pre(dev(xu))), ref.image_classification_preset(xu, crop, resize, MEAN3, STD3))
    xf = philox_f32(8201 + crop, shape)
    np.testing.assert_array_equal(host(pre(dev(xf))), ref.image_classification_preset(xf, crop, resize, MEAN3, STD3))
    # step by step through the same kernels = the fused call
    step = F1.center_crop(F1.resize(dev(xu), [resize]), [crop])
    np.testing.assert_array_equal(host(step), ref.center_crop(ref.resize(xu, [resize]), [crop]))
    from cpu_vision_amd import _lib
    assert _lib.last_kernel().startswith("k_resize_fused" if max(shape[-2:]) / resize <= 7 else "k_resize_h")  # one kernel up to scale 7
    with _two_kernels():
        np.testing.assert_array_equal(host(pre(dev(xu))), ref.image_classification_preset(xu, crop, resize, MEAN3, STD3))
        assert _lib.last_kernel() == "k_resize_h"


def test_preset_full_size_batch_properties():
    """A 1080p uint8 batch through the 256/224 preset: a sample of frames against the oracle, and size-independent
    properties on all of them (a constant image normalises to the constant; batch rows are independent)."""
    n = 24
    x = torch.randint(0, 256, (n, 3, 1080, 1920), dtype=torch.uint8, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    pre = ImageClassification(crop_size=224)
    y = pre(x)
    assert y.shape == (n, 3, 224, 224) and y.dtype == torch.float32
    for i in (0, 11, 23):
        np.testing.assert_array_equal(host(y[i]), ref.image_classification_preset(host(x[i]), 224, 256, MEAN3, STD3))
        assert torch.equal(pre(x[i]), y[i])
    c = torch.full((2, 3, 720, 1280), 128, dtype=torch.uint8, device="cuda")
    want = (np.float32(128) / np.float32(255) - np.asarray(MEAN3, np.float32)) / np.asarray(STD3, np.float32)
    got = host(pre(c))
    for ch in range(3):
        assert np.all(got[:, ch] == want[ch])


def test_preset_feeds_the_first_conv_like_the_reference_pipeline():
    """preset -> Conv2d(3,64,3,pad=1)+ReLU: the two stages of BASELINE's serving path back to back, against the
    oracle's composition."""
    from cpu_vision_amd import functional as F
    xu = philox_u8(8300, (2, 3, 96, 128))
    w = ((philox_f32(8301, (64, 3, 3, 3)) - 0.5) * 0.3).astype(np.float32)
    b = (philox_f32(8302, (64,)) - 0.5).astype(np.float32)
    y = F.conv2d_bias_relu(ImageClassification(crop_size=56, resize_size=64)(dev(xu)), dev(w), dev(b))
    want = ref.conv3x3_bias_relu(ref.image_classification_preset(xu, 56, 64, MEAN3, STD3), w, b)
    np.testing.assert_array_equal(host(y), want)


def test_resize_errors_and_abi_status():
    x = torch.zeros((3, 8, 8), dtype=torch.uint8, device="cuda")
    with pytest.raises(NotImplementedError):
        F1.resize(x.to(torch.int16), [4])
    with pytest.raises(ValueError, match="std evaluated to zero"):
        ImageClassification(crop_size=4, resize_size=4, std=(1.0, 0.0, 1.0))(x)
    lib = mv.load_library()
    y = torch.empty((3, 4, 4), dtype=torch.uint8, device="cuda")
    # one kernel, the width pass's temporary in LDS: no workspace; scale factors above 47 (96-tap windows) keep the two kernels
    assert lib.mv_resize_workspace_bytes(3, 8, 8, 4, 4, 0, 0, 4, 4) == 0
    assert lib.mv_resize_bilinear_aa_u8(x.data_ptr(), y.data_ptr(), 3, 8, 8, 4, 4, 0, 0, 4, 4, None, 0, None) == 0
    big = torch.zeros((1, 8, 500), dtype=torch.uint8, device="cuda")
    yb = torch.empty((1, 8, 8), dtype=torch.uint8, device="cuda")
    assert lib.mv_resize_workspace_bytes(1, 8, 500, 8, 8, 0, 0, 8, 8) == 1 * 8 * 8 * 4
    assert lib.mv_resize_bilinear_aa_u8(big.data_ptr(), yb.data_ptr(), 1, 8, 500, 8, 8, 0, 0, 8, 8, None, 0, None) == -1
    assert b"workspace" in lib.mv_last_error()
    assert lib.mv_resize_bilinear_aa_u8(x.data_ptr(), y.data_ptr(), 3, 8, 8, 0, 4, 0, 0, 4, 4, None, 0, None) == -1
    e = torch.empty((0, 3, 8, 8), dtype=torch.uint8, device="cuda")
    assert ImageClassification(crop_size=4, resize_size=4)(e).shape == (0, 3, 4, 4)


def test_preset_on_pil_input_vs_reference_fixtures():
    """transforms/_presets.py:54-61 takes PIL images: the reference then resizes / crops with PIL itself and runs the tensor half
    (pil_to_tensor -> convert_image_dtype -> normalize) on the result.  Same split here -- PIL's two calls on the host, the tensor
    half on the MI355X -- against outputs generated by the reference (tests/golden/round3_preset_pil.npz): bit-exact when the
    installed Pillow is the one the fixtures were made with (its resampling is PIL's own arithmetic), 2 uint8 steps otherwise."""
    import PIL
    import PIL.Image
    from cpu_vision_amd.presets import ImageClassification
    g = golden("round3_preset_pil")
    same_pillow = str(g["pil_version"][0]) == PIL.__version__
    for name in map(str, g["index"]):
        arr = g[f"{name}__x"]
        img = PIL.Image.fromarray(arr, mode="L" if arr.ndim == 2 else "RGB")
        crop, size = (int(v) for v in g[f"{name}__cfg"])
        pre = ImageClassification(crop_size=crop, resize_size=size, mean=tuple(g[f"{name}__mean"]), std=tuple(g[f"{name}__std"]))
        got = pre(img)
        assert got.is_cuda and got.dtype == torch.float32 and tuple(got.shape) == tuple(g[f"{name}__y"].shape)
        if same_pillow:
            np.testing.assert_array_equal(got.cpu().numpy(), g[f"{name}__y"], err_msg=name)
        else:
            assert np.abs(got.cpu().numpy() - g[f"{name}__y"]).max() <= 2.0 / 255 / float(np.min(g[f"{name}__std"])) + 1e-6
    with pytest.raises(TypeError, match="Tensor or a PIL Image"):
        ImageClassification(crop_size=8)(np.zeros((3, 8, 8), np.uint8))
