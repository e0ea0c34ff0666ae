"""Host-side behaviour that needs no GPU: argument validation / error text (same as the reference's tests
test_transforms_v2.py:3177-3194, 3227-3257, 4716-4719), the kernel registry, transform parameter sampling,
taps, and the loud failure when asked to run off-device."""
from pathlib import Path

import numpy as np
import pytest
import torch

import cpu_vision_amd as mv
from cpu_vision_amd import _registry, functional as F, functional_v1 as F1, transforms, tv_tensors
from tests._util import golden


def img(*shape, dtype=torch.float32):
    return torch.zeros(shape or (3, 17, 11), dtype=dtype)


def test_gaussian_blur_argument_errors():
    image = img()
    with pytest.raises(ValueError, match="kernel_size is a sequence its length should be 2"):
        F.gaussian_blur_image(image, kernel_size=[1, 2, 3])
    for ks in [2, -1]:
        with pytest.raises(ValueError, match="kernel_size should have odd and positive integers"):
            F.gaussian_blur_image(image, kernel_size=ks)
    with pytest.raises(ValueError, match="sigma is a sequence, its length should be 2"):
        F.gaussian_blur_image(image, kernel_size=1, sigma=[1, 2, 3])
    with pytest.raises(TypeError, match="sigma should be either float or sequence of floats"):
        F.gaussian_blur_image(image, kernel_size=1, sigma=object())
    with pytest.raises(ValueError, match="sigma should have positive values"):
        F.gaussian_blur_image(image, kernel_size=1, sigma=-1)
    with pytest.raises(TypeError, match="kernel_size should be int or a sequence of integers"):
        F1.gaussian_blur(image, kernel_size=1.5)


def test_sharpness_argument_errors():
    with pytest.raises(TypeError, match="can have 1 or 3 channels"):
        F.adjust_sharpness(img(4, 8, 8), sharpness_factor=0.5)
    with pytest.raises(ValueError, match="is not non-negative"):
        F.adjust_sharpness(img(), sharpness_factor=-1)
    with pytest.raises(TypeError, match="permitted channel values"):
        F1.adjust_sharpness(img(2, 8, 8), 0.5)


def test_early_returns_do_not_need_a_device():
    e = torch.empty(0, 3, 8, 8)
    assert F.gaussian_blur_image(e, [3, 3]) is e
    assert F.adjust_sharpness_image(e, 0.5) is e
    small = img(3, 2, 9)
    assert F.adjust_sharpness_image(small, 0.5) is small  # _color.py:240
    assert F1.adjust_sharpness(small, 0.5) is small
    for dims in [(0,), (5, 0), (0, 5)]:
        d = torch.empty(dims + (3, 17, 11))
        assert F.gaussian_blur_image(d, [3, 3]).shape == d.shape


def test_off_device_input_raises_loudly():
    with pytest.raises(mv.Mi355VisionError, match="no CPU fallback"):
        F.gaussian_blur(img(), [3, 3])
    with pytest.raises(mv.Mi355VisionError, match="no CPU fallback"):
        F.adjust_sharpness(img(dtype=torch.uint8), 0.5)
    with pytest.raises(mv.Mi355VisionError, match="no CPU fallback"):
        F.conv2d_bias_relu(torch.zeros(1, 3, 8, 8), torch.zeros(64, 3, 3, 3))


def test_taps_bit_identical_to_reference():
    g = golden("gaussian_kernels")
    for k, s in g["cases"]:
        k = int(k)
        np.testing.assert_array_equal(F._get_gaussian_kernel1d(k, s).numpy(), g[f"v2_{k}_{s}"])
        np.testing.assert_array_equal(F1._get_gaussian_kernel1d(k, s).numpy(), g[f"v1_{k}_{s}"])
    np.testing.assert_array_equal(F._get_gaussian_kernel2d([3, 5], [0.8, 0.5]).numpy(), g["v2_2d_3x5"])


def test_registry_dispatch_and_passthrough():
    assert _registry._get_kernel(F.gaussian_blur, torch.Tensor) is F.gaussian_blur_image
    assert _registry._get_kernel(F.gaussian_blur, tv_tensors.Video).__wrapped__ is F.gaussian_blur_video
    assert _registry._get_kernel(F.adjust_sharpness, tv_tensors.Image).__wrapped__ is F.adjust_sharpness_image
    with pytest.raises(TypeError, match="supports inputs of type"):
        _registry._get_kernel(F.gaussian_blur, tv_tensors.Mask)
    m = tv_tensors.Mask(torch.zeros(4, 4))
    assert _registry._get_kernel(F.gaussian_blur, tv_tensors.Mask, allow_passthrough=True)(m, [3, 3]) is m
    # masks / boxes flow through the transform untouched, like the reference (_transform.py:33-35)
    out = transforms.GaussianBlur(3)({"mask": m})
    assert out["mask"] is m


def test_register_kernel_public_rules():
    class MyImage(tv_tensors.TVTensor):
        pass

    calls = []

    @mv.register_kernel(F.gaussian_blur, MyImage)
    def my_blur(inpt, kernel_size, sigma=None):
        calls.append(kernel_size)
        return inpt

    x = MyImage(torch.zeros(3, 4, 4))
    assert F.gaussian_blur(x, [3, 3]) is x and calls == [[3, 3]]
    with pytest.raises(ValueError, match="already has a kernel registered"):
        mv.register_kernel("gaussian_blur", MyImage)(my_blur)
    with pytest.raises(ValueError, match="builtin tv_tensor classes"):
        mv.register_kernel(F.gaussian_blur, tv_tensors.Image)
    with pytest.raises(ValueError, match="subclasses of"):
        mv.register_kernel(F.gaussian_blur, torch.Tensor)
    with pytest.raises(ValueError, match="Could not find functional"):
        mv.register_kernel("bla", MyImage)

    class Other(tv_tensors.TVTensor):
        pass

    with pytest.raises(TypeError, match="supports inputs of type"):  # user tv_tensors never reach the tensor kernel
        F.gaussian_blur(Other(torch.zeros(3, 4, 4)), [3, 3])


def test_gaussian_blur_transform_assertions_and_params():
    with pytest.raises(ValueError, match="Kernel size should be a tuple/list of two integers"):
        transforms.GaussianBlur([10, 12, 14])
    with pytest.raises(ValueError, match="Kernel size value should be an odd and positive number"):
        transforms.GaussianBlur(4)
    with pytest.raises(ValueError, match="If sigma is a sequence its length should be 1 or 2. Got 3"):
        transforms.GaussianBlur(3, sigma=[1, 2, 3])
    with pytest.raises(ValueError, match="sigma values should be positive and of the form"):
        transforms.GaussianBlur(3, sigma=-1.0)
    with pytest.raises(ValueError, match="sigma values should be positive and of the form"):
        transforms.GaussianBlur(3, sigma=[2.0, 1.0])
    with pytest.raises(TypeError, match="sigma should be a number or a sequence of numbers"):
        transforms.GaussianBlur(3, sigma={})
    for sigma in [10.0, [10.0, 12.0], (10, 12.0), [10]]:
        p = transforms.GaussianBlur(3, sigma=sigma)._get_params([])
        if isinstance(sigma, float):
            assert p["sigma"][0] == p["sigma"][1] == sigma
        elif isinstance(sigma, list) and len(sigma) == 1:
            assert p["sigma"][0] == p["sigma"][1] == sigma[0]
        else:
            assert sigma[0] <= p["sigma"][0] <= sigma[1] and p["sigma"][0] == p["sigma"][1]
    with pytest.raises(ValueError, match="must be positive"):
        transforms.GaussianBlurV1(3, sigma=0)
    assert 0.1 <= transforms.GaussianBlurV1.get_params(0.1, 2.0) <= 2.0


def test_random_adjust_sharpness_p0_is_identity():
    x = img()
    assert transforms.RandomAdjustSharpness(2.0, p=0.0)(x) is x
    with pytest.raises(ValueError):
        transforms.RandomAdjustSharpness(2.0, p=1.5)


def test_conv_module_parameter_layout_matches_nn_conv2d():
    from cpu_vision_amd.nn import Conv2dNormActivation, Conv3x3ReLU

    torch.manual_seed(0)
    conv = torch.nn.Conv2d(3, 64, 3, padding=1)
    m = Conv3x3ReLU.from_conv(conv)
    assert m.weight.shape == conv.weight.shape and torch.equal(m.weight, conv.weight) and torch.equal(m.bias, conv.bias)
    m2 = Conv3x3ReLU(3, 64)
    m2.load_state_dict({"weight": conv.weight, "bias": conv.bias})
    assert float(Conv3x3ReLU(3, 64).bias.detach().abs().sum()) == 0.0  # vgg init: bias = 0
    from cpu_vision_amd import mobilenet
    assert Conv2dNormActivation is mobilenet.Conv2dNormActivation  # one class under that name
    blk = Conv2dNormActivation(3, 64, norm_layer=None)
    assert blk[0].bias is not None and blk.out_channels == 64 and [type(m).__name__ for m in blk] == ["Conv2d", "ReLU"]
    # the reference's defaults (ops/misc.py:68-128): BatchNorm2d, ReLU(inplace), bias only without a norm, 'same' padding
    dflt = Conv2dNormActivation(3, 8, kernel_size=5, dilation=2)
    assert [type(m).__name__ for m in dflt] == ["Conv2d", "BatchNorm2d", "ReLU"] and dflt[0].bias is None
    assert dflt[0].padding == (4, 4) and dflt[2].inplace
    with pytest.raises(ValueError):
        Conv3x3ReLU.from_conv(torch.nn.Conv2d(3, 8, 3, padding=0))


def test_resize_geometry_matches_reference_rules():
    """_compute_resized_output_size (functional.py:353-384) and the center_crop box (functional.py:572-594),
    including the padded cases, against the oracle's numpy statement on index images."""
    from oracle import ref
    assert F1._compute_resized_output_size((375, 500), [256]) == [256, 341]
    assert F1._compute_resized_output_size((500, 375), [256]) == [341, 256]
    assert F1._compute_resized_output_size((20, 100), [16], 40) == [8, 40]
    assert F1._compute_resized_output_size((20, 100), [7, 9]) == [7, 9]
    with pytest.raises(ValueError, match="max_size = 10 must be strictly greater"):
        F1._compute_resized_output_size((20, 100), [16], 10)
    for (h, w), crop in [((32, 42), 28), ((16, 48), 24), ((5, 7), 9), ((9, 9), 9), ((10, 3), [4, 8]), ((7, 8), 3)]:
        img = torch.arange(1, 1 + 2 * h * w, dtype=torch.float32).reshape(2, h, w)
        want = ref.center_crop(img.numpy(), crop if isinstance(crop, list) else [crop])
        got = F1.center_crop(img, crop if isinstance(crop, list) else [crop])
        np.testing.assert_array_equal(got.numpy(), want)
        top, left, ch, cw = F1._center_crop_window(h, w, crop)
        assert (ch, cw) == want.shape[-2:]
    with pytest.raises(NotImplementedError):
        F1.resize(torch.zeros(3, 8, 8), [4], interpolation="nearest")
    with pytest.raises(NotImplementedError):
        F1.resize(torch.zeros(3, 8, 8), [4], antialias=False)
    x = torch.zeros(3, 8, 8)
    assert F1.resize(x, [8, 8]) is x  # same size: returned as is, no device needed
    with pytest.raises(mv.Mi355VisionError):
        F1.resize(x, [4])  # CPU tensor: no fallback


def test_linear_k_slicing_plan_and_oracle_order():
    """mv_linear_k_slices / mv_linear_workspace_bytes are host logic: slices cover K exactly once, slice_len is a multiple of
    32, large batches keep the single chain; the oracle's sliced order equals the plain one when one slice covers K."""
    import numpy as np

    from cpu_vision_amd import _lib, functional as F
    from oracle import ref

    lib = _lib.load()
    for n, k, m in [(1, 25088, 4096), (64, 25088, 4096), (1, 4096, 1000), (256, 4096, 4096), (4096, 4096, 4096), (5, 300, 70),
                    (1, 9216, 4096), (0, 128, 128)]:
        slices, slice_len = F.linear_k_slices(n, k, m)
        assert slices >= 1 and slice_len % 32 == 0
        assert slices * slice_len >= k and (slices - 1) * slice_len < k
        ws = int(lib.mv_linear_workspace_bytes(n, k, m))
        assert ws == (slices * n * m * 4 if slices > 1 else 0)
    assert F.linear_k_slices(1, 25088, 4096)[0] > 8          # batch 1: K spread over the chip
    assert F.linear_k_slices(4096, 4096, 4096) == (1, 4096)  # training-size batch: single chain
    rng = np.random.default_rng(0)
    x, w, b = rng.random((3, 100), dtype=np.float32), rng.random((7, 100), dtype=np.float32), rng.random(7, dtype=np.float32)
    np.testing.assert_array_equal(ref.linear_bias_relu(x, w, b, slice_len=128), ref.linear_bias_relu(x, w, b))
    sliced = ref.linear_bias_relu(x, w, b, slice_len=32)
    np.testing.assert_allclose(sliced, ref.linear_bias_relu(x, w, b), rtol=1e-6)


# ----------------------------------------------------------------------------- round 2: PIL entries, ElasticTransform, float64 taps
def test_pil_kernels_are_registered_and_conversion_round_trips():
    import PIL.Image
    from cpu_vision_amd import _pil
    assert _registry._get_kernel(F.gaussian_blur, PIL.Image.Image) is F._gaussian_blur_image_pil     # _misc.py:169-174
    assert _registry._get_kernel(F.adjust_sharpness, PIL.Image.Image) is F._adjust_sharpness_image_pil
    g = golden("round2_api")
    for mode in ("RGB", "L", "RGBA"):
        arr = g[f"pil_{mode}__x"]
        im = PIL.Image.fromarray(arr[:, :, 0] if mode == "L" else arr, mode=mode)
        tns = _pil.pil_to_tensor(im)
        assert tns.dtype == torch.uint8 and tuple(tns.shape) == (arr.shape[2], arr.shape[0], arr.shape[1])
        back = _pil.to_pil_image(tns, mode=mode)
        assert back.mode == mode and np.array_equal(np.asarray(back).reshape(arr.shape), arr)
    with pytest.raises(TypeError, match="pic should be PIL Image"):
        _pil.pil_to_tensor(torch.zeros(3, 4, 4))
    with pytest.raises(ValueError, match="should not have > 4 channels"):
        _pil.to_pil_image(torch.zeros(5, 4, 4, dtype=torch.uint8))
    with pytest.raises(ValueError, match="Only modes"):
        _pil.to_pil_image(torch.zeros(3, 4, 4, dtype=torch.uint8), mode="RGBA")


def test_pil_input_without_a_gpu_fails_loudly():
    import PIL.Image
    if torch.cuda.is_available():
        pytest.skip("box has a GPU")
    im = PIL.Image.fromarray(np.zeros((8, 8, 3), np.uint8))
    with pytest.raises(mv.Mi355VisionError, match="no CPU fallback"):
        F.gaussian_blur(im, [3, 3])
    with pytest.raises(mv.Mi355VisionError, match="no CPU fallback"):
        transforms.GaussianBlur(3)(im)


def test_elastic_transform_constructor_and_query_size():
    tr = transforms.ElasticTransform()
    assert tr.alpha == [50.0, 50.0] and tr.sigma == [5.0, 5.0]
    tr = transforms.ElasticTransform(alpha=(30, 60), sigma=[2.0])
    assert tr.alpha == [30.0, 60.0] and tr.sigma == [2.0, 2.0]
    with pytest.raises(TypeError, match="alpha should be a number or a sequence of numbers"):
        transforms.ElasticTransform(alpha=object())
    with pytest.raises(ValueError, match="sigma is a sequence its length should be 1 or 2"):
        transforms.ElasticTransform(sigma=[1.0, 2.0, 3.0])
    assert transforms.query_size([torch.zeros(3, 5, 7), tv_tensors.Image(torch.zeros(1, 5, 7)), 3, "x"]) == (5, 7)
    with pytest.raises(TypeError, match="No image, video, mask or bounding box was found"):
        transforms.query_size([1, "a"])
    with pytest.raises(ValueError, match="Found multiple HxW dimensions"):
        transforms.query_size([torch.zeros(3, 5, 7), torch.zeros(3, 6, 7)])
    with pytest.raises(NotImplementedError, match="grid_sample"):
        tr._transform(torch.zeros(3, 5, 7), {})


def test_float64_taps_are_built_in_float64():
    """A float64 image gets float64 taps (the reference passes dtype=image.dtype to _get_gaussian_kernel2d, _misc.py:141)."""
    t64, arr = F._host_taps(5, 1.1, False, torch.float64)
    assert t64.dtype == torch.float64 and len(arr) == 5 and arr[2] == float(t64[2])
    t32, _ = F._host_taps(5, 1.1)
    assert t32.dtype == torch.float32 and float(t64[2]) != float(t32[2])
    assert F._taps_dtype(torch.zeros(1, dtype=torch.float64)) == torch.float64
    assert F._taps_dtype(torch.zeros(1, dtype=torch.float16)) == torch.float32


def test_vgg_has_the_reference_module_tree_and_seeded_weights():
    """models/vgg.py:35-87: features = Sequential(Conv2d, ReLU, MaxPool2d, ...), classifier = Sequential(Linear, ReLU, Dropout,
    ...): the reference's state-dict keys, loadable with plain load_state_dict, and a seeded construction that draws from
    the RNG like the reference's (per-parameter sums of the reference's own seeded model: tests/golden/vgg11_forward.npz)."""
    from cpu_vision_amd.nn import vgg11, vgg11_reference_init
    g = golden("vgg11_forward")
    torch.manual_seed(0)
    model = vgg11(num_classes=50)
    sd = model.state_dict()
    assert list(sd.keys()) == [str(n) for n in g["param_names"]]
    assert [k for k in sd if k.startswith("classifier")] == ["classifier.0.weight", "classifier.0.bias", "classifier.3.weight",
                                                             "classifier.3.bias", "classifier.6.weight", "classifier.6.bias"]
    np.testing.assert_allclose([float(v.double().sum()) for v in sd.values()], g["param_sum"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose([float(v.double().abs().sum()) for v in sd.values()], g["param_abs_sum"], rtol=1e-9)
    ref_state = vgg11_reference_init(num_classes=50, seed=0)  # the reference's constructor sequence, restated
    assert all(torch.equal(sd[k], ref_state[k]) for k in ref_state)
    fresh = vgg11(num_classes=50)
    fresh.load_state_dict(ref_state)  # strict: a reference state dict is this module's state dict
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), sd.values()))
    assert not model.training
    with pytest.raises(mv.Mi355VisionError):
        model(torch.zeros(1, 3, 32, 32))  # CPU tensor: no fallback


def test_preset_steps_as_v2_transforms_constructors_and_repr():
    """Resize / CenterCrop / ToDtype / Normalize / Compose mirror the reference's classes (v2/_geometry.py:76-193,
    v2/_misc.py:134-304, v2/_container.py:10-64): same attributes, same repr as the reference's own pipeline, same errors."""
    T = transforms
    pipe = T.Compose([T.Resize(40), T.CenterCrop(32), T.ToDtype(torch.float32, scale=True),
                      T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]), T.GaussianBlur(3, sigma=(0.9, 0.9))])
    assert repr(pipe) == str(golden("round2_api")["pipe_f32__repr"])
    assert T.Resize((10, 20)).size == [10, 20] and T.Resize(None, max_size=7).size is None and T.CenterCrop(5).size == (5, 5)
    with pytest.raises(ValueError, match="max_size must be an integer when size is None"):
        T.Resize(None)
    with pytest.raises(ValueError, match="size can be an integer, a sequence of one or two integers, or None"):
        T.Resize((1, 2, 3))
    with pytest.raises(ValueError, match="Please provide only two dimensions"):
        T.CenterCrop((1, 2, 3))
    with pytest.raises(ValueError, match="dtype must be a dict or a torch.dtype"):
        T.ToDtype("float32")
    with pytest.raises(TypeError, match="should be a sequence of callables"):
        T.Compose(T.Resize(4))
    with pytest.raises(ValueError, match="Pass at least one transform"):
        T.Compose([])
    import PIL.Image
    with pytest.raises(TypeError, match="does not support PIL images"):
        T.Normalize([0.5], [0.5])(PIL.Image.new("L", (4, 4)))
    mask = tv_tensors.Mask(torch.zeros(4, 4, dtype=torch.uint8))
    assert T.ToDtype(torch.float32, scale=True)(mask) is mask                 # a plain dtype only touches images / videos
    assert T.ToDtype({tv_tensors.Mask: torch.int64, "others": None})(mask).dtype == torch.int64
    with pytest.raises(ValueError, match="No dtype was specified for type"):
        T.ToDtype({tv_tensors.Image: torch.float32})(mask)
    x = torch.arange(3 * 6 * 8, dtype=torch.uint8).reshape(3, 6, 8)          # views need no GPU
    assert torch.equal(T.CenterCrop((4, 6))(x), x[:, 1:5, 1:7])


# ----------------------------------------------------------------------------- round 3: the stated summation orders are pinned
def test_k_slice_plans_are_pinned():
    """The GPU parity tests ask the library which K-slice plan it runs and let the oracle restate it -- so the plans themselves
    are pinned HERE, against a committed fixture: a change that moves mv_*_k_slices and the kernels together fails this test
    until tests/golden/make_k_slice_plans.py is re-run on purpose."""
    import importlib.util
    import json
    gen = Path(__file__).parent / "golden" / "make_k_slice_plans.py"
    spec = importlib.util.spec_from_file_location("make_k_slice_plans", gen)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pinned = json.loads((gen.parent / "k_slice_plans.json").read_text())
    now = mod.plans()
    for kind in ("conv3x3", "linear", "conv1x1", "inverted_residual"):
        assert sorted(now[kind]) == sorted(pinned[kind]), f"{kind}: the set of pinned shapes changed"
        for shape, plan in pinned[kind].items():
            assert now[kind][shape] == plan, f"{kind} ({shape}): the library now states {now[kind][shape]}, pinned {plan}"
    # every plan covers K exactly once, in order: slices * slice_len >= K > (slices - 1) * slice_len
    for shape, (slices, sl) in pinned["conv3x3"].items():
        cin = int(shape.split(",")[1])
        assert (slices - 1) * sl < cin <= slices * sl
    for shape, (slices, sl) in pinned["linear"].items():
        k = int(shape.split(",")[1])
        assert (slices - 1) * sl < k <= slices * sl and (slices == 1 or sl % 32 == 0)
    for shape, (slices, sl) in pinned["conv1x1"].items():
        cin = int(shape.split(",")[1])
        assert (slices - 1) * sl < cin <= slices * sl and (slices == 1 or sl % 32 == 0)
    fused = 0
    for shape, (slices, sl) in pinned["inverted_residual"].items():
        hidden, side = int(shape.split(",")[2]), int(shape.split(",")[4])
        assert slices >= 1, shape  # every expanding block of MobileNetV2 runs as one kernel (k_invres, k_invres_wide on 112 / 56-pixel maps)
        if slices:
            fused += 1
            assert (slices - 1) * sl < hidden <= slices * sl and (slices == 1 or sl % 32 == 0)
    assert fused == 12 * 4  # twelve distinct fused block shapes (the first block, without expansion, included) x four batch sizes


def test_slice_plans_depend_on_the_batch_and_the_switch_is_exposed():
    """the plan follows the workgroup count, i.e. the batch size -- stated in the header and pinned here; callers
    that need batch-invariant bits set functional.BATCH_INVARIANT_SUMMATION (single chain for every batch)."""
    assert F.conv3x3_k_slices(1, 512, 14, 14, 512)[0] > 1 and F.conv3x3_k_slices(64, 512, 14, 14, 512)[0] == 1
    assert F.linear_k_slices(1, 25088, 4096)[0] > F.linear_k_slices(256, 25088, 4096)[0] > F.linear_k_slices(1024, 25088, 4096)[0] == 1
    assert F.BATCH_INVARIANT_SUMMATION is False
    import inspect
    assert inspect.signature(F.conv2d_bias_relu).parameters["sliced_k"].default is None
    assert inspect.signature(F.linear_bias_relu).parameters["sliced_k"].default is None
    header = (Path(__file__).parent.parent / "include" / "mi355vision.h").read_text()
    assert "BATCH DEPENDENCE" in header
    # a fused InvertedResidual block with a several-slice plan falls back to three launches under the flag (host logic only)
    import torch
    from cpu_vision_amd.mobilenet import InvertedResidual
    blk = InvertedResidual(64, 64, 1, 6).eval()
    x1, x256 = torch.empty((1, 64, 14, 14)), torch.empty((256, 64, 14, 14))
    assert blk._fused_plan(x1)[0] > 1 and blk._fused_plan(x256) == (1, 384)
    F.BATCH_INVARIANT_SUMMATION = True
    try:
        assert blk._fused_plan(x1) is None and blk._fused_plan(x256) == (1, 384)
    finally:
        F.BATCH_INVARIANT_SUMMATION = False
