#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  It imports the reference's pure-Python package, feeds it seeded
inputs (numpy Philox, so any consumer can regenerate them) and stores inputs and
the reference's outputs as small .npz files.  Only data is written: no reference
source text is copied.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Import note: `import torchvision` from the source tree
fails with an ordinary RuntimeError because the `_C` extension is unbuilt and
`_meta_registrations.py` registers a fake for `torchvision::nms`; defining that
schema first lets the pure-Python package import with `_HAS_OPS=False`.
"""
import sys
import warnings
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")

warnings.filterwarnings("ignore")
torch.library.define("torchvision::nms", "(Tensor dets, Tensor scores, float iou_threshold) -> Tensor")
sys.path.insert(0, str(REF))
sys.dont_write_bytecode = True
import torchvision  # noqa: E402
import torchvision.transforms._functional_tensor as F_t  # noqa: E402
from torchvision.transforms.v2 import functional as F  # noqa: E402
from torchvision.transforms.v2.functional import _misc as F_misc  # noqa: E402


def philox_f32(seed: int, shape) -> np.ndarray:
    """U[0,1) float32 from numpy's Philox bit generator -- reproducible anywhere."""
    return np.random.Generator(np.random.Philox(seed)).random(shape, dtype=np.float32)


def philox_u8(seed: int, shape) -> np.ndarray:
    return np.random.Generator(np.random.Philox(seed)).integers(0, 256, shape, dtype=np.uint8)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def save(name, **arrays):
    path = HERE / f"{name}.npz"
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz: {path.stat().st_size / 1024:.1f} KiB, {len(arrays)} arrays")


# ----------------------------------------------------------------------------- 1-D / 2-D kernels
def gen_kernels():
    out = {}
    cases = [(1, 0.5), (3, 0.8), (3, 0.5), (5, 1.1), (5, 0.15 * 5 + 0.35), (7, 2.0), (9, 0.3), (23, 1.7), (41, 5.0)]
    out["cases"] = np.array(cases, np.float64)
    for k, s in cases:
        out[f"v2_{k}_{s}"] = F_misc._get_gaussian_kernel1d(k, s, torch.float32, torch.device("cpu")).numpy()
        out[f"v1_{k}_{s}"] = F_t._get_gaussian_kernel1d(k, s, torch.float32, torch.device("cpu")).numpy()
    out["v2_2d_3x5"] = F_misc._get_gaussian_kernel2d([3, 5], [0.8, 0.5], torch.float32, torch.device("cpu")).numpy()
    save("gaussian_kernels", **out)


# ----------------------------------------------------------------------------- OpenCV vectors (reference's own fixture)
def gen_opencv():
    # test/assets/gaussian_blur_opencv_results.pt is a pickle of numpy scalars; it is a data file of the
    # reference's own tests (test_transforms_v2.py:3259-3309).  Re-encode it as plain uint8 arrays.
    import numpy
    import _codecs
    # allow-list exactly the three globals the pickle uses (inspected with pickletools); numpy 2 moved
    # numpy.core to numpy._core, so name the old path explicitly
    allow = [(numpy._core.multiarray.scalar, "numpy.core.multiarray.scalar"), numpy.dtype,
             type(numpy.dtype("uint8")), (_codecs.encode, "_codecs.encode")]
    with torch.serialization.safe_globals(allow):
        d = torch.load(REF / "test/assets/gaussian_blur_opencv_results.pt", weights_only=True)
    out = {}
    for key, vals in d.items():
        out[key] = np.asarray([int(v) for v in vals], np.uint8)
    save("opencv_gaussian_blur", **out)
    return out


# ----------------------------------------------------------------------------- gaussian blur
def gen_blur():
    out = {}
    idx = []
    seed = 100
    cfgs = [([3, 3], [0.8, 0.8]), ([3, 5], [0.8, 0.5]), ([5, 5], None), ([23, 23], [1.7, 1.7]), ([5, 3], [1.1, 0.6]),
            ([1, 1], [0.5, 0.5]), ([3, 1], None), ([1, 7], [1.0, 2.0])]
    shapes = [("f32", (3, 37, 53)), ("u8", (3, 37, 53)), ("f32", (1, 26, 28)), ("f32", (2, 1, 3, 17, 11)),
              ("u8", (2, 3, 24, 40)), ("f32", (1, 40, 260)), ("f32", (1, 12, 300))]
    for ks, sg in cfgs:
        for dt, shp in shapes:
            if ks[0] // 2 >= shp[-1] or ks[1] // 2 >= shp[-2]:
                continue
            seed += 1
            x = philox_f32(seed, shp) if dt == "f32" else philox_u8(seed, shp)
            name = f"k{ks[0]}x{ks[1]}_s{'d' if sg is None else '_'.join(map(str, sg))}_{dt}_{'x'.join(map(str, shp))}"
            y2 = F.gaussian_blur_image(t(x), kernel_size=ks, sigma=sg).numpy()
            sg1 = sg if sg is not None else [k * 0.15 + 0.35 for k in ks]
            out[name + "__x"] = x
            out[name + "__y_v2"] = y2
            if len(shp) <= 4:  # the v1 tensor backend only takes (C,H,W) / (B,C,H,W)
                out[name + "__y_v1"] = F_t.gaussian_blur(t(x), ks, sg1).numpy()
            idx.append(name)
    out["index"] = np.array(idx)
    save("gaussian_blur", **out)


# ----------------------------------------------------------------------------- sharpness
def gen_sharpness():
    out = {}
    idx = []
    seed = 200
    for dt, shp in [("u8", (3, 37, 53)), ("f32", (3, 37, 53)), ("u8", (1, 9, 260)), ("f32", (2, 2, 3, 16, 20)),
                    ("u8", (2, 1, 3, 5)), ("u8", (3, 2, 40)), ("f32", (1, 3, 3))]:
        seed += 1
        x = philox_f32(seed, shp) if dt == "f32" else philox_u8(seed, shp)
        out[f"{dt}_{'x'.join(map(str, shp))}__x"] = x
        for f in [0.0, 0.1, 0.4, 0.5, 1.0, 2.0, 3.7]:
            name = f"f{f}_{dt}_{'x'.join(map(str, shp))}"
            out[name + "__y_v2"] = F.adjust_sharpness_image(t(x), sharpness_factor=f).numpy()
            if len(shp) <= 4:
                out[name + "__y_v1"] = F_t.adjust_sharpness(t(x), f).numpy()
            idx.append(name)
    # extreme-contrast image: pins the fused-multiply-add behaviour of the in-place blend (_color.py:270)
    x = np.zeros((1, 16, 64), np.uint8)
    x[:, ::2, ::2] = 255
    x[:, 5:9, 20:40] = 255
    x[:, 6, 25] = 0
    out["u8_extreme__x"] = x
    for f in [0.3, 0.4, 0.6, 0.9, 1.7]:
        name = f"f{f}_u8_extreme"
        out[name + "__y_v2"] = F.adjust_sharpness_image(t(x), sharpness_factor=f).numpy()
        out[name + "__y_v1"] = F_t.adjust_sharpness(t(x), f).numpy()
        idx.append(name)
    # PIL-exact contract of the reference's own test (test_transforms_v2.py:4721-4731): f in {0.1, 0.5, 1.0}
    from PIL import Image, ImageEnhance
    xp = philox_u8(299, (3, 17, 11))
    out["pil__x"] = xp
    for f in [0.1, 0.5, 1.0]:
        pil = Image.fromarray(np.transpose(xp, (1, 2, 0)))
        ypil = np.transpose(np.asarray(ImageEnhance.Sharpness(pil).enhance(f)), (2, 0, 1))
        yref = F.adjust_sharpness_image(t(xp), sharpness_factor=f).numpy()
        assert np.array_equal(ypil, yref), "reference is PIL-exact here"
        out[f"pil__y_{f}"] = np.ascontiguousarray(ypil)
    out["index"] = np.array(idx)
    save("adjust_sharpness", **out)


# ----------------------------------------------------------------------------- box / separable / sobel on the reference primitive
def primitive(x4, k2d, border):
    """The reference's primitive called the way gaussian_blur_image calls it (_misc.py:153-155)."""
    from torch.nn.functional import conv2d, pad
    c = x4.shape[1]
    ky, kx = k2d.shape
    if border == "reflect":
        x4 = pad(x4, [kx // 2, kx // 2, ky // 2, ky // 2], mode="reflect")
    elif border == "zero":
        x4 = pad(x4, [kx // 2, kx // 2, ky // 2, ky // 2])
    return conv2d(x4, k2d.expand(c, 1, ky, kx), groups=c)


def gen_primitive():
    out = {}
    x = philox_f32(301, (1, 1, 32, 40))
    out["box__x"] = x
    out["box__y"] = primitive(t(x), torch.full((3, 3), 1.0 / 9.0), "reflect").numpy()
    x = philox_f32(302, (1, 3, 32, 40))
    k1 = F_misc._get_gaussian_kernel1d(5, 1.1, torch.float32, torch.device("cpu"))
    blur = primitive(primitive(t(x), k1[None, :], "reflect"), k1[:, None], "reflect")
    gxk = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]])
    out["sep__x"] = x
    out["sep__k1d"] = k1.numpy()
    out["sep__blur"] = blur.numpy()
    out["sep__gx"] = primitive(blur, gxk, "reflect").numpy()
    out["sep__gy"] = primitive(blur, gxk.t().contiguous(), "reflect").numpy()
    for border in ["reflect", "zero", "valid"]:
        out[f"sobel_{border}__gx"] = primitive(t(x), gxk, border).numpy()
        out[f"sobel_{border}__gy"] = primitive(t(x), gxk.t().contiguous(), border).numpy()
    # generic taps, all three borders, odd sizes
    x = philox_f32(303, (2, 2, 19, 45))
    out["gen__x"] = x
    for ky, kx in [(3, 3), (5, 3), (1, 5), (7, 7)]:
        w = philox_f32(310 + ky * 10 + kx, (ky, kx)) - 0.5
        out[f"gen_{ky}x{kx}__w"] = w
        for border in ["reflect", "zero", "valid"]:
            out[f"gen_{ky}x{kx}_{border}__y"] = primitive(t(x), t(w), border).numpy()
    save("primitive_filters", **out)


# ----------------------------------------------------------------------------- first CNN layer
def gen_cnn():
    from torchvision.models import vgg11
    from torchvision.ops.misc import Conv2dNormActivation
    out = {}
    torch.manual_seed(0)
    model = vgg11(num_classes=50).eval()  # test_models.py:674-693 (seed 0, 50 classes)
    layer = model.features[0:2]
    w, b = model.features[0].weight.detach(), model.features[0].bias.detach()
    x = philox_f32(401, (2, 3, 16, 20))
    out["vgg11__w"], out["vgg11__b"] = w.numpy(), b.numpy()
    out["vgg11__x"] = x
    with torch.no_grad():
        out["vgg11__y"] = layer(t(x)).numpy()
    # non-zero bias + negative inputs (vgg init sets bias to 0)
    b2 = (philox_f32(402, (64,)) - 0.5) * 0.2
    x2 = philox_f32(403, (1, 3, 33, 47)) * 2 - 1
    out["bias__b"], out["bias__x"] = b2, x2
    with torch.no_grad():
        y = torch.nn.functional.conv2d(t(x2), w, t(b2), padding=1)
        out["bias__y_norelu"] = y.numpy()
        out["bias__y"] = torch.relu(y).numpy()
    # Conv2dNormActivation(norm_layer=None) == conv + bias + ReLU (ops/misc.py:68-172)
    torch.manual_seed(1)
    blk = Conv2dNormActivation(3, 64, kernel_size=3, norm_layer=None).eval()
    out["cna__w"], out["cna__b"] = blk[0].weight.detach().numpy(), blk[0].bias.detach().numpy()
    with torch.no_grad():
        out["cna__y"] = blk(t(x)).numpy()
    # second-layer shape (Cin=64 -> Cout=8 slice) for the general-Cin kernel
    torch.manual_seed(2)
    conv = torch.nn.Conv2d(16, 32, 3, padding=1)
    x3 = philox_f32(404, (2, 16, 12, 36)) - 0.5
    out["c16__w"], out["c16__b"], out["c16__x"] = conv.weight.detach().numpy(), conv.bias.detach().numpy(), x3
    with torch.no_grad():
        out["c16__y"] = torch.relu(conv(t(x3))).numpy()
    save("conv_relu", **out)


# ----------------------------------------------------------------------------- whole small CNN
def gen_vgg():
    """vgg11(num_classes=50), seed 0, eval, on torch.rand(1,3,224,224) -- test_models.py:674-693.  The weights are
    NOT stored (531 MB): the consumer rebuilds them with the same seeded constructor sequence and checks the
    per-layer checksums stored here."""
    from torchvision.models import vgg11
    out = {}
    torch.manual_seed(0)
    model = vgg11(num_classes=50).eval()
    x = torch.rand(1, 3, 224, 224)
    with torch.no_grad():
        feats = model.features(x)
        y = model(x)
        f2 = model.features[0:6](x)  # conv1+relu+pool+conv2+relu+pool
    out["x"] = x.numpy()
    out["y"] = y.numpy()
    out["features"] = feats.numpy()
    out["features_0_6"] = f2.numpy()[:, :8]  # first 8 channels of the 128x56x56 activation
    sums = []
    for name, prm in model.state_dict().items():
        sums.append((name, float(prm.double().sum()), float(prm.double().abs().sum())))
    out["param_names"] = np.array([s[0] for s in sums])
    out["param_sum"] = np.array([s[1] for s in sums], np.float64)
    out["param_abs_sum"] = np.array([s[2] for s in sums], np.float64)
    # the reference's own expect file for this test (data of its test-suite), re-encoded
    exp = torch.load(REF / "test/expect/ModelTester.test_vgg11_expect.pkl", weights_only=True)
    out["reference_expect_pkl"] = exp.detach().numpy()
    save("vgg11_forward", **out)
    # small layers for bit-exact oracle comparisons: maxpool / linear / mid-network convs from the same model
    o2 = {}
    xs = philox_f32(501, (2, 64, 12, 20)) - 0.3
    with torch.no_grad():
        o2["c64_128__x"] = xs
        o2["c64_128__w"] = model.features[3].weight.numpy()
        o2["c64_128__b"] = (philox_f32(502, (128,)) - 0.5) * 0.1
        yy = torch.relu(torch.nn.functional.conv2d(t(xs), model.features[3].weight, t(o2["c64_128__b"]), padding=1))
        o2["c64_128__y"] = yy.numpy()
        o2["pool__y"] = torch.nn.functional.max_pool2d(yy, 2, 2).numpy()
        xl = philox_f32(503, (5, 300)) - 0.5
        wl = (philox_f32(504, (70, 300)) - 0.5) * 0.1
        bl = philox_f32(505, (70,)) - 0.5
        o2["lin__x"], o2["lin__w"], o2["lin__b"] = xl, wl, bl
        o2["lin__y"] = torch.relu(torch.nn.functional.linear(t(xl), t(wl), t(bl))).numpy()
        xa = philox_f32(506, (2, 3, 10, 13))
        o2["avg__x"] = xa
        o2["avg__y"] = torch.nn.functional.adaptive_avg_pool2d(t(xa), (7, 7)).numpy()
    save("cnn_layers", **o2)


# ----------------------------------------------------------------------------- preset tail
def gen_preset():
    """ImageClassification's last two steps (transforms/_presets.py:58-60): convert_image_dtype(float) + normalize,
    and the same feeding the first VGG layer."""
    from torchvision.models import vgg11
    out = {}
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    xu = philox_u8(601, (2, 3, 24, 36))
    xf = F.to_dtype(t(xu), torch.float32, scale=True)
    out["x_u8"] = xu
    out["to_float"] = xf.numpy()
    out["normalized"] = F.normalize(xf, mean=mean, std=std).numpy()
    out["mean"], out["std"] = np.array(mean, np.float64), np.array(std, np.float64)
    xg = philox_u8(602, (1, 1, 17, 19))
    out["gray_u8"] = xg
    out["gray_normalized"] = F.normalize(F.to_dtype(t(xg), torch.float32, scale=True), mean=[0.5], std=[0.25]).numpy()
    torch.manual_seed(0)
    model = vgg11(num_classes=50).eval()
    with torch.no_grad():
        out["vgg_first_layer"] = model.features[0:2](t(out["normalized"])).numpy()
    save("preset_tail", **out)


def gen_resize_preset():
    """The preset's head and the whole ImageClassification.forward (transforms/_presets.py:56-63) through the
    reference's v1 functional API (F.resize -> _functional_tensor.resize -> torch interpolate(antialias=True))."""
    import torchvision.transforms.functional as F1
    from torchvision.transforms._presets import ImageClassification
    out, index = {}, []
    cases = [  # name, dtype, shape, resize size, max_size, crop
        ("land_u8", "u8", (3, 75, 100), [32], None, 28),
        ("port_u8", "u8", (3, 64, 48), [32], None, 28),
        ("batch_u8", "u8", (2, 3, 40, 60), [24], None, 20),
        ("gray_u8", "u8", (1, 33, 47), [16], None, 16),
        ("up_u8", "u8", (3, 20, 30), [48], None, 40),
        ("pad_u8", "u8", (3, 30, 90), [16], None, 24),
        ("hw_u8", "u8", (3, 50, 70), [21, 34], None, 20),
        ("max_u8", "u8", (3, 20, 100), [16], 40, 8),
        ("land_f32", "f32", (3, 75, 100), [32], None, 28),
        ("photo_u8", "u8", (3, 188, 250), [128], None, 112),
    ]
    for k, (name, dt, shape, size, max_size, crop) in enumerate(cases):
        x = philox_u8(700 + k, shape) if dt == "u8" else philox_f32(700 + k, shape)
        if name == "photo_u8":  # photo-like content: smooth field + noise, still seeded
            yy, xx = np.mgrid[0:shape[-2], 0:shape[-1]].astype(np.float32)
            base = 127 + 100 * np.sin(xx / 37.0) * np.cos(yy / 23.0)
            x = np.clip(base[None] + (x.astype(np.float32) - 128) * 0.2 + np.arange(3, dtype=np.float32)[:, None, None] * 9, 0, 255).astype(np.uint8)
        r = F1.resize(t(x), size, max_size=max_size, antialias=True)
        c = F1.center_crop(r, [crop])
        out[f"{name}__x"], out[f"{name}__resized"] = x, r.numpy()
        out[f"{name}__cropped"] = c.numpy()
        out[f"{name}__args"] = np.array(size + [-1 if max_size is None else max_size, crop], np.int64)
        if max_size is None and len(size) == 1:
            nch = shape[-3]
            mean, std = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]) if nch == 3 else ([0.45], [0.25])
            out[f"{name}__preset"] = ImageClassification(crop_size=crop, resize_size=size[0], mean=mean, std=std)(t(x)).numpy()
        index.append(name)
    out["index"] = np.array(index)
    save("resize_preset", **out)


def randomize_norms(model, seed):
    """Seeded, order-dependent perturbation of every norm layer's affine parameters and statistics (identity-like
    defaults would hide a wrong fold).  tests/_util.py::randomize_norms repeats it on the MI355X modules."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, "running_var") and m.running_var is not None:
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.rand(m.bias.shape, generator=g) - 0.5)
                m.running_mean.copy_(torch.rand(m.running_mean.shape, generator=g) * 0.4 - 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 1.5 + 0.4)


def gen_mobilenet():
    """Conv2dNormActivation with BatchNorm2d(eval) / FrozenBatchNorm2d + ReLU6 / Hardswish / ReLU, and
    MobileNetV2 end to end (models/mobilenetv2.py), all through the reference's own modules."""
    from torchvision.models import mobilenet_v2
    from torchvision.ops.misc import Conv2dNormActivation, FrozenBatchNorm2d
    out, index = {}, []
    blocks = [  # name, cin, cout, kernel, stride, groups, norm, act, input hw
        ("stem_bn_relu6", 3, 32, 3, 2, 1, torch.nn.BatchNorm2d, torch.nn.ReLU6, (33, 35)),
        ("stem_frozen_hswish", 3, 16, 3, 2, 1, FrozenBatchNorm2d, torch.nn.Hardswish, (32, 32)),
        ("dw_s1_bn_relu6", 24, 24, 3, 1, 24, torch.nn.BatchNorm2d, torch.nn.ReLU6, (14, 15)),
        ("dw_s2_frozen_relu", 40, 40, 3, 2, 40, FrozenBatchNorm2d, torch.nn.ReLU, (15, 17)),
        ("pw_bn_relu6", 16, 96, 1, 1, 1, torch.nn.BatchNorm2d, torch.nn.ReLU6, (9, 9)),
        ("pw_frozen_linear", 96, 24, 1, 1, 1, FrozenBatchNorm2d, None, (7, 7)),
        ("pw_bn_silu", 20, 50, 1, 1, 1, torch.nn.BatchNorm2d, torch.nn.SiLU, (6, 10)),
    ]
    for k, (name, cin, cout, ks, st, g, norm, act, hw) in enumerate(blocks):
        torch.manual_seed(100 + k)
        blk = Conv2dNormActivation(cin, cout, kernel_size=ks, stride=st, groups=g, norm_layer=norm, activation_layer=act).eval()
        randomize_norms(blk, 200 + k)
        x = philox_f32(800 + k, (2, cin) + hw) * 4 - 2
        with torch.no_grad():
            y = blk(t(x))
        out[f"{name}__x"], out[f"{name}__y"], out[f"{name}__w"] = x, y.numpy(), blk[0].weight.detach().numpy()
        n = blk[1]
        out[f"{name}__norm"] = np.stack([n.weight.detach().numpy(), n.bias.detach().numpy(), n.running_mean.numpy(), n.running_var.numpy()])
        index.append(name)
    out["index"] = np.array(index)
    # the whole network: parameters come from the seed (torch.manual_seed(0) + the reference's constructor), not stored
    torch.manual_seed(0)
    model = mobilenet_v2(num_classes=10).eval()
    randomize_norms(model, 7)
    x = philox_f32(900, (2, 3, 64, 64))
    out["net__x"] = x
    with torch.no_grad():
        a = t(x)
        for i, layer in enumerate(model.features):
            a = layer(a)
            if i in (0, 1, 3, 7, 14):
                out[f"net__features{i}"] = a[:, ::max(1, a.shape[1] // 8), ::2, ::2].numpy()  # strided sample
        out["net__features"] = a.numpy()
        out["net__logits"] = model(t(x)).numpy()
    out["net__checksum"] = np.array([float(sum(p.double().sum() for p in model.parameters()))])
    save("mobilenet_v2", **out)


def gen_deform():
    """the reference's own oracle for deform_conv2d, TestDeformConv.expected_fn (test/test_ops.py:933-980,
    pure Python over torch float64 tensors), run on seeded inputs: the test's own configuration (6 -> 2 channels, 2
    weight groups, 3 offset groups, kernel (3, 2), stride (2, 1), padding (1, 0), dilation (2, 1), input 5 x 4) and a
    few more.  (The native operator itself needs the unbuilt _C extension.)"""
    # test/test_ops.py cannot be imported as a module (its module level touches torch.ops.torchvision.roi_pool, which
    # needs the unbuilt _C extension): compile only the two pure-Python functions out of the reference's file, in memory
    import ast
    import math
    from torch.nn.modules.utils import _pair
    tree = ast.parse((REF / "test" / "test_ops.py").read_text())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "bilinear_interpolate"]
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TestDeformConv")
    wanted += [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "expected_fn"]
    ns = {"math": math, "torch": torch, "_pair": _pair}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), "reference:test/test_ops.py", "exec"), ns)

    class tester:  # expected_fn takes `self` but does not use it
        expected_fn = staticmethod(lambda *a, **k: ns["expected_fn"](None, *a, **k))
    out, index = {}, []
    cases = [  # name, n, cin, cout, (h, w), (kh, kw), stride, pad, dil, groups, offset groups, mask, bias
        ("ref_test_cfg", 3, 6, 2, (5, 4), (3, 2), (2, 1), (1, 0), (2, 1), 2, 3, True, True),
        ("ref_test_cfg_nomask", 2, 6, 2, (5, 4), (3, 2), (2, 1), (1, 0), (2, 1), 2, 3, False, True),
        ("k3_same", 2, 4, 5, (7, 6), (3, 3), (1, 1), (1, 1), (1, 1), 1, 1, True, True),
        ("k3_s2_og2", 1, 4, 3, (9, 8), (3, 3), (2, 2), (1, 1), (1, 1), 1, 2, True, False),
        ("k1", 2, 3, 4, (5, 5), (1, 1), (1, 1), (0, 0), (1, 1), 1, 1, False, True),
        ("big_offsets", 1, 2, 2, (6, 6), (3, 3), (1, 1), (1, 1), (1, 1), 2, 1, True, True),
    ]
    for k, (name, n, cin, cout, (h, w), (kh, kw), st, pd, dl, g, og, use_mask, use_bias) in enumerate(cases):
        oh = (h + 2 * pd[0] - (dl[0] * (kh - 1) + 1)) // st[0] + 1
        ow = (w + 2 * pd[1] - (dl[1] * (kw - 1) + 1)) // st[1] + 1
        rng = np.random.Generator(np.random.Philox(950 + k))
        x = rng.random((n, cin, h, w), dtype=np.float32)
        scale = 4.0 if name == "big_offsets" else 1.0
        offset = (rng.standard_normal((n, og * 2 * kh * kw, oh, ow)) * scale).astype(np.float32)
        mask = rng.standard_normal((n, og * kh * kw, oh, ow)).astype(np.float32) if use_mask else None
        weight = rng.standard_normal((cout, cin // g, kh, kw)).astype(np.float32)
        bias = rng.standard_normal(cout).astype(np.float32) if use_bias else np.zeros(cout, np.float32)
        d = lambda a: None if a is None else t(a).double()  # noqa: E731
        want = tester.expected_fn(d(x), d(weight), d(offset), d(mask), d(bias), stride=st, padding=pd, dilation=dl)
        out[f"{name}__x"], out[f"{name}__offset"], out[f"{name}__weight"], out[f"{name}__bias"] = x, offset, weight, bias
        if use_mask:
            out[f"{name}__mask"] = mask
        out[f"{name}__expected_f64"] = want.numpy()
        out[f"{name}__args"] = np.array(list(st) + list(pd) + list(dl) + [int(use_bias)], np.int64)
        index.append(name)
    out["index"] = np.array(index)
    save("deform_conv2d", **out)


def gen_alexnet():
    """alexnet(num_classes=50), seed 0, eval, on torch.rand(1,3,224,224) -- test_models.py:674-693 -- plus the reference's
    own expect file.  Weights come from the seed (the consumer rebuilds them with the same constructor sequence)."""
    from torchvision.models import alexnet
    out = {}
    torch.manual_seed(0)
    model = alexnet(num_classes=50).eval()
    x = torch.rand(1, 3, 224, 224)
    with torch.no_grad():
        out["y"] = model(x).numpy()
        out["x_checksum"] = np.array([float(x.double().sum())])  # x = the seeded stream right after construction; not stored
        out["features_0_3"] = model.features[0:3](x).numpy()[:, :8]   # conv 11x11 s4 + relu + maxpool(3,2): 8 of 64 channels
        out["features_0_6"] = model.features[0:6](x).numpy()[:, :8]   # ... + conv 5x5 + relu + maxpool
        out["features"] = model.features(x).numpy()
    out["checksum"] = np.array([float(sum(p.double().sum() for p in model.parameters()))])
    exp = torch.load(REF / "test/expect/ModelTester.test_alexnet_expect.pkl", weights_only=True)
    out["reference_expect_pkl"] = exp.detach().numpy()
    save("alexnet_forward", **out)


# ----------------------------------------------------------------------------- round 2: float64, PIL, ElasticTransform, K > 63
def philox_f64(seed: int, shape) -> np.ndarray:
    return np.random.Generator(np.random.Philox(seed)).random(shape)


def gen_round2():
    from PIL import Image
    from torchvision.transforms import v2
    out = {}
    # float64 images: the reference builds the taps and convolves in float64 (_misc.py:139-155)
    idx = []
    seed = 900
    for ks, sg in [([3, 3], None), ([5, 3], [1.1, 0.6]), ([23, 23], [1.7, 1.7]), ([7, 7], [2.0, 2.0])]:
        for shp in [(3, 37, 53), (2, 1, 3, 27, 31)]:
            seed += 1
            x = philox_f64(seed, shp)
            name = f"f64_k{ks[0]}x{ks[1]}_{'x'.join(map(str, shp))}"
            out[name + "__x"] = x
            out[name + "__y_v2"] = F.gaussian_blur_image(t(x), kernel_size=ks, sigma=sg).numpy()
            sg1 = sg if sg is not None else [k * 0.15 + 0.35 for k in ks]
            if len(shp) <= 4:
                out[name + "__y_v1"] = F_t.gaussian_blur(t(x), ks, sg1).numpy()
            out[name + "__ks"] = np.array(ks)
            out[name + "__sigma"] = np.array(sg1)
            idx.append(name)
    out["f64_blur_index"] = np.array(idx)
    x = philox_f64(950, (3, 21, 34))
    out["f64_sharp__x"] = x
    for f in [0.0, 0.4, 1.0, 2.3]:
        out[f"f64_sharp__y_v2_{f}"] = F.adjust_sharpness_image(t(x), sharpness_factor=f).numpy()
        out[f"f64_sharp__y_v1_{f}"] = F_t.adjust_sharpness(t(x), f).numpy()
    # kernel sides above 63 (ElasticTransform with sigma >= 8 asks for int(8 * sigma + 1) | 1 taps)
    xk = philox_f32(960, (1, 80, 96))
    out["k65__x"] = xk
    out["k65__y"] = F.gaussian_blur_image(t(xk), kernel_size=[65, 65], sigma=[9.0, 9.0]).numpy()
    xku = philox_u8(961, (1, 80, 96))
    out["k65u8__x"] = xku
    out["k65u8__y"] = F.gaussian_blur_image(t(xku), kernel_size=[65, 65], sigma=[9.0, 9.0]).numpy()
    # ElasticTransform._get_params (v2/_geometry.py:1054-1075), seeded host generator
    for tag, alpha, sigma, hw in [("default", 50.0, 5.0, (48, 64)), ("aniso", (30.0, 60.0), (2.0, 3.0), (33, 41)),
                                  ("sigma9", 40.0, 9.0, (80, 96)), ("nosmooth", 10.0, 0.0, (9, 12))]:
        torch.manual_seed(1234)
        tr = v2.ElasticTransform(alpha=alpha, sigma=sigma)
        params = tr._get_params([torch.zeros(3, *hw)])
        out[f"elastic_{tag}__displacement"] = params["displacement"].numpy()
        out[f"elastic_{tag}__alpha"] = np.array(tr.alpha)
        out[f"elastic_{tag}__sigma"] = np.array(tr.sigma)
        out[f"elastic_{tag}__hw"] = np.array(hw)
    # PIL entry points (_misc.py:169-174 for the blur; PIL's own ImageEnhance.Sharpness for adjust_sharpness, _color.py:283)
    for mode, c in [("RGB", 3), ("L", 1), ("RGBA", 4)]:
        arr = philox_u8(970 + c, (29, 43, c))
        img = Image.fromarray(arr[:, :, 0] if c == 1 else arr, mode=mode)
        out[f"pil_{mode}__x"] = arr
        out[f"pil_{mode}__blur_5x3"] = np.asarray(F.gaussian_blur(img, kernel_size=[5, 3], sigma=[1.2, 0.7]))
        out[f"pil_{mode}__blur_9"] = np.asarray(F.gaussian_blur(img, kernel_size=[9, 9]))
        for f in [0.0, 0.5, 1.7]:
            out[f"pil_{mode}__sharp_{f}"] = np.asarray(F.adjust_sharpness(img, sharpness_factor=f))
    # the classification preset's steps as v2 transforms, composed with the blur (v2/_geometry.py:76-193, v2/_misc.py:134-304,
    # v2/_container.py:10-64).  float32 input: the interpolation the reference runs on a device tensor of any dtype
    # (_geometry.py:222-254 casts uint8 to float32 there; on the CPU it would take ATen's uint8 kernel instead)
    xf = philox_f32(980, (3, 75, 100))
    pipe = v2.Compose([v2.Resize(40), v2.CenterCrop(32), v2.ToDtype(torch.float32, scale=True),
                       v2.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225]), v2.GaussianBlur(3, sigma=(0.9, 0.9))])
    out["pipe_f32__x"] = xf
    out["pipe_f32__y"] = pipe(t(xf)).numpy()
    out["pipe_f32__repr"] = np.array(repr(pipe))
    xu = philox_u8(981, (2, 3, 36, 44))
    pipe_u8 = v2.Compose([v2.CenterCrop((30, 50)), v2.GaussianBlur((5, 3), sigma=(1.1, 1.1)), v2.ToDtype(torch.float32, scale=True),
                          v2.Normalize([0.5, 0.4, 0.3], [0.2, 0.25, 0.3])])
    out["pipe_u8__x"] = xu
    out["pipe_u8__y"] = pipe_u8(t(xu)).numpy()
    save("round2_api", **out)


def gen_round3():
    """Round 3: ImageClassification on PIL input (transforms/_presets.py:54-61).  For a PIL image the reference resizes and
    crops with PIL itself (F.resize / F.center_crop dispatch to PIL), then pil_to_tensor -> convert_image_dtype -> normalize."""
    import PIL.Image
    from torchvision.transforms._presets import ImageClassification
    out = {}
    cases = {"rgb_photo": ((375, 500, 3), "RGB", 224, 256), "gray": ((100, 80), "L", 64, 72), "rgb_small_padded": ((40, 30, 3), "RGB", 48, 32),
             "rgb_portrait": ((200, 150, 3), "RGB", 96, 100)}
    for name, (shape, mode, crop, size) in cases.items():
        arr = philox_u8(3100 + shape[0], shape)
        img = PIL.Image.fromarray(arr, mode=mode)
        mean, std = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)) if mode == "RGB" else ((0.5,), (0.25,))
        y = ImageClassification(crop_size=crop, resize_size=size, mean=mean, std=std)(img)
        out[f"{name}__x"] = arr
        out[f"{name}__y"] = y.numpy()
        out[f"{name}__cfg"] = np.array([crop, size], np.int64)
        out[f"{name}__mean"], out[f"{name}__std"] = np.array(mean, np.float64), np.array(std, np.float64)
    out["index"] = np.array(sorted(cases))
    out["pil_version"] = np.array([PIL.__version__])
    save("round3_preset_pil", **out)


if __name__ == "__main__":
    if "--only-round3" in sys.argv:
        gen_round3()
        sys.exit(0)
    torch.set_num_threads(1)
    if "--only-round2" in sys.argv:
        gen_round2()
        sys.exit(0)
    gen_kernels()
    gen_opencv()
    gen_blur()
    gen_sharpness()
    gen_primitive()
    gen_cnn()
    gen_vgg()
    gen_preset()
    gen_resize_preset()
    gen_mobilenet()
    gen_deform()
    gen_alexnet()
    gen_round2()
    (HERE / "PROVENANCE.txt").write_text(
        "Fixtures generated by tests/golden/make_golden.py from the reference at /root/reference\n"
        f"(torchvision {Path(REF / 'version.txt').read_text().strip()}), torch {torch.__version__}, numpy {np.__version__}.\n"
        "opencv_gaussian_blur.npz re-encodes the reference's own test/assets/gaussian_blur_opencv_results.pt.\n"
    )
