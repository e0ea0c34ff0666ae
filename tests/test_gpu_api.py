"""The reference's kernel contract (`check_kernel`, test_transforms_v2.py:154-184) applied to the MI355X kernels:
no in-place mutation, dtype/device preserved, GPU-vs-CPU closeness, batched == unbatched, degenerate batch dims;
plus the transform classes and stream semantics.  `-m gpu` only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cpu_vision_amd as mv  # noqa: E402
from cpu_vision_amd import functional as F, transforms, tv_tensors  # noqa: E402
from oracle import ref_torch  # noqa: E402


def make_image(size=(17, 11), channels=3, batch_dims=(), dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    shape = (*batch_dims, channels, *size)
    if dtype.is_floating_point:
        return torch.rand(shape, generator=g).to(dtype).cuda()
    return torch.randint(0, 256, shape, generator=g, dtype=torch.uint8).to(dtype).cuda()


def check_kernel(kernel, cpu_kernel, inpt, *args, atol=1e-5, **kwargs):
    version = inpt._version
    out = kernel(inpt, *args, **kwargs)
    assert inpt._version == version, "kernel must not modify its input"
    assert out.dtype == inpt.dtype and out.device == inpt.device
    # GPU vs the reference's CPU path (_check_kernel_cuda_vs_cpu)
    want = cpu_kernel(inpt.cpu(), *args, **kwargs)
    if inpt.dtype == torch.uint8:
        assert (out.cpu().int() - want.int()).abs().max() <= 1
    else:
        torch.testing.assert_close(out.cpu(), want, rtol=1e-5, atol=atol)
    # batched vs unbatched
    for batch_dims in [(2,), (2, 1)]:
        rep = [*batch_dims, *[1] * inpt.ndim]
        torch.testing.assert_close(kernel(inpt.repeat(rep), *args, **kwargs), out.repeat(rep), rtol=0, atol=0)
    for dims in [(0,), (5, 0), (0, 5)]:
        e = torch.empty(dims + inpt.shape, dtype=inpt.dtype, device=inpt.device)
        assert kernel(e, *args, **kwargs).shape[: -inpt.ndim] == dims


def _ref_blur(x, kernel_size, sigma=None):
    ks, sg = F._check_gaussian_args(kernel_size, sigma)
    return ref_torch.gaussian_blur_image(x, ks, sg)


@pytest.mark.parametrize("kernel_size", [1, 3, (3, 1), [3, 5]])
@pytest.mark.parametrize("sigma", [None, 1.0, 1, (0.5,), [0.3], (0.3, 0.7), [0.9, 0.2]])
def test_gaussian_blur_kernel_image(kernel_size, sigma):
    """TestGaussianBlur.test_kernel_image, test_transforms_v2.py:3166-3175."""
    check_kernel(F.gaussian_blur_image, _ref_blur, make_image(), kernel_size=kernel_size, sigma=sigma)


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32, torch.float16, torch.float64, torch.int16])
def test_gaussian_blur_dtypes(dtype):
    x = make_image(dtype=dtype, size=(24, 36))
    out = F.gaussian_blur_image(x, [5, 3], [1.0, 0.7])
    assert out.dtype == dtype
    want = ref_torch.gaussian_blur_image(x.cpu().to(torch.float32 if dtype in (torch.float16,) else dtype), [5, 3], [1.0, 0.7])
    tol = {torch.float16: 2e-3, torch.float64: 1e-13}.get(dtype, 1e-5 if dtype.is_floating_point else 1)
    assert (out.cpu().double() - want.double()).abs().max() <= tol


def test_gaussian_blur_video_and_tv_tensor_types():
    v = tv_tensors.Video(make_image(batch_dims=(4,)))
    out = F.gaussian_blur(v, kernel_size=(3, 3))
    assert type(out) is tv_tensors.Video and out.shape == v.shape
    im = tv_tensors.Image(make_image())
    out = F.gaussian_blur(im, [3, 3], [0.8, 0.8])
    assert type(out) is tv_tensors.Image
    torch.testing.assert_close(out.as_subclass(torch.Tensor), F.gaussian_blur_image(im.as_subclass(torch.Tensor), [3, 3], [0.8, 0.8]), rtol=0, atol=0)
    check_kernel(F.gaussian_blur_video, _ref_blur, make_image(batch_dims=(3,)), kernel_size=(3, 3))


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
def test_adjust_sharpness_kernel_image(dtype):
    """TestAdjustSharpness.test_kernel_image, test_transforms_v2.py:4686-4689."""
    check_kernel(F.adjust_sharpness_image, ref_torch.adjust_sharpness_image, make_image(dtype=dtype), sharpness_factor=0.5)
    check_kernel(F.adjust_sharpness_video, ref_torch.adjust_sharpness_image, make_image(dtype=dtype, batch_dims=(2,)),
                 sharpness_factor=1.7)


def test_adjust_sharpness_uint8_exact_vs_reference_cpu_path():
    for f in (0.1, 0.5, 1.0, 2.0):
        x = make_image(dtype=torch.uint8, size=(37, 53), seed=3)
        assert torch.equal(F.adjust_sharpness(x, f).cpu(), ref_torch.adjust_sharpness_image(x.cpu(), f))


def test_non_contiguous_input():
    base = make_image(size=(40, 64))
    x = base[..., ::2, 1:33]  # strided view
    out = F.gaussian_blur_image(x, [3, 3])
    torch.testing.assert_close(out, F.gaussian_blur_image(x.contiguous(), [3, 3]), rtol=0, atol=0)
    xt = base.transpose(-1, -2)
    torch.testing.assert_close(F.adjust_sharpness_image(xt, 0.3), F.adjust_sharpness_image(xt.contiguous(), 0.3), rtol=0, atol=0)


def test_runs_on_the_callers_stream():
    x = make_image(size=(256, 512))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y = x * 2.0  # produced on s; the blur must be ordered after it on the same stream
        out = F.gaussian_blur_image(y, [3, 3], [0.8, 0.8])
    s.synchronize()
    torch.testing.assert_close(out, F.gaussian_blur_image(x * 2.0, [3, 3], [0.8, 0.8]), rtol=0, atol=0)


@pytest.mark.parametrize("sigma", [5, 2.0, (0.5, 2), [1.3, 2.7]])
def test_gaussian_blur_transform(sigma):
    t = transforms.GaussianBlur(kernel_size=3, sigma=sigma)
    img = make_image()
    mask = tv_tensors.Mask(torch.zeros(17, 11, device="cuda"))
    torch.manual_seed(1)
    out = t({"image": tv_tensors.Image(img), "mask": mask})
    assert type(out["image"]) is tv_tensors.Image and out["mask"] is mask
    torch.manual_seed(1)
    s = t._get_params([])["sigma"]
    torch.testing.assert_close(out["image"].as_subclass(torch.Tensor), F.gaussian_blur_image(img, [3, 3], s), rtol=0, atol=0)
    # pure tensor: treated as the image
    torch.manual_seed(1)
    torch.testing.assert_close(t(img), F.gaussian_blur_image(img, [3, 3], s), rtol=0, atol=0)
    v1 = transforms.GaussianBlurV1(3, sigma=(0.5, 2.0))
    torch.manual_seed(2)
    o1 = v1(img)
    torch.manual_seed(2)
    s1 = v1.get_params(0.5, 2.0)
    from cpu_vision_amd import functional_v1 as F1
    torch.testing.assert_close(o1, F1.gaussian_blur(img, [3, 3], [s1, s1]), rtol=0, atol=0)


def test_random_adjust_sharpness_transform():
    img = make_image(dtype=torch.uint8)
    out = transforms.RandomAdjustSharpness(sharpness_factor=0.5, p=1)(img)
    assert torch.equal(out, F.adjust_sharpness_image(img, 0.5))
    vid = tv_tensors.Video(make_image(batch_dims=(2,)))
    assert type(transforms.RandomAdjustSharpness(2.0, p=1)(vid)) is tv_tensors.Video


def test_elastic_transform_style_large_kernel():
    """ElasticTransform._get_params blurs a 1x1xHxW field with k = int(8*sigma+1)|1 (v2/_geometry.py:1054-1075)."""
    sigma = 5.0
    k = int(8 * sigma + 1)
    k += (k % 2 == 0)
    x = torch.rand(1, 1, 64, 80, generator=torch.Generator().manual_seed(0)).cuda() * 2 - 1
    out = F.gaussian_blur(x, [k, k], [sigma, sigma])
    want = ref_torch.gaussian_blur_image(x.cpu(), [k, k], [sigma, sigma])
    torch.testing.assert_close(out.cpu(), want, rtol=1e-5, atol=1e-6)


def test_vgg_first_layer_module_matches_torch_cpu():
    from cpu_vision_amd.nn import Conv3x3ReLU
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(3, 64, 3, padding=1)
    torch.nn.init.kaiming_normal_(conv.weight, mode="fan_out", nonlinearity="relu")
    x = torch.rand(2, 3, 56, 72)
    want = torch.relu(conv(x)).detach()
    got = Conv3x3ReLU.from_conv(conv).cuda()(x.cuda())
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 3, 40, 64), (3, 37, 53), (1, 64, 1023), (1, 9, 130)])
@pytest.mark.parametrize("ks", [(3, 3), (5, 5), (7, 7), (3, 5), (9, 3), (11, 11), (1, 7), (23, 23)])
def test_half_precision_storage_is_the_fp32_path_rounded_once(dtype, shape, ks):
    """fp16 / bf16 images: the fused tile kernel (fp32 arithmetic, one rounding on store) equals
    .to(float32) -> the 2-D pass -> .to(dtype) bit for bit, and the oracle; sides above 11 take the conversion path."""
    import numpy as np
    from oracle import ref
    if ks[0] // 2 >= shape[-1] or ks[1] // 2 >= shape[-2]:
        pytest.skip("reflect padding must be smaller than the image")
    g = torch.Generator().manual_seed(hash((shape, ks)) % 1000)
    x = (torch.rand(shape, generator=g) * 4 - 2).to(dtype).cuda()
    sg = [0.5 + ks[0] / 5.0, 0.5 + ks[1] / 6.0]
    got = F.gaussian_blur_image(x, list(ks), sg)
    assert got.dtype == dtype and got.shape == x.shape
    kx, ky = F._get_gaussian_kernel1d(ks[0], sg[0]).numpy(), F._get_gaussian_kernel1d(ks[1], sg[1]).numpy()
    xf = x.float().cpu().numpy()
    want32 = ref.separable_blur(xf, kx, ky) if max(ks) > 11 else ref.gaussian_blur(xf, kx, ky)
    want = torch.from_numpy(want32).to(dtype)
    assert torch.equal(got.cpu(), want)
