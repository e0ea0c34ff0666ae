"""BASELINE.json's configurations at FULL size on the MI355X (`-m gpu`): direct comparison with the oracle where it
finishes in seconds, size-independent properties (batch invariance, linearity, gain on constants, determinism)
where it does not.  Inputs are numpy-Philox U[0,1) as SURVEY.md 8(d) prescribes."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cpu_vision_amd import _lib, functional as F  # noqa: E402
from oracle import ref  # noqa: E402
from tests._util import assert_conv_close, philox_f32  # noqa: E402


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def k1d(k, s):
    return F._get_gaussian_kernel1d(k, s).numpy()


def test_cfg1_box3x3_512():
    x = philox_f32(1000, (1, 1, 512, 512))
    got = F.box_filter(dev(x), 3, "reflect").cpu().numpy()
    np.testing.assert_array_equal(got, ref.box_filter(x, 3))


def test_cfg2_gaussian3x3_1080p():
    x = philox_f32(2000, (1, 3, 1080, 1920))
    got = F.gaussian_blur(dev(x), [3, 3]).cpu().numpy()  # default sigma 0.8
    k = k1d(3, 0.8)
    np.testing.assert_array_equal(got, ref.gaussian_blur(x, k, k))


def test_metric_gaussian3x3_4k_frame_and_lds_tile_variant(monkeypatch):
    x = philox_f32(5000, (3, 2160, 3840))
    k = k1d(3, 0.8)
    want = ref.gaussian_blur(x, k, k)
    xd = dev(x)
    np.testing.assert_array_equal(F.gaussian_blur(xd, [3, 3]).cpu().numpy(), want)
    assert _lib.last_kernel().startswith("k_dwtile<f32,3x3"), _lib.last_kernel()
    # the register-window formulation of the same op must agree bit for bit with the LDS-halo-tile one (forced kernels
    # exist in the tuning build only)
    with _lib.tuning_library():
        monkeypatch.setenv("MV_FORCE_REG3X3", "1")
        np.testing.assert_array_equal(F.gaussian_blur(xd, [3, 3]).cpu().numpy(), want)
        assert _lib.last_kernel() == "k_dw3x3"
        # strip height must not change the result
        for rows in ("8", "20", "2160"):
            monkeypatch.setenv("MV_DW3X3_ROWS", rows)
            np.testing.assert_array_equal(F.gaussian_blur(xd, [3, 3]).cpu().numpy(), want)
    # the product library ignores the environment
    np.testing.assert_array_equal(F.gaussian_blur(xd, [3, 3]).cpu().numpy(), want)
    assert _lib.last_kernel().startswith("k_dwtile<f32,3x3"), _lib.last_kernel()


def test_cfg3_separable5x5_then_sobel_4k():
    x = philox_f32(3000, (1, 3, 2160, 3840))
    k = k1d(5, 1.1)
    gx, gy = F.gaussian_sobel(dev(x), [5, 5], [1.1, 1.1])
    ogx, ogy = ref.gaussian_sobel(x, k, k)
    np.testing.assert_array_equal(gx.cpu().numpy(), ogx)
    np.testing.assert_array_equal(gy.cpu().numpy(), ogy)
    # unfused composition through the public operators gives the same bits
    blur = F.separable_gaussian_blur(dev(x), [5, 5], [1.1, 1.1])
    gx2, gy2 = F.sobel(blur, "reflect")
    assert torch.equal(gx2, gx) and torch.equal(gy2, gy)


def test_cfg4_conv3x3x64_relu_batch256():
    n = 256
    g = torch.Generator(device="cuda").manual_seed(4000)
    x = torch.rand((n, 3, 224, 224), generator=g, device="cuda")
    w = torch.randn((64, 3, 3, 3), generator=g, device="cuda") * (2.0 / (64 * 9)) ** 0.5  # kaiming fan_out (vgg.py:55)
    b = (torch.rand((64,), generator=g, device="cuda") - 0.5) * 0.2
    y = F.conv2d_bias_relu(x, w, b)
    assert y.shape == (n, 64, 224, 224) and float(y.min()) >= 0.0
    wn, bn = w.cpu().numpy(), b.cpu().numpy()
    for i in (0, 101, 255):  # sampled images against the oracle, bit for bit
        np.testing.assert_array_equal(y[i:i + 1].cpu().numpy(), ref.conv3x3_bias_relu(x[i:i + 1].cpu().numpy(), wn, bn))
    # batch invariance + determinism
    assert torch.equal(F.conv2d_bias_relu(x[7:9], w, b), y[7:9])
    assert torch.equal(F.conv2d_bias_relu(x, w, b), y)
    # b = 0 run (the vgg initialisation)
    y0 = F.conv2d_bias_relu(x[:4], w, None)
    np.testing.assert_array_equal(y0.cpu().numpy(), ref.conv3x3_bias_relu(x[:4].cpu().numpy(), wn, None))


def test_cfg5_batch_of_4k_frames_properties():
    """A 16-frame slice of cfg5's per-GPU shard (1.6 GB in, 1.6 GB out) -- too big to push through the oracle
    whole, so: sampled frames bit-exact vs the oracle, batch invariance, linearity, unit gain on constants."""
    n = 16
    g = torch.Generator(device="cuda").manual_seed(5000)
    x = torch.rand((n, 3, 2160, 3840), generator=g, device="cuda")
    y = F.gaussian_blur(x, [3, 3])
    k = k1d(3, 0.8)
    for i in (0, 9, 15):
        np.testing.assert_array_equal(y[i].cpu().numpy(), ref.gaussian_blur(x[i].cpu().numpy(), k, k))
    assert torch.equal(F.gaussian_blur(x[5], [3, 3]), y[5])                      # batch invariance
    x2 = torch.rand((2, 3, 2160, 3840), generator=g, device="cuda")
    lhs = F.gaussian_blur(0.25 * x[:2] + 0.5 * x2, [3, 3])
    rhs = 0.25 * y[:2] + 0.5 * F.gaussian_blur(x2, [3, 3])
    assert float((lhs - rhs).abs().max()) <= 2e-6                                # linearity
    c = torch.full((1, 3, 2160, 3840), 0.7311, device="cuda")
    assert float((F.gaussian_blur(c, [3, 3]) - 0.7311).abs().max()) <= 2e-7      # sum of taps == 1
    assert float((y.double().mean() - x.double().mean()).abs()) <= 1e-6          # mean preserved (reflect ~ symmetric)


def _one_launch_blur(x):
    """One mv_gaussian_blur_f32 launch over the whole batch, through the C ABI (what bench.py times)."""
    lib = _lib.load()
    y = torch.empty_like(x)
    k1 = F._get_gaussian_kernel1d(3, 0.8)
    tx, ty = _lib.taps_from_tensor(k1), _lib.taps_from_tensor(k1)
    planes = x.shape[0] * x.shape[1]
    _lib.check(lib.mv_gaussian_blur_f32(x.data_ptr(), y.data_ptr(), planes, x.shape[2], x.shape[3], tx, 3, ty, 3,
                                        torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return y


def test_cfg5_real_shard_128_frames_one_launch_high_frames_vs_oracle():
    """The REAL per-GPU shard of cfg5 / the headline launch: 128 frames of 3x2160x3840 fp32 in ONE launch (12.7 GB in,
    12.7 GB out; element offsets up to 3.18e9 > 2^31, byte offsets up to 1.27e10 > 2^32, 777 600 workgroups).  Frames
    from the bottom, the middle, across the 2^31-element and 2^32-byte boundaries and the very last one are compared
    with the C oracle bit for bit; every frame goes through a whole-output property (per-frame mean preserved)."""
    n, c, h, w = 128, 3, 2160, 3840
    g = torch.Generator(device="cuda").manual_seed(5128)
    x = torch.empty((n, c, h, w), dtype=torch.float32, device="cuda")
    for i in range(0, n, 16):
        x[i:i + 16].uniform_(0.0, 1.0, generator=g)
    y = _one_launch_blur(x)
    assert _lib.last_kernel().startswith("k_dwtile<f32,3x3"), _lib.last_kernel()
    k = k1d(3, 0.8)
    per_frame = c * h * w
    f31 = (2 ** 31) // per_frame       # the frame holding element offset 2^31
    f32b = (2 ** 32) // (per_frame * 4)  # the frame holding byte offset 2^32
    for i in sorted({0, f32b, f31, 63, 64, 127}):
        np.testing.assert_array_equal(y[i].cpu().numpy(), ref.gaussian_blur(x[i].cpu().numpy(), k, k), err_msg=f"frame {i}")
    # every frame: the blur of a U[0,1) frame keeps its mean to ~1e-6 (reflect border, taps sum to 1) -- a frame that was
    # skipped, written twice at a wrong offset or left uninitialised would miss by ~0.5
    mx = x.view(n, -1).double().mean(1)
    my = y.view(n, -1).double().mean(1)
    assert float((mx - my).abs().max()) <= 5e-6
    # batch invariance: the same frames as a 2-frame launch
    assert torch.equal(_one_launch_blur(x[126:128].contiguous()), y[126:128])


def test_cfg2_real_batch_96_frames_1080p_one_launch_vs_oracle():
    n, c, h, w = 96, 3, 1080, 1920
    g = torch.Generator(device="cuda").manual_seed(2096)
    x = torch.rand((n, c, h, w), generator=g, device="cuda")
    y = _one_launch_blur(x)
    k = k1d(3, 0.8)
    for i in (0, 47, 95):
        np.testing.assert_array_equal(y[i].cpu().numpy(), ref.gaussian_blur(x[i].cpu().numpy(), k, k), err_msg=f"frame {i}")
    mx = x.view(n, -1).double().mean(1)
    my = y.view(n, -1).double().mean(1)
    assert float((mx - my).abs().max()) <= 1e-5
