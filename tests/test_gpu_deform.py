"""SURVEY.md 8(f).4 -- torchvision::deform_conv2d forward on the MI355X: against the reference's own test oracle
(TestDeformConv.expected_fn outputs, golden fixture) at the reference's tolerance, and bit for bit against the CPU oracle."""
import contextlib
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import cpu_vision_amd as mv  # noqa: E402
from cpu_vision_amd import ops  # noqa: E402
from oracle import ref  # noqa: E402
from tests._util import golden, philox_f32  # noqa: E402
from tests.test_oracle_golden import _deform_case  # noqa: E402


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("fused", [False, True], ids=["library", "fused"])
def test_vs_reference_expected_fn_and_oracle(fused):
    g = golden("deform_conv2d")
    for name in map(str, g["index"]):
        x, off, w, b, st, pd, dl, mask = _deform_case(g, name)
        with (_fused() if fused else contextlib.nullcontext()):
            got = host(ops.deform_conv2d(dev(x), dev(off), dev(w), dev(b), stride=st, padding=pd, dilation=dl, mask=dev(mask)))
        np.testing.assert_allclose(got, g[f"{name}__expected_f64"], rtol=1e-5, atol=1e-5, err_msg=f"{name} vs the reference's expected_fn")
        np.testing.assert_array_equal(got, ref.deform_conv2d(x, off, w, b, st, pd, dl, mask), err_msg=f"{name} vs oracle")


@contextlib.contextmanager
def _unfused(direct=False):
    """The columns-workspace form for geometries the product library runs fused: tuning build + MV_DEFORM_UNFUSED."""
    from cpu_vision_amd import _lib
    with _lib.tuning_library():
        os.environ["MV_DEFORM_UNFUSED"] = "1"
        if direct:
            os.environ["MV_DEFORM_DIRECT"] = "1"
        try:
            yield
        finally:
            os.environ.pop("MV_DEFORM_UNFUSED")
            os.environ.pop("MV_DEFORM_DIRECT", None)


@contextlib.contextmanager
def _fused():
    """The fused kernel wherever its tiles fit, also for launches so small that the library would take the optional workspace."""
    old, ops.FUSED_WHENEVER_POSSIBLE = ops.FUSED_WHENEVER_POSSIBLE, True
    try:
        yield
    finally:
        ops.FUSED_WHENEVER_POSSIBLE = old


def _case(seed, n, cin, cout, h, w, kh, kw, st, pd, dl, groups, og, use_mask, use_bias, scale=1.5):
    rng = np.random.Generator(np.random.Philox(seed))
    oh = (h + 2 * pd[0] - (dl[0] * (kh - 1) + 1)) // st[0] + 1
    ow = (w + 2 * pd[1] - (dl[1] * (kw - 1) + 1)) // st[1] + 1
    x = rng.random((n, cin, h, w), dtype=np.float32) * 2 - 1
    off = (rng.standard_normal((n, og * 2 * kh * kw, oh, ow)) * scale).astype(np.float32)
    mask = rng.random((n, og * kh * kw, oh, ow), dtype=np.float32) if use_mask else None
    wt = ((rng.random((cout, cin // groups, kh, kw), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    b = (rng.random(cout, dtype=np.float32) - 0.5) if use_bias else None
    return x, off, wt, b, mask


@pytest.mark.parametrize("n,cin,cout,h,w,k,st,pd,dl,groups,og,use_mask,use_bias", [
    (2, 16, 32, 20, 24, (3, 3), (1, 1), (1, 1), (1, 1), 1, 1, True, True),      # DCNv2 block
    (3, 6, 2, 5, 4, (3, 2), (2, 1), (1, 0), (2, 1), 2, 3, True, True),          # the reference test's configuration
    (1, 64, 64, 28, 28, (3, 3), (1, 1), (1, 1), (1, 1), 1, 4, False, False),    # DCNv1, 4 offset groups
    (2, 8, 12, 17, 13, (5, 3), (2, 2), (2, 1), (1, 1), 4, 2, True, False),
    (1, 3, 5, 7, 7, (1, 1), (1, 1), (0, 0), (1, 1), 1, 1, True, True),
    (2, 4, 4, 9, 31, (3, 3), (1, 2), (1, 1), (1, 3), 2, 1, False, True),
    (1, 2, 3, 3, 3, (3, 3), (1, 1), (0, 0), (1, 1), 1, 1, True, True),          # single output pixel
    (2, 16, 40, 20, 24, (3, 3), (1, 1), (1, 1), (1, 1), 1, 2, True, True),      # 64-channel workgroups, 4-channel chunks
    (1, 32, 130, 19, 21, (3, 3), (1, 1), (1, 1), (1, 1), 1, 4, True, False),    # two channel blocks, the K walk crosses offset groups
    (1, 24, 72, 33, 18, (3, 3), (2, 2), (1, 1), (1, 1), 3, 1, False, True),     # weight groups, stride 2
    (2, 8, 6, 16, 16, (7, 7), (1, 1), (3, 3), (1, 1), 1, 2, True, True),        # 49 taps: columns workspace
    (1, 5, 3, 9, 40, (1, 5), (1, 1), (0, 2), (1, 1), 1, 5, True, True),         # odd K per chunk (one channel x 5 taps)
    (2, 16, 200, 13, 21, (3, 3), (1, 1), (1, 1), (1, 1), 1, 2, True, True),     # > 128 output channels: 256-channel x 64-pixel tiles
    (1, 6, 160, 9, 18, (3, 2), (1, 1), (1, 0), (1, 1), 1, 3, True, False),      # the same tiles, run-time taps (2-channel chunks)
    (1, 8, 300, 6, 33, (3, 3), (1, 1), (1, 1), (1, 1), 1, 1, False, True),      # two 256-channel blocks, the second mostly empty
])
@pytest.mark.parametrize("path", [contextlib.nullcontext, _fused, _unfused], ids=["library", "fused", "columns"])
def test_bit_exact_vs_oracle(n, cin, cout, h, w, k, st, pd, dl, groups, og, use_mask, use_bias, path):
    """Every geometry through the library's choice (the fused kernel wherever its tiles fit) and through the columns-workspace
    form (tuning build, MV_DEFORM_UNFUSED): the same chain per output, so both equal the oracle bit for bit."""
    x, off, wt, b, mask = _case(11000 + cin * 7 + h, n, cin, cout, h, w, k[0], k[1], st, pd, dl, groups, og, use_mask, use_bias)
    with path():
        got = host(ops.deform_conv2d(dev(x), dev(off), dev(wt), dev(b), stride=st, padding=pd, dilation=dl, mask=dev(mask)))
    np.testing.assert_array_equal(got, ref.deform_conv2d(x, off, wt, b, st, pd, dl, mask))


def test_which_kernel_runs():
    """mv_deform_conv2d_needs_workspace() and mv_last_kernel(): 3x3 / 5x3 / 1x1 layers run the fused kernel without a
    workspace; 7x7 (49 taps) needs the columns workspace and says so when it is missing."""
    from cpu_vision_amd import _lib
    lib = mv.load_library()
    assert lib.mv_deform_conv2d_needs_workspace(8, 256, 256, 64, 64, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1) == 0
    assert lib.mv_deform_conv2d_needs_workspace(1, 256, 256, 64, 64, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1) == 2    # 64 workgroups: optional
    assert lib.mv_deform_conv2d_needs_workspace(3, 6, 2, 5, 4, 3, 2, 2, 1, 1, 0, 2, 1, 2, 3) == 2
    assert lib.mv_deform_conv2d_needs_workspace(64, 8, 8, 20, 20, 7, 7, 1, 1, 3, 3, 1, 1, 1, 1) == 1
    assert lib.mv_deform_conv2d_needs_workspace(64, 8, 8, 200, 200, 3, 3, 9, 9, 1, 1, 1, 1, 1, 1) == 1   # stride 9: the window outgrows LDS
    assert lib.mv_deform_conv2d_needs_workspace(64, 7, 8, 20, 20, 3, 3, 1, 1, 1, 1, 1, 1, 2, 1) == 1     # bad channel split
    x, off, wt, b, mask = _case(11210, 1, 16, 8, 12, 12, 3, 3, (1, 1), (1, 1), (1, 1), 1, 1, True, True)
    ops.deform_conv2d(dev(x), dev(off), dev(wt), dev(b), padding=(1, 1), mask=dev(mask))
    assert _lib.last_kernel().startswith("k_conv1x1")  # one workgroup's worth: the optional workspace is taken
    with _fused():
        ops.deform_conv2d(dev(x), dev(off), dev(wt), dev(b), padding=(1, 1), mask=dev(mask))
    assert _lib.last_kernel() == "k_deform_fused<1,3x3,cb4>"
    x, off, wt, b, mask = _case(11211, 1, 6, 70, 12, 12, 3, 2, (1, 1), (1, 1), (1, 1), 1, 3, True, True)
    with _fused():
        ops.deform_conv2d(dev(x), dev(off), dev(wt), dev(b), padding=(1, 1), mask=dev(mask))
    assert _lib.last_kernel() == "k_deform_fused<4,taps6,cb2>"
    x, off, wt, b, mask = _case(11213, 1, 8, 256, 12, 12, 3, 3, (1, 1), (1, 1), (1, 1), 1, 1, True, True)
    with _fused():
        ops.deform_conv2d(dev(x), dev(off), dev(wt), dev(b), padding=(1, 1), mask=dev(mask))
    assert _lib.last_kernel() == "k_deform_fused<8,3x3,cb4>"
    x, off, wt, b, mask = _case(11212, 1, 4, 4, 12, 12, 7, 7, (1, 1), (3, 3), (1, 1), 1, 1, True, True)
    got = host(ops.deform_conv2d(dev(x), dev(off), dev(wt), dev(b), padding=(3, 3), mask=dev(mask)))
    assert _lib.last_kernel().startswith("k_conv1x1")
    np.testing.assert_array_equal(got, ref.deform_conv2d(x, off, wt, b, (1, 1), (3, 3), (1, 1), mask))


@pytest.mark.parametrize("scale", [0.0, 3.0, 12.0, 60.0])
def test_offsets_inside_and_far_outside_the_staged_window(scale):
    """k_deform_fused (and k_deform_im2col_lds) serve corners from an LDS window that covers |offset| <= 6; larger offsets (and
    samples far outside the image) take the global path per (tap, pixel).  Both paths in one launch, bit-exact against the
    oracle; plus the two columns-workspace kernels (tuning build) on the same inputs."""
    from cpu_vision_amd import _lib
    x, off, wt, b, mask = _case(11300 + int(scale), 2, 12, 10, 37, 45, 3, 3, (1, 1), (1, 1), (1, 1), 1, 2, True, True, scale=scale)
    want = ref.deform_conv2d(x, off, wt, b, (1, 1), (1, 1), (1, 1), mask)
    x2, off2, wt2, b2, m2 = _case(11400 + int(scale), 1, 8, 4, 30, 41, 3, 5, (2, 1), (2, 3), (2, 1), 2, 4, False, True, scale=scale)
    with _fused():
        got = host(ops.deform_conv2d(dev(x), dev(off), dev(wt), dev(b), padding=(1, 1), mask=dev(mask)))
        assert _lib.last_kernel().startswith("k_deform_fused")
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(host(ops.deform_conv2d(dev(x2), dev(off2), dev(wt2), dev(b2), stride=(2, 1), padding=(2, 3), dilation=(2, 1))),
                                      ref.deform_conv2d(x2, off2, wt2, b2, (2, 1), (2, 3), (2, 1), None))
    for direct in (False, True):  # k_deform_im2col_lds, then the direct-gather kernel, + GEMM
        with _unfused(direct):
            np.testing.assert_array_equal(host(ops.deform_conv2d(dev(x), dev(off), dev(wt), dev(b), padding=(1, 1), mask=dev(mask))), want)
            assert _lib.last_kernel().startswith("k_conv1x1")


def test_samples_exactly_on_the_outside_boundary_next_to_non_finite_pixels():
    """h == -1 / w == -1 exactly (zero offsets + padding 1 put the first kernel row / column there) must return 0
    like the reference's `outside` test (deform_conv2d_kernel.cpp:88-90) and not 0 x (border pixel) from the LDS window -- which is
    NaN when that pixel is inf / NaN.  Fused kernel, the library's choice, and both columns kernels: all equal the oracle, whose
    non-finite pattern equals the reference's native kernel (tests/_ref_deform_worker.py, last case)."""
    rng = np.random.Generator(np.random.Philox(11500))
    n, cin, cout, h, w = 1, 4, 6, 9, 11
    x = rng.random((n, cin, h, w), dtype=np.float32) * 2 - 1
    x[0, 0, 0, 0], x[0, 1, 0, 5], x[0, 2, 4, 0], x[0, 3, h - 1, w - 1] = np.inf, -np.inf, np.inf, np.nan
    off = np.zeros((n, 18, h, w), np.float32)
    off[0, :, 4:, :] = (rng.standard_normal((18, h - 4, w)) * 0.7).astype(np.float32)
    off[0, 0, 6, 3] = -6.0  # h = -1 exactly through a non-zero offset as well (row 6, tap (0, 0): 6 - 1 + 0 - 6)
    wt = ((rng.random((cout, cin, 3, 3), dtype=np.float32) - 0.5) * 0.5).astype(np.float32)
    want = ref.deform_conv2d(x, off, wt, None, (1, 1), (1, 1), (1, 1), None)
    assert np.isfinite(want).sum() > 0 and (~np.isfinite(want)).sum() > 0
    assert np.isfinite(want[0, :, 0, 2]).all()  # top row, away from the non-finite pixels: its h == -1 samples contribute exactly 0
    paths = [contextlib.nullcontext(), _fused(), _unfused(False), _unfused(True)]
    for path in paths:
        with path:
            got = host(ops.deform_conv2d(dev(x), dev(off), dev(wt), None, padding=(1, 1)))
        np.testing.assert_array_equal(got, want)  # NaNs compare equal position by position


def test_zero_offsets_equal_the_conv_kernels_and_passes_split_the_batch(monkeypatch):
    """Zero offsets, no mask = conv2d: equal to the oracle's conv bit for bit.  A workspace that holds one image at a time
    (several passes) gives the same result as one pass."""
    x = philox_f32(11100, (5, 3, 12, 10)) * 2 - 1
    w = (philox_f32(11101, (7, 3, 3, 3)) - 0.5)
    b = philox_f32(11102, (7,))
    off = np.zeros((5, 18, 12, 10), np.float32)
    for path in (contextlib.nullcontext, _fused):
        with path():
            got = host(ops.deform_conv2d(dev(x), dev(off), dev(w), dev(b), padding=(1, 1)))
        np.testing.assert_array_equal(got, ref.conv2d_affine_act(x, w, b, None, None, None, 1, 1, 1, 0, None))
    x2, off2, wt2, b2, m2 = _case(11110, 5, 8, 6, 9, 9, 3, 3, (1, 1), (1, 1), (1, 1), 1, 1, True, True)
    one = host(ops.deform_conv2d(dev(x2), dev(off2), dev(wt2), dev(b2), padding=(1, 1), mask=dev(m2)))
    monkeypatch.setattr(ops, "MAX_WORKSPACE_BYTES", 8 * 9 * 81 * 4 * 2)  # two images per pass
    with _unfused():
        np.testing.assert_array_equal(host(ops.deform_conv2d(dev(x2), dev(off2), dev(wt2), dev(b2), padding=(1, 1), mask=dev(m2))), one)


def test_module_and_dispatcher_registration():
    """DeformConv2d mirrors the reference's module (parameters, init stream, repr); register_torchvision_op() makes
    torch.ops.torchvision.deform_conv2d -- the operator the reference's Python calls -- run this kernel for device tensors."""
    torch.manual_seed(3)
    layer = ops.DeformConv2d(6, 2, (3, 2), stride=(2, 1), padding=(1, 0), dilation=(2, 1), groups=2)
    assert repr(layer) == "DeformConv2d(6, 2, kernel_size=(3, 2), stride=(2, 1), padding=(1, 0), dilation=(2, 1), groups=2)"
    assert set(layer.state_dict()) == {"weight", "bias"} and layer.weight.shape == (2, 3, 3, 2)
    x, off, _, _, mask = _case(11200, 4, 6, 2, 5, 4, 3, 2, (2, 1), (1, 0), (2, 1), 2, 3, True, True)
    want = ref.deform_conv2d(x, off, layer.weight.detach().numpy(), layer.bias.detach().numpy(), (2, 1), (1, 0), (2, 1), mask)
    layer = layer.cuda()
    np.testing.assert_array_equal(host(layer(dev(x), dev(off), dev(mask))), want)
    ops.register_torchvision_op()
    ops.register_torchvision_op()  # idempotent
    got = torch.ops.torchvision.deform_conv2d(dev(x), layer.weight, dev(off), dev(mask), layer.bias, 2, 1, 1, 0, 2, 1, 2, 3, True)
    np.testing.assert_array_equal(host(got), want)
    # the reference's calling convention without a mask: a zero-sized placeholder and use_mask = False
    placeholder = torch.zeros((4, 1), device="cuda")
    got = torch.ops.torchvision.deform_conv2d(dev(x), layer.weight, dev(off), placeholder, layer.bias, 2, 1, 1, 0, 2, 1, 2, 3, False)
    np.testing.assert_array_equal(host(got), ref.deform_conv2d(x, off, host(layer.weight), host(layer.bias), (2, 1), (1, 0), (2, 1), None))


def test_errors_match_the_reference_messages():
    x = torch.zeros((1, 4, 8, 8), device="cuda")
    w = torch.zeros((2, 4, 3, 3), device="cuda")
    with pytest.raises(RuntimeError, match="offset.shape\\[1\\] is not valid: got: 20 expected: 18"):
        ops._deform_conv2d_impl(x, w, torch.zeros((1, 20, 6, 6), device="cuda"), None, None, 1, 1, 0, 0, 1, 1, 1, 1, False)
    with pytest.raises(RuntimeError, match="the shape of the offset tensor at dimension 1 is not valid"):
        ops.deform_conv2d(x, torch.zeros((1, 9, 6, 6), device="cuda"), w)
    with pytest.raises(RuntimeError, match="offset output dims: \\(5, 6\\) - computed output dims: \\(6, 6\\)"):
        ops.deform_conv2d(x, torch.zeros((1, 18, 5, 6), device="cuda"), w)
    with pytest.raises(RuntimeError, match="invalid batch size of offset"):
        ops.deform_conv2d(x, torch.zeros((2, 18, 6, 6), device="cuda"), w)
    with pytest.raises(RuntimeError, match="mask.shape\\[1\\] is not valid"):
        ops.deform_conv2d(x, torch.zeros((1, 18, 6, 6), device="cuda"), w, mask=torch.zeros((1, 8, 6, 6), device="cuda"))
    with pytest.raises(mv.Mi355VisionError):
        ops.deform_conv2d(x.cpu(), torch.zeros((1, 18, 6, 6)), w.cpu())
    assert ops.deform_conv2d(x[:0], torch.zeros((0, 18, 6, 6), device="cuda"), w).shape == (0, 2, 6, 6)
    lib = mv.load_library()
    assert lib.mv_deform_conv2d_workspace_bytes(2, 4, 8, 8, 3, 3, 1, 1, 0, 0, 1, 1) == 2 * 4 * 9 * 36 * 4
    y = torch.empty((1, 2, 6, 6), device="cuda")
    off = torch.zeros((1, 18, 6, 6), device="cuda")
    assert lib.mv_deform_conv2d_f32(x.data_ptr(), w.data_ptr(), off.data_ptr(), None, None, y.data_ptr(), 1, 4, 8, 8, 2, 3, 3, 1, 1, 0, 0,
                                    1, 1, 1, 1, 0, None, 0, None) == 0  # fused: no workspace needed
    w7 = torch.zeros((2, 4, 7, 7), device="cuda")
    y7, off7 = torch.empty((1, 2, 2, 2), device="cuda"), torch.zeros((1, 98, 2, 2), device="cuda")
    assert lib.mv_deform_conv2d_f32(x.data_ptr(), w7.data_ptr(), off7.data_ptr(), None, None, y7.data_ptr(), 1, 4, 8, 8, 2, 7, 7, 1, 1, 0, 0,
                                    1, 1, 1, 1, 0, None, 0, None) == -1 and b"workspace" in lib.mv_last_error()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-5), (torch.float16, 2e-3), (torch.bfloat16, 2e-2)])
def test_reference_test_configuration_in_its_own_dtypes(dtype, tol):
    """TestDeformConv.test_forward (test/test_ops.py:1054-1080) feeds float64 (class attribute dtype, :931) and, on the GPU,
    float16 tensors through torch.ops.torchvision.deform_conv2d and compares with expected_fn at tol = 1e-5 (2e-3 for half).
    Same configuration (6 -> 2 channels, 2 weight groups, 3 offset groups, kernel (3, 2), stride (2, 1), padding (1, 0),
    dilation (2, 1), batch 4), same operator entry, the reference's stored expected_fn outputs."""
    g = golden("deform_conv2d")
    name = str(g["index"][0])
    x, off, w, b, st, pd, dl, mask = _deform_case(g, name)
    want = g[f"{name}__expected_f64"]
    ops.register_torchvision_op()
    t = lambda a: None if a is None else dev(a).to(dtype)  # noqa: E731
    n_w, n_o = x.shape[1] // w.shape[1], off.shape[1] // (2 * w.shape[2] * w.shape[3])
    use_mask = mask is not None
    mk = t(mask) if use_mask else torch.zeros((x.shape[0], 1), device="cuda", dtype=dtype)
    got = torch.ops.torchvision.deform_conv2d(t(x), t(w), t(off), mk, t(b), st[0], st[1], pd[0], pd[1], dl[0], dl[1], n_w, n_o, use_mask)
    assert got.dtype == dtype
    torch.testing.assert_close(got.double().cpu(), torch.from_numpy(want).double(), rtol=tol, atol=tol)
    got2 = ops.deform_conv2d(t(x), t(off), t(w), t(b), stride=st, padding=pd, dilation=dl, mask=t(mask) if use_mask else None)
    assert got2.dtype == dtype and torch.equal(got2, got)
    with pytest.raises(RuntimeError, match="floating-point"):
        ops.deform_conv2d(dev(x).to(torch.int32), dev(off), dev(w))


def test_forward_only_results_raise_on_backward_instead_of_dropping_gradients():
    """The reference registers a backward for deform_conv2d (and nn.Conv2d has one); these kernels are forward-only, so a call
    recorded by autograd returns a tensor whose backward raises -- parameters never silently stay without gradients."""
    from cpu_vision_amd import functional as F
    from cpu_vision_amd.nn import Conv3x3ReLU
    layer = ops.DeformConv2d(4, 4, 3, padding=1).cuda()
    x = torch.rand(1, 4, 8, 8, device="cuda")
    off = torch.zeros(1, 18, 8, 8, device="cuda")
    out = layer(x, off)
    assert out.requires_grad
    with pytest.raises(RuntimeError, match="forward-only"):
        out.sum().backward()
    with torch.no_grad():
        assert not layer(x, off).requires_grad
    ops.register_torchvision_op()
    out = torch.ops.torchvision.deform_conv2d(x, layer.weight, off, torch.zeros(1, 1, device="cuda"), layer.bias, 1, 1, 1, 1, 1, 1, 1, 1, False)
    assert out.requires_grad
    with pytest.raises(RuntimeError, match="forward-only"):
        out.sum().backward()
    conv = Conv3x3ReLU(3, 8).cuda()
    y = conv(torch.rand(1, 3, 8, 8, device="cuda"))
    with pytest.raises(RuntimeError, match="forward-only"):
        y.mean().backward()
    xg = torch.rand(1, 3, 8, 8, device="cuda", requires_grad=True)
    with pytest.raises(RuntimeError, match="forward-only"):
        F.conv2d_bias_relu(xg, conv.weight.detach(), None).sum().backward()
    assert not F.conv2d_bias_relu(xg.detach(), conv.weight.detach(), None).requires_grad


def test_seeded_sweep_of_geometries_through_the_fused_kernel():
    """40 random geometries (kernel 1..5 x 1..5, stride 1..3, dilation 1..2, padding 0..3, weight / offset groups, channel counts
    that give 1-, 2-, 4- and 8-channel chunks, odd and even K per chunk, all four tile widths) -- every one the fused kernel
    accepts must equal the oracle bit for bit; the rest runs the two-kernel form against the same oracle."""
    from cpu_vision_amd import _lib
    rng = np.random.Generator(np.random.Philox(20261004))
    fused = 0
    for case in range(40):
        kh, kw = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        st = (int(rng.integers(1, 4)), int(rng.integers(1, 4)))
        dl = (int(rng.integers(1, 3)), int(rng.integers(1, 3)))
        pd = (int(rng.integers(0, 4)), int(rng.integers(0, 4)))
        groups = int(rng.choice([1, 1, 2, 3]))
        og = int(rng.choice([1, 1, 2, 4]))
        cin = groups * og * int(rng.choice([1, 2, 4, 8])) * int(rng.choice([1, 1, 3]))
        cout = groups * int(rng.choice([1, 5, 24, 40, 70, 150]))
        h = int(rng.integers(dl[0] * (kh - 1) + 1, 30)) if dl[0] * (kh - 1) + 1 < 30 else dl[0] * (kh - 1) + 1
        w = int(rng.integers(dl[1] * (kw - 1) + 1, 40)) if dl[1] * (kw - 1) + 1 < 40 else dl[1] * (kw - 1) + 1
        n = int(rng.integers(1, 4))
        use_mask, use_bias = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        x, off, wt, b, mask = _case(12000 + case, n, cin, cout, h, w, kh, kw, st, pd, dl, groups, og, use_mask, use_bias, scale=float(rng.choice([0.5, 2.0, 5.0])))
        with _fused():
            got = host(ops.deform_conv2d(dev(x), dev(off), dev(wt), dev(b), stride=st, padding=pd, dilation=dl, mask=dev(mask)))
        fused += _lib.last_kernel().startswith("k_deform_fused")
        np.testing.assert_array_equal(got, ref.deform_conv2d(x, off, wt, b, st, pd, dl, mask),
                                      err_msg=f"case {case}: n={n} {cin}->{cout} {h}x{w} k={kh}x{kw} s={st} p={pd} d={dl} g={groups} og={og} ({_lib.last_kernel()})")
    assert fused >= 30
