"""Boundary B2 built for real: cpu-vision_amd/torch_shim/deform_conv2d_shim.cpp -- a C++ torch
extension that registers torchvision::deform_conv2d for the CUDA (= HIP), Meta and Autocast dispatch keys with
TORCH_LIBRARY_IMPL over the C ABI of libmi355vision.so, as the reference does in csrc/ops/cuda/deform_conv2d_kernel.cu:1323,
torchvision/_meta_registrations.py:177-198 and csrc/ops/autocast/deform_conv2d_kernel.cpp:12-52.  The first non-Python
consumer of include/mi355vision.h.  Each test drives tests/_shim_worker.py in a fresh process (the shim and the Python
registration of cpu_vision_amd.ops both claim the operator's CUDA key)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
WORKER = ROOT / "tests" / "_shim_worker.py"


def _run(mode):
    r = subprocess.run([sys.executable, str(WORKER), mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_shim_links_the_c_abi_and_registers_native_kernels():
    """No GPU needed: the library loads (it links libmi355vision.so through its $ORIGIN rpath), the operator has C++ kernels
    for the CUDA, Meta and AutocastCUDA keys, and the fake kernel gives the reference's output shape rule."""
    import shutil
    so = ROOT / "cpu-vision_amd" / "lib" / "libmi355vision_torch.so"
    assert so.exists(), "run __graft_entry__.build() (or python cpu-vision_amd/torch_shim/build.py)"
    if shutil.which("readelf"):
        dyn = subprocess.run(["readelf", "-d", str(so)], capture_output=True, text=True).stdout
        assert "libmi355vision.so" in dyn and "$ORIGIN" in dyn
    res = _run("cpu")
    assert res["has_cuda_kernel"] and res["has_meta_kernel"] and res["has_autocast_kernel"]
    assert res["meta_shape"] == [4, 2, 2, 3] and res["meta_device"] == "meta" and res["fake_shape"] == [4, 2, 2, 3]


@pytest.mark.gpu
def test_reference_test_configuration_through_the_cpp_dispatcher_kernel():
    """torch.ops.torchvision.deform_conv2d -> C++ shim -> mv_deform_conv2d_f32: TestDeformConv's own configuration is bit-exact
    against the oracle (DCNv2 and the v1 call with placeholder tensors), float64 tensors come back as float64 within the
    reference's 1e-5, autocast widens half inputs and returns half, torch.compile traces through the fake kernel, bad ranks
    raise RuntimeError, and both the fused kernel and the columns-workspace path are reached from C++."""
    res = _run("gpu")
    assert res["bit_exact_vs_oracle"] and res["v1_bit_exact_vs_oracle"]
    assert res["f64_dtype"] == "torch.float64" and res["f64_max_abs_err"] <= 1e-5
    assert res["autocast_dtype"] == "torch.float16" and res["autocast_max_abs_err_vs_oracle_on_rounded_inputs"] <= 2e-2
    assert res["compiled_equals_eager"] is True, res["compiled_equals_eager"]
    assert res["bad_rank_raises"] is True
    assert res["fused_bit_exact_vs_oracle"] and res["columns_bit_exact_vs_oracle"]
