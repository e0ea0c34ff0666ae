#!/usr/bin/env python3
"""Builds cpu-vision_amd/lib/libmi355vision_torch.so: the C++ dispatcher shim of boundary B2 (deform_conv2d_shim.cpp) --
TORCH_LIBRARY_IMPL(torchvision, CUDA | Meta | Autocast) over the C ABI of libmi355vision.so.

    python cpu-vision_amd/torch_shim/build.py

torch.utils.cpp_extension drives g++ (one .cpp, no device code: the kernels are in libmi355vision.so); the result links
against libmi355vision.so through an $ORIGIN rpath, so both travel together.  __graft_entry__.build() calls this.
"""
from __future__ import annotations

import os
import shutil
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
PKG = HERE.parent
LIB_DIR = PKG / "lib"
OUT = LIB_DIR / "libmi355vision_torch.so"
SRC = HERE / "deform_conv2d_shim.cpp"
HEADER = PKG.parent / "include" / "mi355vision.h"


def build(force: bool = False, verbose: bool = False) -> Path:
    core = LIB_DIR / "libmi355vision.so"
    if not core.exists():
        raise RuntimeError(f"{core} not found: build the kernels first (python cpu-vision_amd/_build.py)")
    newest = max(SRC.stat().st_mtime, HEADER.stat().st_mtime, Path(__file__).stat().st_mtime)
    if OUT.exists() and OUT.stat().st_mtime >= newest and not force:
        return OUT
    import torch
    from torch.utils import cpp_extension

    build_dir = PKG / "build_torch_shim"
    build_dir.mkdir(exist_ok=True)
    torch_lib = Path(torch.__file__).resolve().parent / "lib"
    os.environ.setdefault("MAX_JOBS", "4")
    cpp_extension.load(
        name="mi355vision_torch", sources=[str(SRC)], build_directory=str(build_dir), is_python_module=False, verbose=verbose,
        extra_include_paths=[str(HEADER.parent), "/opt/rocm/include"],
        # the HIP runtime headers behind c10/hip/* need to know the platform; this is a build flag, not a code path
        extra_cflags=["-O2", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-Wno-deprecated-declarations"],
        # rpath: $$ is ninja's escape for $, the quotes keep the shell from expanding $ORIGIN
        extra_ldflags=[f"-L{LIB_DIR}", "-lmi355vision", "-Wl,-rpath,'$$ORIGIN'", "-Wl,-rpath,'$$ORIGIN/../lib'",
                       f"-L{torch_lib}", "-lc10_hip", "-ltorch_hip"])
    shutil.copy2(build_dir / "mi355vision_torch.so", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
