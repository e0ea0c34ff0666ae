#!/usr/bin/env python3
"""Builds the one native kernel the reference owns on this path, its CPU `torchvision::deform_conv2d`, from the sources WHERE
THEY LIE under /root/reference into oracle/_ref/ (git-ignored; travels to the GPU box with the snapshot like every built
.so).  Test infrastructure: a second, reference-owned oracle for SURVEY.md 8f.4 (deformable im2col + GEMM) and, with zero
offsets, for the dense conv of cfg4.  Only tests/ load it (in a subprocess, tests/test_oracle_golden.py).

    python oracle/build_ref.py        # no-op when /root/reference is absent (the GPU box uses the prebuilt file)

Recipe: torch.utils.cpp_extension.load over three of the reference's own files (no reference build system, no generated
code, no stand-in headers): csrc/vision.cpp (library fragment), csrc/ops/deform_conv2d.cpp (operator schema + dispatcher
entry), csrc/ops/cpu/deform_conv2d_kernel.cpp (the CPU kernel); it links against the torch that is installed in the image.
No source is copied into the repository.
"""
from __future__ import annotations

import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/torchvision/csrc")
OUT = HERE / "_ref"
NAME = "ref_deform_conv2d_cpu"


def built_library() -> Path | None:
    p = OUT / f"{NAME}.so"
    return p if p.exists() else None


def build(verbose: bool = False) -> Path | None:
    if not REF.exists():
        return built_library()
    srcs = [REF / "vision.cpp", REF / "ops" / "deform_conv2d.cpp", REF / "ops" / "cpu" / "deform_conv2d_kernel.cpp"]
    so = OUT / f"{NAME}.so"
    if so.exists() and all(so.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return so
    OUT.mkdir(exist_ok=True)
    from torch.utils.cpp_extension import load
    load(name=NAME, sources=[str(s) for s in srcs], extra_include_paths=[str(REF)], build_directory=str(OUT),
         is_python_module=False, verbose=verbose, extra_cflags=["-O2"])
    return built_library()


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
