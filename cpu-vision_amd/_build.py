"""Builds libmi355vision.so (hand-written gfx950 HIP kernels + the C ABI) in-tree with hipcc.

    python cpu-vision_amd/_build.py [--force]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
.so is git-ignored but travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
# experiment hook (tools/, never the product build): MV_BUILD_VARIANT=name + MV_HIPCC_EXTRA="flags" builds
# lib/libmi355vision_<name>.so next to the product library; load it with MI355VISION_LIB=<path>
_VARIANT = os.environ.get("MV_BUILD_VARIANT", "")
OBJ = HERE / ("build_" + _VARIANT if _VARIANT else "build")
LIB = HERE / "lib" / ("libmi355vision_" + _VARIANT + ".so" if _VARIANT else "libmi355vision.so")
SOURCES = ["abi.hip", "dw3x3.hip", "dw3x3_u8.hip", "dwk_u8.hip", "dwtile.hip", "separable.hip", "sepfast.hip", "sepstream.hip", "conv3x3_mfma.hip", "conv3x3_c3.hip", "conv3x3_gen.hip", "cnn_ops.hip", "linear_mfma.hip", "resize.hip", "convnorm.hip", "deform.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # no implicit contraction: every fused multiply-add in the kernels is an explicit fmaf(), so the
    # rounding sequence is exactly the oracle's
    "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
] + (os.environ.get("MV_HIPCC_EXTRA", "").split() if _VARIANT else [])


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    LIB.parent.mkdir(exist_ok=True)
    headers = [CSRC / "mv_common.h", HERE.parent / "include" / "mi355vision.h"]
    jobs = []
    for src in SOURCES:
        obj = OBJ / (src + ".o")
        if force or _stale(obj, [CSRC / src, *headers]):
            extra = ["-Rpass-analysis=kernel-resource-usage"] if verbose else []
            jobs.append([HIPCC, *FLAGS, *extra, "-c", str(CSRC / src), "-o", str(obj)])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        logs = list(ex.map(run, jobs))
    if verbose:
        for log in logs:
            sys.stderr.write(log)
    objs = [str(OBJ / (s + ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
