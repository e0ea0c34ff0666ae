"""Builds libmi355vision.so (hand-written gfx950 HIP kernels + the C ABI) in-tree with hipcc.

    python cpu-vision_amd/_build.py [--force] [--tuning]

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the resulting
.so is git-ignored but travels to the GPU box with the repo snapshot.

Two kinds of library come out of the same sources:
  lib/libmi355vision.so          the product: no environment lookups, kernel selection from the arguments alone;
  lib/libmi355vision_tuning.so   -DMV_TUNING: the MV_* knobs of tools/ (forced kernels, strip heights, ...) are read
                                 through tools/tuning/mv_tuning.h.  Only tools/ and the tests that force an
                                 alternative kernel load it (cpu_vision_amd._lib.load_tuning()).
Named experiment variants (MV_BUILD_VARIANT=name + MV_HIPCC_EXTRA="flags") are tuning builds with extra flags.
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
TUNING_INC = HERE.parent / "tools" / "tuning"
SOURCES = ["abi.hip", "dw3x3.hip", "dw3x3_u8.hip", "dwk_u8.hip", "tiefix_u8.hip", "dwtile.hip", "dwf64.hip", "separable.hip", "sepfast.hip", "sepstream.hip", "conv3x3_mfma.hip", "conv3x3_c3.hip", "conv3x3_gen.hip", "cnn_ops.hip", "linear_mfma.hip", "resize.hip", "convnorm.hip", "conv_igemm.hip", "invres.hip", "deform.hip", "deform_fused.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # no implicit contraction: every fused multiply-add in the kernels is an explicit fmaf(), so the
    # rounding sequence is exactly the oracle's
    "-ffp-contract=off", "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
]


def _paths(variant: str):
    obj = HERE / ("build_" + variant if variant else "build")
    lib = HERE / "lib" / ("libmi355vision_" + variant + ".so" if variant else "libmi355vision.so")
    return obj, lib


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_id(flags) -> str:
    """First 16 hex digits of the SHA-256 over every source, header and compile flag of the library."""
    h = hashlib.sha256()
    for f in sorted(CSRC.iterdir()):
        if f.suffix in (".hip", ".h"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    h.update((HERE.parent / "include" / "mi355vision.h").read_bytes())
    h.update(" ".join(flags).encode())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False, variant: str = "", extra_flags=()) -> Path:
    """variant "" = the product library; any other name = a -DMV_TUNING build (plus `extra_flags`)."""
    obj_dir, lib = _paths(variant)
    obj_dir.mkdir(exist_ok=True)
    lib.parent.mkdir(exist_ok=True)
    flags = list(FLAGS)
    if variant:
        flags += ["-DMV_TUNING", f"-I{TUNING_INC}", *extra_flags]
    headers = [CSRC / "mv_common.h", HERE.parent / "include" / "mi355vision.h", CSRC / "mv_epilogue.h", CSRC / "mv_deform.h", CSRC / "mv_conv.h", CSRC / "mv_invres.h", CSRC / "mv_act.h"]
    if variant:
        headers.append(TUNING_INC / "mv_tuning.h")
    bid = build_id(flags)
    id_file = obj_dir / "build_id.txt"  # abi.hip carries the id: rebuild it whenever any source changed
    id_changed = not id_file.exists() or id_file.read_text() != bid
    jobs = []
    # experiment variants that touch one kernel: MV_VARIANT_SOURCES="dwk_u8.hip" compiles only those files with the extra
    # flags; every other object comes from the plain tuning build
    only = set(os.environ.get("MV_VARIANT_SOURCES", "").split()) if variant and variant != "tuning" else set()
    shared_dir = _paths("tuning")[0]
    # sources that never call tune_env() compile to the same object with or without -DMV_TUNING: the tuning build links the
    # product build's objects for them (built first by build_all)
    product_dir = _paths("")[0]
    same_as_product = set()
    if variant and not extra_flags:
        same_as_product = {src for src in SOURCES if not _reaches_knobs(CSRC / src)
                           and (product_dir / (src + ".o")).exists()
                           and not _stale(product_dir / (src + ".o"), [CSRC / src, *headers[:6]])}
    for src in SOURCES:
        if (only and src not in only) or src in same_as_product:
            continue
        obj = obj_dir / (src + ".o")
        if force or _stale(obj, [CSRC / src, *headers]) or (src == "abi.hip" and id_changed):
            extra = ["-Rpass-analysis=kernel-resource-usage"] if verbose else []
            if src == "abi.hip":
                extra.append(f'-DMV_BUILD_ID="{bid}"')
            jobs.append([HIPCC, *flags, *extra, "-c", str(CSRC / src), "-o", str(obj)])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, max(1, len(jobs)))) as ex:
        logs = list(ex.map(run, jobs))
    if verbose:
        for log in logs:
            sys.stderr.write(log)
    def obj_of(src):
        if src in same_as_product:
            return product_dir / (src + ".o")
        if not only or src in only:
            return obj_dir / (src + ".o")
        shared = shared_dir / (src + ".o")  # single-file variants: everything else from the tuning build (or, for sources
        return shared if shared.exists() else product_dir / (src + ".o")  # without knobs, from the product build)
    objs = [str(obj_of(s)) for s in SOURCES]
    if force or jobs or _stale(lib, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *objs])
    id_file.write_text(bid)
    return lib


_KNOB_TOKENS = re.compile(r"\btune_env\b|\bMV_TUNING\b")
_LOCAL_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _reaches_knobs(path: Path, _seen=None) -> bool:
    """True if this translation unit can compile differently under -DMV_TUNING: the source, or any project header it includes
    (transitively), mentions tune_env / MV_TUNING.  mv_common.h is exempt -- it only DEFINES tune_env (a constant nullptr in
    the product build), which changes nothing for a source that never calls it; any other header that wraps a knob makes
    every source including it a tuning source."""
    _seen = set() if _seen is None else _seen
    if path in _seen or not path.exists():
        return False
    _seen.add(path)
    text = path.read_text()
    if path.name != "mv_common.h" and _KNOB_TOKENS.search(text):
        return True
    return any(_reaches_knobs(path.parent / inc, _seen) for inc in _LOCAL_INCLUDE.findall(text))


def build_all(force: bool = False, verbose: bool = False):
    """The product library and the tuning library (what __graft_entry__.build() calls)."""
    return build(force, verbose), build(force, verbose, variant="tuning")


# experiment hook of tools/: MV_BUILD_VARIANT=name MV_HIPCC_EXTRA="-DFOO=1" python cpu-vision_amd/_build.py
_VARIANT = os.environ.get("MV_BUILD_VARIANT", "")

if __name__ == "__main__":
    force, verbose = "--force" in sys.argv, "--verbose" in sys.argv
    if _VARIANT:
        print(build(force, verbose, variant=_VARIANT, extra_flags=os.environ.get("MV_HIPCC_EXTRA", "").split()))
    elif "--tuning" in sys.argv:
        print(build(force, verbose, variant="tuning"))
    else:
        for p in build_all(force, verbose):
            print(p)
