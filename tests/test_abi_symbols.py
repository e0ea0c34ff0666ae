"""The C-ABI library builds, loads and exports every symbol include/mi355vision.h declares (no compute)."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "mi355vision.h"


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(mv_[a-z0-9_]+)\s*\(", text)))


def declared_prototypes():
    """{name: (return class, [argument classes])} parsed from the header: 'ptr', 'i32', 'i64', 'f32', 'f64', 'str', None."""
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#[^\n]*", "", text, flags=re.M)  # preprocessor lines
    text = text.replace('extern "C"', "").replace("{", ";").replace("}", ";")

    def cls(decl: str):
        decl = decl.strip()
        if "*" in decl:
            return "str" if re.match(r"const\s+char\s*\*\s*$", decl) else "ptr"
        base = re.sub(r"\b(const|unsigned|signed)\b", "", decl).split()
        kind = base[0] if base else "void"
        return {"int": "i32", "int64_t": "i64", "double": "f64", "float": "f32", "void": None}[kind]

    protos = {}
    for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_\s\*]*?)\b(mv_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", text):
        arglist = [a for a in (x.strip() for x in args.split(",")) if a and a != "void"]
        protos[name] = (cls(ret), [cls(a) for a in arglist])
    return protos


def ctypes_class(t):
    if t is None:
        return None
    if t is ctypes.c_char_p:
        return "str"
    if t is ctypes.c_void_p or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
        return "ptr"
    return {ctypes.c_int: "i32", ctypes.c_int64: "i64", ctypes.c_double: "f64", ctypes.c_float: "f32"}[t]


def test_binding_signatures_match_the_header_prototypes():
    """Arity and the class of every argument / return value of _lib.SYMBOLS against include/mi355vision.h (which the
    library's own sources include, so the compiler checks the definitions against the same text)."""
    from cpu_vision_amd import _lib

    protos = declared_prototypes()
    assert sorted(protos) == declared_symbols()
    for name, (res, args) in _lib.SYMBOLS.items():
        want_res, want_args = protos[name]
        got_args = [ctypes_class(a) for a in args]
        assert len(got_args) == len(want_args), f"{name}: binding passes {len(got_args)} arguments, header declares {len(want_args)}"
        assert got_args == want_args, f"{name}: binding {got_args} != header {want_args}"
        assert ctypes_class(res) == want_res, f"{name}: return {ctypes_class(res)} != header {want_res}"


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for s in ["mv_depthwise_conv2d_f32", "mv_depthwise_conv2d_u8", "mv_gaussian_blur_f32", "mv_gaussian_blur_u8",
              "mv_separable_blur_f32", "mv_sobel_f32", "mv_gaussian_sobel_f32", "mv_sharpness_f32", "mv_sharpness_u8",
              "mv_conv3x3_bias_relu_f32", "mv_last_error", "mv_abi_version", "mv_device_count"]:
        assert s in syms


def test_library_exports_every_declared_symbol():
    import cpu_vision_amd as mv
    from cpu_vision_amd import _lib

    assert mv.LIB_PATH.exists(), "run python cpu-vision_amd/_build.py (or __graft_entry__.build())"
    lib = ctypes.CDLL(str(mv.LIB_PATH))
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in mi355vision.h but not exported"
    # and the ctypes binding covers exactly the declared surface
    assert sorted(_lib.SYMBOLS) == declared_symbols()
    assert mv.load_library().mv_abi_version() == 1


def test_abi_rejects_bad_arguments_without_a_device():
    """Validation happens before any launch, so these calls are safe on a box with no GPU."""
    from cpu_vision_amd import _lib

    lib = _lib.load()
    t = _lib.taps([0.25, 0.5, 0.25])
    # even kernel size
    rc = lib.mv_gaussian_blur_f32(1, 2, 1, 8, 8, t, 2, t, 3, None)
    assert rc == -1 and b"odd" in lib.mv_last_error()
    # reflect padding must be smaller than the image
    rc = lib.mv_gaussian_blur_f32(1, 2, 1, 1, 8, t, 3, t, 3, None)
    assert rc == -1 and b"reflect" in lib.mv_last_error()
    # aliasing
    rc = lib.mv_gaussian_blur_f32(16, 16, 1, 8, 8, t, 3, t, 3, None)
    assert rc == -1 and b"alias" in lib.mv_last_error()
    # negative sharpness
    rc = lib.mv_sharpness_u8(1, 2, 1, 8, 8, -0.5, 0, None)
    assert rc == -1 and b"non-negative" in lib.mv_last_error()
    # empty work is a no-op success
    assert lib.mv_gaussian_blur_f32(None, None, 0, 8, 8, t, 3, t, 3, None) == 0
    with pytest.raises(ValueError):
        _lib.check(lib.mv_sobel_f32(1, 2, 2, 1, 8, 8, 1, None))  # gx == gy
