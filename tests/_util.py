"""Shared helpers for the test-suite: seeded inputs, golden loading, the parity tolerance."""
from __future__ import annotations

from functools import lru_cache
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def philox_f32(seed: int, shape) -> np.ndarray:
    """U[0,1) float32 from numpy's Philox generator (same generator as tests/golden/make_golden.py)."""
    return np.random.Generator(np.random.Philox(seed)).random(shape, dtype=np.float32)


def philox_u8(seed: int, shape) -> np.ndarray:
    return np.random.Generator(np.random.Philox(seed)).integers(0, 256, shape, dtype=np.uint8)


@lru_cache(maxsize=None)
def golden(name: str):
    return np.load(GOLDEN / f"{name}.npz", allow_pickle=False)


def conv_tol(ref: np.ndarray, w_abs_sum: float, x_abs_max: float, rel: float = 1e-5, floor: float = 1e-6):
    """SURVEY.md 8(d) / BASELINE.md: |a-b| <= 1e-5*|b| + 1e-6 * sum|w| * max|x|.

    A pure 1e-5 relative bound is ill-posed at ReLU zeros and Sobel zero crossings, hence the
    absolute floor scaled by the filter's gain."""
    return rel * np.abs(ref.astype(np.float64)) + floor * w_abs_sum * x_abs_max


def assert_conv_close(actual, ref, w_abs_sum=1.0, x_abs_max=1.0, rel=1e-5, floor=1e-6, what=""):
    actual = np.asarray(actual)
    ref = np.asarray(ref)
    assert actual.shape == ref.shape, f"{what}: shape {actual.shape} vs {ref.shape}"
    err = np.abs(actual.astype(np.float64) - ref.astype(np.float64))
    tol = conv_tol(ref, w_abs_sum, x_abs_max, rel, floor)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {int(bad.sum())}/{err.size} elements out of tolerance; worst at {i}: "
                             f"got {actual[i]!r} want {ref[i]!r} (err {err[i]:.3e} > tol {tol[i]:.3e})")
