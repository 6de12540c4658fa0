"""Shared helpers for the parity tests."""
from __future__ import annotations

import numpy as np


def half_to_f32(bits: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(bits, np.uint16).view(np.float16).astype(np.float32)


def fp16_ulp(ref: np.ndarray) -> np.ndarray:
    """Size of one fp16 ulp at |ref| (float32 array); subnormal range -> 2^-24."""
    a = np.abs(ref.astype(np.float32))
    e = np.floor(np.log2(np.maximum(a, np.float32(2.0 ** -14))))
    return np.exp2(e - 10).astype(np.float32)


def hdr_mismatch(gpu_bits: np.ndarray, ref_bits: np.ndarray, exclude: np.ndarray | None = None):
    """Compare two RGBA16F images under the north-star tolerance max(1e-3, 1 fp16 ulp of |ref|), NaN == NaN.
    Returns (number of failing channel values, worst excess, mask of failing pixels)."""
    g, r = half_to_f32(gpu_bits), half_to_f32(ref_bits)
    both_nan = np.isnan(g) & np.isnan(r)
    tol = np.maximum(np.float32(1e-3), fp16_ulp(np.where(np.isfinite(r), r, 0)))
    diff = np.abs(g - r)
    same_inf = np.isinf(g) & np.isinf(r) & (np.sign(g) == np.sign(r))
    bad = ~(both_nan | same_inf) & ~(diff <= tol)
    if exclude is not None:
        bad &= ~exclude[..., None].astype(bool)
    excess = np.where(bad & np.isfinite(diff), diff - tol, 0)
    return int(bad.sum()), float(excess.max()) if bad.any() else 0.0, bad.any(axis=-1)
