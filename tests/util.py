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
    Returns (number of failing channel values, worst excess, mask of failing pixels).

    `exclude`: pixels left out of the strict comparison. When it is the oracle's FragileMask (a shadow compare within 1e-5
    of flipping: a correctly rounded kernel may land on either side) those pixels are NOT skipped: each channel must lie,
    within the same tolerance, either on the reference value or inside the interval spanned by the oracle's two forced
    evaluations (every tie fails / every tie passes), which brackets every admissible outcome (the colour is monotone in
    each tap's compare). A kernel cannot write anything else there."""
    g, r = half_to_f32(gpu_bits), half_to_f32(ref_bits)
    both_nan = np.isnan(g) & np.isnan(r)
    tol = np.maximum(np.float32(1e-3), fp16_ulp(np.where(np.isfinite(r), r, 0)))
    diff = np.abs(g - r)
    same_inf = np.isinf(g) & np.isinf(r) & (np.sign(g) == np.sign(r))
    bad = ~(both_nan | same_inf) & ~(diff <= tol)
    if exclude is not None:
        ex = np.asarray(exclude).astype(bool)
        lo_bits, hi_bits = getattr(exclude, "lo", None), getattr(exclude, "hi", None)
        if lo_bits is not None and hi_bits is not None and ex.any():
            lo, hi = half_to_f32(lo_bits), half_to_f32(hi_bits)
            a, b = np.minimum(lo, hi), np.maximum(lo, hi)
            tol_i = np.maximum(np.float32(1e-3), fp16_ulp(np.where(np.isfinite(b), b, 0)))
            inside = (g >= a - tol_i) & (g <= b + tol_i)
            bad &= ~(ex[..., None] & inside)
        else:
            bad &= ~ex[..., None]
    excess = np.where(bad & np.isfinite(diff), diff - tol, 0)
    return int(bad.sum()), float(excess.max()) if bad.any() else 0.0, bad.any(axis=-1)
