"""ctypes wrapper of the CPU oracle (oracle/ur_oracle.cpp). TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module. PARITY UNPINNED by the
reference (it has no tests or golden vectors for this path; see the header of ur_oracle.cpp).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "_build" / "liburoracle.so"
_lib = None


class MipDesc(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


def build(force: bool = False) -> Path:
    src = _HERE / "ur_oracle.cpp"
    hdr = _HERE.parent / "include" / "ur_hotpath.h"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        r = subprocess.run(["make", "-C", str(_HERE), "-B", "all"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"oracle build failed:\n{r.stdout}\n{r.stderr}")
    return _LIB_PATH


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not _LIB_PATH.exists():
            build()
        _lib = C.CDLL(str(_LIB_PATH))
        _lib.uro_h2f.restype = C.c_float
        _lib.uro_h2f.argtypes = [C.c_uint16]
        _lib.uro_f2h.restype = C.c_uint16
        _lib.uro_f2h.argtypes = [C.c_float]
        _lib.uro_hzb_layout.restype = C.c_uint32
        _lib.uro_env_cube_texels.restype = C.c_size_t
        _lib.uro_env_cube_texels.argtypes = [C.c_uint32, C.c_uint32]
        _lib.uro_sample_cmp.restype = C.c_float
        _lib.uro_sample_cmp.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float]
        _lib.uro_srgb_to_linear.restype = C.c_float
        _lib.uro_srgb_to_linear.argtypes = [C.c_uint32]
        _lib.uro_hardware_threads.restype = C.c_int
    return _lib


def _p(a: np.ndarray):
    assert a.flags.c_contiguous, "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def set_threads(n: int) -> None:
    load().uro_set_threads(int(n))


def hardware_threads() -> int:
    return int(load().uro_hardware_threads())


def h2f(h: int) -> float:
    return float(load().uro_h2f(int(h)))


def f2h(f: float) -> int:
    return int(load().uro_f2h(float(f)))


def hzb_layout(w: int, h: int):
    """Packed layout (CreateHZBResources sizing). Returns (list of (offset,w,h), total floats)."""
    mips = (MipDesc * 16)()
    n = C.c_uint32(0)
    total = load().uro_hzb_layout(C.c_uint32(w), C.c_uint32(h), mips, C.byref(n))
    return [(mips[i].offset, mips[i].width, mips[i].height) for i in range(n.value)], int(total)


def _mips_array(mips):
    arr = (MipDesc * 16)()
    for i, (o, w, h) in enumerate(mips):
        arr[i].offset, arr[i].width, arr[i].height = int(o), int(w), int(h)
    return arr


def build_hzb(depth: np.ndarray, mips, total_floats: int) -> np.ndarray:
    """depth: (h, w) float32. mips: [(offset,w,h)] (any layout). Returns the flat HZB buffer (untouched gaps are NaN)."""
    depth = np.ascontiguousarray(depth, np.float32)
    h, w = depth.shape
    out = np.full(total_floats, np.nan, np.float32)
    load().uro_build_hzb(_p(depth), C.c_uint32(w), C.c_uint32(h), _p(out), _mips_array(mips), C.c_uint32(len(mips)))
    return out


def cull_indirect_args(constants: np.ndarray, bounds: np.ndarray, hzb: np.ndarray | None, mips, indirect_args: np.ndarray,
                       index_base: int = 0):
    """Returns (indirect_args_out uint32[n,16], stats[2], visible_idx, visible_count)."""
    constants = np.ascontiguousarray(constants, np.uint32)
    n = int(constants[40])
    bounds = np.ascontiguousarray(bounds, np.float32)
    args = np.ascontiguousarray(indirect_args, np.uint32).copy()
    stats = np.zeros(2, np.uint32)
    vis = np.zeros(max(n, 1), np.uint32)
    cnt = np.zeros(1, np.uint32)
    hz = np.ascontiguousarray(hzb, np.float32) if hzb is not None else np.zeros(1, np.float32)
    load().uro_cull_indirect_args(_p(constants), _p(bounds), _p(hz), _mips_array(mips or []), _p(args), _p(stats), _p(vis), _p(cnt),
                                  C.c_uint32(index_base))
    return args, stats, vis[:int(cnt[0])].copy(), int(cnt[0])


def cpu_frustum(planes24: np.ndarray, bounds: np.ndarray) -> np.ndarray:
    planes24 = np.ascontiguousarray(planes24, np.float32)
    bounds = np.ascontiguousarray(bounds, np.float32)
    n = bounds.shape[0]
    out = np.zeros(n, np.uint8)
    load().uro_cpu_frustum(_p(planes24), _p(bounds), C.c_uint32(n), _p(out))
    return out


class FragileMask(np.ndarray):
    """The fragile-pixel mask of deferred_lighting (1 where a shadow compare is within 1e-5 of flipping) carrying, when any
    pixel is fragile, the two images that bracket every admissible outcome of those pixels: `lo` (every tie fails) and
    `hi` (every tie passes). tests/util.py:hdr_mismatch holds fragile pixels to that interval."""
    lo = None
    hi = None

    def __array_finalize__(self, obj):
        if obj is not None and getattr(obj, "shape", None) == self.shape:
            self.lo, self.hi = getattr(obj, "lo", None), getattr(obj, "hi", None)


def deferred_lighting(scene, A, B, Cc, shadow, env_cube, env_base, env_mips, lut, hdr, w, h, row0=0, rows=None,
                      want_fragile=False, tie_mode=0):
    """scene: a ctypes struct laid out as ur_scene_constants. Arrays are band-local. Returns hdr_out[, fragile].
    tie_mode: 0 the reference's compare; +1 / -1 every shadow compare within 1e-5 of flipping passes / fails."""
    rows = A.shape[0] if rows is None else rows
    if want_fragile and tie_mode == 0:
        out, frag = deferred_lighting(scene, A, B, Cc, shadow, env_cube, env_base, env_mips, lut, hdr, w, h, row0, rows, True, tie_mode=None)
        frag = frag.view(FragileMask)
        if frag.any():
            frag.lo = deferred_lighting(scene, A, B, Cc, shadow, env_cube, env_base, env_mips, lut, hdr, w, h, row0, rows, False, tie_mode=-1)
            frag.hi = deferred_lighting(scene, A, B, Cc, shadow, env_cube, env_base, env_mips, lut, hdr, w, h, row0, rows, False, tie_mode=+1)
        return out, frag
    tie_mode = tie_mode or 0
    A = np.ascontiguousarray(A, np.uint16); B = np.ascontiguousarray(B, np.uint16); Cc = np.ascontiguousarray(Cc, np.uint32)
    out = np.ascontiguousarray(hdr, np.uint16).copy()
    lut = np.ascontiguousarray(lut, np.uint16)
    env_cube = np.ascontiguousarray(env_cube, np.uint16)
    sh = np.ascontiguousarray(shadow, np.float32) if shadow is not None else None
    frag = np.zeros((rows, w), np.uint8) if want_fragile else None
    load().uro_set_shadow_tie_mode(C.c_int(tie_mode))
    try:
        load().uro_deferred_lighting(C.byref(scene), _p(A), _p(B), _p(Cc), _p(sh) if sh is not None else None, _p(env_cube),
                                     C.c_uint32(env_base), C.c_uint32(env_mips), _p(lut), C.c_uint32(lut.shape[1]), C.c_uint32(lut.shape[0]),
                                     _p(out), C.c_uint32(w), C.c_uint32(h), C.c_uint32(row0), C.c_uint32(rows),
                                     _p(frag) if frag is not None else None)
    finally:
        load().uro_set_shadow_tie_mode(C.c_int(0))
    return (out, frag) if want_fragile else out


def sky_atmosphere(sky, depth, hdr, w, h, row0=0, rows=None):
    rows = depth.shape[0] if rows is None else rows
    depth = np.ascontiguousarray(depth, np.float32)
    out = np.ascontiguousarray(hdr, np.uint16).copy()
    load().uro_sky_atmosphere(C.byref(sky), _p(depth), _p(out), C.c_uint32(w), C.c_uint32(h), C.c_uint32(row0), C.c_uint32(rows))
    return out


class TonemapConstants(C.Structure):
    _fields_ = [("EnableTonemap", C.c_uint32), ("EnableAutoExposure", C.c_uint32), ("Exposure", C.c_float), ("Gamma", C.c_float)]


def tonemap(hdr: np.ndarray, exposure=1.0, gamma=2.2, enable_tonemap=True, exposure_ev=None) -> np.ndarray:
    """hdr: (..., 4) uint16 RGBA16F bit patterns -> (...) uint32 R8G8B8A8_UNORM."""
    hdr = np.ascontiguousarray(hdr, np.uint16)
    out = np.zeros(hdr.shape[:-1], np.uint32)
    k = TonemapConstants(int(enable_tonemap), int(exposure_ev is not None), exposure, gamma)
    ev = np.array([exposure_ev], np.float32) if exposure_ev is not None else None
    load().uro_tonemap(C.byref(k), _p(hdr), _p(ev) if ev is not None else None, _p(out), C.c_uint32(out.size))
    return out


def temporal_aa(current: np.ndarray, history: np.ndarray, history_weight: float, use_history: bool, row0: int = 0, rows=None) -> np.ndarray:
    """current: (H, W, 4) uint16 full frame; history: (rows, W, 4) band. Returns the resolved band."""
    current = np.ascontiguousarray(current, np.uint16)
    H, W = current.shape[:2]
    rows = H - row0 if rows is None else rows
    history = np.ascontiguousarray(history, np.uint16)
    out = np.zeros((rows, W, 4), np.uint16)
    load().uro_temporal_aa(_p(current), _p(history), _p(out), C.c_float(history_weight), C.c_uint32(int(use_history)), C.c_uint32(W), C.c_uint32(H),
                           C.c_uint32(row0), C.c_uint32(rows))
    return out


def env_cube_texels(base: int, mips: int) -> int:
    return int(load().uro_env_cube_texels(base, mips))


def stage_env_cube(src: np.ndarray, base: int, mips: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint16)
    out = np.zeros((env_cube_texels(base, mips), 4), np.uint16)
    load().uro_stage_env_cube(_p(src), C.c_uint32(base), C.c_uint32(mips), _p(out))
    return out


def evaluate_pbr(albedo, metallic, roughness, F0, N, V, L) -> np.ndarray:
    f = lambda v: np.ascontiguousarray(v, np.float32)
    out = np.zeros(3, np.float32)
    a, f0, n, v, l = f(albedo), f(F0), f(N), f(V), f(L)
    load().uro_evaluate_pbr(_p(a), C.c_float(metallic), C.c_float(roughness), _p(f0), _p(n), _p(v), _p(l), _p(out))
    return out


def apply_atmosphere(sky, view_dir) -> np.ndarray:
    out = np.zeros(3, np.float32)
    d = np.ascontiguousarray(view_dir, np.float32)
    load().uro_apply_atmosphere(C.byref(sky), _p(d), _p(out))
    return out


def sample_cube_level(env_cube, base, mips, direction, level) -> np.ndarray:
    out = np.zeros(3, np.float32)
    e = np.ascontiguousarray(env_cube, np.uint16)
    d = np.ascontiguousarray(direction, np.float32)
    load().uro_sample_cube_level(_p(e), C.c_uint32(base), C.c_uint32(mips), _p(d), C.c_float(level), _p(out))
    return out


def select_cube_face(direction):
    d = np.ascontiguousarray(direction, np.float32)
    face = C.c_int(0)
    uv = np.zeros(2, np.float32)
    load().uro_select_cube_face(_p(d), C.byref(face), _p(uv))
    return face.value, float(uv[0]), float(uv[1])


def sample_cmp(shadow: np.ndarray, u: float, v: float, cmp: float) -> float:
    s = np.ascontiguousarray(shadow, np.float32)
    return float(load().uro_sample_cmp(_p(s), s.shape[1], s.shape[0], u, v, cmp))


def sample_lut(lut: np.ndarray, u: float, v: float) -> np.ndarray:
    l = np.ascontiguousarray(lut, np.uint16)
    out = np.zeros(2, np.float32)
    load().uro_sample_lut(_p(l), C.c_uint32(l.shape[1]), C.c_uint32(l.shape[0]), C.c_float(u), C.c_float(v), _p(out))
    return out


def srgb_to_linear(byte: int) -> float:
    return float(load().uro_srgb_to_linear(int(byte)))
