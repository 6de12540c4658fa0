"""Seeded synthetic inputs for the hot path (numpy; host side, setup time only).

Everything derives from a counter-based hash h = pcg_hash(seed ^ x*0x9E3779B1 ^ y*0x85EBCA77 ^ channel*0xC2B2AE3D), so any
rank can generate any row band without communication.

Two G-buffer generators:
  * gbuffer_scene — a ray-cast analytic atrium (floor, walls, covered side aisles, two colonnades, spheres, open roof)
    seen from the shipped Sponza camera. Depth, normals and materials are piecewise smooth like a rasterised scene; this
    is the bench workload ("Sponza ... G-buffer precomputed" — the 10.8 MB Sponza geometry blob is not in the reference
    checkout, .MISSING_LARGE_BLOBS:7, so the real G-buffer cannot be rasterised).
  * gbuffer_iid — SURVEY.md §8d's independent-per-pixel variant (random normal / depth / material per pixel, 64x64
    background blobs). It is the stress case: every lane gathers a different env-cube, LUT and shadow-map line.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

SEED_BASE = 0x5EED0000


# ---------------------------------------------------------------------------------------------------------------------
# hashing
# ---------------------------------------------------------------------------------------------------------------------
def pcg_hash(v: np.ndarray) -> np.ndarray:
    v = np.asarray(v, dtype=np.uint32)
    with np.errstate(over="ignore"):
        state = v * np.uint32(747796405) + np.uint32(2891336453)
        word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
        return (word >> np.uint32(22)) ^ word


def hash_u32(seed: int, x: np.ndarray, y: np.ndarray, channel: int) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint32)
    y = np.asarray(y, dtype=np.uint32)
    with np.errstate(over="ignore"):
        k = np.uint32(seed & 0xFFFFFFFF) ^ (x * np.uint32(0x9E3779B1)) ^ (y * np.uint32(0x85EBCA77)) ^ np.uint32((channel * 0xC2B2AE3D) & 0xFFFFFFFF)
    return pcg_hash(k)


def hash_unit(seed: int, x, y, channel: int) -> np.ndarray:
    """float32 in [0,1) with 24 random bits."""
    return (hash_u32(seed, x, y, channel) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def _grid(w: int, row0: int, rows: int):
    x = np.arange(w, dtype=np.uint32)[None, :].repeat(rows, 0)
    y = (np.arange(rows, dtype=np.uint32) + np.uint32(row0))[:, None].repeat(w, 1)
    return x, y


def _half4(x, y, z, w) -> np.ndarray:
    """(rows, w, 4) uint16 holding binary16 bit patterns (RTE from float32)."""
    return np.stack([np.asarray(c, np.float32).astype(np.float16) for c in (x, y, z, w)], axis=-1).view(np.uint16)


def _srgb_encode(lin: np.ndarray) -> np.ndarray:
    lin = np.clip(lin, 0.0, 1.0).astype(np.float64)
    s = np.where(lin <= 0.0031308, lin * 12.92, 1.055 * np.power(lin, 1.0 / 2.4) - 0.055)
    return np.clip(np.rint(s * 255.0), 0, 255).astype(np.uint32)


@dataclass
class GBuffer:
    """Band-local G-buffer: rows [row0,row0+rows) of a w x h frame."""
    w: int
    h: int
    row0: int
    rows: int
    A: np.ndarray      # (rows, w, 4) uint16: view normal.xyz, -viewZ   (R16G16B16A16_FLOAT)
    B: np.ndarray      # (rows, w, 4) uint16: specular, metallic, roughness, 1
    C: np.ndarray      # (rows, w) uint32: R8G8B8A8_UNORM_SRGB, R in the low byte
    hdr: np.ndarray    # (rows, w, 4) uint16: emissive, 1
    depth: np.ndarray  # (rows, w) float32: reverse-Z Near/viewZ, 0 = cleared

    @property
    def background_fraction(self) -> float:
        return float((self.depth == 0).mean())


def _emissive_hdr(seed, x, y) -> np.ndarray:
    em = hash_unit(seed, x, y, 20) < np.float32(0.01)
    e = [np.where(em, hash_unit(seed, x, y, 21 + c) * np.float32(4.0), np.float32(0.0)) for c in range(3)]
    return _half4(e[0], e[1], e[2], np.ones_like(e[0]))


def quantize_d24(depth: np.ndarray) -> np.ndarray:
    """D24_UNORM as the R24_UNORM_X8 SRV returns it (DeferredRenderer.cpp:3087-3091)."""
    q = np.rint(depth.astype(np.float64) * (2 ** 24 - 1))
    return (q / (2 ** 24 - 1)).astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# SURVEY §8d independent-per-pixel G-buffer
# ---------------------------------------------------------------------------------------------------------------------
def gbuffer_iid(w: int, h: int, seed: int, row0: int = 0, rows: int | None = None, near: float = 0.1) -> GBuffer:
    rows = h - row0 if rows is None else rows
    x, y = _grid(w, row0, rows)
    bg = hash_unit(seed, x >> np.uint32(6), y >> np.uint32(6), 1) < np.float32(0.15)
    # unit normal uniformly on the camera-facing hemisphere (view-space z < 0)
    nz = -np.maximum(hash_unit(seed, x, y, 2), np.float32(1e-3))
    phi = hash_unit(seed, x, y, 3) * np.float32(2.0 * math.pi)
    r = np.sqrt(np.maximum(np.float32(0.0), np.float32(1.0) - nz * nz))
    nx, ny = r * np.cos(phi), r * np.sin(phi)
    z = np.float32(0.5) * np.exp(hash_unit(seed, x, y, 4) * np.float32(math.log(60.0 / 0.5))).astype(np.float32)
    msel = hash_unit(seed, x, y, 5)
    metallic = np.where(msel < 0.1, np.float32(0.0), np.where(msel < 0.2, np.float32(1.0), hash_unit(seed, x, y, 6)))
    rough = np.float32(0.045) + hash_unit(seed, x, y, 7) * np.float32(1.0 - 0.045)
    one, zero = np.ones_like(z), np.zeros_like(z)
    A = _half4(np.where(bg, zero, nx), np.where(bg, zero, ny), np.where(bg, zero, nz), np.where(bg, one, -z))
    B = _half4(np.where(bg, zero, np.float32(0.04)), np.where(bg, zero, metallic), np.where(bg, zero, rough), one)
    c = hash_u32(seed, x, y, 8)
    C = np.where(bg, np.uint32(0xFF000000), (c & np.uint32(0x00FFFFFF)) | np.uint32(0xFF000000)).astype(np.uint32)
    depth = np.where(bg, np.float32(0.0), np.float32(near) / z).astype(np.float32)
    hdr = _emissive_hdr(seed, x, y)
    hdr[bg] = np.array([0, 0, 0, 0x3C00], np.uint16)
    return GBuffer(w, h, row0, rows, A, B, C, hdr, depth)


# ---------------------------------------------------------------------------------------------------------------------
# analytic atrium (world space, y up, metres) — the scene-like bench G-buffer and its shadow map
# ---------------------------------------------------------------------------------------------------------------------
ATRIUM_MIN = np.array([-14.2, 0.0, -11.0], np.float32)
ATRIUM_MAX = np.array([23.0, 14.3, 11.8], np.float32)
AISLE_Z = 3.2          # the roof is open for |z - ZC| < AISLE_Z
ZC = 0.4
_COLUMN_R, _COLUMN_H = 0.45, 7.5


def _columns():
    cols = []
    for zc in (ZC - AISLE_Z, ZC + AISLE_Z):
        for xc in np.arange(-12.0, 21.0, 4.0):
            cols.append((float(xc), float(zc)))
    return cols


_SPHERES = [(6.0, 1.0, 0.4, 1.0), (-2.0, 0.8, -0.9, 0.8), (2.0, 0.6, 1.6, 0.6), (-8.0, 1.4, 0.2, 1.4)]


def _raycast(o: np.ndarray, d: np.ndarray):
    """o: (3,) or (...,3) origins, d: (...,3) unit directions. Returns t (inf = miss), world normal (...,3), primitive id."""
    shape = d.shape[:-1]
    o = np.broadcast_to(np.asarray(o, np.float32), d.shape)
    best_t = np.full(shape, np.inf, np.float32)
    best_n = np.zeros(d.shape, np.float32)
    best_id = np.zeros(shape, np.int32)
    eps = np.float32(1e-4)

    def commit(t, n, pid, mask):
        nonlocal best_t, best_n, best_id
        m = mask & (t > eps) & (t < best_t)
        best_t = np.where(m, t, best_t)
        best_id = np.where(m, np.int32(pid), best_id)
        best_n = np.where(m[..., None], n, best_n)

    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.float32(1.0) / d
        # axis-aligned planes: (axis, coordinate, inward normal sign, id, extra mask fn)
        planes = [
            (1, ATRIUM_MIN[1], +1.0, 1, None),                    # floor
            (2, ATRIUM_MIN[2], +1.0, 2, None), (2, ATRIUM_MAX[2], -1.0, 3, None),  # long walls
            (0, ATRIUM_MIN[0], +1.0, 4, None), (0, ATRIUM_MAX[0], -1.0, 5, None),  # end walls
            (1, ATRIUM_MAX[1], -1.0, 6, "aisle"),                 # ceiling over the side aisles only
        ]
        for axis, coord, sgn, pid, extra in planes:
            t = (np.float32(coord) - o[..., axis]) * inv[..., axis]
            p = o + d * t[..., None]
            inside = np.ones(shape, bool)
            for a in range(3):
                if a != axis:
                    inside &= (p[..., a] >= ATRIUM_MIN[a] - 1e-3) & (p[..., a] <= ATRIUM_MAX[a] + 1e-3)
            if extra == "aisle":
                inside &= np.abs(p[..., 2] - np.float32(ZC)) >= np.float32(AISLE_Z)
            n = np.zeros(d.shape, np.float32)
            n[..., axis] = sgn
            commit(t, n, pid, inside & np.isfinite(t))
        # vertical cylinders
        for k, (xc, zc) in enumerate(_columns()):
            ox, oz = o[..., 0] - np.float32(xc), o[..., 2] - np.float32(zc)
            a = d[..., 0] * d[..., 0] + d[..., 2] * d[..., 2]
            b = ox * d[..., 0] + oz * d[..., 2]
            c = ox * ox + oz * oz - np.float32(_COLUMN_R * _COLUMN_R)
            disc = b * b - a * c
            t = (-b - np.sqrt(np.maximum(disc, 0))) / a
            py = o[..., 1] + d[..., 1] * t
            ok = (disc > 0) & (a > 1e-12) & (py >= 0) & (py <= _COLUMN_H)
            n = np.zeros(d.shape, np.float32)
            n[..., 0] = (ox + d[..., 0] * t) / np.float32(_COLUMN_R)
            n[..., 2] = (oz + d[..., 2] * t) / np.float32(_COLUMN_R)
            commit(t, n, 16 + k, ok & np.isfinite(t))
        # spheres
        for k, (sx, sy, sz, sr) in enumerate(_SPHERES):
            oc = o - np.array([sx, sy, sz], np.float32)
            b = (oc * d).sum(-1)
            c = (oc * oc).sum(-1) - np.float32(sr * sr)
            disc = b * b - c
            t = -b - np.sqrt(np.maximum(disc, 0))
            n = (oc + d * t[..., None]) / np.float32(sr)
            commit(t, n.astype(np.float32), 48 + k, (disc > 0) & np.isfinite(t))
    return best_t, best_n, best_id


def _material(seed: int, pid: np.ndarray, p: np.ndarray, x, y):
    """albedo (linear rgb), metallic, roughness for primitive ids at world points p."""
    pal = np.array([[0.5, 0.5, 0.5], [0.55, 0.48, 0.38], [0.62, 0.55, 0.45], [0.62, 0.55, 0.45], [0.5, 0.42, 0.36],
                    [0.5, 0.42, 0.36], [0.4, 0.38, 0.36]], np.float32)
    is_col = (pid >= 16) & (pid < 48)
    is_sph = pid >= 48
    base = pal[np.clip(pid, 0, 6)]
    base = np.where(is_col[..., None], np.array([0.70, 0.66, 0.58], np.float32), base)
    base = np.where(is_sph[..., None], np.array([0.95, 0.64, 0.35], np.float32), base)
    # brick / tile pattern from world position + a little per-pixel texture noise
    u = p[..., 0] * np.float32(1.7) + p[..., 2] * np.float32(0.9)
    v = p[..., 1] * np.float32(2.3) + p[..., 2] * np.float32(1.3)
    mortar = ((np.abs(u - np.floor(u) - np.float32(0.5)) > np.float32(0.46)) | (np.abs(v - np.floor(v) - np.float32(0.5)) > np.float32(0.44)))
    tone = np.float32(0.85) + np.float32(0.15) * np.sin(np.floor(u) * np.float32(12.9898) + np.floor(v) * np.float32(78.233))
    noise = (hash_unit(seed, x, y, 9) - np.float32(0.5)) * np.float32(0.06)
    albedo = np.clip(base * np.where(mortar, np.float32(0.55), tone)[..., None] + noise[..., None], 0.02, 1.0).astype(np.float32)
    rough = np.clip(np.float32(0.55) + np.float32(0.35) * np.sin(p[..., 0] * np.float32(0.8)) * np.cos(p[..., 2] * np.float32(0.6) + p[..., 1] * np.float32(0.4)),
                    0.08, 1.0).astype(np.float32)
    rough = np.where(pid == 1, np.clip(rough * np.float32(0.5), 0.06, 1.0), rough)   # polished floor
    rough = np.where(is_sph, np.float32(0.2) + np.float32(0.15) * (pid - 48).astype(np.float32), rough)
    metallic = np.where(is_sph, np.float32(1.0), np.float32(0.0)).astype(np.float32)
    return albedo, metallic, rough.astype(np.float32)


def gbuffer_scene(view: np.ndarray, proj: np.ndarray, camera_pos: np.ndarray, w: int, h: int, seed: int, row0: int = 0,
                  rows: int | None = None, chunk_rows: int = 256) -> GBuffer:
    """Ray-cast the atrium through the given camera (row-major, row-vector View/Projection as in hostmath)."""
    rows = h - row0 if rows is None else rows
    V = np.asarray(view, np.float32).reshape(4, 4)
    P = np.asarray(proj, np.float32).reshape(4, 4)
    near = float(P[3, 2])
    Rv = V[:3, :3]  # world -> view rotation (row-vector: v_view = v_world @ Rv)
    out = []
    for c0 in range(0, rows, chunk_rows):
        cr = min(chunk_rows, rows - c0)
        x, y = _grid(w, row0 + c0, cr)
        ndcx = (x.astype(np.float32) + np.float32(0.5)) / np.float32(w) * np.float32(2.0) - np.float32(1.0)
        ndcy = np.float32(1.0) - (y.astype(np.float32) + np.float32(0.5)) / np.float32(h) * np.float32(2.0)
        dv = np.stack([ndcx / P[0, 0], ndcy / P[1, 1], np.ones_like(ndcx)], -1).astype(np.float32)
        inv_len = np.float32(1.0) / np.sqrt((dv * dv).sum(-1))
        dw = (dv @ Rv.T) * inv_len[..., None]
        t, n, pid = _raycast(np.asarray(camera_pos, np.float32), dw.astype(np.float32))
        hit = np.isfinite(t)
        tt = np.where(hit, t, np.float32(1.0))
        p = np.asarray(camera_pos, np.float32) + dw * tt[..., None]
        zview = tt * inv_len  # t * unit_dir_view.z (dv.z == 1)
        # smooth bump so normals are not piecewise constant
        bump = np.stack([np.sin(p[..., 1] * 3.1 + p[..., 2] * 2.3), np.sin(p[..., 0] * 2.7 + p[..., 2] * 3.7), np.sin(p[..., 0] * 3.3 + p[..., 1] * 2.9)], -1)
        nw = n + np.float32(0.08) * bump.astype(np.float32)
        nv = nw @ Rv
        nv = nv / np.sqrt((nv * nv).sum(-1, keepdims=True) + np.float32(1e-20))
        nv[..., 2] = np.minimum(nv[..., 2], np.float32(-0.02))  # keep facing the camera
        nv = nv / np.sqrt((nv * nv).sum(-1, keepdims=True))
        albedo, metallic, rough = _material(seed, pid, p, x, y)
        one, zero = np.ones_like(zview), np.zeros_like(zview)
        A = _half4(np.where(hit, nv[..., 0], zero), np.where(hit, nv[..., 1], zero), np.where(hit, nv[..., 2], zero), np.where(hit, -zview, one))
        B = _half4(np.where(hit, np.float32(0.04), zero), np.where(hit, metallic, zero), np.where(hit, rough, zero), one)
        sr = _srgb_encode(albedo)
        C = np.where(hit, sr[..., 0] | (sr[..., 1] << 8) | (sr[..., 2] << 16) | np.uint32(0xFF000000), np.uint32(0xFF000000)).astype(np.uint32)
        depth = np.where(hit, np.float32(near) / zview, np.float32(0.0)).astype(np.float32)
        hdr = _emissive_hdr(seed, x, y)
        hdr[~hit] = np.array([0, 0, 0, 0x3C00], np.uint16)
        out.append((A, B, C, hdr, depth))
    A, B, C, hdr, depth = (np.concatenate([o[i] for o in out], 0) for i in range(5))
    return GBuffer(w, h, row0, rows, A, B, C, hdr, depth)


def shadow_map_scene(light_view_proj: np.ndarray, size: int = 2048, bias: float = 2e-3, chunk_rows: int = 256) -> np.ndarray:
    """Orthographic light depth of the atrium (standard Z, RendererUtils.cpp:1117-1137), + a constant bias baked in."""
    M = np.asarray(light_view_proj, np.float64).reshape(4, 4)
    Minv = np.linalg.inv(M)
    out = np.empty((size, size), np.float32)
    for r0 in range(0, size, chunk_rows):
        cr = min(chunk_rows, size - r0)
        xs = (np.arange(size) + 0.5) / size * 2.0 - 1.0
        ys = 1.0 - (np.arange(r0, r0 + cr) + 0.5) / size * 2.0
        gx, gy = np.meshgrid(xs, ys)
        p0 = np.stack([gx, gy, np.zeros_like(gx), np.ones_like(gx)], -1) @ Minv
        p1 = np.stack([gx, gy, np.ones_like(gx), np.ones_like(gx)], -1) @ Minv
        o = p0[..., :3]
        seg = p1[..., :3] - o
        seg_len = np.sqrt((seg * seg).sum(-1))
        d = (seg / seg_len[..., None]).astype(np.float32)
        t, _, _ = _raycast(o.astype(np.float32), d)
        z = np.where(np.isfinite(t), t / seg_len.astype(np.float32), np.float32(1.0))
        out[r0:r0 + cr] = np.clip(z + np.float32(bias), 0.0, 1.0)
    return out


def shadow_map_noise(size: int, seed: int) -> np.ndarray:
    """tilted plane + hashed noise in [0.2, 0.9]."""
    x, y = _grid(size, 0, size)
    fx = x.astype(np.float32) / np.float32(size)
    fy = y.astype(np.float32) / np.float32(size)
    s = np.float32(0.2) + np.float32(0.7) * (np.float32(0.45) * fx + np.float32(0.30) * fy + np.float32(0.25) * hash_unit(seed, x, y, 30))
    return s.astype(np.float32)


# ---------------------------------------------------------------------------------------------------------------------
# procedural IBL tables (stand-ins until the shipped DDS assets are decoded — SURVEY.md §8f-2)
# ---------------------------------------------------------------------------------------------------------------------
def _face_dirs(n: int, face: int) -> np.ndarray:
    c = (np.arange(n, dtype=np.float64) + 0.5) / n * 2.0 - 1.0
    s, t = np.meshgrid(c, c)  # s along u (columns), t along v (rows)
    one = np.ones_like(s)
    d = {0: (one, -t, -s), 1: (-one, -t, s), 2: (s, one, t), 3: (s, -one, -t), 4: (s, -t, one), 5: (-s, -t, -one)}[face]
    v = np.stack(d, -1)
    return v / np.sqrt((v * v).sum(-1, keepdims=True))


def env_cube_procedural(base: int = 256, mips: int = 9, sun_dir=(0.0, 0.966, 0.259)) -> np.ndarray:
    """RGBA16F cube in DDS order (face-major, mips inner), flattened (texels, 4) uint16. A sky gradient with a sun lobe
    whose width grows with the mip, i.e. the shape of a prefiltered (pmrem) chain."""
    sun = np.asarray(sun_dir, np.float64)
    sun = sun / np.sqrt((sun * sun).sum())
    faces = []
    for f in range(6):
        chain = []
        for m in range(mips):
            n = max(1, base >> m)
            d = _face_dirs(n, f)
            up = d[..., 1]
            sky = np.stack([0.25 + 0.25 * (1 - up), 0.38 + 0.22 * (1 - up), 0.62 + 0.12 * (1 - up)], -1)
            ground = np.array([0.16, 0.14, 0.12])
            blend = 1.0 / (1.0 + np.exp(-up * (24.0 / (1 + m * m))))
            col = ground * (1 - blend[..., None]) + sky * blend[..., None]
            sharp = 2048.0 / (4.0 ** m) + 1.0
            lobe = np.power(np.clip((d * sun).sum(-1), 0, 1), sharp) * (24.0 / (1.0 + 1.5 * m * m))
            col = col + lobe[..., None] * np.array([1.0, 0.92, 0.78])
            rgba = np.concatenate([col, np.ones_like(col[..., :1])], -1).astype(np.float32)
            chain.append(rgba.astype(np.float16).view(np.uint16).reshape(-1, 4))
        faces.append(np.concatenate(chain, 0))
    return np.ascontiguousarray(np.concatenate(faces, 0))


def brdf_lut_procedural(w: int = 128, h: int = 32) -> np.ndarray:
    """(h, w, 2) uint16 RG16_UNORM: Karis' analytic fit of the split-sum scale/bias; x = NdotV, y = roughness."""
    nv = (np.arange(w, dtype=np.float64) + 0.5) / w
    r = (np.arange(h, dtype=np.float64) + 0.5) / h
    NV, R = np.meshgrid(nv, r)
    c0 = np.array([-1.0, -0.0275, -0.572, 0.022])
    c1 = np.array([1.0, 0.0425, 1.04, -0.04])
    rr = [R * c0[i] + c1[i] for i in range(4)]
    a004 = np.minimum(rr[0] * rr[0], np.exp2(-9.28 * NV)) * rr[0] + rr[1]
    A = np.clip(-1.04 * a004 + rr[2], 0, 1)
    B = np.clip(1.04 * a004 + rr[3], 0, 1)
    return np.ascontiguousarray(np.stack([np.rint(A * 65535), np.rint(B * 65535)], -1).astype(np.uint16))


# ---------------------------------------------------------------------------------------------------------------------
# instances
# ---------------------------------------------------------------------------------------------------------------------
def instances_random(n: int, seed: int, center=(0.0, 0.0, 0.0), box: float = 400.0, first: int = 0) -> np.ndarray:
    """(n, 2, 4) float32 AABBs: centre uniform in a box^3 cube around `center`, half-extent log-uniform [0.05, 5] per axis
    (SURVEY §8d config 5). `first` offsets the instance index so ranks can generate disjoint ranges."""
    i = np.arange(first, first + n, dtype=np.uint32)
    z = np.zeros_like(i)
    c = np.stack([(hash_unit(seed, i, z, 40 + a) - np.float32(0.5)) * np.float32(box) + np.float32(center[a]) for a in range(3)], -1)
    e = np.stack([np.float32(0.05) * np.exp(hash_unit(seed, i, z, 43 + a) * np.float32(math.log(5.0 / 0.05))) for a in range(3)], -1).astype(np.float32)
    b = np.zeros((n, 2, 4), np.float32)
    b[:, 0, :3] = c - e
    b[:, 1, :3] = c + e
    return b


def instances_replicated(aabb_min, aabb_max, count: int) -> np.ndarray:
    """Sponza: 25 draw commands that all share the mesh-level AABB."""
    b = np.zeros((count, 2, 4), np.float32)
    b[:, 0, :3] = np.asarray(aabb_min, np.float32)
    b[:, 1, :3] = np.asarray(aabb_max, np.float32)
    return b


def indirect_args_initial(n: int) -> np.ndarray:
    """n 64-byte FIndirectDrawCommands as uint32[n,16] with InstanceCount = 1, StartInstanceLocation = index and
    recognisable filler elsewhere, so tests can prove only dword 11 (byte 44) is written (DeferredRenderer.cpp:3352-3356)."""
    a = np.empty((n, 16), np.uint32)
    a[:] = (np.arange(16, dtype=np.uint32) * np.uint32(0x01010101) + np.uint32(0xA5000000))[None, :]
    a[:, 10] = 36          # IndexCountPerInstance
    a[:, 11] = 1           # InstanceCount  <- byte 44
    a[:, 14] = np.arange(n, dtype=np.uint32)  # StartInstanceLocation
    return a
