"""Caller-side constants for the hot path: thin Python over the C++ host math (include/ur_host.h), plus the camera /
light presets of the reference's shipped scenes (Assets/Scenes/{sponza,Duck,pica_pica}.json).

Everything numeric is computed by libur_hotpath.so (csrc/host_math.cpp); this module only marshals.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field

import numpy as np

from . import lib as _lib


def _f(*v) -> np.ndarray:
    return np.asarray(v, dtype=np.float32).reshape(-1).copy()


def look_to_lh(eye, direction, up=(0.0, 1.0, 0.0)) -> np.ndarray:
    out = np.zeros(16, np.float32)
    _lib.load().ur_host_look_to_lh(_lib.fptr(_f(*eye)), _lib.fptr(_f(*direction)), _lib.fptr(_f(*up)), _lib.fptr(out))
    return out


def reverse_z_projection(fov_y: float, aspect: float, near: float = 0.1) -> np.ndarray:
    out = np.zeros(16, np.float32)
    _lib.load().ur_host_reverse_z_projection(fov_y, aspect, near, _lib.fptr(out))
    return out


def mat_mul(a, b) -> np.ndarray:
    out = np.zeros(16, np.float32)
    _lib.load().ur_host_mat_mul(_lib.fptr(_f(*a)), _lib.fptr(_f(*b)), _lib.fptr(out))
    return out


def mat_inverse(m) -> np.ndarray:
    out = np.zeros(16, np.float32)
    if not _lib.load().ur_host_mat_inverse(_lib.fptr(_f(*m)), _lib.fptr(out)):
        raise ValueError("singular matrix")
    return out


def frustum_planes(view_proj) -> np.ndarray:
    out = np.zeros(24, np.float32)
    _lib.load().ur_host_frustum_planes(_lib.fptr(_f(*view_proj)), _lib.fptr(out))
    return out


def is_aabb_in_frustum(planes, bmin, bmax) -> bool:
    return bool(_lib.load().ur_host_is_aabb_in_frustum(_lib.fptr(_f(*planes)), _lib.fptr(_f(*bmin)), _lib.fptr(_f(*bmax))))


def light_view_projection(center, radius: float, light_dir) -> np.ndarray:
    out = np.zeros(16, np.float32)
    _lib.load().ur_host_light_view_projection(_lib.fptr(_f(*center)), radius, _lib.fptr(_f(*light_dir)), _lib.fptr(out))
    return out


def pack_culling_constants(view, proj, model_count: int, hzb_enabled: bool, hzb_mip_count: int, hzb_width: int,
                           hzb_height: int, debug_print: bool = False) -> np.ndarray:
    out = np.zeros(_lib.UR_CULL_CONSTANT_DWORDS, np.uint32)
    _lib.load().ur_host_pack_culling_constants(_lib.fptr(_f(*view)), _lib.fptr(_f(*proj)), model_count, int(hzb_enabled),
                                               hzb_mip_count, hzb_width, hzb_height, int(debug_print),
                                               out.ctypes.data_as(C.POINTER(C.c_uint32)))
    return out


def direction_from_euler_degrees(pitch: float, yaw: float) -> np.ndarray:
    out = np.zeros(3, np.float32)
    _lib.load().ur_host_direction_from_euler_degrees(pitch, yaw, _lib.fptr(out))
    return out


def camera_forward_from_euler_degrees(pitch: float, yaw: float) -> np.ndarray:
    out = np.zeros(3, np.float32)
    _lib.load().ur_host_camera_forward_from_euler_degrees(pitch, yaw, _lib.fptr(out))
    return out


def light_direction_roundtrip(json_dir) -> np.ndarray:
    out = np.zeros(3, np.float32)
    _lib.load().ur_host_light_direction_roundtrip(_lib.fptr(_f(*json_dir)), _lib.fptr(out))
    return out


@dataclass
class ScenePreset:
    """Camera/light values copied from Assets/Scenes/*.json; bounds as the renderer derives them (RendererUtils.cpp:46-82,
    277-286,533-540: per-model sphere bounds -> scene box -> centre/radius)."""
    name: str
    camera_position: tuple
    camera_rotation_deg: tuple | None = None  # (pitch, yaw, roll)
    camera_look_at: tuple | None = None
    fov_y_deg: float = 60.0
    light_rotation_deg: tuple | None = None
    light_direction: tuple | None = None
    light_intensity: float = 1.0
    light_color: tuple = (1.0, 1.0, 1.0)
    # world AABB of the scene's models (min, max); Sponza: accessor min/max of Assets/sponza/untitled.gltf through the
    # node's +90deg X rotation, LH z flip (GltfLoader.cpp:823), scale 0.01, translate (5,0,0) (sponza.json)
    model_aabb: tuple = ((-1.0, -1.0, -1.0), (1.0, 1.0, 1.0))
    instance_count: int = 1
    # scene centre / radius as CreateSceneModelsFromJson derives them (union of per-command spheres); values produced by
    # ur_scene_extract (csrc/scene.cpp) on the shipped scene files and pinned by tests/test_scene.py
    scene_center: tuple | None = None
    scene_radius: float | None = None


SCENES = {
    "sponza": ScenePreset("sponza", (14.327, 0.762, 0.571), camera_rotation_deg=(-12.6, 261.8, 0.0), fov_y_deg=60.0,
                          light_rotation_deg=(-75.0, 0.0, 0.0), light_intensity=1.0, light_color=(1.0, 1.0, 1.0),
                          model_aabb=((-14.209459, -1.264425, -11.054260), (22.999081, 14.294332, 11.828071)), instance_count=25,
                          scene_center=(4.3948116, 6.5149517, 0.38690376), scene_radius=40.15736),
    "duck": ScenePreset("duck", (0.0, 1.5, 4.0), camera_look_at=(0.0, 1.0, 0.0), fov_y_deg=60.0,
                        light_direction=(-0.5, -1.0, -0.3), light_intensity=3.0, light_color=(1.0, 0.95, 0.9),
                        model_aabb=((-0.692985, 0.09929369, -0.539252), (0.96179897, 1.6396999, 0.61328197)), instance_count=1,
                        scene_center=(0.13440704, 0.8694968, 0.03701496), scene_radius=2.1976402),
    "pica_pica": ScenePreset("pica_pica", (-13.482, 20.457, -42.455), camera_rotation_deg=(19.199, 16.599, 0.0), fov_y_deg=60.0,
                             light_rotation_deg=(-45.0, -135.0, 0.0), light_intensity=1.0, light_color=(1.0, 0.95, 0.9),
                             model_aabb=((-36.92157, -1.1695042, -18.168373), (27.365425, 18.529217, 32.19241)), instance_count=170,
                             scene_center=(-3.2157097, 3.8195744, 1.1750641), scene_radius=62.74809),
}


@dataclass
class FrameConstants:
    width: int
    height: int
    view: np.ndarray
    proj: np.ndarray
    camera_position: np.ndarray
    light_direction: np.ndarray
    scene: _lib.SceneConstants
    sky: _lib.SkyConstants
    scene_center: np.ndarray
    scene_radius: float
    sky_radius: float
    near: float = 0.1
    extra: dict = field(default_factory=dict)


def build_frame_constants(preset: ScenePreset | str, width: int, height: int, *, shadow_strength: float = 1.0,
                          shadow_bias: float = 0.0, shadow_size: int = 2048, env_mip_count: int = 9,
                          near: float = 0.1) -> FrameConstants:
    """What FApplication/FDeferredRenderer compute per frame before recording the four passes."""
    if isinstance(preset, str):
        preset = SCENES[preset]
    L = _lib.load()
    pos = _f(*preset.camera_position)
    if preset.camera_look_at is not None:
        d = _f(*preset.camera_look_at) - pos
        fwd = (d / np.float32(np.sqrt(np.float32((d * d).sum())))).astype(np.float32)
    else:
        fwd = camera_forward_from_euler_degrees(preset.camera_rotation_deg[0], preset.camera_rotation_deg[1])
    view = look_to_lh(pos, fwd)
    proj = reverse_z_projection(math.radians(preset.fov_y_deg), width / height, near)
    if preset.light_direction is not None:
        jd = _f(*preset.light_direction)
    else:
        jd = direction_from_euler_degrees(preset.light_rotation_deg[0], preset.light_rotation_deg[1])
    light_dir = light_direction_roundtrip(jd)
    if preset.scene_center is not None and preset.scene_radius is not None:
        center, scene_radius = _f(*preset.scene_center), float(preset.scene_radius)
    else:  # one model sphere (centre of the AABB, half its diagonal, >= 1) -> box -> centre / radius
        mn, mx = _f(*preset.model_aabb[0]), _f(*preset.model_aabb[1])
        center = ((mn + mx) * np.float32(0.5)).astype(np.float32)
        model_radius = max(float(np.sqrt(((mx - mn) ** 2).sum())) * 0.5, 1.0)
        scene_radius = max(float(np.sqrt(3.0 * (2.0 * model_radius) ** 2)) * 0.5, 1.0)
    sky_radius = max(scene_radius * 5.0, 100.0)  # DeferredRenderer.cpp:349
    lvp = light_view_projection(center, scene_radius, light_dir)
    scene = _lib.SceneConstants()
    L.ur_host_fill_scene_constants(_lib.fptr(view), _lib.fptr(proj), _lib.fptr(pos), preset.light_intensity, _lib.fptr(light_dir),
                                   _lib.fptr(_f(*preset.light_color)), _lib.fptr(lvp), shadow_strength, shadow_bias,
                                   float(shadow_size), float(shadow_size), float(env_mip_count), C.byref(scene))
    sky = _lib.SkyConstants()
    L.ur_host_fill_sky_constants(_lib.fptr(view), _lib.fptr(proj), _lib.fptr(pos), sky_radius, _lib.fptr(light_dir),
                                 _lib.fptr(_f(*preset.light_color)), C.byref(sky))
    return FrameConstants(width, height, view, proj, pos, light_dir, scene, sky, center, scene_radius, sky_radius, near)
