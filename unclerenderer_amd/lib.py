"""ctypes binding of the C-ABI (include/ur_hotpath.h, include/ur_host.h).

The shared library is the product; there is no Python or CPU fallback. If it is missing, importing the compute
entry points raises — build it with `python -m unclerenderer_amd.build` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "_build" / "libur_hotpath.so"
if os.environ.get("UR_HOTPATH_LIB"):  # diagnostic builds (e.g. the in-kernel-stamps build of tools/stamps_lighting.py)
    _LIB_PATH = Path(os.environ["UR_HOTPATH_LIB"]).resolve()

UR_OK = 0
UR_EINVAL, UR_EHIP, UR_ENOMEM, UR_ENODEVICE, UR_EUNSUPPORTED, UR_ETIMEOUT = -1, -2, -3, -4, -5, -6
UR_MAX_HZB_MIPS = 16
UR_CULL_CONSTANT_DWORDS = 46
UR_INDIRECT_COMMAND_STRIDE = 64
UR_INDIRECT_INSTANCE_COUNT_OFFSET = 44


# ur_set_option keys (include/ur_hotpath.h)
UR_OPT_LIGHTING_STREAM, UR_OPT_LIGHTING_WAVES_PER_WG, UR_OPT_LIGHTING_TILED_WAVES, UR_OPT_LIGHTING_LEAVE_CUS, UR_OPT_RIDE_WALKERS = 1, 2, 3, 4, 5
UR_OPT_CULL_STORE, UR_OPT_LIGHTING_BALANCE, UR_OPT_BALANCE_POOL_16THS, UR_OPT_BALANCE_CHUNK_SHIFT, UR_OPT_DEBUG_HZB_RIDE_STALL = 7, 8, 9, 10, 11


class MipDesc(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


class HzbSlice(C.Structure):
    """ur_hzb_slice: one contiguous run of floats of the HZB allocation."""
    _fields_ = [("offset", C.c_uint32), ("count", C.c_uint32)]


class SceneConstants(C.Structure):
    """FSceneConstants (Source/Render/RendererUtils.h:41-79)."""
    _fields_ = [
        ("World", C.c_float * 16), ("View", C.c_float * 16), ("ViewInverse", C.c_float * 16), ("Projection", C.c_float * 16),
        ("BaseColor", C.c_float * 3), ("LightIntensity", C.c_float),
        ("LightDirection", C.c_float * 3), ("Padding1", C.c_float),
        ("CameraPosition", C.c_float * 3), ("Padding2", C.c_float),
        ("LightColor", C.c_float * 3), ("Padding3", C.c_float),
        ("EmissiveFactor", C.c_float * 3), ("Padding4", C.c_float),
        ("LightViewProjection", C.c_float * 16),
        ("ShadowStrength", C.c_float), ("ShadowBias", C.c_float), ("ShadowMapSize", C.c_float * 2),
        ("MetallicFactor", C.c_float), ("RoughnessFactor", C.c_float), ("BaseColorAlpha", C.c_float), ("AlphaCutoff", C.c_float),
        ("AlphaMode", C.c_uint32), ("PaddingMaterial", C.c_uint32 * 3),
        ("BaseColorTransformOffsetScale", C.c_float * 4), ("BaseColorTransformRotation", C.c_float * 4),
        ("MetallicRoughnessTransformOffsetScale", C.c_float * 4), ("MetallicRoughnessTransformRotation", C.c_float * 4),
        ("NormalTransformOffsetScale", C.c_float * 4), ("NormalTransformRotation", C.c_float * 4),
        ("EmissiveTransformOffsetScale", C.c_float * 4), ("EmissiveTransformRotation", C.c_float * 4),
        ("EnvMapMipCount", C.c_float), ("PaddingEnvMap", C.c_float * 3),
        ("ObjectId", C.c_uint32), ("PaddingObjectId", C.c_float * 3),
    ]


class SkyConstants(C.Structure):
    """FSkyAtmosphereConstants (Source/Render/RendererUtils.h:81-92)."""
    _fields_ = [
        ("World", C.c_float * 16), ("View", C.c_float * 16), ("Projection", C.c_float * 16),
        ("CameraPosition", C.c_float * 3), ("Padding0", C.c_float),
        ("LightDirection", C.c_float * 3), ("Padding1", C.c_float),
        ("LightColor", C.c_float * 3), ("Padding2", C.c_float),
    ]


class LightingTables(C.Structure):
    _fields_ = [
        ("shadow_map", C.c_void_p), ("env_cube", C.c_void_p), ("env_base_size", C.c_uint32), ("env_mip_count", C.c_uint32),
        ("brdf_lut_rg16", C.c_void_p), ("lut_width", C.c_uint32), ("lut_height", C.c_uint32), ("env_cube_texels", C.c_uint64),
    ]


class DdsInfo(C.Structure):
    """ur_dds_info (include/ur_assets.h)."""
    _fields_ = [(n, C.c_uint32) for n in ("width", "height", "mip_count", "slices", "is_cube", "dxgi_format", "header_size", "block_dim", "bytes_per_block")]


class SceneModel(C.Structure):
    """ur_scene_model (include/ur_scene.h)."""
    _fields_ = [("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3), ("center", C.c_float * 3), ("radius", C.c_float),
                ("pipeline_key", C.c_uint32), ("material_index", C.c_uint32), ("model_index", C.c_uint32), ("node_order", C.c_uint32),
                ("mesh_index", C.c_uint32), ("primitive_index", C.c_uint32)]


class SceneSummary(C.Structure):
    _fields_ = [("model_count", C.c_uint32), ("scene_center", C.c_float * 3), ("scene_radius", C.c_float)]


class TonemapConstants(C.Structure):
    """TonemapParams (Shaders/Tonemap.hlsl:22-28)."""
    _fields_ = [("EnableTonemap", C.c_uint32), ("EnableAutoExposure", C.c_uint32), ("Exposure", C.c_float), ("Gamma", C.c_float)]


class FrameResources(C.Structure):
    """ur_frame_resources (include/ur_frame.h)."""
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("row0", C.c_uint32), ("rows", C.c_uint32),
        ("gbuffer_a", C.c_void_p), ("gbuffer_b", C.c_void_p), ("gbuffer_c", C.c_void_p), ("depth_band", C.c_void_p),
        ("lighting_band", C.c_void_p), ("depth_full", C.c_void_p), ("hzb", C.c_void_p),
        ("hzb_mips", MipDesc * UR_MAX_HZB_MIPS), ("hzb_mip_count", C.c_uint32),
        ("tables", LightingTables),
        ("model_bounds", C.c_void_p), ("indirect_args", C.c_void_p), ("indirect_command_count", C.c_uint32),
        ("instance_index_base", C.c_uint32), ("visible_indices", C.c_void_p), ("visible_count", C.c_void_p), ("cull_stats", C.c_void_p),
        ("tonemap_band", C.c_void_p),
    ]


UR_FRAME_INDIRECT_DRAW, UR_FRAME_HZB, UR_FRAME_DEPTH_PREPASS, UR_FRAME_SHADOWS, UR_FRAME_SKY = 0x1, 0x2, 0x4, 0x8, 0x10
UR_FRAME_FUSE_LIGHTING_SKY, UR_FRAME_GPU_TIMING, UR_FRAME_GRAPH_DUMP, UR_FRAME_BARRIER_LOGS = 0x20, 0x40, 0x80, 0x100
UR_FRAME_ASYNC_COMPUTE, UR_FRAME_ASYNC_NO_JOIN, UR_FRAME_TONEMAP, UR_FRAME_TIME_LIGHTING = 0x200, 0x400, 0x800, 0x1000
UR_FRAME_HZB_TAIL_WITH_LIGHTING = 0x2000
UR_FRAME_HZB_WITH_LIGHTING = 0x4000
UR_FRAME_TIME_LIGHTING_RECORD_COST = 0x8000
UR_FRAME_TIME_LIGHTING_KERNEL = 0x10000
UR_FRAME_HZB_SHARD = 0x20000
UR_FRAME_DEFAULT = UR_FRAME_INDIRECT_DRAW | UR_FRAME_HZB | UR_FRAME_DEPTH_PREPASS | UR_FRAME_SHADOWS | UR_FRAME_SKY

assert C.sizeof(SceneConstants) == 608 and C.sizeof(SkyConstants) == 240

# name -> (restype, argtypes); every symbol declared in include/*.h
_VP, _U32, _F = C.c_void_p, C.c_uint32, C.c_float
_FP = C.POINTER(C.c_float)
SIGNATURES = {
    # ur_hotpath.h
    "ur_create": (_VP, [C.c_int, _VP]),
    "ur_destroy": (None, [_VP]),
    "ur_reserve": (C.c_int, [_VP, _U32]),
    "ur_defer_hzb_tail": (C.c_int, [_VP, C.c_int]),
    "ur_flush": (C.c_int, [_VP]),
    "ur_debug_set_hzb_timeout": (C.c_int, [_VP]),
    "ur_debug_lighting_schedule": (C.c_int, [_VP, C.POINTER(_U32)]),
    "ur_debug_stream_ceiling": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_uint64, _VP, _VP]),
    "ur_debug_timeline": (C.c_int, [_VP, _VP, _U32]),
    "ur_time_next_lighting": (C.c_int, [_VP, _VP, _VP]),
    "ur_time_next_cull": (C.c_int, [_VP, _VP]),
    "ur_time_cull_carried": (C.c_int, [_VP]),
    "ur_set_option": (C.c_int, [_VP, C.c_int, C.c_int]),
    "ur_get_option": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_int)]),
    "ur_last_error": (C.c_char_p, []),
    "ur_version": (C.c_char_p, []),
    "ur_hzb_layout": (_U32, [_U32, _U32, C.POINTER(MipDesc), C.POINTER(_U32)]),
    "ur_build_hzb": (C.c_int, [_VP, _VP, _U32, _U32, _VP, C.POINTER(MipDesc), _U32]),
    "ur_hzb_band_pieces": (C.c_int, [_U32, _U32, _U32, C.POINTER(_U32), C.POINTER(_U32)]),
    "ur_hzb_band_slices": (C.c_int, [C.POINTER(MipDesc), _U32, _U32, _U32, C.POINTER(HzbSlice)]),
    "ur_build_hzb_band": (C.c_int, [_VP, _VP, _U32, _U32, _VP, C.POINTER(MipDesc), _U32, _U32, _U32]),
    "ur_build_hzb_tail": (C.c_int, [_VP, _VP, C.POINTER(MipDesc), _U32]),
    "ur_cull_indirect_args": (C.c_int, [_VP, C.POINTER(_U32), _VP, _VP, C.POINTER(MipDesc), _VP, _VP, _VP, _VP]),
    "ur_cull_indirect_args_ex": (C.c_int, [_VP, C.POINTER(_U32), _VP, _VP, C.POINTER(MipDesc), _VP, _VP, _VP, _VP, _U32]),
    "ur_env_cube_texels": (C.c_size_t, [_U32, _U32]),
    "ur_stage_env_cube": (C.c_int, [_VP, _VP, _U32, _U32, _VP]),
    "ur_deferred_lighting": (C.c_int, [_VP, C.POINTER(SceneConstants), _VP, _VP, _VP, C.POINTER(LightingTables), _VP, _U32, _U32, _U32, _U32]),
    "ur_sky_atmosphere": (C.c_int, [_VP, C.POINTER(SkyConstants), _VP, _VP, _U32, _U32, _U32, _U32]),
    "ur_deferred_lighting_sky": (C.c_int, [_VP, C.POINTER(SceneConstants), C.POINTER(SkyConstants), _VP, _VP, _VP, _VP,
                                           C.POINTER(LightingTables), _VP, _U32, _U32, _U32, _U32]),
    "ur_tonemap": (C.c_int, [_VP, C.POINTER(TonemapConstants), _VP, _VP, _VP, _U32, _U32]),
    "ur_temporal_aa": (C.c_int, [_VP, _VP, _VP, _VP, _F, _U32, _U32, _U32, _U32, _U32]),
    "ur_allgather_rows": (C.c_int, [_VP, _VP, _VP, _U32, _U32, _U32, _U32]),
    "ur_allgather_rows_bytes": (C.c_int, [_VP, _VP, _VP, _U32, _U32, _U32, _U32]),
    "ur_allgather_rows_bytes_ex": (C.c_int, [_VP, _VP, _VP, _U32, _U32, _U32, _U32, C.c_int]),
    # ur_assets.h
    "ur_dds_parse": (C.c_int, [_VP, C.c_size_t, C.POINTER(DdsInfo)]),
    "ur_dds_texel_count": (C.c_size_t, [C.POINTER(DdsInfo)]),
    "ur_dds_decode_rgba16f": (C.c_int, [_VP, C.c_size_t, C.POINTER(DdsInfo), _VP, C.POINTER(_U32)]),
    "ur_dds_copy_rg16": (C.c_int, [_VP, C.c_size_t, C.POINTER(DdsInfo), _VP]),
    "ur_bc6h_decode_block": (C.c_int, [_VP, C.c_int, _VP]),
    "ur_bc6h_block_endpoints": (C.c_int, [_VP, C.c_int, _VP]),
    # ur_scene.h
    "ur_scene_model_count": (C.c_int, [C.c_char_p]),
    "ur_scene_model_path": (C.c_int, [C.c_char_p, _U32, C.c_char_p, _U32]),
    "ur_scene_extract": (C.c_int, [C.c_char_p, C.POINTER(C.c_char_p), _U32, C.POINTER(SceneModel), _U32, C.POINTER(SceneSummary)]),
    # ur_frame.h
    "ur_frame_create": (_VP, [_VP, _VP, _U32, C.c_int, C.c_int]),
    "ur_frame_destroy": (None, [_VP]),
    "ur_frame_render": (C.c_int, [_VP, C.POINTER(FrameResources), C.POINTER(_U32), C.POINTER(SceneConstants), C.POINTER(SkyConstants), _U32]),
    "ur_frame_join_async": (None, [_VP]),
    "ur_frame_lighting_times": (_U32, [_VP, _FP, _U32]),
    "ur_frame_lighting_times_ex": (_U32, [_VP, _FP, _FP, _U32]),
    "ur_frame_hzb_ready": (C.c_int, [_VP]),
    "ur_frame_reset_hzb": (None, [_VP]),
    "ur_frame_report": (_U32, [_VP, C.c_char_p, _U32]),
    "ur_rg_timing_stats": (_U32, [C.c_char_p, _U32]),
    # ur_host.h
    "ur_host_look_to_lh": (None, [_FP, _FP, _FP, _FP]),
    "ur_host_look_at_lh": (None, [_FP, _FP, _FP, _FP]),
    "ur_host_reverse_z_projection": (None, [_F, _F, _F, _FP]),
    "ur_host_orthographic_lh": (None, [_F, _F, _F, _F, _FP]),
    "ur_host_mat_mul": (None, [_FP, _FP, _FP]),
    "ur_host_mat_inverse": (C.c_int, [_FP, _FP]),
    "ur_host_frustum_planes": (None, [_FP, _FP]),
    "ur_host_is_aabb_in_frustum": (C.c_int, [_FP, _FP, _FP]),
    "ur_host_light_view_projection": (None, [_FP, _F, _FP, _FP]),
    "ur_host_pack_culling_constants": (None, [_FP, _FP, _U32, _U32, _U32, _U32, _U32, _U32, C.POINTER(_U32)]),
    "ur_host_fill_scene_constants": (None, [_FP, _FP, _FP, _F, _FP, _FP, _FP, _F, _F, _F, _F, _F, C.POINTER(SceneConstants)]),
    "ur_host_fill_sky_constants": (None, [_FP, _FP, _FP, _F, _FP, _FP, C.POINTER(SkyConstants)]),
    "ur_host_direction_from_euler_degrees": (None, [_F, _F, _FP]),
    "ur_host_camera_forward_from_euler_degrees": (None, [_F, _F, _FP]),
    "ur_host_light_direction_roundtrip": (None, [_FP, _FP]),
}

_lib = None


def library_path() -> Path:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load libur_hotpath.so; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise RuntimeError(
            f"{_LIB_PATH} is missing: the HIP extension is not built. Run `python -m unclerenderer_amd.build`. "
            "There is no CPU fallback for the hot path.")
    # One HIP runtime per process: torch bundles its own libamdhip64 (SONAME libamdhip64.so.7) but asks for it by file
    # name, so if the system copy were loaded first the process would end up with two runtimes and the second one sees
    # no device (KFD allows one open per process). Importing torch first makes our NEEDED libamdhip64.so.7 resolve to
    # the copy torch already loaded. A C++ host that does not use torch simply gets /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(str(_LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class UrError(RuntimeError):
    def __init__(self, code: int, where: str):
        msg = load().ur_last_error().decode(errors="replace")
        super().__init__(f"{where} failed with {code}: {msg}")
        self.code = code


def check(code: int, where: str) -> None:
    if code != UR_OK:
        raise UrError(code, where)


def fptr(a: np.ndarray):
    """float32 numpy array -> float* (the array must stay alive for the duration of the call)."""
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(_FP)
