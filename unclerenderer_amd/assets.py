"""IBL asset loading: DDS (BC6H cube, RG16 LUT) -> host arrays ready for HotPath.stage_env_cube / make_tables.
All decoding happens in libur_hotpath.so (csrc/dds.cpp); this module only reads the file and marshals."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import lib as _lib


def parse_dds(data: bytes) -> _lib.DdsInfo:
    info = _lib.DdsInfo()
    rc = _lib.load().ur_dds_parse(data, len(data), C.byref(info))
    if rc != 0:
        raise ValueError(f"ur_dds_parse failed ({rc})")
    return info


def load_env_cube_dds(path) -> tuple[np.ndarray, int, int, int]:
    """Returns (texels (n,4) uint16 in DDS order, base size, mip count, reserved-block count)."""
    data = Path(path).read_bytes()
    info = parse_dds(data)
    if not info.is_cube or info.slices != 6 or info.width != info.height:
        raise ValueError("not a single cube map")
    n = int(_lib.load().ur_dds_texel_count(C.byref(info)))
    out = np.zeros((n, 4), np.uint16)
    bad = C.c_uint32(0)
    rc = _lib.load().ur_dds_decode_rgba16f(data, len(data), C.byref(info), out.ctypes.data_as(C.c_void_p), C.byref(bad))
    if rc != 0:
        raise ValueError(f"ur_dds_decode_rgba16f failed ({rc})")
    return out, int(info.width), int(info.mip_count), int(bad.value)


def load_brdf_lut_dds(path) -> np.ndarray:
    """Returns (h, w, 2) uint16 R16G16_UNORM."""
    data = Path(path).read_bytes()
    info = parse_dds(data)
    out = np.zeros((info.height, info.width, 2), np.uint16)
    rc = _lib.load().ur_dds_copy_rg16(data, len(data), C.byref(info), out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ValueError(f"ur_dds_copy_rg16 failed ({rc})")
    return out
