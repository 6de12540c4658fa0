"""The C-ABI library loads and exports every symbol include/*.h declares; host-only entry points behave (no GPU)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    names = set()
    for h in (ROOT / "include").glob("*.h"):
        text = re.sub(r"/\*.*?\*/", "", h.read_text(), flags=re.S)
        names |= set(re.findall(r"\b(ur_[a-z0-9_]+)\s*\(", text))
    return names


def test_every_declared_symbol_is_exported_and_bound(urlib):
    from unclerenderer_amd import lib
    decl = declared_symbols()
    assert len(decl) >= 35
    for name in sorted(decl):
        assert hasattr(urlib, name), f"{name} declared in include/ but not exported"
    assert decl == set(lib.SIGNATURES), (decl ^ set(lib.SIGNATURES))


def test_struct_layouts(urlib):
    from unclerenderer_amd import lib
    assert C.sizeof(lib.SceneConstants) == 608  # sizeof(FSceneConstants), RendererUtils.h:41-79
    assert C.sizeof(lib.SkyConstants) == 240    # sizeof(FSkyAtmosphereConstants)
    assert lib.SceneConstants.LightViewProjection.offset == 336
    assert lib.SceneConstants.ShadowStrength.offset == 400
    assert lib.SceneConstants.EnvMapMipCount.offset == 576
    assert C.sizeof(lib.MipDesc) == 12
    assert C.sizeof(lib.LightingTables) == 48 and lib.LightingTables.env_cube_texels.offset == 40  # ur_lighting_tables (the layout tag is its last field)
    assert C.sizeof(lib.HzbSlice) == 8
    assert lib.UR_INDIRECT_COMMAND_STRIDE == 64 and lib.UR_INDIRECT_INSTANCE_COUNT_OFFSET == 44


def test_hzb_layout_matches_create_hzb_resources(urlib):
    """CreateHZBResources (DeferredRenderer.cpp:2801-2835): chains listed in SURVEY.md §8 a10."""
    from unclerenderer_amd.hotpath import HzbLayout
    dims = lambda w, h: [(m[1], m[2]) for m in HzbLayout(w, h).as_list()]
    assert dims(1920, 1080) == [(960, 540), (480, 270), (240, 135), (120, 67), (60, 33), (30, 16), (15, 8), (7, 4), (3, 2), (1, 1)]
    assert len(dims(512, 512)) == 9 and dims(512, 512)[0] == (256, 256)
    assert len(dims(3840, 2160)) == 11 and len(dims(7680, 4320)) == 12
    assert dims(1, 1) == [(1, 1)] and dims(3, 5) == [(2, 3), (1, 1)]
    lay = HzbLayout(3840, 2160)
    assert sum(w * h for _, w, h in lay.as_list()) == 2764655
    assert all(off % 64 == 0 for off, _, _ in lay.as_list())
    mips = (urlib.ur_hzb_layout.argtypes[2]._type_ * 16)()
    n = C.c_uint32(0)
    assert urlib.ur_hzb_layout(0, 4, mips, C.byref(n)) == 0


def test_env_cube_texels(urlib):
    e = [max(1, 256 >> m) + 2 for m in range(9)]
    assert urlib.ur_env_cube_texels(256, 9) == 6 * sum(x * x for x in e) + 9 * sum(x * (x - 1) for x in e)  # bordered faces + RGB row pairs (12-byte entries, in 8-byte units)
    assert urlib.ur_env_cube_texels(0, 9) == 0 and urlib.ur_env_cube_texels(256, 17) == 0


def test_no_gpu_fails_loudly(urlib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    assert not urlib.ur_create(0, None)
    assert b"no HIP device" in urlib.ur_last_error()
    from unclerenderer_amd.hotpath import HotPath
    with pytest.raises(RuntimeError):
        HotPath(0)


def test_null_context_is_einval(urlib):
    from unclerenderer_amd import lib
    assert urlib.ur_build_hzb(None, None, 0, 0, None, None, 0) == lib.UR_EINVAL
    assert urlib.ur_cull_indirect_args(None, None, None, None, None, None, None, None, None) == lib.UR_EINVAL
    assert urlib.ur_reserve(None, 10) == lib.UR_EINVAL
    assert urlib.ur_frame_render(None, None, None, None, None, 0) == lib.UR_EINVAL
    assert b"gfx950" in urlib.ur_version()


def test_product_never_reaches_for_the_oracle():
    """The oracle is test infrastructure: nothing under unclerenderer_amd/ (Python or native) may import, include, link or
    load it, and the shipped library must not depend on liburoracle."""
    import re
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|#\s*include\s*[\"<][^\">]*oracle)|liburoracle|oracle/_build|oracle\.oracle", re.M)
    bad = []
    for f in (root / "unclerenderer_amd").rglob("*"):
        if f.is_file() and f.suffix in {".py", ".hip", ".cpp", ".h", ".hpp"} and "_build" not in f.parts:
            if pat.search(f.read_text(errors="replace")):
                bad.append(str(f.relative_to(root)))
    assert not bad, bad
    lib = root / "unclerenderer_amd" / "csrc" / "_build" / "libur_hotpath.so"
    if lib.exists():
        needed = subprocess.run(["readelf", "-d", str(lib)], capture_output=True, text=True).stdout
        assert "oracle" not in needed


def test_hzb_band_helpers_without_a_gpu(urlib):
    """ur_hzb_band_pieces / ur_hzb_band_slices are host-only: the piece rows of the ranks tile the wide launch, their slices tile mips 0-4."""
    from unclerenderer_amd.hotpath import HzbLayout
    for (w, h, n) in [(3840, 2160, 8), (1920, 1080, 3), (1904, 1052, 2), (7680, 4320, 8)]:
        lay = HzbLayout(w, h)
        pieces = [lay.band_pieces(n, r) for r in range(n)]
        assert pieces[0][0] == 0 and pieces[-1][0] + pieces[-1][1] == (h + 31) // 32
        assert all(a[0] + a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        total = sum(c for p in pieces for _, c in lay.band_slices(*p))
        assert total == sum(lay.mips[k].width * lay.mips[k].height for k in range(5))
    a, b = C.c_uint32(0), C.c_uint32(0)
    assert urlib.ur_hzb_band_pieces(2160, 7, 0, C.byref(a), C.byref(b)) != 0  # 7 does not divide 2160
