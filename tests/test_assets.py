"""DDS container + BC6H decode (next row). Host-only code in libur_hotpath.so.

PARITY UNPINNED: the reference ships no decoder and no decoded image (D3D12 hardware decodes BC6H). Pinned by
hand-built blocks whose texels follow from the published unquantisation arithmetic, and by properties of the shipped
prefiltered cube that a wrong bit layout or table would break (every mode valid, energy preserved along the mip
chain, no discontinuity at 4x4 block borders)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from unclerenderer_amd import assets, lib

ASSETS = Path(__file__).parent / "golden" / "assets"


def _block(bits: dict) -> bytes:
    v = 0
    for (pos, width), val in bits.items():
        assert 0 <= val < (1 << width)
        v |= val << pos
    return v.to_bytes(16, "little")


def _decode(block: bytes, signed=True):
    out = np.zeros((16, 4), np.uint16)
    ok = lib.load().ur_bc6h_decode_block(block, int(signed), out.ctypes.data_as(C.c_void_p))
    return ok, out


def test_bc6h_hand_built_blocks(urlib):
    # mode 11 (00011): one region, 10-bit endpoints stored directly; rw = rx = 255, all indices 0
    ok, t = _decode(_block({(0, 5): 3, (5, 10): 255, (35, 10): 255}))
    # unquantize(255, 10 bits, signed) = ((255 << 15) + 0x4000) >> 9 = 16352; finish = (16352 * 31) >> 5 = 15841 = 0x3DE1
    assert ok and (t[:, 0] == 0x3DE1).all() and (t[:, 1] == 0).all() and (t[:, 2] == 0).all() and (t[:, 3] == 0x3C00).all()
    # endpoints 0 and +511 (-> 0x7FFF): index 15 gives the largest half 65504 (0x7BFF), index 8 (weight 34) gives 0x41DF
    ok, t = _decode(_block({(0, 5): 3, (35, 10): 511, (68, 4): 15, (72, 4): 8}))
    assert ok and t[0, 0] == 0 and t[1, 0] == 0x7BFF and t[2, 0] == 0x41DF and (t[3:, 0] == 0).all()
    # signed: rw = -1 -> unquantize(-1) = -96 -> finish -(96 * 31 >> 5) = -93 -> sign-magnitude half 0x805D
    ok, t = _decode(_block({(0, 5): 3, (5, 10): 0x3FF, (35, 10): 0x3FF}))
    assert ok and (t[:, 0] == 0x805D).all()
    # unsigned format: same bits are +1023 -> 0xFFFF -> (65535 * 31) >> 6 = 31743 = 0x7BFF
    ok, t = _decode(_block({(0, 5): 3, (5, 10): 0x3FF, (35, 10): 0x3FF}), signed=False)
    assert ok and (t[:, 0] == 0x7BFF).all()
    # mode 1 (00): two regions, rw = 100, deltas 0 -> every endpoint 100 -> uniform block whatever the partition/indices
    ok, t = _decode(_block({(0, 2): 0, (5, 10): 100, (77, 5): 13, (82, 46): (1 << 46) - 1}))
    want = ((((100 << 15) + 0x4000) >> 9) * 31) >> 5
    assert ok and (t[:, 0] == want).all() and (t[:, 1] == 0).all()
    # mode 12 (00111) transformed: rw = 200 (11-bit), rx = -3 (9-bit delta) -> e1 = 197
    ok, t = _decode(_block({(0, 5): 7, (5, 10): 200, (35, 9): 0x1FD, (68, 4): 15}))
    e = lambda x: (((x << 15) + 0x4000) >> 10) * 31 >> 5
    assert ok and t[0, 0] == e(200) and t[1, 0] == e(197)
    # reserved mode (10011) decodes to zero and reports it
    ok, t = _decode(_block({(0, 5): 0x13}))
    assert not ok and (t[:, :3] == 0).all()


def test_shipped_cube_chain_is_consistent(urlib):
    cube, base, mips, bad = assets.load_env_cube_dds(ASSETS / "output_pmrem.dds")
    assert (base, mips, bad) == (256, 9, 0)
    assert cube.shape[0] == 6 * sum(max(1, 256 >> m) ** 2 for m in range(9)) == 524286
    f = cube.view(np.float16).astype(np.float32)
    assert np.isfinite(f).all() and (f[:, 3] == 1).all() and f[:, :3].min() > 0 and f[:, :3].max() < 64
    stride = cube.shape[0] // 6
    off, means = 0, []
    for m in range(9):
        n = max(1, 256 >> m)
        a = np.stack([f[k * stride + off: k * stride + off + n * n].reshape(n, n, 4)[..., :3] for k in range(6)])
        means.append(a.mean(axis=(0, 1, 2)))
        if n >= 16:  # a wrong partition table / index layout shows up as steps at block borders
            dx = np.abs(np.diff(a, axis=2))
            edge = (np.arange(n - 1) % 4) == 3
            assert dx[:, :, edge].mean() < 2.0 * dx[:, :, ~edge].mean()
        off += n * n
    means = np.array(means)
    assert np.allclose(means, means[0], rtol=0.02)  # a prefiltered chain preserves the mean radiance


def test_shipped_lut(urlib):
    lut = assets.load_brdf_lut_dds(ASSETS / "PreintegratedGF.dds")
    assert lut.shape == (32, 128, 2)
    a = lut.astype(np.float64) / 65535
    assert tuple(lut[0, 127]) == (65535, 0)            # smooth surface, normal incidence: F0 * 1 + 0
    assert a[0, 0, 1] > 0.9 and a[0, 0, 0] < 0.05      # grazing: the bias term takes over
    assert (np.diff(a[0, :, 0]) > -2e-3).all()         # smooth row: the scale grows with NdotV
    assert (a.sum(-1) < 1.02).all()


def test_dds_rejects_garbage(urlib):
    info = lib.DdsInfo()
    assert urlib.ur_dds_parse(b"\0" * 200, 200, C.byref(info)) == -2
    assert urlib.ur_dds_parse(b"DDS ", 4, C.byref(info)) == -1
    data = bytearray((ASSETS / "output_pmrem.dds").read_bytes())
    assert urlib.ur_dds_parse(bytes(data[:4000]), 4000, C.byref(info)) == -2  # truncated
    with pytest.raises(ValueError):
        assets.load_env_cube_dds(ASSETS / "PreintegratedGF.dds")
