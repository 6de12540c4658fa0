"""The oracle against the committed fixtures (tests/golden/hotpath_golden.npz, minted by tests/golden/make_golden.py).
Integer/min paths must reproduce bit for bit; the HDR images within one fp16 ulp (libm pow/exp may differ by an ulp
between hosts). The fixtures pin the oracle over time; the reference itself has no vectors for this path."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from tests.util import hdr_mismatch
from unclerenderer_amd import lib, synth

GOLD = Path(__file__).parent / "golden" / "hotpath_golden.npz"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD, allow_pickle=False)


@pytest.mark.parametrize("w,h", [(3, 5), (17, 9), (64, 36), (129, 67)])
def test_hzb_golden(oracle, gold, w, h):
    mips, total = oracle.hzb_layout(w, h)
    out = oracle.build_hzb(gold[f"hzb_{w}x{h}_depth"], mips, total)
    assert np.array_equal(out.view(np.uint32), gold[f"hzb_{w}x{h}_out"].view(np.uint32))


def test_cull_golden(oracle, gold):
    mips = [tuple(int(v) for v in m) for m in gold["cull_mips"]]
    n = gold["cull_bounds"].shape[0]
    args, stats, vis, cnt = oracle.cull_indirect_args(gold["cull_consts"], gold["cull_bounds"], gold["cull_hzb"], mips, synth.indirect_args_initial(n))
    assert np.array_equal(args[:, 11], gold["cull_words"])
    assert np.array_equal(stats, gold["cull_stats"]) and np.array_equal(vis, gold["cull_visible"])
    assert 0 < cnt < n and stats[1] > 0  # the fixture exercises frustum rejects, occlusion rejects and survivors


@pytest.mark.parametrize("mode", ["iid", "scene"])
def test_lighting_golden(oracle, gold, mode):
    scene = lib.SceneConstants.from_buffer_copy(gold["light_scene_bytes"].tobytes())
    sky = lib.SkyConstants.from_buffer_copy(gold["light_sky_bytes"].tobytes())
    w, h = 48, 27
    g = lambda k: gold[f"light_{mode}_{k}"]
    lit, frag = oracle.deferred_lighting(scene, g("A"), g("B"), g("C"), gold["light_shadow"], gold["light_env"], 8, 4, gold["light_lut"], g("hdr"), w, h,
                                         want_fragile=True)
    nbad, worst, _ = hdr_mismatch(lit, g("lit"), exclude=frag | g("fragile"))
    assert nbad == 0, (nbad, worst)
    final = oracle.sky_atmosphere(sky, g("depth"), lit, w, h)
    nbad, worst, _ = hdr_mismatch(final, g("final"), exclude=frag | g("fragile"))
    assert nbad == 0, (nbad, worst)
    assert (final.view(np.float16)[..., 3][g("depth") == 0] == 1).all()
