"""The BASELINE.json configurations at their full sizes, through the C-ABI, against the oracle (on bands) and through
size-independent properties (band split == whole frame, every pixel finite, alpha bookkeeping):

  C2  Sponza 1920x1080 DeferredLighting GGX+IBL (fused with Sky), shipped IBL tables, 2048^2 shadow map
  C4  pica_pica 3840x2160 screen-tiled 8 ways (the camera / light of Assets/Scenes/pica_pica.json)
  C5  1 M instance AABBs culled against the 12-mip HZB that ur_build_hzb makes from a 7680x4320 depth, and the 8K G-buffer lit
plus the kernel instantiations no other test reaches: IRR_LDS = false (the irradiance mip is larger than 2x2, so it is
gathered from memory instead of LDS) and the 12-waves-per-workgroup build (ur_set_option)."""

from pathlib import Path

import numpy as np
import pytest

from tests.util import hdr_mismatch

pytestmark = pytest.mark.gpu
ASSETS = Path(__file__).parent / "golden" / "assets"


def _shipped_tables(hotpath, shadow):
    from unclerenderer_amd import assets
    from unclerenderer_amd.hotpath import to_device
    env, base, mips, bad = assets.load_env_cube_dds(ASSETS / "output_pmrem.dds")
    lut = assets.load_brdf_lut_dds(ASSETS / "PreintegratedGF.dds")
    assert (base, mips, bad) == (256, 9, 0)
    return env, lut, hotpath.make_tables(to_device(shadow), hotpath.stage_env_cube(env, base, mips), base, mips, to_device(lut))


def _bands_equal_whole_and_oracle(hotpath, oracle, fc, g, shadow, env, lut, tables, w, h, n_bands, oracle_bands):
    import torch
    from unclerenderer_amd.hotpath import to_device
    dA, dB, dC, dD = to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth)
    whole = to_device(g.hdr)
    hotpath.deferred_lighting_sky(fc.scene, fc.sky, dA, dB, dC, dD, tables, whole, w, h)
    parts = to_device(g.hdr)
    rows = h // n_bands
    assert rows * n_bands == h
    for r in range(n_bands):
        sl = slice(r * rows, (r + 1) * rows)
        hotpath.deferred_lighting_sky(fc.scene, fc.sky, dA[sl], dB[sl], dC[sl], dD[sl], tables, parts[sl], w, h, r * rows, rows)
    torch.cuda.synchronize()
    assert torch.equal(whole, parts), f"{n_bands} row bands differ from the whole frame"
    out = whole.cpu().numpy().view(np.float16).astype(np.float32)
    assert np.isfinite(out).all(), "after the fused sky pass no pixel may be NaN/inf"
    assert (out[..., 3][g.depth == 0] == 1.0).all() and (out[..., 3][g.depth > 0] == 2.0).all()
    for r0, nr in oracle_bands:
        sl = slice(r0, r0 + nr)
        lit, frag = oracle.deferred_lighting(fc.scene, g.A[sl], g.B[sl], g.C[sl], shadow, env, 256, 9, lut, g.hdr[sl], w, h, r0, nr, want_fragile=True)
        ref = oracle.sky_atmosphere(fc.sky, g.depth[sl], lit, w, h, r0, nr)
        nbad, worst, _ = hdr_mismatch(whole[sl].cpu().numpy().view(np.uint16), ref, exclude=frag)
        assert nbad == 0, f"rows {r0}..{r0 + nr}: {nbad} channel values beyond max(1e-3, 1 ulp fp16) (worst excess {worst})"
        assert frag.mean() < 5e-3


def test_c2_sponza_1080p_lighting(hotpath, oracle):
    from unclerenderer_amd import hostmath, synth
    w, h = 1920, 1080
    fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=2048, env_mip_count=9)
    g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, synth.SEED_BASE + 2)
    shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    env, lut, tables = _shipped_tables(hotpath, shadow)
    # 1080 rows = 8 bands of 135 (not a multiple of the 4-row tile: every band ends in a partial tile row)
    _bands_equal_whole_and_oracle(hotpath, oracle, fc, g, shadow, env, lut, tables, w, h, 8, [(0, 32), (700, 32)])


def test_c2_sponza_1080p_lighting_only_pass(hotpath, oracle):
    """The same configuration as the reference records it: Lighting (additive, every pixel) then Sky, two launches."""
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import to_device
    w, h = 1920, 1080
    fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=2048, env_mip_count=9)
    g = synth.gbuffer_iid(w, h, synth.SEED_BASE + 2)  # SURVEY.md section 8d's generator
    shadow = synth.shadow_map_noise(2048, synth.SEED_BASE + 2)
    env, lut, tables = _shipped_tables(hotpath, shadow)
    dA, dB, dC, dD = to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth)
    two, fused = to_device(g.hdr), to_device(g.hdr)
    hotpath.deferred_lighting(fc.scene, dA, dB, dC, tables, two, w, h)
    hotpath.sky_atmosphere(fc.sky, dD, two, w, h)
    hotpath.deferred_lighting_sky(fc.scene, fc.sky, dA, dB, dC, dD, tables, fused, w, h)
    torch.cuda.synchronize()
    assert torch.equal(two, fused), "fused launch == Lighting then Sky, bit for bit"
    r0, nr = 512, 24
    sl = slice(r0, r0 + nr)
    lit, frag = oracle.deferred_lighting(fc.scene, g.A[sl], g.B[sl], g.C[sl], shadow, env, 256, 9, lut, g.hdr[sl], w, h, r0, nr, want_fragile=True)
    ref = oracle.sky_atmosphere(fc.sky, g.depth[sl], lit, w, h, r0, nr)
    nbad, worst, _ = hdr_mismatch(fused[sl].cpu().numpy().view(np.uint16), ref, exclude=frag)
    assert nbad == 0, (nbad, worst)


def test_c4_pica_pica_4k_eight_bands(hotpath, oracle):
    from unclerenderer_amd import hostmath, synth
    w, h = 3840, 2160
    fc = hostmath.build_frame_constants("pica_pica", w, h, shadow_size=2048, env_mip_count=9)
    assert tuple(np.round(fc.camera_position, 3)) == (-13.482, 20.457, -42.455)
    g = synth.gbuffer_iid(w, h, synth.SEED_BASE + 4)
    shadow = synth.shadow_map_noise(2048, synth.SEED_BASE + 4)
    env, lut, tables = _shipped_tables(hotpath, shadow)
    _bands_equal_whole_and_oracle(hotpath, oracle, fc, g, shadow, env, lut, tables, w, h, 8, [(270 * 3 - 16, 32), (2160 - 24, 24)])


def test_c5_cull_1m_against_the_8k_hzb(hotpath, oracle):
    """1 M instances against the chain ur_build_hzb produces from a 7680x4320 depth (12 mips, three launches): HZB bit-exact
    against the oracle's chain, InstanceCount words / visible list / counters bit-exact against the oracle's cull of it."""
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import HzbLayout, to_device
    w, h, n = 7680, 4320, 1_000_000
    fc = hostmath.build_frame_constants("sponza", w, h)
    depth = synth.gbuffer_iid(w, h, synth.SEED_BASE + 5).depth
    lay = HzbLayout(w, h)
    assert lay.count == 12
    hzb = torch.zeros(lay.total, device="cuda")
    hotpath.build_hzb(to_device(depth), hzb, lay)
    ref_hzb = np.nan_to_num(oracle.build_hzb(depth, lay.as_list(), lay.total))
    got = hzb.cpu().numpy()
    for (off, mw, mh) in lay.as_list():
        assert np.array_equal(got[off:off + mw * mh].view(np.uint32), ref_hzb[off:off + mw * mh].view(np.uint32)), f"mip {mw}x{mh}"
    bounds = synth.instances_random(n, synth.SEED_BASE + 5, center=fc.camera_position, box=400.0)
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, n, True, lay.count, lay.width, lay.height, True)
    args0 = synth.indirect_args_initial(n)
    ref_args, ref_stats, ref_vis, ref_cnt = oracle.cull_indirect_args(consts, bounds, ref_hzb, lay.as_list(), args0)
    d_args, d_stats = to_device(args0), torch.zeros(2, dtype=torch.int32, device="cuda")
    d_vis, d_cnt = torch.full((n,), -1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    hotpath.cull_indirect_args(consts, to_device(bounds), hzb, lay, d_args, d_stats, d_vis, d_cnt)  # the HZB the GPU built
    torch.cuda.synchronize()
    assert np.array_equal(d_args.cpu().numpy().view(np.uint32), ref_args)
    assert int(d_cnt.cpu()[0]) == ref_cnt and np.array_equal(d_vis.cpu().numpy().view(np.uint32)[:ref_cnt], ref_vis)
    assert np.array_equal(d_stats.cpu().numpy().view(np.uint32), ref_stats)
    assert ref_cnt > 1000 and ref_stats[0] > 0 and ref_stats[1] > 0, "both rejection paths and the visible path are exercised"
    assert np.all(np.diff(ref_vis.astype(np.int64)) > 0)
    # the same million with the words' present values taken from the context's record (UR_OPT_CULL_STORE = 4): the launch above left no
    # record (flavour 3), so the first launch reads the words; the second - another camera - runs from the record
    from unclerenderer_amd import lib
    hotpath.set_option(lib.UR_OPT_CULL_STORE, 4)
    try:
        d_bounds = to_device(bounds)
        hotpath.cull_indirect_args(consts, d_bounds, hzb, lay, d_args, None, d_vis, d_cnt)
        fc2 = hostmath.build_frame_constants("pica_pica", w, h)
        consts2 = hostmath.pack_culling_constants(fc2.view, fc2.proj, n, True, lay.count, lay.width, lay.height, False)
        ref_args2, _, ref_vis2, ref_cnt2 = oracle.cull_indirect_args(consts2, bounds, ref_hzb, lay.as_list(), ref_args)
        hotpath.cull_indirect_args(consts2, d_bounds, hzb, lay, d_args, None, d_vis, d_cnt)
        torch.cuda.synchronize()
        assert np.array_equal(d_args.cpu().numpy().view(np.uint32), ref_args2)
        assert int(d_cnt.cpu()[0]) == ref_cnt2 and np.array_equal(d_vis.cpu().numpy().view(np.uint32)[:ref_cnt2], ref_vis2)
        assert (ref_args2[:, 11] != ref_args[:, 11]).sum() > 100, "the second camera changes sides for many instances"
    finally:
        hotpath.set_option(lib.UR_OPT_CULL_STORE, 3)


def test_c5_8k_gbuffer_lighting(hotpath, oracle):
    from unclerenderer_amd import hostmath, synth
    w, h = 7680, 4320
    fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=2048, env_mip_count=9)
    g = synth.gbuffer_iid(w, h, synth.SEED_BASE + 5)
    shadow = synth.shadow_map_noise(2048, synth.SEED_BASE + 5)
    env, lut, tables = _shipped_tables(hotpath, shadow)
    _bands_equal_whole_and_oracle(hotpath, oracle, fc, g, shadow, env, lut, tables, w, h, 8, [(540 * 5 - 8, 16)])


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("shadows", [False, True])
@pytest.mark.parametrize("env_mip_count,irr_n", [(3, 8), (5, 2), (6, 1)])
def test_every_irradiance_table_form(hotpath, oracle, fused, shadows, env_mip_count, irr_n):
    """The irradiance lookup reads mip EnvMapMipCount - 1 of the cube: 1x1 and 2x2 faces live in LDS as per-cell polynomials
    (24 / 54 cells: the larger table reaches up to the workgroup's tile counter, which must not be touched), anything larger
    (here 8x8) is gathered from memory like the prefiltered taps (IRR_LDS = false)."""
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import to_device
    w, h = 320, 180
    fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=256, shadow_strength=1.0 if shadows else 0.0, env_mip_count=env_mip_count)
    assert max(1, 32 >> (env_mip_count - 1)) == irr_n
    env, lut, shadow = synth.env_cube_procedural(32, 6), synth.brdf_lut_procedural(128, 32), synth.shadow_map_noise(256, 41)
    tables = hotpath.make_tables(to_device(shadow) if shadows else None, hotpath.stage_env_cube(env, 32, 6), 32, 6, to_device(lut))
    for g in (synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, 41), synth.gbuffer_iid(w, h, 41)):
        lit, frag = oracle.deferred_lighting(fc.scene, g.A, g.B, g.C, shadow if shadows else None, env, 32, 6, lut, g.hdr, w, h, want_fragile=True)
        ref = oracle.sky_atmosphere(fc.sky, g.depth, lit, w, h) if fused else lit
        hdr = to_device(g.hdr)
        if fused:
            hotpath.deferred_lighting_sky(fc.scene, fc.sky, to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth), tables, hdr, w, h)
        else:
            hotpath.deferred_lighting(fc.scene, to_device(g.A), to_device(g.B), to_device(g.C), tables, hdr, w, h)
        torch.cuda.synchronize()
        nbad, worst, _ = hdr_mismatch(hdr.cpu().numpy().view(np.uint16), ref, exclude=frag)
        assert nbad == 0, (nbad, worst)


def test_twelve_wave_workgroups_give_the_same_bits(hotpath):
    """UR_OPT_LIGHTING_WAVES_PER_WG = 12 (3 waves per SIMD) runs the same arithmetic on a different work split: bit-identical
    output to the default 16-wave build, for the LDS and the gathered irradiance, fused and not, and with partial tile rows.
    Both builds run in this process, on one context (ur_set_option)."""
    import torch
    from tests.test_gpu_parity import _device_tables, _lighting_inputs
    from unclerenderer_amd import lib
    from unclerenderer_amd.hotpath import to_device
    try:
        for name, (w, h, mips) in {"lds": (320, 180, 6.0), "lds2": (320, 180, 5.0), "mem": (320, 180, 3.0), "partial": (272, 33, 6.0)}.items():
            fc, g, shadow, env, lut = _lighting_inputs("sponza", w, h, seed=33, mode="scene")
            fc.scene.EnvMapMipCount = mips
            tables = _device_tables(hotpath, shadow, env, lut)
            for fused in (0, 1):
                got = {}
                for wpb in (16, 12):
                    hotpath.set_option(lib.UR_OPT_LIGHTING_WAVES_PER_WG, wpb)
                    hdr = to_device(g.hdr)
                    if fused:
                        hotpath.deferred_lighting_sky(fc.scene, fc.sky, to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth), tables, hdr, w, h)
                    else:
                        hotpath.deferred_lighting(fc.scene, to_device(g.A), to_device(g.B), to_device(g.C), tables, hdr, w, h)
                    torch.cuda.synchronize()
                    got[wpb] = hdr.cpu().numpy().view(np.uint16)
                assert np.array_equal(got[16], got[12]), (name, fused)
    finally:
        hotpath.set_option(lib.UR_OPT_LIGHTING_WAVES_PER_WG, 16)


@pytest.mark.parametrize("d24", [False, True])
def test_c1_duck_512_whole_frame(hotpath, oracle, d24):
    """BASELINE config 1 at its own size, as ONE frame: Duck's camera, light (intensity 3: HDR values above 2.0, where the
    tolerance is one fp16 ulp - SURVEY.md H6) and its one draw command (Assets/Scenes/Duck.json through csrc/scene.cpp), 512 x 512,
    shipped IBL tables, driven through ur_frame_render twice: frame 1 culls on the frustum alone and builds the HZB, frame 2 culls
    against it (DeferredRenderer.cpp:519,1210). HZB, InstanceCount words, visible list and counters bit-exact; the whole HDR frame
    within max(1e-3, 1 ulp fp16). d24: the depth buffer quantised as the R24_UNORM_X8 SRV returns it (DeferredRenderer.cpp:3087-3096)."""
    import torch
    from unclerenderer_amd import hostmath, lib, scene, synth
    from unclerenderer_amd.hotpath import Frame, HzbLayout, to_device
    w = h = 512
    fc = hostmath.build_frame_constants("duck", w, h, shadow_size=2048, env_mip_count=9)
    assert fc.scene.LightIntensity == 3.0
    g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, synth.SEED_BASE + 1)
    depth = synth.quantize_d24(g.depth) if d24 else g.depth
    if d24:
        assert (depth != g.depth).any() and (depth[g.depth == 0] == 0).all()
    shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    env, lut, tables = _shipped_tables(hotpath, shadow)
    sb = scene.load_scene_bounds(ASSETS / "Scenes" / "Duck.json")
    assert sb.count == 1
    lay = HzbLayout(w, h)
    assert lay.count == 9  # 256^2 ... 1
    dA, dB, dC, dD = to_device(g.A), to_device(g.B), to_device(g.C), to_device(depth)
    d_hzb = torch.zeros(lay.total, device="cuda")
    args0 = synth.indirect_args_initial(1)
    d_args, d_stats = to_device(args0), torch.zeros(2, dtype=torch.int32, device="cuda")
    d_vis, d_cnt = torch.full((1,), -1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, 0, False, 0, 0, 0, True)  # dwords 40-44 are filled by the frame
    ref_hzb = np.nan_to_num(oracle.build_hzb(depth, lay.as_list(), lay.total))
    lit, frag = oracle.deferred_lighting(fc.scene, g.A, g.B, g.C, shadow, env, 256, 9, lut, g.hdr, w, h, want_fragile=True)
    ref_hdr = oracle.sky_atmosphere(fc.sky, depth, lit, w, h)
    assert (ref_hdr.view(np.float16).astype(np.float32)[..., :3] > 2.0).mean() > 0.002, "the fixture must reach the one-ulp branch of the tolerance"
    frame = Frame(hotpath)
    try:
        for k, flags in enumerate((lib.UR_FRAME_DEFAULT | lib.UR_FRAME_FUSE_LIGHTING_SKY,
                                   lib.UR_FRAME_DEFAULT | lib.UR_FRAME_FUSE_LIGHTING_SKY | lib.UR_FRAME_HZB_WITH_LIGHTING)):
            hzb_on = k == 1  # the first frame has no HZB yet
            assert frame.hzb_ready == hzb_on
            d_args.copy_(to_device(args0)); d_stats.zero_()
            hdr = to_device(g.hdr)
            res = Frame.resources(w, h, 0, h, dA, dB, dC, dD, hdr, dD, d_hzb, lay, tables, to_device(sb.bounds), d_args, 1, 0, d_vis, d_cnt, d_stats)
            frame.render(res, consts, fc.scene, fc.sky, flags)
            torch.cuda.synchronize()
            c = hostmath.pack_culling_constants(fc.view, fc.proj, 1, hzb_on, lay.count, lay.width, lay.height, True)
            ref_args, ref_stats, ref_vis, ref_cnt = oracle.cull_indirect_args(c, sb.bounds, ref_hzb if hzb_on else None, lay.as_list(), args0)
            assert np.array_equal(d_args.cpu().numpy().view(np.uint32), ref_args), k
            assert int(d_cnt.cpu()[0]) == ref_cnt and np.array_equal(d_vis.cpu().numpy().view(np.uint32)[:ref_cnt], ref_vis), k
            assert np.array_equal(d_stats.cpu().numpy().view(np.uint32), ref_stats), k
            assert np.array_equal(d_hzb.cpu().numpy().view(np.uint32), ref_hzb.view(np.uint32)), k
            nbad, worst, _ = hdr_mismatch(hdr.cpu().numpy().view(np.uint16), ref_hdr, exclude=frag)
            assert nbad == 0, (k, nbad, worst)
        assert frag.mean() < 5e-3
    finally:
        frame.close()
