"""ur_defer_hzb_tail (include/ur_hotpath.h): the single-workgroup tail of the HZB chain held back and run as an extra
workgroup of the next streaming Lighting launch, or on its own by everything that reads / rewrites the HZB. The HZB and
the HDR band are bit for bit what the separate launches give, whichever way the tail ends up running."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(hotpath, w, h, seed=7):
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import HzbLayout, to_device
    fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=128, env_mip_count=5)
    g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, seed)
    shadow, env, lut = synth.shadow_map_noise(128, seed), synth.env_cube_procedural(16, 5), synth.brdf_lut_procedural(128, 32)
    tables = hotpath.make_tables(to_device(shadow), hotpath.stage_env_cube(env, 16, 5), 16, 5, to_device(lut))
    lay = HzbLayout(w, h)
    dev = dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth))
    return fc, g, tables, lay, dev


def _reference(hotpath, fc, g, tables, lay, dev, w, h):
    import torch
    from unclerenderer_amd.hotpath import to_device
    hzb = torch.full((lay.total,), -1.0, device="cuda")
    hdr = to_device(g.hdr)
    hotpath.build_hzb(dev["D"], hzb, lay)
    hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"], dev["B"], dev["C"], dev["D"], tables, hdr, w, h)
    torch.cuda.synchronize()
    return hzb, hdr


@pytest.mark.parametrize("w,h", [(1024, 512), (512, 256), (3840, 2160)])
def test_tail_rides_with_streaming_lighting(hotpath, w, h):
    import torch
    from unclerenderer_amd.hotpath import to_device
    fc, g, tables, lay, dev = _setup(hotpath, w, h)
    assert lay.count > 5, "the chain must be long enough to have a single-workgroup tail"
    ref_hzb, ref_hdr = _reference(hotpath, fc, g, tables, lay, dev, w, h)
    hzb = torch.full((lay.total,), -1.0, device="cuda")
    hdr = to_device(g.hdr)
    hotpath.defer_hzb_tail(True)
    try:
        hotpath.build_hzb(dev["D"], hzb, lay)
        torch.cuda.synchronize()
        tail_off = lay.as_list()[-1][0]
        assert float(hzb[tail_off]) == -1.0, "the tail is held back: the last mip is still untouched"
        hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"], dev["B"], dev["C"], dev["D"], tables, hdr, w, h)
        torch.cuda.synchronize()
        assert torch.equal(hzb, ref_hzb), "the Lighting launch carried the tail"
        assert torch.equal(hdr, ref_hdr)
        # nothing is pending any more: a flush launches nothing and changes nothing
        hzb2 = hzb.clone()
        hotpath.flush()
        torch.cuda.synchronize()
        assert torch.equal(hzb, hzb2)
    finally:
        hotpath.defer_hzb_tail(False)


def test_tail_flushed_by_flush_cull_rebuild_and_disable(hotpath, oracle):
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import to_device
    w, h = 640, 360
    fc, g, tables, lay, dev = _setup(hotpath, w, h, seed=11)
    ref_hzb, _ = _reference(hotpath, fc, g, tables, lay, dev, w, h)
    last = lay.as_list()[-1][0]

    def held_back():
        hzb = torch.full((lay.total,), -1.0, device="cuda")
        hotpath.defer_hzb_tail(True)
        hotpath.build_hzb(dev["D"], hzb, lay)
        torch.cuda.synchronize()
        assert float(hzb[last]) == -1.0
        return hzb

    try:
        # explicit flush
        hzb = held_back()
        hotpath.flush()
        torch.cuda.synchronize()
        assert torch.equal(hzb, ref_hzb)
        # switching the mode off
        hzb = held_back()
        hotpath.defer_hzb_tail(False)
        torch.cuda.synchronize()
        assert torch.equal(hzb, ref_hzb)
        # a cull reads the chain: the tail runs first, and the result is the one the complete chain gives
        n = 3000
        bounds = synth.instances_random(n, 5, center=fc.camera_position, box=60.0)
        args0 = synth.indirect_args_initial(n)
        c = hostmath.pack_culling_constants(fc.view, fc.proj, n, True, lay.count, lay.width, lay.height, True)
        ref = oracle.cull_indirect_args(c, bounds, np.nan_to_num(oracle.build_hzb(g.depth, lay.as_list(), lay.total)), lay.as_list(), args0)
        hzb = held_back()
        d_args, d_stats = to_device(args0), torch.zeros(2, dtype=torch.int32, device="cuda")
        d_vis, d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        hotpath.cull_indirect_args(c, to_device(bounds), hzb, lay, d_args, d_stats, d_vis, d_cnt)
        torch.cuda.synchronize()
        assert torch.equal(hzb, ref_hzb)
        assert np.array_equal(d_args.cpu().numpy().view(np.uint32), ref[0]) and int(d_cnt.cpu()[0]) == ref[3]
        # a second build on the same context: the first chain's tail runs before the second chain starts
        hzb = held_back()
        hzb_b = torch.full((lay.total,), -1.0, device="cuda")
        hotpath.build_hzb(dev["D"], hzb_b, lay)
        hotpath.flush()
        torch.cuda.synchronize()
        assert torch.equal(hzb, ref_hzb) and torch.equal(hzb_b, ref_hzb)
        # a lighting launch that takes the per-tile kernel (width not a multiple of 16) cannot carry the tail: it sends it out
        # on its own in front, so the chain is complete after ANY Lighting launch on the context
        w2, h2 = 200, 64
        fc2, g2, tables2, lay2, dev2 = _setup(hotpath, w2, h2, seed=3)
        hdr2, ref2 = to_device(g2.hdr), to_device(g2.hdr)
        hotpath.defer_hzb_tail(False)
        hotpath.deferred_lighting_sky(fc2.scene, fc2.sky, dev2["A"], dev2["B"], dev2["C"], dev2["D"], tables2, ref2, w2, h2)
        hzb = held_back()
        hotpath.deferred_lighting_sky(fc2.scene, fc2.sky, dev2["A"], dev2["B"], dev2["C"], dev2["D"], tables2, hdr2, w2, h2)
        torch.cuda.synchronize()
        assert torch.equal(hdr2, ref2)
        assert torch.equal(hzb, ref_hzb)
        hzb_before = hzb.clone()
        hotpath.flush()  # nothing left to launch
        torch.cuda.synchronize()
        assert torch.equal(hzb, hzb_before)
    finally:
        hotpath.defer_hzb_tail(False)


@pytest.mark.parametrize("w,h", [(6001, 3999), (7680, 4320)])
def test_large_chain_two_launches_when_deferred(hotpath, w, h):
    """Mip 4 does not fit the tail's LDS but mip 5 does: with the tail held back the first launch produces five levels and
    the tail reads mip 4 from memory (two launches instead of three). Same bits as the plain build."""
    import torch
    from unclerenderer_amd.hotpath import HzbLayout
    lay = HzbLayout(w, h)
    depth = torch.rand(h * w, device="cuda")
    depth[torch.rand(h * w, device="cuda") < 0.1] = 0.0
    ref = torch.full((lay.total,), -1.0, device="cuda")
    hotpath.build_hzb(depth, ref, lay)
    hzb = torch.full((lay.total,), -1.0, device="cuda")
    hotpath.defer_hzb_tail(True)
    try:
        hotpath.build_hzb(depth, hzb, lay)
        torch.cuda.synchronize()
        off5 = lay.as_list()[5][0]
        assert float(hzb[off5]) == -1.0 and float(hzb[lay.as_list()[4][0]]) != -1.0, "mip 4 written by the first launch, mip 5 held back"
        hotpath.flush()
        torch.cuda.synchronize()
        assert torch.equal(hzb, ref)
    finally:
        hotpath.defer_hzb_tail(False)


def test_frame_flag_gives_the_same_frame(hotpath):
    import torch
    from unclerenderer_amd import hostmath, lib, synth
    from unclerenderer_amd.hotpath import Frame, to_device
    w, h, n = 1024, 576, 25
    fc, g, tables, lay, dev = _setup(hotpath, w, h, seed=19)
    bounds = to_device(synth.instances_random(n, 2, center=fc.camera_position, box=60.0))
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, 0, False, 0, 0, 0, False)
    outs = []
    for extra in (0, lib.UR_FRAME_HZB_TAIL_WITH_LIGHTING, lib.UR_FRAME_HZB_WITH_LIGHTING):
        frame = Frame(hotpath)
        hzb = torch.zeros(lay.total, device="cuda")
        d_args = to_device(synth.indirect_args_initial(n))
        d_vis, d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
        hdr = to_device(g.hdr)
        for _ in range(2):  # the second frame's cull reads the first frame's HZB
            hdr.copy_(to_device(g.hdr))
            res = Frame.resources(w, h, 0, h, dev["A"], dev["B"], dev["C"], dev["D"], hdr, dev["D"], hzb, lay, tables, bounds, d_args, n, 0, d_vis, d_cnt, None)
            frame.render(res, consts, fc.scene, fc.sky, lib.UR_FRAME_DEFAULT | lib.UR_FRAME_FUSE_LIGHTING_SKY | extra)
        torch.cuda.synchronize()
        outs.append((hzb.clone(), hdr.clone(), d_args.clone(), d_vis.clone(), d_cnt.clone()))
        frame.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)


def test_destroy_discards_a_held_back_tail(oracle):
    """ur_destroy must not launch a tail whose HZB buffer the caller may already have freed: the context drops it."""
    import torch
    from unclerenderer_amd.hotpath import HotPath
    hp = HotPath(0)
    w, h = 640, 360
    fc, g, tables, lay, dev = _setup(hp, w, h, seed=23)
    hzb = torch.full((lay.total,), -1.0, device="cuda")
    hp.defer_hzb_tail(True)
    hp.build_hzb(dev["D"], hzb, lay)
    torch.cuda.synchronize()
    last = lay.as_list()[-1][0]
    assert float(hzb[last]) == -1.0
    keep = hzb.clone()
    del hzb
    torch.cuda.empty_cache()       # the buffer the tail points into is gone
    hp.close()                     # ur_destroy: nothing is launched
    torch.cuda.synchronize()
    assert float(keep[last]) == -1.0


def test_cull_rejects_a_chain_that_does_not_halve(hotpath):
    """ur_cull_indirect_args_ex validates EVERY level it may index (the kernel reads hzb + mips[l].offset with pitch
    mips[l].width up to HZBMipCount - 1), not only mips[0] against HZBWidth/HZBHeight."""
    import torch
    from unclerenderer_amd import hostmath, lib, synth
    from unclerenderer_amd.hotpath import HzbLayout, to_device
    w, h, n = 640, 360, 64
    fc = hostmath.build_frame_constants("sponza", w, h)
    lay = HzbLayout(w, h)
    hzb = torch.zeros(lay.total, device="cuda")
    bounds = to_device(synth.instances_random(n, 5, center=fc.camera_position, box=60.0))
    d_args = to_device(synth.indirect_args_initial(n))
    c = hostmath.pack_culling_constants(fc.view, fc.proj, n, True, lay.count, lay.width, lay.height, False)
    hotpath.cull_indirect_args(c, bounds, hzb, lay, d_args)  # the good chain passes
    bad = HzbLayout(w, h)
    bad.mips[3].width += 1
    with pytest.raises(RuntimeError, match="halve"):
        hotpath.cull_indirect_args(c, bounds, hzb, bad, d_args)
    bad = HzbLayout(w, h)
    bad.mips[2].offset = bad.mips[1].offset  # overlapping levels
    with pytest.raises(RuntimeError, match="halve"):
        hotpath.cull_indirect_args(c, bounds, hzb, bad, d_args)
    torch.cuda.synchronize()


@pytest.mark.parametrize("w,h", [(1024, 512), (512, 256), (1920, 1080), (3840, 2160), (7680, 4320), (1000, 300)])
def test_whole_chain_rides_with_streaming_lighting(hotpath, w, h):
    """ur_defer_hzb_tail(ctx, 2): ur_build_hzb launches nothing; the Lighting launch's workgroups walk the wide launch's
    128x32 pieces (one wave each) and its extra workgroup reduces the tail once they have all arrived. HZB and HDR are
    bit for bit what the separate launches give, launch after launch (the arrival counter resets itself)."""
    import torch
    from unclerenderer_amd.hotpath import to_device
    if w * h > 4096 * 2160:
        fc, g, tables, lay, dev = _setup(hotpath, 512, 256)  # small tables; the big frame's G-buffer is synthetic noise
        import numpy as np
        from unclerenderer_amd import hostmath, synth
        from unclerenderer_amd.hotpath import HzbLayout
        fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=128, env_mip_count=5)
        g = synth.gbuffer_iid(w, h, 7)
        lay = HzbLayout(w, h)
        dev = dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth))
    else:
        fc, g, tables, lay, dev = _setup(hotpath, w, h)
    assert lay.count > 5
    ref_hzb, ref_hdr = _reference(hotpath, fc, g, tables, lay, dev, w, h)
    hotpath.defer_hzb_tail(2)
    try:
        for it in range(3):
            hzb = torch.full((lay.total,), -1.0, device="cuda")
            hdr = to_device(g.hdr)
            hotpath.build_hzb(dev["D"], hzb, lay)
            torch.cuda.synchronize()
            assert float(hzb[0]) == -1.0 and float(hzb[lay.as_list()[-1][0]]) == -1.0, "nothing of the chain has been launched yet"
            hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"], dev["B"], dev["C"], dev["D"], tables, hdr, w, h)
            torch.cuda.synchronize()
            assert torch.equal(hzb, ref_hzb), f"launch {it}: the Lighting launch built the whole chain"
            assert torch.equal(hdr, ref_hdr)
        # held back, then flushed without a Lighting launch: the ordinary launches
        hzb = torch.full((lay.total,), -1.0, device="cuda")
        hotpath.build_hzb(dev["D"], hzb, lay)
        hotpath.flush()
        torch.cuda.synchronize()
        assert torch.equal(hzb, ref_hzb)
        # held back, then a per-tile Lighting launch (lighting-only over 3 rows of a 24-pixel-wide frame): flushed in front
        hzb = torch.full((lay.total,), -1.0, device="cuda")
        hotpath.build_hzb(dev["D"], hzb, lay)
        fc2, g2, tables2, lay2, dev2 = _setup(hotpath, 24, 8, seed=3)
        hotpath.deferred_lighting_sky(fc2.scene, fc2.sky, dev2["A"], dev2["B"], dev2["C"], dev2["D"], tables2, to_device(g2.hdr), 24, 8)
        torch.cuda.synchronize()
        assert torch.equal(hzb, ref_hzb)
        # switching down to mode 1 flushes a held-back wide launch; the tail alone then rides as before
        hzb = torch.full((lay.total,), -1.0, device="cuda")
        hotpath.build_hzb(dev["D"], hzb, lay)
        hotpath.defer_hzb_tail(1)
        torch.cuda.synchronize()
        assert torch.equal(hzb, ref_hzb)
    finally:
        hotpath.defer_hzb_tail(0)


def test_debug_timeline_orders_the_launches(hotpath):
    """ur_debug_timeline: every cull and streaming Lighting launch folds the GPU's constant clock into the next {first entry,
    last exit} pair of the caller's ring, in launch order; with the chain riding the Lighting pair spans the HZB work too."""
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import to_device
    w, h, n = 1024, 512, 25
    fc, g, tables, lay, dev = _setup(hotpath, w, h, seed=5)
    bounds = to_device(synth.instances_random(n, 2, center=fc.camera_position, box=60.0))
    d_args = to_device(synth.indirect_args_initial(n))
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, n, False, 0, 0, 0, False)
    hzb = torch.zeros(lay.total, device="cuda")
    pairs = torch.zeros((8, 2), dtype=torch.int64, device="cuda")
    pairs[:, 0] = -1
    torch.cuda.synchronize()
    hotpath.debug_timeline(pairs)
    hotpath.defer_hzb_tail(2)
    try:
        for _ in range(3):
            hotpath.cull_indirect_args(consts, bounds, None, None, d_args)
            hotpath.build_hzb(dev["D"], hzb, lay)   # held back: no launch, no pair
            hdr = to_device(g.hdr)
            hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"], dev["B"], dev["C"], dev["D"], tables, hdr, w, h)
        torch.cuda.synchronize()
    finally:
        hotpath.defer_hzb_tail(0)
        hotpath.debug_timeline(None)
    p = pairs.cpu().numpy().view("uint64")
    used = p[:6]
    assert (p[6:, 1] == 0).all() and (used[:, 1] != 0).all(), "six launches, six pairs"
    assert (used[:, 1] > used[:, 0]).all(), "exit after entry"
    assert (used[1:, 0] >= used[:-1, 1]).all(), "launches of one stream do not overlap: each entry is behind the previous exit"
    spans_us = (used[:, 1] - used[:, 0]).astype("float64") * 1e-2
    assert spans_us[0::2].max() < spans_us[1::2].min(), "cull launches are shorter than Lighting launches"
    # switched off: further launches stamp nothing
    before = pairs.clone()
    hotpath.cull_indirect_args(consts, bounds, None, None, d_args)
    torch.cuda.synchronize()
    assert torch.equal(pairs, before)


@pytest.mark.parametrize("rows,row0", [(270, 810), (540, 0)])
def test_whole_chain_rides_with_a_short_band(hotpath, rows, row0):
    """A rank that shades a 1/8 or 1/4 band of the 4K frame still builds the WHOLE frame's HZB: the chain would outlast the
    shading if one wave per workgroup walked its pieces, so every wave does (the RIDE_ALL instantiation). Same bits."""
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import HzbLayout, to_device
    w, h = 3840, 2160
    fc0, g0, tables, lay0, dev0 = _setup(hotpath, 512, 256)
    fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=128, env_mip_count=5)
    g = synth.gbuffer_iid(w, h, 9)
    lay = HzbLayout(w, h)
    sl = slice(row0, row0 + rows)
    A, B, Cc, D_full = to_device(g.A[sl]), to_device(g.B[sl]), to_device(g.C[sl]), to_device(g.depth)
    D_band = D_full[sl]
    ref_hzb = torch.full((lay.total,), -1.0, device="cuda")
    ref_hdr = to_device(g.hdr[sl])
    hotpath.build_hzb(D_full, ref_hzb, lay)
    hotpath.deferred_lighting_sky(fc.scene, fc.sky, A, B, Cc, D_band, tables, ref_hdr, w, h, row0, rows)
    torch.cuda.synchronize()
    hotpath.defer_hzb_tail(2)
    try:
        for it in range(2):
            hzb = torch.full((lay.total,), -1.0, device="cuda")
            hdr = to_device(g.hdr[sl])
            hotpath.build_hzb(D_full, hzb, lay)
            torch.cuda.synchronize()
            assert float(hzb[0]) == -1.0
            hotpath.deferred_lighting_sky(fc.scene, fc.sky, A, B, Cc, D_band, tables, hdr, w, h, row0, rows)
            torch.cuda.synchronize()
            assert torch.equal(hzb, ref_hzb) and torch.equal(hdr, ref_hdr), it
    finally:
        hotpath.defer_hzb_tail(0)


def test_riding_tail_timeout_is_reported_once_by_the_next_entry_point(hotpath):
    """The riding tail workgroup's wait for its producers is bounded; giving up sets a host-visible flag (UR_ETIMEOUT,
    include/ur_hotpath.h). The host side of that path, driven by the debug entry point that sets the flag as the kernel
    does: each entry point that depends on the HZB reports it exactly once, does nothing else in that call, and the context
    is usable afterwards (the same calls then succeed and give the reference bits)."""
    import torch
    from unclerenderer_amd import hostmath, lib, synth
    from unclerenderer_amd.hotpath import to_device
    w, h = 1024, 512
    fc, g, tables, lay, dev = _setup(hotpath, w, h)
    ref_hzb, ref_hdr = _reference(hotpath, fc, g, tables, lay, dev, w, h)
    L = lib.load()
    # ur_flush
    assert L.ur_debug_set_hzb_timeout(hotpath.ctx) == lib.UR_OK
    assert L.ur_flush(hotpath.ctx) == lib.UR_ETIMEOUT
    assert b"gave up waiting" in L.ur_last_error()
    assert L.ur_flush(hotpath.ctx) == lib.UR_OK, "reported once"
    # ur_build_hzb: refuses (nothing is launched), then builds
    hzb = torch.full((lay.total,), -1.0, device="cuda")
    assert L.ur_debug_set_hzb_timeout(hotpath.ctx) == lib.UR_OK
    with pytest.raises(lib.UrError) as e:
        hotpath.build_hzb(dev["D"], hzb, lay)
    assert e.value.code == lib.UR_ETIMEOUT
    torch.cuda.synchronize()
    assert float(hzb.max()) == -1.0, "the refused call launched nothing"
    hotpath.build_hzb(dev["D"], hzb, lay)
    torch.cuda.synchronize()
    assert torch.equal(hzb, ref_hzb)
    # ur_cull_indirect_args: a cull against a stale HZB is refused, not run
    n = 500
    bounds = to_device(synth.instances_random(n, 5, center=fc.camera_position, box=60.0))
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, n, True, lay.count, lay.width, lay.height, False)
    args0 = synth.indirect_args_initial(n)
    d_args = to_device(args0)
    assert L.ur_debug_set_hzb_timeout(hotpath.ctx) == lib.UR_OK
    with pytest.raises(lib.UrError) as e:
        hotpath.cull_indirect_args(consts, bounds, hzb, lay, d_args)
    assert e.value.code == lib.UR_ETIMEOUT
    torch.cuda.synchronize()
    assert np.array_equal(d_args.cpu().numpy().view(np.uint32).reshape(args0.shape), args0), "the refused cull wrote nothing"
    hotpath.cull_indirect_args(consts, bounds, hzb, lay, d_args)
    torch.cuda.synchronize()
    # the whole chain riding a Lighting launch still works on this context (the arrival counter was reset by the report)
    hzb2 = torch.full((lay.total,), -1.0, device="cuda")
    hdr = to_device(g.hdr)
    hotpath.defer_hzb_tail(2)
    try:
        hotpath.build_hzb(dev["D"], hzb2, lay)
        hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"], dev["B"], dev["C"], dev["D"], tables, hdr, w, h)
        torch.cuda.synchronize()
    finally:
        hotpath.defer_hzb_tail(False)
    assert torch.equal(hzb2, ref_hzb) and torch.equal(hdr, ref_hdr)
    assert L.ur_flush(hotpath.ctx) == lib.UR_OK, "a launch that completed its wait leaves no flag behind"


def test_frame_render_reports_a_riding_tail_timeout(hotpath):
    import torch
    from unclerenderer_amd import hostmath, lib
    from unclerenderer_amd.hotpath import Frame, to_device
    w, h = 512, 256
    fc, g, tables, lay, dev = _setup(hotpath, w, h)
    hzb = torch.zeros(lay.total, device="cuda")
    hdr = to_device(g.hdr)
    frame = Frame(hotpath)
    res = Frame.resources(w, h, 0, h, dev["A"], dev["B"], dev["C"], dev["D"], hdr, dev["D"], hzb, lay, tables)
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, 0, True, lay.count, lay.width, lay.height, False)
    flags = lib.UR_FRAME_DEFAULT | lib.UR_FRAME_FUSE_LIGHTING_SKY | lib.UR_FRAME_HZB_WITH_LIGHTING
    frame.render(res, consts, fc.scene, fc.sky, flags)
    torch.cuda.synchronize()
    L = lib.load()
    assert L.ur_debug_set_hzb_timeout(hotpath.ctx) == lib.UR_OK
    with pytest.raises(lib.UrError) as e:
        frame.render(res, consts, fc.scene, fc.sky, flags)
    assert e.value.code == lib.UR_ETIMEOUT
    frame.render(res, consts, fc.scene, fc.sky, flags)  # reported once
    torch.cuda.synchronize()
    frame.close()


def test_the_kernel_side_of_a_riding_tail_timeout(hotpath):
    """The REAL time-out path (round 3 tested only the host's half, through ur_debug_set_hzb_timeout): with
    UR_OPT_DEBUG_HZB_RIDE_STALL the riding tail workgroup expects one arrival more than it has producers and gives up after ~1 ms.
    It must raise the host-visible flag (UR_ETIMEOUT at the next entry point), leave the arrival word alone and mark the words
    stale; riding launches queued BEFORE the host has noticed (their tails find a leftover count) report as well instead of passing
    silently with a wrong HZB; and after the host's reset the same context rides correctly again."""
    import torch
    from unclerenderer_amd import lib
    from unclerenderer_amd.hotpath import to_device
    w, h = 1024, 512
    fc, g, tables, lay, dev = _setup(hotpath, w, h)
    ref_hzb, ref_hdr = _reference(hotpath, fc, g, tables, lay, dev, w, h)
    L = lib.load()

    def ride(n=1):
        outs = []
        hotpath.defer_hzb_tail(2)
        try:
            for _ in range(n):
                hzb = torch.full((lay.total,), -1.0, device="cuda")
                hdr = to_device(g.hdr)
                # ur_build_hzb reports a time-out it finds: take the entry points apart from the checks with the raw calls
                rc = L.ur_build_hzb(hotpath.ctx, dev["D"].data_ptr(), w, h, hzb.data_ptr(), lay.mips, lay.count)
                outs.append((rc, hzb, hdr))
                if rc == lib.UR_OK:
                    hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"], dev["B"], dev["C"], dev["D"], tables, hdr, w, h)
        finally:
            L.ur_defer_hzb_tail(hotpath.ctx, 0)  # (its flush may itself report the time-out: the raw call does not raise)
        torch.cuda.synchronize()
        return outs

    hotpath.set_option(lib.UR_OPT_DEBUG_HZB_RIDE_STALL, 1)
    try:
        (rc, hzb, hdr), = ride()
        assert rc == lib.UR_OK
        assert torch.equal(hdr, ref_hdr), "the shading itself is complete"
        assert L.ur_flush(hotpath.ctx) == lib.UR_ETIMEOUT and b"gave up waiting" in L.ur_last_error()
        assert L.ur_flush(hotpath.ctx) == lib.UR_OK
    finally:
        hotpath.set_option(lib.UR_OPT_DEBUG_HZB_RIDE_STALL, 0)
    # stall again, then queue a NORMAL riding launch behind it before the host looks: its tail finds the words stale and reports
    hotpath.set_option(lib.UR_OPT_DEBUG_HZB_RIDE_STALL, 1)
    try:
        hzb_a, hdr_a = torch.full((lay.total,), -1.0, device="cuda"), to_device(g.hdr)
        hotpath.defer_hzb_tail(2)
        assert L.ur_build_hzb(hotpath.ctx, dev["D"].data_ptr(), w, h, hzb_a.data_ptr(), lay.mips, lay.count) == lib.UR_OK
        hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"], dev["B"], dev["C"], dev["D"], tables, hdr_a, w, h)
        hotpath.set_option(lib.UR_OPT_DEBUG_HZB_RIDE_STALL, 0)
        hzb_b, hdr_b = torch.full((lay.total,), -1.0, device="cuda"), to_device(g.hdr)
        rc_b = L.ur_build_hzb(hotpath.ctx, dev["D"].data_ptr(), w, h, hzb_b.data_ptr(), lay.mips, lay.count)
        reported = rc_b == lib.UR_ETIMEOUT
        if rc_b == lib.UR_OK:  # queued before the first launch's flag was seen (the usual case: the host runs ahead of the GPU)
            try:
                hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"], dev["B"], dev["C"], dev["D"], tables, hdr_b, w, h)
            except lib.UrError as e:  # (the Lighting entry points look at the flag too: the first launch may have raised it by now)
                assert e.code == lib.UR_ETIMEOUT
                reported = True
            torch.cuda.synchronize()
            if not reported:
                assert L.ur_flush(hotpath.ctx) == lib.UR_ETIMEOUT, "the launch behind a timed-out one must not pass silently"
        else:
            assert rc_b == lib.UR_ETIMEOUT
        torch.cuda.synchronize()
    finally:
        hotpath.set_option(lib.UR_OPT_DEBUG_HZB_RIDE_STALL, 0)
        L.ur_defer_hzb_tail(hotpath.ctx, 0)
    while L.ur_flush(hotpath.ctx) == lib.UR_ETIMEOUT:  # (each report resets the words behind everything queued so far)
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    # clean again: three riding frames in a row give the reference bits
    for rc, hzb, hdr in ride(3):
        assert rc == lib.UR_OK and torch.equal(hzb, ref_hzb) and torch.equal(hdr, ref_hdr)
    assert L.ur_flush(hotpath.ctx) == lib.UR_OK
