"""Band-sharded Build HZB (ur_build_hzb_band / ur_build_hzb_tail, SURVEY.md section 8e row 3's alternative): every rank's band
launch writes exactly its slices of mips 0..4 with the whole-frame launch's bits, the ranks' slices tile those levels, and the
tail behind the exchange completes the chain - byte for byte ur_build_hzb's HZB (Shaders/BuildHZB.hlsl:34-126 through the
reference's dispatch grouping, DeferredRenderer.cpp:1046-1207), as separate launches and riding a Lighting launch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,worlds", [(3840, 2160, (2, 8)), (1920, 1080, (3, 8)), (1918, 1082, (2,)), (6001, 3999, (3,)), (7680, 4320, (8,)), (640, 360, (5,))])
def test_band_launches_tile_the_whole_chain(hotpath, oracle, w, h, worlds):
    import torch
    from unclerenderer_amd import synth
    from unclerenderer_amd.hotpath import HzbLayout, to_device
    depth = synth.hash_unit(29, *synth._grid(w, 0, h), 0).astype(np.float32)
    depth[(synth.hash_unit(30, *synth._grid(w, 0, h), 0) < 0.1)] = 0.0  # some cleared pixels
    lay = HzbLayout(w, h)
    dD = to_device(depth)
    ref = torch.full((lay.total,), -1.0, device="cuda")
    hotpath.build_hzb(dD, ref, lay)
    torch.cuda.synchronize()
    valid = torch.zeros(lay.total, dtype=torch.bool, device="cuda")  # (mips start on 256-byte boundaries: the gaps hold nothing)
    for off, mw, mh in lay.as_list():
        valid[off:off + mw * mh] = True
    if w * h <= 1920 * 1080:
        want = np.nan_to_num(oracle.build_hzb(depth, lay.as_list(), lay.total)).view(np.uint32)
        assert np.array_equal(ref.cpu().numpy().view(np.uint32)[valid.cpu().numpy()], want[valid.cpu().numpy()])
    for world in worlds:
        whole = torch.full((lay.total,), -1.0, device="cuda")
        for r in range(world):
            p0, pn = lay.band_pieces(world, r)
            mine = torch.full((lay.total,), -1.0, device="cuda")
            hotpath.build_hzb_band(dD, mine, lay, p0, pn)
            torch.cuda.synchronize()
            written = mine != -1.0
            expect = torch.zeros(lay.total, dtype=torch.bool, device="cuda")
            for off, cnt in lay.band_slices(p0, pn):
                expect[off:off + cnt] = True
                assert torch.equal(mine[off:off + cnt], ref[off:off + cnt]), (world, r)
                whole[off:off + cnt] = mine[off:off + cnt]  # (the exchange)
            assert torch.equal(written, expect), f"rank {r} of {world} wrote outside its slices"
        hotpath.build_hzb_tail(whole, lay)
        torch.cuda.synchronize()
        assert torch.equal(whole[valid], ref[valid]), world


def test_band_pieces_ride_the_lighting_launch(hotpath):
    """With ur_defer_hzb_tail(ctx, 2) a band's pieces ride the rank's streaming Lighting launch (no tail, no arrival counter: the
    tail waits for the exchange): same slices, same HDR; the riding whole chain still works on the same context afterwards."""
    import torch
    from tests.test_gpu_hzb_tail import _reference, _setup
    from unclerenderer_amd.hotpath import to_device
    w, h = 3840, 2160
    fc, g, tables, lay, dev = _setup(hotpath, w, h)
    ref_hzb, ref_hdr = _reference(hotpath, fc, g, tables, lay, dev, w, h)
    world = 8
    for rank in (0, 3, 7):
        rows = h // world
        sl = slice(rank * rows, (rank + 1) * rows)
        p0, pn = lay.band_pieces(world, rank)
        hzb = torch.full((lay.total,), -1.0, device="cuda")
        hdr = to_device(g.hdr[sl])
        hotpath.defer_hzb_tail(2)
        try:
            hotpath.build_hzb_band(dev["D"], hzb, lay, p0, pn)
            torch.cuda.synchronize()
            assert float(hzb.max()) == -1.0, "held back: nothing launched yet"
            hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"][sl], dev["B"][sl], dev["C"][sl], dev["D"][sl], tables, hdr, w, h, rank * rows, rows)
            torch.cuda.synchronize()
            assert hotpath.lighting_schedule()["hzb_pieces"] == pn * ((lay.width + 63) // 64)
        finally:
            hotpath.defer_hzb_tail(0)
        assert torch.equal(hdr, ref_hdr[sl]), rank
        expect = torch.zeros(lay.total, dtype=torch.bool, device="cuda")
        for off, cnt in lay.band_slices(p0, pn):
            expect[off:off + cnt] = True
            assert torch.equal(hzb[off:off + cnt], ref_hzb[off:off + cnt]), rank
        assert torch.equal(hzb != -1.0, expect), rank
    # the whole chain riding (arrival counter + tail workgroup) is untouched by the band launches before it
    hzb2, hdr2 = torch.full((lay.total,), -1.0, device="cuda"), to_device(g.hdr)
    hotpath.defer_hzb_tail(2)
    try:
        hotpath.build_hzb(dev["D"], hzb2, lay)
        hotpath.deferred_lighting_sky(fc.scene, fc.sky, dev["A"], dev["B"], dev["C"], dev["D"], tables, hdr2, w, h)
        torch.cuda.synchronize()
    finally:
        hotpath.defer_hzb_tail(0)
    assert torch.equal(hzb2, ref_hzb) and torch.equal(hdr2, ref_hdr)
    hotpath.flush()


def test_band_build_rejects_what_it_cannot_do(hotpath):
    import torch
    from unclerenderer_amd import lib
    from unclerenderer_amd.hotpath import HzbLayout
    lay = HzbLayout(64, 64)  # 6 mips: not a five-level launch plus a tail worth sharding? (mip 5 is 1x1: it is) -> allowed
    d, hz = torch.zeros(64 * 64, device="cuda"), torch.zeros(lay.total, device="cuda")
    hotpath.build_hzb_band(d, hz, lay, 0, 2)
    small = HzbLayout(16, 16)  # 4 mips: no tail to split off
    with pytest.raises(lib.UrError) as e:
        hotpath.build_hzb_band(torch.zeros(256, device="cuda"), torch.zeros(small.total, device="cuda"), small, 0, 1)
    assert e.value.code == lib.UR_EUNSUPPORTED
    with pytest.raises(lib.UrError) as e:
        hotpath.build_hzb_band(d, hz, lay, 1, 2)  # piece rows [1, 3) of 2
    assert e.value.code == lib.UR_EINVAL
    L = lib.load()
    import ctypes as C
    a, b = C.c_uint32(0), C.c_uint32(0)
    assert L.ur_hzb_band_pieces(2160, 7, 0, C.byref(a), C.byref(b)) == lib.UR_EINVAL  # 7 does not divide 2160
    assert L.ur_hzb_band_pieces(2160, 8, 8, C.byref(a), C.byref(b)) == lib.UR_EINVAL
