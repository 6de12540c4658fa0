"""Inter-workgroup balancing of the streaming lighting launch (UR_OPT_LIGHTING_BALANCE, csrc/lighting.hip struct Balance): the
last part of a launch's tiles is claimed by the workgroups at run time instead of being dealt statically. Which workgroup
shades a tile must not change a bit of it: every schedule is compared byte for byte with the all-static one, over launches in
a row (the claim words must be back at zero after each), with the Build HZB chain riding, and on a band."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(w, h, seed, mode="scene"):
    from tests.test_gpu_parity import _lighting_inputs
    return _lighting_inputs("sponza", w, h, seed=seed, mode=mode)


@pytest.mark.parametrize("w,h,pool,chunk", [(1920, 1080, 3, 2), (1920, 1080, 8, 2), (1280, 720, 6, 2), (2560, 1440, 6, 4), (2560, 1440, 8, 3), (1920, 1083, 6, 3), (3840, 2160, 3, 4)])
def test_balanced_schedule_gives_the_same_bits(hotpath, w, h, pool, chunk):
    import torch
    from tests.test_gpu_parity import _device_tables
    from unclerenderer_amd import lib
    from unclerenderer_amd.hotpath import to_device
    fc, g, shadow, env, lut = _inputs(w, h, 71)
    tables = _device_tables(hotpath, shadow, env, lut)
    dA, dB, dC, dD = to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth)

    def shade(n=1):
        outs = []
        for _ in range(n):
            hdr = to_device(g.hdr)
            hotpath.deferred_lighting_sky(fc.scene, fc.sky, dA, dB, dC, dD, tables, hdr, w, h)
            outs.append(hdr)
        torch.cuda.synchronize()
        return [o.cpu().numpy().view(np.uint16) for o in outs]

    try:
        hotpath.set_option(lib.UR_OPT_LIGHTING_BALANCE, 0)
        ref = shade()[0]
        assert hotpath.lighting_schedule()["pool_chunks"] == 0
        hotpath.set_option(lib.UR_OPT_LIGHTING_BALANCE, 1)
        hotpath.set_option(lib.UR_OPT_BALANCE_POOL_16THS, pool)
        hotpath.set_option(lib.UR_OPT_BALANCE_CHUNK_SHIFT, chunk)
        outs = shade(4)  # four launches in a row: each must find the claim words at zero
        sched = hotpath.lighting_schedule()
        assert sched["pool_chunks"] > 0 and sched["static_tiles"] < sched["tiles"], sched  # the run-time part was really used
        for k, o in enumerate(outs):
            assert np.array_equal(o, ref), (k, sched)
        hotpath.flush()  # would report UR_ETIMEOUT had a wave given up waiting for a claim
    finally:
        hotpath.set_option(lib.UR_OPT_LIGHTING_BALANCE, 1)
        hotpath.set_option(lib.UR_OPT_BALANCE_POOL_16THS, 3)
        hotpath.set_option(lib.UR_OPT_BALANCE_CHUNK_SHIFT, 4)


def test_balanced_schedule_with_the_hzb_chain_riding_and_on_a_band(hotpath, oracle):
    """The same with the Build HZB chain riding the launch (walker waves + tail workgroup beside the claims), on the whole frame and
    on a band of it, for the 12-wave build too; HZB bit-exact against the oracle, HDR byte-equal to the static schedule."""
    import torch
    from tests.test_gpu_parity import _device_tables
    from unclerenderer_amd import lib
    from unclerenderer_amd.hotpath import HzbLayout, to_device
    w, h = 1920, 1080
    fc, g, shadow, env, lut = _inputs(w, h, 72)
    tables = _device_tables(hotpath, shadow, env, lut)
    dA, dB, dC, dD = to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth)
    lay = HzbLayout(w, h)
    ref_hzb = np.nan_to_num(oracle.build_hzb(g.depth, lay.as_list(), lay.total))

    def frame(row0, rows):
        hzb = torch.zeros(lay.total, device="cuda")
        hdr = to_device(g.hdr[row0:row0 + rows])
        hotpath.defer_hzb_tail(2)
        hotpath.build_hzb(dD, hzb, lay)
        hotpath.deferred_lighting_sky(fc.scene, fc.sky, dA[row0:row0 + rows], dB[row0:row0 + rows], dC[row0:row0 + rows], dD[row0:row0 + rows], tables, hdr, w, h, row0, rows)
        hotpath.defer_hzb_tail(0)
        torch.cuda.synchronize()
        return hdr.cpu().numpy().view(np.uint16), hzb.cpu().numpy().view(np.uint32), hotpath.lighting_schedule()

    try:
        for wpb in (16, 12):
            hotpath.set_option(lib.UR_OPT_LIGHTING_WAVES_PER_WG, wpb)
            for row0, rows, pool in ((0, h, 3), (540, 540, 6), (0, 272, 6)):  # (the last band is too short for a run-time part: static)
                hotpath.set_option(lib.UR_OPT_LIGHTING_BALANCE, 0)
                ref, _, _ = frame(row0, rows)
                hotpath.set_option(lib.UR_OPT_LIGHTING_BALANCE, 1)
                hotpath.set_option(lib.UR_OPT_BALANCE_POOL_16THS, pool)
                hotpath.set_option(lib.UR_OPT_BALANCE_CHUNK_SHIFT, 2)
                for k in range(3):
                    out, hzb, sched = frame(row0, rows)
                    assert (sched["pool_chunks"] > 0) == (rows >= 540), (wpb, row0, rows, sched)
                    if wpb == 16:
                        assert sched["hzb_pieces"] > 0, sched  # (the 12-wave build sends the chain out in front)
                    assert np.array_equal(out, ref), (wpb, row0, rows, k, sched)
                    assert np.array_equal(hzb, ref_hzb.view(np.uint32)), (wpb, row0, rows, k)
        hotpath.flush()
    finally:
        hotpath.set_option(lib.UR_OPT_LIGHTING_WAVES_PER_WG, 16)
        hotpath.set_option(lib.UR_OPT_LIGHTING_BALANCE, 1)
        hotpath.set_option(lib.UR_OPT_BALANCE_POOL_16THS, 3)
        hotpath.set_option(lib.UR_OPT_BALANCE_CHUNK_SHIFT, 4)


def test_options_are_validated_and_per_context(urlib):
    import ctypes as C
    import torch
    from unclerenderer_amd import lib
    from unclerenderer_amd.hotpath import HotPath
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    a, b = HotPath(0), HotPath(0)
    try:
        a.set_option(lib.UR_OPT_LIGHTING_WAVES_PER_WG, 12)
        assert a.get_option(lib.UR_OPT_LIGHTING_WAVES_PER_WG) == 12 and b.get_option(lib.UR_OPT_LIGHTING_WAVES_PER_WG) == 16
        L = lib.load()
        assert L.ur_set_option(a.ctx, lib.UR_OPT_LIGHTING_WAVES_PER_WG, 14) == lib.UR_EINVAL
        assert L.ur_set_option(a.ctx, 999, 1) == lib.UR_EINVAL
        assert L.ur_set_option(a.ctx, lib.UR_OPT_BALANCE_CHUNK_SHIFT, 7) == lib.UR_EINVAL
        v = C.c_int(0)
        assert L.ur_get_option(a.ctx, 999, C.byref(v)) == lib.UR_EINVAL
    finally:
        a.close()
        b.close()


def test_balanced_launch_replays_from_a_hip_graph(urlib):
    """A captured launch with run-time tile claims is replayed as often as the host likes: the claim words are put back to zero by
    the launch itself (the workgroup whose last claim comes last), not by the host, so a replay finds them as the capture did."""
    import torch
    from tests.test_gpu_parity import _device_tables
    from unclerenderer_amd import lib
    from unclerenderer_amd.hotpath import HotPath, to_device
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h = 1920, 1080
    fc, g, shadow, env, lut = _inputs(w, h, 73)
    stream = torch.cuda.Stream()
    hp = HotPath(0, stream)
    try:
        with torch.cuda.stream(stream):
            tables = _device_tables(hp, shadow, env, lut)
            dA, dB, dC, dD = to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth)
            hdr0, hdr = to_device(g.hdr), to_device(g.hdr)
            hp.set_option(lib.UR_OPT_LIGHTING_BALANCE, 0)
            hp.deferred_lighting_sky(fc.scene, fc.sky, dA, dB, dC, dD, tables, hdr, w, h)
            stream.synchronize()
            ref = hdr.clone()
            hp.set_option(lib.UR_OPT_LIGHTING_BALANCE, 1)
            hp.set_option(lib.UR_OPT_BALANCE_CHUNK_SHIFT, 2)
            hdr.copy_(hdr0)
            hp.deferred_lighting_sky(fc.scene, fc.sky, dA, dB, dC, dD, tables, hdr, w, h)  # (function attributes are set before the capture)
            stream.synchronize()
            assert hp.lighting_schedule()["pool_chunks"] > 0 and torch.equal(hdr, ref)
            graph = torch.cuda.CUDAGraph()
            hdr.copy_(hdr0)
            with torch.cuda.graph(graph, stream=stream):
                hp.deferred_lighting_sky(fc.scene, fc.sky, dA, dB, dC, dD, tables, hdr, w, h)
            for k in range(4):
                hdr.copy_(hdr0)
                graph.replay()
                stream.synchronize()
                assert torch.equal(hdr, ref), k
            hp.flush()
    finally:
        hp.close()
