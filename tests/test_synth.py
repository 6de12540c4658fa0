"""Synthetic-input generator: deterministic, band-sliceable (any rank can generate any rows), well-formed."""
import numpy as np

from unclerenderer_amd import hostmath, synth


def test_hash_is_counter_based():
    x, y = np.arange(7, dtype=np.uint32), np.arange(7, dtype=np.uint32) * 3
    a = synth.hash_u32(5, x, y, 2)
    assert np.array_equal(a, synth.hash_u32(5, x, y, 2))
    assert not np.array_equal(a, synth.hash_u32(6, x, y, 2)) and not np.array_equal(a, synth.hash_u32(5, x, y, 3))
    u = synth.hash_unit(1, np.arange(100000, dtype=np.uint32), np.zeros(100000, np.uint32), 0)
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01
    assert int(synth.pcg_hash(np.array([0], np.uint32))[0]) == 129708002  # pcg_hash(0), fixed point of the definition


def test_bands_equal_slices_of_the_frame(urlib):
    w, h = 96, 54
    fc = hostmath.build_frame_constants("sponza", w, h)
    for gen in (lambda r0, rows: synth.gbuffer_iid(w, h, 4, r0, rows),
                lambda r0, rows: synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, 4, r0, rows, chunk_rows=16)):
        whole = gen(0, h)
        band = gen(18, 18)
        for k in ("A", "B", "C", "hdr", "depth"):
            assert np.array_equal(getattr(band, k), getattr(whole, k)[18:36]), k


def test_gbuffer_is_well_formed(urlib):
    w, h = 448, 256  # 7 x 4 background blobs of 64 x 64 for the iid generator
    fc = hostmath.build_frame_constants("sponza", w, h)
    for g in (synth.gbuffer_iid(w, h, 8), synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, 8)):
        a = g.A.view(np.float16).astype(np.float32)
        geo = g.depth > 0
        assert geo.any() and (~geo).any()
        n = a[geo][:, :3]
        assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=3e-3)
        assert (n[:, 2] < 0).all()                       # camera-facing in view space
        assert (a[geo][:, 3] < 0).all()                  # A.w = -viewZ
        np.testing.assert_allclose(g.depth[geo], 0.1 / -a[geo][:, 3], rtol=2e-3)
        assert (a[~geo] == [0, 0, 0, 1]).all()           # cleared texels (DeferredRenderer.cpp:757-761)
        b = g.B.view(np.float16).astype(np.float32)
        assert (b[geo][:, 2] >= 0.04).all() and (b[geo][:, 2] <= 1).all() and (b[..., 3] == 1).all()
        assert ((g.C >> 24) == 255).all()
        assert (g.hdr.view(np.float16)[..., 3] == 1).all()
    # scene mode: geometry is in front of the sky sphere everywhere
    assert g.depth[g.depth > 0].min() > 0.1 / 75.0


def test_instances_and_args():
    b = synth.instances_random(1000, 3, center=(1, 2, 3), box=400.0)
    assert (b[:, 1, :3] > b[:, 0, :3]).all()
    e = (b[:, 1, :3] - b[:, 0, :3]) / 2
    assert e.min() >= 0.0499 and e.max() <= 5.001
    assert np.array_equal(synth.instances_random(10, 3, first=20), synth.instances_random(30, 3)[20:])
    a = synth.indirect_args_initial(5)
    assert a.shape == (5, 16) and (a[:, 11] == 1).all() and a[:, 14].tolist() == [0, 1, 2, 3, 4] and a.itemsize * 16 == 64
