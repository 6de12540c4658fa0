"""N > 1 path on CPU: two gloo ranks shard the frame by row bands and the cull by instance ranges, exchange with the
same collectives bench.py uses (unclerenderer_amd/dist.py), and must reproduce the single-rank result byte for byte.
The per-band compute here is the ORACLE (this is a CPU test of the sharding + collectives, not of the HIP kernels;
the GPU suite checks band-split == whole-frame on the device)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from unclerenderer_amd import dist as urdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, n_inst, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as o
        from unclerenderer_amd import hostmath, synth
        fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=64, env_mip_count=4)
        plan = urdist.plan_bands(h, world, rank)
        g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, 17, plan.row0, plan.rows)
        shadow, env, lut = synth.shadow_map_noise(64, 17), synth.env_cube_procedural(8, 4), synth.brdf_lut_procedural(16, 8)
        lit = o.deferred_lighting(fc.scene, g.A, g.B, g.C, shadow, env, 8, 4, lut, g.hdr, w, h, plan.row0, plan.rows)
        band = o.sky_atmosphere(fc.sky, g.depth, lit, w, h, plan.row0, plan.rows)
        hdr_full = torch.zeros((h, w, 4), dtype=torch.int16)
        urdist.allgather_hdr(hdr_full, torch.from_numpy(band.view(np.int16)))
        # the overlapped form bench.py uses: a Work handle, waited for before the buffers are touched again
        hdr_async = torch.zeros((h, w, 4), dtype=torch.int16)
        work = urdist.allgather_hdr(hdr_async, torch.from_numpy(band.view(np.int16)), async_op=True)
        assert work is not None
        work.wait()
        assert torch.equal(hdr_async, hdr_full)
        # the direct form (N - 1 isend / irecv pairs per rank in one batch; on RCCL a grouped ncclSend / ncclRecv): same bytes,
        # blocking and as a handle
        hdr_direct = torch.zeros((h, w, 4), dtype=torch.int16)
        assert urdist.allgather_hdr(hdr_direct, torch.from_numpy(band.view(np.int16)), mode="direct") is None
        assert torch.equal(hdr_direct, hdr_full)
        hdr_direct2 = torch.zeros((h, w, 4), dtype=torch.int16)
        work = urdist.allgather_hdr(hdr_direct2, torch.from_numpy(band.view(np.int16)), async_op=True, mode="direct")
        work.wait()
        assert torch.equal(hdr_direct2, hdr_full)
        # Tonemap ahead of the gather (bench.py --gather-ldr): the band is tonemapped where it was shaded and the 4-byte
        # pixels travel instead of the 8-byte ones
        ldr_band = o.tonemap(band, exposure=0.9, gamma=2.2)
        ldr_full = torch.zeros((h, w), dtype=torch.int32)
        work = urdist.allgather_rows(ldr_full, torch.from_numpy(ldr_band.view(np.int32)), async_op=True)
        work.wait()
        ldr_direct = torch.zeros((h, w), dtype=torch.int32)
        urdist.allgather_rows(ldr_direct, torch.from_numpy(ldr_band.view(np.int32)), mode="direct")
        assert torch.equal(ldr_direct, ldr_full)
        assert ldr_full.element_size() * ldr_full[0].numel() * 2 == hdr_full.element_size() * hdr_full[0].numel()  # half the bytes per row
        # cull: instance ranges + replicated HZB
        depth_full = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, 17).depth
        mips, total = o.hzb_layout(w, h)
        hzb = np.nan_to_num(o.build_hzb(depth_full, mips, total))
        i0, i1 = urdist.plan_instances(n_inst, world, rank)
        bounds = synth.instances_random(i1 - i0, 17, center=fc.camera_position, box=60.0, first=i0)
        consts = hostmath.pack_culling_constants(fc.view, fc.proj, i1 - i0, True, len(mips), mips[0][1], mips[0][2], False)
        _, _, vis, cnt = o.cull_indirect_args(consts, bounds, hzb, mips, synth.indirect_args_initial(i1 - i0), index_base=i0)
        pad = np.zeros(max(i1 - i0, 1), np.int32); pad[:cnt] = vis.view(np.int32)
        all_vis, total_cnt = urdist.allgather_visible(torch.from_numpy(pad), torch.tensor([cnt], dtype=torch.int32))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), hdr=hdr_full.numpy(), ldr=ldr_full.numpy(), vis=all_vis.numpy(), cnt=total_cnt)
    finally:
        dist.destroy_process_group()


def test_two_ranks_reproduce_single_rank(tmp_path, oracle, urlib):
    w, h, n = 64, 36, 3001
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), w, h, n, str(tmp_path)), nprocs=world, join=True)
    from unclerenderer_amd import hostmath, synth
    fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=64, env_mip_count=4)
    g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, 17)
    shadow, env, lut = synth.shadow_map_noise(64, 17), synth.env_cube_procedural(8, 4), synth.brdf_lut_procedural(16, 8)
    lit = oracle.deferred_lighting(fc.scene, g.A, g.B, g.C, shadow, env, 8, 4, lut, g.hdr, w, h)
    ref = oracle.sky_atmosphere(fc.sky, g.depth, lit, w, h)
    ref_ldr = oracle.tonemap(ref, exposure=0.9, gamma=2.2)
    mips, total = oracle.hzb_layout(w, h)
    hzb = np.nan_to_num(oracle.build_hzb(g.depth, mips, total))
    bounds = synth.instances_random(n, 17, center=fc.camera_position, box=60.0)
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, n, True, len(mips), mips[0][1], mips[0][2], False)
    _, _, ref_vis, ref_cnt = oracle.cull_indirect_args(consts, bounds, hzb, mips, synth.indirect_args_initial(n))
    assert 0 < ref_cnt < n
    for r in range(world):
        d = np.load(tmp_path / f"rank{r}.npz")
        assert np.array_equal(d["hdr"].view(np.uint16), ref), f"rank {r}: gathered HDR differs from the single-rank frame"
        assert np.array_equal(d["ldr"].view(np.uint32), ref_ldr), f"rank {r}: gathered tonemapped frame differs from tonemapping the single-rank frame"
        assert int(d["cnt"]) == ref_cnt and np.array_equal(d["vis"].view(np.uint32), ref_vis)


def test_plans():
    assert urdist.plan_bands(2160, 8, 3) == urdist.BandPlan(3, 8, 2160, 810, 270)
    with pytest.raises(ValueError):
        urdist.plan_bands(2161, 8, 0)
    ranges = [urdist.plan_instances(25, 8, r) for r in range(8)]
    assert ranges[0][0] == 0 and ranges[-1][1] == 25 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    h = torch.zeros((4, 2, 4), dtype=torch.int16)
    urdist.allgather_hdr(h, torch.ones((4, 2, 4), dtype=torch.int16))  # world 1: plain copy
    assert (h == 1).all()


def _hzb_worker(rank, world, port, w, h, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as o
        from unclerenderer_amd import synth
        from unclerenderer_amd.hotpath import HzbLayout
        depth = (synth.hash_unit(23, *synth._grid(w, 0, h), 0)).astype(np.float32)
        lay = HzbLayout(w, h)
        ref = np.nan_to_num(o.build_hzb(depth, lay.as_list(), lay.total))
        mine = np.full(lay.total, -1.0, np.float32)  # what this rank has after its band launch: its own slices of mips 0..4, nothing else
        for off, cnt in lay.band_slices(*lay.band_pieces(world, rank)):
            mine[off:off + cnt] = ref[off:off + cnt]
        t = torch.from_numpy(mine)
        sent, _ = urdist.allgather_hzb_slices(t, lay)
        sent2, work = urdist.allgather_hzb_slices(t, lay, async_op=True)  # idempotent; the handle form
        work.wait()
        assert sent == sent2 == 4 * sum(c for _, c in lay.band_slices(*lay.band_pieces(world, rank)))
        np.save(os.path.join(out_dir, f"hzb{rank}.npy"), t.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_band_sharded_hzb_slices_gather_into_the_whole_levels(tmp_path, oracle, urlib, world):
    """Band-sharded Build HZB (ur_build_hzb_band + dist.allgather_hzb_slices): the piece rows of the ranks partition the wide
    launch's pieces, their slices partition mips 0..4, and after the peer-to-peer exchange every rank holds those five levels
    whole (the tail levels are each rank's own launch afterwards)."""
    from unclerenderer_amd import synth
    from unclerenderer_amd.hotpath import HzbLayout
    w, h = 640, 360
    lay = HzbLayout(w, h)
    pieces = [lay.band_pieces(world, r) for r in range(world)]
    assert pieces[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(pieces, pieces[1:])) and sum(p[1] for p in pieces) == (h + 31) // 32
    covered = np.zeros(lay.total, np.int32)
    for p in pieces:
        for off, cnt in lay.band_slices(*p):
            covered[off:off + cnt] += 1
    for k in range(lay.count):
        m = lay.mips[k]
        lvl = covered[m.offset:m.offset + m.width * m.height]
        assert (lvl == (1 if k < 5 else 0)).all(), k
    mp.spawn(_hzb_worker, args=(world, _free_port(), w, h, str(tmp_path)), nprocs=world, join=True)
    depth = (synth.hash_unit(23, *synth._grid(w, 0, h), 0)).astype(np.float32)
    ref = np.nan_to_num(oracle.build_hzb(depth, lay.as_list(), lay.total))
    for r in range(world):
        got = np.load(tmp_path / f"hzb{r}.npy")
        for k in range(5):
            m = lay.mips[k]
            sl = slice(m.offset, m.offset + m.width * m.height)
            assert np.array_equal(got[sl].view(np.uint32), ref[sl].view(np.uint32)), (r, k)
