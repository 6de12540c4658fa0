"""One rank of the multi-rank GPU tests (tests/test_gpu_multirank.py): started as a fresh process by tests/_spawner.py,
one per rank, all on GPU 0 with the gloo backend (RCCL refuses two ranks on one device; the sharding, the index bases and
the gathers of unclerenderer_amd/dist.py are the same calls bench.py makes under RCCL). Every rank runs the HIP path on
ITS shard — Frame.render on its row band with its instance range and index_base, or the 1 M-instance cull on its range —
gathers through dist.py, and writes what it ended up with: sha256 of every gathered array (all ranks must agree), and on
rank 0 the arrays themselves for the byte comparison with the single-rank HIP result.

    RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/_multirank_worker.py --case c4|c5 --out DIR
"""
import argparse
import hashlib
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
ASSETS = ROOT / "tests" / "golden" / "assets"


def sha(t) -> str:
    return hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()


def c4_inputs(w, h, row0, rows):
    """pica_pica's camera and light at w x h (BASELINE config 4), SURVEY section 8d's generator: any rank makes any band."""
    from unclerenderer_amd import hostmath, scene, synth
    fc = hostmath.build_frame_constants("pica_pica", w, h, shadow_size=2048, env_mip_count=9)
    seed = synth.SEED_BASE + 4
    g = synth.gbuffer_iid(w, h, seed, row0, rows)
    depth_full = g.depth if rows == h else synth.gbuffer_iid(w, h, seed).depth
    shadow = synth.shadow_map_noise(2048, seed)
    sb = scene.load_scene_bounds(ASSETS / "Scenes" / "pica_pica.json")
    return fc, g, depth_full, shadow, sb.bounds


def c4_render(hp, w, h, rank, world, frames=2, shard_hzb=False):
    """The frame as bench.py drives it (render graph: cull with last frame's HZB -> Build HZB riding the fused Lighting+Sky
    launch -> Tonemap), `frames` times so that the last cull reads an HZB. Returns the device tensors of the band."""
    import torch
    from unclerenderer_amd import assets, hostmath, lib, synth
    from unclerenderer_amd import dist as urdist
    from unclerenderer_amd.hotpath import Frame, HzbLayout, to_device
    plan = urdist.plan_bands(h, world, rank)
    fc, g, depth_full, shadow, bounds = c4_inputs(w, h, plan.row0, plan.rows)
    env, base, mips, _ = assets.load_env_cube_dds(ASSETS / "output_pmrem.dds")
    lut = assets.load_brdf_lut_dds(ASSETS / "PreintegratedGF.dds")
    tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, base, mips), base, mips, to_device(lut))
    lay = HzbLayout(w, h)
    n = bounds.shape[0]
    i0, i1 = urdist.plan_instances(n, world, rank)
    d = dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), depth_band=to_device(g.depth), depth_full=to_device(depth_full),
             hdr_band=to_device(g.hdr), ldr_band=torch.zeros((plan.rows, w), dtype=torch.int32, device="cuda"),
             hzb=torch.zeros(lay.total, dtype=torch.float32, device="cuda"),
             bounds=to_device(np.ascontiguousarray(bounds[i0:i1])), args=to_device(synth.indirect_args_initial(n)[i0:i1]),
             vis=torch.zeros(max(1, i1 - i0), dtype=torch.int32, device="cuda"), cnt=torch.zeros(1, dtype=torch.int32, device="cuda"))
    frame = Frame(hp, frames_in_flight=3, rank=rank, world_size=world)
    res = Frame.resources(w, h, plan.row0, plan.rows, d["A"], d["B"], d["C"], d["depth_band"], d["hdr_band"], d["depth_full"], d["hzb"], lay, tables,
                          d["bounds"], d["args"], i1 - i0, i0, d["vis"], d["cnt"], None, d["ldr_band"])
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, i1 - i0, True, lay.count, lay.width, lay.height, False)
    flags = lib.UR_FRAME_DEFAULT | lib.UR_FRAME_FUSE_LIGHTING_SKY | lib.UR_FRAME_HZB_WITH_LIGHTING | lib.UR_FRAME_TONEMAP
    if shard_hzb:
        flags |= lib.UR_FRAME_HZB_SHARD  # (one rank: the whole chain as usual)
    hdr0 = d["hdr_band"].clone()
    for k in range(frames):
        d["hdr_band"].copy_(hdr0)  # (the Lighting pass blends into its target: every frame starts from the emissive pre-fill)
        frame.render(res, consts, fc.scene, fc.sky, flags)
        if shard_hzb and world > 1:
            # band-sharded Build HZB: the frame built this rank's pieces of mips 0..4 (riding its Lighting launch); the ranks
            # exchange the slices peer to peer and every rank runs the tail - what bench.py --hzb shard does per frame
            d["hzb_sent"] = urdist.allgather_hzb_slices(d["hzb"], lay)[0]
            hp.build_hzb_tail(d["hzb"], lay)
    torch.cuda.synchronize()
    frame.close()
    return plan, d, n


def gather_all(torch, urdist, plan, d, w, h, n, world):
    """Every gather of dist.py on the band's results; device tensors throughout."""
    out = {}
    for mode in ("ring", "direct"):
        full = torch.zeros((h, w, 4), dtype=torch.int16, device="cuda")
        urdist.allgather_hdr(full, d["hdr_band"], mode=mode)
        out[f"hdr_{mode}"] = full
        ldr = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        urdist.allgather_rows(ldr, d["ldr_band"], mode=mode)
        out[f"ldr_{mode}"] = ldr
    full = torch.zeros((h, w, 4), dtype=torch.int16, device="cuda")
    work = urdist.allgather_hdr(full, d["hdr_band"], async_op=True)
    if work is not None:
        work.wait()
    out["hdr_async"] = full
    vis, cnt = urdist.allgather_visible(d["vis"], d["cnt"])
    out["vis"] = vis[:cnt].clone()
    if n % world == 0:  # the InstanceCount words: equal command slices gathered like rows
        words = torch.zeros((n, 16), dtype=torch.int32, device="cuda")
        urdist.allgather_rows(words, d["args"].view(torch.int32).view(-1, 16))
        out["args"] = words
    torch.cuda.synchronize()
    return out


def c5_inputs(n):
    from unclerenderer_amd import hostmath, synth
    w, h = 7680, 4320
    fc = hostmath.build_frame_constants("sponza", w, h)
    depth = synth.gbuffer_iid(w, h, synth.SEED_BASE + 5).depth
    bounds = synth.instances_random(n, synth.SEED_BASE + 5, center=fc.camera_position, box=400.0)
    return w, h, fc, depth, bounds


def c5_cull(hp, n, rank, world):
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd import dist as urdist
    from unclerenderer_amd.hotpath import HzbLayout, to_device
    w, h, fc, depth, bounds = c5_inputs(n)
    lay = HzbLayout(w, h)
    hzb = torch.zeros(lay.total, device="cuda")
    hp.build_hzb(to_device(depth), hzb, lay)  # replicated: every rank builds the whole chain
    i0, i1 = urdist.plan_instances(n, world, rank)
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, i1 - i0, True, lay.count, lay.width, lay.height, True)
    d = dict(args=to_device(synth.indirect_args_initial(n)[i0:i1]), stats=torch.zeros(2, dtype=torch.int32, device="cuda"),
             vis=torch.full((max(1, i1 - i0),), -1, dtype=torch.int32, device="cuda"), cnt=torch.zeros(1, dtype=torch.int32, device="cuda"))
    hp.cull_indirect_args(consts, to_device(np.ascontiguousarray(bounds[i0:i1])), hzb, lay, d["args"], d["stats"], d["vis"], d["cnt"], index_base=i0)
    torch.cuda.synchronize()
    return d, hzb


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--instances", type=int, default=1_000_000)
    ap.add_argument("--shard-hzb", action="store_true")
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    from unclerenderer_amd import dist as urdist
    from unclerenderer_amd.hotpath import HotPath
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hp = HotPath(0)
    try:
        if a.case == "c4":
            w, h = a.width, a.height
            plan, d, n = c4_render(hp, w, h, rank, world, shard_hzb=a.shard_hzb)
            out = gather_all(torch, urdist, plan, d, w, h, n, world)
            if a.shard_hzb:
                out["hzb"] = d["hzb"].clone()  # complete on every rank: gathered slices + the replicated tail
        else:
            n = a.instances
            d, _ = c5_cull(hp, n, rank, world)
            vis, cnt = urdist.allgather_visible(d["vis"], d["cnt"])
            words = torch.zeros((n, 16), dtype=torch.int32, device="cuda")
            urdist.allgather_rows(words, d["args"].view(torch.int32).view(-1, 16), mode="direct")
            stats = d["stats"].clone()
            dist.all_reduce(stats)  # the two counters are sums over instances
            torch.cuda.synchronize()
            out = {"vis": vis[:cnt].clone(), "args": words, "stats": stats}
        digest = {k: sha(v) for k, v in out.items()}
        Path(a.out).mkdir(parents=True, exist_ok=True)
        (Path(a.out) / f"rank{rank}.json").write_text(json.dumps(digest))
        if rank == 0:
            np.savez(Path(a.out) / "rank0.npz", **{k: v.cpu().numpy() for k, v in out.items()})
        dist.barrier()
    finally:
        hp.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
