#!/usr/bin/env python3
"""Soak of the run-time tile claims with the Build HZB chain riding: hundreds of thousands of frames through ur_frame_render at three sizes
(4K defaults; 1080p with 4-tile chunks; 1440p with 8-tile chunks), HDR and HZB compared with the first frame every 997 frames, and the
context flushed at the end (a wave that gave up waiting for a claim would surface there as UR_ETIMEOUT).  python tools/soak_balance.py"""
import sys, time, hashlib
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
import numpy as np, torch
from unclerenderer_amd import assets, hostmath, lib, synth
from unclerenderer_amd.hotpath import Frame, HotPath, HzbLayout, to_device
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
def run(W, H, opts, frames, tag):
    hp = HotPath(0)
    for k, v in opts: hp.set_option(k, v)
    fc = hostmath.build_frame_constants("sponza", W, H)
    g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, synth.SEED_BASE + 3)
    shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    ad = ROOT / "tests" / "golden" / "assets"
    env = assets.load_env_cube_dds(ad / "output_pmrem.dds")[0]; lut = assets.load_brdf_lut_dds(ad / "PreintegratedGF.dds")
    tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, 256, 9), 256, 9, to_device(lut))
    lay = HzbLayout(W, H)
    hdr0 = to_device(g.hdr)
    bufs = [dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth), hdr=hdr0.clone()) for _ in range(2)]
    preset = hostmath.SCENES["sponza"]; n = preset.instance_count
    d_bounds = to_device(synth.instances_replicated(*preset.model_aabb, n)); d_args = to_device(synth.indirect_args_initial(n))
    d_vis, d_cnt = torch.zeros(n, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, n, True, lay.count, lay.width, lay.height, False)
    hzb = torch.zeros(lay.total, device="cuda")
    frame = Frame(hp)
    flags = lib.UR_FRAME_DEFAULT | lib.UR_FRAME_FUSE_LIGHTING_SKY | lib.UR_FRAME_HZB_WITH_LIGHTING
    for b in bufs:
        b["res"] = Frame.resources(W, H, 0, H, b["A"], b["B"], b["C"], b["D"], b["hdr"], b["D"], hzb, lay, tables, d_bounds, d_args, n, 0, d_vis, d_cnt)
    # reference digest
    b = bufs[0]; b["hdr"].copy_(hdr0); frame.render(b["res"], consts, fc.scene, fc.sky, flags); torch.cuda.synchronize()
    ref = b["hdr"].clone(); ref_hzb = hzb.clone()
    t0 = time.time(); bad = 0
    for k in range(frames):
        b = bufs[k % 2]
        if k % 997 == 0:
            b["hdr"].copy_(hdr0)
        frame.render(b["res"], consts, fc.scene, fc.sky, flags)
        if k % 997 == 0:
            torch.cuda.synchronize()
            if not (torch.equal(b["hdr"], ref) and torch.equal(hzb, ref_hzb)): bad += 1
        if k % 20000 == 0: print(tag, k, round(time.time() - t0, 1), "s", flush=True)
    torch.cuda.synchronize(); hp.flush()
    print(tag, "frames", frames, "mismatching checks", bad, "schedule", hp.lighting_schedule(), round(time.time() - t0, 1), "s", flush=True)
    frame.close(); hp.close()
    return bad
bad = run(3840, 2160, [], 60000, "4K default")
bad += run(1920, 1080, [(10, 2), (9, 6)], 150000, "1080p chunks of 4, pool 6/16")
bad += run(2560, 1440, [(10, 3), (9, 8)], 100000, "1440p chunks of 8, pool 8/16")
print("SOAK", "OK" if bad == 0 else "FAILED")
