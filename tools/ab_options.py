#!/usr/bin/env python3
"""Same-box, same-process A/B of launch-shape options (ur_set_option) on the bench workload: the 4K Sponza frame.

    python tools/ab_options.py static:8=0 dyn:8=1 dyn8:8=1,10=3 [--rounds 5] [--size 3840x2160] [--rows 2160]

Every set is NAME:opt=value[,opt=value...] with the UR_OPT_* numbers of include/ur_hotpath.h. The sets are run interleaved
(round-robin, --rounds times) on ONE context, so clock state and box are shared. Per set and round:
  alone  fused Lighting+Sky launches back to back over four cold buffer sets between one event pair (us per launch)
  frame  ur_frame_render (cull of 25 + Lighting carrying the Build HZB chain), us per frame between one event pair, and the
         Lighting dispatch's own duration from events carried on the dispatches (every 4th frame)
The HDR outputs of all sets are compared byte for byte (options must not change results)."""
import argparse
import hashlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sets", nargs="+")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--rows", type=int, default=0, help="shade only the first ROWS rows (a rank's band)")
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--gbuffer", default="scene")
    ap.add_argument("--no-frame", action="store_true")
    ap.add_argument("--settle", type=int, default=1500, help="untimed launches / frames of a set in front of each of its measurements (the clock follows the load over milliseconds)")
    a = ap.parse_args()
    import torch
    from unclerenderer_amd import assets, hostmath, lib, synth
    from unclerenderer_amd.hotpath import Frame, HotPath, HzbLayout, to_device

    W, H = (int(v) for v in a.size.split("x"))
    rows = a.rows or H
    sets = []
    for spec in a.sets:
        name, _, kv = spec.partition(":")
        sets.append((name, [tuple(int(x) for x in p.split("=")) for p in kv.split(",") if p]))
    hp = HotPath(0)
    defaults = {k: hp.get_option(k) for k in (1, 2, 3, 4, 5, 7, 8, 9, 10)}
    fc = hostmath.build_frame_constants("sponza", W, H, shadow_size=2048, env_mip_count=9)
    if a.gbuffer == "scene":
        g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, synth.SEED_BASE + 3)
        shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    else:
        g = synth.gbuffer_iid(W, H, synth.SEED_BASE + 3)
        shadow = synth.shadow_map_noise(2048, synth.SEED_BASE + 3)
    ad = ROOT / "tests" / "golden" / "assets"
    env = assets.load_env_cube_dds(ad / "output_pmrem.dds")[0]
    lut = assets.load_brdf_lut_dds(ad / "PreintegratedGF.dds")
    tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, 256, 9), 256, 9, to_device(lut))
    lay = HzbLayout(W, H)
    ring = 4
    hdr0 = to_device(g.hdr)
    bufs = [dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth), hdr=hdr0.clone()) for _ in range(ring)]
    preset = hostmath.SCENES["sponza"]
    n_inst = preset.instance_count
    d_bounds = to_device(synth.instances_replicated(*preset.model_aabb, n_inst))
    d_args = to_device(synth.indirect_args_initial(n_inst))
    d_vis, d_cnt = torch.zeros(n_inst, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, n_inst, True, lay.count, lay.width, lay.height, False)
    hzb = torch.zeros(lay.total, device="cuda")
    frame = Frame(hp)
    flags = lib.UR_FRAME_DEFAULT | lib.UR_FRAME_FUSE_LIGHTING_SKY | lib.UR_FRAME_HZB_WITH_LIGHTING
    for b in bufs:
        b["res"] = Frame.resources(W, H, 0, rows, b["A"], b["B"], b["C"], b["D"], b["hdr"], b["D"], hzb, lay, tables, d_bounds, d_args, n_inst, 0, d_vis, d_cnt)

    def apply(opts):
        for k, v in defaults.items():
            hp.set_option(k, v)
        for k, v in opts:
            hp.set_option(k, v)

    def alone(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(n):
            b = bufs[k % ring]
            hp.deferred_lighting_sky(fc.scene, fc.sky, b["A"], b["B"], b["C"], b["D"], tables, b["hdr"], W, H, 0, rows)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    def frames(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(n):
            frame.render(bufs[k % ring]["res"], consts, fc.scene, fc.sky, flags | (lib.UR_FRAME_TIME_LIGHTING_KERNEL if k % 4 == 0 else 0))
        e1.record()
        torch.cuda.synchronize()
        t = frame.lighting_times_and_record_cost_ms()[0]
        return e0.elapsed_time(e1) * 1e3 / n, float(np.mean(t)) * 1e3 if t.size else float("nan")

    # results must not depend on the options
    digests = {}
    for name, opts in sets:
        apply(opts)
        b = bufs[0]
        b["hdr"].copy_(hdr0)
        hp.deferred_lighting_sky(fc.scene, fc.sky, b["A"], b["B"], b["C"], b["D"], tables, b["hdr"], W, H, 0, rows)
        torch.cuda.synchronize()
        digests[name] = hashlib.sha256(b["hdr"].cpu().numpy().tobytes()).hexdigest()[:16]
    print("hdr digests:", digests, "SAME" if len(set(digests.values())) == 1 else "DIFFERENT", flush=True)

    apply(sets[0][1])
    alone(1500)  # clock ramp
    res = {name: {"alone": [], "frame": [], "light": []} for name, _ in sets}
    for r in range(a.rounds):
        for name, opts in sets:
            apply(opts)
            alone(a.settle)
            res[name]["alone"].append(alone(a.iters))
            if not a.no_frame:
                frames(a.settle)
                f, l = frames(a.iters)
                res[name]["frame"].append(f)
                res[name]["light"].append(l)
        print("round", r, {n: (round(v["alone"][-1], 2), round(v["frame"][-1], 2) if v["frame"] else None, round(v["light"][-1], 2) if v["light"] else None) for n, v in res.items()}, flush=True)
    print(f"{'set':12s} {'alone med':>10s} {'alone min':>10s} {'frame med':>10s} {'light-in-frame med':>19s}   ({W}x{H}, rows {rows}, {a.gbuffer})")
    for name, v in res.items():
        fm = np.median(v["frame"]) if v["frame"] else float("nan")
        lm = np.median(v["light"]) if v["light"] else float("nan")
        print(f"{name:12s} {np.median(v['alone']):10.2f} {np.min(v['alone']):10.2f} {fm:10.2f} {lm:19.2f}")
    try:
        hp.flush()
        print("flush: ok")
    except Exception as e:  # UR_ETIMEOUT would surface here
        print("flush:", e)


if __name__ == "__main__":
    main()
