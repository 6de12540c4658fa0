#!/usr/bin/env python3
"""What one rank of N launches per frame for Build HZB + Lighting, replicated against band-sharded (bench.py --hzb):
measured on ONE GPU by giving it the work of rank `r` of N = 4K rows / band rows.

    python tools/hzb_shard_timing.py [--size 3840x2160] [--bands 270 540 1080 2160] [--rank-of-band middle]

Per band height, back-to-back pairs over four cold buffer sets, us per frame-part of one rank:
  lighting alone      the band's fused Lighting+Sky launch
  replicated, rides   the WHOLE chain rides the band's launch (every rank builds the whole HZB; no exchange)
  sharded, rides      the band's pieces of mips 0-4 ride the launch; + the replicated single-workgroup tail as a launch of its own
                      (it runs behind the ranks' exchange of the slices), timed separately
and the bytes a rank adds to the per-frame exchange in sharded mode (its mips 0-4 slices, to every peer) beside its HDR band."""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--bands", type=int, nargs="+", default=[270, 540, 1080, 2160])
    ap.add_argument("--iters", type=int, default=600)
    a = ap.parse_args()
    import torch
    from unclerenderer_amd import assets, hostmath, synth
    from unclerenderer_amd.hotpath import HotPath, HzbLayout, to_device
    W, H = (int(v) for v in a.size.split("x"))
    hp = HotPath(0)
    fc = hostmath.build_frame_constants("sponza", W, H)
    g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, synth.SEED_BASE + 3)
    shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    ad = ROOT / "tests" / "golden" / "assets"
    env = assets.load_env_cube_dds(ad / "output_pmrem.dds")[0]
    lut = assets.load_brdf_lut_dds(ad / "PreintegratedGF.dds")
    tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, 256, 9), 256, 9, to_device(lut))
    ring = 4
    sets = [dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth), hdr=to_device(g.hdr)) for _ in range(ring)]
    lay = HzbLayout(W, H)
    hzb = torch.zeros(lay.total, device="cuda")

    def run(fn, n=a.iters, warm=1200, lighting=True):
        """(loop us per iteration between one event pair - host-bound for launches this short: a ctypes call is ~5 us -, mean duration
        of the Lighting DISPATCH from events carried on every 8th one: the GPU's own figure, what rocprofv3 would report)"""
        for k in range(warm):
            fn(k)
        pairs = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(n):
            if lighting and k % 8 == 0:
                x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                x.record(); y.record()
                pairs.append((x, y))
                hp.time_next_lighting(x, y)
            fn(k)
        e1.record()
        torch.cuda.synchronize()
        loop = e0.elapsed_time(e1) * 1e3 / n
        disp = float(np.mean([x.elapsed_time(y) for x, y in pairs])) * 1e3 if pairs else loop
        return loop, disp

    print(f"{W}x{H}; per rank and frame, us (back-to-back, cold buffer sets); N = {H} / band rows")
    print("dispatch = the Lighting launch's own duration (events carried on the dispatch); loop = back-to-back iterations, host-bound below ~15 us")
    print(f"{'band':>5s} {'N':>3s} | {'lighting':>9s} {'replicated':>11s} {'sharded':>8s}  (dispatch us) | {'lighting':>9s} {'replicated':>11s} {'sharded':>8s} (loop us) | {'tail launch':>11s} {'HDR band B':>11s} {'HZB slices B':>13s}")
    for band in a.bands:
        world = H // band
        rank = world // 2
        row0 = rank * band
        sl = slice(row0, row0 + band)
        p0, pn = lay.band_pieces(world, rank)

        def light(k):
            s = sets[k % ring]
            hp.deferred_lighting_sky(fc.scene, fc.sky, s["A"][sl], s["B"][sl], s["C"][sl], s["D"][sl], tables, s["hdr"][sl], W, H, row0, band)

        def replicated(k):
            hp.build_hzb(sets[k % ring]["D"], hzb, lay)
            light(k)

        def sharded(k):
            hp.build_hzb_band(sets[k % ring]["D"], hzb, lay, p0, pn)
            light(k)

        t_light = run(light)
        hp.defer_hzb_tail(2)
        t_rep = run(replicated)
        t_shard = run(sharded)
        hp.defer_hzb_tail(0)
        t_tail = run(lambda k: hp.build_hzb_tail(hzb, lay), n=300, warm=300, lighting=False)[0]
        slice_bytes = 4 * sum(c for _, c in lay.band_slices(p0, pn))
        print(f"{band:5d} {world:3d} | {t_light[1]:9.1f} {t_rep[1]:11.1f} {t_shard[1]:8.1f}               | {t_light[0]:9.1f} {t_rep[0]:11.1f} {t_shard[0]:8.1f}           | {t_tail:11.1f} {band * W * 8:11d} {slice_bytes if world > 1 else 0:13d}", flush=True)
    hp.flush()


if __name__ == "__main__":
    main()
