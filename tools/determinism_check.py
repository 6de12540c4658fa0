#!/usr/bin/env python3
"""Run-to-run determinism of the fused lighting kernel: the same inputs shaded N times must give the same bytes every
time (the streaming kernel claims tiles dynamically and prefetches them by LDS-DMA: any ordering hole would show up as a
run that differs).   python tools/determinism_check.py [--runs 200]"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=200)
    a = ap.parse_args()
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import HotPath, to_device
    hp = HotPath(0)
    bad = 0
    for (w, h, mode) in [(3840, 2160, "scene"), (1920, 1080, "iid"), (272, 33, "scene")]:
        fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=512, env_mip_count=6)
        g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, 7) if mode == "scene" else synth.gbuffer_iid(w, h, 7)
        shadow = synth.shadow_map_noise(512, 7)
        env, lut = synth.env_cube_procedural(32, 6), synth.brdf_lut_procedural(128, 32)
        tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, 32, 6), 32, 6, to_device(lut))
        A, B, C, D, hdr0 = to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth), to_device(g.hdr)
        ref = None
        diff_runs = 0
        for r in range(a.runs):
            out = hdr0.clone()
            hp.deferred_lighting_sky(fc.scene, fc.sky, A, B, C, D, tables, out, w, h)
            if ref is None:
                torch.cuda.synchronize()
                ref = out.clone()
            elif not torch.equal(out, ref):
                diff_runs += 1
        torch.cuda.synchronize()
        print(f"{w}x{h} {mode}: {a.runs} runs, {diff_runs} differ from the first")
        bad += diff_runs
    raise SystemExit(1 if bad else 0)


if __name__ == "__main__":
    main()
