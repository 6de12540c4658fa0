#!/usr/bin/env python3
"""Per-kernel timing on one GPU (development aid; bench.py is the contract benchmark).

    python tools/bench_kernels.py [--gbuffer scene|iid|both] [--iters 50] [--hzb] [--cull]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def time_events(torch, fn, iters, warm=5, reps=7):
    """Back-to-back launches between ONE event pair (steady-state per-launch time), repeated; returns (median, min) us."""
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(iters):
            fn(k)
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / iters)
    return float(np.median(out)), float(min(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gbuffer", default="both")
    ap.add_argument("--iters", type=int, default=300)  # long enough for the chip to reach its sustained clock
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--ring", type=int, default=4)
    ap.add_argument("--hzb", action="store_true")
    ap.add_argument("--cull", action="store_true")
    ap.add_argument("--no-light", action="store_true")
    ap.add_argument("--post", action="store_true", help="the rows behind the path: Tonemap (12 B/pixel) and TemporalAA resolve (24 B/pixel)")
    ap.add_argument("--cache", default="", help="directory for the generated inputs (npz): repeated runs on one box skip the generators")
    ap.add_argument("--no-shadows", action="store_true", help="ShadowStrength = 0: the SHADOWS=false instantiation")
    ap.add_argument("--tag", default="", help="printed in front of every line (which library variant this is)")
    a = ap.parse_args()
    import torch
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import HotPath, HzbLayout, to_device
    hp = HotPath(0)
    W, H = a.width, a.height
    fc = hostmath.build_frame_constants("sponza", W, H)
    if a.no_shadows:
        fc.scene.ShadowStrength = 0.0
    asset_dir = Path(__file__).resolve().parent.parent / "tests" / "golden" / "assets"
    if (asset_dir / "output_pmrem.dds").exists():
        from unclerenderer_amd import assets
        env = assets.load_env_cube_dds(asset_dir / "output_pmrem.dds")[0]
        lut = assets.load_brdf_lut_dds(asset_dir / "PreintegratedGF.dds")
    else:
        env = synth.env_cube_procedural(256, 9, sun_dir=fc.light_direction)
        lut = synth.brdf_lut_procedural(128, 32)
    d_env = hp.stage_env_cube(env, 256, 9)
    modes = ["scene", "iid"] if a.gbuffer == "both" else [a.gbuffer]
    for mode in modes if not a.no_light else []:
        t0 = time.time()
        cache = Path(a.cache) / f"g_{mode}_{W}x{H}.npz" if a.cache else None
        if cache is not None and cache.exists():
            z = np.load(cache)
            g = synth.GBuffer(W, H, 0, H, z["A"], z["B"], z["C"], z["hdr"], z["depth"])
            shadow = z["shadow"]
        else:
            if mode == "scene":
                g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, 3)
                shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
            else:
                g = synth.gbuffer_iid(W, H, 3)
                shadow = synth.shadow_map_noise(2048, 3)
            if cache is not None:
                cache.parent.mkdir(parents=True, exist_ok=True)
                np.savez(cache, A=g.A, B=g.B, C=g.C, hdr=g.hdr, depth=g.depth, shadow=shadow)
        tables = hp.make_tables(to_device(shadow), d_env, 256, 9, to_device(lut))
        sets = [dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth), hdr=to_device(g.hdr)) for _ in range(a.ring)]
        n_sky = int((g.depth == 0).sum()); n_geo = g.depth.size - n_sky
        nbytes = 40 * n_geo + 12 * n_sky

        def fused(k):
            s = sets[k % a.ring]
            hp.deferred_lighting_sky(fc.scene, fc.sky, s["A"], s["B"], s["C"], s["D"], tables, s["hdr"], W, H)
        med, mn = time_events(torch, fused, a.iters)
        print(f"{a.tag}[{mode}] fused lighting+sky {W}x{H}: median {med:.1f} us  min {mn:.1f} us  {nbytes / med / 1e3:.0f} GB/s  "
              f"{100 * nbytes / med / 1e3 / 8000:.1f}% of 8 TB/s  {W * H / med:.0f} Mpx/s  (gen {time.time() - t0:.1f}s, sky {n_sky / g.depth.size:.3f})", flush=True)
    if a.hzb:
        for (w, h) in [(1920, 1080), (3840, 2160), (7680, 4320)]:
            lay = HzbLayout(w, h)
            d = torch.rand(h * w, device="cuda")
            hz = torch.zeros(lay.total, device="cuda")
            med, mn = time_events(torch, lambda k: hp.build_hzb(d, hz, lay), a.iters)
            b = 4 * (w * h + lay.mip_texels())
            print(f"[hzb] {w}x{h}: median {med:.1f} us min {mn:.1f} us {b / med / 1e3:.0f} GB/s ({100 * b / med / 1e3 / 8000:.1f}%)", flush=True)
    if a.cull:
        for n in [25, 170, 100_000, 1_000_000, 8_000_000]:
            W8, H8 = 7680, 4320
            fc8 = hostmath.build_frame_constants("sponza", W8, H8)
            lay = HzbLayout(W8, H8)
            hz = torch.rand(lay.total, device="cuda") * 0.01
            b = to_device(synth.instances_random(n, 5, center=fc8.camera_position, box=400.0))
            args = to_device(synth.indirect_args_initial(n))
            vis = torch.zeros(n, dtype=torch.int32, device="cuda"); cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
            c = hostmath.pack_culling_constants(fc8.view, fc8.proj, n, True, lay.count, lay.width, lay.height, False)
            med, mn = time_events(torch, lambda k: hp.cull_indirect_args(c, b, hz, lay, args, None, vis, cnt), a.iters)
            med2, _ = time_events(torch, lambda k: hp.cull_indirect_args(c, b, hz, lay, args), a.iters)
            print(f"[cull] n={n}: with list median {med:.1f} us min {mn:.1f} ({n / med:.0f} M inst/s); words only {med2:.1f} us; visible {int(cnt.cpu()[0])}", flush=True)
    if a.post:
        for (w, h) in [(1920, 1080), (3840, 2160), (7680, 4320)]:
            ring = 6 if w < 7680 else 3  # distinct buffer sets: the inputs of a launch are not in cache from the previous one
            hdr = [(torch.rand((h, w, 4), device="cuda") * 4.0).to(torch.float16).view(torch.int16) for _ in range(ring)]
            hist = [(torch.rand((h, w, 4), device="cuda") * 4.0).to(torch.float16).view(torch.int16) for _ in range(ring)]
            out8 = [torch.zeros((h, w), dtype=torch.int32, device="cuda") for _ in range(ring)]
            out16 = [torch.zeros((h, w, 4), dtype=torch.int16, device="cuda") for _ in range(ring)]
            med, mn = time_events(torch, lambda k: hp.tonemap(hdr[k % ring], out8[k % ring], w, h), a.iters)
            b = 12 * w * h
            print(f"[tonemap] {w}x{h}: median {med:.1f} us min {mn:.1f} us {b / med / 1e3:.0f} GB/s ({100 * b / med / 1e3 / 8000:.1f}% of 8 TB/s)", flush=True)
            med, mn = time_events(torch, lambda k: hp.temporal_aa(hdr[k % ring], hist[k % ring], out16[k % ring], 0.9, True, w, h), a.iters)
            b = 24 * w * h
            print(f"[taa] {w}x{h}: median {med:.1f} us min {mn:.1f} us {b / med / 1e3:.0f} GB/s ({100 * b / med / 1e3 / 8000:.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
