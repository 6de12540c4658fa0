#!/usr/bin/env python3
"""Mint the golden fixtures (tests/golden/hotpath_golden.npz) from the CPU oracle on seeded inputs.

The reference ships no golden vectors for this path and cannot run here, so these fixtures pin the
ORACLE AGAINST ITSELF over time (regression), not against the reference: parity with the reference stays "unpinned".
Run from the repository root:  python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

from oracle import oracle as o  # noqa: E402
from unclerenderer_amd import hostmath, synth  # noqa: E402


def build():
    o.build()
    out = {}
    # ---- HZB chains (incl. the H8 quirk size 17x9 and odd sizes)
    for (w, h) in [(3, 5), (17, 9), (64, 36), (129, 67)]:
        d = np.random.default_rng(w * 1000 + h).random((h, w), dtype=np.float32)
        mips, total = o.hzb_layout(w, h)
        out[f"hzb_{w}x{h}_depth"] = d
        out[f"hzb_{w}x{h}_out"] = o.build_hzb(d, mips, total)
    # ---- cull: 512 random instances + HZB from a scene depth
    w, h = 160, 90
    fc = hostmath.build_frame_constants("sponza", w, h)
    g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, 123)
    mips, total = o.hzb_layout(w, h)
    hzb = np.nan_to_num(o.build_hzb(g.depth, mips, total))
    n = 512
    bounds = synth.instances_random(n, 123, center=fc.camera_position, box=60.0)
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, n, True, len(mips), mips[0][1], mips[0][2], True)
    args, stats, vis, cnt = o.cull_indirect_args(consts, bounds, hzb, mips, synth.indirect_args_initial(n))
    out.update(cull_depth=g.depth, cull_hzb=hzb, cull_bounds=bounds, cull_consts=consts, cull_words=args[:, 11].copy(), cull_stats=stats,
               cull_visible=vis, cull_mips=np.array(mips, np.uint32))
    # ---- lighting + sky: 48x27 tile, both G-buffer generators
    w, h = 48, 27
    fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=64, env_mip_count=4)
    shadow = synth.shadow_map_noise(64, 9)
    env = synth.env_cube_procedural(8, 4)
    lut = synth.brdf_lut_procedural(16, 8)
    out.update(light_shadow=shadow, light_env=env, light_lut=lut, light_scene_bytes=np.frombuffer(bytes(fc.scene), np.uint8).copy(),
               light_sky_bytes=np.frombuffer(bytes(fc.sky), np.uint8).copy())
    for mode in ("iid", "scene"):
        g = synth.gbuffer_iid(w, h, 9) if mode == "iid" else synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, 9)
        lit, frag = o.deferred_lighting(fc.scene, g.A, g.B, g.C, shadow, env, 8, 4, lut, g.hdr, w, h, want_fragile=True)
        final = o.sky_atmosphere(fc.sky, g.depth, lit, w, h)
        out.update({f"light_{mode}_A": g.A, f"light_{mode}_B": g.B, f"light_{mode}_C": g.C, f"light_{mode}_hdr": g.hdr, f"light_{mode}_depth": g.depth,
                    f"light_{mode}_lit": lit, f"light_{mode}_final": final, f"light_{mode}_fragile": frag})
    return out


if __name__ == "__main__":
    data = build()
    path = Path(__file__).resolve().parent / "hotpath_golden.npz"
    np.savez_compressed(path, **data)
    print(f"wrote {path} ({path.stat().st_size / 1024:.0f} KiB, {len(data)} arrays)")
