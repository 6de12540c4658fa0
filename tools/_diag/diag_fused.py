import numpy as np, torch, sys
sys.path.insert(0, '.')
from unclerenderer_amd import hostmath, synth, assets
from unclerenderer_amd.hotpath import HotPath, to_device
from pathlib import Path
hp = HotPath(0)
A = Path('tests/golden/assets')
env, base, mips, bad = assets.load_env_cube_dds(A / "output_pmrem.dds")
lut = assets.load_brdf_lut_dds(A / "PreintegratedGF.dds")
for (w, h, gen) in [(1920, 1080, 'iid'), (1920, 1080, 'scene'), (256, 144, 'iid')]:
    fc = hostmath.build_frame_constants("sponza", w, h, shadow_size=2048, env_mip_count=9)
    g = synth.gbuffer_iid(w, h, synth.SEED_BASE + 2) if gen == 'iid' else synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, w, h, synth.SEED_BASE + 2)
    shadow = synth.shadow_map_noise(2048, synth.SEED_BASE + 2)
    tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, base, mips), base, mips, to_device(lut))
    dA, dB, dC, dD = to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth)
    two, fused, two_b = to_device(g.hdr), to_device(g.hdr), to_device(g.hdr)
    hp.deferred_lighting(fc.scene, dA, dB, dC, tables, two, w, h)
    lit_only = two.clone()
    hp.sky_atmosphere(fc.sky, dD, two, w, h)
    hp.deferred_lighting_sky(fc.scene, fc.sky, dA, dB, dC, dD, tables, fused, w, h)
    hp.deferred_lighting(fc.scene, dA, dB, dC, tables, two_b, w, h)
    torch.cuda.synchronize()
    a = two.cpu().numpy().view(np.uint16); b = fused.cpu().numpy().view(np.uint16)
    print(gen, w, h, 'lighting repeatable:', torch.equal(lit_only, two_b))
    d = (a != b).any(-1)
    print('  differing pixels', int(d.sum()), 'of', d.size, ' on sky:', int((d & (g.depth == 0)).sum()), ' on geometry:', int((d & (g.depth > 0)).sum()))
    if d.any():
        ys, xs = np.nonzero(d)
        fa = a.view(np.float16).astype(np.float32); fb = b.view(np.float16).astype(np.float32)
        diff = np.abs(fa - fb)[d]
        print('  max abs diff', np.nanmax(diff), ' NaN in either', int(np.isnan(fa[d]).any(-1).sum()), int(np.isnan(fb[d]).any(-1).sum()))
        for k in range(min(6, len(ys))):
            y, x = ys[k], xs[k]
            print('   ', y, x, 'depth', g.depth[y, x], 'two', fa[y, x], 'fused', fb[y, x], 'tile', x // 16, y // 4)
        print('  rows hist (y%4):', np.bincount(ys % 4, minlength=4), ' cols (x%16):', np.bincount(xs % 16, minlength=16))
