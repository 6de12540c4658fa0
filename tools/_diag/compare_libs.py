"""sha256 of what the fused Lighting+Sky launch (4K scene G-buffer, shipped tables) and the riding Build HZB chain write, for the library
selected by UR_HOTPATH_LIB (or the product): two libraries that execute the same operations in another order print the same line."""
import hashlib, sys
import numpy as np, torch
sys.path.insert(0, '.')
from pathlib import Path
from unclerenderer_amd import hostmath, synth, assets
from unclerenderer_amd.hotpath import HotPath, HzbLayout, to_device
hp = HotPath(0)
W, H = 3840, 2160
fc = hostmath.build_frame_constants("sponza", W, H)
A = Path('tests/golden/assets')
env = assets.load_env_cube_dds(A / "output_pmrem.dds")[0]
lut = assets.load_brdf_lut_dds(A / "PreintegratedGF.dds")
cache = Path('/tmp/urcache/g_scene_3840x2160.npz')
if cache.exists():
    z = np.load(cache); g = synth.GBuffer(W, H, 0, H, z["A"], z["B"], z["C"], z["hdr"], z["depth"]); shadow = z["shadow"]
else:
    g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, 3)
    shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    cache.parent.mkdir(parents=True, exist_ok=True)
    np.savez(cache, A=g.A, B=g.B, C=g.C, hdr=g.hdr, depth=g.depth, shadow=shadow)
tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, 256, 9), 256, 9, to_device(lut))
d = dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth))
lay = HzbLayout(W, H)
out = []
for ride in (0, 2):
    hdr = to_device(g.hdr)
    hzb = torch.zeros(lay.total, device="cuda")
    hp.defer_hzb_tail(ride)
    hp.build_hzb(d["D"], hzb, lay)
    hp.deferred_lighting_sky(fc.scene, fc.sky, d["A"], d["B"], d["C"], d["D"], tables, hdr, W, H)
    hp.defer_hzb_tail(0)
    torch.cuda.synchronize()
    out.append(hashlib.sha256(hdr.cpu().numpy().tobytes()).hexdigest()[:16])
    out.append(hashlib.sha256(hzb.cpu().numpy().tobytes()).hexdigest()[:16])
print("hdr/hzb separate, hdr/hzb riding:", *out)
