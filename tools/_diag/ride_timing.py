"""Lighting alone / Build HZB + Lighting as separate launches / the whole chain riding with the Lighting launch: us per pair."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from unclerenderer_amd import hostmath, synth, assets
from unclerenderer_amd.hotpath import HotPath, HzbLayout, to_device
from pathlib import Path
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
tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, 256, 9), 256, 9, to_device(lut))
ring = 4
sets = [dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth), hdr=to_device(g.hdr)) for _ in range(ring)]
lay = HzbLayout(W, H)
hzb = torch.zeros(lay.total, device="cuda")
import os
BAND = int(os.environ.get("BAND", H))  # rows of the frame this "rank" shades (the HZB is always the whole frame's)
def light(k):
    s = sets[k % ring]
    hp.deferred_lighting_sky(fc.scene, fc.sky, s["A"][:BAND], s["B"][:BAND], s["C"][:BAND], s["D"][:BAND], tables, s["hdr"][:BAND], W, H, 0, BAND)
def both(k):
    hp.build_hzb(sets[k % ring]["D"], hzb, lay)
    light(k)
def run(fn, n=400, warm=300):
    for k in range(warm): fn(k)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(n): fn(k)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
tag = (sys.argv[1] if len(sys.argv) > 1 else '') + f" band={BAND}"
print(tag, "lighting alone            %.1f us" % run(light))
for mode, name in ((0, "separate launches (3)"), (1, "tail rides (2 launches)"), (2, "whole chain rides (1)")):
    hp.defer_hzb_tail(mode)
    print(tag, "hzb + lighting, %-24s %.1f us" % (name, run(both)))
hp.defer_hzb_tail(0)
ref = torch.zeros(lay.total, device="cuda"); hp.build_hzb(sets[0]["D"], ref, lay); hp.defer_hzb_tail(2); both(0); hp.defer_hzb_tail(0); torch.cuda.synchronize()
print(tag, "ride == separate:", bool(torch.equal(ref, hzb)))
