"""Feasibility probe: does the frame's small cull launch overlap the persistent Lighting launch (whole Build HZB chain riding)
when it is issued on a second HIP stream? NO synchronisation between the two streams here (the cull races with the chain
that rewrites the HZB it reads: timing only, results are ignored). us per frame: lighting+hzb alone / + cull on the same
stream / + cull on a second stream."""
import gc, os, sys, time, numpy as np, torch
sys.path.insert(0, '.')
from unclerenderer_amd import hostmath, synth, assets
from unclerenderer_amd.hotpath import HotPath, HzbLayout, to_device
from pathlib import Path
W, H = 3840, 2160
sA = torch.cuda.Stream()
sB = torch.cuda.Stream(priority=int(os.environ.get("PRIO", "-1")))
hp, hpB = HotPath(0, sA), HotPath(0, sB)
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
n_inst = 25
bounds = to_device(synth.instances_replicated(np.array([-20.0, -2.0, -12.0], np.float32), np.array([20.0, 16.0, 12.0], np.float32), n_inst))
d_args = to_device(synth.indirect_args_initial(n_inst))
d_vis = torch.zeros(n_inst, dtype=torch.int32, device="cuda"); d_cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
consts = hostmath.pack_culling_constants(fc.view, fc.proj, n_inst, True, lay.count, lay.width, lay.height, False)
hp.defer_hzb_tail(2)
torch.cuda.synchronize()


def frame(k, cull):
    s = sets[k % ring]
    if cull == 1:
        hp.cull_indirect_args(consts, bounds, hzb, lay, d_args, None, d_vis, d_cnt)
    elif cull == 2:
        hpB.cull_indirect_args(consts, bounds, hzb, lay, d_args, None, d_vis, d_cnt)
    hp.build_hzb(s["D"], hzb, lay)
    hp.deferred_lighting_sky(fc.scene, fc.sky, s["A"], s["B"], s["C"], s["D"], tables, s["hdr"], W, H, 0, H)


def run(cull, n=3000, warm=1500):
    gc.collect(); gc.disable()
    for k in range(warm): frame(k, cull)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n): frame(k, cull)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e6
    gc.enable()
    return dt


for rep in range(2):
    for cull, name in ((0, "lighting + riding hzb, no cull"), (1, "cull on the same stream"), (2, "cull on a second stream (unsynchronised)")):
        print("%-44s %.2f us/frame" % (name, run(cull)), flush=True)
hp.defer_hzb_tail(0)
torch.cuda.synchronize()
