#!/usr/bin/env python3
"""Where does one iteration of the streaming lighting kernel spend its cycles? (diagnostic build with in-kernel stamps)

    python tools/build_variants.py --base r02 stamps=-DUR_STAMPS   # on the build host (the round-2 kernel source carries the stamps)
    python tools/stamps_lighting.py                               # on the GPU box
Reads SHARES, not lengths: the stamps' fences forbid overlaps the real kernel has.
"""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    from unclerenderer_amd import hostmath, synth, lib
    from unclerenderer_amd.hotpath import HotPath, to_device
    hp = HotPath(0)
    W, H = 3840, 2160
    fc = hostmath.build_frame_constants("sponza", W, H)
    from unclerenderer_amd import assets
    asset_dir = Path(__file__).resolve().parent.parent / "tests" / "golden" / "assets"
    env = assets.load_env_cube_dds(asset_dir / "output_pmrem.dds")[0]
    lut = assets.load_brdf_lut_dds(asset_dir / "PreintegratedGF.dds")
    cache = Path("/tmp/urcache/g_scene_3840x2160.npz")  # written by tools/bench_kernels.py --cache /tmp/urcache
    if cache.exists():
        z = np.load(cache)
        g, shadow = synth.GBuffer(W, H, 0, H, z["A"], z["B"], z["C"], z["hdr"], z["depth"]), z["shadow"]
    else:
        g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, 3)
        shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, 256, 9), 256, 9, to_device(lut))
    A, B, C, D, hdr = to_device(g.A), to_device(g.B), to_device(g.C), to_device(g.depth), to_device(g.hdr)
    for _ in range(400):  # sustained clocks: the stamps read back are those of the last launch
        hp.deferred_lighting_sky(fc.scene, fc.sky, A, B, C, D, tables, hdr, W, H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        hp.deferred_lighting_sky(fc.scene, fc.sky, A, B, C, D, tables, hdr, W, H)
    e1.record()
    torch.cuda.synchronize()
    print(f"stamped build: {e0.elapsed_time(e1) * 5.0:.1f} us per launch (includes the stamp buffer's memset)")
    L = lib.load()
    waves = 4096
    out = (ctypes.c_ulonglong * (waves * 16))()
    fn = L.ur_debug_stamps
    fn.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    rc = fn(out, waves)
    assert rc == 0, rc
    raw = np.frombuffer(out, dtype=np.uint64).reshape(waves, 16)
    a = raw.astype(np.float64)
    a[:, 5] = (raw[:, 5] & np.uint64(0xFFFF)).astype(np.float64)
    ticks = (raw[:, 5] >> np.uint64(16)).astype(np.float64)  # whole-kernel span of the wave in 100 MHz ticks
    live = a[:, 7] > 0
    mhz = 100.0 * a[live, 7].sum() / ticks[live].sum()
    print(f"shader clock during the launch: {mhz:.0f} MHz (s_memtime against s_memrealtime)")
    q = np.percentile(a[live, 7], [0, 50, 100])
    print(f"per-wave whole-kernel cycles min/median/max = {q[0]:.0f} {q[1]:.0f} {q[2]:.0f}  ({q[2] / mhz:.1f} us for the slowest)")
    q = np.percentile(a[live, 6], [0, 50, 100])
    print(f"per-wave prologue cycles (entry -> first iteration) min/median/max = {q[0]:.0f} {q[1]:.0f} {q[2]:.0f}  ({q[1] / mhz:.2f} us median)")
    for k, n in ((8, "table loads issued, tile walk set up"), (9, "tile DMAs issued"), (10, "tables converted and in LDS"), (11, "after the barrier"), (6, "tile DMAs landed")):
        q = np.percentile(a[live, k], [0, 50, 100])
        print(f"   prologue, {n:45s} min/median/max = {q[0]:.0f} {q[1]:.0f} {q[2]:.0f}")
    t0 = a[live, 12]
    print(f"   wave entry spread (max - min of t0, counters may differ across XCDs): {t0.max() - t0.min():.0f} cycles")
    a = a[a[:, 5] > 0]
    it = a[:, 5].sum()
    names = ["top -> gathers issued", "LDS lookups + BRDF math", "shadow filter + gather wait", "vmcnt(0) at the prefetch point", "DMA issue + cube filter + combine + store"]
    tot = a[:, :5].sum()
    print(f"{len(a)} waves, {it:.0f} shaded iterations, {tot / it:.0f} cycles per iteration (stamped build)")
    for i, n in enumerate(names):
        print(f"  {n:45s} {a[:, i].sum() / it:8.0f} cycles  {100 * a[:, i].sum() / tot:5.1f} %")
    print(f"  of the last segment: DMA issue (prefetch point)   {a[:, 13].sum() / it:8.0f} cycles; the HDR store (convert + issue) {a[:, 14].sum() / it:8.0f} cycles")
    per_wave = a[:, :5].sum(axis=1)
    q = np.percentile(per_wave, [0, 5, 25, 50, 75, 95, 100])
    print("per-wave loop cycles (sum over its iterations): min/5/25/50/75/95/max =", " ".join(f"{x:.0f}" for x in q))
    per_it = per_wave / a[:, 5]
    q = np.percentile(per_it, [0, 5, 25, 50, 75, 95, 100])
    print("per-wave cycles per iteration:                   min/5/25/50/75/95/max =", " ".join(f"{x:.0f}" for x in q))
    # by workgroup (16 waves each) and by XCD (workgroup index mod 8)
    wg = per_wave[: (len(per_wave) // 16) * 16].reshape(-1, 16).mean(axis=1)
    print("per-workgroup mean loop cycles: min/median/max =", f"{wg.min():.0f} {np.median(wg):.0f} {wg.max():.0f}")
    for x in range(8):
        print(f"   workgroups = {x} mod 8: mean {wg[x::8].mean():.0f}")


if __name__ == "__main__":
    main()
