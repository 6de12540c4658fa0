#!/usr/bin/env python3
"""Where does the end of a Lighting launch go? A diagnostic build of the product kernel in which EVERY wave stamps the
constant 100 MHz clock and its XCC id when it enters the kernel and when it leaves the tile loop (the product stamps one
{first entry, last exit} pair per launch: ur_debug_timeline). The stamps land behind the launch's pair in the timeline buffer.

    python tools/wave_exit_stamps.py --build                 # here (no GPU): csrc/_build/variants/libur_wavestamps.so
    python tools/wave_exit_stamps.py --run [--width 3840 --height 2160] [--launches 5] [--cache DIR]      # on the GPU box

Printed per launch: the launch's span, the spread of wave exits inside a workgroup (the dynamic claim's quantisation: one
iteration), the spread of the workgroups' LAST exits (imbalance between workgroups), both split by XCD (workgroups of one XCD
share an L2), and what perfect balance at three scopes (workgroup / XCD / chip) would end the launch at."""
import argparse
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
WAVES = 16
MAXG = 256

ENTRY_OLD = "    ur::timeline_entry(p.timeline);\n    if (blockIdx.x >= p.hot.groups) {"
ENTRY_NEW = ("    ur::timeline_entry(p.timeline);\n"
             "    if (p.timeline != nullptr && (threadIdx.x & 63u) == 0u && blockIdx.x < %dU)\n"
             "        { p.timeline[2u + %du + blockIdx.x * %du + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime();\n"
             "          p.timeline[2u + %du + blockIdx.x * %du + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); }\n"
             "    if (blockIdx.x >= p.hot.groups) {") % (MAXG, MAXG * WAVES, WAVES, 3 * MAXG * WAVES, WAVES)
EXIT_OLD = "    unsigned long long* const tl = fresh_params()->timeline;\n"
EXIT_NEW = (EXIT_OLD +
            "    if (tl != nullptr && lane == 0 && blockIdx.x < %dU)\n"
            "        { tl[2u + blockIdx.x * %du + wave] = (__builtin_amdgcn_s_memrealtime() << 4) | (unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u);\n"
            "          tl[2u + %du + blockIdx.x * %du + wave] = __builtin_amdgcn_s_memtime(); }\n") % (MAXG, WAVES, 2 * MAXG * WAVES, WAVES)


CLAIMS_OLD = "        ur::timeline_exit(tl, lane == 0 && left == WPB - 1u);\n"
CLAIMS_NEW = (CLAIMS_OLD +
              "        if (lane == 0 && left == WPB - 1u && blockIdx.x < %dU) tl[2u + %du + blockIdx.x] = work[0];\n") % (MAXG, 4 * MAXG * WAVES)


def build():
    from unclerenderer_amd import build as b
    b.build()
    src = (b.CSRC / "lighting.hip").read_text()
    assert src.count(ENTRY_OLD) == 1 and src.count(EXIT_OLD) == 1 and src.count(CLAIMS_OLD) == 1, "the product source moved: update the anchors"
    tmp = b.CSRC / "_lighting_wavestamps.hip"
    tmp.write_text(src.replace(ENTRY_OLD, ENTRY_NEW).replace(EXIT_OLD, EXIT_NEW).replace(CLAIMS_OLD, CLAIMS_NEW))
    out = b.OUT / "variants"
    out.mkdir(parents=True, exist_ok=True)
    obj, lib = out / "lighting_wavestamps.o", out / "libur_wavestamps.so"
    try:
        subprocess.run([b.hipcc()] + b.COMMON + dict(b.SOURCES)["lighting.hip"] + ["-c", str(tmp), "-o", str(obj)], check=True)
        objs = [str(b.OUT / (s.replace("/", "_") + ".o")) for s, _ in b.SOURCES if s != "lighting.hip"] + [str(obj)]
        subprocess.run([b.hipcc(), f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", str(lib)] + objs + ["-ldl", "-lpthread"], check=True)
    finally:
        tmp.unlink()
        if obj.exists():
            obj.unlink()
    print(lib)


def run(a):
    lib = ROOT / "unclerenderer_amd" / "csrc" / "_build" / "variants" / "libur_wavestamps.so"
    assert lib.exists(), "build it first (--build)"
    os.environ["UR_HOTPATH_LIB"] = str(lib)
    import torch
    from unclerenderer_amd import assets, hostmath, synth
    from unclerenderer_amd.hotpath import HotPath, HzbLayout, to_device
    hp = HotPath(0)
    hp.set_option(8, a.balance)  # UR_OPT_LIGHTING_BALANCE
    W, H = a.width, a.height
    fc = hostmath.build_frame_constants("sponza", W, H)
    ad = ROOT / "tests" / "golden" / "assets"
    env = assets.load_env_cube_dds(ad / "output_pmrem.dds")[0]
    lut = assets.load_brdf_lut_dds(ad / "PreintegratedGF.dds")
    d_env = hp.stage_env_cube(env, 256, 9)
    cache = Path(a.cache) / f"g_scene_{W}x{H}.npz" if a.cache else None
    if cache is not None and cache.exists():
        z = np.load(cache)
        g = synth.GBuffer(W, H, 0, H, z["A"], z["B"], z["C"], z["hdr"], z["depth"])
        shadow = z["shadow"]
    else:
        g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, 3)
        shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
        if cache is not None:
            cache.parent.mkdir(parents=True, exist_ok=True)
            np.savez(cache, A=g.A, B=g.B, C=g.C, hdr=g.hdr, depth=g.depth, shadow=shadow)
    tables = hp.make_tables(to_device(shadow), d_env, 256, 9, to_device(lut))
    sets = [dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth), hdr=to_device(g.hdr)) for _ in range(4)]
    lay = HzbLayout(W, H)
    hzb = torch.zeros(lay.total, device="cuda")
    if a.ride:
        hp.defer_hzb_tail(2)

    def launch(k):
        s = sets[k % 4]
        if a.ride:
            hp.build_hzb(s["D"], hzb, lay)
        hp.deferred_lighting_sky(fc.scene, fc.sky, s["A"], s["B"], s["C"], s["D"], tables, s["hdr"], W, H)

    for k in range(300):  # the chip's clock ramps over the first milliseconds
        launch(k)
    torch.cuda.synchronize()
    n = 1 + 2 * MAXG * WAVES + MAXG // 2  # (+ the workgroups' claim counts) pair 0 = the launch's own {entry, exit}; then exits, entries (100 MHz clock), exits, entries (s_memtime)
    history = []  # per launch: each workgroup's mean wave exit minus the launch's mean (what a static re-deal could take out)
    for rep in range(a.launches):
        tl = torch.zeros((n, 2), dtype=torch.int64, device="cuda")
        tl[0, 0] = -1
        for k in range(8):
            launch(rep * 9 + k)
        hp.debug_timeline(tl[:1])  # capacity ONE pair: the next launch takes it, the variant's stamps land behind it
        launch(rep * 9 + 8)
        torch.cuda.synchronize()
        hp.debug_timeline(None)
        raw = tl.cpu().numpy().view(np.uint64).reshape(-1)
        t_in, t_out = int(raw[0]), int(raw[1])
        ex = raw[2:2 + MAXG * WAVES].reshape(MAXG, WAVES)
        en = raw[2 + MAXG * WAVES:2 + 2 * MAXG * WAVES].reshape(MAXG, WAVES)
        mex = raw[2 + 2 * MAXG * WAVES:2 + 3 * MAXG * WAVES].reshape(MAXG, WAVES)
        men = raw[2 + 3 * MAXG * WAVES:2 + 4 * MAXG * WAVES].reshape(MAXG, WAVES)
        claims = raw[2 + 4 * MAXG * WAVES:2 + 4 * MAXG * WAVES + MAXG].astype(np.int64)
        used = (ex != 0).any(axis=1)
        G = int(used.sum())
        xcc = (ex[used] & np.uint64(15)).astype(np.int64)
        assert (xcc == xcc[:, :1]).all(), "a workgroup lives on one XCD"
        xcc = xcc[:, 0]
        e = ((ex[used] >> np.uint64(4)).astype(np.int64) - t_in) * 0.01  # us since the launch's first entry
        s = (en[used].astype(np.int64) - t_in) * 0.01
        wg_last, wg_first, wg_mean = e.max(axis=1), e.min(axis=1), e.mean(axis=1)
        print(f"launch {rep}: span {0.01 * (t_out - t_in):.2f} us, {G} lighting workgroups on XCDs {sorted(set(xcc.tolist()))} ({np.bincount(xcc).tolist()} each); "
              f"wave entries {s.min():.2f}..{s.max():.2f} us (workgroup medians {np.median(s, axis=1).min():.2f}..{np.median(s, axis=1).max():.2f})")
        print(f"   wave exits: first {e.min():.2f}, mean {e.mean():.2f}, last {e.max():.2f} us; inside a workgroup last - first: median {np.median(wg_last - wg_first):.2f}, max {(wg_last - wg_first).max():.2f} us")
        print(f"   workgroups' LAST exits: min {wg_last.min():.2f}, median {np.median(wg_last):.2f}, max {wg_last.max():.2f}; workgroup MEAN exits: min {wg_mean.min():.2f} max {wg_mean.max():.2f} (sd {wg_mean.std():.2f})")
        rows = []
        for x in sorted(set(xcc.tolist())):
            m = xcc == x
            rows.append((x, int(m.sum()), e[m].mean(), wg_last[m].max(), wg_mean[m].min(), wg_mean[m].max()))
        print("   per XCD (id, workgroups, mean wave exit, last exit, slowest/fastest workgroup mean): " + "; ".join(f"{x}: {c} {me:.2f} {la:.2f} [{lo:.2f},{hi:.2f}]" for x, c, me, la, lo, hi in rows))
        xm = np.array([r[2] for r in rows])
        cl = claims[used]
        print(f"   tiles drawn per workgroup (LDS claim counter at exit; schedule {hp.lighting_schedule()}): min {cl.min()} median {int(np.median(cl))} max {cl.max()}; per XCD mean: "
              + " ".join(f"{x}: {cl[xcc == x].mean():.1f}" for x in sorted(set(xcc.tolist()))))
        # s_memtime ticks per 100 MHz tick over a wave's life, by XCD: do the XCDs run at one clock?
        dt_real = ((ex[used] >> np.uint64(4)).astype(np.int64) - en[used].astype(np.int64)).astype(np.float64)
        dt_mem = (mex[used].astype(np.int64) - men[used].astype(np.int64)).astype(np.float64)
        ratio = dt_mem / np.maximum(dt_real, 1.0)
        print("   s_memtime ticks per 100 MHz tick over a wave's life, per XCD: " + " ".join(f"{x}: {ratio[xcc == x].mean():.3f}" for x in sorted(set(xcc.tolist()))))
        history.append(wg_mean - e.mean())
        print(f"   if waves could be balanced ... inside a workgroup: launch ends at {wg_mean.max():.2f}; inside an XCD: {xm.max():.2f}; over the chip: {e.mean():.2f} (now {e.max():.2f}); "
              f"spread of XCD means {xm.max() - xm.min():.2f} us")
    if len(history) >= 2:
        h = np.array(history)
        cc = np.corrcoef(h)
        print("persistence of the per-workgroup deviations (mean wave exit of a workgroup minus the launch's mean), launch to launch:")
        print("   correlation between consecutive launches: " + " ".join(f"{cc[i, i + 1]:.2f}" for i in range(len(h) - 1)) + f"; sd of a deviation {h.std(axis=1).mean():.2f} us")
        resid = h[1:] - h[:-1]
        avg = h[:-1].cumsum(axis=0) / np.arange(1, len(h))[:, None]
        resid_avg = h[1:] - avg
        print(f"   sd left after taking out the previous launch's deviation: {resid.std(axis=1).mean():.2f} us; the running mean of all previous launches: {resid_avg.std(axis=1)[-1]:.2f} us (last launch)")
        print(f"   slowest workgroup above the mean: now {h.max(axis=1).mean():.2f} us; after the previous launch's correction {resid.max(axis=1).mean():.2f} us; after the running mean's {resid_avg.max(axis=1)[-1]:.2f} us")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--launches", type=int, default=5)
    ap.add_argument("--ride", action="store_true", help="carry the Build HZB chain in the launch (the bench default)")
    ap.add_argument("--cache", default="")
    ap.add_argument("--balance", type=int, default=1, help="UR_OPT_LIGHTING_BALANCE for the run")
    a = ap.parse_args()
    if a.build:
        build()
    if a.run:
        t0 = time.time()
        run(a)
        print(f"({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
