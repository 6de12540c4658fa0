#!/usr/bin/env python3
"""Is the chip holding its clock down under the Lighting launch? Samples the GPU's power / clock sensors (amdgpu hwmon + pp_dpm
files in sysfs, read-only; `rocm-smi` as a fallback) from a side thread while the 4K fused Lighting launch runs back to back for a
few seconds, and prints what they read beside the launch time of each second.

    python tools/power_probe.py [--seconds 4] [--balance 1]"""
import argparse
import glob
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def read(path):
    try:
        return Path(path).read_text().strip()
    except Exception:
        return None


def sensors():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
            v = read(f"{hw}/{name}")
            if v is not None:
                out[f"{hw.split('/')[4]}:{name}"] = v
    for dev in glob.glob("/sys/class/drm/card*/device"):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "gpu_busy_percent"):
            v = read(f"{dev}/{name}")
            if v is not None:
                out[f"{dev.split('/')[4]}:{name}"] = " | ".join(l for l in v.splitlines() if "*" in l) or v[:60]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--balance", type=int, default=1)
    ap.add_argument("--no-shadows", action="store_true", help="ShadowStrength = 0: the SHADOWS = false instantiation (no shadow gathers, no PCF)")
    ap.add_argument("--gbuffer", choices=["scene", "iid"], default="scene")
    ap.add_argument("--kernel", choices=["lighting", "stream"], default="lighting", help="stream: the plain four-reads-one-write streaming kernel of the same byte count (ur_debug_stream_ceiling)")
    a = ap.parse_args()
    import torch
    from unclerenderer_amd import assets, hostmath, synth
    from unclerenderer_amd.hotpath import HotPath, to_device
    print("idle sensors:", sensors(), flush=True)
    hp = HotPath(0)
    hp.set_option(8, a.balance)
    W, H = 3840, 2160
    fc = hostmath.build_frame_constants("sponza", W, H)
    if a.no_shadows:
        fc.scene.ShadowStrength = 0.0
    if a.gbuffer == "scene":
        g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, synth.SEED_BASE + 3)
        shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    else:
        g = synth.gbuffer_iid(W, H, synth.SEED_BASE + 3)
        shadow = synth.shadow_map_noise(2048, synth.SEED_BASE + 3)
    ad = ROOT / "tests" / "golden" / "assets"
    env = assets.load_env_cube_dds(ad / "output_pmrem.dds")[0]
    lut = assets.load_brdf_lut_dds(ad / "PreintegratedGF.dds")
    tables = hp.make_tables(to_device(shadow), hp.stage_env_cube(env, 256, 9), 256, 9, to_device(lut))
    bufs = [dict(A=to_device(g.A), B=to_device(g.B), C=to_device(g.C), D=to_device(g.depth), hdr=to_device(g.hdr)) for _ in range(4)]
    n16 = 345_000_000 // 80
    stream_sets = None
    if a.kernel == "stream":
        gen = torch.Generator(device="cuda"); gen.manual_seed(1)
        stream_sets = [([(torch.randint(0, 0x3FFF, (n16 * 8,), dtype=torch.int16, device="cuda", generator=gen) | 0x3000) for _ in range(4)],
                        torch.empty(n16 * 8, dtype=torch.int16, device="cuda")) for _ in range(3)]
    stop = threading.Event()
    samples = []

    def poll():
        while not stop.is_set():
            samples.append((time.perf_counter(), sensors()))
            time.sleep(0.1)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < a.seconds:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(2000):
            if stream_sets is not None:
                hp.stream_ceiling(*stream_sets[k % 3])
                continue
            b = bufs[k % 4]
            hp.deferred_lighting_sky(fc.scene, fc.sky, b["A"], b["B"], b["C"], b["D"], tables, b["hdr"], W, H)
        e1.record()
        torch.cuda.synchronize()
        print(f"t={time.perf_counter() - t0:5.2f}s  {e0.elapsed_time(e1) * 1e3 / 2000:.2f} us per launch", flush=True)
    stop.set()
    th.join()
    # only the card that is busy with this process (the box's other cards show in sysfs too)
    busy = {k.split(":")[0] for _, s in samples for k, v in s.items() if k.endswith("gpu_busy_percent") and v.isdigit() and int(v) > 50}
    keys = sorted({k for _, s in samples for k in s if k.split(":")[0] in busy})
    for k in keys:
        vals = [s.get(k) for _, s in samples]
        print(k, "->", vals[:: max(1, len(vals) // 12)])
        if k.endswith(("power1_input", "freq1_input")):
            nums = [float(v) for v in vals[len(vals) // 3:] if v and v.isdigit()]  # the steady part
            if nums:
                print(f"   steady mean {np.mean(nums) / 1e6:.0f} {'W' if 'power' in k else 'MHz'}")
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20)
        print(r.stdout[-1500:])
    except Exception as e:
        print("rocm-smi:", e)


if __name__ == "__main__":
    main()
