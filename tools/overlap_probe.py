#!/usr/bin/env python3
"""Does the HBM-bound HZB build overlap the VALU-bound lighting kernel when issued on two HIP streams? (development aid)"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from unclerenderer_amd import hostmath, synth
from unclerenderer_amd.hotpath import HotPath, HzbLayout, to_device

W, H = 3840, 2160
sA = torch.cuda.Stream()
sB = torch.cuda.Stream(priority=-1)
hpA, hpB = HotPath(0, sA), HotPath(0, sB)
fc = hostmath.build_frame_constants("sponza", W, H)
g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, 3)
shadow = synth.shadow_map_noise(2048, 3)
tables = hpA.make_tables(to_device(shadow), hpA.stage_env_cube(synth.env_cube_procedural(256, 9), 256, 9), 256, 9, to_device(synth.brdf_lut_procedural()))
A, B, C, D, hdr = (to_device(x) for x in (g.A, g.B, g.C, g.depth, g.hdr))
lay = HzbLayout(W, H)
hzb = torch.zeros(lay.total, device="cuda")
torch.cuda.synchronize()


def light():
    hpA.deferred_lighting_sky(fc.scene, fc.sky, A, B, C, D, tables, hdr, W, H)


def hz():
    hpB.build_hzb(D, hzb, lay)


def run(fa, fb, n=50):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if fa: fa()
        if fb: fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for _ in range(3):
    light(); hz()
print("lighting only      %.1f us" % run(light, None))
print("hzb only           %.1f us" % run(None, hz))
print("both, two streams  %.1f us" % run(light, hz))
