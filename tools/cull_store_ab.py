#!/usr/bin/env python3
"""Cull of 1 M instances against an 8K chain under the UR_OPT_CULL_STORE flavours, over warm (same buffers) and cold (four sets cycled)
inputs, with the command buffer as the previous frame left it ("coherent": no word changes) or reset to InstanceCount = 1 in front of
every launch ("reset": every culled instance's word changes). us per call, batches between one event pair.

    python tools/cull_store_ab.py [--instances 1000000]
"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=1_000_000)
    ap.add_argument("--words-only", action="store_true", help="no visible list (one launch)")
    a = ap.parse_args()
    import torch
    from unclerenderer_amd import hostmath, lib, synth
    from unclerenderer_amd.hotpath import HotPath, HzbLayout, to_device

    hp = HotPath(0)
    n = a.instances
    W8, H8 = 7680, 4320
    fc = hostmath.build_frame_constants("sponza", W8, H8)
    lay = HzbLayout(W8, H8)
    g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W8, H8, synth.SEED_BASE + 5)
    hzb = torch.zeros(lay.total, dtype=torch.float32, device="cuda")
    hp.build_hzb(to_device(g.depth), hzb, lay)
    torch.cuda.synchronize()
    del g
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, n, True, lay.count, lay.width, lay.height, False)
    bounds0 = to_device(synth.instances_random(n, synth.SEED_BASE + 5, center=fc.camera_position, box=400.0))
    args0 = to_device(synth.indirect_args_initial(n))
    ring = 4
    sets = [(bounds0.clone(), args0.clone(), torch.zeros(n, dtype=torch.int32, device="cuda")) for _ in range(ring)]
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")

    one_buffer = [False]  # flavour 4 keeps its record per command buffer: the bounds cycle, the commands do not (they are not read at all)

    def call(k):
        b, ar, v = sets[k]
        if one_buffer[0]:
            ar = sets[0][1]
        hp.cull_indirect_args(consts, b, hzb, lay, ar, None, None if a.words_only else v, None if a.words_only else cnt)

    def timed(cold, reset, batch=16, reps=9):
        out = []
        t = 0
        for _ in range(3):
            call(0)
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if reset:  # per-call events: the reset copy lies outside each pair
                tot = 0.0
                evs = []
                for _ in range(batch):
                    k = (t % ring) if cold else 0
                    t += 1
                    sets[k][1].copy_(args0)
                    x, y = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    x.record(); call(k); y.record()
                    evs.append((x, y))
                torch.cuda.synchronize()
                out.append(float(np.median([x.elapsed_time(y) for x, y in evs])) * 1e3)
            else:
                e0.record()
                for _ in range(batch):
                    call((t % ring) if cold else 0)
                    t += 1
                e1.record()
                torch.cuda.synchronize()
                out.append(e0.elapsed_time(e1) * 1e3 / batch)
        return float(np.median(out))

    ref = None
    print(f"{n} instances, {'words only' if a.words_only else 'words + visible list'}; us per call")
    print("flavour                          warm coherent   cold coherent   warm reset*   cold reset*     (* per-call event pairs: +~3 us)")
    for fl, name in ((0, "plain"), (1, "nontemporal"), (2, "write-through, every word"), (3, "write-through, changed words"),
                     (4, "... from the context's record")):
        hp.set_option(lib.UR_OPT_CULL_STORE, fl)
        one_buffer[0] = fl == 4
        for k in range(ring):
            sets[k][1].copy_(args0)
            call(k)
        torch.cuda.synchronize()
        words = sets[0][1].cpu().numpy().view(np.uint32).copy()
        if ref is None:
            ref = words
        assert np.array_equal(ref, words), "flavours disagree"
        if fl == 4:  # (a reset by somebody else is what the option's contract excludes)
            print(f"{fl} {name:30s} {timed(False, False):10.2f} {timed(True, False):15.2f}             -             -")
        else:
            print(f"{fl} {name:30s} {timed(False, False):10.2f} {timed(True, False):15.2f} {timed(False, True):13.2f} {timed(True, True):13.2f}")


if __name__ == "__main__":
    main()
