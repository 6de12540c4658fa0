#!/usr/bin/env python3
"""bench.py — the reference's headline metric on MI355X: shaded Mpixels/s of the 4K deferred frame (BuildHZB +
CullIndirectArgs + DeferredLighting/Sky) on the Sponza constants, with % of HBM roofline for the dominant kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one frame of BASELINE.json configs[2] on synthetic inputs already resident in HBM:
  cull (Sponza's 25 commands, last frame's HZB) -> BuildHZB (full chain) -> DeferredLighting+Sky (fused kernel).
N > 1: the frame is sharded by screen row bands (rank r shades rows [r*H/N,(r+1)*H/N)), the cull by instance ranges, the
HZB build is replicated, and the HDR bands are all-gathered with RCCL (torch.distributed backend "nccl") every frame.
Rank 0 prints ONE JSON line. The CPU oracle is used only for the cpu_baseline leg (rank 0, N == 1).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy reaches


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # ~0.1 ms per frame: 200 warm-up frames let the chip reach its sustained clock (20 ms; with 20 the first timed frames
    # still run ~10 % slower), 1000 timed frames are 0.1 s
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    # 200 frames (16 ms) are not enough: the next 20 frames still run 3-4 % slower than the steady state (84.4 against 80.6-81.4
    # us/frame with 600 and more); 1000 frames are 80 ms of untimed work
    ap.add_argument("--min-warmup", type=int, default=1000, help="lower bound on the untimed warm-up frames (clock ramp)")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--gbuffer", choices=["scene", "iid"], default="scene")
    ap.add_argument("--ring", type=int, default=4, help="distinct frame-buffer sets cycled through so inputs are cache-cold")
    ap.add_argument("--cull-instances", type=int, default=1_000_000)
    ap.add_argument("--async-compute", action="store_true",
                    help="put the visibility passes on the graph's async-compute stream. Off by default: the streaming lighting kernel "
                         "keeps every CU's register file full, so the passes no longer run beside it (measured: same frame time)")
    ap.add_argument("--hzb-launch", choices=["ride", "tail-rides", "separate"], default="ride",
                    help="ride: the whole Build HZB chain rides along with the Lighting launch (two launches per frame); tail-rides: only its "
                         "single-workgroup tail does (three launches); separate: Build HZB is launches of its own (four)")
    ap.add_argument("--hzb", choices=["replicate", "shard"], default="replicate",
                    help="N > 1: replicate = every rank builds the whole HZB chain from the whole depth buffer (no exchange); shard = a rank builds "
                         "mips 0-4 for the 128x32 pieces its rows own (riding its Lighting launch), the slices cross peer to peer and the "
                         "single-workgroup tail runs on every rank behind the exchange (ur_build_hzb_band, dist.allgather_hzb_slices)")
    ap.add_argument("--separate-hzb-tail", action="store_true",
                    help="launch the single-workgroup tail of Build HZB on its own (three visibility launches per frame) instead of "
                         "letting it ride along with the Lighting launch as an extra workgroup")
    ap.add_argument("--explicit-stream", action="store_true", help="run everything on one explicitly created stream instead of the default (null) stream")
    ap.add_argument("--graph", action="store_true",
                    help="capture one frame per buffer set in a HIP graph (torch.cuda.CUDAGraph) and replay it; frames that carry the "
                         "Lighting event pair are still submitted eagerly")
    ap.add_argument("--sync-gather", action="store_true", help="N > 1: wait for each frame's HDR all-gather before the next frame starts")
    ap.add_argument("--gather-ldr", action="store_true",
                    help="add the Tonemap pass behind Lighting+Sky (Tonemap.hlsl) and, with N > 1, all-gather the tonemapped RGBA8 bands "
                         "(4 B/pixel over xGMI) instead of the RGBA16F ones (8 B/pixel)")
    ap.add_argument("--gather", choices=["ring", "direct"], default="ring",
                    help="N > 1: ring = one all_gather_into_tensor per frame (RCCL's ring / tree); direct = N - 1 grouped send/receive pairs "
                         "per rank, one per xGMI link (unclerenderer_amd/dist.py, ur_allgather_rows_bytes_ex)")
    ap.add_argument("--light-every", type=int, default=0, help="sample the Lighting dispatch's duration on one timed frame in this many (0 = 2 for runs of up to 64 steps, 8 beyond)")
    ap.add_argument("--no-light-events", action="store_true", help="do not bracket the Lighting pass with events inside the timed region")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="host pacing: the host waits (once every 64 frames) until the GPU is within this many frames of it (the reference "
                         "keeps 3 frames in flight, Core/Application.cpp:567-573). 0 = unpaced (default): the host runs ~8x ahead of the GPU, "
                         "which is what hides its own pauses")
    ap.add_argument("--no-spin-fence", action="store_true", help="end the timed region with torch.cuda.synchronize() alone instead of spinning on a "
                                                                 "stream-written word of pinned memory in front of it")
    ap.add_argument("--python-gc", action="store_true",
                    help="leave Python's cyclic garbage collector on during the frames. Off by default: a full collection of this process's "
                         "heap takes the submitting thread 35-55 ms at an allocation count that falls inside the timed region, and whenever "
                         "that is longer than the work queued ahead the GPU idles for the difference (the 'slow' bench process, DESIGN.md section 7)")
    ap.add_argument("--timeline", action="store_true",
                    help="GPU-side timeline of the timed region (ur_debug_timeline: each cull / Lighting launch stamps its first entry and "
                         "last exit with the constant 100 MHz clock): per-kernel spans and the gaps between consecutive launches, without a profiler")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the 1M-instance cull and iid side measurements")
    return ap.parse_args()


def _self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N rank processes ourselves, as a CHILD
    `python -m torch.distributed.run`, before this process has imported torch or touched a GPU (a process that has
    initialised HIP must never exec or fork GPU work on this pool). Rank 0's JSON line goes to the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(args))
    import torch
    import torch.distributed as dist

    from unclerenderer_amd import dist as urdist
    from unclerenderer_amd import hostmath, synth
    from unclerenderer_amd.hotpath import HotPath, HzbLayout, to_device

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world  # launched under torch.distributed.run: the launcher's world size is the truth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # UR_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share devices;
    # RCCL refuses that). The measured configuration is always nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("UR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    W, H, N = args.width, args.height, world
    if H % N != 0:
        raise SystemExit(f"height {H} not divisible by {N} ranks")
    band = H // N
    row0 = rank * band
    if args.graph or args.explicit_stream:
        # stream capture is not allowed on the legacy default stream: the whole run uses one explicit stream
        torch.cuda.set_stream(torch.cuda.Stream(device=local_rank))
    hp = HotPath(local_rank)
    dev = local_rank

    # ---------------- inputs (synthetic, generated per rank for its own band; constants from the shipped Sponza scene)
    preset = hostmath.SCENES["sponza"]
    fc = hostmath.build_frame_constants(preset, W, H, shadow_size=2048, env_mip_count=9)
    seed = synth.SEED_BASE + 3
    t_gen = time.time()
    if args.gbuffer == "scene":
        g = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, seed, row0, band)
        shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    else:
        g = synth.gbuffer_iid(W, H, seed, row0, band)
        shadow = synth.shadow_map_noise(2048, seed)
    # HZB build is replicated: every rank needs the full-frame depth
    if N == 1:
        depth_full = g.depth
    elif args.gbuffer == "scene":
        depth_full = synth.gbuffer_scene(fc.view, fc.proj, fc.camera_position, W, H, seed).depth
    else:
        depth_full = synth.gbuffer_iid(W, H, seed).depth
    # IBL tables: the reference's shipped assets (BC6H_SF16 cube decoded once on the host, RG16 LUT), kept as data
    # fixtures under tests/golden/assets; procedural stand-ins of the same shape if they are absent
    asset_dir = ROOT / "tests" / "golden" / "assets"
    if (asset_dir / "output_pmrem.dds").exists() and (asset_dir / "PreintegratedGF.dds").exists():
        from unclerenderer_amd import assets
        env, env_base, env_mips, _ = assets.load_env_cube_dds(asset_dir / "output_pmrem.dds")
        lut = assets.load_brdf_lut_dds(asset_dir / "PreintegratedGF.dds")
        ibl_desc = "shipped output_pmrem.dds (BC6H_SF16 256^2 x 9 mips, decoded at setup) + PreintegratedGF.dds (RG16 128x32)"
    else:
        env, env_base, env_mips = synth.env_cube_procedural(256, 9, sun_dir=fc.light_direction), 256, 9
        lut = synth.brdf_lut_procedural(128, 32)
        ibl_desc = "procedural 256^2x9 cube + analytic LUT"
    assert (env_base, env_mips) == (256, 9)
    gen_s = time.time() - t_gen

    d_env_cube, d_lut = hp.stage_env_cube(env, 256, 9), to_device(lut, dev)
    tables = hp.make_tables(to_device(shadow, dev), d_env_cube, 256, 9, d_lut)
    lay = HzbLayout(W, H)
    ring = max(1, args.ring)
    sets = []
    for _ in range(ring):
        s = {
            "A": to_device(g.A, dev), "B": to_device(g.B, dev), "C": to_device(g.C, dev),
            "depth_full": to_device(depth_full, dev),
            "hdr_full": torch.zeros((H, W, 4), dtype=torch.int16, device=f"cuda:{dev}"),
        }
        s["hdr_full"][row0:row0 + band] = to_device(g.hdr, dev)
        s["depth_band"] = s["depth_full"][row0:row0 + band]
        # N == 1: shade straight into the frame; N > 1: shade a band buffer, RCCL gathers the bands into the frame
        s["hdr_band"] = s["hdr_full"][row0:row0 + band] if N == 1 else to_device(g.hdr, dev)
        if args.gather_ldr:
            s["ldr_full"] = torch.zeros((H, W), dtype=torch.int32, device=f"cuda:{dev}")
            s["ldr_band"] = s["ldr_full"][row0:row0 + band] if N == 1 else torch.zeros((band, W), dtype=torch.int32, device=f"cuda:{dev}")
        sets.append(s)

    # Sponza cull: 25 commands sharing one AABB, sharded by instance range
    n_inst = preset.instance_count
    i0, i1 = rank * n_inst // N, (rank + 1) * n_inst // N
    bounds = synth.instances_replicated(*preset.model_aabb, n_inst)[i0:i1]
    d_bounds = to_device(bounds, dev) if i1 > i0 else None
    d_args = to_device(synth.indirect_args_initial(n_inst)[i0:i1], dev) if i1 > i0 else None
    d_vis = torch.zeros(max(1, i1 - i0), dtype=torch.int32, device=f"cuda:{dev}")
    d_cnt = torch.zeros(1, dtype=torch.int32, device=f"cuda:{dev}")
    cull_consts = hostmath.pack_culling_constants(fc.view, fc.proj, i1 - i0, True, lay.count, lay.width, lay.height, False)

    # The frame is driven through the render graph (csrc/frame/HotPathRenderer.cpp): GPU Culling -> Build HZB -> Lighting
    # (+Sky fused), all on the main stream unless --async-compute moves the two visibility passes to the graph's
    # async-compute stream. The HZB is ONE
    # buffer: the cull of frame k reads what frame k-1 built (DeferredRenderer.cpp:519-542).
    from unclerenderer_amd import lib as urlib
    from unclerenderer_amd.hotpath import Frame
    frame = Frame(hp, frames_in_flight=3, rank=rank, world_size=N)
    hzb = torch.zeros(lay.total, dtype=torch.float32, device=f"cuda:{dev}")
    flags = urlib.UR_FRAME_DEFAULT | urlib.UR_FRAME_FUSE_LIGHTING_SKY
    if args.separate_hzb_tail:
        args.hzb_launch = "separate"
    if args.hzb_launch == "ride":
        flags |= urlib.UR_FRAME_HZB_WITH_LIGHTING  # (ignored with --async-compute)
    elif args.hzb_launch == "tail-rides":
        flags |= urlib.UR_FRAME_HZB_TAIL_WITH_LIGHTING
    if args.gather_ldr:
        flags |= urlib.UR_FRAME_TONEMAP
    shard_hzb = args.hzb == "shard" and N > 1 and not args.async_compute
    if shard_hzb:
        flags |= urlib.UR_FRAME_HZB_SHARD
    hzb_sent = [0]
    if args.async_compute:
        # visibility passes on the async-compute stream; joined once before the timed region closes (nothing on the
        # main stream consumes their outputs or overwrites their inputs inside the loop)
        flags |= urlib.UR_FRAME_ASYNC_COMPUTE | urlib.UR_FRAME_ASYNC_NO_JOIN
    for s in sets:
        s["res"] = Frame.resources(W, H, row0, band, s["A"], s["B"], s["C"], s["depth_band"], s["hdr_band"], s["depth_full"], hzb, lay, tables,
                                   d_bounds, d_args, i1 - i0, i0, d_vis, d_cnt, None, s.get("ldr_band"))

    # roofline leg: the Lighting launch of the sampled timed frames is timed by HIP events carried ON DISPATCHES
    # (hipExtLaunchKernel on the stream the kernels are launched on): the stop event rides on the Lighting kernel's own
    # dispatch, the start event on the cull launch directly in front of it (ur_time_next_cull / ur_time_next_lighting; the
    # frame is exactly those two launches) — their distance is the interval from the end of what precedes the Lighting
    # kernel to the kernel's end as the command processor stamps them, the quantity rocprofv3's kernel trace reports for the
    # dispatch (profiles/: same command), and NOTHING enters the queue for the measurement. (Where no cull launch precedes —
    # other --hzb-launch modes — the start event is a marker in front of the kernel, ~8 us of queue time.) A sampled frame
    # costs ~1.7 us (the two dispatches run with their timestamps enabled): every second frame of a short run (the driver's
    # --steps 20 gives 10 samples), one in eight of a long one.
    timed_flags = flags if args.no_light_events else (flags | urlib.UR_FRAME_TIME_LIGHTING_KERNEL)
    light_every = args.light_every if args.light_every > 0 else (2 if args.steps <= 64 else 8)

    from collections import deque
    pace_marks = deque()
    pace_every = 64
    submitted = [0]

    def pace():
        """Every 64th frame leaves an event behind; the host never runs more than --frames-in-flight frames ahead of it."""
        if args.frames_in_flight <= 0:
            return
        submitted[0] += 1
        if submitted[0] % pace_every == 0:
            e = torch.cuda.Event()
            e.record()
            pace_marks.append(e)
            while len(pace_marks) > max(1, args.frames_in_flight // pace_every):
                pace_marks.popleft().synchronize()

    def step(k: int, timed: bool):
        pace()
        s = sets[k % ring]
        # one frame in eight carries the event pair: an event record costs ~4 us of queue time on this stack, which would
        # otherwise inflate every timed frame by ~6 %
        if s.get("gather") is not None:  # this buffer set's previous all-gather must have finished before it is shaded into again
            s["gather"].wait()
            s["gather"] = None
        if args.graph and "graph" in s and not (timed and k % light_every == 0):
            s["graph"].replay()
        else:
            frame.render(s["res"], cull_consts, fc.scene, fc.sky, timed_flags if (timed and k % light_every == 0) else flags)
        if shard_hzb:
            # the frame built this rank's pieces of mips 0-4: exchange the slices (peer to peer, into place) and run the tail. The
            # next frame's cull reads this HZB, so the exchange is waited for here (it is ~1.4 MB per peer at 4K / 8)
            hzb_sent[0] = urdist.allgather_hzb_slices(hzb, lay)[0]
            hp.build_hzb_tail(hzb, lay)
        if N > 1:
            # RCCL all-gather of the bands on the communication stream, behind this frame's passes; the next frames (other
            # buffer sets of the ring) are shaded while it runs — frames in flight, as the reference keeps three
            if args.gather_ldr:
                s["gather"] = urdist.allgather_rows(s["ldr_full"], s["ldr_band"], async_op=not args.sync_gather, mode=args.gather)
            else:
                s["gather"] = urdist.allgather_hdr(s["hdr_full"], s["hdr_band"], async_op=not args.sync_gather, mode=args.gather)

    # The host notices the end of the queue by SPINNING on a word of pinned memory that the stream writes behind the last frame
    # (hipStreamWriteValue32: a queue packet, no kernel), and only then calls torch.cuda.synchronize(), which finds nothing left to
    # wait for: a blocking synchronize alone adds the runtime's wake-up latency (tens of microseconds) to whatever it brackets, which
    # is 2-3 % of a 20-frame timed region. (--no-spin-fence: the plain synchronize.)
    spin = {"hip": None, "flag": None, "np": None, "seq": 0, "used": 0}
    if not args.no_spin_fence and not args.graph:
        try:
            import ctypes
            spin["hip"] = ctypes.CDLL("libamdhip64.so")
            spin["hip"].hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
            spin["hip"].hipStreamWriteValue32.restype = ctypes.c_int
            spin["flag"] = torch.zeros(16, dtype=torch.int32).pin_memory()
            spin["np"] = spin["flag"].numpy()
        except Exception:
            spin["hip"] = None

    def spin_until_queue_is_empty():
        if spin["hip"] is None:
            return
        spin["seq"] += 1
        rc = spin["hip"].hipStreamWriteValue32(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(spin["flag"].data_ptr()),
                                               ctypes.c_uint32(spin["seq"]), 0)
        if rc != 0:  # not supported on this stack: the plain synchronize does the waiting
            spin["hip"] = None
            return
        word, want, t_end = spin["np"], spin["seq"], time.perf_counter() + 10.0
        while word[0] != want:
            if time.perf_counter() > t_end:
                break
        spin["used"] += 1

    def fence():
        for s in sets:
            if s.get("gather") is not None:
                s["gather"].wait()
                s["gather"] = None
        if args.async_compute:
            frame.join_async()
        spin_until_queue_is_empty()
        torch.cuda.synchronize()
        if N > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # The chip needs ~10-20 ms of continuous work to reach its sustained clock and a 4K frame is ~0.1 ms: never fewer than
    # `--min-warmup` untimed frames, whatever W says (they are warm-up steps like the others; K timed steps follow).
    if args.graph:
        for k in range(2 * ring):  # everything lazy (workspace, function attributes) happens before the capture
            step(k, False)
        fence()
        for s in sets:
            g_ = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_, stream=torch.cuda.current_stream()):
                frame.render(s["res"], cull_consts, fc.scene, fc.sky, flags)
            s["graph"] = g_
    # Untimed frames, in this order: a clock ramp of max(0, min_warmup - W) frames, the host-cost probe's 32, then the W
    # warm-up steps the caller asked for, then the K timed ones. All of them are reported (clock_ramp_frames, warmup).
    import gc
    if not args.python_gc:
        gc.collect()
        gc.freeze()   # everything allocated during set-up is out of the collector's sight
        gc.disable()  # and no collection starts while frames are being submitted (re-enabled behind the timed region)
    ramp_frames = max(0, args.min_warmup - args.warmup)
    for k in range(ramp_frames):
        step(k, False)
    fence()
    # host cost of one frame's submission with an empty queue (no back-pressure): if this approaches ms_per_step the run
    # is submission-bound and the GPU idles between passes
    th = time.perf_counter()
    for k in range(32):
        step(k, False)
    host_unthrottled_ms = (time.perf_counter() - th) / 32 * 1e3
    ramp_frames += 32
    for k in range(args.warmup):
        step(k, False)
    fence()
    tl = None
    if args.timeline:
        tl = torch.empty((4 * args.steps + 64, 2), dtype=torch.int64, device=f"cuda:{dev}")
        tl[:, 0] = -1  # = ~0 as uint64: the entry stamp is an atomic minimum
        tl[:, 1] = 0
        torch.cuda.synchronize()
        hp.debug_timeline(tl)
    host_t = np.zeros(args.steps + 1)
    t0 = time.perf_counter()
    host_t[0] = t0
    for k in range(args.steps):
        step(args.warmup + k, True)
        host_t[k + 1] = time.perf_counter()
    t_enqueued = time.perf_counter() - t0  # host time to submit K frames (if ~= dt the run is submission-bound)
    fence()
    dt = time.perf_counter() - t0
    if not args.python_gc:
        gc.enable()
    timeline = None
    if tl is not None:
        hp.debug_timeline(None)
        pairs = tl.cpu().numpy().view(np.uint64)
        pairs = pairs[pairs[:, 1] != 0]
        if pairs.shape[0] >= 8:
            t_in, t_out = pairs[:, 0].astype(np.float64) * 1e-2, pairs[:, 1].astype(np.float64) * 1e-2  # 100 MHz ticks -> us
            span = t_out - t_in
            gap = t_in[1:] - t_out[:-1]
            big = span > 20.0  # the Lighting launches (the cull of a few commands is microseconds)
            q = lambda a: [round(float(x), 2) for x in np.percentile(a, [50, 90, 100])] if a.size else None
            timeline = {
                "launches": int(pairs.shape[0]), "clock": "s_memrealtime, 100 MHz; [median, p90, max] in us",
                "lighting_span_us": q(span[big]), "other_span_us": q(span[~big]),
                "gap_before_lighting_us": q(gap[big[1:]]), "gap_before_other_us": q(gap[~big[1:]]),
                "period_between_lighting_entries_us": q(np.diff(t_in[big])),
            }
            # the largest GPU-idle gaps, with the frame they precede, beside the host's largest pauses between two submissions:
            # a GPU gap of milliseconds that sits where the host paused is the host thread not being scheduled / blocked in the
            # runtime, not anything the kernels do
            frame_of = np.cumsum(big) - big  # launch index -> frame index (a frame ends with its Lighting launch)
            top = np.argsort(gap)[-3:][::-1]
            timeline["largest_gaps"] = [{"us": round(float(gap[i]), 1), "before_frame": int(frame_of[i + 1])} for i in top]
            hd = np.diff(host_t) * 1e6
            toph = np.argsort(hd)[-3:][::-1]
            timeline["largest_host_pauses"] = [{"us": round(float(hd[i]), 1), "submitting_frame": int(i)} for i in toph]
            timeline["host_submit_us_median"] = round(float(np.median(hd)), 1)
    if N > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{dev}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    light_ms, _ = (a.astype(np.float64) for a in frame.lighting_times_and_record_cost_ms())  # inside the timed region
    # The same frames once more with Python's collector left on (a host that drives frames from Python without freezing it):
    # reported beside `value`, never instead of it
    gc_on = None
    if not args.python_gc and N == 1:
        gc.unfreeze()  # (the timed region above ran with the set-up heap frozen and the collector off)
        t1 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + k, False)
        fence()
        dt_gc = time.perf_counter() - t1
        gc_on = {"value": W * H * args.steps / dt_gc / 1e6, "ms_per_step": dt_gc / args.steps * 1e3}
    # secondary (untimed frames): the Lighting pass between an ordinary event pair (hipEventRecord in front and behind), with a
    # third event recorded right behind each pair: what round 2 reported (bracket, and bracket minus one record's cost)
    for k in range(32):
        s_ = sets[k % ring]
        if s_.get("gather") is not None:
            s_["gather"].wait()
            s_["gather"] = None
        frame.render(s_["res"], cull_consts, fc.scene, fc.sky, flags | urlib.UR_FRAME_TIME_LIGHTING_RECORD_COST)
    fence()
    bracket_ms, record_ms = (a.astype(np.float64) for a in frame.lighting_times_and_record_cost_ms())
    record_ms = record_ms[record_ms >= 0]
    if record_ms.size == 0:
        record_ms = np.zeros(1)
    if bracket_ms.size == 0:
        bracket_ms = np.zeros(1)
    # the same kernel in a loop of its own (no visibility passes around it): ONE event pair around the whole batch, so that
    # no event record sits between two launches (the figure includes the ~2 us boundary between back-to-back launches)
    n_alone = min(max(args.steps, 50), 300)
    for k in range(20):
        s = sets[k % ring]
        hp.deferred_lighting_sky(fc.scene, fc.sky, s["A"], s["B"], s["C"], s["depth_band"], tables, s["hdr_band"], W, H, row0, band)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n_alone):
        s = sets[k % ring]
        hp.deferred_lighting_sky(fc.scene, fc.sky, s["A"], s["B"], s["C"], s["depth_band"], tables, s["hdr_band"], W, H, row0, band)
    e1.record()
    torch.cuda.synchronize()
    alone_ms = np.array([e0.elapsed_time(e1) / n_alone], dtype=np.float64)
    if light_ms.size == 0:
        light_ms = alone_ms
    n_sky = int((g.depth == 0).sum())
    n_geo = g.depth.size - n_sky
    # algorithmic bytes of one fused launch on this rank: geometry pixels read A 8 + B 8 + C 4 + depth 4 + HDR 8 and write
    # HDR 8 (= 40 B); sky pixels read depth 4 and write HDR 8 (= 12 B). Side tables are cache-resident and excluded.
    light_only_bytes = 40 * n_geo + 12 * n_sky
    # --hzb-launch ride: the same launch also reads the full depth buffer once and writes every HZB mip (Build HZB's algorithmic
    # bytes): they are part of what THIS launch moves
    rides = args.hzb_launch == "ride" and not args.async_compute
    hzb_bytes = 4 * (W * H + lay.mip_texels())
    light_bytes = light_only_bytes + (hzb_bytes if rides else 0)
    # ... counting the depth buffer ONCE: the riding chain's read of it is the launch's second (the tile DMA has fetched the
    # same rows for the sky test), so the launch's compulsory bytes are smaller by one depth buffer
    light_bytes_dedup = light_bytes - (4 * W * band if rides else 0)  # (at N > 1: the rows of this rank's band)
    light_avg_s = float(light_ms.mean()) * 1e-3        # dispatch begin -> end, averaged over the samples of the timed region
    bracket_avg_s = float(bracket_ms.mean()) * 1e-3    # secondary: [record, launch, record] of the untimed frames behind it
    record_avg_s = float(record_ms.mean()) * 1e-3
    achieved = light_bytes / light_avg_s / 1e9

    result = {
        "metric": "shaded Mpixels/s, 4K deferred pass (BuildHZB + CullIndirectArgs + DeferredLighting/Sky), Sponza; % HBM roofline",
        "value": W * H * args.steps / dt / 1e6,
        "unit": "Mpixels/s",
        "n_gpus": N,
        "steps": args.steps,
        "warmup": args.warmup,  # the W warm-up steps, run immediately before the timed region ...
        "clock_ramp_frames": ramp_frames,  # ... and the untimed frames run BEFORE those (clock ramp to --min-warmup, 32 for the host-cost probe)
        "untimed_frames_total": ramp_frames + args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "host_submit_ms_per_step": t_enqueued / args.steps * 1e3,
        "end_of_region": ("spin on a stream-written pinned word, then torch.cuda.synchronize()" if spin["used"] else "torch.cuda.synchronize()"),
        "host_submit_unthrottled_ms_per_step": host_unthrottled_ms,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",  # the arithmetic type of the path (G-buffer and HDR targets are fp16 / unorm8 in memory: config.storage)
        "data": "synthetic",
        "config": {
            "workload": f"Sponza {W}x{H} full pipeline: cull(25) + BuildHZB({lay.count} mips) + DeferredLighting+Sky fused"
                        + (" + Tonemap" if args.gather_ldr else "")
                        + (f", {N} row bands + RCCL {'all-gather' if args.gather == 'ring' else 'direct send/recv gather'} of " + ("tonemapped RGBA8" if args.gather_ldr else "RGBA16F HDR") + (" (overlapped with the next frames)" if not args.sync_gather else "") if N > 1 else ""),
            "storage": "G-buffer A/B and HDR RGBA16F, G-buffer C RGBA8 sRGB, depth / shadow map / HZB fp32",
            "gbuffer": args.gbuffer, "background_fraction": round(float(n_sky) / g.depth.size, 4),
            "ibl_tables": ibl_desc,
            "frame_buffer_ring": ring, "parallelism": f"rowbands{N}",
            "hzb_build": ("band-sharded: mips 0-4 per rank + peer-to-peer exchange of %d B per peer + replicated tail" % hzb_sent[0]) if shard_hzb else ("replicated on every rank" if N > 1 else "one rank"),
            "async_compute": args.async_compute, "hip_graph": bool(args.graph), "hzb_launch": "separate launches" if (args.hzb_launch == "separate" or args.async_compute) else ("whole chain rides with the Lighting launch" if args.hzb_launch == "ride" else "tail rides with the Lighting launch"), "driver": "FRenderGraph (csrc/frame/HotPathRenderer.cpp)", "frames_in_flight": args.frames_in_flight,
        },
        "roofline": {
            "kernel": "lighting_stream_kernel<FUSED>" + (" carrying the Build HZB chain (wave pieces + tail workgroup)" if rides else ""), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            # the same launch against its COMPULSORY bytes (the depth buffer counted once) and against what a plain streaming kernel of
            # this byte count sustains on this box in this run (stream_ceiling_leg: four reads, one write, nothing computed)
            "frac_dedup": light_bytes_dedup / light_avg_s / 1e9 / HBM_PEAK_GBS, "bytes_per_launch_dedup": light_bytes_dedup,
            "stream_ceiling_GBps": None, "frac_of_stream_ceiling": None, "stream_ceiling_us": None,
            "bytes_per_launch": light_bytes, "avg_launch_us": light_avg_s * 1e6, "min_launch_us": float(light_ms.min()) * 1e3,
            "median_launch_us": float(np.median(light_ms)) * 1e3,
            "timing": "HIP events carried on dispatches (hipExtLaunchKernel): end of the cull launch in front -> end of the Lighting launch, timed region",
            "lighting_bytes": light_only_bytes, "hzb_bytes_in_launch": hzb_bytes if rides else 0,
            "frac_lighting_bytes_only": light_only_bytes / light_avg_s / 1e9 / HBM_PEAK_GBS,
            "shade_only_mpixels_per_s": g.depth.size * N / light_avg_s / 1e6,
            "launches_sampled": int(light_ms.size), "alone_on_stream_us": float(alone_ms.mean()) * 1e3,
            "alone_on_stream_frac": light_only_bytes / (float(alone_ms.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "event_bracket_us": bracket_avg_s * 1e6, "event_record_us": record_avg_s * 1e6,
            "frac_event_bracket": light_bytes / bracket_avg_s / 1e9 / HBM_PEAK_GBS if bracket_avg_s > 0 else None,
            "frac_event_bracket_minus_record": light_bytes / (bracket_avg_s - record_avg_s) / 1e9 / HBM_PEAK_GBS if bracket_avg_s > record_avg_s else None,
            "schedule": hp.lighting_schedule(),
        },
    }
    if rank == 0 and not args.no_extras:
        # the practical ceiling, on this box, in this run, at this launch's byte count (untimed leg)
        sc = stream_ceiling_leg(hp, torch, light_bytes, dev)
        r_ = result["roofline"]
        r_["stream_ceiling_GBps"], r_["stream_ceiling_us"] = sc["GBps"], sc["dispatch_us"]
        r_["frac_of_stream_ceiling"] = achieved / sc["GBps"]
        r_["stream_ceiling"] = sc
    traffic_file = ROOT / "profiles" / "traffic_latest.json"
    if traffic_file.exists():
        try:
            result["roofline"]["traffic"] = json.loads(traffic_file.read_text()).get("lighting_kernel_fused_bytes_per_launch")
        except Exception:
            pass

    if rank == 0 and N == 1 and not args.no_extras:
        result["extras"] = ex = side_measurements(args, hp, fc, torch, to_device, HzbLayout, synth, hostmath, dev, (d_env_cube, d_lut))
        # the same kernel on SURVEY.md §8d's own generator travels in the roofline block too (the headline input is the
        # coherent "scene" G-buffer; on independent-per-pixel normals every lane gathers its own cube/shadow line)
        result["roofline"]["other_inputs"] = {k: {"us": ex[k]["us"], "achieved": ex[k]["GBps"], "frac": ex[k]["frac_hbm"]}
                                              for k in ("lighting_iid", "lighting_1080p_scene", "lighting_8k_scene")}
    if rank == 0 and N == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(fc, g, shadow, env, lut, bounds, lay, W, H)
    if timeline is not None:
        result["timeline"] = timeline
    if gc_on is not None:
        result["with_python_gc_on"] = gc_on
    if rank == 0:
        result["setup_seconds"] = round(gen_s, 1)
        print(json.dumps(result))
    if N > 1:
        dist.destroy_process_group()


def stream_ceiling_leg(hp, torch, nbytes, dev):
    """The practical ceiling at this launch's byte count (untimed extras leg): a plain four-reads-one-write streaming kernel
    (csrc/stream_ceiling.hip) moving `nbytes` per launch over three cold buffer sets of pseudo-random finite data, timed the way the
    Lighting launch is: events carried on the dispatch (every 8th launch), and the back-to-back loop figure beside it."""
    n16 = max(1, nbytes // (5 * 16))
    ring = 3
    gen = torch.Generator(device=f"cuda:{dev}")
    gen.manual_seed(1234)
    sets = []
    for _ in range(ring):
        ins = [(torch.randint(0, 0x3FFF, (n16 * 8,), dtype=torch.int16, device=f"cuda:{dev}", generator=gen) | 0x3000) for _ in range(4)]  # fp16 values in (0.125, 2)
        sets.append((ins, torch.empty(n16 * 8, dtype=torch.int16, device=f"cuda:{dev}")))
    for k in range(400):
        hp.stream_ceiling(*sets[k % ring])
    pairs = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    iters = 400
    for k in range(iters):
        if k % 8 == 0:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()  # (torch creates the HIP events at their first record)
            pairs.append((a, b))
            hp.stream_ceiling(*sets[k % ring], start=a, stop=b)
        else:
            hp.stream_ceiling(*sets[k % ring])
    e1.record()
    torch.cuda.synchronize()
    loop_us = e0.elapsed_time(e1) * 1e3 / iters
    disp_us = float(np.mean([a.elapsed_time(b) for a, b in pairs])) * 1e3
    moved = n16 * 16 * 5
    del sets
    torch.cuda.empty_cache()
    return {"bytes_per_launch": moved, "dispatch_us": disp_us, "loop_us": loop_us, "GBps": moved / disp_us / 1e3, "loop_GBps": moved / loop_us / 1e3}


def _time_events(torch, fn, iters, warm=5, batch=16):
    """Per-call time of fn: `batch` calls back to back between ONE event pair, `iters` such pairs (an event record costs ~5 us of queue
    time on this stack: around a single call it would be a fifth of what a 25-us kernel reads). Returns (median, min) seconds per call."""
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch):
            fn()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in evs]) * 1e-3 / batch
    return float(np.median(t)), float(t.min())


def tables_for(hp, to_device, synth, fc, kind, dev, env_dev):
    """Lighting side tables: the scene generator's shadow map (or the noise one for the iid G-buffer), the staged cube, the LUT."""
    if kind == "scene":
        shadow = synth.shadow_map_scene(np.ctypeslib.as_array(fc.scene.LightViewProjection), 2048)
    else:
        shadow = synth.shadow_map_noise(2048, synth.SEED_BASE + 3)
    return hp.make_tables(to_device(shadow, dev), env_dev[0], 256, 9, env_dev[1])


def light_leg(hp, torch, to_device, fc, g, tables, W, H, dev, ring, iters, warm):
    """One fused Lighting+Sky launch per iteration over `ring` cold buffer sets; per-launch time from ONE event pair."""
    sets = [dict(A=to_device(g.A, dev), B=to_device(g.B, dev), C=to_device(g.C, dev), D=to_device(g.depth, dev), hdr=to_device(g.hdr, dev)) for _ in range(ring)]
    n_sky = int((g.depth == 0).sum())
    nbytes = 40 * (g.depth.size - n_sky) + 12 * n_sky

    def fn(k):
        s = sets[k % ring]
        hp.deferred_lighting_sky(fc.scene, fc.sky, s["A"], s["B"], s["C"], s["D"], tables, s["hdr"], W, H)
    for k in range(warm):  # the chip needs ~10-20 ms of work before its clock is the sustained one
        fn(k)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(iters):
        fn(k)
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) * 1e-3 / iters
    del sets
    torch.cuda.empty_cache()
    return {"size": f"{W}x{H}", "background_fraction": round(n_sky / g.depth.size, 4), "bytes_per_launch": nbytes, "us": t * 1e6,
            "GBps": nbytes / t / 1e9, "frac_hbm": nbytes / t / 1e9 / HBM_PEAK_GBS, "mpixels_per_s": W * H / t / 1e6, "launches": iters}


def side_measurements(args, hp, fc, torch, to_device, HzbLayout, synth, hostmath, dev, env_dev):
    """Culled instances/s on BASELINE config 5's instance set (1 M AABBs against an 8K-frame HZB) and per-kernel times."""
    out = {}
    W8, H8 = 7680, 4320
    n = args.cull_instances
    fc8 = hostmath.build_frame_constants("sponza", W8, H8)
    lay8 = HzbLayout(W8, H8)
    g8 = synth.gbuffer_scene(fc8.view, fc8.proj, fc8.camera_position, W8, H8, synth.SEED_BASE + 5)
    depth8 = to_device(g8.depth, dev)
    hzb8 = torch.zeros(lay8.total, dtype=torch.float32, device=f"cuda:{dev}")
    # cold inputs, as in the lighting legs: three depth buffers and chains (3 x 177 MB) so that no launch finds its source in the
    # 256-MB memory-side cache the launch before left there; the same buffers over and over ("warm") is reported beside it
    kring = 3
    depths = [depth8] + [depth8.clone() for _ in range(kring - 1)]
    hzbs = [hzb8] + [torch.zeros_like(hzb8) for _ in range(kring - 1)]
    turn = [0]

    def hzb_cold():
        k = turn[0] % kring
        turn[0] += 1
        hp.build_hzb(depths[k], hzbs[k], lay8)

    med, mn = _time_events(torch, hzb_cold, 12, batch=15)
    med_w, _ = _time_events(torch, lambda: hp.build_hzb(depth8, hzb8, lay8), 12)
    hzb_bytes = 4 * (W8 * H8 + lay8.mip_texels())
    out["build_hzb_8k"] = {"median_us": med * 1e6, "GBps": hzb_bytes / med / 1e9, "frac_hbm": hzb_bytes / med / 1e9 / HBM_PEAK_GBS,
                           "inputs": "three buffer sets cycled (cold)", "same_buffers_us": med_w * 1e6}
    del depths[1:], hzbs[1:]
    bounds = to_device(synth.instances_random(n, synth.SEED_BASE + 5, center=fc8.camera_position, box=400.0), dev)
    d_args = to_device(synth.indirect_args_initial(n), dev)
    d_vis = torch.zeros(n, dtype=torch.int32, device=f"cuda:{dev}")
    d_cnt = torch.zeros(1, dtype=torch.int32, device=f"cuda:{dev}")
    d_stats = torch.zeros(2, dtype=torch.int32, device=f"cuda:{dev}")
    consts_dbg = hostmath.pack_culling_constants(fc8.view, fc8.proj, n, True, lay8.count, lay8.width, lay8.height, True)
    hp.cull_indirect_args(consts_dbg, bounds, hzb8, lay8, d_args, d_stats, d_vis, d_cnt)
    torch.cuda.synchronize()
    frustum_culled, occluded = (int(v) for v in d_stats.cpu().numpy().view(np.uint32))
    visible = int(d_cnt.cpu().numpy().view(np.uint32)[0])
    consts = hostmath.pack_culling_constants(fc8.view, fc8.proj, n, True, lay8.count, lay8.width, lay8.height, False)
    cring = 6  # 6 x (32 MB of bounds + 64 MB of commands + the list): cold, as above
    csets = [(bounds, d_args, d_vis)] + [(bounds.clone(), d_args.clone(), torch.zeros_like(d_vis)) for _ in range(cring - 1)]

    def cull_cold():
        b_, a_, v_ = csets[turn[0] % cring]
        turn[0] += 1
        hp.cull_indirect_args(consts, b_, hzb8, lay8, a_, None, v_, d_cnt)

    med, mn = _time_events(torch, cull_cold, 12)
    med_w, _ = _time_events(torch, lambda: hp.cull_indirect_args(consts, bounds, hzb8, lay8, d_args, None, d_vis, d_cnt), 12)
    # The launch leaves a word that already holds its value alone (UR_OPT_CULL_STORE = 3), and the legs above find the command buffer as
    # the previous launch left it - the steady frame. The other end: every command back to InstanceCount = 1 ahead of its launch (a first
    # frame, a camera cut: 99 % of these words then change). The reset of a set runs three launches ahead of its use so that the copy
    # has left the caches again; per-launch event pairs (the copies lie outside them; ~3 us of event cost inside).
    args_initial = d_args.clone()
    args_initial.copy_(to_device(synth.indirect_args_initial(n), dev))
    evs = []
    for k in range(cring):
        csets[k][1].copy_(args_initial)
    for i in range(40):
        b_, a_, v_ = csets[i % cring]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hp.cull_indirect_args(consts, b_, hzb8, lay8, a_, None, v_, d_cnt)
        e1.record()
        evs.append((e0, e1))
        csets[(i + 3) % cring][1].copy_(args_initial)  # used three launches from now
    torch.cuda.synchronize()
    med_reset = float(np.median([x.elapsed_time(y) for x, y in evs[8:]])) * 1e-3
    # UR_OPT_CULL_STORE = 4 (opt-in: the caller promises that only this context's culls write the words): the present values come from
    # the context's one-bit-per-instance record of its previous launch on the same command buffer instead of from the 64 MB of command
    # lines. One command buffer (the record is per buffer; it is not read at all), the bounds still cycle cold.
    from unclerenderer_amd import lib as urlib_
    hp.set_option(urlib_.UR_OPT_CULL_STORE, 4)

    def cull_record():
        b_, _, v_ = csets[turn[0] % cring]
        turn[0] += 1
        hp.cull_indirect_args(consts, b_, hzb8, lay8, d_args, None, v_, d_cnt)

    med_rec, _ = _time_events(torch, cull_record, 12)
    hp.set_option(urlib_.UR_OPT_CULL_STORE, 3)
    del csets[1:], args_initial
    f_frustum = 1.0 - frustum_culled / n
    cull_bytes = n * (36 + 16 * f_frustum) + 4 * visible
    # SURVEY.md section 8d asks for both accountings: the algorithmic 4 B per InstanceCount word, and the 64-byte line each of those
    # words is alone in (FIndirectDrawCommand stride 64: the layout is the reference's, the write amplification comes with it)
    line_bytes = n * (32 + 64 + 16 * f_frustum) + 4 * visible
    out["cull_1m"] = {"instances": n, "visible": visible, "frustum_culled": frustum_culled, "occluded": occluded,
                      "median_us": med * 1e6, "instances_per_s": n / med, "algorithmic_GBps": cull_bytes / med / 1e9,
                      "frac_hbm": cull_bytes / med / 1e9 / HBM_PEAK_GBS,
                      "with_64B_store_lines_GBps": line_bytes / med / 1e9, "frac_hbm_with_64B_store_lines": line_bytes / med / 1e9 / HBM_PEAK_GBS,
                      "inputs": "six buffer sets cycled (cold); command buffers as the previous frame left them (no word changes)",
                      "same_buffers_us": med_w * 1e6, "every_word_reset_first_us": med_reset * 1e6,
                      "from_the_contexts_record_us": med_rec * 1e6, "from_the_contexts_record_instances_per_s": n / med_rec}
    # ---- fused Lighting+Sky on the other G-buffers the contract names: the independent-per-pixel generator
    #      at the frame size (the stress case: every lane gathers its own cube / LUT / shadow line), C2's 1920x1080 and C5's
    #      7680x4320, Sponza constants, shipped IBL tables. Back-to-back launches over cold buffer sets between ONE event pair.
    del depth8, hzb8, bounds, d_args, d_vis
    torch.cuda.empty_cache()
    out["lighting_8k_scene"] = light_leg(hp, torch, to_device, fc8, g8, tables_for(hp, to_device, synth, fc8, "scene", dev, env_dev), W8, H8, dev, ring=2, iters=150, warm=60)
    del g8
    fc2 = hostmath.build_frame_constants("sponza", 1920, 1080)
    g2 = synth.gbuffer_scene(fc2.view, fc2.proj, fc2.camera_position, 1920, 1080, synth.SEED_BASE + 2)
    out["lighting_1080p_scene"] = light_leg(hp, torch, to_device, fc2, g2, tables_for(hp, to_device, synth, fc2, "scene", dev, env_dev), 1920, 1080, dev, ring=8, iters=1200, warm=800)
    W, H = args.width, args.height
    fci = hostmath.build_frame_constants("sponza", W, H)
    gi = synth.gbuffer_iid(W, H, synth.SEED_BASE + 3)
    out["lighting_iid"] = light_leg(hp, torch, to_device, fci, gi, tables_for(hp, to_device, synth, fci, "iid", dev, env_dev), W, H, dev, ring=3, iters=200, warm=80)
    del gi, g2
    torch.cuda.empty_cache()
    # the rows behind the path: Tonemap and TemporalAA at the frame size, back-to-back launches over
    # three cold buffer sets between ONE event pair
    W, H, ring = args.width, args.height, 3
    hdr = [(torch.rand((H, W, 4), device=f"cuda:{dev}") * 4.0).to(torch.float16).view(torch.int16) for _ in range(ring)]
    hist = [(torch.rand((H, W, 4), device=f"cuda:{dev}") * 4.0).to(torch.float16).view(torch.int16) for _ in range(ring)]
    out8 = [torch.zeros((H, W), dtype=torch.int32, device=f"cuda:{dev}") for _ in range(ring)]
    out16 = [torch.zeros((H, W, 4), dtype=torch.int16, device=f"cuda:{dev}") for _ in range(ring)]

    def loop(fn, iters=600):
        for k in range(400):  # ~10-20 ms of work before the clock is the sustained one
            fn(k)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(iters):
            fn(k)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e-3 / iters

    t = loop(lambda k: hp.tonemap(hdr[k % ring], out8[k % ring], W, H))
    out["tonemap"] = {"size": f"{W}x{H}", "us": t * 1e6, "GBps": 12 * W * H / t / 1e9, "frac_hbm": 12 * W * H / t / 1e9 / HBM_PEAK_GBS}
    t = loop(lambda k: hp.temporal_aa(hdr[k % ring], hist[k % ring], out16[k % ring], 0.9, True, W, H))
    out["temporal_aa"] = {"size": f"{W}x{H}", "us": t * 1e6, "GBps": 24 * W * H / t / 1e9, "frac_hbm": 24 * W * H / t / 1e9 / HBM_PEAK_GBS}
    return out


def cpu_baseline(fc, g, shadow, env, lut, bounds, lay, W, H):
    """The scalar C++ oracle (oracle/ur_oracle.cpp, -O2 -ffp-contract=off) timed on this box's host cores on one full
    frame of the same workload. kind = "port": the reference's D3D12/HLSL cannot run here."""
    from oracle import oracle as o
    from unclerenderer_amd import hostmath, synth
    o.build()
    # the GPU box gives one GPU a 16-core share of the host; never oversubscribe it
    cores = max(1, min(16, len(os.sched_getaffinity(0)), o.hardware_threads()))
    o.set_threads(cores)
    t0 = time.perf_counter()
    hzb = o.build_hzb(g.depth, lay.as_list(), lay.total)
    consts = hostmath.pack_culling_constants(fc.view, fc.proj, bounds.shape[0], True, lay.count, lay.width, lay.height, False)
    o.cull_indirect_args(consts, bounds, np.nan_to_num(hzb), lay.as_list(), synth.indirect_args_initial(bounds.shape[0]))
    t1 = time.perf_counter()
    lit = o.deferred_lighting(fc.scene, g.A, g.B, g.C, shadow, env, 256, 9, lut, g.hdr, W, H)
    o.sky_atmosphere(fc.sky, g.depth, lit, W, H)
    t2 = time.perf_counter()
    return {"value": W * H / (t2 - t0) / 1e6, "unit": "Mpixels/s", "cores": cores, "kind": "port",
            "sample": f"1 full {W}x{H} frame: BuildHZB+cull single-thread {t1 - t0:.2f}s, lighting+sky on {cores} threads {t2 - t1:.2f}s"}


if __name__ == "__main__":
    main()
