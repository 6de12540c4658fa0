"""N > 1 on the HIP path, checked (the reference is single-adapter, Source/RHI/DX12Device.cpp:127-133):
two FRESH rank processes (children of tests/_spawner.py, which never touches the GPU; both ranks on GPU 0, gloo backend —
RCCL refuses two ranks on one device) each run the HIP kernels on their shard exactly as bench.py does under RCCL
(Frame.render on the row band with the instance range and index_base; the 1 M-instance cull on the range), gather through
unclerenderer_amd/dist.py (HDR ring / direct / overlapped, tonemapped RGBA8, the visible list, the InstanceCount words), and
the result must be, byte for byte, what ONE rank's HIP frame gives in this process:

  C4  pica_pica's camera and light, 3840x2160, its 170 draw commands, the render-graph frame with Build HZB riding Lighting
  C5  1 M instance AABBs against the 12-mip chain of a 7680x4320 depth
"""
import json
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
WORKER = str(ROOT / "tests" / "_multirank_worker.py")


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(spawn_ranks, case, out, world=2, extra=()):
    port = _port()
    envs = [dict(RANK=r, LOCAL_RANK=r, WORLD_SIZE=world, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=0, OMP_NUM_THREADS=4)
            for r in range(world)]
    res = spawn_ranks([sys.executable, WORKER, "--case", case, "--out", str(out), *extra], envs, timeout=540)
    assert res["rc"] == [0] * world, "rank processes failed:\n" + "\n----\n".join(res["tail"])
    digests = [json.loads((out / f"rank{r}.json").read_text()) for r in range(world)]
    for r in range(1, world):
        assert digests[r] == digests[0], f"rank {r} gathered different bytes than rank 0"
    return np.load(out / "rank0.npz")


def test_c4_two_ranks_reproduce_the_single_rank_hip_frame(hotpath, spawn_ranks, tmp_path):
    import torch
    from tests._multirank_worker import c4_render
    w, h = 3840, 2160
    got = _run_ranks(spawn_ranks, "c4", tmp_path)
    plan, d, n = c4_render(hotpath, w, h, 0, 1)  # the same code path with one rank: the whole frame, every instance
    assert plan.rows == h and n == 170
    hdr = d["hdr_band"].cpu().numpy()
    ldr = d["ldr_band"].cpu().numpy()
    cnt = int(d["cnt"].cpu()[0])
    vis = d["vis"].cpu().numpy()[:cnt]
    assert np.isfinite(hdr.view(np.float16).astype(np.float32)).all()
    for key in ("hdr_ring", "hdr_direct", "hdr_async"):
        assert np.array_equal(got[key], hdr), f"{key}: the gathered bands differ from the single-rank HIP frame"
    for key in ("ldr_ring", "ldr_direct"):
        assert np.array_equal(got[key], ldr), f"{key}: the gathered tonemapped bands differ from the single-rank HIP frame"
    assert 0 < cnt <= n and np.array_equal(got["vis"], vis), "rank-order concatenation of the per-rank lists == the single-rank list"
    assert np.all(np.diff(got["vis"].astype(np.int64)) > 0)
    assert np.array_equal(got["args"], d["args"].cpu().numpy().view(np.int32).reshape(-1, 16)), "InstanceCount words"
    torch.cuda.synchronize()


def test_c5_two_ranks_cull_1m_like_one(hotpath, spawn_ranks, tmp_path):
    from tests._multirank_worker import c5_cull
    n = 1_000_000
    got = _run_ranks(spawn_ranks, "c5", tmp_path, extra=("--instances", str(n)))
    d, _ = c5_cull(hotpath, n, 0, 1)
    cnt = int(d["cnt"].cpu()[0])
    assert cnt > 1000
    assert np.array_equal(got["vis"], d["vis"].cpu().numpy()[:cnt]), "visible list"
    assert np.array_equal(got["args"], d["args"].cpu().numpy().view(np.int32).reshape(-1, 16)), "InstanceCount words"
    assert np.array_equal(got["stats"], d["stats"].cpu().numpy()), "frustum-culled / occluded counters sum to the single-rank ones"


@pytest.mark.parametrize("w,h", [(3840, 2160), (1904, 1052)])
def test_c4_two_ranks_with_band_sharded_hzb(hotpath, spawn_ranks, tmp_path, w, h):
    """The same frame with Build HZB band-sharded (UR_FRAME_HZB_SHARD): each rank builds mips 0..4 for the 128x32 pieces its rows own,
    riding its Lighting launch, the slices cross peer to peer (dist.allgather_hzb_slices) and the tail is replicated behind the
    exchange. After two frames the HZB on every rank, the cull that read it (visible list, InstanceCount words) and the HDR / LDR
    frames are byte for byte the single-rank chain's - at 4K and at an odd size (1052 rows: bands of 526 rows, 16.4 pieces)."""
    import torch
    from tests._multirank_worker import c4_render
    got = _run_ranks(spawn_ranks, "c4", tmp_path, extra=("--shard-hzb", "--width", str(w), "--height", str(h)))
    plan, d, n = c4_render(hotpath, w, h, 0, 1)
    torch.cuda.synchronize()
    from unclerenderer_amd.hotpath import HzbLayout
    lay = HzbLayout(w, h)
    valid = np.zeros(lay.total, bool)
    for off, mw, mh in lay.as_list():
        valid[off:off + mw * mh] = True
    ref_hzb = d["hzb"].cpu().numpy()
    assert np.array_equal(got["hzb"].view(np.uint32)[valid], ref_hzb.view(np.uint32)[valid]), "gathered slices + replicated tail == the single-rank chain"
    assert np.array_equal(got["hdr_ring"], d["hdr_band"].cpu().numpy())
    assert np.array_equal(got["ldr_direct"], d["ldr_band"].cpu().numpy())
    cnt = int(d["cnt"].cpu()[0])
    assert np.array_equal(got["vis"], d["vis"].cpu().numpy()[:cnt]) and cnt > 0
    assert np.array_equal(got["args"], d["args"].cpu().numpy().view(np.int32).reshape(-1, 16))
