"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on
the GPU box, "gloo" in the CPU tests). The reference is single-GPU; this is new design:

  * DeferredLighting/Sky: contiguous row bands, rank r shades rows [r*H/N, (r+1)*H/N); ONE all-gather of the RGBA16F
    bands per frame rebuilds the full HDR buffer on every rank (equal counts, N | H).
  * CullIndirectArgs: contiguous instance ranges; the per-rank visible lists are already ascending and carry global
    indices (index_base), so concatenating them in rank order is the single-GPU list bit for bit. Counts are
    all-gathered first, then the lists are all-gathered padded to the largest count.
  * Tonemap ahead of the gather: the band is tonemapped to R8G8B8A8 on the rank that shaded it and the
    4-byte pixels are gathered instead (allgather_rows), halving the xGMI payload.
  * BuildHZB: replicated (every rank builds the full chain from the full depth) — no exchange; or band-sharded
    (allgather_hzb_slices): a rank builds mips 0..4 for the 128x32 source pieces its rows own, the slices (5 contiguous runs of
    floats per rank) are exchanged peer to peer straight into place, and every rank runs the single-workgroup tail behind it.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class BandPlan:
    rank: int
    world: int
    height: int
    row0: int
    rows: int


def plan_bands(height: int, world: int, rank: int) -> BandPlan:
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    if height % world != 0:
        raise ValueError(f"frame height {height} is not divisible by {world} ranks (the HDR all-gather needs equal bands)")
    rows = height // world
    return BandPlan(rank, world, height, rank * rows, rows)


def plan_instances(count: int, world: int, rank: int) -> tuple[int, int]:
    """[first, last) of the contiguous instance range of `rank` (sizes differ by at most one)."""
    return rank * count // world, (rank + 1) * count // world


class _Works:
    """The requests of one direct gather behind the Work interface the ring form returns."""

    def __init__(self, reqs, finish=None):
        self._reqs, self._finish = reqs, finish

    def wait(self):
        for r in self._reqs:
            r.wait()
        if self._finish is not None:
            self._finish()
            self._finish = None
        return True


def _allgather_direct(full: torch.Tensor, band: torch.Tensor, group, async_op: bool):
    """Every rank sends its band to every peer and receives every peer's band into that peer's rows: N - 1 send/receive
    pairs per rank in ONE batch (on "nccl" = RCCL: ncclGroupStart ... ncclSend / ncclRecv ... ncclGroupEnd), each pair over the
    one xGMI link its two GPUs share, all in flight together (SURVEY.md H1's direct-link floor) instead of RCCL's ring or
    tree all-gather. Same bytes in `full` afterwards."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    fb = full.view(torch.uint8).view(world, -1)
    mine = band.contiguous().view(torch.uint8).view(-1)
    if mine.data_ptr() != fb[rank].data_ptr():
        fb[rank].copy_(mine)
    # gloo moves host memory only: device tensors are staged through the host around the exchange (the CPU tests, and the
    # rehearsal of several ranks on one GPU); RCCL takes the device pointers as they are
    staged = dist.get_backend(group) == "gloo" and full.is_cuda
    src = mine.cpu() if staged else mine
    dst = torch.empty((world, mine.numel()), dtype=torch.uint8) if staged else fb
    ops = []
    for k in range(1, world):  # peers in the order rank + 1, rank + 2, ...: at any moment every rank sends to a different peer
        to, frm = (rank + k) % world, (rank - k) % world
        ops.append(dist.P2POp(dist.isend, src, dist.get_global_rank(group, to) if group is not None else to, group))
        ops.append(dist.P2POp(dist.irecv, dst[frm], dist.get_global_rank(group, frm) if group is not None else frm, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []

    def finish():
        if staged:
            for frm in range(world):
                if frm != rank:
                    fb[frm].copy_(dst[frm])
    w = _Works(reqs, finish)
    if async_op:
        return w
    w.wait()
    return None


def allgather_hdr(hdr_full: torch.Tensor, band: torch.Tensor, group=None, async_op: bool = False, mode: str = "ring"):
    """hdr_full: (H, W, 4) 16-bit tensor on every rank; band: this rank's (H/N, W, 4) rows. One collective
    (mode "ring": all_gather_into_tensor, RCCL's choice of ring or tree) or one batch of N - 1 send/receive pairs per rank
    (mode "direct": _allgather_direct).

    async_op=True returns the collective's Work handle (None when there is nothing to exchange): the gather runs on the
    backend's communication stream behind everything already queued on the current stream, and `work.wait()` makes the
    current stream wait for it — the caller overlaps it with the next frame's passes and waits before it touches
    `hdr_full` or `band` again."""
    if mode not in ("ring", "direct"):
        raise ValueError(f"gather mode {mode!r} (ring | direct)")
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        if band.data_ptr() != hdr_full.data_ptr():
            hdr_full.copy_(band.view_as(hdr_full))
        return None
    if mode == "direct":
        return _allgather_direct(hdr_full, band, group, async_op)
    # transported as bytes: neither RCCL nor gloo has a 16-bit integer type, and the payload is opaque fp16 bit patterns
    work = dist.all_gather_into_tensor(hdr_full.view(torch.uint8).view(-1), band.contiguous().view(torch.uint8).view(-1), group=group,
                                       async_op=async_op)
    return work if async_op else None


def allgather_rows(full: torch.Tensor, band: torch.Tensor, group=None, async_op: bool = False, mode: str = "ring"):
    """The same collective for any row-major image whose leading dimension is rows (the tonemapped R8G8B8A8 band,
    Tonemap.hlsl:57-79: 4 B/pixel instead of the HDR band's 8): `full` (H, ...) on every rank, `band` this rank's H/N rows."""
    return allgather_hdr(full, band, group=group, async_op=async_op, mode=mode)


def allgather_visible(visible_idx: torch.Tensor, visible_count: torch.Tensor, group=None) -> tuple[torch.Tensor, int]:
    """Concatenate the per-rank ascending visible lists in rank order. visible_idx holds global indices (index_base);
    visible_count is a 1-element integer tensor. Returns (list, total)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        n = int(visible_count.item())
        return visible_idx[:n].clone(), n
    world = dist.get_world_size(group)
    counts = torch.zeros(world, dtype=visible_count.dtype, device=visible_count.device)
    dist.all_gather_into_tensor(counts, visible_count.reshape(1), group=group)
    counts_host = [int(c) for c in counts.cpu()]
    cap = max(max(counts_host), 1)
    send = torch.zeros(cap, dtype=visible_idx.dtype, device=visible_idx.device)
    n = counts_host[dist.get_rank(group)]
    send[:n] = visible_idx[:n]
    recv = torch.zeros(world * cap, dtype=visible_idx.dtype, device=visible_idx.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    parts = [recv[r * cap: r * cap + counts_host[r]] for r in range(world)]
    return torch.cat(parts), sum(counts_host)


def allgather_hzb_slices(hzb: torch.Tensor, layout, group=None, async_op: bool = False):
    """Band-sharded Build HZB (ur_build_hzb_band): rank r has written mips 0..4 for ITS piece rows into `hzb` (the whole HZB
    allocation, fp32); send those five runs to every peer and receive every peer's into place - 5 (N - 1) send/receive pairs per
    rank in ONE batch (RCCL: one grouped call, every pair over the one xGMI link its two GPUs share). Slices differ in size by a
    piece row, so this is peer-to-peer, not an all_gather of equal counts. Afterwards mips 0..4 are complete on every rank
    (run HotPath.build_hzb_tail behind it). Returns bytes sent per peer (the added xGMI payload), and the Work when async."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0, None
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    slices = [layout.band_slices(*layout.band_pieces(world, r)) for r in range(world)]
    staged = dist.get_backend(group) == "gloo" and hzb.is_cuda
    host = hzb.cpu() if staged else hzb  # (gloo moves host memory only: rehearsals and the CPU tests)
    flat = host.view(-1)
    ops = []
    for k in range(1, world):
        to, frm = (rank + k) % world, (rank - k) % world
        for off, cnt in slices[rank]:
            if cnt:
                ops.append(dist.P2POp(dist.isend, flat[off:off + cnt], dist.get_global_rank(group, to) if group is not None else to, group))
        for off, cnt in slices[frm]:
            if cnt:
                ops.append(dist.P2POp(dist.irecv, flat[off:off + cnt], dist.get_global_rank(group, frm) if group is not None else frm, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []

    def finish():
        if staged:
            for r in range(world):
                if r != rank:
                    for off, cnt in slices[r]:
                        hzb.view(-1)[off:off + cnt].copy_(flat[off:off + cnt])
    w = _Works(reqs, finish)
    sent = 4 * sum(c for _, c in slices[rank])
    if async_op:
        return sent, w
    w.wait()
    return sent, None
