"""Multi-GPU sharding of independent camera streams.

Streams share nothing (own MOG2 model, own output), so the hot path shards with
NO data-path collective: stream s lives on rank s // ceil(S/N) for its whole
life.  torch.distributed (backend "nccl" == RCCL over xGMI on the GPU box,
"gloo" in the CPU tests) is used only for
  * scatter_frames: the stream->rank scatter when frames originate on one rank
    (a camera host); per-rank ingest skips it, and
  * gather_positions: N tiny position records back to one rank.
"""
import numpy as np
import torch
import torch.distributed as dist

POS_FIELDS = 8   # valid, first_pixel, x, y, area, a00, a10, a01


def stream_partition(n_streams_total, world, rank):
    """Contiguous block of global stream ids owned by `rank` (possibly empty)."""
    per = -(-n_streams_total // world)
    lo = min(rank * per, n_streams_total)
    hi = min(lo + per, n_streams_total)
    return range(lo, hi)


def owner_of(stream, n_streams_total, world):
    per = -(-n_streams_total // world)
    return stream // per


def scatter_frames(frames_root, n_streams_total, frame_shape, device, src=0, group=None):
    """Root holds frames_root[n_streams_total, *frame_shape] (uint8); every rank returns its own
    [n_local, *frame_shape] block on `device`.  One send per peer (xGMI is point-to-point: the
    root's 7 links carry the 7 blocks in parallel)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-n_streams_total // world)
    out = torch.empty((per,) + tuple(frame_shape), dtype=torch.uint8, device=device)
    chunks = None
    if rank == src:
        assert frames_root.shape[0] == n_streams_total
        pad = per * world - n_streams_total
        fr = frames_root.to(device)
        if pad:
            fr = torch.cat([fr, torch.zeros((pad,) + tuple(frame_shape), dtype=torch.uint8, device=device)])
        chunks = [c.contiguous() for c in fr.split(per)]
    dist.scatter(out, chunks, src=src, group=group)
    return out[:len(stream_partition(n_streams_total, world, rank))]


def pack_positions(positions):
    """list of Position2D -> float64 tensor [n, POS_FIELDS] (int64 sums are exact in float64: < 2^53)."""
    a = np.zeros((len(positions), POS_FIELDS), np.float64)
    for i, p in enumerate(positions):
        a[i] = (float(p.position_valid), float(p.first_pixel), p.x, p.y, p.area, float(p.a00), float(p.a10),
                float(p.a01))
    return torch.from_numpy(a)


def gather_positions(local_positions, n_streams_total, device, dst=0, group=None):
    """Every rank contributes its streams' positions; rank dst gets [n_streams_total, POS_FIELDS]
    in global stream order, the others None."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = -(-n_streams_total // world)
    mine = torch.zeros((per, POS_FIELDS), dtype=torch.float64, device=device)
    if len(local_positions):
        mine[:len(local_positions)] = pack_positions(local_positions).to(device)
    bufs = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    dist.gather(mine, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat(bufs)[:n_streams_total].cpu()
