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


class FrameScatterPipe:
    """Double-buffered stream->rank scatter for the topology where all frames originate on ONE rank (a camera
    host): the frames of step t+1 travel while step t computes.

    One 1080p stream is 6.2 MB; at BASELINE config 4 (64 streams, 8 per GPU) the root sends 49.8 MB to each of
    its 7 peers per step -- about 0.33 ms on one xGMI link (point-to-point: the 7 blocks use the 7 links in
    parallel, which is why this is send/recv per peer and not a ring collective) -- the same order as the
    0.25-0.5 ms compute step, so the transfer has to hide under the previous step:

        pipe.post(0, frames0)                 # root: frames0[n_streams_total, ...]; other ranks: None
        for t in range(T):
            if t + 1 < T: pipe.post(t + 1, frames(t + 1))      # starts moving while ...
            local = pipe.take(t)                               # ... step t is taken and computed
            compute(local)

    post(t) uses buffer slot t % depth, so take(t)'s tensor stays valid until post(t + depth) -- and post(t + depth)
    must not start before the kernel that READS the slot has finished: the receive runs on RCCL's stream, the
    per-pixel kernel on the hot path's.  Pass the HotPath as `consumer`: post() then calls its input_consumed()
    (oatgpu_track_input_consumed) before it reuses a slot, and take() orders the consumer's HIP stream behind the
    transfer with a stream-to-stream wait instead of blocking the host (without a consumer it synchronises the
    device's current stream, which serialises the overlap the pipe exists for).  Backend "nccl" (= RCCL) on the
    GPU box; the tests drive the same code over gloo.  Unmeasured on multi-GPU hardware so far (the builder has
    one GPU at a time); bench.py's N > 1 runs time it as their `scatter_ingest` leg beside the per-rank-ingest `value`.

    via_host=True: the transport runs between page-locked HOST buffers and take() uploads the block -- for backends
    that cannot send from device memory (gloo: the 2-rank smoke tests on one GPU).  Never the RCCL path."""

    def __init__(self, n_streams_total, frame_shape, device, src=0, group=None, depth=2, consumer=None, via_host=False):
        self.total, self.shape, self.device, self.src, self.group = n_streams_total, tuple(frame_shape), device, src, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.per = -(-n_streams_total // self.world)
        self.mine = stream_partition(n_streams_total, self.world, self.rank)
        self.depth = depth
        self.consumer = consumer
        self.buf = [torch.empty((self.per,) + self.shape, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.via_host = bool(via_host) and torch.device(device).type == "cuda"
        self.xbuf = None
        if self.via_host and self.rank != src:
            self.xbuf = [torch.empty((self.per,) + self.shape, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.up_ev = [None] * depth         # via_host: the upload out of xbuf[k] has finished (post() must not receive into it before)
        self.work = [None] * depth          # outstanding requests of the slot
        self.keep = [None] * depth          # root: the frame tensor being sent out of
        self.step_of = [None] * depth
        self.bytes_per_peer = self.per * int(np.prod(self.shape))

    def post(self, t, frames_root=None):
        k = t % self.depth
        if self.work[k] is not None:
            raise RuntimeError(f"slot of step {self.step_of[k]} was never taken")
        if self.step_of[k] is not None and self.consumer is not None:
            self.consumer.input_consumed()          # the kernel reading this slot's previous frames has finished
        ops = []
        if self.rank == self.src:
            fr = frames_root if frames_root.device == self.buf[k].device else frames_root.to(self.device, non_blocking=True)
            assert fr.shape[0] == self.total and tuple(fr.shape[1:]) == self.shape
            out = fr.cpu() if (self.via_host and self.world > 1) else fr        # (host transport: one D2H of the whole set)
            self.keep[k] = (fr, out)
            for peer in range(self.world):
                blk = stream_partition(self.total, self.world, peer)
                if not len(blk):
                    continue
                if peer == self.rank:
                    self.buf[k][:len(blk)].copy_(fr[blk.start:blk.stop], non_blocking=True)
                else:
                    ops.append(dist.P2POp(dist.isend, out[blk.start:blk.stop].contiguous(), peer, self.group))
        elif len(self.mine):
            if self.via_host and self.up_ev[k] is not None:
                self.up_ev[k].synchronize()         # (the receive below writes xbuf[k] from a CPU thread -- the
                self.up_ev[k] = None                #  asynchronous upload that read it for step t - depth must be through)
            dst = self.xbuf[k] if self.via_host else self.buf[k]
            ops.append(dist.P2POp(dist.irecv, dst[:len(self.mine)], self.src, self.group))
        self.work[k] = dist.batch_isend_irecv(ops) if ops else []
        self.step_of[k] = t

    def take(self, t):
        k = t % self.depth
        if self.step_of[k] != t or self.work[k] is None:
            raise RuntimeError(f"step {t} was not posted (slot holds {self.step_of[k]})")
        for w in self.work[k]:
            w.wait()
        if self.xbuf is not None and len(self.mine):
            self.buf[k][:len(self.mine)].copy_(self.xbuf[k][:len(self.mine)], non_blocking=True)
            self.up_ev[k] = torch.cuda.Event()
            self.up_ev[k].record(torch.cuda.current_stream(self.buf[k].device))
        if self.buf[k].is_cuda:
            cur = torch.cuda.current_stream(self.buf[k].device)
            if self.consumer is not None and self.consumer.get_stream():
                # the hot path runs on its own HIP streams: order ITS stream behind the transfer, never the host
                torch.cuda.ExternalStream(self.consumer.get_stream(), device=self.buf[k].device).wait_stream(cur)
            else:
                cur.synchronize()
        self.work[k] = None
        self.keep[k] = None
        return self.buf[k][:len(self.mine)]
