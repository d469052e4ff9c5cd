"""CPU tests, world_size 2 over gloo: the N > 1 plumbing (stream partition, stream->rank scatter,
position gather).  The data path itself has no collective (streams are independent)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oat_amd import dist as od
from oat_amd.components import Position2D


def test_partition_covers_every_stream_once():
    for total in (1, 7, 8, 16, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                blk = list(od.stream_partition(total, world, r))
                assert all(od.owner_of(s, total, world) == r for s in blk)
                seen += blk
            assert seen == list(range(total))
    assert list(od.stream_partition(64, 8, 3)) == list(range(24, 32))     # BASELINE config 4: 8 per GPU


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape = (6, 8, 3)
        rng = np.random.default_rng(5)
        allf = torch.from_numpy(rng.integers(0, 256, (total,) + shape, dtype=np.uint8))
        got = od.scatter_frames(allf if rank == 0 else None, total, shape, torch.device("cpu"), src=0)
        mine = od.stream_partition(total, world, rank)
        ok = got.shape[0] == len(mine) and all((got[i] == allf[s]).all() for i, s in enumerate(mine))
        # each rank "detects" a position that encodes its global stream id
        pos = [Position2D(True, float(s), float(2 * s), 3.0 * s, -(2 ** 40) - s, s, -s, s) for s in mine]
        allp = od.gather_positions(pos, total, torch.device("cpu"), dst=0)
        if rank == 0:
            ok = ok and allp.shape == (total, od.POS_FIELDS)
            for s in range(total):
                ok = ok and allp[s].tolist() == [1.0, float(s), float(s), float(2 * s), 3.0 * s,
                                                 float(-(2 ** 40) - s), float(s), float(-s)]
        else:
            ok = ok and allp is None
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8])
def test_scatter_and_gather_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}


def _pipe_worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape, T = (4, 5, 3), 7
        mine = od.stream_partition(total, world, rank)

        def frames(t):          # every byte encodes (step, stream)
            f = torch.empty((total,) + shape, dtype=torch.uint8)
            for s in range(total):
                f[s] = (17 * t + 3 * s) % 251
            return f
        pipe = od.FrameScatterPipe(total, shape, torch.device("cpu"), src=0, depth=2)
        ok = True
        pipe.post(0, frames(0) if rank == 0 else None)
        for t in range(T):
            if t + 1 < T:
                pipe.post(t + 1, frames(t + 1) if rank == 0 else None)     # in flight while step t is consumed
            local = pipe.take(t)
            ok = ok and local.shape[0] == len(mine)
            for i, s in enumerate(mine):
                ok = ok and bool((local[i] == (17 * t + 3 * s) % 251).all())
        # protocol errors are reported, not silently reordered
        try:
            pipe.take(T + 5)
            ok = False
        except RuntimeError:
            pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [3, 8])
def test_double_buffered_scatter_keeps_step_order_world2(total):
    """The transfer of step t+1 is posted before step t is taken; every rank must still see
    exactly its own streams of exactly step t."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(2))
    assert res == {0: True, 1: True}
