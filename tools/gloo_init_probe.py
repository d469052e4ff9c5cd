#!/usr/bin/env python3
"""How long does an 8-rank gloo world take to come up on this box?  python -m torch.distributed.run --nproc-per-node 8 ... tools/gloo_init_probe.py"""
import os
import time
t0 = time.perf_counter()
import torch
import torch.distributed as dist
t1 = time.perf_counter()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
t2 = time.perf_counter()
dist.barrier()
t3 = time.perf_counter()
x = torch.ones(4, dtype=torch.float64)
dist.all_reduce(x)
t4 = time.perf_counter()
objs = [None] * world
dist.all_gather_object(objs, dict(rank=rank))
t5 = time.perf_counter()
print(f"rank {rank}: import {t1 - t0:.2f} s, init_process_group {t2 - t1:.2f} s, first barrier {t3 - t2:.2f} s, all_reduce {t4 - t3:.2f} s, "
      f"all_gather_object {t5 - t4:.2f} s   GLOO_SOCKET_IFNAME={os.environ.get('GLOO_SOCKET_IFNAME')}", flush=True)
dist.destroy_process_group()
