#!/usr/bin/env python3
"""Does initialising RCCL BEFORE the library creates its four streams cost the hot path?  (The runtime places a process's streams on
its hardware queues in creation order -- DESIGN section 4 -- and an N > 1 bench.py rank calls dist.init_process_group first.)

    python tools/nccl_order_probe.py MODE WORKLOAD      MODE: none | nccl_first | context_first     (fresh process each)

none: no process group; nccl_first: init_process_group(nccl, world 1, device_id) + all_reduce + barrier, then the context;
context_first: a throw-away context (the library's streams exist from then on), then RCCL, then the measured context."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode, wl = sys.argv[1], sys.argv[2]
    import torch
    import bench
    dev = torch.device("cuda", 0)
    bench.open_device_with_retry(dev, 0)
    import torch.distributed as dist

    def rccl():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(bench.free_port()))
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=dev)
        t = torch.ones(8, device=dev)
        dist.all_reduce(t)
        dist.barrier()
    if mode == "context_first":
        import oat_amd
        oat_amd.HotPath(16, 64, n_streams=1, device=0).close()
    if mode != "none":
        rccl()
    leg = bench.Leg(wl, 0, 0, pool=24)

    def barrier():
        leg.hp.synchronize()
        torch.cuda.synchronize()
        if mode != "none":
            dist.barrier()
    K = {"vga1": 1000, "1080p1": 1000}.get(wl, 600)
    tr = bench.timed_run(leg, K, 100, barrier, 8, age_frames=300, export=False, spin=0.0)
    k1 = bench.k1_ms(tr["prof"])[0]
    print(f"{mode:14s} {wl:8s} fps {leg.ns * K / tr['block_s']:9.1f}  step {tr['block_s'] / K * 1e6:7.2f} us  K1 {k1 * 1e3:6.1f} us", flush=True)
    leg.close()
    if mode != "none":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
