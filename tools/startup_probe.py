#!/usr/bin/env python3
"""Where does the start-up of N processes on ONE GPU go?  (round 6: `bench.py --gpus 8 --backend gloo` on one device spends
10+ minutes before its first timed step.)   python tools/startup_probe.py N [pool]   -> one line per process and phase"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys, time
t0 = time.perf_counter(); T = {}
def lap(k):
    global t0
    t1 = time.perf_counter(); T[k] = round(t1 - t0, 2); t0 = t1
sys.path.insert(0, %r)
import torch
lap("import torch")
if os.environ.get("PROBE_GLOO"):
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    lap("gloo init")
if os.environ.get("PROBE_PIN"):
    import bench as _b
    _b.pin_to_gpu_node(0)
    lap("pin to the GPU's NUMA node (%%d cpus)" %% len(os.sched_getaffinity(0)))
torch.cuda.set_device(0)
x = torch.empty(1 << 20, device="cuda"); torch.cuda.synchronize()
lap("first allocation")
g = torch.Generator(device="cuda"); g.manual_seed(1)
f = torch.randint(-6, 7, (1, 2160, 3840, 3), device="cuda", dtype=torch.int16, generator=g); torch.cuda.synchronize()
lap("first randint 4K")
m = ((torch.arange(2160 * 3840, device="cuda") %% 64) == 17).view(2160, 3840)
f[:, m] += 60; torch.cuda.synchronize()
lap("first masked add")
u = f.clamp_(0, 255).to(torch.uint8); torch.cuda.synchronize()
lap("first clamp / cast")
import bench
pool = bench.make_pool_device(2160, 3840, 1, int(sys.argv[1]), int(os.environ.get("RANK", "0")), torch.device("cuda", 0)); torch.cuda.synchronize()
lap("pool of %%s 4K frames" %% sys.argv[1])
hp = bench.make_hotpath(bench.WORKLOADS["4k1"], 0)
lap("HotPath create")
hp.track_dev(pool[0].data_ptr())
lap("first frame through the library")
print(json.dumps(dict(rank=int(os.environ.get("RANK", "0")), **T)), flush=True)
""" % ROOT


def main():
    if os.environ.get("PROBE_CHILD"):            # under torch.distributed.run: this process IS a rank
        sys.argv = [sys.argv[0], sys.argv[1] if len(sys.argv) > 1 else "48"]
        exec(CHILD)
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    pool = sys.argv[2] if len(sys.argv) > 2 else "48"
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", CHILD, pool], env=dict(os.environ, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, text=True) for r in range(n)]
    for p in procs:
        out, _ = p.communicate()
        print(out.strip())
    print(f"{n} process(es): {time.perf_counter() - t0:.1f} s wall")


if __name__ == "__main__":
    main()
