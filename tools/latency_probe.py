#!/usr/bin/env python3
"""Single-frame latency of the fused chain: SYNCHRONOUS steps (one frame in, wait, one result out), so that no
kernel of one frame overlaps another frame's -- what a camera-bound pipeline sees, and the clean way to read the
back-half kernels' own durations out of a kernel trace:

    python tools/latency_probe.py [--workload 1080p1] [--steps 300]
    rocprofv3 --kernel-trace -d /tmp/kt -o r -- python tools/latency_probe.py ...   (tools/ktrace_latency.sh)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="1080p1")
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()
    import torch
    import bench
    leg = bench.Leg(a.workload, 0, 0, pool=24)
    leg.init()
    for i in range(50):
        leg.hp.track_dev(leg.pool[leg.pool_index(1 + i)].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        leg.hp.track_dev(leg.pool[leg.pool_index(51 + i)].data_ptr())
    el = time.perf_counter() - t0
    print(f"{a.workload}: {el / a.steps * 1e3:.4f} ms per synchronous step ({a.steps} steps)")
    leg.close()


if __name__ == "__main__":
    main()
