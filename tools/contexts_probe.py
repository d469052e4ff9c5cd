#!/usr/bin/env python3
"""Eight contexts on one device against one context of eight times the streams.

    python tools/contexts_probe.py [--contexts 8] [--workload 1080p8] [--steps 200]

All contexts of a device in one process share ONE A stream, three B streams and the copy streams (DESIGN.md section 4:
ROCm 7.2 spreads a process's streams over four hardware queues, two streams on one queue serialise).  `oat-track-hip
--gpu-index 0,0,..` and a multi-camera host process are this shape.  Measured here: C contexts of S streams each, one
driving thread per context (ctypes releases the GIL inside the library), every thread running the pipelined sequence
call over frames resident in HBM -- against ONE context of C x S streams; frames per second over all streams, and per
context.  Prints one JSON line."""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bench


def run_contexts(name, C, steps, warm, pool):
    legs = [bench.Leg(name, 0, r, pool=pool) for r in range(C)]
    for l in legs:
        l.init()
        l.age(120)
        l.run(warm)
        l.hp.synchronize()
    prepared = [l.prepare(steps) for l in legs]
    times = [0.0] * C
    start = threading.Barrier(C + 1)

    def drive(i):
        start.wait()
        t0 = time.perf_counter()
        legs[i].run(steps, prepared[i])
        legs[i].hp.synchronize()
        times[i] = time.perf_counter() - t0
    th = [threading.Thread(target=drive, args=(i,)) for i in range(C)]
    for t in th:
        t.start()
    torch.cuda.synchronize()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    ns = legs[0].ns
    for l in legs:
        l.close()
    torch.cuda.empty_cache()
    return dict(contexts=C, streams_per_context=ns, steps=steps, wall_s=wall, fps_total=C * ns * steps / wall,
                fps_per_context=[ns * steps / t for t in times])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contexts", type=int, default=8)
    ap.add_argument("--workload", default="1080p8")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--pool", type=int, default=12)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    w = bench.WORKLOADS[a.workload]
    big = f"{a.workload}x{a.contexts}"
    bench.WORKLOADS[big] = dict(w, streams=w["streams"] * a.contexts)
    out = dict(workload=a.workload, what="C contexts x S streams (one driving thread each, shared A/B/copy streams) against ONE "
                                         "context of C x S streams, frames resident in HBM, pipelined sequence call")
    out["many"] = run_contexts(a.workload, a.contexts, a.steps, a.warmup, a.pool)
    out["one"] = run_contexts(big, 1, a.steps, a.warmup, a.pool)
    out["many_over_one"] = out["many"]["fps_total"] / out["one"]["fps_total"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
