#!/usr/bin/env python3
"""When do k_rowscan's workgroups START and how long do they RUN beside the per-pixel kernel?

    OATGPU_MEASURE_PY=1 OATGPU_LIB=build/variants/liboatgpu_rst.so python tools/rowscan_probe.py [--workload 4k1] [--mode load|alone]

Needs a -DOATGPU_RS_TIMING build (make variant NAME=rst DEFS=-DOATGPU_RS_TIMING): every workgroup of stream 0 stamps the
100 MHz wall clock at its first and last instruction; launches are told apart by their ticket (early order).  Prints, per
launch (median over launches): the span from the first workgroup's start to the last one's end (what a kernel trace calls
the duration), how the workgroups' START times spread over it, and how long one workgroup runs.
"""
import argparse
import ctypes as C
import os
import statistics as st
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="4k1")
    ap.add_argument("--mode", default="load", choices=["load", "alone"])
    ap.add_argument("--steps", type=int, default=100)
    a = ap.parse_args()
    import torch
    import bench
    leg = bench.Leg(a.workload, 0, 0, pool=24)
    leg.init()
    leg.age(300)
    lib = leg.hp.lib
    lib.oatgpu_debug_rs_timing.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    lib.oatgpu_debug_rs_timing.restype = C.c_int
    if a.mode == "load":
        leg.run(a.steps)
    else:
        for i in range(a.steps):
            leg.hp.track_dev(leg.pool[leg.pool_index(leg.step + i)].data_ptr())
    torch.cuda.synchronize()
    N = 1 << 16
    buf = (C.c_longlong * (N * 5))()
    n = lib.oatgpu_debug_rs_timing(buf, N)
    launches = defaultdict(list)
    for r in range(n):
        t0, t1, wg, tag, k1e = buf[r * 5], buf[r * 5 + 1], buf[r * 5 + 2], buf[r * 5 + 3], buf[r * 5 + 4]
        launches[tag].append((t0, t1, wg, k1e))
    nwg = (leg.wl["rows"] + 3) // 4
    # (early order: the two frames of a step carry the same ticket on their two scratch sets -- 2 x nwg entries a tag, run
    # side by side on two streams: split by first / second occurrence of a workgroup index in start order)
    full = []
    for k, v in launches.items():
        if k == 0:
            continue
        v = sorted(v)
        if len(v) % nwg:
            continue
        # (the scratch sets count their tickets separately: up to four launches -- sets 0..3 -- carry the same tag and run close
        # together; the i-th occurrence of a row group in start order goes to the i-th of them)
        parts, seen = [[] for _ in range(len(v) // nwg)], {}
        for w in v:
            i = seen.get(w[2], 0)
            seen[w[2]] = i + 1
            if i < len(parts):
                parts[i].append(w)
        full += [p_ for p_ in parts if len(p_) == nwg]
    if a.mode == "alone":      # untagged: cut the one list into launches of nwg workgroups by time
        allw = sorted(w for v in launches.values() for w in v)
        full = [allw[i:i + nwg] for i in range(0, len(allw) - nwg + 1, nwg)]
    full = full[len(full) // 4:]
    def us(x): return x / 100.0
    span = [us(max(w[1] for w in L) - L[0][0]) for L in full]
    start50 = [us(sorted(w[0] for w in L)[len(L) // 2] - L[0][0]) for L in full]
    start90 = [us(sorted(w[0] for w in L)[int(len(L) * 0.9)] - L[0][0]) for L in full]
    startmax = [us(max(w[0] for w in L) - L[0][0]) for L in full]
    run50 = [us(st.median(w[1] - w[0] for w in L)) for L in full]
    run90 = [us(sorted(w[1] - w[0] for w in L)[int(len(L) * 0.9)]) for L in full]
    runmax = [us(max(w[1] - w[0] for w in L)) for L in full]
    m = st.median
    # the per-pixel launch the row scan waited for ended at the latest stamp any of its workgroups saw that lies BEFORE the launch's first
    # start (the next per-pixel launch stamps later ones while the row scan is still trickling in)
    dep = []
    for L in full:
        first = L[0][0]
        seen = [w[3] for w in L if 0 < w[3] <= first]
        if seen:
            dep.append(us(first - max(seen)))
    print(f"{a.workload} {a.mode}: {len(full)} launches x {nwg} workgroups; median over launches, us: first start -> last end {m(span):.1f}; "
          f"workgroup START after the first one's: p50 {m(start50):.1f}, p90 {m(start90):.1f}, last {m(startmax):.1f}; "
          f"one workgroup RUNS: p50 {m(run50):.1f}, p90 {m(run90):.1f}, longest {m(runmax):.1f}; "
          + (f"first workgroup starts {m(dep):.1f} us (p90 {sorted(dep)[int(0.9 * len(dep))]:.1f}) behind the END of its per-pixel launch's last workgroups ({len(dep)} launches)" if dep else "no per-pixel end stamps"))
    leg.close()


if __name__ == "__main__":
    main()
