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
    buf = (C.c_longlong * (N * 4))()
    n = lib.oatgpu_debug_rs_timing(buf, N)
    launches = defaultdict(list)
    for r in range(n):
        t0, t1, wg, tag = buf[r * 4], buf[r * 4 + 1], buf[r * 4 + 2], buf[r * 4 + 3]
        launches[tag].append((t0, t1, wg))
    nwg = (leg.wl["rows"] + 3) // 4
    # (early order: the two frames of a step carry the same ticket on their two scratch sets -- 2 x nwg entries a tag, run
    # side by side on two streams: split by first / second occurrence of a workgroup index in start order)
    full = []
    for k, v in launches.items():
        if k == 0:
            continue
        v = sorted(v)
        if len(v) == nwg:
            full.append(v)
        elif len(v) == 2 * nwg:
            seen, one, two = set(), [], []
            for w in v:
                (two if w[2] in seen else one).append(w)
                seen.add(w[2])
            if len(one) == nwg and len(two) == nwg:
                full += [one, two]
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
    print(f"{a.workload} {a.mode}: {len(full)} launches x {nwg} workgroups; median over launches, us: first start -> last end {m(span):.1f}; "
          f"workgroup START after the first one's: p50 {m(start50):.1f}, p90 {m(start90):.1f}, last {m(startmax):.1f}; "
          f"one workgroup RUNS: p50 {m(run50):.1f}, p90 {m(run90):.1f}, longest {m(runmax):.1f}")
    leg.close()


if __name__ == "__main__":
    main()
