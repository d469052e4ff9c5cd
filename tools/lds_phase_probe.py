#!/usr/bin/env python3
"""In-kernel phase times of k_blob_lds, beside the per-pixel kernel (pipelined run) and alone (synchronous steps):

    OATGPU_LIB=build/variants/liboatgpu_ldst.so python tools/lds_phase_probe.py [--workload 4k1] [--mode load|alone]

Needs a -DOATGPU_LDS_TIMING build (make variant NAME=ldst DEFS=-DOATGPU_LDS_TIMING): its k_blob_lds stamps the 100 MHz
wall clock between its phases into a ring in device memory, which oatgpu_debug_lds_timing copies out.  Run under
rocprofv3 --kernel-trace (tools/lds_phase_probe.sh) the same launches' dispatch durations stand beside the totals:
duration - total = what a launch waits for before its workgroup runs.
"""
import argparse
import ctypes as C
import os
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="4k1")
    ap.add_argument("--mode", default="load", choices=["load", "alone"])
    ap.add_argument("--steps", type=int, default=600)
    a = ap.parse_args()
    import torch
    import bench
    leg = bench.Leg(a.workload, 0, 0, pool=24)
    leg.init()
    leg.age(200)
    lib = leg.hp.lib
    lib.oatgpu_debug_lds_timing.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    lib.oatgpu_debug_lds_timing.restype = C.c_int
    if a.mode == "load":
        leg.run(a.steps)
    else:
        for i in range(a.steps):
            leg.hp.track_dev(leg.pool[leg.pool_index(leg.step + i)].data_ptr())
    torch.cuda.synchronize()
    buf = (C.c_longlong * (4096 * 10))()
    n = lib.oatgpu_debug_lds_timing(buf, 4096)
    rows = [[(buf[r * 10 + q + 1] - buf[r * 10 + q]) / 100.0 for q in range(8)] + [(buf[r * 10 + 8] - buf[r * 10]) / 100.0]
            for r in range(n)]
    rows = rows[-min(len(rows), a.steps):]
    names = ["A rows", "B run list", "C unions", "D flatten", "E setup", "E own loop", "E barrier wait", "F select", "total"]
    print(f"{a.workload} {a.mode}: {len(rows)} launches of k_blob_lds, median us: "
          + ", ".join(f"{nm} {st.median(r[i] for r in rows):.1f}" for i, nm in enumerate(names))
          + f"; total p90 {sorted(r[8] for r in rows)[int(0.9 * len(rows))]:.1f}, max {max(r[8] for r in rows):.1f}")
    leg.close()


if __name__ == "__main__":
    main()
