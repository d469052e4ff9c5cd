#!/usr/bin/env python3
"""Does the dense per-pixel launch's time depend on WHERE its model landed?  (the one-frame launch ran
264 us in one process and 293 us in another on the same box, its two-frame sibling 264-269 us in both.)

    OATGPU_MEASURE_PY=1 OATGPU_LIB=build/variants/liboatgpu_meas.so python tools/dense_placement_probe.py [--rounds 6]

Every round: a filler allocation of a different size shifts the addresses, then a fresh dense 4K context per form (one / two
frames a launch) is warmed (1 200 steps) and timed (300 steps, HIP events around the kernel); printed with the model's
device address (measurement builds export it)."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--forms", default="1:1:64,2:1:64,1:0:64,2:0:64,1:1:256,1:0:256",
                    help="frames a launch : early blob order : threads of a per-pixel workgroup, comma separated")
    a = ap.parse_args()
    import torch
    import bench
    torch.cuda.set_device(0)
    keep = []
    for r in range(a.rounds):
        keep.append(torch.empty((r * 37 + 5) << 20, dtype=torch.uint8, device="cuda"))       # shifts what comes next
        for nf, early, wg in [tuple(int(x) for x in f.split(":")) for f in a.forms.split(",")]:
            leg = bench.Leg("4k1", 0, 0, dense=True, pool=10)
            leg.hp.set_fusion(nf)
            leg.hp.set_k1_workgroup(wg)
            leg.hp.set_early_blob(bool(early))
            lib = leg.hp.lib
            addr = (C.c_ulonglong * 4)()
            try:
                lib.oatgpu_debug_addresses.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
                lib.oatgpu_debug_addresses(leg.hp.ctx, addr)
            except AttributeError:
                pass
            tr = bench.timed_run(leg, 300, 1200, lambda: (leg.hp.synchronize(), torch.cuda.synchronize()), 2, age_frames=60, export=False, min_ms=0.0)
            k1 = bench.k1_ms(tr["prof"])[0]
            print(f"round {r} frames/launch {nf} early {early} wg {wg}: k_mog_fused {k1 * 1e3:7.1f} us  step {tr['block_s'] / 300 * 1e6:7.1f} us   model @ {addr[0]:#x} (mod 2 MiB {addr[0] % (2 << 20):#x}, mod 1 GiB {addr[0] % (1 << 30):#x}), "
                  f"counters @ {addr[1]:#x}, thr @ {addr[2]:#x}", flush=True)
            leg.close()
            del leg
    del keep


if __name__ == "__main__":
    main()
