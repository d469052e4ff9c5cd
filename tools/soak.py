#!/usr/bin/env python3
"""Soak test (GPU box): many thousands of pipelined frames through the fused track path -- several
camera streams, deep ring, position filter on -- with EVERY result compared against the CPU oracle
chain (tests/oracle_lib.py).  Small frames on purpose: launches and cross-stream hand-offs are then
as dense as they get, which is what a race in the ring / event / ticket logic would need.

    python tools/soak.py [--frames 20000] [--rows 240 --cols 320] [--streams 3] [--ring 6]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def g_xy(g):
    return g.x, g.y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=20000)
    ap.add_argument("--rows", type=int, default=240)
    ap.add_argument("--cols", type=int, default=320)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--ring", type=int, default=6)
    ap.add_argument("--pool", type=int, default=97)
    ap.add_argument("--no-kalman", action="store_true",
                    help="position filter off: with ONE stream of 4 MP and more the steps then take the early blob dispatch and "
                         "the one-wave-a-workgroup per-pixel kernel (the library's default there, r05)")
    ap.add_argument("--threads", type=int, default=8, help="row workers of the oracle chain")
    args = ap.parse_args()

    import torch
    import oat_amd
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream, disc_hsv_window

    rows, cols, n = args.rows, args.cols, args.streams
    sts = [SyntheticStream(rows, cols, 1000 + s, n_discs=1 + s % 2) for s in range(n)]
    # a pool of distinct frame sets; some without discs so that the filter coasts and drops
    pool = [np.stack([st.frame(t, with_discs=(t > 0 and t % 23 not in (7, 8, 9, 10, 11, 12))) for st in sts])
            for t in range(args.pool)]
    dev = torch.device("cuda:0")
    dpool = [torch.from_numpy(p).to(dev) for p in pool]
    torch.cuda.synchronize()

    kal = dict(dt=0.01, timeout=0.04, sigma_accel=25.0, sigma_noise=1.0)
    hp = oat_amd.HotPath(rows, cols, n_streams=n, ring_depth=args.ring, adaptation_coeff=0.01, erode=3, dilate=5,
                         area=(10.0, 1e5), **disc_hsv_window())
    if not args.no_kalman:
        hp.set_kalman(True, **kal)
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=5,
                     min_area=10.0, max_area=1e5)
    orc = [O.Mog2(rows, cols, 3) for _ in range(n)]
    okal = [O.Kalman(**kal) for _ in range(n)]

    hp.set_fusion(2)       # device frames: two frames a launch is opt-in (oatgpu_set_fusion); the pool stays untouched
    order = [(t * 7 + t // args.pool) % args.pool for t in range(args.frames)]
    got = []
    t0 = time.perf_counter()
    rng = np.random.default_rng(1)
    for t in range(args.frames):
        # irregular drain pattern: sometimes collect early, sometimes let the ring fill up
        while hp.outstanding() == args.ring or (hp.outstanding() and rng.random() < 0.15):
            got.append(hp.collect())
        hp.enqueue_dev(dpool[order[t]].data_ptr())
    while hp.outstanding():
        got.append(hp.collect())
    t_gpu = time.perf_counter() - t0

    bad = tracked = found = 0
    t0 = time.perf_counter()
    for t in range(args.frames):
        for s in range(n):
            d, _ = O.chain_step(orc[s], pool[order[t]][s], 0.01, p, nthreads=args.threads)
            k = okal[s].filter(d["valid"], d["x"], d["y"]) if not args.no_kalman else dict(
                position_valid=d["valid"], x=g_xy(got[t][s])[0], y=g_xy(got[t][s])[1], vx=got[t][s].vx, vy=got[t][s].vy)
            g = got[t][s]
            ok = (g.raw_valid == d["valid"] and g.position_valid == k["position_valid"]
                  and (g.x, g.y, g.vx, g.vy) == (k["x"], k["y"], k["vx"], k["vy"])
                  and (not d["valid"] or (g.a00, g.a10, g.a01, g.raw_x, g.raw_y) == (d["a00"], d["a10"], d["a01"], d["x"], d["y"])))
            bad += not ok
            found += d["valid"]
            tracked += k["position_valid"]
            if not ok and bad <= 5:
                print("MISMATCH frame", t, "stream", s, g, d, k)
    t_cpu = time.perf_counter() - t0
    print(f"soak: {args.frames} frames x {n} streams {cols}x{rows}, ring {args.ring}: {bad} mismatches; "
          f"{found} detections, {tracked} tracked positions; GPU {t_gpu:.1f} s ({args.frames * n / t_gpu:.0f} fps), "
          f"oracle {t_cpu:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
