#!/usr/bin/env python3
"""How the per-step time of a long run evolves (model age): windows of 400 steps, K1 / back-half event times, the
number of threshold-mask pixels and the model's mode histogram.  python tools/drift_probe.py [workload] [windows]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "4k1"
leg = bench.Leg(name, 0, 0)
leg.init()
for w in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    leg.hp.profile(4)
    leg.hp.profile_reset()
    t0 = time.perf_counter()
    leg.run(400)
    leg.hp.synchronize()
    el = time.perf_counter() - t0
    p = leg.hp.profile_read()
    leg.hp.profile(0)
    n = max(p["steps"], 1)
    thr = int((leg.hp.read_mask(0) != 0).sum())
    nm, wgt, _, _, _ = leg.hp.mog_state(0)
    live = ((wgt != 0) & (np.arange(wgt.shape[1])[None, :] < nm[:, None])).sum(1)
    print(f"frames {leg.step:6d}  step {el / 400 * 1e6:7.1f} us  K1 {bench.k1_ms(p)[0] * 1e3:7.1f}  blob {p['blob_ms'] / n * 1e3:7.1f}  "
          f"thr px {thr:7d}  modesUsed {nm.mean():.3f} live {live.mean():.3f}  hist {np.bincount(nm, minlength=6)[:6].tolist()}", flush=True)
