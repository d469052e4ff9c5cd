#!/usr/bin/env python3
"""Bisect a difference between one frame a launch and two frames a launch (oatgpu_set_fusion): final model of both
forms after T frames of a scenario, for growing T; prints where they first part and in which plane / slot."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oat_amd  # noqa: E402
from test_gpu_parity import _noisy_sequence  # noqa: E402


def run(frames, rates, fusion, ring, drains=()):
    n, rows, cols = frames[0].shape[:3]
    win = dict(h_thresh=(100, 125), s_thresh=(150, 256), v_thresh=(100, 256))
    hp = oat_amd.HotPath(rows, cols, n_streams=n, ring_depth=ring, erode=0, dilate=3, area=(5.0, 1e6), **win)
    hp.set_fusion(fusion)
    for t, (f, lr) in enumerate(zip(frames, rates)):
        hp.learning_coeff_ = lr
        hp.enqueue(list(f))
        if t in drains:
            while hp.outstanding():
                hp.collect()
        elif hp.outstanding() >= ring:
            hp.collect()
    while hp.outstanding():
        hp.collect()
    st = [hp.mog_state(s) for s in range(n)]
    hp.close()
    return st


def first_split(name, rates, drains=(), ring=3, seed=77, noise=9):
    rng = np.random.default_rng(seed)
    frames = _noisy_sequence(rng, 2, 48, 130, 3, len(rates), noise=noise)
    for T in range(1, len(rates) + 1):
        a = run(frames[:T], rates[:T], 1, ring, drains)
        b = run(frames[:T], rates[:T], 2, ring, drains)
        for s in range(2):
            nm = a[s][0]
            live = np.arange(5)[None, :] < nm[:, None]
            bad = {}
            if not (a[s][0] == b[s][0]).all():
                bad["nmodes"] = int((a[s][0] != b[s][0]).sum())
            for nme, i in (("w", 1), ("v", 2)):
                d = (a[s][i] != b[s][i]) & ~(np.isnan(a[s][i]) & np.isnan(b[s][i])) & live
                if d.any():
                    bad[nme] = (int(d.sum()), np.argwhere(d)[:3].tolist())
            d = ((a[s][3] != b[s][3]) & ~(np.isnan(a[s][3]) & np.isnan(b[s][3]))).any(-1) & live
            if d.any():
                px, k = np.argwhere(d)[0]
                bad["m"] = (int(d.sum()), np.argwhere(d)[:3].tolist(), a[s][3][px].tolist(), b[s][3][px].tolist(),
                            a[s][1][px].tolist(), a[s][2][px].tolist(), b[s][2][px].tolist(), int(nm[px]))
            if bad:
                print(f"{name}: first difference after T={T} frames (rates {rates[max(0, T - 3):T]}), stream {s}: {bad}")
                return
    print(f"{name}: identical through {len(rates)} frames")


if __name__ == "__main__":
    R = [-1.0, -1.0, -1.0, 0.01, 0.3, 0.0, 0.0, 0.05, 1.0, 0.05, 0.05, -1.0, 0.2, 0.2, 1.5, -1.0, 0.01, 0.01, 0.01,
         0.5, 0.0, 0.1, 0.1]
    first_split("test rates, no drains", R)
    first_split("test rates, drains", R, drains={0, 3, 4, 9, 10, 11, 17})
    first_split("constant 0.3", [0.3] * 16)
    first_split("constant 0.02", [0.02] * 16)
    first_split("auto", [-1.0] * 16)
    first_split("alternating 0.3 / 0.01", [0.3, 0.01] * 8)
    first_split("with zeros", [0.05, 0.05, 0.0, 0.0, 0.05, 0.0, 0.05, 0.05, 0.0, 0.05, 0.05])
    first_split("with re-initialisation", [0.05, 0.05, 0.05, 1.0, 0.05, 0.05, 0.05, 0.05, 1.0, 0.05, 0.05])
