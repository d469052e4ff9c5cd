#!/usr/bin/env python3
"""Differential fuzzing on the GPU box: random geometries, stream counts, ring depths, detector and
filter settings and frame contents through the fused track path (synchronous, pipelined from device
memory and pipelined from host memory, chosen at random), every result compared with the CPU
oracle chain; morphology masks are compared too on the synchronous configurations.

    python tools/fuzz.py [--configs 300] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_frames(rng, rows, cols, n, nframes, channels):
    base = rng.integers(30, 200, (n, rows, cols, channels)).astype(np.int16)
    out = []
    nblobs = int(rng.integers(0, 4))
    blobs = [(rng.integers(0, rows), rng.integers(0, cols), rng.integers(1, max(2, min(rows, cols) // 3)),
              rng.integers(-3, 4), rng.integers(-3, 4)) for _ in range(nblobs)]
    colour = np.array([255, 64, 0][:channels] if channels == 3 else [250])
    for t in range(nframes):
        f = base + rng.integers(-10, 11, base.shape)
        if rng.random() < 0.1:
            f += rng.integers(-60, 61)
        for s in range(n):
            for (y, x, r, vy, vx) in blobs:
                yy, xx = (y + vy * t + 7 * s) % rows, (x + vx * t + 11 * s) % cols
                f[s, max(0, yy - r):yy + r + 1, max(0, xx - r):xx + r + 1] = colour
        if rng.random() < 0.05:
            f[:] = 0
        out.append(np.clip(f, 0, 255).astype(np.uint8).reshape((n, rows, cols, 3) if channels == 3 else (n, rows, cols)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1, help="run this configuration alone (the others still draw their random numbers) and say what differs")
    args = ap.parse_args()
    import torch
    import oat_amd
    import oracle_lib as O

    rng = np.random.default_rng(args.seed)
    dev = torch.device("cuda:0")
    checked = bad = 0
    for ci in range(args.configs):
        rows = int(rng.choice([1, 2, 3, 7, 31, 64, 65, 97, 128, 200, 255]))
        cols = int(rng.choice([1, 2, 5, 63, 64, 65, 127, 128, 129, 300, 511, 640, 4100]))
        if rows * cols > 400000:
            rows = max(1, 400000 // cols)
        n = int(rng.choice([1, 1, 2, 3, 5]))
        channels = int(rng.choice([3, 3, 1]))
        ring = int(rng.integers(1, 7))
        nframes = int(rng.integers(3, 14))
        lr = float(rng.choice([0.0, 0.01, 0.2, -1.0]))
        e, d = int(rng.choice([0, 0, 2, 3, 5, 8, 21])), int(rng.choice([0, 0, 2, 3, 7, 10, 33]))
        area = (float(rng.choice([0.0, 0.5, 4.0, 30.0])), float(rng.choice([50.0, 1e4, 1e9])))
        if channels == 3:
            win = dict(h_thresh=(int(rng.choice([0, 100])), int(rng.choice([125, 256]))),
                       s_thresh=(int(rng.choice([0, 150])), 256), v_thresh=(int(rng.choice([1, 100])), 256))
        else:
            win = dict(h_thresh=(int(rng.choice([1, 200])), 256))
        use_kal = rng.random() < 0.4
        kal = dict(dt=float(rng.choice([0.01, 0.02])), timeout=float(rng.choice([0.0, 0.03, 0.2])),
                   sigma_accel=float(rng.choice([5.0, 40.0])), sigma_noise=float(rng.choice([0.0, 1.0])))
        mode = str(rng.choice(["sync", "dev", "host"]))
        frames = make_frames(rng, rows, cols, n, nframes, channels)
        restore = int(rng.integers(0, 2))          # both readings of MOG2Invoker's mode count (oracle/mog2.c)
        if args.only >= 0 and ci != args.only:
            continue

        hp = oat_amd.HotPath(rows, cols, n_streams=n, ring_depth=ring, channels=channels, adaptation_coeff=lr,
                             erode=e, dilate=d, area=area, mog_restore_nmodes=restore, **win)
        if use_kal:
            hp.set_kalman(True, **kal)
        pkw = dict(erode=e, dilate=d, min_area=area[0], max_area=area[1])
        if channels == 3:
            pkw.update(h_lo=win["h_thresh"][0], h_hi=win["h_thresh"][1], s_lo=win["s_thresh"][0], s_hi=256,
                       v_lo=win["v_thresh"][0], v_hi=256)
        else:
            pkw.update(h_lo=win["h_thresh"][0], h_hi=256)
        p = O.hsv_params(**pkw)
        orc = [O.Mog2(rows, cols, channels, params=dict(restore_nmodes=restore)) for _ in range(n)]
        okal = [O.Kalman(**kal) for _ in range(n)]

        got, masks = [], []
        if mode == "sync":
            for f in frames:
                got.append(hp.track(list(f)))
                masks.append([hp.read_mask(1, stream=s) for s in range(n)])
        else:
            if mode == "dev" and (len(frames) + n + ring) % 2 == 0:
                hp.set_fusion(2)               # opt in: two frames a launch for device frames too (the buffers stay untouched)
            bufs = [torch.from_numpy(f).to(dev) for f in frames] if mode == "dev" else None
            torch.cuda.synchronize()
            for t, f in enumerate(frames):
                if hp.outstanding() == ring:
                    got.append(hp.collect())
                if mode == "dev":
                    hp.enqueue_dev(bufs[t].data_ptr())
                else:
                    hp.enqueue(list(f))
            while hp.outstanding():
                got.append(hp.collect())

        ok = True
        for t, f in enumerate(frames):
            for s in range(n):
                dres, thr = O.chain_step(orc[s], f[s], lr, p)
                g = got[t][s]
                if use_kal:
                    k = okal[s].filter(dres["valid"], dres["x"], dres["y"])
                    ok &= (g.position_valid == k["position_valid"] and (g.x, g.y, g.vx, g.vy) == (k["x"], k["y"], k["vx"], k["vy"]))
                    ok &= g.raw_valid == dres["valid"] and (not dres["valid"] or (g.raw_x, g.raw_y) == (dres["x"], dres["y"]))
                else:
                    ok &= g.position_valid == dres["valid"]
                    ok &= (not dres["valid"]) or ((g.a00, g.a10, g.a01, g.first_pixel, g.x, g.y) ==
                                                  (dres["a00"], dres["a10"], dres["a01"], dres["first_pixel"], dres["x"], dres["y"]))
                if masks:
                    ok &= bool((masks[t][s] == thr).all())
                if args.only >= 0:
                    print(f"frame {t} stream {s}: got valid {g.position_valid} a00 {g.a00} a10 {g.a10} a01 {g.a01} first {g.first_pixel} | want {dres}")
                checked += 1
        if not ok:
            bad += 1
            print("MISMATCH config", ci, dict(rows=rows, cols=cols, n=n, channels=channels, ring=ring, nframes=nframes, lr=lr,
                                             e=e, d=d, area=area, win=win, kal=kal if use_kal else None, mode=mode))
        hp.close()
    print(f"fuzz: {args.configs} configurations, {checked} stream-frames checked, {bad} configurations with a mismatch")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
