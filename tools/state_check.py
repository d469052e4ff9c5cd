#!/usr/bin/env python3
"""Long-run FULL-MODEL parity of the pipelined path at a real frame size (tools/state_check.py [--rows R --cols C
--frames N --pool P]): synthetic SURVEY 8d frames (flickering pixels, moving discs) go through
oatgpu_track_enqueue / collect (ring 4) for N frames; then the device's exported model -- counters, weights,
variances, means -- must equal the oracle's bit for bit, and every position of the run must be the oracle's.
The short sequences of tests/test_gpu_parity.py compare the model on small frames; this is the same check where
the kernel runs at full occupancy for hundreds of frames."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(rows, cols, frames, pool, alpha=0.01, streams=1, audited=6, dense=False, log=print, fusion=None):
    import oat_amd
    import oracle_lib as O
    from oat_amd.synth import SyntheticStream, disc_hsv_window
    if dense:
        # bench.py's dense model: every pixel cycles through five well separated colours (own phase per pixel) --
        # five live modes everywhere, the library switches the slot-1..4 loads to the streaming cache policy
        # (the NTLD instantiation of the per-pixel kernel) after its first density probes
        rng = np.random.default_rng(0xD0)
        table = np.array([[20, 30, 40], [90, 200, 60], [200, 60, 120], [240, 240, 230], [40, 130, 220]], np.int16)
        phase = rng.integers(0, 5, (streams, rows, cols))
        pool = max(5, pool // 5 * 5)
        fr = [[np.clip(table[(phase[s] + t) % 5] + rng.integers(-5, 6, (rows, cols, 3)), 0, 255).astype(np.uint8)
               for s in range(streams)] for t in range(pool)]
    else:
        st = [SyntheticStream(rows, cols, s, n_discs=2) for s in range(streams)]
        fr = [[st[s].frame(9 * t, with_discs=t > 0) for s in range(streams)] for t in range(pool)]
    hp = oat_amd.HotPath(rows, cols, n_streams=streams, adaptation_coeff=alpha, erode=3, dilate=7,
                         area=(20.0, 1e5), ring_depth=4, **disc_hsv_window())
    if fusion:
        hp.set_fusion(fusion)
    got = []
    t0 = time.perf_counter()
    for t in range(frames):
        hp.enqueue(fr[t % pool])
        if hp.outstanding() >= 4:
            got.append(hp.collect())
    while hp.outstanding():
        got.append(hp.collect())
    t_gpu = time.perf_counter() - t0
    # ... and `audited` more frames through the traffic-audit instantiation of the same kernel (oatgpu_traffic_audit):
    # it must leave the model exactly as the product kernel would
    if audited:
        hp.traffic_audit(True)
        for t in range(frames, frames + audited):           # same ring pattern: the audited launches take two frames too
            hp.enqueue(fr[t % pool])
            if hp.outstanding() >= 4:
                got.append(hp.collect())
        while hp.outstanding():
            got.append(hp.collect())
        hp.traffic_read()
        hp.traffic_audit(False)
    frames += audited
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=7, min_area=20.0,
                     max_area=1e5)
    bad = 0
    bad_at = []
    for s in range(streams):
        orc = O.Mog2(rows, cols, 3)
        for t in range(frames):
            w = O.chain_step(orc, fr[t % pool][s], alpha, p, nthreads=min(os.cpu_count() or 1, 32))[0]
            g = got[t][s]
            if g.position_valid != w["valid"] or (w["valid"] and (g.a00, g.a10, g.a01) != (w["a00"], w["a10"], w["a01"])):
                bad += 1
                bad_at.append(t)
        nm_g, w_g, v_g, m_g, _ = hp.mog_state(s)
        nm_o, w_o, v_o, m_o = orc.state()
        live = np.arange(w_o.shape[1])[None, :] < nm_o[:, None]
        d = dict(count=int((nm_g != nm_o).sum()), weight=int((w_g[live] != w_o[live]).sum()),
                 variance=int((v_g[live] != v_o[live]).sum()), mean=int((m_g[live] != m_o[live]).sum()))
        log(f"stream {s}: {frames} frames {cols}x{rows}: position mismatches {bad}, model differences {d} "
            f"(of {int(live.sum())} live modes); GPU {t_gpu:.2f} s" + (f"; positions differ at frames {bad_at[:12]}" if bad_at else ""))
        bad += sum(d.values())
    hp.close()
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--pool", type=int, default=24)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--audited", type=int, default=6)
    ap.add_argument("--dense", action="store_true")
    ap.add_argument("--fusion", type=int, default=0, help="oatgpu_set_fusion (0 = the library's default)")
    a = ap.parse_args()
    sys.exit(1 if run(a.rows, a.cols, a.frames, a.pool, streams=a.streams, audited=a.audited, dense=a.dense, fusion=a.fusion or None) else 0)
