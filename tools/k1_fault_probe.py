#!/usr/bin/env python3
"""Single-launch differential probe for the per-pixel kernel K1 (tools/k1_fault_probe.py [--rows R --cols C]).

Round 2 saw an instantiation of k_mog_fused (the traffic-audit one, two frames a launch) leave a model that was
wrong and different from run to run while the product instantiations stayed bit-exact.  This probe isolates ONE
launch: a model aged by the pipelined path is exported (S0); then, from S0 every time,
  * the product two-frame launch on frames (A, B)          -> R   (checked against the oracle)
  * the audited two-frame launch on the same frames, N times -> Q_i
  * the same with one frame a launch (fusion 1)
and every Q_i is compared with R field by field.  For the pixels that differ it prints which lanes of the wave they
sit in, which fields differ, the values, and whether the set is the same from repeat to repeat -- the fingerprint
needed to tell a miscompiled path from a missing wait from a hardware hazard.  OATGPU_LIB selects the library
(make variant builds)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def diff_report(tag, R, Q, S0, fa, fb, cols, log, verbose=4):
    nm_r, w_r, v_r, m_r = R
    nm_q, w_q, v_q, m_q = Q
    d_nm = nm_r != nm_q
    d_w = (w_r.view(np.uint32) != w_q.view(np.uint32)).any(1)
    d_v = (v_r.view(np.uint32) != v_q.view(np.uint32)).any(1)
    d_m = (m_r.view(np.uint32) != m_q.view(np.uint32)).any((1, 2))
    bad = d_nm | d_w | d_v | d_m
    n = int(bad.sum())
    if not n:
        log(f"  {tag}: identical")
        return set()
    idx = np.flatnonzero(bad)
    lanes = np.bincount(idx % 64, minlength=64)
    waves = np.unique(idx // 64).size
    top = ", ".join(f"{l}:{c}" for l, c in sorted(enumerate(lanes), key=lambda t: -t[1])[:6] if c)
    log(f"  {tag}: {n} pixels differ in {waves} waves (count {int(d_nm.sum())}, weight {int(d_w.sum())}, "
        f"variance {int(d_v.sum())}, mean {int(d_m.sum())}); lanes {top}; rows {idx[0] // cols}..{idx[-1] // cols}")
    for p in idx[:verbose]:
        log(f"    px {p} (lane {p % 64}, x {p % cols}, y {p // cols}) A={fa.reshape(-1, 3)[p].tolist()} B={fb.reshape(-1, 3)[p].tolist()}")
        log(f"      S0  n={S0[0][p]} w={S0[1][p].tolist()} v={S0[2][p].tolist()} m0={S0[3][p][:, 0].tolist()}")
        log(f"      ref n={nm_r[p]} w={w_r[p].tolist()} v={v_r[p].tolist()} m0={m_r[p][:, 0].tolist()}")
        log(f"      got n={nm_q[p]} w={w_q[p].tolist()} v={v_q[p].tolist()} m0={m_q[p][:, 0].tolist()}")
    return set(idx.tolist())


def run(rows, cols, age, reps, alpha=0.01, check_oracle=True, log=print):
    import oat_amd
    from oat_amd import ffi
    from oat_amd.synth import SyntheticStream, disc_hsv_window
    log(f"library: {ffi.lib_path()}")
    st = SyntheticStream(rows, cols, 0, n_discs=2)
    fr = [st.frame(9 * t, with_discs=t > 0) for t in range(age + 2)]
    hp = oat_amd.HotPath(rows, cols, n_streams=1, adaptation_coeff=alpha, erode=3, dilate=7, area=(20.0, 1e5),
                         ring_depth=4, **disc_hsv_window())
    for t in range(age):
        hp.enqueue([fr[t]])
        if hp.outstanding() >= 4:
            hp.collect()
    while hp.outstanding():
        hp.collect()
    S0 = hp.mog_state(0)
    fa, fb = fr[age], fr[age + 1]
    total_bad = 0

    def launch(fusion, audited):
        hp.set_mog_state(*S0[:4], S0[4], stream=0)
        hp.set_fusion(fusion)
        if audited:
            hp.traffic_audit(True)
        hp.enqueue([fa])
        hp.enqueue([fb])
        hp.collect()
        hp.collect()
        if audited:
            hp.traffic_read()
            hp.traffic_audit(False)
        return hp.mog_state(0)[:4]

    for fusion in (2, 1):
        R = launch(fusion, False)
        if check_oracle:
            import oracle_lib as O
            orc = O.Mog2(rows, cols, 3)
            orc.set_state(S0[0], S0[1], S0[2], S0[3], S0[4])
            for f in (fa, fb):
                orc.apply(f, alpha)
            nm_o, w_o, v_o, m_o = orc.state()
            live = np.arange(w_o.shape[1])[None, :] < nm_o[:, None]
            dd = int((R[0] != nm_o).sum()) + int((R[1][live] != w_o[live]).sum()) + int((R[2][live] != v_o[live]).sum()) + int((R[3][live] != m_o[live]).sum())
            log(f"fusion {fusion}: product launch vs oracle: {dd} differences")
            total_bad += dd
        R2 = launch(fusion, False)
        s = diff_report(f"fusion {fusion} product, repeat", R, R2, S0, fa, fb, cols, log)
        total_bad += len(s)
        sets = []
        for i in range(reps):
            Q = launch(fusion, True)
            sets.append(diff_report(f"fusion {fusion} AUDITED #{i}", R, Q, S0, fa, fb, cols, log, verbose=4 if i == 0 else 0))
        if any(sets):
            common = set.intersection(*sets) if all(sets) else set()
            union = set.union(*sets)
            log(f"fusion {fusion}: audited differing pixels: union {len(union)}, common to all repeats {len(common)}")
            total_bad += len(union)
    hp.close()
    return total_bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--age", type=int, default=40)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    sys.exit(1 if run(a.rows, a.cols, a.age, a.reps, check_oracle=not a.no_oracle) else 0)
