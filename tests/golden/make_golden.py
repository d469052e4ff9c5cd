#!/usr/bin/env python3
"""Writes the hand-derived golden vectors for the hot path (tests/golden/*.json).

Nothing here imports the oracle or the product: every expected value is either
typed in from a hand derivation (comment beside it) or produced by the tiny
independent float32 restatement of one MOG2 pixel below (`mog2_pixel_trace`),
written from the algorithm description in SURVEY.md 8a row 1 -- so the C oracle
(oracle/mog2.c) and this script are two separate restatements that must agree.

The reference (jonnew/Oat) holds NO golden data for this path,
so these are this repo's own known answers: PARITY UNPINNED.
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def mog2_pixel_trace(pixels, rates, nmix=5, restore=True):
    """One pixel, 3 channels.  pixels: list of (b,g,r); rates: learning rate per frame.
    restore: the mode count is set back to its value at entry after the renormalisation
    (MOG2Invoker's `nmodes = nNewModes;`, see oracle/mog2.c "Mode count"); False = pruning shrinks it.
    Returns per frame: mask, nmodes, weights, variances, means (python floats of f32)."""
    Tb, TB, Tg = f32(16), f32(0.9), f32(9)
    varInit, varMin, varMax, tau = f32(15), f32(4), f32(75), f32(0.5)
    CT = np.float32(0.05)
    w = [f32(0)] * nmix
    var = [f32(0)] * nmix
    mu = [[f32(0)] * 3 for _ in range(nmix)]
    nmodes = 0
    nframes = 0
    out = []
    for px, rate in zip(pixels, rates):
        if nframes == 0 or rate >= 1:
            w = [f32(0)] * nmix; var = [f32(0)] * nmix
            mu = [[f32(0)] * 3 for _ in range(nmix)]; nmodes = 0; nframes = 0
        nframes += 1
        lr = rate if (rate >= 0 and nframes > 1) else 1.0 / min(2 * nframes, 500)
        alphaT = f32(lr)
        prune = f32(-lr * float(CT))          # product in double, then to float
        alpha1 = f32(1) - alphaT
        x = [f32(v) for v in px]
        background = False
        fits = False
        total = f32(0)
        mode = 0
        n_entry = nmodes
        while mode < nmodes:
            weight = f32(f32(alpha1 * w[mode]) + prune)
            swaps = 0
            if not fits:
                v = var[mode]
                d = [f32(mu[mode][c] - x[c]) for c in range(3)]
                dist2 = f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))
                if total < TB and dist2 < f32(Tb * v):
                    background = True
                if dist2 < f32(Tg * v):
                    fits = True
                    weight = f32(weight + alphaT)
                    k = f32(alphaT / weight)
                    for c in range(3):
                        mu[mode][c] = f32(mu[mode][c] - f32(k * d[c]))
                    vn = f32(v + f32(k * f32(dist2 - v)))
                    vn = varMin if vn < varMin else vn; vn = varMax if vn > varMax else vn     # OpenCV MAX / MIN macros: a NaN stays NaN
                    var[mode] = f32(vn)
                    i = mode
                    while i > 0:
                        if weight < w[i - 1]:
                            break
                        swaps += 1
                        w[i], w[i - 1] = w[i - 1], w[i]
                        var[i], var[i - 1] = var[i - 1], var[i]
                        mu[i], mu[i - 1] = mu[i - 1], mu[i]
                        i -= 1
            if weight < -prune:
                weight = f32(0)
                nmodes -= 1
            w[mode - swaps] = weight
            total = f32(total + weight)
            mode += 1
        with np.errstate(divide="ignore"):
            inv = f32(f32(1) / total)
        for m in range(nmodes):
            w[m] = f32(w[m] * inv)
        if restore:
            nmodes = n_entry
        if not fits and alphaT > 0:
            if nmodes == nmix:
                mode = nmix - 1
            else:
                mode = nmodes
                nmodes += 1
            if nmodes == 1:
                w[mode] = f32(1)
            else:
                w[mode] = alphaT
                for i in range(nmodes - 1):
                    w[i] = f32(w[i] * alpha1)
            mu[mode] = list(x)
            var[mode] = varInit
            i = nmodes - 1
            while i > 0:
                if alphaT < w[i - 1]:
                    break
                w[i], w[i - 1] = w[i - 1], w[i]
                var[i], var[i - 1] = var[i - 1], var[i]
                mu[i], mu[i - 1] = mu[i - 1], mu[i]
                i -= 1
        # mask
        if background:
            mask = 0
        else:
            mask = 255
            tw = f32(0)
            for m in range(nmodes):
                num = f32(0); den = f32(0)
                for c in range(3):
                    num = f32(num + f32(x[c] * mu[m][c]))
                    den = f32(den + f32(mu[m][c] * mu[m][c]))
                if den == 0:
                    break
                if num <= den and num >= f32(tau * den):
                    a = f32(num / den)
                    d2 = f32(0)
                    for c in range(3):
                        dd = f32(f32(a * mu[m][c]) - x[c])
                        d2 = f32(d2 + f32(dd * dd))
                    if d2 < f32(f32(f32(Tb * var[m]) * a) * a):
                        mask = 127
                        break
                tw = f32(tw + w[m])
                if tw > TB:
                    break
        out.append(dict(mask=int(mask), nmodes=int(nmodes),
                        weight=[float(v) for v in w[:nmodes]],
                        variance=[float(v) for v in var[:nmodes]],
                        mean=[[float(c) for c in mu[m]] for m in range(nmodes)]))
    return out


def kalman_trace(samples, dt=0.02, timeout=0.0, sigma_accel=5.0, sigma_noise=0.0):
    """Independent float64 restatement of `posifilt kalman` (KalmanFilter2D.cpp:95-210 over
    cv::KalmanFilter(4,2,0,CV_64F)) with numpy matrix algebra and np.linalg.solve -- a different
    evaluation order from oracle/kalman.c on purpose: the two must agree to rounding, not bit for bit.
    samples: list of (valid, x, y).  Returns per sample [valid, x, y, vx, vy]."""
    A = np.eye(4); Q = np.eye(4); R = np.eye(2); H = np.zeros((2, 4))      # KalmanFilter::init
    x_pre = np.zeros(4); x_post = np.zeros(4); P_pre = np.zeros((4, 4)); P_post = np.zeros((4, 4))
    reported = np.full(4, 6.0)          # Mat_<double>{4, 1, CV_64F}: filled with 6.0
    meas = np.full(2, 6.0)
    aliased = False
    found, missing = False, 0
    threshold = int(timeout / dt)
    out = []
    for valid, mx, my in samples:
        if valid:
            meas = np.array([mx, my], float)
            missing = 0
            if not found:
                A = np.eye(4); A[0, 1] = dt; A[2, 3] = dt
                H = np.zeros((2, 4)); H[0, 0] = 1; H[1, 2] = 1
                q = np.array([[dt ** 4 / 4, dt ** 3 / 2], [dt ** 3 / 2, dt ** 2]]) * sigma_accel ** 2
                Q = np.zeros((4, 4)); Q[:2, :2] = q; Q[2:, 2:] = q
                R = np.eye(2) * sigma_noise ** 2
                P_pre = np.eye(4) * 1000.0
                x_pre = np.array([mx, 0.0, my, 0.0]); x_post = x_pre.copy()
            found = True
        else:
            missing += 1
        if missing >= threshold:
            found = False
        if found:
            x_pre = A @ x_post
            P_pre = A @ P_post @ A.T + Q
            x_post = x_pre.copy(); P_post = P_pre.copy()
            aliased = True
            S = H @ P_pre @ H.T + R
            K = np.linalg.solve(S, H @ P_pre).T
            x_post = x_pre + K @ (meas - H @ x_pre)
            P_post = P_pre - K @ (H @ P_pre)
        r = x_pre if aliased else reported
        out.append([bool(found), float(r[0]), float(r[2]), float(r[1]), float(r[3])])
    return out


def bgr2grey_restated(bgr):
    """COLOR_BGR2GRAY on 8U written out as the closed form of RGB2Gray<uchar>'s table sums
    (python ints, no table): (1868 B + 9617 G + 4899 R + 2^13) >> 14."""
    bgr = np.asarray(bgr, np.int64)
    return ((1868 * bgr[..., 0] + 9617 * bgr[..., 1] + 4899 * bgr[..., 2] + (1 << 13)) >> 14).astype(np.uint8)


def hsv2bgr_restated(hsv):
    """COLOR_HSV2BGR on 8U as OpenCV 3.1.0 does it (HSV2RGB_b over HSV2RGB_f, hrange 180), vectorised in
    numpy float32 -- every operation rounds to float32 on its own, np.rint = round half to even (cvRound)."""
    f = np.float32
    hsv = np.asarray(hsv, np.uint8)
    h = hsv[..., 0].astype(f)
    s = hsv[..., 1].astype(f) * f(f(1) / f(255))
    v = hsv[..., 2].astype(f) * f(f(1) / f(255))
    h = h * f(f(6) / f(180))
    h = np.where(h >= f(6), h - f(6), h).astype(f)
    sector = np.floor(h).astype(np.int64)
    h = (h - sector.astype(f)).astype(f)
    one = f(1)
    tab = np.stack([v, v * (one - s), v * (one - s * h), v * (one - s * (one - h))], -1).astype(f)
    sector_data = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])
    idx = sector_data[sector]                                   # (..., 3) -> which tab entry is b, g, r
    bgr = np.take_along_axis(tab, idx, -1)
    bgr = np.where((s == 0)[..., None], v[..., None], bgr).astype(f)
    return np.clip(np.rint(bgr * f(255)), 0, 255).astype(np.uint8)


def main():
    rng = np.random.default_rng(20260929)

    # ---- HSV known answers (the well-known OpenCV hues) ----
    hsv = dict(bgr=[[0, 0, 255], [0, 255, 0], [255, 0, 0], [0, 255, 255], [128, 128, 128],
                    [0, 0, 0], [10, 200, 100], [201, 200, 199], [255, 255, 255], [255, 0, 255],
                    [255, 255, 0]],
               hsv=[[0, 255, 255], [60, 255, 255], [120, 255, 255], [30, 255, 255], [0, 0, 128],
                    [0, 0, 0], [46, 242, 200], [105, 3, 201], [0, 0, 255], [150, 255, 255],
                    [90, 255, 255]])
    json.dump(hsv, open(os.path.join(HERE, "hsv_kat.json"), "w"), indent=1)

    # ---- the other conversions of `framefilt col` (Color.h:45-51): known answers from the restatements above ----
    # grey: the well-known luma of the primaries (blue 29, green 150, red 76), yellow 226, white 255
    # hsv -> bgr: the primaries/secondaries come back exactly; h = 180..255 wraps round (h * 6/180 >= 6)
    colours = [[255, 0, 0], [0, 255, 0], [0, 0, 255], [0, 255, 255], [255, 255, 255], [0, 0, 0], [128, 128, 128],
               [10, 200, 100], [201, 200, 199], [255, 64, 0], [1, 1, 1], [254, 255, 253]]
    rng_cvt = np.random.default_rng(20260930)      # its own generator: the vectors below keep theirs
    colours += rng_cvt.integers(0, 256, (52, 3)).tolist()
    hsvs = [[0, 255, 255], [60, 255, 255], [120, 255, 255], [30, 255, 255], [0, 0, 128], [0, 0, 0], [46, 242, 200],
            [105, 3, 201], [179, 255, 255], [180, 255, 255], [255, 255, 255], [90, 128, 77], [15, 1, 254]]
    hsvs += rng_cvt.integers(0, 256, (51, 3)).tolist()
    cvt = dict(bgr=colours, grey=bgr2grey_restated(colours).tolist(),
               hsv=hsvs, bgr_of_hsv=hsv2bgr_restated(hsvs).tolist())
    assert cvt["grey"][:5] == [29, 150, 76, 226, 255]
    assert cvt["bgr_of_hsv"][:4] == [[0, 0, 255], [0, 255, 0], [255, 0, 0], [0, 255, 255]]
    json.dump(cvt, open(os.path.join(HERE, "cvt_color_kat.json"), "w"))

    # ---- contour known answers (hand derivations in comments) ----
    def blank(h, w): return [[0] * w for _ in range(h)]
    cases = []
    # single pixel: polygon of one vertex, area 0 -> never selected
    g = blank(7, 7); g[3][3] = 1
    cases.append(dict(name="single_pixel", img=g, valid=False, area=0.0))
    # 1 x 5 line: degenerate polygon, area 0
    g = blank(7, 9)
    for x in range(2, 7): g[3][x] = 1
    cases.append(dict(name="line_1x5", img=g, valid=False, area=0.0))
    # 7x4 rect at (3,2): polygon through pixel centres (3,2)-(9,5): area 6*3=18, centroid (6,3.5)
    g = blank(12, 16)
    for y in range(2, 6):
        for x in range(3, 10): g[y][x] = 1
    cases.append(dict(name="rect_7x4", img=g, valid=True, area=18.0, x=6.0, y=3.5))
    # L-shape: pixels x in [2,6], y in [2,3] plus x in [2,3], y in [4,7].  8-connected border
    # following cuts the concave corner diagonally ((3,4)->(4,3)), so the polygon is
    # (2,2)->(2,7)->(3,7)->(3,4)->(4,3)->(6,3)->(6,2): rectangle A (2..6 x 2..3) = 4,
    # rectangle B (2..3 x 3..7) = 4, corner triangle (3,3),(3,4),(4,3) = 0.5 -> area 8.5.
    # First moments: A 4*(4,2.5) + B 4*(2.5,5) + T 0.5*(10/3,10/3) = (166/6, 190/6)
    # -> raw Green sums (|a00|,|a10|,|a01|) = (17,166,190); centroid by contourMoments' epilogue.
    g = blank(10, 9)
    for y in (2, 3):
        for x in range(2, 7): g[y][x] = 1
    for y in range(4, 8):
        for x in (2, 3): g[y][x] = 1
    cases.append(dict(name="L_shape", img=g, valid=True, area=8.5,
                      x=(166 * 0.16666666666666666) / 8.5, y=(190 * 0.16666666666666666) / 8.5))
    # blob touching the frame: 5x5 square in the top-left corner; OpenCV 3.1 zeroes the
    # outer frame first, so the surviving pixels are x,y in [1,4]: area 3*3=9, centroid (2.5,2.5)
    g = blank(9, 9)
    for y in range(0, 5):
        for x in range(0, 5): g[y][x] = 1
    cases.append(dict(name="touch_edge", img=g, valid=True, area=9.0, x=2.5, y=2.5))
    # ring (outer 1..11, wall 2 thick) with a 3x3 blob nested in its hole: RETR_EXTERNAL
    # reports only the ring's OUTER border: area 10*10=100, centroid (6,6).
    g = blank(13, 13)
    for y in range(1, 12):
        for x in range(1, 12): g[y][x] = 1
    for y in range(3, 10):
        for x in range(3, 10): g[y][x] = 0
    for y in range(5, 8):
        for x in range(5, 8): g[y][x] = 1
    cases.append(dict(name="ring_nested", img=g, valid=True, area=100.0, x=6.0, y=6.0))
    # same image, area window [1, 50): the nested 3x3 blob (area 4) must NOT be found
    cases.append(dict(name="ring_nested_window", img=g, valid=False, area=0.0, min_area=1.0, max_area=50.0))
    # two equal 4x4 squares (area 9 each): strict '>' over the reversed list keeps the one
    # discovered LAST in raster order -> the lower one at (2..5, 8..11): centroid (3.5, 9.5)
    g = blank(14, 14)
    for y in range(2, 6):
        for x in range(7, 11): g[y][x] = 1
    for y in range(8, 12):
        for x in range(2, 6): g[y][x] = 1
    cases.append(dict(name="tie_break", img=g, valid=True, area=9.0, x=3.5, y=9.5))
    # same row tie: left (2..5) and right (8..11) squares on rows 2..5: right one is later in raster
    g = blank(9, 14)
    for y in range(2, 6):
        for x in range(2, 6): g[y][x] = 1
        for x in range(8, 12): g[y][x] = 1
    cases.append(dict(name="tie_break_row", img=g, valid=True, area=9.0, x=9.5, y=3.5))
    # diagonal 2-pixel chain: polygon (2,2)->(3,3)->back: area 0
    g = blank(7, 7); g[2][2] = 1; g[3][3] = 1
    cases.append(dict(name="diag2", img=g, valid=False, area=0.0))
    # plus sign, arms length 1 around (4,4): polygon is the diamond (4,3),(3,4),(4,5),(5,4): area 2, centroid (4,4)
    g = blank(9, 9)
    for (x, y) in ((4, 3), (3, 4), (4, 4), (5, 4), (4, 5)): g[y][x] = 1
    cases.append(dict(name="plus", img=g, valid=True, area=2.0, x=4.0, y=4.0))
    # area window: 7x4 rect (18) and 3x3 square (4); window [1,10) must pick the square at (12..14, 8..10)
    g = blank(14, 18)
    for y in range(2, 6):
        for x in range(3, 10): g[y][x] = 1
    for y in range(8, 11):
        for x in range(12, 15): g[y][x] = 1
    cases.append(dict(name="area_window", img=g, valid=True, area=4.0, x=13.0, y=9.0, min_area=1.0, max_area=10.0))
    json.dump(cases, open(os.path.join(HERE, "contours.json"), "w"))

    # ---- morphology impulse answers: computed from the window definition ----
    # dilate(k) of an impulse at (x0,y0): out==255 for x in [x0-(k-1)+a, x0+a], a=k//2 (clipped)
    morph = []
    for k in (1, 2, 3, 7, 10, 13):
        a = k // 2
        for (x0, y0) in ((0, 0), (8, 6), (19, 15), (0, 15), (10, 0)):
            morph.append(dict(k=k, rows=16, cols=20, x0=x0, y0=y0,
                              x_lo=max(0, x0 - (k - 1) + a), x_hi=min(19, x0 + a),
                              y_lo=max(0, y0 - (k - 1) + a), y_hi=min(15, y0 + a)))
    json.dump(morph, open(os.path.join(HERE, "morph_impulse.json"), "w"))

    # ---- MOG2 single-pixel traces ----
    traces = []
    base = (120, 130, 140)
    for name, rate in (("alpha0", 0.0), ("alpha001", 0.01), ("alpha05", 0.5), ("alpha1", 1.0), ("alpha_neg", -1.0)):
        pix = []
        for t in range(24):
            if t % 7 == 6:
                pix.append((30, 200, 90))
            elif t % 5 == 4:
                pix.append((0, 0, 0))
            else:
                pix.append(tuple(int(b + rng.integers(-9, 10)) for b in base))
        for restore in (1, 0):
            traces.append(dict(name=name + ("" if restore else "_shrink"), rate=rate, restore=restore, pixels=pix,
                               frames=mog2_pixel_trace(pix, [rate] * len(pix), restore=bool(restore))))
    # five live modes: cycle through 6 well separated colours with a quick learner
    cols = [(10, 10, 10), (60, 200, 30), (200, 40, 90), (250, 250, 250), (20, 120, 240), (128, 0, 128)]
    pix = [cols[(t * 5 + t // 3) % 6] for t in range(40)]
    for restore in (1, 0):
        traces.append(dict(name="five_modes" + ("" if restore else "_shrink"), rate=0.05, restore=restore, pixels=pix,
                           frames=mog2_pixel_trace(pix, [0.05] * len(pix), restore=bool(restore))))
    # pruning: colours seen once decay below the prune threshold within a few frames at rate 0.3; coming
    # back to them later separates the two readings of the mode count (a zero-weight slot is revived /
    # a fresh mode with varInit is made), as does filling all five slots and replacing the last one
    pix = [cols[0]] * 3 + [cols[1]] + [cols[0]] * 9 + [cols[1]] * 2 + [cols[2], cols[3], cols[4], cols[5]] + \
          [cols[0]] * 8 + [cols[3], cols[1], cols[5], cols[0], cols[2]] + [cols[0]] * 6 + [cols[4]] * 3
    for restore in (1, 0):
        traces.append(dict(name="prune_revive" + ("" if restore else "_shrink"), rate=0.3, restore=restore, pixels=pix,
                           frames=mog2_pixel_trace(pix, [0.3] * len(pix), restore=bool(restore))))
    json.dump(traces, open(os.path.join(HERE, "mog2_trace.json"), "w"))
    # ---- posifilt kalman traces ----
    krng = np.random.default_rng(4242)
    def walk(n, gaps=()):
        xs = []
        for t in range(n):
            valid = not any(a <= t < b for a, b in gaps)
            xs.append((valid, 300.0 + 4.0 * t + float(krng.normal(0, 1.5)), 200.0 - 2.5 * t + float(krng.normal(0, 1.5))))
        return xs
    ktr = []
    for name, kw, samples in (
            ("default_timeout0_never_tracks", {}, walk(12)),
            ("timeout_100ms", dict(timeout=0.1), walk(40)),
            ("short_gap_uses_stale_measurement", dict(timeout=0.1), walk(40, gaps=((10, 13),))),
            ("long_gap_reinitialises", dict(timeout=0.1, sigma_noise=2.0), walk(60, gaps=((15, 30), (45, 47)))),
            ("starts_invalid", dict(timeout=0.2, sigma_accel=50.0, sigma_noise=1.0, dt=0.01), walk(50, gaps=((0, 6),))),
    ):
        ktr.append(dict(name=name, params=kw, samples=[[bool(v), x, y] for v, x, y in samples],
                        out=kalman_trace(samples, **kw)))
    json.dump(ktr, open(os.path.join(HERE, "kalman_trace.json"), "w"))
    print("wrote golden vectors to", HERE)


if __name__ == "__main__":
    main()
