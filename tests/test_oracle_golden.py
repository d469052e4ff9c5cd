"""CPU tests: the C oracle against the hand-derived golden vectors (tests/golden).

The reference has no tests for this path: these known answers
are this repo's own pins -- PARITY UNPINNED by the reference.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O


def _load(golden_dir, name):
    with open(os.path.join(golden_dir, name)) as f:
        return json.load(f)


def test_hsv_known_answers(golden_dir):
    g = _load(golden_dir, "hsv_kat.json")
    got = O.bgr2hsv(np.array(g["bgr"], np.uint8))
    assert got.tolist() == g["hsv"]


def test_hsv_range_and_exhaustive_selfcheck():
    # every (b,g,r): H < 180, V == max, S == 0 iff grey
    b, g, r = np.meshgrid(np.arange(0, 256, 3), np.arange(256), np.arange(0, 256, 5), indexing="ij")
    bgr = np.stack([b, g, r], -1).reshape(-1, 3).astype(np.uint8)
    hsv = O.bgr2hsv(bgr)
    assert hsv[:, 0].max() < 180
    assert (hsv[:, 2] == bgr.max(1)).all()
    grey = (bgr.max(1) == bgr.min(1))
    assert ((hsv[:, 1] == 0) == grey).all()


@pytest.mark.parametrize("lo,hi,expect", [
    ((0, 0, 0), (256, 256, 256), "all"),       # Oat defaults: all-pass, 256 saturates to 255
    ((10, 0, 0), (5, 256, 256), "none"),       # lo > hi -> empty
    ((256, 0, 0), (256, 256, 256), "none"),    # lo > 255 -> empty
    ((100, 100, 100), (100, 100, 100), "eq"),  # both bounds inclusive
])
def test_inrange_bounds(lo, hi, expect):
    src = np.array([[[100, 100, 100], [99, 100, 100], [255, 255, 255], [0, 0, 0]]], np.uint8)
    got = O.inrange3(src, lo, hi).ravel().tolist()
    want = {"all": [255] * 4, "none": [0] * 4, "eq": [255, 0, 0, 0]}[expect]
    assert got == want
    g1 = O.inrange1(np.array([[0, 99, 100, 255]], np.uint8), 100, 256).ravel().tolist()
    assert g1 == [0, 0, 255, 255]


def test_morphology_impulses(golden_dir):
    for c in _load(golden_dir, "morph_impulse.json"):
        img = np.zeros((c["rows"], c["cols"]), np.uint8)
        img[c["y0"], c["x0"]] = 255
        want = np.zeros_like(img)
        want[c["y_lo"]:c["y_hi"] + 1, c["x_lo"]:c["x_hi"] + 1] = 255
        assert (O.dilate(img, c["k"]) == want).all(), c
        # erosion is the dual on the complement (border 255 <-> border 0)
        assert (O.erode(255 - img, c["k"]) == 255 - want).all(), c


def test_morphology_borders_and_even_kernel():
    full = np.full((9, 11), 255, np.uint8)
    assert (O.erode(full, 7) == 255).all()          # erode pads with 255
    assert (O.dilate(np.zeros_like(full), 7) == 0).all()   # dilate pads with 0
    # even k=10: anchor 5 -> window [x-5, x+4] (asymmetric)
    row = np.zeros((1, 30), np.uint8); row[0, 15] = 255
    d = O.dilate(row, 10)
    assert np.nonzero(d[0])[0].tolist() == list(range(11, 21))


def test_contour_known_answers(golden_dir):
    for c in _load(golden_dir, "contours.json"):
        img = (np.array(c["img"], np.uint8) * 255)
        kw = dict(min_area=c.get("min_area", 0.0), max_area=c.get("max_area", float(np.finfo(np.float64).max)))
        for fn in (O.sift_contours, O.sift_cracks):
            d = fn(img, **kw)
            assert d["valid"] == c["valid"], (c["name"], fn.__name__, d)
            assert d["area"] == c["area"], (c["name"], fn.__name__, d)
            if c["valid"]:
                assert d["x"] == c["x"] and d["y"] == c["y"], (c["name"], fn.__name__, d)


def test_contour_list_order_is_reverse_discovery():
    img = np.zeros((12, 12), np.uint8)
    img[2:5, 2:5] = 255
    img[7:10, 6:9] = 255
    cs = O.find_contours(img)
    assert [c["start"] for c in cs] == [(6, 7), (2, 2)]
    # CHAIN_APPROX_SIMPLE: a rectangle has exactly 4 vertices
    assert all(len(c["points"]) == 4 for c in cs)


def test_mog2_single_pixel_traces(golden_dir):
    names = set()
    for tr in _load(golden_dir, "mog2_trace.json"):
        names.add(tr["name"])
        m = O.Mog2(1, 1, 3, params=dict(restore_nmodes=tr.get("restore", 1)))
        for t, (px, want) in enumerate(zip(tr["pixels"], tr["frames"])):
            mask = m.apply(np.array(px, np.uint8).reshape(1, 1, 3), tr["rate"])
            nm, w, v, mu = m.state()
            k = want["nmodes"]
            assert int(mask[0, 0]) == want["mask"], (tr["name"], t)
            assert int(nm[0]) == k, (tr["name"], t)
            assert w[0, :k].tolist() == [np.float32(x) for x in want["weight"]], (tr["name"], t)
            assert v[0, :k].tolist() == [np.float32(x) for x in want["variance"]], (tr["name"], t)
            assert mu[0, :k].tolist() == [[np.float32(c) for c in r] for r in want["mean"]], (tr["name"], t)
    # both readings of the mode count (oracle/mog2.c "Mode count") are pinned by their own traces
    assert {"prune_revive", "prune_revive_shrink", "alpha05", "alpha05_shrink"} <= names


@pytest.mark.parametrize("restore", [1, 0])
@pytest.mark.parametrize("rate", [0.3, 0.02, -1.0])
def test_mog2_random_pixels_against_the_python_restatement(restore, rate):
    """The per-pixel Python restatement that made the golden traces (tests/golden/make_golden.py, written from the
    description of MOG2Invoker, float32 step by step) on RANDOM pixel histories -- two-level flicker, slow drift, rare
    outliers, enough to prune, revive and replace modes -- against the C oracle run on all of them as one image:
    mask, mode count, weights, variances, means of every pixel and frame, bit for bit, for both readings of the count."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    G = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(G)
    rng = np.random.default_rng(41 + restore)
    npx, nfr = 24, 45
    base = rng.integers(30, 220, (npx, 3))
    alt = rng.integers(0, 256, (npx, 3))
    frames = np.empty((nfr, npx, 3), np.uint8)
    for t in range(nfr):
        f = base + rng.integers(-6, 7, (npx, 3)) + (t // 9)
        flip = rng.random(npx) < 0.25
        f[flip] = alt[flip] + rng.integers(-3, 4, (int(flip.sum()), 3))
        wild = rng.random(npx) < 0.05
        f[wild] = rng.integers(0, 256, (int(wild.sum()), 3))
        frames[t] = np.clip(f, 0, 255)
    m = O.Mog2(1, npx, 3, params=dict(restore_nmodes=restore))
    got = []
    for t in range(nfr):
        mask = m.apply(frames[t].reshape(1, npx, 3), rate)
        nm, w, v, mu = m.state()
        got.append((mask[0].copy(), nm.copy(), w.copy(), v.copy(), mu.copy()))
    for p in range(npx):
        tr = G.mog2_pixel_trace([tuple(int(c) for c in frames[t, p]) for t in range(nfr)], [rate] * nfr, restore=bool(restore))
        for t, want in enumerate(tr):
            mask, nm, w, v, mu = got[t]
            k = want["nmodes"]
            assert int(mask[p]) == want["mask"] and int(nm[p]) == k, (p, t)
            assert w[p, :k].tolist() == [np.float32(x) for x in want["weight"]], (p, t)
            assert v[p, :k].tolist() == [np.float32(x) for x in want["variance"]], (p, t)
            assert mu[p, :k].tolist() == [[np.float32(c) for c in r] for r in want["mean"]], (p, t)
    counts = np.stack([g[1] for g in got])
    assert counts.max() == 5 and (counts[-1] >= 2).sum() >= npx // 2          # the histories really exercised the mixture


def test_mog2_mode_count_readings_differ_only_after_a_prune():
    """restore_nmodes = 1 (MOG2Invoker's `nmodes = nNewModes;`): modesUsed never decreases and a pruned
    slot keeps weight 0; restore_nmodes = 0: the count shrinks.  Before the first prune the two readings
    are the same computation."""
    rng = np.random.default_rng(11)
    a, b = O.Mog2(6, 7, 3), O.Mog2(6, 7, 3, params=dict(restore_nmodes=0))
    prev = np.zeros(42, np.uint8)
    differed = False
    for t in range(40):
        f = rng.integers(0, 256, (6, 7, 3), dtype=np.uint8) if t % 3 == 0 else np.full((6, 7, 3), 100, np.uint8)
        ma, mb = a.apply(f, 0.3), b.apply(f, 0.3)
        nma, wa, _, _ = a.state()
        nmb, wb, _, _ = b.state()
        assert (nma >= prev).all()                       # never shrinks
        assert (nma >= nmb).all()
        prev = nma
        if not differed and (nma != nmb).any():
            differed = True
            # the pruned mode is still counted, with weight exactly 0 (in ITS slot: the loop ends at the
            # prune, so slots behind it keep their stale weights and the zero need not be the last one)
            p = int(np.flatnonzero(nma != nmb)[0])
            assert wa[p, :nma[p]].min() == 0.0 and wb[p, :nmb[p]].min() >= 0.0
        if not differed:
            assert (ma == mb).all() and (wa == wb).all()
    assert differed


def test_mog2_frame1_and_frozen_model():
    rng = np.random.default_rng(3)
    f1 = rng.integers(1, 256, (8, 9, 3), dtype=np.uint8)
    f1[0, 0] = 0                                  # pure black pixel
    m = O.Mog2(8, 9, 3)
    out, mask = m.filter(f1, 0.0)
    # frame 1: alpha = 0.5, every pixel creates mode 0; shadow test sees a = 1 -> 127, black -> 255
    assert mask[0, 0] == 255 and (np.delete(mask.ravel(), 0) == 127).all()
    assert (out == f1).all()                      # Oat keeps mask != 0 -> first output is the input
    # frames >= 2 with Oat's default -a 0: pixel zeroed iff ||x - x1||^2 < 16*15 (float compare)
    f2 = np.clip(f1.astype(int) + rng.integers(-14, 15, f1.shape), 0, 255).astype(np.uint8)
    out2, mask2 = m.filter(f2, 0.0)
    d2 = ((f2.astype(np.float32) - f1.astype(np.float32)) ** 2).sum(-1)
    assert ((mask2 == 0) == (d2 < 240)).all()
    assert (out2[mask2 == 0] == 0).all() and (out2[mask2 != 0] == f2[mask2 != 0]).all()
    nm, w, v, mu = m.state()
    assert (nm == 1).all() and (w[:, 0] == 1).all() and (v[:, 0] == 15).all()
    assert (mu[:, 0].reshape(8, 9, 3) == f1).all()   # frozen: state still frame 1


def test_mog2_row_parallel_equals_serial():
    rng = np.random.default_rng(4)
    a, b = O.Mog2(17, 23, 3), O.Mog2(17, 23, 3)
    for t in range(12):
        f = rng.integers(0, 256, (17, 23, 3), dtype=np.uint8)
        o1, m1 = a.filter(f, 0.05, nthreads=1)
        o2, m2 = b.filter(f, 0.05, nthreads=4)
        assert (o1 == o2).all() and (m1 == m2).all()
    for x, y in zip(a.state(), b.state()):
        assert (x == y).all()


def test_chain_on_the_worker_pool_is_independent_of_the_thread_count_and_leaves_its_input_alone():
    """oracle/pool.c: persistent row workers, woken per stage.  The whole chain (MOG2 + setTo + HSV + inRange + erode +
    dilate + contours) must give the same detection, threshold image and model for 1, 3, 8 and 40 workers -- growing and
    shrinking the job count between calls -- and must not touch the caller's frame (the workers copy their rows)."""
    from oat_amd.synth import SyntheticStream
    rows, cols = 90, 140
    st = SyntheticStream(rows, cols, 3, n_discs=1, radius=9)
    frames = [st.frame(t, with_discs=t > 0) for t in range(10)]
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=5, min_area=10.0, max_area=1e5)
    runs = {}
    for order, nts in enumerate(((1,) * 10, (3, 40, 8, 1, 40, 3, 8, 8, 40, 2))):
        m = O.Mog2(rows, cols, 3)
        out = []
        for f, nt in zip(frames, nts):
            keep = f.copy()
            d, thr = O.chain_step(m, f, 0.02, p, nthreads=nt)
            assert (f == keep).all()
            out.append((d, thr.copy()))
        runs[order] = (out, m.state())
    for (d1, t1), (d2, t2) in zip(runs[0][0], runs[1][0]):
        assert d1 == d2 and (t1 == t2).all()
    for x, y in zip(runs[0][1], runs[1][1]):
        assert (x == y).all()
    assert sum(d["valid"] for d, _ in runs[0][0]) >= 4


def test_blur_known_answers_and_dilation_equivalence():
    """cv::blur semantics (DifferenceDetector.cpp:160-161) and the property the GPU path relies on:
    for k <= 22 the box blur of a {0,255} image is non-zero exactly where the k x k dilation is,
    except on the outermost ring (BORDER_REFLECT_101), which findContours zeroes anyway."""
    img = np.zeros((7, 9), np.uint8)
    img[3, 4] = 255
    b3 = O.blur(img, 3)
    assert b3[2:5, 3:6].tolist() == [[28] * 3] * 3 and b3.sum() == 28 * 9          # 255/9 = 28.33 -> 28
    b2 = O.blur(img, 2)
    assert b2[3:5, 4:6].tolist() == [[64, 64], [64, 64]]                            # 63.75 -> 64; window [x-1, x]
    edge = np.zeros((5, 6), np.uint8)
    edge[2, 1] = 255
    assert O.blur(edge, 2)[2, 0] == 64          # reflect-101: pixel x=1 also feeds output x=0 (window [-1,0] -> {1,0})
    rng = np.random.default_rng(5)
    for k in (2, 3, 4, 7, 10, 22):
        for _ in range(10):
            h, w = int(rng.integers(k + 2, 50)), int(rng.integers(k + 2, 60))
            im = (rng.random((h, w)) < 0.1).astype(np.uint8) * 255
            b, d = O.blur(im, k) > 0, O.dilate(im, k) > 0
            assert (b[1:-1, 1:-1] == d[1:-1, 1:-1]).all()
            assert O.sift_contours(b.astype(np.uint8) * 255) == O.sift_contours(d.astype(np.uint8) * 255)
    one = np.zeros((60, 60), np.uint8)
    one[30, 30] = 255
    assert O.blur(one, 22).max() == 1 and O.blur(one, 23).max() == 0            # why the GPU path stops at 22


def test_diff_detector_first_frame_and_motion():
    d = O.Diff(40, 50, diff_threshold=10, blur=2)
    f0 = np.full((40, 50), 30, np.uint8)
    r, thr = d.detect(f0)
    assert (thr == f0).all() and r["valid"] and r["area"] == (50 - 3) * (40 - 3)   # first frame analysed as is
    f1 = f0.copy()
    f1[10:20, 15:30] = 200
    r, thr = d.detect(f1)
    assert r["valid"] and r["area"] == 15 * 10      # 15x10 box grown by the 2x2 blur -> 16x11 pixels -> (16-1)*(11-1)
    r, thr = d.detect(f1)
    assert not r["valid"] and thr.max() == 0        # nothing moved


def test_grey_conversion_and_sibling_filters_known_answers():
    """cv::cvtColor(BGR2GRAY) fixed point (R2Y 4899, G2Y 9617, B2Y 1868, shift 14), framefilt thresh and bsub."""
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 100]]], np.uint8)
    # 0.114 B + 0.587 G + 0.299 R: blue 29, green 150, red 76; (1868*10 + 9617*200 + 4899*100 + 8192) >> 14 = 148
    assert O.bgr2grey(px).tolist() == [[255, 0, 29, 150, 76, 148]]
    out = O.thresh_filter(px, 30, 150)              # keeps grey in [30,150]: green(150), red(76), the last (148)
    assert out[0].tolist() == [[0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 100]]
    # bsub, alpha 0: first frame is the background -> zeros; later frames subtract it with saturation
    b = O.Bsub(1, 3, 1, 0.0)
    assert b.filter(np.array([[10, 100, 250]], np.uint8)).tolist() == [[0, 0, 0]]
    assert b.filter(np.array([[5, 130, 255]], np.uint8)).tolist() == [[0, 30, 5]]
    # alpha 0.5: bg_f = 0.5*frame + 0.5*bg_f; bg = cvRound (half to even): 10 -> (10+21)/2 = 15.5 -> 16
    b = O.Bsub(1, 1, 1, 0.5)
    assert b.filter(np.array([[10]], np.uint8)).tolist() == [[0]]
    assert b.filter(np.array([[21]], np.uint8)).tolist() == [[21 - 16]]
    assert b.filter(np.array([[21]], np.uint8)).tolist() == [[21 - 18]]      # 0.5*21 + 0.5*15.5 = 18.25 -> 18


def test_kalman_traces_against_independent_numpy_restatement(golden_dir):
    """oracle/kalman.c vs tests/golden/kalman_trace.json (numpy matrix algebra + linalg.solve):
    flags identical, values equal to rounding (the two use different evaluation orders)."""
    for tr in json.load(open(os.path.join(golden_dir, "kalman_trace.json"))):
        k = O.Kalman(**tr["params"])
        for t, ((v, x, y), w) in enumerate(zip(tr["samples"], tr["out"])):
            o = k.filter(v, x, y)
            assert o["position_valid"] == o["velocity_valid"] == w[0], (tr["name"], t)
            for got, want in zip((o["x"], o["y"], o["vx"], o["vy"]), w[1:]):
                assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (tr["name"], t, got, want)


def test_kalman_reference_quirks():
    """The observable oddities of KalmanFilter2D.cpp a drop-in has to keep."""
    k = O.Kalman()                                   # default --timeout 0: threshold 0, never tracks
    for t in range(5):
        o = k.filter(True, 10.0 + t, 20.0)
        assert not o["position_valid"] and (o["x"], o["y"], o["vx"], o["vy"]) == (6.0, 6.0, 6.0, 6.0)
    k = O.Kalman(timeout=0.1)                        # threshold 5
    o = k.filter(False, 0, 0)                        # nothing seen yet
    assert not o["position_valid"] and o["x"] == 6.0
    o = k.filter(True, 100.0, 50.0)                  # first detection: predicted state = the measurement
    assert o["position_valid"] and (o["x"], o["y"], o["vx"], o["vy"]) == (100.0, 50.0, 0.0, 0.0)
    o = k.filter(True, 102.0, 50.0)                  # sigma_noise 0: the correction trusted the measurement fully
    assert o["x"] == 100.0 and o["vx"] == 0.0        # ... but the report is the PREDICTION made before it
    o = k.filter(True, 104.0, 50.0)                  # P' = Q on the first step => velocity gain 2/dt: v = 2 px * 100 /s
    assert abs(o["vx"] - 200.0) < 1e-9 and abs(o["x"] - 106.0) < 1e-9 and o["vy"] == 0.0
    seen = [k.filter(False, 0, 0)["position_valid"] for _ in range(6)]
    assert seen == [True, True, True, True, False, False]       # 5th miss reaches the threshold
    o = k.filter(True, 300.0, 10.0)                  # re-initialised at the new measurement
    assert o["position_valid"] and (o["x"], o["y"], o["vx"], o["vy"]) == (300.0, 10.0, 0.0, 0.0)


# ---- the other conversions of `framefilt col` (oat::color_conv_table, Color.h:45-51) ----

def test_cvt_color_known_answers(golden_dir):
    g = _load(golden_dir, "cvt_color_kat.json")
    bgr = np.array(g["bgr"], np.uint8)[None]
    assert O.bgr2grey(bgr)[0].tolist() == g["grey"]
    hsv = np.array(g["hsv"], np.uint8)[None]
    assert O.hsv2bgr(hsv)[0].tolist() == g["bgr_of_hsv"]
    grey = np.arange(256, dtype=np.uint8)[None]
    assert (O.grey2bgr(grey) == grey[..., None]).all()


def test_cvt_color_exhaustive_against_the_numpy_restatement():
    """All 2^24 inputs of BGR->GREY and HSV->BGR (h beyond 179 included): the C oracle against the vectorised
    float32 / integer restatement in tests/golden/make_golden.py, written separately from it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_golden", os.path.join(os.path.dirname(__file__), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    c = np.arange(256, dtype=np.uint8)
    for a0 in range(0, 256, 32):                     # 8 slabs of 2^21 colours keep the float32 temporaries small
        a, b, d = np.meshgrid(c[a0:a0 + 32], c, c, indexing="ij")
        px = np.stack([a, b, d], -1).reshape(1, -1, 3)
        assert (O.bgr2grey(px) == mg.bgr2grey_restated(px)).all()
        assert (O.hsv2bgr(px) == mg.hsv2bgr_restated(px)).all()


def test_cvt_color_properties():
    c = np.arange(256, dtype=np.uint8)
    b, g, r = np.meshgrid(c[::5], c[::3], c[::7], indexing="ij")
    bgr = np.stack([b, g, r], -1).reshape(1, -1, 3)
    grey = O.bgr2grey(bgr).astype(int)
    assert (grey >= bgr.min(-1)).all() and (grey <= bgr.max(-1)).all()          # a convex combination
    eq = np.stack([c, c, c], -1)[None]
    assert (O.bgr2grey(eq)[0] == c).all()                                        # 1868 + 9617 + 4899 = 2^14
    # BGR -> HSV -> BGR comes back within the quantisation of the 8-bit HSV grid (a hue step is 2 degrees)
    back = O.hsv2bgr(O.bgr2hsv(bgr)).astype(int)
    assert np.abs(back - bgr).max() <= 6
    # s == 0 -> grey of value v whatever the hue; v == 0 -> black
    hsv = np.stack([c, np.zeros_like(c), c[::-1]], -1)[None]
    assert (O.hsv2bgr(hsv) == c[::-1][None, :, None]).all()
    assert O.hsv2bgr(np.stack([c, c, np.zeros_like(c)], -1)[None]).max() == 0


def test_cvt_color_table_is_the_references():
    """oat::color_conv_table (Color.h:45-51): -1 nothing to be done, -2 not possible, else bytes per pixel."""
    B, G, C3, H = O.BINARY, O.GREY, O.BGR, O.HSV
    want = {(B, B): -1, (B, G): -1, (B, C3): 3, (B, H): -2,
            (G, B): -1, (G, G): -1, (G, C3): 3, (G, H): -2,
            (C3, B): 1, (C3, G): 1, (C3, C3): -1, (C3, H): 3,
            (H, B): -2, (H, G): -2, (H, C3): 3, (H, H): -1}
    rng = np.random.default_rng(5)
    for (src, dst), code in want.items():
        f = rng.integers(0, 256, (3, 5, 3) if src >= 2 else (3, 5)).astype(np.uint8)
        rc, out = O.cvt_color(f, src, dst)
        assert rc == code, (src, dst)
        if code > 0:
            ref = {(C3, H): O.bgr2hsv, (H, C3): O.hsv2bgr}.get((src, dst), O.bgr2grey if src == C3 else O.grey2bgr)(f)
            assert out.shape == ref.shape and (out == ref).all()


def test_homography_filter_known_answers():
    """posifilt homography (HomographyTransform2D.cpp:62-107) = cv::perspectiveTransform on one point; hand-computed:
    identity; scale + offset (the velocity loses the offset, :79-89); a projective row; |w| <= FLT_EPSILON -> (0, 0);
    invalid parts are left alone."""
    assert O.homography(np.eye(3), True, 12.5, -3.0, True, 1.0, 2.0) == (12.5, -3.0, 1.0, 2.0)
    h = [2.0, 0.0, 10.0, 0.0, 0.5, -4.0, 0.0, 0.0, 1.0]
    assert O.homography(h, True, 3.0, 8.0, True, 1.0, 2.0) == (16.0, 0.0, 2.0, 1.0)
    # w = 0.01 x + 1: x = 100 -> w = 2; (x', y') = ((x + 2y) / 2, y / 2)
    h = [1.0, 2.0, 0.0, 0.0, 1.0, 0.0, 0.01, 0.0, 1.0]
    x, y, vx, vy = O.homography(h, True, 100.0, 10.0, False, 7.0, 7.0)
    assert (x, y, vx, vy) == ((100.0 + 20.0) * (1.0 / 2.0), 10.0 * (1.0 / 2.0), 7.0, 7.0)
    # the reference multiplies by the reciprocal of w: 1/3 rounds, so x * (1/3) is not x / 3 for every x
    h = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 3.0]
    assert O.homography(h, True, 5.0, 7.0)[:2] == (5.0 * (1.0 / 3.0), 7.0 * (1.0 / 3.0))
    h = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1e-8]          # |w| <= FLT_EPSILON
    assert O.homography(h, True, 5.0, 7.0)[:2] == (0.0, 0.0)
    assert O.homography(h, False, 5.0, 7.0)[:2] == (5.0, 7.0)


def test_pipelined_chain_equals_the_sequential_chain():
    """oracle/pipeline.c (bench.py's cpu_baseline with the reference's stage pipelining: framefilt mog | framefilt col +
    posidet front | findContours as three concurrent stages, FrameFilter.cpp:59-98 / PositionDetector.cpp:58-99): the
    detections of a frame sequence are those of oat_chain_step frame by frame, pipelined or not, whatever the number of
    row workers per stage, BGR and GREY; the model ends bit-identical."""
    from oat_amd.synth import SyntheticStream
    rows, cols, n = 96, 160, 14
    st = SyntheticStream(rows, cols, 5, n_discs=2, radius=7)
    frames = [st.frame(t, with_discs=t > 0) for t in range(n)]
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=3, dilate=5, min_area=5.0, max_area=1e5)
    ref = O.Mog2(rows, cols, 3)
    want = [O.chain_step(ref, f, 0.01, p)[0] for f in frames]
    assert sum(w["valid"] for w in want) >= n // 2
    for pipelined, tf, tm in ((False, 1, 1), (False, 3, 3), (True, 1, 1), (True, 4, 2), (True, 2, 5)):
        m = O.Mog2(rows, cols, 3)
        el, busy, got = O.pipeline_run(m, frames, 0, n, 0.01, p, t_front=tf, t_mid=tm, pipelined=pipelined)
        assert got == want, (pipelined, tf, tm)
        assert el > 0 and all(b > 0 for b in busy)
        a, b = m.state(), ref.state()
        assert all(np.array_equal(x, y, equal_nan=x.dtype.kind == "f") for x, y in zip(a, b))
    # the pool wraps: frame i = frames[(first + i) % len]
    m = O.Mog2(rows, cols, 3)
    _, _, got = O.pipeline_run(m, frames[:5], 0, 5, 0.01, p, 2, 2, True)
    _, _, got2 = O.pipeline_run(m, frames[:5], 2, 4, 0.01, p, 2, 2, True)
    ref2 = O.Mog2(rows, cols, 3)
    want2 = [O.chain_step(ref2, frames[i], 0.01, p)[0] for i in (0, 1, 2, 3, 4, 2, 3, 4, 0)]
    assert got + got2 == want2
    # GREY chain
    grey = [f[..., 1].copy() for f in frames]
    pg = O.hsv_params(h_lo=120, h_hi=256, erode=0, dilate=3, min_area=2.0, max_area=1e5)
    refg = O.Mog2(rows, cols, 1)
    wantg = [O.chain_step(refg, f, 0.01, pg)[0] for f in grey]
    mg = O.Mog2(rows, cols, 1)
    assert O.pipeline_run(mg, grey, 0, n, 0.01, pg, 3, 2, True)[2] == wantg


@pytest.mark.parametrize("e,d", [(0, 10), (3, 7), (7, 7), (2, 4), (13, 1), (1, 13)])
def test_chain_morphology_equals_the_single_stage_morphology(e, d):
    """The chain's row-parallel, vectorised erode / dilate (oracle/contours.c chain_worker, r04) against the per-pixel
    window form of oracle/pixels.c (oat_erode_rect / oat_dilate_rect, the one the known answers and the scipy statement
    pin): same threshold image after morphology for odd and EVEN sizes (anchor k / 2, not reflected), blobs touching every
    border, any number of row workers."""
    rng = np.random.default_rng(100 * e + d)
    rows, cols = 61, 83
    bgr = np.zeros((rows, cols, 3), np.uint8)
    bgr[rng.random((rows, cols)) < 0.35] = (255, 64, 0)                       # passes the window below
    bgr[0:3, :] = (255, 64, 0); bgr[:, -2:] = (255, 64, 0); bgr[-1, 5:9] = (255, 64, 0); bgr[20:44, 0] = (255, 64, 0)
    p = O.hsv_params(h_lo=100, h_hi=125, s_lo=150, s_hi=256, v_lo=100, v_hi=256, erode=e, dilate=d, min_area=1.0, max_area=1e9)
    hsv = O.bgr2hsv(bgr)
    thr = O.inrange3(hsv, (100, 150, 100), (125, 256, 256))
    if e > 0:
        thr = O.erode(thr, e)
    if d > 0:
        thr = O.dilate(thr, d)
    want_det, want_thr = O.detect_hsv(hsv, p)
    assert (want_thr == thr).all()
    for nt in (1, 3, 8):
        m = O.Mog2(rows, cols, 3)
        got_det, got_thr = O.chain_step(m, bgr, 0.0, p, nthreads=nt)        # frame 1: MOG2 keeps every non-black pixel
        assert (got_thr == thr).all(), (e, d, nt)
        assert got_det == want_det


def test_mog2_clamp_keeps_a_nan_variance_like_opencvs_macros():
    """oracle/mog2.c follows OpenCV's MAX(a,b) ((a) < (b) ? (b) : (a)) / MIN(a,b) ((a) > (b) ? (b) : (a)) in operand order: a pruned
    slot re-matched at learning rate 0 gets k = 0 / 0, and its NaN variance compares false both times and stays NaN (VERDICT r04
    weak-1: rounds 1-4 returned varMin).  [OCV-mem]: recalled macro definitions -- tests/test_opencv_crosscheck.py
    test_mog2_nan_variance_clamp checks the masks against a real cv2 where there is one."""
    rows, cols = 2, 2
    a = np.full((rows, cols, 3), 40, np.uint8)
    b = np.full((rows, cols, 3), 200, np.uint8)
    orc = O.Mog2(rows, cols, 3)
    for f, r in [(a, 0.3)] * 3 + [(b, 0.3)] + [(a, 0.3)] * 40:
        orc.apply(f, r)
    nm, w, v, m = orc.state()
    assert nm[0] == 2 and w[0, 1] == 0.0 and np.isfinite(v[0, 1])          # slot 1 was pruned and keeps its place
    mask = orc.apply(b, 0.0)                                                # ... and is matched again at rate 0
    nm, w, v, m = orc.state()
    assert np.isnan(v[0, 1]) and np.isnan(m[0, 1]).all() and w[0, 1] == 0.0 and v[0, 0] == 4.0
    assert (mask == 255).all()                                              # weight 0 < TB never makes it background... a plain foreground pixel
    assert (orc.apply(b, 0.0) == 255).all() and (orc.apply(a, 0.0) == 0).all()   # the NaN slot never matches again; mode 0 still does


# ---------------------------------------------------------------------------------------------------------------------------
# Fixtures harvested from a REAL OpenCV (tools/harvest_opencv_golden.py).  None exists yet: no image this
# repo has run in has a cv2 -- the oracle is "parity unpinned".  The day a box has one, `tools/probe_opencv.sh` leaves
# gpurun_out/opencv_golden/opencv_*.json; copied into tests/golden/ they are consumed HERE, on every CPU run, for good.
# ---------------------------------------------------------------------------------------------------------------------------
def _harvest():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("harvest_opencv_golden", os.path.join(root, "tools", "harvest_opencv_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["harvest_opencv_golden"] = mod
    spec.loader.exec_module(mod)
    return mod


def _check_opencv_fixture(H, name, body):
    """The oracle on the fixture's inputs (rebuilt from its recipe) against the outputs OpenCV gave.  -> what was compared."""
    if name == "mog2_trace":
        r = body["recipe"]
        verdict = {}
        for rate, packed in body["masks_by_rate"].items():
            want = H.unpack(packed)
            for restore in (1, 0):
                orc = O.Mog2(r["rows"], r["cols"], 3, params=dict(restore_nmodes=restore))
                frames = H.mog2_frames(r["seed"], r["rows"], r["cols"], r["frames"])
                verdict[(float(rate), restore)] = all((orc.apply(f, float(rate)) == want[t]).all() for t, f in enumerate(frames))
        good = [q for q in (1, 0) if all(v for (_, rq), v in verdict.items() if rq == q)]
        assert good, f"neither reading of the MOG2 mode count reproduces OpenCV {body['opencv_version']}: {verdict}"
        assert 1 in good, f"OpenCV {body['opencv_version']} prunes the mode count: flip the default of mog_restore_nmodes ({verdict})"
        return f"MOG2 masks, {len(verdict)} (rate, mode-count reading) runs"
    if name == "mog2_nan_clamp":
        want = H.unpack(body["masks"])
        orc = O.Mog2(4, 4, 3)
        for t, (f, rate) in enumerate(H.nan_clamp_sequence()):
            assert (orc.apply(f, rate) == want[t]).all(), t
        return f"NaN-clamp sequence, {len(want)} frames"
    if name == "contours":
        r = body["recipe"]
        for i, (img, want) in enumerate(zip(H.contour_images(r["seed"], r["n"]), body["contours"])):
            mine = O.find_contours(img)
            assert len(mine) == len(want), i
            for a, b in zip(mine, want):                  # SAME LIST ORDER: siftContours' tie-break depends on it
                assert list(a["start"]) == b["start"], i
                assert abs(a["m00"] - b["m00"]) < 1e-9 and abs(a["m10"] - b["m10"]) < 1e-6 and abs(a["m01"] - b["m01"]) < 1e-6, i
        return f"external contours of {r['n']} images: count, list order, moments"
    if name == "morphology":
        for k, d in body["by_k"].items():
            for img, e, di in zip(H.morph_images(), d["erode"], d["dilate"]):
                assert (O.erode(img, int(k)) == H.unpack(e)).all(), ("erode", k)
                assert (O.dilate(img, int(k)) == H.unpack(di)).all(), ("dilate (even k: anchor / reflection)", k)
        return f"rect erode / dilate, k in {sorted(int(k) for k in body['by_k'])}"
    if name == "hsv":
        bgr = np.random.default_rng(0).integers(0, 256, (256, 4096, 3), dtype=np.uint8)
        hsv = O.bgr2hsv(bgr)
        assert (hsv == H.unpack(body["hsv"])).all()
        assert (O.inrange3(hsv, (100, 150, 100), (125, 256, 256)) == H.unpack(body["inrange_100_150_100__125_256_256"])).all()
        return "BGR2HSV + inRange on 2^20 colours"
    raise AssertionError(f"unknown fixture {name}")


def test_oracle_against_harvested_opencv_fixtures(golden_dir):
    import glob
    files = sorted(glob.glob(os.path.join(golden_dir, "opencv_*.json")))
    if not files:
        pytest.skip("no tests/golden/opencv_*.json: no OpenCV has been reachable yet (tools/harvest_opencv_golden.py writes them "
                    "where `import cv2` works; tools/probe_opencv.sh runs it on every GPU box) -- parity unpinned")
    H = _harvest()
    for f in files:
        body = json.load(open(f))
        what = _check_opencv_fixture(H, os.path.basename(f)[len("opencv_"):-len(".json")], body)
        print(f"{os.path.basename(f)} (OpenCV {body['opencv_version']}): {what}: identical")


def test_the_harvest_and_its_consumer_run_end_to_end_on_a_producer_made_of_the_oracle(tmp_path):
    """PLUMBING ONLY, and it says so: there is no cv2 here, so the harvest script is driven with a producer that answers the
    cv2 calls it makes out of the ORACLE itself -- the fixtures written are the oracle's own outputs and pin nothing.  What this
    holds is that the day a real cv2 is passed in, harvest() -> JSON -> _check_opencv_fixture() works: recipes rebuild the same
    inputs, pack / unpack round-trip, every fixture kind has a consumer."""
    H = _harvest()

    class _Mog:
        def __init__(self):
            self.m = None

        def apply(self, f, learningRate=-1):
            if self.m is None:
                self.m = O.Mog2(f.shape[0], f.shape[1], 3)
            return self.m.apply(f, learningRate)

    class _OracleAsProducer:
        __version__ = "0.0-oracle-self-test"
        RETR_EXTERNAL = CHAIN_APPROX_SIMPLE = MORPH_RECT = COLOR_BGR2HSV = 0
        createBackgroundSubtractorMOG2 = staticmethod(lambda: _Mog())
        getStructuringElement = staticmethod(lambda shape, size: size[0])
        erode = staticmethod(lambda img, k: O.erode(img, k))
        dilate = staticmethod(lambda img, k: O.dilate(img, k))
        cvtColor = staticmethod(lambda img, code: O.bgr2hsv(img))
        inRange = staticmethod(lambda img, lo, hi: O.inrange3(img, lo, hi))

        @staticmethod
        def findContours(img, mode, method):
            cs = O.find_contours(img)
            return [[np.array([c["start"]]), c] for c in cs], None          # (contours, hierarchy); moments() reads c[1]

        @staticmethod
        def moments(c):
            return dict(m00=c[1]["m00"], m10=c[1]["m10"], m01=c[1]["m01"])
    written = H.harvest(_OracleAsProducer, str(tmp_path))
    assert sorted(os.path.basename(p) for p in written) == ["opencv_contours.json", "opencv_hsv.json", "opencv_mog2_nan_clamp.json",
                                                            "opencv_mog2_trace.json", "opencv_morphology.json"]
    for p in written:
        body = json.load(open(p))
        assert body["opencv_version"] == "0.0-oracle-self-test"
        assert _check_opencv_fixture(H, os.path.basename(p)[len("opencv_"):-len(".json")], body)
