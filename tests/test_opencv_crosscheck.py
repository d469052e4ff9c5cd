"""The oracle against a REAL OpenCV -- the check that would pin parity (SURVEY.md 8c last row).

Skipped where cv2 is missing, which is everywhere this repo has run so far: neither the build container nor the
GPU box has OpenCV in any form (profiles/r02_opencv_probe.txt).  It is written so that whoever has `cv2` only
has to run `pytest tests/test_opencv_crosscheck.py -rs`:

  * the reference pins OpenCV 3.1.0 (README.md:1613); cv2 >= 3.2 pads instead of zeroing the 1-px frame in
    findContours, so the contour images keep an empty border ring and both behaviours give the same result;
  * the three points the judge could not verify from memory are targeted first: the MOG2 mode count
    (`nmodes = nNewModes;`, oracle/mog2.c "Mode count") -- the test says WHICH reading matches --, the list order
    of findContours (tie-break) and the even-k dilation anchor.
Call sites being checked: BackgroundSubtractorMOG.cpp:82-83,124; ColorConvert.cpp:104; HSVDetector.cpp:146-156;
DetectorFunc.cpp:41-63."""
import numpy as np
import pytest

import oracle_lib as O

cv2 = pytest.importorskip("cv2", reason="no OpenCV in this image (profiles/r02_opencv_probe.txt): parity stays unpinned")


def test_bgr2hsv_matches_cvtcolor():
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (256, 4096, 3), dtype=np.uint8)
    assert (O.bgr2hsv(bgr) == cv2.cvtColor(bgr, cv2.COLOR_BGR2HSV)).all()


def test_the_other_framefilt_col_conversions_match_cvtcolor():
    """Color.h:45-51: BGR2GRAY, GRAY2BGR (any version) and HSV2BGR -- the latter only against OpenCV < 3.4, the
    float routine the oracle restates; later versions convert in fixed point and differ by a level here and there."""
    rng = np.random.default_rng(1)
    px = rng.integers(0, 256, (256, 4096, 3), dtype=np.uint8)
    assert (O.bgr2grey(px) == cv2.cvtColor(px, cv2.COLOR_BGR2GRAY)).all()
    assert (O.grey2bgr(px[..., 0]) == cv2.cvtColor(px[..., 0].copy(), cv2.COLOR_GRAY2BGR)).all()
    diff = np.abs(O.hsv2bgr(px).astype(int) - cv2.cvtColor(px, cv2.COLOR_HSV2BGR).astype(int)).max()
    major, minor = (int(v) for v in cv2.__version__.split(".")[:2])
    assert diff == 0 if (major, minor) < (3, 4) else diff <= 1, diff


@pytest.mark.parametrize("k", [2, 3, 4, 7, 10])
def test_rect_morphology_matches_cv(k):
    rng = np.random.default_rng(k)
    img = np.where(rng.random((61, 83)) < 0.5, 255, 0).astype(np.uint8)
    el = cv2.getStructuringElement(cv2.MORPH_RECT, (k, k))
    assert (O.erode(img, k) == cv2.erode(img, el)).all()
    assert (O.dilate(img, k) == cv2.dilate(img, el)).all(), "even-k dilation anchor / reflection differs"


def _ringed(img):
    img = img.copy()
    img[:2, :] = img[-2:, :] = 0
    img[:, :2] = img[:, -2:] = 0
    return img


def test_external_contours_moments_and_list_order_match_cv():
    rng = np.random.default_rng(3)
    for i in range(200):
        img = _ringed(np.where(rng.random((50, 70)) < rng.choice([0.2, 0.5]), 255, 0).astype(np.uint8))
        res = cv2.findContours(img.copy(), cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)
        contours = res[-2]
        mine = O.find_contours(img)
        assert len(mine) == len(contours), i
        for a, b in zip(mine, contours):                      # SAME ORDER: the tie-break of siftContours depends on it
            m = cv2.moments(b)
            assert a["start"] == tuple(int(v) for v in b[0][0]), i
            assert abs(a["m00"] - m["m00"]) < 1e-9 and abs(a["m10"] - m["m10"]) < 1e-6 and abs(a["m01"] - m["m01"]) < 1e-6, i


def test_mog2_masks_match_cv_and_report_the_mode_count_reading():
    rng = np.random.default_rng(11)
    rows, cols = 48, 64
    base = rng.integers(0, 256, (rows, cols, 3)).astype(np.int16)
    alt = rng.integers(0, 256, (rows, cols, 3)).astype(np.int16)
    verdict = {}
    for rate in (0.0, 0.01, 0.3):
        for restore in (1, 0):
            ref = cv2.createBackgroundSubtractorMOG2()
            orc = O.Mog2(rows, cols, 3, params=dict(restore_nmodes=restore))
            r2 = np.random.default_rng(5)
            same = True
            for t in range(150):
                f = np.where(r2.random((rows, cols, 1)) < 0.25, alt, base) + r2.integers(-12, 13, (rows, cols, 3))
                f = np.clip(f, 0, 255).astype(np.uint8)
                same = same and bool((ref.apply(f, learningRate=rate) == orc.apply(f, rate)).all())
            verdict[(rate, restore)] = same
    print("MOG2 mask identity with cv2 by (rate, restore_nmodes):", verdict)
    assert verdict[(0.0, 1)] and verdict[(0.0, 0)], "frozen-model masks differ from OpenCV"
    good = [r for r in (1, 0) if all(verdict[(rate, r)] for rate in (0.0, 0.01, 0.3))]
    assert good, f"neither reading of the mode count reproduces OpenCV {cv2.__version__}: {verdict}"
    assert 1 in good, ("OpenCV %s prunes the mode count (restore_nmodes = 0): flip the default of "
                       "oatgpu_config.mog_restore_nmodes / oat_mog2_params.restore_nmodes" % cv2.__version__)


def test_mog2_nan_variance_clamp():
    """The fourth open point: `varnew = MAX(varnew, varMin); varnew = MIN(varnew, varMax);` when varnew
    is NaN.  One pixel: learn at 0.3 until a second mode has been created and pruned again (its slot keeps weight 0 with
    `nmodes = nNewModes;`), then show that slot's colour at learning rate 0 -- k = alphaT / weight = 0 / 0.  With OpenCV's
    macros (a comparison with a NaN is false) the variance stays NaN and that slot can never match again; had the clamp
    returned varMin the slot would still be NaN in its mean -- the masks cannot tell the two apart, so this test reads the
    oracle's STATE and compares the masks of a long tail with cv2: it passes if the oracle's reading (NaN kept) is
    at least mask-compatible, and prints the oracle's variance for the record."""
    rows, cols = 4, 4
    a = np.full((rows, cols, 3), 40, np.uint8)
    b = np.full((rows, cols, 3), 200, np.uint8)
    ref = cv2.createBackgroundSubtractorMOG2()
    orc = O.Mog2(rows, cols, 3)
    seq = [(a, 0.3)] * 3 + [(b, 0.3)] + [(a, 0.3)] * 40 + [(b, 0.0)] * 3 + [(a, 0.0), (b, 0.0), (a, 0.01), (b, 0.01)] * 5
    for t, (f, rate) in enumerate(seq):
        assert (ref.apply(f, learningRate=rate) == orc.apply(f, rate)).all(), t
    nm, w, v, m = orc.state()
    print("oracle variances of pixel 0 after the 0/0 update:", v[0].tolist(), "weights:", w[0].tolist())
