"""A SECOND, independently written statement of the integer stages, built on what this image does have
(scipy.ndimage), so that the oracle and its golden vectors stop sharing one author and one formulation
(VERDICT r01, next-round item 1c).  Nothing here follows borders or uses Green's theorem:

  * morphology: scipy's minimum/maximum filters with an explicit window (OpenCV's anchor k/2, constant
    border 1 for erosion / 0 for dilation = morphologyDefaultBorderValue for 8-bit masks);
  * external contours: 8-connected labelling, hole filling with a 4-connected background flood, and the polygon
    through the border pixels' centres measured by CELL DECOMPOSITION -- every 2x2 block of pixel centres with
    four corners inside the filled component is a unit square, with three corners a half-square triangle --
    which gives cv::moments' m00, m10, m01 of the outer contour without ever tracing it.

What this cannot settle is what only a real OpenCV can: tests/test_opencv_crosscheck.py (skipped without cv2)."""
import numpy as np
import pytest
from scipy import ndimage

import oracle_lib as O

EIGHT = np.ones((3, 3), bool)


def _zero_frame(img):
    """cvStartFindContours in OpenCV 3.1 zeroes the 1-pixel image frame before following borders."""
    f = (np.asarray(img) != 0)
    f = f.copy()
    f[0, :] = f[-1, :] = False
    f[:, 0] = f[:, -1] = False
    return f


def external_components_by_cells(img):
    """{first pixel (x, y) in raster order: (m00, m10, m01)} of every EXTERNAL 8-connected component."""
    f = _zero_frame(img)
    lab, n = ndimage.label(f, structure=EIGHT)
    # outside = background 4-connected to the frame (the frame itself is background now)
    bg, _ = ndimage.label(~f)                       # default structure = 4-connectivity
    outside = bg == bg[0, 0]
    touches_outside = ndimage.binary_dilation(outside, structure=EIGHT)   # fg is 8-connected: diagonal contact counts
    out = {}
    for k in range(1, n + 1):
        comp = lab == k
        if not (comp & touches_outside).any():
            continue                                 # nested in a hole of another component: not RETR_EXTERNAL
        filled = ndimage.binary_fill_holes(comp)     # 4-connected background flood: what the outer border encloses
        a, b = filled[:-1, :-1], filled[:-1, 1:]
        c, d = filled[1:, :-1], filled[1:, 1:]
        cnt = a.astype(int) + b + c + d
        ys, xs = np.mgrid[0:filled.shape[0] - 1, 0:filled.shape[1] - 1]
        full = cnt == 4
        m00 = full.sum() * 1.0
        m10 = (xs[full] + 0.5).sum()
        m01 = (ys[full] + 0.5).sum()
        tri = cnt == 3
        # centroid of the triangle = mean of its three present corners
        sx = (a * xs + b * (xs + 1) + c * xs + d * (xs + 1))[tri] / 3.0
        sy = (a * ys + b * ys + c * (ys + 1) + d * (ys + 1))[tri] / 3.0
        m00 += 0.5 * tri.sum()
        m10 += 0.5 * sx.sum()
        m01 += 0.5 * sy.sum()
        yy, xx = np.nonzero(comp)
        first = np.lexsort((xx, yy))[0]
        out[(int(xx[first]), int(yy[first]))] = (m00, m10, m01)
    return out


def _random_mask(rng, h, w, kind):
    img = np.zeros((h, w), np.uint8)
    if kind == "noise":
        img[rng.random((h, w)) < rng.choice([0.15, 0.4, 0.6])] = 255
    elif kind == "blobs":
        yy, xx = np.mgrid[0:h, 0:w]
        for _ in range(rng.integers(1, 7)):
            cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(1, max(3, min(h, w) // 3))
            if rng.random() < 0.5:
                img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 255
            else:
                img[max(cy - r, 0):cy + r, max(cx - r // 2, 0):cx + r] = 255
        for _ in range(rng.integers(0, 4)):          # punch holes, some with islands inside
            cy, cx, r = rng.integers(0, h), rng.integers(0, w), rng.integers(1, 6)
            img[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0
            if rng.random() < 0.5:
                img[cy, cx] = 255
    else:                                            # thin lines and diagonal chains
        for _ in range(rng.integers(1, 5)):
            y, x = rng.integers(0, h), rng.integers(0, w)
            for _ in range(rng.integers(3, 40)):
                img[y % h, x % w] = 255
                y += rng.integers(-1, 2)
                x += rng.integers(-1, 2)
    return img


def test_external_contour_moments_by_cell_decomposition():
    rng = np.random.default_rng(20260930)
    checked = 0
    for i in range(400):
        h, w = rng.integers(3, 70), rng.integers(3, 90)
        img = _random_mask(rng, h, w, ("noise", "blobs", "lines")[i % 3])
        want = external_components_by_cells(img)
        got = {c["start"]: c for c in O.find_contours(img)}
        assert set(got) == set(want), (i, sorted(got), sorted(want))
        for k, (m00, m10, m01) in want.items():
            c = got[k]
            assert abs(c["m00"] - m00) < 1e-9, (i, k, c["m00"], m00)
            assert abs(c["m10"] - m10) < 1e-6 * max(1.0, abs(m10)) and abs(c["m01"] - m01) < 1e-6 * max(1.0, abs(m01)), (i, k)
            checked += 1
    assert checked > 2000


def test_selected_blob_is_the_largest_external_component():
    """siftContours (DetectorFunc.cpp:41-63) = largest m00 inside [min, max); centroid = m10/m00, m01/m00."""
    rng = np.random.default_rng(7)
    hits = 0
    for i in range(200):
        img = _random_mask(rng, 60, 80, "blobs")
        comps = external_components_by_cells(img)
        lo, hi = 3.0, 900.0
        cands = sorted(((m[0], k) for k, m in comps.items() if lo <= m[0] < hi), reverse=True)
        d = O.sift_contours(img, lo, hi)
        if not cands:
            assert not d["valid"]
            continue
        if len(cands) > 1 and cands[0][0] == cands[1][0]:
            continue                                 # area ties are the tie-break tests' business
        m00, m10, m01 = comps[cands[0][1]]
        assert d["valid"] and d["area"] == m00
        assert abs(d["x"] - m10 / m00) < 1e-9 and abs(d["y"] - m01 / m00) < 1e-9
        hits += 1
    assert hits > 100


@pytest.mark.parametrize("k", [2, 3, 4, 5, 7, 10, 13])
def test_rect_morphology_equals_scipy_window_filters(k):
    """erode / dilate with a k x k rectangle: window [x - k//2, x - k//2 + k - 1] in both directions (anchor k/2,
    the kernel NOT reflected for dilation), outside the image 1 for erosion and 0 for dilation."""
    rng = np.random.default_rng(k)
    for shape in ((37, 53), (8, 5), (64, 64), (1, 9)):
        img = np.where(rng.random(shape) < 0.55, 255, 0).astype(np.uint8)
        ero = ndimage.minimum_filter(img, size=k, mode="constant", cval=255, origin=0)
        dil = ndimage.maximum_filter(img, size=k, mode="constant", cval=0, origin=0)
        assert (O.erode(img, k) == ero).all(), (k, shape)
        assert (O.dilate(img, k) == dil).all(), (k, shape)


def test_inrange_equals_numpy():
    rng = np.random.default_rng(1)
    hsv = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    for lo, hi in (((0, 0, 0), (256, 256, 256)), ((100, 150, 100), (125, 256, 256)), ((30, 40, 50), (20, 256, 256)),
                   ((0, 0, 255), (180, 255, 255))):
        l = np.array(lo)
        h = np.minimum(np.array(hi), 255)
        want = ((hsv >= l) & (hsv <= h)).all(-1) & (np.array(lo) <= np.array(hi)).all()
        assert ((O.inrange3(hsv, lo, hi) != 0) == want).all(), (lo, hi)
