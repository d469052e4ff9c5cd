"""CPU tests: the sequential OpenCV-3.1 border follower (oat_find_contours_external)
against two independent order-free restatements (the C crack formulation the HIP
kernels implement, and a numpy/scipy one written here)."""
import numpy as np
from scipy import ndimage as ndi

import oracle_lib as O

_T = [(-1, 0, -1, 1, 0, 1), (0, 1, 1, 1, 1, 0), (1, 0, 1, -1, 0, -1), (0, -1, -1, -1, -1, 0)]


def crack_all(img):
    H, W = img.shape
    fg = np.zeros((H, W), bool)
    fg[1:-1, 1:-1] = img[1:-1, 1:-1] > 0
    lab, _ = ndi.label(fg, structure=np.ones((3, 3)))
    bl, _ = ndi.label(~fg, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    outside = bl == bl[0, 0]
    res, first = {}, {}
    for y, x in zip(*np.nonzero(fg)):
        L = lab[y, x]
        first.setdefault(L, (int(x), int(y)))
        for bx, by, Bx, By, Ax, Ay in _T:
            if not outside[y + by, x + bx]:
                continue
            r = res.setdefault(L, [0, 0, 0])
            if fg[y + By, x + Bx]:
                qx, qy = x + Bx, y + By
            elif fg[y + Ay, x + Ax]:
                qx, qy = x + Ax, y + Ay
            else:
                continue
            d = int(x) * int(qy) - int(qx) * int(y)
            r[0] += d; r[1] += d * int(x + qx); r[2] += d * int(y + qy)
    return sorted((first[L], tuple(v)) for L, v in res.items())


def _images(n, seed):
    rng = np.random.default_rng(seed)
    for it in range(n):
        h, w = int(rng.integers(3, 48)), int(rng.integers(3, 48))
        img = (rng.random((h, w)) < rng.uniform(0.2, 0.85)).astype(np.uint8) * 255
        if it % 3 == 0:
            img = O.dilate(img, int(rng.integers(2, 4)))
        if it % 5 == 0:
            img = O.erode(img, 2)
        if it % 7 == 0:      # nested rings + salt noise: holes inside holes
            img[:] = 0
            step = int(rng.integers(2, 4))
            for k in range(0, min(h, w) // 2 - 1, step):
                img[k + 1:h - k - 1, k + 1:w - k - 1] = 255 if (k // step) % 2 == 0 else 0
            img ^= ((rng.random((h, w)) < 0.05) * 255).astype(np.uint8)
        yield img


def test_sequential_equals_order_free_full_list():
    for img in _images(600, 11):
        seq = sorted((c["start"], (int(c["a00"]), int(c["a10"]), int(c["a01"]))) for c in O.find_contours(img))
        assert seq == crack_all(img)


def test_sift_equal_between_formulations():
    rng = np.random.default_rng(12)
    for img in _images(1500, 13):
        lo = float(rng.choice([0.0, 0.0, 2.0, 10.0]))
        hi = float(rng.choice([np.finfo(np.float64).max, 50.0, 200.0]))
        assert O.sift_contours(img, lo, hi) == O.sift_cracks(img, lo, hi)


def test_degenerate_shapes():
    for shape in ((1, 1), (2, 5), (5, 2), (3, 3)):
        img = np.full(shape, 255, np.uint8)
        a, b = O.sift_contours(img), O.sift_cracks(img)
        assert a == b and not a["valid"]
