#!/usr/bin/env python3
"""Harvest golden vectors from a REAL OpenCV, wherever one turns up.

The oracle (oracle/*.c) restates OpenCV 3.1.0 routines Oat calls (README.md:1613; call sites
src/framefilter/BackgroundSubtractorMOG.cpp:82-83,124-125, src/framefilter/ColorConvert.cpp:104,
src/positiondetector/HSVDetector.cpp:146-156, src/positiondetector/DetectorFunc.cpp:41-50) and has never met the real
thing: no image this repo has run in has a cv2 (profiles/r02_opencv_probe.txt) -- "parity unpinned".  GPU boxes are fresh
for every call; should ONE of them ever have `import cv2`, this script turns that moment into fixtures that stay:

    python tools/harvest_opencv_golden.py --out gpurun_out/opencv_golden      (tools/probe_opencv.sh runs exactly this)
    cp gpurun_out/opencv_golden/opencv_*.json tests/golden/                   (then commit them)

Every fixture holds cv2.__version__, the RECIPE of its inputs (seeded numpy -- tests/test_oracle_golden.py rebuilds them bit
for bit through inputs_of() below) and OpenCV's outputs (zlib + base64).  The four [OCV-mem] points come first: the MOG2 mode
count reading, findContours' list order on tie-break images, the even-k dilation anchor, the NaN variance clamp.  Fixtures
are DATA: inputs and expected outputs, nothing of anybody's source.  Without cv2 the script says so and exits 0."""
import argparse
import base64
import json
import os
import sys
import zlib

import numpy as np


def pack(a):
    a = np.ascontiguousarray(a)
    return dict(dtype=str(a.dtype), shape=list(a.shape), z=base64.b64encode(zlib.compress(a.tobytes(), 9)).decode())


def unpack(d):
    return np.frombuffer(zlib.decompress(base64.b64decode(d["z"])), dtype=d["dtype"]).reshape(d["shape"]).copy()


# ---- input recipes (shared with the consumer test: same seeds, same arithmetic) ----
def mog2_frames(seed, rows, cols, n):
    """two colour fields a pixel flips between + noise: every mode-loop path is walked (fits, swaps, prunes, new modes)"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (rows, cols, 3)).astype(np.int16)
    alt = rng.integers(0, 256, (rows, cols, 3)).astype(np.int16)
    out = []
    for _ in range(n):
        f = np.where(rng.random((rows, cols, 1)) < 0.25, alt, base) + rng.integers(-12, 13, (rows, cols, 3))
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return out


def nan_clamp_sequence():
    a = np.full((4, 4, 3), 40, np.uint8)
    b = np.full((4, 4, 3), 200, np.uint8)
    return [(a, 0.3)] * 3 + [(b, 0.3)] + [(a, 0.3)] * 40 + [(b, 0.0)] * 3 + [(a, 0.0), (b, 0.0), (a, 0.01), (b, 0.01)] * 5


def contour_images(seed, n):
    """random binary images with an empty two-pixel ring (3.1.0 zeroes the frame, >= 3.2 pads: same result either way), the
    first four the tie-break shapes: equal rectangles side by side / above each other / nested in a ring / touching diagonally"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        img = np.zeros((50, 70), np.uint8)
        if i == 0:
            img[10:20, 5:25] = 255; img[10:20, 40:60] = 255
        elif i == 1:
            img[5:15, 20:40] = 255; img[30:40, 20:40] = 255
        elif i == 2:
            img[5:45, 5:65] = 255; img[10:40, 10:60] = 0; img[20:30, 25:45] = 255
        elif i == 3:
            img[10:20, 10:20] = 255; img[20:30, 20:30] = 255
        else:
            img = np.where(rng.random((50, 70)) < rng.choice([0.2, 0.5]), 255, 0).astype(np.uint8)
            img[:2, :] = img[-2:, :] = 0
            img[:, :2] = img[:, -2:] = 0
        out.append(img)
    return out


def morph_images():
    imp = np.zeros((31, 37), np.uint8)
    for y, x in ((0, 0), (0, 36), (30, 0), (30, 36), (15, 18), (0, 18), (15, 0)):
        imp[y, x] = 255
    rng = np.random.default_rng(41)
    return [imp, 255 - imp, np.where(rng.random((61, 83)) < 0.5, 255, 0).astype(np.uint8)]


def harvest(cv2, out):
    os.makedirs(out, exist_ok=True)
    ver = cv2.__version__
    written = []

    def dump(name, body):
        body = dict(opencv_version=ver, harvested_by="tools/harvest_opencv_golden.py", **body)
        p = os.path.join(out, f"opencv_{name}.json")
        with open(p, "w") as f:
            json.dump(body, f)
        written.append(p)
    # 1. MOG2 mask traces (BackgroundSubtractorMOG.cpp:82-83,124: all defaults, apply(frame, mask, rate))
    rows, cols, n = 64, 64, 40
    tr = {}
    for rate in (0.0, 0.01, 0.3):
        ref = cv2.createBackgroundSubtractorMOG2()
        masks = [ref.apply(f, learningRate=rate) for f in mog2_frames(5, rows, cols, n)]
        tr[str(rate)] = pack(np.stack(masks))
    dump("mog2_trace", dict(recipe=dict(fn="mog2_frames", seed=5, rows=rows, cols=cols, frames=n), masks_by_rate=tr))
    # 2. the NaN clamp sequence (one 4 x 4 image)
    ref = cv2.createBackgroundSubtractorMOG2()
    dump("mog2_nan_clamp", dict(recipe=dict(fn="nan_clamp_sequence"),
                                masks=pack(np.stack([ref.apply(f, learningRate=r) for f, r in nan_clamp_sequence()]))))
    # 3. findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) list order + moments (DetectorFunc.cpp:41-50)
    recs = []
    for img in contour_images(3, 60):
        res = cv2.findContours(img.copy(), cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)
        cs = res[-2]
        recs.append([dict(start=[int(v) for v in c[0][0]], m00=cv2.moments(c)["m00"], m10=cv2.moments(c)["m10"],
                          m01=cv2.moments(c)["m01"]) for c in cs])
    dump("contours", dict(recipe=dict(fn="contour_images", seed=3, n=60), contours=recs))
    # 4. rect erode / dilate incl. even k (HSVDetector.cpp:152-156, 253-273)
    mo = {}
    for k in (2, 3, 4, 7, 10, 13):
        el = cv2.getStructuringElement(cv2.MORPH_RECT, (k, k))
        mo[str(k)] = dict(erode=[pack(cv2.erode(i, el)) for i in morph_images()], dilate=[pack(cv2.dilate(i, el)) for i in morph_images()])
    dump("morphology", dict(recipe=dict(fn="morph_images"), by_k=mo))
    # 5. BGR2HSV on 2^20 random colours + inRange with a 256 upper bound (ColorConvert.cpp:104, HSVDetector.cpp:146-149)
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 256, (256, 4096, 3), dtype=np.uint8)
    hsv = cv2.cvtColor(bgr, cv2.COLOR_BGR2HSV)
    thr = cv2.inRange(hsv, (100, 150, 100), (125, 256, 256))
    dump("hsv", dict(recipe=dict(fn="rng0_bgr_256x4096"), hsv=pack(hsv), inrange_100_150_100__125_256_256=pack(thr)))
    return written


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join("gpurun_out", "opencv_golden"))
    a = ap.parse_args()
    try:
        import cv2
    except Exception as e:
        print(f"no cv2 ({type(e).__name__}: {e}): nothing to harvest; the oracle stays unpinned")
        return 0
    files = harvest(cv2, a.out)
    print(f"OpenCV {cv2.__version__}: wrote {len(files)} fixtures under {a.out}: copy opencv_*.json into tests/golden/ and commit")
    for f in files:
        print("  ", f, os.path.getsize(f), "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
