"""Deterministic synthetic camera streams for tests and bench.py.

Per stream s (seed 0x0A7 + s): a smooth mid-grey BGR gradient with fresh uniform
noise of amplitude +-6 every frame (MOG2 keeps 1-2 modes on most pixels), every
64th pixel flickering between two levels (forces >= 3 modes there), and 1-3
saturated coloured discs on Lissajous paths, kept away from the frame edge so
the centroid ground truth is unambiguous.  Host-side numpy only; no part of the
product or the oracle.
"""
import numpy as np

DISC_BGR = ((255, 64, 0), (0, 64, 255), (40, 255, 40))


class SyntheticStream:
    def __init__(self, rows, cols, stream=0, n_discs=1, radius=None, noise=6, flicker=True):
        self.rows, self.cols, self.stream = rows, cols, stream
        self.rng = np.random.default_rng(0x0A7 + stream)
        self.noise = noise
        yy, xx = np.mgrid[0:rows, 0:cols]
        self.yy, self.xx = yy, xx
        base = 110 + 30 * (xx / max(cols - 1, 1)) + 20 * (yy / max(rows - 1, 1))
        self.base = np.stack([base, base + 6, base - 5], -1).astype(np.int16)
        self.flick = None
        if flicker:
            self.flick = ((yy * cols + xx) % 64 == 17)
        rmin = max(4, min(rows, cols) // 40)
        self.discs = []
        for d in range(n_discs):
            r = radius if radius else int(self.rng.integers(rmin, 2 * rmin + 1))
            self.discs.append(dict(
                r=r, col=DISC_BGR[d % 3],
                ax=(cols / 2 - r - 24) * (0.55 + 0.15 * d), ay=(rows / 2 - r - 24) * (0.5 + 0.2 * d),
                fx=0.013 * (1 + d) + 0.002 * stream, fy=0.017 * (1 + 0.5 * d) + 0.001 * stream,
                px=self.rng.uniform(0, 6.28), py=self.rng.uniform(0, 6.28)))
        self.t = 0

    def disc_centres(self, t):
        out = []
        for d in self.discs:
            cx = self.cols / 2 + d["ax"] * np.sin(d["fx"] * t + d["px"])
            cy = self.rows / 2 + d["ay"] * np.sin(d["fy"] * t + d["py"])
            out.append((int(round(cx)), int(round(cy)), d["r"], d["col"]))
        return out

    def frame(self, t=None, with_discs=True):
        if t is None:
            t = self.t
            self.t += 1
        n = self.rng.integers(-self.noise, self.noise + 1, self.base.shape, dtype=np.int16)
        f = self.base + n
        if self.flick is not None:
            f[self.flick] += 60 if (t & 1) else -50
        f = np.clip(f, 0, 255).astype(np.uint8)
        if with_discs:
            for cx, cy, r, col in self.disc_centres(t):
                y0, y1 = max(cy - r, 0), min(cy + r + 1, self.rows)
                x0, x1 = max(cx - r, 0), min(cx + r + 1, self.cols)
                sub = (self.xx[y0:y1, x0:x1] - cx) ** 2 + (self.yy[y0:y1, x0:x1] - cy) ** 2 <= r * r
                f[y0:y1, x0:x1][sub] = col
        return f


def disc_hsv_window():
    """H/S/V windows that bracket DISC_BGR[0] = BGR(255,64,0) -> HSV(112,255,255); v_min >= 1 so
    pixels zeroed by the background subtractor fail the window."""
    return dict(h_thresh=(100, 125), s_thresh=(150, 256), v_thresh=(100, 256))


def make_pool(rows, cols, n_streams, n_frames, n_discs=1, warm=0):
    """[n_frames][n_streams, rows, cols, 3] uint8 frame sets (bench input pool)."""
    streams = [SyntheticStream(rows, cols, s, n_discs=n_discs) for s in range(n_streams)]
    pool = []
    for t in range(n_frames):
        pool.append(np.stack([st.frame(t + warm) for st in streams]))
    return pool
