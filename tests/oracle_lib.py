"""ctypes binding of oracle/liboat_oracle.so (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(_ROOT, "oracle", "liboat_oracle.so")


class Mog2Params(C.Structure):
    _fields_ = [("history", C.c_int), ("nmixtures", C.c_int),
                ("var_threshold", C.c_float), ("background_ratio", C.c_float),
                ("var_threshold_gen", C.c_float), ("var_init", C.c_float),
                ("var_min", C.c_float), ("var_max", C.c_float), ("ct", C.c_float),
                ("detect_shadows", C.c_int), ("shadow_value", C.c_uint8), ("tau", C.c_float),
                ("restore_nmodes", C.c_int)]


class HsvParams(C.Structure):
    _fields_ = [("h_lo", C.c_int), ("h_hi", C.c_int), ("s_lo", C.c_int), ("s_hi", C.c_int),
                ("v_lo", C.c_int), ("v_hi", C.c_int), ("erode", C.c_int), ("dilate", C.c_int),
                ("min_area", C.c_double), ("max_area", C.c_double)]


class Detection(C.Structure):
    _fields_ = [("valid", C.c_int), ("x", C.c_double), ("y", C.c_double), ("area", C.c_double),
                ("a00", C.c_int64), ("a10", C.c_int64), ("a01", C.c_int64),
                ("first_pixel", C.c_int32)]

    def as_dict(self):
        return dict(valid=bool(self.valid), x=self.x, y=self.y, area=self.area,
                    a00=self.a00, a10=self.a10, a01=self.a01, first_pixel=self.first_pixel)


class Contour(C.Structure):
    _fields_ = [("start_x", C.c_int), ("start_y", C.c_int), ("npoints", C.c_int),
                ("first_point", C.c_int),
                ("a00", C.c_double), ("a10", C.c_double), ("a01", C.c_double),
                ("m00", C.c_double), ("m10", C.c_double), ("m01", C.c_double)]


class Contours(C.Structure):
    _fields_ = [("count", C.c_int), ("c", C.POINTER(Contour)), ("points", C.POINTER(C.c_int)),
                ("npoints_total", C.c_int)]


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_ROOT, "oracle")])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    u8p = C.POINTER(C.c_uint8)
    lib.oat_mog2_default_params.argtypes = [C.POINTER(Mog2Params)]
    lib.oat_mog2_create.restype = C.c_void_p
    lib.oat_mog2_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(Mog2Params)]
    lib.oat_mog2_destroy.argtypes = [C.c_void_p]
    lib.oat_mog2_apply.argtypes = [C.c_void_p, u8p, u8p, C.c_double]
    lib.oat_mog2_filter.argtypes = [C.c_void_p, u8p, u8p, C.c_double]
    lib.oat_mog2_filter_mt.argtypes = [C.c_void_p, u8p, u8p, C.c_double, C.c_int]
    lib.oat_mog2_nframes.argtypes = [C.c_void_p]
    lib.oat_mog2_modes_used.restype = u8p
    lib.oat_mog2_modes_used.argtypes = [C.c_void_p]
    fp = C.POINTER(C.c_float)
    lib.oat_mog2_get_state.argtypes = [C.c_void_p, fp, fp, fp]
    lib.oat_mog2_set_state.argtypes = [C.c_void_p, u8p, fp, fp, fp, C.c_int]
    lib.oat_bgr2hsv.argtypes = [u8p, u8p, C.c_size_t]
    ip = C.POINTER(C.c_int)
    lib.oat_inrange3.argtypes = [u8p, C.c_size_t, ip, ip, u8p]
    lib.oat_inrange1.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, u8p]
    lib.oat_erode_rect.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
    lib.oat_dilate_rect.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
    lib.oat_find_contours_external.restype = C.POINTER(Contours)
    lib.oat_find_contours_external.argtypes = [u8p, C.c_int, C.c_int]
    lib.oat_contours_free.argtypes = [C.POINTER(Contours)]
    lib.oat_sift_contours.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(Detection)]
    lib.oat_sift_cracks.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(Detection)]
    lib.oat_hsv_default_params.argtypes = [C.POINTER(HsvParams)]
    lib.oat_detect_hsv.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(HsvParams), u8p, C.POINTER(Detection)]
    lib.oat_detect_thresh.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(HsvParams), u8p, C.POINTER(Detection)]
    lib.oat_chain_step_from.argtypes = [C.c_void_p, u8p, u8p, C.c_int, C.c_int, C.c_double, C.POINTER(HsvParams),
                                        u8p, u8p, C.POINTER(Detection), C.c_int]
    lib.oat_chain_step.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_double, C.POINTER(HsvParams),
                                   u8p, u8p, C.POINTER(Detection), C.c_int]
    lib.oat_pipeline_run.restype = C.c_double
    lib.oat_pipeline_run.argtypes = [C.c_void_p, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                     C.POINTER(HsvParams), C.c_int, C.c_int, C.c_int, C.POINTER(Detection),
                                     C.POINTER(C.c_double)]
    lib.oat_bsub_create.restype = C.c_void_p
    lib.oat_bsub_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
    lib.oat_bsub_destroy.argtypes = [C.c_void_p]
    lib.oat_bsub_filter.argtypes = [C.c_void_p, u8p]
    lib.oat_bgr2grey.argtypes = [u8p, u8p, C.c_size_t]
    lib.oat_grey2bgr.argtypes = [u8p, u8p, C.c_size_t]
    lib.oat_hsv2bgr.argtypes = [u8p, u8p, C.c_size_t]
    lib.oat_cvt_color.argtypes = [C.c_int, C.c_int, u8p, u8p, C.c_size_t]
    lib.oat_cvt_color.restype = C.c_int
    lib.oat_thresh_filter.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, C.c_int]
    lib.oat_blur_box.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int]
    lib.oat_diff_create.restype = C.c_void_p
    lib.oat_diff_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
    lib.oat_diff_destroy.argtypes = [C.c_void_p]
    lib.oat_diff_detect.argtypes = [C.c_void_p, u8p, u8p, C.POINTER(Detection)]
    return lib


lib = _load()
# several processes of one job on one host (torch.distributed.run: bench.py --gpus N): every process's pinned row workers
# get their own stretch of the machine's cpus
if os.environ.get("LOCAL_RANK") and os.environ.get("LOCAL_WORLD_SIZE"):
    try:
        _lr, _lw = int(os.environ["LOCAL_RANK"]), max(int(os.environ["LOCAL_WORLD_SIZE"]), 1)
        lib.oat_pool_set_cpu_offset(_lr * ((os.cpu_count() or 1) // _lw))
    except (ValueError, AttributeError):
        pass
_u8p = C.POINTER(C.c_uint8)


def _p(a):
    return a.ctypes.data_as(_u8p)


def _c(a, dtype=np.uint8):
    return np.ascontiguousarray(a, dtype=dtype)


class Mog2:
    """cv::BackgroundSubtractorMOG2 with OpenCV defaults (oracle)."""

    def __init__(self, rows, cols, channels=3, params=None):
        self.rows, self.cols, self.ch = rows, cols, channels
        p = Mog2Params()
        lib.oat_mog2_default_params(C.byref(p))
        if params:
            for k, v in params.items():
                setattr(p, k, v)
        self.params = p
        self.h = lib.oat_mog2_create(rows, cols, channels, C.byref(p))

    def __del__(self):
        if getattr(self, "h", None):
            lib.oat_mog2_destroy(self.h)
            self.h = None

    def apply(self, image, lr):
        image = _c(image)
        mask = np.empty((self.rows, self.cols), np.uint8)
        lib.oat_mog2_apply(self.h, _p(image), _p(mask), float(lr))
        return mask

    def filter(self, frame, lr, nthreads=1):
        """BackgroundSubtractorMOG::filter: returns (filtered frame, mask)."""
        frame = _c(frame).copy()
        mask = np.empty((self.rows, self.cols), np.uint8)
        lib.oat_mog2_filter_mt(self.h, _p(frame), _p(mask), float(lr), int(nthreads))
        return frame, mask

    def set_state(self, nm, w, v, m, nframes):
        """Continue from a model exported elsewhere (HotPath.mog_state() order: modes_used, weight, variance, mean)."""
        nm = _c(nm)
        w, v, m = (np.ascontiguousarray(a, np.float32) for a in (w, v, m))
        f = C.POINTER(C.c_float)
        lib.oat_mog2_set_state(self.h, _p(nm), w.ctypes.data_as(f), v.ctypes.data_as(f), m.ctypes.data_as(f), int(nframes))

    def state(self):
        n = self.rows * self.cols
        k = self.params.nmixtures
        w = np.empty((n, k), np.float32)
        v = np.empty((n, k), np.float32)
        m = np.empty((n, k, self.ch), np.float32)
        f = C.POINTER(C.c_float)
        lib.oat_mog2_get_state(self.h, w.ctypes.data_as(f), v.ctypes.data_as(f), m.ctypes.data_as(f))
        nm = np.ctypeslib.as_array(lib.oat_mog2_modes_used(self.h), shape=(n,)).copy()
        return nm, w, v, m


def bgr2hsv(bgr):
    bgr = _c(bgr)
    out = np.empty_like(bgr)
    lib.oat_bgr2hsv(_p(bgr), _p(out), bgr.size // 3)
    return out


def inrange3(src, lo, hi):
    src = _c(src)
    n = src.size // 3
    out = np.empty(src.shape[:-1], np.uint8)
    lo_a = (C.c_int * 3)(*lo)
    hi_a = (C.c_int * 3)(*hi)
    lib.oat_inrange3(_p(src), n, lo_a, hi_a, _p(out))
    return out


def inrange1(src, lo, hi):
    src = _c(src)
    out = np.empty_like(src)
    lib.oat_inrange1(_p(src), src.size, int(lo), int(hi), _p(out))
    return out


def erode(img, k):
    img = _c(img)
    out = np.empty_like(img)
    lib.oat_erode_rect(_p(img), _p(out), img.shape[0], img.shape[1], int(k))
    return out


def dilate(img, k):
    img = _c(img)
    out = np.empty_like(img)
    lib.oat_dilate_rect(_p(img), _p(out), img.shape[0], img.shape[1], int(k))
    return out


def find_contours(img):
    """Returns list (in OpenCV list order) of dicts with start, points, sums, moments."""
    img = _c(img).copy()
    cs = lib.oat_find_contours_external(_p(img), img.shape[0], img.shape[1])
    out = []
    pts = cs.contents.points
    for i in range(cs.contents.count):
        c = cs.contents.c[i]
        p = [(pts[2 * (c.first_point + j)], pts[2 * (c.first_point + j) + 1]) for j in range(c.npoints)]
        out.append(dict(start=(c.start_x, c.start_y), points=p, a00=c.a00, a10=c.a10, a01=c.a01,
                        m00=c.m00, m10=c.m10, m01=c.m01))
    lib.oat_contours_free(cs)
    return out


def sift_contours(thr, min_area=0.0, max_area=float(np.finfo(np.float64).max)):
    thr = _c(thr).copy()
    d = Detection()
    lib.oat_sift_contours(_p(thr), thr.shape[0], thr.shape[1], min_area, max_area, C.byref(d))
    return d.as_dict()


def sift_cracks(thr, min_area=0.0, max_area=float(np.finfo(np.float64).max)):
    thr = _c(thr)
    d = Detection()
    lib.oat_sift_cracks(_p(thr), thr.shape[0], thr.shape[1], min_area, max_area, C.byref(d))
    return d.as_dict()


def hsv_params(**kw):
    p = HsvParams()
    lib.oat_hsv_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def detect_hsv(hsv, p):
    hsv = _c(hsv)
    thr = np.empty(hsv.shape[:2], np.uint8)
    d = Detection()
    lib.oat_detect_hsv(_p(hsv), hsv.shape[0], hsv.shape[1], C.byref(p), _p(thr), C.byref(d))
    return d.as_dict(), thr


def detect_thresh(grey, p):
    grey = _c(grey)
    thr = np.empty(grey.shape[:2], np.uint8)
    d = Detection()
    lib.oat_detect_thresh(_p(grey), grey.shape[0], grey.shape[1], C.byref(p), _p(thr), C.byref(d))
    return d.as_dict(), thr


def chain_step(mog, frame, lr, p, nthreads=1):
    """mog filter -> (bgr2hsv ->) detect on one frame (frame is not modified); GREY when mog.ch == 1."""
    frame = _c(frame)
    n = mog.rows * mog.cols
    # the work buffers live with the model (no 66 MB of fresh pages per 4K frame); the frame is copied row block by
    # row block inside the oracle's workers
    if getattr(mog, "_work", None) is None:
        mog._work = (np.empty(frame.size, np.uint8), np.empty(5 * n, np.uint8))
    work, scratch = mog._work
    thr = np.empty((mog.rows, mog.cols), np.uint8)
    d = Detection()
    lib.oat_chain_step_from(mog.h, _p(frame), _p(work), mog.rows, mog.cols, float(lr), C.byref(p),
                            _p(scratch), _p(thr), C.byref(d), int(nthreads))
    return d.as_dict(), thr


def pipeline_run(mog, frames, first, n, lr, p, t_front=1, t_mid=1, pipelined=True, keep=True):
    """oat_pipeline_run (oracle/pipeline.c): n frames, frame i = frames[(first + i) % len(frames)], through the chain
    with the reference's stage pipelining (three concurrent stages) or, pipelined=False, stage after stage.
    -> (wall seconds, [busy seconds of mog, col+inRange+morphology, contours], [detection dicts] or None)."""
    fs = [_c(f) for f in frames]
    ptrs = (_u8p * len(fs))(*[_p(f) for f in fs])
    out = (Detection * n)() if keep else None
    st = (C.c_double * 3)()
    el = lib.oat_pipeline_run(mog.h, ptrs, len(fs), int(first), int(n), mog.rows, mog.cols, float(lr), C.byref(p),
                              int(t_front), int(t_mid), 1 if pipelined else 0, out, st)
    if el < 0:
        raise MemoryError("oat_pipeline_run")
    return el, [st[0], st[1], st[2]], ([d.as_dict() for d in out] if keep else None)


def blur(img, k):
    img = _c(img)
    out = np.empty_like(img)
    lib.oat_blur_box(_p(img), _p(out), img.shape[0], img.shape[1], int(k))
    return out


class Diff:
    """posidet diff (DifferenceDetector) oracle."""

    def __init__(self, rows, cols, diff_threshold=10, blur=2, min_area=0.0, max_area=float(np.finfo(np.float64).max)):
        self.rows, self.cols = rows, cols
        self.h = lib.oat_diff_create(rows, cols, diff_threshold, blur, min_area, max_area)

    def __del__(self):
        if getattr(self, "h", None):
            lib.oat_diff_destroy(self.h)
            self.h = None

    def detect(self, grey):
        grey = _c(grey)
        thr = np.empty((self.rows, self.cols), np.uint8)
        d = Detection()
        lib.oat_diff_detect(self.h, _p(grey), _p(thr), C.byref(d))
        return d.as_dict(), thr


class Bsub:
    """framefilt bsub oracle."""

    def __init__(self, rows, cols, channels=3, alpha=0.0):
        self.h = lib.oat_bsub_create(rows, cols, channels, float(alpha))

    def __del__(self):
        if getattr(self, "h", None):
            lib.oat_bsub_destroy(self.h)
            self.h = None

    def filter(self, frame):
        f = _c(frame).copy()
        lib.oat_bsub_filter(self.h, _p(f))
        return f


def bgr2grey(bgr):
    bgr = _c(bgr)
    out = np.empty(bgr.shape[:-1], np.uint8)
    lib.oat_bgr2grey(_p(bgr), _p(out), out.size)
    return out


def grey2bgr(grey):
    grey = _c(grey)
    out = np.empty(grey.shape + (3,), np.uint8)
    lib.oat_grey2bgr(_p(grey), _p(out), grey.size)
    return out


def hsv2bgr(hsv):
    hsv = _c(hsv)
    out = np.empty_like(hsv)
    lib.oat_hsv2bgr(_p(hsv), _p(out), hsv.size // 3)
    return out


BINARY, GREY, BGR, HSV = 0, 1, 2, 3            # oat::PixelColor (Color.h:29-34)


def cvt_color(frame, src, dst):
    """ColorConvert::filter through oat::color_conv_table; returns (code, frame): code = bytes per
    output pixel, -1 nothing to be done, -2 not possible (frame None then)."""
    f = _c(frame)
    ch_in = 3 if src >= 2 else 1
    npx = f.size // ch_in
    out = np.empty(npx * 3, np.uint8)
    rc = lib.oat_cvt_color(int(src), int(dst), _p(f), _p(out), npx)
    if rc < 0:
        return rc, None
    shape = f.shape[:-1] if ch_in == 3 else f.shape
    return rc, (out[:npx * 3].reshape(shape + (3,)) if rc == 3 else out[:npx].reshape(shape))


def thresh_filter(frame, i_min, i_max):
    f = _c(frame).copy()
    ch = 3 if f.ndim == 3 else 1
    lib.oat_thresh_filter(_p(f), f.size // ch, ch, int(i_min), int(i_max))
    return f


def homography(h, valid, x, y, velocity_valid=False, vx=0.0, vy=0.0):
    """posifilt homography on one position (oat_homography_filter); returns (x, y, vx, vy)."""
    H = (C.c_double * 9)(*[float(v) for v in np.asarray(h, np.float64).reshape(9)])
    cx, cy, cvx, cvy = C.c_double(x), C.c_double(y), C.c_double(vx), C.c_double(vy)
    lib.oat_homography_filter.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.oat_homography_filter.restype = None
    lib.oat_homography_filter(H, int(bool(valid)), C.byref(cx), C.byref(cy), int(bool(velocity_valid)),
                              C.byref(cvx), C.byref(cvy))
    return cx.value, cy.value, cvx.value, cvy.value


class KalmanParams(C.Structure):
    _fields_ = [("dt", C.c_double), ("timeout", C.c_double), ("sigma_accel", C.c_double),
                ("sigma_noise", C.c_double)]


class KalmanOut(C.Structure):
    _fields_ = [("position_valid", C.c_int), ("velocity_valid", C.c_int), ("x", C.c_double), ("y", C.c_double),
                ("vx", C.c_double), ("vy", C.c_double)]


class Kalman:
    """posifilt kalman oracle (KalmanFilter2D.cpp:95-210)."""

    def __init__(self, dt=0.02, timeout=0.0, sigma_accel=5.0, sigma_noise=0.0):
        lib.oat_kalman_create.restype = C.c_void_p
        lib.oat_kalman_create.argtypes = [C.POINTER(KalmanParams)]
        lib.oat_kalman_destroy.argtypes = [C.c_void_p]
        lib.oat_kalman_filter.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.POINTER(KalmanOut)]
        p = KalmanParams(dt, timeout, sigma_accel, sigma_noise)
        self.h = lib.oat_kalman_create(C.byref(p))

    def __del__(self):
        if getattr(self, "h", None):
            lib.oat_kalman_destroy(self.h)
            self.h = None

    def filter(self, valid, x, y):
        o = KalmanOut()
        lib.oat_kalman_filter(self.h, int(bool(valid)), float(x), float(y), C.byref(o))
        return dict(position_valid=bool(o.position_valid), velocity_valid=bool(o.velocity_valid), x=o.x, y=o.y,
                    vx=o.vx, vy=o.vy)
