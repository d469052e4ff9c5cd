"""Host-side mirror of the reference's operator interface for the hot path.

Class and method names follow jonnew/Oat so that the parity tests read like the
reference's call sites:

    BackgroundSubtractorMOG.filter(frame)      src/framefilter/BackgroundSubtractorMOG.cpp:114-127
    ColorConvert.filter(frame)                 src/framefilter/ColorConvert.cpp:101-107
    HSVDetector.detectPosition(frame, pos)     src/positiondetector/HSVDetector.cpp:142-173
    SimpleThreshold.detectPosition(frame, pos) src/positiondetector/SimpleThreshold.cpp:114-134

``HotPath`` is the fused, batched form (N camera streams, one launch per stage).
All pixel work happens in liboatgpu.so on the GPU.
"""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import ffi

DBL_MAX = float(np.finfo(np.float64).max)


@dataclass
class Position2D:
    """The fields posidet writes into oat::Position2D (lib/datatypes/Position2D.h:112-127)."""
    position_valid: bool = False
    x: float = 0.0
    y: float = 0.0
    area: float = 0.0          # siftContours' object_area out-parameter
    a00: int = 0
    a10: int = 0
    a01: int = 0
    first_pixel: int = -1
    velocity_valid: bool = False   # set by the position filter (HotPath.set_kalman)
    vx: float = 0.0
    vy: float = 0.0
    raw_valid: bool = False        # the detector's own result when the filter is on
    raw_x: float = 0.0
    raw_y: float = 0.0

    @staticmethod
    def from_c(p):
        return Position2D(bool(p.valid), p.x, p.y, p.area, p.a00, p.a10, p.a01, p.first_pixel,
                          bool(p.velocity_valid), p.vx, p.vy, bool(p.raw_valid), p.raw_x, p.raw_y)


def _frame(a, shape):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.shape != shape:
        raise ValueError(f"expected frame of shape {shape}, got {a.shape}")
    return a


class _Context:
    """Owns one oatgpu_ctx."""

    def __init__(self, rows, cols, n_streams=1, device=0, ring_depth=4, **overrides):
        self.lib = ffi.load()
        cfg = ffi.Config()
        self.lib.oatgpu_default_config(C.byref(cfg))
        cfg.rows, cfg.cols, cfg.n_streams, cfg.device, cfg.ring_depth = rows, cols, n_streams, device, ring_depth
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown option {k}")
            setattr(cfg, k, v)
        self.cfg = cfg
        self.rows, self.cols, self.n_streams = rows, cols, n_streams
        self.channels = cfg.channels
        self.frame_shape = (rows, cols, 3) if cfg.channels == 3 else (rows, cols)
        self.ctx = self.lib.oatgpu_create(C.byref(cfg))
        if not self.ctx:
            raise ffi.OatGpuError(-1, (self.lib.oatgpu_last_error(None) or b"").decode())

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.oatgpu_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        ffi.check(self.lib, self.ctx, rc)

    def set_roi_mask(self, mask, stream=0):
        """framefilt mask fused in front of mog: mask (rows, cols) uint8, nonzero = keep; None clears."""
        if mask is None:
            self._chk(self.lib.oatgpu_set_roi_mask(self.ctx, stream, None))
        else:
            m = _frame(mask, (self.rows, self.cols))
            self._chk(self.lib.oatgpu_set_roi_mask(self.ctx, stream, ffi.u8(m)))

    def traffic_audit(self, on=True):
        """Count what the fused per-pixel kernel loads and stores (slower launches; never time them)."""
        self._chk(self.lib.oatgpu_traffic_audit(self.ctx, 1 if on else 0))

    def traffic_read(self):
        t = ffi.Traffic()
        self._chk(self.lib.oatgpu_traffic_read(self.ctx, C.byref(t)))
        return {n: getattr(t, n) for n, _ in ffi.Traffic._fields_}

    def measure_hbm(self, nbytes=1 << 30, reps=5):
        """(read_GBps, copy_GBps) of plain streaming kernels on this device."""
        r, c = C.c_double(0), C.c_double(0)
        self._chk(self.lib.oatgpu_measure_hbm(self.ctx, int(nbytes), int(reps), C.byref(r), C.byref(c)))
        return r.value, c.value

    # -- taps ------------------------------------------------------------
    def read_mask(self, which=ffi.TAP_MORPH, stream=0):
        out = np.empty((self.rows, self.cols), np.uint8)
        self._chk(self.lib.oatgpu_read_mask(self.ctx, stream, which, ffi.u8(out)))
        return out

    def mog_state(self, stream=0):
        n, k = self.rows * self.cols, self.cfg.nmixtures
        nm = np.empty(n, np.uint8)
        w = np.empty((n, k), np.float32)
        v = np.empty((n, k), np.float32)
        m = np.empty((n, k, self.channels), np.float32)
        nf = C.c_int32(0)
        self._chk(self.lib.oatgpu_mog_get_state(self.ctx, stream, ffi.u8(nm), ffi.f32(w), ffi.f32(v), ffi.f32(m),
                                                C.byref(nf)))
        return nm, w, v, m, nf.value

    def set_mog_state(self, nm, w, v, m, nframes, stream=0):
        nm = np.ascontiguousarray(nm, np.uint8)
        w, v, m = (np.ascontiguousarray(a, np.float32) for a in (w, v, m))
        self._chk(self.lib.oatgpu_mog_set_state(self.ctx, stream, ffi.u8(nm), ffi.f32(w), ffi.f32(v), ffi.f32(m),
                                                int(nframes)))


    def save_mog_state(self, path, stream=0):
        """Checkpoint the MOG2 model of one stream to a file (oatgpu_mog_save)."""
        self._chk(self.lib.oatgpu_mog_save(self.ctx, stream, os.fsencode(path)))

    def load_mog_state(self, path, stream=0):
        """Resume from a checkpoint; geometry / channels / mixture count must match."""
        self._chk(self.lib.oatgpu_mog_load(self.ctx, stream, os.fsencode(path)))


class BackgroundSubtractorMOG(_Context):
    """framefilt mog.  ``adaptation_coeff`` is the reference's -a option (default 0)."""

    def __init__(self, rows, cols, adaptation_coeff=0.0, **kw):
        if not 0.0 <= adaptation_coeff <= 1.0:
            raise ValueError("adaptation-coeff must be in [0, 1]")   # BackgroundSubtractorMOG.cpp:86-88
        super().__init__(rows, cols, **kw)
        self.learning_coeff_ = float(adaptation_coeff)

    def filter(self, frame, stream=0):
        """In place, like FrameFilter::filter(cv::Mat&); also returns the frame."""
        f = _frame(frame, self.frame_shape)
        out = frame if (isinstance(frame, np.ndarray) and frame.flags.c_contiguous and frame.dtype == np.uint8) else f
        self._chk(self.lib.oatgpu_mog_filter(self.ctx, stream, ffi.u8(f), ffi.u8(out), self.learning_coeff_))
        return out

    def apply(self, frame, learning_rate=None, stream=0):
        """cv::BackgroundSubtractorMOG2::apply: returns the {0,127,255} mask."""
        f = _frame(frame, self.frame_shape)
        mask = np.empty((self.rows, self.cols), np.uint8)
        lr = self.learning_coeff_ if learning_rate is None else float(learning_rate)
        self._chk(self.lib.oatgpu_mog_apply(self.ctx, stream, ffi.u8(f), ffi.u8(mask), lr))
        return mask


class BackgroundSubtractor(_Context):
    """framefilt bsub (BackgroundSubtractor.cpp): -a adaptation coefficient, first frame = background."""

    def __init__(self, rows, cols, adaptation_coeff=0.0, **kw):
        super().__init__(rows, cols, **kw)
        self.alpha_ = float(adaptation_coeff)

    def filter(self, frame, stream=0):
        f = _frame(frame, self.frame_shape)
        out = np.empty_like(f)
        self._chk(self.lib.oatgpu_bsub_filter(self.ctx, stream, ffi.u8(f), ffi.u8(out), self.alpha_))
        return out


class Threshold(_Context):
    """framefilt thresh (Threshold.cpp): -I [min,max] intensity passband."""

    def __init__(self, rows, cols, intensity=(0, 256), **kw):
        super().__init__(rows, cols, **kw)
        self.i_min_, self.i_max_ = int(intensity[0]), int(intensity[1])

    def filter(self, frame):
        f = _frame(frame, self.frame_shape)
        out = np.empty_like(f)
        self._chk(self.lib.oatgpu_thresh_filter(self.ctx, ffi.u8(f), ffi.u8(out), self.i_min_, self.i_max_))
        return out


PIX_BINARY, PIX_GREY, PIX_BGR, PIX_HSV = 0, 1, 2, 3      # oat::PixelColor (Color.h:29-34)
_COLOR_NAMES = {"BINARY": PIX_BINARY, "GREY": PIX_GREY, "BGR": PIX_BGR, "HSV": PIX_HSV}


class ColorConvert(_Context):
    """framefilt col -C COLOR (ColorConvert.cpp): `color` = the colour to convert to, as the reference's
    --color option (default HSV, the bridge in front of `posidet hsv`); the source colour is what the
    frame source carries (`from_color`, default BGR).  Pairs with nothing to do or not possible raise
    with the reference's texts."""

    def __init__(self, rows, cols, color="HSV", from_color="BGR", **kw):
        super().__init__(rows, cols, **kw)
        self.color_ = _COLOR_NAMES[color] if isinstance(color, str) else int(color)
        self.from_ = _COLOR_NAMES[from_color] if isinstance(from_color, str) else int(from_color)

    def filter(self, frame):
        shape_in = (self.rows, self.cols, 3) if self.from_ >= PIX_BGR else (self.rows, self.cols)
        f = _frame(frame, shape_in)
        out = np.empty((self.rows, self.cols, 3) if self.color_ >= PIX_BGR else (self.rows, self.cols), np.uint8)
        self._chk(self.lib.oatgpu_cvt_color(self.ctx, self.from_, self.color_, ffi.u8(f), ffi.u8(out)))
        return out


class _Detector(_Context):
    def _set(self, **kw):
        c = self.cfg
        for k, v in kw.items():
            setattr(c, k, v)
        self._chk(self.lib.oatgpu_set_detector(self.ctx, c.h_lo, c.h_hi, c.s_lo, c.s_hi, c.v_lo, c.v_hi,
                                               c.erode, c.dilate, c.min_area, c.max_area))


class HSVDetector(_Detector):
    """posidet hsv.  Options follow HSVDetector.cpp:49-75 (-H -S -V -e -d -a)."""

    def __init__(self, rows, cols, h_thresh=(0, 256), s_thresh=(0, 256), v_thresh=(0, 256),
                 erode=0, dilate=10, area=(0.0, DBL_MAX), **kw):
        super().__init__(rows, cols, h_lo=h_thresh[0], h_hi=h_thresh[1], s_lo=s_thresh[0], s_hi=s_thresh[1],
                         v_lo=v_thresh[0], v_hi=v_thresh[1], erode=erode, dilate=dilate,
                         min_area=area[0], max_area=area[1], **kw)

    def detectPosition(self, frame, position=None, stream=0):
        f = _frame(frame, (self.rows, self.cols, 3))
        p = ffi.Position()
        self._chk(self.lib.oatgpu_detect_hsv(self.ctx, stream, ffi.u8(f), C.byref(p)))
        return _merge(position, p)


class SimpleThreshold(_Detector):
    """posidet thresh.  Options follow SimpleThreshold.cpp:49-69 (-T -e -d -a); erode/dilate default off."""

    def __init__(self, rows, cols, thresh=(0, 256), erode=0, dilate=0, area=(0.0, DBL_MAX), **kw):
        super().__init__(rows, cols, h_lo=thresh[0], h_hi=thresh[1], erode=erode, dilate=dilate,
                         min_area=area[0], max_area=area[1], **kw)

    def detectPosition(self, frame, position=None, stream=0):
        f = _frame(frame, (self.rows, self.cols))
        p = ffi.Position()
        self._chk(self.lib.oatgpu_detect_thresh(self.ctx, stream, ffi.u8(f), C.byref(p)))
        return _merge(position, p)


class DifferenceDetector(_Detector):
    """posidet diff.  Options follow DifferenceDetector.cpp:47-64 (-d diff-threshold, -b blur, -a area)."""

    def __init__(self, rows, cols, diff_threshold=10, blur=2, area=(0.0, DBL_MAX), **kw):
        super().__init__(rows, cols, diff_threshold=diff_threshold, blur=blur, erode=0, dilate=0,
                         min_area=area[0], max_area=area[1], **kw)

    def detectPosition(self, frame, position=None, stream=0):
        f = _frame(frame, (self.rows, self.cols))
        p = ffi.Position()
        self._chk(self.lib.oatgpu_detect_diff(self.ctx, stream, ffi.u8(f), C.byref(p)))
        return _merge(position, p)


def _merge(position, p):
    """siftContours only writes x/y when a blob is found (DetectorFunc.cpp:46,58-60)."""
    new = Position2D.from_c(p)
    if position is None:
        return new
    position.position_valid = new.position_valid
    position.area = new.area
    if new.position_valid:
        position.x, position.y = new.x, new.y
    position.a00, position.a10, position.a01, position.first_pixel = new.a00, new.a10, new.a01, new.first_pixel
    return position


class HotPath(_Context):
    """The fused chain for N camera streams: mog + setTo + BGR2HSV + inRange + erode + dilate + blob."""

    def __init__(self, rows, cols, n_streams=1, adaptation_coeff=0.0, h_thresh=(0, 256), s_thresh=(0, 256),
                 v_thresh=(0, 256), erode=0, dilate=10, area=(0.0, DBL_MAX), **kw):
        super().__init__(rows, cols, n_streams=n_streams, h_lo=h_thresh[0], h_hi=h_thresh[1], s_lo=s_thresh[0],
                         s_hi=s_thresh[1], v_lo=v_thresh[0], v_hi=v_thresh[1], erode=erode, dilate=dilate,
                         min_area=area[0], max_area=area[1], **kw)
        self.learning_coeff_ = float(adaptation_coeff)
        self._pos = (ffi.Position * n_streams)()

    def _out(self):
        return [Position2D.from_c(p) for p in self._pos]

    def track(self, frames):
        """frames: sequence of n_streams host arrays (rows, cols, 3), or (rows, cols) when channels == 1."""
        fs = [_frame(f, self.frame_shape) for f in frames]
        ptrs = (ffi._u8p * len(fs))(*[ffi.u8(f) for f in fs])
        self._chk(self.lib.oatgpu_track_batch(self.ctx, ptrs, len(fs), self.learning_coeff_, self._pos))
        return self._out()

    def track_dev(self, dev_ptr):
        """dev_ptr: device address of n_streams*rows*cols*3 bytes (e.g. torch tensor .data_ptr())."""
        self._chk(self.lib.oatgpu_track_batch_dev(self.ctx, C.c_void_p(dev_ptr), self.learning_coeff_, self._pos))
        return self._out()

    def track_sequence_dev(self, dev_ptrs, timed=False):
        """A recorded sequence in one call (oatgpu_track_sequence_dev): dev_ptrs[t] = device address of
        frame set t; returns one list of Position2D per frame -- and, with timed=True
        (oatgpu_track_sequence_dev_timed), the seconds after the call's entry at which each frame's result was collected."""
        n = len(dev_ptrs)
        arr = (C.c_void_p * n)(*dev_ptrs)
        out = (ffi.Position * (n * self.n_streams))()
        if timed:
            done = (C.c_double * max(n, 1))()
            self._chk(self.lib.oatgpu_track_sequence_dev_timed(self.ctx, arr, n, self.learning_coeff_, out, done))
        else:
            self._chk(self.lib.oatgpu_track_sequence_dev(self.ctx, arr, n, self.learning_coeff_, out))
        res = [[Position2D.from_c(out[t * self.n_streams + s]) for s in range(self.n_streams)] for t in range(n)]
        return (res, [done[t] for t in range(n)]) if timed else res

    def enqueue(self, frames):
        """Pipelined host-frame form (oatgpu_track_enqueue): frames must stay untouched until the
        matching collect(); they are kept referenced here until then."""
        fs = [_frame(f, self.frame_shape) for f in frames]
        ptrs = (ffi._u8p * len(fs))(*[ffi.u8(f) for f in fs])
        self._chk(self.lib.oatgpu_track_enqueue(self.ctx, ptrs, len(fs), self.learning_coeff_))
        self._held = getattr(self, "_held", [])
        self._held.append(fs)

    def enqueue_dev(self, dev_ptr, keepalive=None):
        """Pipelined device-frame form (oatgpu_track_enqueue_dev): dev_ptr -> n_streams*rows*cols*channels bytes.
        By default the per-pixel kernel that reads the frame is queued on the context's stream inside the call.
        After set_fusion(2) it may go out with the NEXT frame's (two frames a launch): the buffer must then stay
        valid and untouched until the frame's result was collected or input_consumed() returned -- pass the owning
        object (a torch tensor) as `keepalive` and it is held here until the matching collect()."""
        self._chk(self.lib.oatgpu_track_enqueue_dev(self.ctx, C.c_void_p(dev_ptr), self.learning_coeff_))
        self._held = getattr(self, "_held", [])
        self._held.append(keepalive)

    def input_consumed(self):
        """Block until every frame handed over so far has been read out of the caller's buffers
        (oatgpu_track_input_consumed): host frames copied, device frames read by their per-pixel kernel."""
        self._chk(self.lib.oatgpu_track_input_consumed(self.ctx))

    def stage(self, stream, frame):
        """oatgpu_track_stage: ONE camera's frame of the next frame set starts its H2D copy now."""
        f = _frame(frame, self.frame_shape)
        self._staged_keep = getattr(self, "_staged_keep", []) + [f]
        self._chk(self.lib.oatgpu_track_stage(self.ctx, int(stream), ffi.u8(f)))

    def enqueue_staged(self):
        """oatgpu_track_enqueue_staged: every stream staged -> the set is registered like enqueue()."""
        self._chk(self.lib.oatgpu_track_enqueue_staged(self.ctx, self.learning_coeff_))
        self._held = getattr(self, "_held", [])
        self._held.append(self._staged_keep)
        self._staged_keep = []

    def set_stage_copy(self, mode):
        """oatgpu_set_stage_copy: 0 = stage() copies by DMA (default), 1 = by a kernel reading the host frame in place."""
        self._chk(self.lib.oatgpu_set_stage_copy(self.ctx, int(mode)))

    def stage_abort(self):
        """oatgpu_track_stage_abort: give up a partly staged set (copies already started are waited for)."""
        self._chk(self.lib.oatgpu_track_stage_abort(self.ctx))
        self._staged_keep = []

    def input_consumed_stream(self, stream):
        """The frame of ONE camera stream of the latest enqueue() has left the caller's buffer
        (oatgpu_track_input_consumed_stream; call for the streams in ascending order)."""
        self._chk(self.lib.oatgpu_track_input_consumed_stream(self.ctx, int(stream)))

    def collect(self):
        self._chk(self.lib.oatgpu_track_collect(self.ctx, self._pos))
        if getattr(self, "_held", None):
            self._held.pop(0)
        return self._out()

    def ready(self):
        """True if collect() would return without blocking (oatgpu_track_ready)."""
        rc = self.lib.oatgpu_track_ready(self.ctx)
        if rc < 0:
            self._chk(rc)
        return rc == 1

    def outstanding(self):
        return self.lib.oatgpu_track_outstanding(self.ctx)

    def set_kalman(self, enable=True, dt=0.02, timeout=0.0, sigma_accel=5.0, sigma_noise=0.0):
        """`posifilt kalman` on the batch (KalmanFilter2D.cpp:63-141; option names and defaults are the
        reference's).  (Re)starts every stream's filter."""
        self._chk(self.lib.oatgpu_set_kalman(self.ctx, int(bool(enable)), dt, timeout, sigma_accel, sigma_noise))

    def set_homography(self, h=None):
        """`posifilt homography` behind the detector / the position filter (HomographyTransform2D.cpp): h = nine numbers,
        row-major [h11, h12, ..., h33]; None turns it off."""
        if h is None:
            self._chk(self.lib.oatgpu_set_homography(self.ctx, 0, None))
        else:
            a = (C.c_double * 9)(*[float(v) for v in np.asarray(h, np.float64).reshape(9)])
            self._chk(self.lib.oatgpu_set_homography(self.ctx, 1, a))

    def set_stream(self, hip_stream):
        self._chk(self.lib.oatgpu_set_stream(self.ctx, C.c_void_p(hip_stream)))

    def get_stream(self):
        """hipStream_t (as an integer) the per-pixel kernel is launched on (oatgpu_get_stream)."""
        return int(self.lib.oatgpu_get_stream(self.ctx) or 0)

    def synchronize(self):
        self._chk(self.lib.oatgpu_synchronize(self.ctx))

    def set_fusion(self, frames_per_launch):
        """Frames per launch of the fused per-pixel kernel on the pipelined path: 1 or 2 (default 2)."""
        self._chk(self.lib.oatgpu_set_fusion(self.ctx, int(frames_per_launch)))

    def set_early_blob(self, on=True):
        """oatgpu_set_early_blob: the blob workgroup of a device-frame step is dispatched ahead of its row scan.
        None: the library's choice by shape (at most three streams, 4 MP a step and more), True / False: forced."""
        self._chk(self.lib.oatgpu_set_early_blob(self.ctx, -1 if on is None else (1 if on else 0)))

    def set_k1_workgroup(self, threads=0):
        """oatgpu_set_k1_workgroup: threads of a workgroup of the per-pixel kernel, 0 = by path (default), 64 or 256."""
        self._chk(self.lib.oatgpu_set_k1_workgroup(self.ctx, int(threads)))

    def last_step_shape(self):
        """(threads of a per-pixel workgroup, blob workgroup dispatched early?) of the latest pipelined step."""
        wg, early = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.oatgpu_last_step_shape(self.ctx, C.byref(wg), C.byref(early)))
        return wg.value, bool(early.value)

    def early_blob_timeouts(self):
        """Frames whose parked blob workgroup gave up waiting for its row scan (oatgpu_early_blob_timeouts)."""
        return int(self.lib.oatgpu_early_blob_timeouts(self.ctx))

    def profile(self, every=1):
        """every = 0/False: off; 1/True: time every step; N: time every Nth step."""
        self._chk(self.lib.oatgpu_profile_enable(self.ctx, int(every)))

    def profile_reset(self):
        self._chk(self.lib.oatgpu_profile_reset(self.ctx))

    def profile_read(self):
        p = ffi.Profile()
        self._chk(self.lib.oatgpu_profile_read(self.ctx, C.byref(p)))
        return dict(steps=p.steps, mog_ms=p.mog_ms, morph_ms=p.morph_ms, blob_ms=p.blob_ms, total_ms=p.total_ms,
                    event_pair_ms=p.event_pair_ms, mog_frames=p.mog_frames, dropped=p.dropped)
