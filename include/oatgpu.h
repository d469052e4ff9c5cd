/*
 * oatgpu.h -- C ABI of liboatgpu.so: the MI355X (gfx950) implementation of Oat's
 * per-frame image hot path
 *
 *      framefilt mog  ->  framefilt col (BGR2HSV)  ->  posidet hsv | thresh
 *
 * Plain pointers and sizes only; no C++ or torch types, no exceptions.  Every
 * entry point names the reference interface it replaces (paths relative to
 * jonnew/Oat).  INTEGRATION.md shows the binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   - return 0 on success, a negative OATGPU_E_* code on failure;
 *     oatgpu_last_error() gives the text.
 *   - a pipelined call (oatgpu_track_enqueue*, _sequence_dev, _batch*) whose kernel launches fail half-way is FATAL
 *     for its context: the frame counts, learning-rate schedule and possibly the model have moved ahead of the
 *     results, so every later pipelined call returns OATGPU_E_HIP ("context unusable ...") instead of silently
 *     losing parity with the reference; results already outstanding can be collected, then destroy the context
 *     (the reference's component would have thrown out of process() and exited, framefilter/main.cpp:278-295).
 *   - the caller owns every host buffer; the context owns all device state
 *     (MOG2 model planes, bit masks, label tables, result ring).
 *   - one context per host thread (not re-entrant).  The HIP streams the work is
 *     issued on belong to the device and are shared by all contexts of a process
 *     (DESIGN.md section 4); oatgpu_set_stream makes a context issue its per-pixel
 *     kernel and copies on an external HIP stream instead.
 *   - pixel buffers are packed, rows*cols*channels bytes, no row padding
 *     (lib/datatypes/Frame.h:92-103, lib/shmemdf/Sink.h:289-290).
 *   - "stream" below = one camera stream (an independent Oat pipeline), not a
 *     HIP stream, unless it says HIP.
 */
#ifndef OATGPU_H
#define OATGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OATGPU_ABI_VERSION 9     /* 2: oatgpu_position grew (filter outputs), new entry points; 3: oatgpu_config.mog_restore_nmodes; 4: oatgpu_cvt_color, oatgpu_set_fusion, oatgpu_set_homography, oatgpu_profile.mog_frames; 5: oatgpu_track_sequence_dev_timed, oatgpu_track_enqueue_dev pairs frames only after oatgpu_set_fusion(2); 6: oatgpu_track_input_consumed_stream, oatgpu_track_stage, oatgpu_track_enqueue_staged; 7: oatgpu_track_stage_abort, oatgpu_set_early_blob, oatgpu_set_stage_copy, oatgpu_set_deferred / _fetch_frame / _fetch_position, a failed pipelined launch is fatal for its context; 8: oatgpu_track_sequence_dev_latency, oatgpu_device_open_retries, oatgpu_set_k1_workgroup, oatgpu_last_step_shape, oatgpu_early_blob_timeouts, a parked blob workgroup that times out switches early dispatch off for its context; 9: oatgpu_profile.dropped, every scratch set is allocated by oatgpu_create (an out-of-memory is reported there, never in the middle of a step) */

enum {
    OATGPU_OK = 0,
    OATGPU_E_INVALID = -1,    /* bad argument / configuration               */
    OATGPU_E_HIP = -2,        /* HIP runtime error (text in last_error)     */
    OATGPU_E_NOMEM = -3,
    OATGPU_E_RING_FULL = -4,  /* track_enqueue without a free result slot   */
    OATGPU_E_RING_EMPTY = -5, /* track_collect with nothing outstanding     */
    OATGPU_E_NODEVICE = -6    /* no usable HIP device                       */
};

/* Context configuration.  Fill with oatgpu_default_config() first; it sets the
 * reference's defaults:
 *   MOG2   cv::createBackgroundSubtractorMOG2() with all defaults
 *          (src/framefilter/BackgroundSubtractorMOG.cpp:82-83)
 *   hsv    HSVDetector.h:77-94 / HSVDetector.cpp:34-47 (all-pass thresholds,
 *          erode off, dilate 10, area [0, DBL_MAX))
 */
typedef struct oatgpu_config {
    int32_t device;            /* HIP device ordinal                          */
    int32_t n_streams;         /* camera streams batched in this context      */
    int32_t rows, cols;        /* frame geometry, identical for all streams   */
    int32_t ring_depth;        /* outstanding track_enqueue results (>=1)     */
    int32_t channels;          /* 3: BGR frames (mog -> col HSV -> posidet hsv)
                                  1: GREY frames (mog -> posidet thresh; -T = h_lo/h_hi) */

    /* --- MOG2 (BackgroundSubtractorMOG.cpp:82-83) --- */
    int32_t history;           /* 500 */
    int32_t nmixtures;         /* 5 (1..5 supported)                          */
    float var_threshold;       /* Tb 16   */
    float background_ratio;    /* TB 0.9  */
    float var_threshold_gen;   /* Tg 9    */
    float var_init;            /* 15      */
    float var_min;             /* 4       */
    float var_max;             /* 75      */
    float ct;                  /* 0.05; must lie in [0, 0.5): OpenCV accepts any value, the kernel relies on a matched
                                  mode (weight >= alpha (1 - ct)) never being prunable (weight < alpha ct) */
    float tau;                 /* 0.5     */
    int32_t detect_shadows;    /* 1       */
    int32_t shadow_value;      /* 127     */

    /* --- detector (HSVDetector.cpp:77-140, SimpleThreshold.cpp:71-112) --- */
    int32_t h_lo, h_hi;        /* -H [min,max] in [0,256]; also -T for thresh */
    int32_t s_lo, s_hi;        /* -S */
    int32_t v_lo, v_hi;        /* -V */
    int32_t erode;             /* -e, 0 = off                                 */
    int32_t dilate;            /* -d, 0 = off                                 */
    double min_area, max_area; /* -a [min,max)                                */

    /* --- posidet diff (DifferenceDetector.h:63-76) --- */
    int32_t diff_threshold;    /* -d, default 10                              */
    int32_t blur;              /* -b, default 2; 0 = off; <= 22 supported     */

    /* --- MOG2 mode count (MOG2Invoker's `nmodes = nNewModes;`) --- */
    int32_t mog_restore_nmodes; /* 1 (default): the per-pixel mode count is set back to its value at
                                  entry after the weight renormalisation, as cv::BackgroundSubtractorMOG2
                                  does -- a pruned mode keeps its slot with weight 0 and modesUsed never
                                  shrinks; 0: a pruned mode leaves the count (round 1's reading).
                                  oracle/mog2.c "Mode count" has the derivation; no OpenCV is available
                                  here to settle it, hence a switch rather than a constant. */
    int32_t reserved_;
} oatgpu_config;

/* What posidet writes into oat::Position2D (src/positiondetector/DetectorFunc.cpp:46,58-60)
 * plus the raw integer Green sums the centroid is derived from. */
typedef struct oatgpu_position {
    int32_t valid;             /* Position2D::position_valid                  */
    int32_t first_pixel;       /* raster index of the blob's first pixel, -1  */
    double x, y;               /* Position2D::position (pixels)               */
    double area;               /* siftContours' area out-parameter            */
    int64_t a00, a10, a01;     /* exact contour sums (cv::moments internals)  */
    /* With oatgpu_set_kalman on (fused track calls only), valid/x/y above are what
     * `posifilt kalman` would hand downstream (Position2D::position_valid/position) and: */
    int32_t velocity_valid;    /* Position2D::velocity_valid (0 when the filter is off) */
    int32_t raw_valid;         /* the detector's own position_valid              */
    double vx, vy;             /* Position2D::velocity (position units / s)      */
    double raw_x, raw_y;       /* the detector's own centroid                    */
} oatgpu_position;

/* Per-stage device time accumulated while profiling is enabled (HIP events on
 * the context's HIP stream). */
typedef struct oatgpu_profile {
    int64_t steps;             /* track steps measured                        */
    double mog_ms;             /* fused MOG2+mask+HSV+inRange kernel          */
    double morph_ms;           /* erode + dilate                              */
    double blob_ms;            /* labelling + contour sums + selection        */
    double total_ms;           /* first event to last event of each step      */
    double event_pair_ms;      /* calibration: elapsed time of an event pair around an EMPTY
                                  kernel on the same HIP stream (what mog_ms contains per step
                                  besides kernel execution); measured at profile_enable */
    int64_t mog_frames;        /* frames the `steps` measured launches of the fused kernel covered
                                  (steps .. 2 * steps, see oatgpu_set_fusion)              */
    int64_t dropped;           /* samples left out: their per-pixel launch read more than 8 x the running average and 1 ms above it -- a host
                                  thread descheduled between the event record and the launch call puts its absence into the
                                  pair; at most 4 in a row (a 5th is a change of regime and is taken)  */
} oatgpu_profile;

typedef struct oatgpu_ctx oatgpu_ctx;

int oatgpu_abi_version(void);
int oatgpu_default_config(oatgpu_config *cfg);

/* Object lifetime == the reference component's (MOG model and scratch are
 * members: BackgroundSubtractorMOG.h:72-73, HSVDetector.h:83).
 * Device memory, all of it allocated HERE (an out-of-memory is OATGPU_E_NOMEM from oatgpu_create, never a failure in the middle of a
 * pipelined step): per camera stream and pixel 101 B of model + 150 B of back-half scratch (five sets of 30 B: three launch orders'
 * worth plus the repair set) + ring_depth / 8 B of threshold words + the staging frame; 1080p: 0.52 GB a stream, 4K: 2.1 GB.  The
 * host-frame path (oatgpu_track_enqueue / _stage) adds ring_depth staging frames on first use. */
oatgpu_ctx *oatgpu_create(const oatgpu_config *cfg);   /* NULL on failure; oatgpu_last_error(NULL) */
void oatgpu_destroy(oatgpu_ctx *ctx);
const char *oatgpu_last_error(const oatgpu_ctx *ctx);

/* Host-memory helpers for callers that do not link HIP themselves.  Frames handed to the
 * stage-by-stage calls may live anywhere; when they live in page-locked memory (a shared-memory
 * segment registered with oatgpu_host_register, or a buffer from oatgpu_host_alloc) the copies
 * are direct DMA instead of bounce-buffered. */
/* Devices and where they hang: oatgpu_device_count() HIP devices are visible; oatgpu_device_numa_node(i) is the NUMA node
 * of device i's PCIe slot (from its bus id through sysfs), -1 if unknown.  A multi-device component keeps the thread that
 * drives device i on that node's CPUs (host/oat_track_hip.cpp --gpu-index D0,D1,..): its launches, its shared-memory reads
 * and the staging copies then stay on the socket the GPU is attached to.  No reference counterpart. */
int oatgpu_device_count(void);
/* oatgpu_create opens its device with a bounded retry (several processes opening one freshly booted device at the same
 * instant can see the first runtime calls fail transiently); this is how many retries this process has needed so far. */
int oatgpu_device_open_retries(void);
int oatgpu_device_numa_node(int32_t device);
int oatgpu_host_register(void *ptr, size_t bytes);
int oatgpu_host_unregister(void *ptr);
void *oatgpu_host_alloc(size_t bytes);
void oatgpu_host_free(void *ptr);

/* HIP stream plumbing (hipStream_t passed as void*). */
int oatgpu_set_stream(oatgpu_ctx *ctx, void *hip_stream);
void *oatgpu_get_stream(oatgpu_ctx *ctx);
int oatgpu_synchronize(oatgpu_ctx *ctx);

/* Frames per launch of the fused per-pixel kernel on the pipelined path, 1 or 2.  The MOG2 update is a recurrence
 * per pixel, so two consecutive frames of a stream can be taken on ONE pass over its model (kept in registers
 * between them).  With 2, an enqueue only REGISTERS its frame; the kernels are launched when the next frame is
 * enqueued -- or as soon as the frame's result is asked for (oatgpu_track_collect / oatgpu_track_ready reaching
 * that frame, oatgpu_track_input_consumed) or any synchronous entry point runs, then for the one frame alone; a
 * caller that collects every frame before it enqueues the next (oatgpu_track_batch*, a camera-bound component
 * loop) never waits for a second frame.  Results, their order, the threshold images and the model are
 * bit-identical either way (FrameFilter.cpp:59-98 / PositionDetector.cpp:58-99: one token out per token in, in
 * order).
 * DEFAULT (no call): two frames a launch wherever the LIBRARY owns the frame's lifetime -- oatgpu_track_enqueue
 * (host frames: copied to a staging slot inside the call) and oatgpu_track_sequence_dev (all frames handed over at
 * once) -- and ONE frame a launch for oatgpu_track_enqueue_dev, whose kernel is then queued inside the call as it
 * always was, so a caller that reuses its device buffer in stream order stays correct.  oatgpu_set_fusion(2) opts
 * oatgpu_track_enqueue_dev in: from then on a frame handed to it must stay valid and UNTOUCHED until its result
 * was collected or oatgpu_track_input_consumed returned.  oatgpu_set_fusion(1) switches pairing off everywhere.
 * While a traffic audit is on (oatgpu_traffic_audit) GREY contexts launch one frame at a time. */
int oatgpu_set_fusion(oatgpu_ctx *ctx, int32_t frames_per_launch);

/* Early dispatch of the blob-analysis workgroup.  On the pipelined device-frame path of steps of 4 MP and more, the big
 * workgroups that label a step's masks (findContours + moments, DetectorFunc.cpp:41-63) are submitted on a HIP stream of
 * their own together with the step's other kernels and WAIT ON THE DEVICE for their frames' row scans: they take their
 * wave slots while the per-pixel kernel of an earlier frame drains instead of queueing for them on the frame's critical path.
 * on = -1 (the default): by shape -- contexts of at most three streams, whose per-pixel kernel then runs with one wave a
 * workgroup (4K: 18.5 k -> 19.2 k fps; 2 x 1080p: 64.8 k -> 70.3 k; 3 x: 69.3 k -> 72-74 k; four and more lose); 0: never; 1: every eligible step (several streams: row scan + blob analysis beside the
 * per-pixel kernel 16 x 1080p 427 -> 69 us for 1-3 % of the frame rate).  Results are identical either way.
 * Switch it off (0) under tools that serialise kernel dispatches -- a counter-collecting profiler (rocprofv3 --pmc) would
 * let the waiting workgroup run before the row scan it waits for; the kernel then gives up after 100 ms and the frame is
 * redone by the global kernels: correct, but slow. */
int oatgpu_set_early_blob(oatgpu_ctx *ctx, int32_t on);
/* 1 once a parked blob workgroup of this context has given up after 100 ms because its row scan was not dispatched beside it
 * (a tool that serialises kernel dispatches), else 0.  It switches early dispatch off for the context (oatgpu_last_error
 * says so) and raises a flag on the device at which the workgroups parked behind it decline without waiting; all those
 * frames are redone by the global kernels: results are unaffected, and the episode costs its caller 100 ms once. */
int64_t oatgpu_early_blob_timeouts(const oatgpu_ctx *ctx);
/* Threads of a workgroup of the fused per-pixel kernel on the pipelined path: 0 (default) by path -- 64 (one wave a
 * workgroup) where the step's blob workgroup is dispatched early, where a LONE frame (nothing else outstanding) of a
 * context that would use the early order is launched, or the model is dense; 256 otherwise -- or 64 / 256
 * whatever the path.  Results are identical.  For profiling the shipped instantiation under a tool that needs
 * oatgpu_set_early_blob(0) (bench.py's counter passes). */
int oatgpu_set_k1_workgroup(oatgpu_ctx *ctx, int32_t threads);
/* How the latest pipelined step was launched: threads of a per-pixel workgroup (0 before the first step) and whether its
 * blob workgroup was dispatched early.  Either pointer may be NULL. */
int oatgpu_last_step_shape(const oatgpu_ctx *ctx, int32_t *k1_workgroup, int32_t *early_blob);

/* Re-configure the detector between frames (what the reference's tuning GUI
 * mutates: HSVDetector.cpp:175-251). */
int oatgpu_set_detector(oatgpu_ctx *ctx, int32_t h_lo, int32_t h_hi, int32_t s_lo, int32_t s_hi,
                        int32_t v_lo, int32_t v_hi, int32_t erode, int32_t dilate,
                        double min_area, double max_area);

/* `posifilt kalman` (src/positionfilter/KalmanFilter2D.cpp:63-141) applied to every stream's
 * detections inside the fused track calls (oatgpu_track_batch/_dev/_enqueue_dev), on the device,
 * in frame order.  dt: --dt (s, default 0.02); timeout: --timeout (s, default 0 -- with which the
 * reference's filter never tracks); sigma_accel: --sigma-accel (default 5); sigma_noise:
 * --sigma-noise (default 0).  All must be >= 0 and dt > 0 (TOMLSanitize lower bound 0).  Calling it
 * (re)starts every stream's filter from the reference's initial state; enable = 0 turns it off.
 * The single-stage oatgpu_detect_* calls are never filtered. */
int oatgpu_set_kalman(oatgpu_ctx *ctx, int32_t enable, double dt, double timeout, double sigma_accel,
                      double sigma_noise);

/* `posifilt homography` on the batch (src/positionfilter/HomographyTransform2D.cpp:62-107) behind the detector and,
 * when it is on, the position filter: every result of the fused track calls goes through cv::perspectiveTransform with
 * the row-major 3x3 matrix h9 (--homography [h11,h12,...,h33], :44-58) -- position where valid, velocity where valid
 * with the matrix' offsets zeroed (:79-89).  raw_x / raw_y keep the detector's pixels.  The caller marks the
 * Position2D it publishes as WORLD units with this matrix (Position2D::setCoordSystem, :102).  enable = 0: off. */
int oatgpu_set_homography(oatgpu_ctx *ctx, int32_t enable, const double *h9);

/* `framefilt mask` fused in front of mog (src/framefilter/FrameMasker.cpp:71-75:
 * frame.setTo(0, roi_mask == 0)): roi_mask is rows*cols bytes, nonzero = keep; NULL removes the
 * mask of that stream.  Applies to oatgpu_mog_apply / _filter and the fused track calls. */
int oatgpu_set_roi_mask(oatgpu_ctx *ctx, int32_t stream_ix, const uint8_t *roi_mask);

/* BackgroundSubtractor::filter (`framefilt bsub`, src/framefilter/BackgroundSubtractor.cpp:87-100):
 * the first frame of a camera stream becomes its background; alpha > 0 adapts it
 * (cv::accumulateWeighted); out = in - background, saturating.  rows*cols*channels bytes; out may
 * equal in. */
int oatgpu_bsub_filter(oatgpu_ctx *ctx, int32_t stream_ix, const uint8_t *frame_in, uint8_t *frame_out,
                       double alpha);

/* `framefilt bsub -f FILE` (BackgroundSubtractor.cpp:63-71): the background image of stream s comes from
 * the caller (rows*cols*channels bytes) instead of the first frame.  As in the reference it cannot adapt
 * afterwards (its fp32 accumulator is never made): oatgpu_bsub_filter with alpha > 0 then fails. */
int oatgpu_bsub_set_background(oatgpu_ctx *ctx, int32_t stream, const uint8_t *image);

/* FrameMasker::filter (FrameMasker.cpp:71-75) as a stage of its own: out = in where the ROI of stream s
 * (oatgpu_set_roi_mask) is non-zero, 0 elsewhere; without a ROI the frame passes unchanged.  In place allowed. */
int oatgpu_mask_filter(oatgpu_ctx *ctx, int32_t stream, const uint8_t *in, uint8_t *out);

/* Threshold::filter (`framefilt thresh`, src/framefilter/Threshold.cpp:67-81): grey conversion for BGR
 * contexts, inRange [i_min, i_max] (0..256), pixels outside are zeroed.  out may equal in. */
int oatgpu_thresh_filter(oatgpu_ctx *ctx, const uint8_t *frame_in, uint8_t *frame_out, int32_t i_min,
                         int32_t i_max);

/* Deferred completion of the stage-by-stage operators below (default off).  With on = 1 a frame filter (oatgpu_mog_filter,
 * _bsub_filter, _mask_filter, _thresh_filter, _bgr2hsv, _cvt_color) or a detector (oatgpu_detect_hsv / _thresh / _diff)
 * returns as soon as its INPUT frame has been read -- the point where the reference posts its SOURCE, right after its
 * memcpy (FrameFilter.cpp:73-80, PositionDetector.cpp:78-86) -- leaves its output argument untouched (it may be NULL for the
 * detectors) and keeps the result on the device; oatgpu_fetch_frame / oatgpu_fetch_position then deliver it: a component
 * posts its SOURCE, waits for its SINK and has the filtered frame copied STRAIGHT into the sink's (page-locked) shared-memory
 * frame, without the staging copy in between (host/component.hpp).  One result may be waiting at a time: the next
 * stage-by-stage call before the fetch fails with OATGPU_E_INVALID.  Results are those of the plain calls. */
int oatgpu_set_deferred(oatgpu_ctx *ctx, int32_t on);
int oatgpu_fetch_frame(oatgpu_ctx *ctx, uint8_t *frame_out);
int oatgpu_fetch_position(oatgpu_ctx *ctx, oatgpu_position *out);

/* ---- stage-by-stage operators (host buffers), one call == one reference call ---- */

/* cv::BackgroundSubtractorMOG2::apply(frame, mask, learning_rate)
 * (BackgroundSubtractorMOG.cpp:124).  fgmask_out: rows*cols, values {0,127,255}. */
int oatgpu_mog_apply(oatgpu_ctx *ctx, int32_t stream_ix, const uint8_t *bgr_in,
                     uint8_t *fgmask_out, double learning_rate);

/* BackgroundSubtractorMOG::filter(cv::Mat&) CPU-branch semantics
 * (BackgroundSubtractorMOG.cpp:114-127): apply + frame.setTo(0, mask == 0).
 * bgr_out may equal bgr_in. */
int oatgpu_mog_filter(oatgpu_ctx *ctx, int32_t stream_ix, const uint8_t *bgr_in,
                      uint8_t *bgr_out, double learning_rate);

/* ColorConvert::filter with COLOR_BGR2HSV (ColorConvert.cpp:101-107). */
int oatgpu_bgr2hsv(oatgpu_ctx *ctx, const uint8_t *bgr_in, uint8_t *hsv_out);

/* ColorConvert::filter for any pair of colours (ColorConvert.cpp:101-107): from_color / to_color are
 * oat::PixelColor values (BINARY 0, GREY 1, BGR 2, HSV 3; Color.h:29-34), the cvtColor code comes from
 * oat::color_conv_table (Color.h:45-51): BGR -> GREY|BINARY (COLOR_BGR2GRAY -- the bridge `posidet thresh`
 * and `posidet diff` need, SimpleThreshold.cpp:46, DifferenceDetector.cpp:44), GREY|BINARY -> BGR, BGR -> HSV,
 * HSV -> BGR.  in: rows*cols*(1|3) bytes, out: rows*cols*(1|3) bytes, must not overlap unless equal in size.
 * OATGPU_E_INVALID with the reference's texts for pairs with nothing to do (ColorConvert.cpp:79-85) or
 * not possible (Color.h:88-95). */
int oatgpu_cvt_color(oatgpu_ctx *ctx, int32_t from_color, int32_t to_color, const uint8_t *in, uint8_t *out);

/* HSVDetector::detectPosition (HSVDetector.cpp:142-173): hsv_in rows*cols*3. */
int oatgpu_detect_hsv(oatgpu_ctx *ctx, int32_t stream_ix, const uint8_t *hsv_in,
                      oatgpu_position *out);

/* SimpleThreshold::detectPosition (SimpleThreshold.cpp:114-134): grey_in rows*cols;
 * uses h_lo/h_hi as -T [min,max]. */
int oatgpu_detect_thresh(oatgpu_ctx *ctx, int32_t stream_ix, const uint8_t *grey_in,
                         oatgpu_position *out);

/* DifferenceDetector::detectPosition (DifferenceDetector.cpp:98-173): grey_in rows*cols.  Stateful
 * per camera stream (the previous frame); the first frame of a stream is analysed as is, like the
 * reference.  For blur <= 22 a box blur of a {0,255} image is non-zero exactly where the k x k
 * dilation is (away from the outermost ring, which findContours zeroes), so the blur runs as a
 * bit-mask dilation. */
int oatgpu_detect_diff(oatgpu_ctx *ctx, int32_t stream_ix, const uint8_t *grey_in, oatgpu_position *out);

/* ---- fused hot path: mog + setTo + BGR2HSV + inRange + erode + dilate + blob ---- */

/* One frame for EVERY stream of the context (the whole chain
 * FrameFilter::process -> ColorConvert -> PositionDetector::process,
 * FrameFilter.cpp:59-98, PositionDetector.cpp:58-99, minus the shm hand-offs).
 * frames_host[i] -> rows*cols*channels bytes of stream i.  out[n_streams].
 * With channels == 1 the chain is framefilt mog (GREY) -> posidet thresh. */
int oatgpu_track_batch(oatgpu_ctx *ctx, const uint8_t *const *frames_host, int32_t n,
                       double learning_rate, oatgpu_position *out);

/* Same with the frames already resident in device memory:
 * frames_dev = n_streams*rows*cols*channels bytes, stream-major. */
int oatgpu_track_batch_dev(oatgpu_ctx *ctx, const void *frames_dev, double learning_rate,
                           oatgpu_position *out);

/* Pipelined form: enqueue returns at once; collect returns results in enqueue
 * order (exactly one result set per enqueued frame set, SURVEY.md 8b token
 * discipline).  Up to ring_depth enqueues may be outstanding.  By default the per-pixel
 * kernel that reads frames_dev is queued on the context's stream inside this call; after
 * oatgpu_set_fusion(2) it may go out with the NEXT frame's instead, and frames_dev must stay
 * valid and untouched until the frame's result was collected or oatgpu_track_input_consumed
 * returned (see oatgpu_set_fusion). */
int oatgpu_track_enqueue_dev(oatgpu_ctx *ctx, const void *frames_dev, double learning_rate);
/* Pipelined form for frames in HOST memory (what a camera or a shared-memory SOURCE hands over):
 * the frames are copied to a per-slot device buffer on a copy stream of their own, so the copy of
 * frame t+1 overlaps the kernels of frame t.  With page-locked frames (oatgpu_host_alloc /
 * oatgpu_host_register) the copies are direct DMA; the caller must leave frames_host[i] untouched
 * until the matching oatgpu_track_collect returns. */
int oatgpu_track_enqueue(oatgpu_ctx *ctx, const uint8_t *const *frames_host, int32_t n, double learning_rate);
int oatgpu_track_collect(oatgpu_ctx *ctx, oatgpu_position *out);
/* oatgpu_track_enqueue camera by camera, for components whose cameras do not deliver at the same instant:
 * oatgpu_track_stage starts the H2D copy of ONE camera's frame of the NEXT frame set as soon as that camera has
 * one (PositionDetector.cpp:63-75 waits for its one source; an N-camera component need not hold the link idle
 * until the slowest of N has delivered); once every stream 0..n_streams-1 has been staged,
 * oatgpu_track_enqueue_staged registers the set exactly as oatgpu_track_enqueue would have.  While a set is being
 * staged oatgpu_track_input_consumed_stream(i) waits for stream i's own copy (any staged stream, any order).
 * OATGPU_E_RING_FULL from the first oatgpu_track_stage of a set when ring_depth sets are outstanding. */
int oatgpu_track_stage(oatgpu_ctx *ctx, int32_t stream_ix, const uint8_t *frame_host);
int oatgpu_track_enqueue_staged(oatgpu_ctx *ctx, double learning_rate);
/* How oatgpu_track_stage moves a camera's frame: 0 (default) a DMA copy (hipMemcpyAsync); 1 a small copy KERNEL that reads
 * the frame in place over PCIe -- for frames in page-locked, mapped host memory (oatgpu_host_register'ed shared-memory
 * segments, oatgpu_host_alloc) that are 16-byte aligned; anything else takes the DMA path.  A launch costs the host a
 * quarter of a DMA copy's set-up, which is what an N-camera round is short of. */
int oatgpu_set_stage_copy(oatgpu_ctx *ctx, int32_t mode);
/* Gives up a partly staged frame set: a camera ended in the middle of a round (its SINK went END, lib/shmemdf/Source.h:187-215)
 * or a call failed after the first oatgpu_track_stage.  Waits for the copies already started (their SOURCEs may be posted
 * afterwards), registers nothing, owes no result; the next oatgpu_track_stage starts a new set.  No-op without one. */
int oatgpu_track_stage_abort(oatgpu_ctx *ctx);

/* oatgpu_track_input_consumed blocks until every frame handed over so far has been read out of the caller's
 * buffers -- host frames (oatgpu_track_enqueue): their H2D copies are done; device frames
 * (oatgpu_track_enqueue_dev): the per-pixel kernel that reads them has finished (a frame that was only
 * registered, oatgpu_set_fusion(2), is launched first).  From then on the caller may release or overwrite
 * them, i.e. post() the shared-memory SOURCEs while the device is still computing (the reference releases
 * its source right after its memcpy, FrameFilter.cpp:73-80 / PositionDetector.cpp:78-86).
 * oatgpu_track_ready: 1 if oatgpu_track_collect would return without blocking, 0 if the oldest outstanding
 * result is still being computed (or nothing is outstanding), < 0 on error.  (If that oldest frame is still only
 * registered -- oatgpu_set_fusion -- the call launches it; if its speculative back half declined the frame, the
 * call launches the global kernels on it and reports 0 until they are done.) */
int oatgpu_track_input_consumed(oatgpu_ctx *ctx);
/* The same for ONE camera stream of the latest oatgpu_track_enqueue: returns when frames_host[stream_ix] has been
 * read (its H2D copy is done), so an N-camera component posts every SOURCE as its own frame leaves shared memory
 * instead of after all N copies -- the frames travel one after the other over one PCIe link, and the upstream
 * writers of the first cameras refill their segments while the later copies are still running (PositionDetector.cpp:78-86
 * releases its one source right after its one memcpy; this is that rule per camera).  Call it for the streams in
 * ascending order.  Per-stream events are recorded from the first call on (that first call waits for the whole
 * set); for device frames it is oatgpu_track_input_consumed. */
int oatgpu_track_input_consumed_stream(oatgpu_ctx *ctx, int32_t stream_ix);
int oatgpu_track_ready(oatgpu_ctx *ctx);

/* A whole recorded sequence through the pipelined path in one call (what `oat frameserve file`
 * feeding the chain amounts to, without a host round trip per frame): frames_dev[t] = frame set t
 * in device memory (n_streams*rows*cols*channels bytes, stream-major), out[t*n_streams + s] = the
 * result of stream s for frame t.  Equivalent to the enqueue/collect loop with the ring kept full;
 * nothing may be outstanding when it is called. */
int oatgpu_track_sequence_dev(oatgpu_ctx *ctx, const void *const *frames_dev, int32_t n_frames,
                              double learning_rate, oatgpu_position *out);
/* The same, also reporting WHEN each frame's result was collected: done_s[t] = seconds since the call was
 * entered (steady clock) at which frame t's result set had been handed out.  With the ring kept full the
 * difference done_s[t+K] - done_s[t] is the steady-state time of K steps, free of the pipeline's fill and drain
 * (bench.py's block timing).  done_s may be NULL. */
int oatgpu_track_sequence_dev_timed(oatgpu_ctx *ctx, const void *const *frames_dev, int32_t n_frames,
                                    double learning_rate, oatgpu_position *out, double *done_s);
/* ... and WHEN each frame was handed to oatgpu_track_enqueue_dev inside the call: enq_s[t], same clock; done_s[t] -
 * enq_s[t] is the time a frame spends in the pipeline with the ring kept full (bench.py latency_us).  Either may be NULL. */
int oatgpu_track_sequence_dev_latency(oatgpu_ctx *ctx, const void *const *frames_dev, int32_t n_frames,
                                      double learning_rate, oatgpu_position *out, double *done_s, double *enq_s);
int oatgpu_track_outstanding(const oatgpu_ctx *ctx);

/* ---- parity taps / model checkpoint (not in the reference; for tests and resume) ---- */

enum {
    OATGPU_TAP_THRESHOLD = 0,  /* inRange output (after the fused kernel)        */
    OATGPU_TAP_MORPH = 1,      /* after erode/dilate == reference threshold_frame_ */
    OATGPU_TAP_FINAL = 2       /* after the 1-px frame zeroing findContours does */
};
/* Unpacks the device bit mask of the last processed frame of a stream to
 * rows*cols bytes {0,255}. */
int oatgpu_read_mask(oatgpu_ctx *ctx, int32_t stream_ix, int32_t which, uint8_t *out);

/* MOG2 model of one stream in the oracle's (OpenCV's) logical layout:
 * modes_used[rows*cols], weight/variance[rows*cols*nmix], mean[rows*cols*nmix*channels].
 * Entries of unused modes (>= modes_used) are unspecified on get.
 * set_state / load take ANY model.  One that is not what a run of this library leaves -- a weight that is neither 0 nor in
 * [2^-62, 4], a mean that is not finite, at least 2^20 in size or -0.f, a variance outside [var_min, var_max] -- is advanced
 * by the kernel instantiations that keep the compiler's IEEE division and compute every update (exact for any input,
 * several times slower) until the stream is re-initialised or a plain model is imported; every other model by the product
 * instantiations. */
int oatgpu_mog_get_state(oatgpu_ctx *ctx, int32_t stream_ix, uint8_t *modes_used, float *weight,
                         float *variance, float *mean, int32_t *nframes);
int oatgpu_mog_set_state(oatgpu_ctx *ctx, int32_t stream_ix, const uint8_t *modes_used,
                         const float *weight, const float *variance, const float *mean,
                         int32_t nframes);

/* MOG2 model checkpoint of one stream (the reference has none: a restarted `framefilt mog`
 * relearns its background for ~history frames).  save writes PATH atomically (PATH.tmp + rename);
 * load refuses a file whose geometry, channel count or mixture count differ from the context's.
 * Resuming from a checkpoint continues bit-identically to an uninterrupted run. */
int oatgpu_mog_save(oatgpu_ctx *ctx, int32_t stream_ix, const char *path);
int oatgpu_mog_load(oatgpu_ctx *ctx, int32_t stream_ix, const char *path);

/* ---- measurement ---- */
/* on = 0: off; on = 1: time every step; on = N > 1: time every Nth step (each timed step costs
 * five hipEventRecord calls, ~20 us of host time -- sample when the host is the bottleneck). */
int oatgpu_profile_enable(oatgpu_ctx *ctx, int32_t on);
int oatgpu_profile_read(oatgpu_ctx *ctx, oatgpu_profile *out);   /* synchronises */
int oatgpu_profile_reset(oatgpu_ctx *ctx);

/* Traffic audit of the fused per-pixel kernel (measurement; bench.py's `useful_bytes_per_px`).  While on,
 * the kernel runs in an instantiation that also counts, for every load and store it executes, the bytes its
 * lanes ask for and the 32-byte sectors / 64-byte half lines those requests touch.  Results are unchanged
 * (same predicates, same arithmetic); the audited launches are slower and must not be timed. */
typedef struct oatgpu_traffic {
    int64_t launches;                 /* audited kernel launches                          */
    int64_t pixels;                   /* pixels processed by them                         */
    int64_t lane_bytes_read;          /* bytes requested by the lanes: what the arithmetic */
    int64_t lane_bytes_written;       /*   can depend on / has changed ("useful" bytes)    */
    int64_t sector32_bytes_read;      /* the same requests in whole 32-byte sectors        */
    int64_t sector32_bytes_written;
    int64_t sector64_bytes_read;      /* ... and in whole 64-byte half lines               */
    int64_t sector64_bytes_written;
} oatgpu_traffic;
int oatgpu_traffic_audit(oatgpu_ctx *ctx, int32_t on);             /* on != 0: zero the counters and start */
int oatgpu_traffic_read(oatgpu_ctx *ctx, oatgpu_traffic *out);     /* synchronises */

/* Achievable HBM rates of THIS device, measured with plain streaming kernels (16 B/lane, `bytes`
 * per buffer, best of `reps`): *read_gbps for a read-only sum, *copy_gbps for read+write of a copy
 * (bytes moved = 2*bytes).  The spec peak (8 TB/s on MI355X) is never reached by any kernel; these
 * are the ceilings the hot path is compared against besides the spec. */
int oatgpu_measure_hbm(oatgpu_ctx *ctx, size_t bytes, int32_t reps, double *read_gbps, double *copy_gbps);

#ifdef __cplusplus
}
#endif
#endif /* OATGPU_H */
