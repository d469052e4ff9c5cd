/*
 * oat_oracle.h -- CPU ORACLE for the Oat hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the arithmetic that jonnew/Oat's
 *   framefilt mog  ->  framefilt col (BGR2HSV)  ->  posidet hsv|thresh
 * chain performs on the CPU.  Oat itself is 3-15 lines of glue per stage; the
 * arithmetic lives in OpenCV 3.x (pinned by the reference's README.md:1613 to
 * 3.1.0), which is NOT vendored under /root/reference and is not installed in
 * this image.  Every function below therefore cites (a) the reference call
 * site it stands in for and (b) the OpenCV 3.1.0 routine whose published
 * algorithm it restates.
 *
 * PARITY UNPINNED: the reference holds no golden vectors / known-answer tests
 * for this path (its tests cover lib/shmemdf only) and
 * neither the reference nor OpenCV can be built here.  The oracle is pinned
 * only by hand-derived known answers (tests/golden/) and by cross-checking
 * two independent formulations of the contour stage.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this library.  The product path (oat_amd/) never does.
 */
#ifndef OAT_ORACLE_H
#define OAT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ MOG2 -- */

/* cv::createBackgroundSubtractorMOG2() defaults, as used (all defaults) by
 * BackgroundSubtractorMOG.cpp:82-83. */
typedef struct {
    int history;          /* 500  */
    int nmixtures;        /* 5    */
    float var_threshold;  /* Tb = 16   */
    float background_ratio; /* TB = 0.9 */
    float var_threshold_gen; /* Tg = 9 */
    float var_init;       /* 15 */
    float var_min;        /* 4  */
    float var_max;        /* 75 = 5*var_init */
    float ct;             /* 0.05 complexity reduction */
    int detect_shadows;   /* 1 */
    uint8_t shadow_value; /* 127 */
    float tau;            /* 0.5 */
    int restore_nmodes;   /* 1: `nmodes = nNewModes;` after the renormalisation (see mog2.c, "mode count");
                             0: a pruned mode leaves modesUsed */
} oat_mog2_params;

void oat_mog2_default_params(oat_mog2_params *p);

typedef struct oat_mog2 oat_mog2;

/* channels: 3 (BGR) or 1 (GREY) */
oat_mog2 *oat_mog2_create(int rows, int cols, int channels, const oat_mog2_params *p);
void oat_mog2_destroy(oat_mog2 *m);

/* cv::BackgroundSubtractorMOG2::apply(image, fgmask, learningRate)
 * (BackgroundSubtractorMOG.cpp:124).  mask in {0,127,255}. */
void oat_mog2_apply(oat_mog2 *m, const uint8_t *image, uint8_t *mask, double learning_rate);

/* BackgroundSubtractorMOG::filter CPU branch, BackgroundSubtractorMOG.cpp:124-125:
 * apply() then frame.setTo(0, mask == 0).  frame is modified in place; mask is
 * an rows*cols scratch/out buffer. */
void oat_mog2_filter(oat_mog2 *m, uint8_t *frame, uint8_t *mask, double learning_rate);

/* row-parallel variant of oat_mog2_filter (what OpenCV's parallel_for_ does);
 * results identical, rows are independent. */
void oat_mog2_filter_mt(oat_mog2 *m, uint8_t *frame, uint8_t *mask, double learning_rate, int nthreads);
void oat_mog2_filter_from(oat_mog2 *m, const uint8_t *src, uint8_t *frame, uint8_t *mask, double learning_rate, int nthreads);

/* pool.c: persistent workers for the row-parallel stages (created on first use, woken per stage).  oat_pool_run uses the
 * calling thread's current pool (oat_pool_make_current; NULL = the process-wide default pool). */
typedef struct oat_pool oat_pool;
oat_pool *oat_pool_create(void);
void oat_pool_make_current(oat_pool *p);
void oat_pool_run(void *(*fn)(void *), void *jobs, size_t stride, int njobs);
int oat_pool_max(void);
void oat_pool_set_cpu_offset(int offset);   /* default pool: first cpu (in the spread order) of its pinned workers */

/* State inspection (tests / parity): per pixel, `nmixtures` entries. */
void oat_mog2_set_state(oat_mog2 *m, const uint8_t *modes_used, const float *weight, const float *variance,
                        const float *mean, int nframes);   /* test plumbing: continue from an exported model */
int oat_mog2_nframes(const oat_mog2 *m);
int oat_mog2_channels(const oat_mog2 *m);
const uint8_t *oat_mog2_modes_used(const oat_mog2 *m);         /* rows*cols */
/* weight[p*nmix+k], variance[p*nmix+k], mean[(p*nmix+k)*ch+c] */
void oat_mog2_get_state(const oat_mog2 *m, float *weight, float *variance, float *mean);

/* ------------------------------------------------------------- colour ----- */

/* cv::cvtColor(frame, out, COLOR_BGR2HSV) for CV_8UC3 (ColorConvert.cpp:104;
 * OpenCV RGB2HSV_b, hrange 180, hsv_shift 12). */
void oat_bgr2hsv(const uint8_t *bgr, uint8_t *hsv, size_t npixels);

/* The other entries of oat::color_conv_table (Color.h:45-51), which `framefilt col`
 * reaches through the same cv::cvtColor call (ColorConvert.cpp:104): BGR -> GREY
 * (COLOR_BGR2GRAY, what `posidet thresh` / `diff` need in front of them,
 * SimpleThreshold.cpp:46, DifferenceDetector.cpp:44), GREY -> BGR, HSV -> BGR. */
void oat_bgr2grey(const uint8_t *bgr, uint8_t *grey, size_t npixels);
void oat_grey2bgr(const uint8_t *grey, uint8_t *bgr, size_t npixels);
void oat_hsv2bgr(const uint8_t *hsv, uint8_t *bgr, size_t npixels);
/* table lookup + conversion; from/to = oat::PixelColor.  Returns bytes per output
 * pixel, -1 "nothing to be done", -2 "not possible". */
int oat_cvt_color(int from, int to, const uint8_t *src, uint8_t *dst, size_t npixels);

/* cv::inRange on 3-channel / 1-channel 8U data with integer scalar bounds
 * (HSVDetector.cpp:146-149, SimpleThreshold.cpp:171-174).  Bounds inclusive;
 * lo > hi or lo > 255 -> empty; hi saturates to 255. */
void oat_inrange3(const uint8_t *src, size_t npixels, const int lo[3], const int hi[3], uint8_t *dst);
void oat_inrange1(const uint8_t *src, size_t npixels, int lo, int hi, uint8_t *dst);

/* ---------------------------------------------------------- morphology ---- */

/* cv::erode / cv::dilate with getStructuringElement(MORPH_RECT, Size(k,k)),
 * default anchor (k/2,k/2), 1 iteration, BORDER_CONSTANT with
 * morphologyDefaultBorderValue (HSVDetector.cpp:152-156, :253-273).
 * k <= 0 is not a legal call (Oat switches the stage off instead); k == 1 copies.
 * src and dst may alias. */
void oat_erode_rect(const uint8_t *src, uint8_t *dst, int rows, int cols, int k);
void oat_dilate_rect(const uint8_t *src, uint8_t *dst, int rows, int cols, int k);

/* ------------------------------------------------------------ contours ---- */

typedef struct {
    int start_x, start_y;   /* pixel where the raster scan met the border    */
    int npoints;            /* CHAIN_APPROX_SIMPLE vertex count              */
    int first_point;        /* offset into oat_contours.points (x,y pairs)   */
    double a00, a10, a01;   /* raw Green sums of cv::moments(contourMoments) */
    double m00, m10, m01;   /* spatial moments as cv::moments returns them   */
} oat_contour;

typedef struct {
    int count;
    oat_contour *c;         /* in LIST order = reverse discovery order       */
    int *points;            /* x0,y0,x1,y1,...                               */
    int npoints_total;
} oat_contours;

/* cv::findContours(img, contours, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) of
 * OpenCV 3.1.0 (DetectorFunc.cpp:41-43) followed by cv::moments on each
 * contour (DetectorFunc.cpp:50).  img (rows*cols u8, nonzero = foreground)
 * is destroyed exactly as the reference warns (DetectorFunc.cpp:40). */
oat_contours *oat_find_contours_external(uint8_t *img, int rows, int cols);
void oat_contours_free(oat_contours *cs);

typedef struct {
    int valid;          /* Position2D::position_valid                        */
    double x, y;        /* Position2D::position                              */
    double area;        /* object_area_ out-parameter of siftContours        */
    /* diagnostics (not part of the reference's output) */
    int64_t a00, a10, a01;
    int32_t first_pixel;    /* raster index of the winning component's first pixel, -1 if none */
} oat_detection;

/* oat::siftContours, DetectorFunc.cpp:31-66 (destroys thr). */
void oat_sift_contours(uint8_t *thr, int rows, int cols, double min_area, double max_area,
                       oat_detection *out);

/* Order-free restatement of the same thing: 8-connected
 * foreground labels, 4-connected "outside" background, directed crack edges,
 * exact int64 Green sums.  This is the formulation the HIP kernels implement;
 * tests prove it equal to oat_sift_contours.  Does not modify thr. */
void oat_sift_cracks(const uint8_t *thr, int rows, int cols, double min_area, double max_area,
                     oat_detection *out);

/* ------------------------------------------------ sibling frame filters ---- */

/* framefilt bsub, BackgroundSubtractor.cpp:71-100: first frame becomes the background (u8 + f32
 * copies); alpha > 0: cv::accumulateWeighted (bg_f = src*a + bg_f*b in float, a = (float)alpha,
 * b = 1 - a) then convertTo(CV_8U) (cvRound, saturate); frame = frame - background (saturating). */
typedef struct oat_bsub oat_bsub;
oat_bsub *oat_bsub_create(int rows, int cols, int channels, double alpha);
void oat_bsub_destroy(oat_bsub *b);
void oat_bsub_filter(oat_bsub *b, uint8_t *frame);

/* framefilt thresh, Threshold.cpp:67-81: grey conversion (BGR frames) -> inRange(i_min, i_max) ->
 * frame.setTo(0, thresh == 0).  channels 3 (BGR) or 1 (GREY); in place. */
void oat_thresh_filter(uint8_t *frame, size_t npixels, int channels, int i_min, int i_max);

/* cv::blur(src, dst, Size(k,k)) on 8U: normalised box filter, anchor (k/2,k/2), BORDER_REFLECT_101,
 * dst = cvRound(sum * (1.0 / (k*k)))  (DifferenceDetector.cpp:160-161).  src/dst may alias. */
void oat_blur_box(const uint8_t *src, uint8_t *dst, int rows, int cols, int k);

/* DifferenceDetector (posidet diff), DifferenceDetector.cpp:98-173: stateful (last frame). */
typedef struct oat_diff oat_diff;
oat_diff *oat_diff_create(int rows, int cols, int diff_threshold /*10*/, int blur /*2, 0 = off*/,
                          double min_area, double max_area);
void oat_diff_destroy(oat_diff *d);
/* grey: rows*cols.  thr_out (may be NULL) receives threshold_frame_ before findContours destroys it. */
void oat_diff_detect(oat_diff *d, const uint8_t *grey, uint8_t *thr_out, oat_detection *out);

/* ----------------------------------------------------- detector chains ---- */

typedef struct {
    int h_lo, h_hi, s_lo, s_hi, v_lo, v_hi;   /* HSVDetector.h:86-88 defaults 0..256 */
    int erode;      /* 0 = off (HSVDetector.cpp:42)   */
    int dilate;     /* 10 default on (HSVDetector.cpp:43) */
    double min_area, max_area;                /* HSVDetector.h:93-94 */
} oat_hsv_params;

void oat_hsv_default_params(oat_hsv_params *p);

/* HSVDetector::detectPosition, HSVDetector.cpp:142-173 (tuning GUI excluded).
 * hsv: rows*cols*3.  thr_out (rows*cols) receives the post-morphology
 * threshold image BEFORE findContours destroys it (may be NULL). */
void oat_detect_hsv(const uint8_t *hsv, int rows, int cols, const oat_hsv_params *p,
                    uint8_t *thr_out, oat_detection *out);

/* SimpleThreshold::detectPosition, SimpleThreshold.cpp:114-134,169-182.
 * grey: rows*cols.  Uses h_lo/h_hi of p as t_min/t_max. */
void oat_detect_thresh(const uint8_t *grey, int rows, int cols, const oat_hsv_params *p,
                       uint8_t *thr_out, oat_detection *out);

/* Whole chain for one frame.  3-channel model: mog filter -> BGR2HSV -> detect_hsv
 * (frameserve -> framefilt mog -> framefilt col -C HSV -> posidet hsv).  1-channel model:
 * mog filter -> detect_thresh with [h_lo,h_hi] (frameserve -C GREY -> framefilt mog -> posidet thresh).
 * frame is consumed (modified).  scratch must hold rows*cols*5 bytes.
 * nthreads > 1 row-parallelises the per-pixel stages. */
void oat_chain_step(oat_mog2 *m, uint8_t *frame, int rows, int cols, double learning_rate,
                    const oat_hsv_params *p, uint8_t *scratch, uint8_t *thr_out,
                    oat_detection *out, int nthreads);
/* the same on a read-only input: src is copied into work (rows*cols*channels bytes) inside the row workers */
void oat_chain_step_from(oat_mog2 *m, const uint8_t *src, uint8_t *work, int rows, int cols, double learning_rate,
                         const oat_hsv_params *p, uint8_t *scratch, uint8_t *thr_out,
                         oat_detection *out, int nthreads);

/* stage 2 of 3 of the chain (framefilt col + posidet up to the morphology), see contours.c */
void oat_chain_middle(const uint8_t *work, int channels, int rows, int cols, const oat_hsv_params *p, uint8_t *hsv,
                      uint8_t *thr, uint8_t *tmp, int nthreads);

/* pipeline.c -- the chain over a sequence of frames with the reference's STAGE PIPELINING: Oat runs framefilt mog,
 * framefilt col and posidet hsv as three concurrent processes (FrameFilter.cpp:59-98, PositionDetector.cpp:58-99,
 * README.md:1708-1714), so its throughput is 1 / max(stage), not 1 / sum(stage).  n frames, frame i = frames[(first + i)
 * % nfile] (read-only), detections to out[i] (may be NULL).  pipelined = 0: the stages one after the other per frame on
 * the calling thread with t_front row workers (what oat_chain_step does); 1: three stage threads -- mog with t_front row
 * workers, col + inRange + morphology with t_mid, contour following alone -- two frame buffers between neighbours (a
 * sink's one shared slot plus the consumer's own copy).  stage_s[3] receives the seconds each stage was busy.  Returns
 * the wall-clock seconds of the n frames.  Results are those of n oat_chain_step calls. */
double oat_pipeline_run(oat_mog2 *m, const uint8_t *const *frames, int nfile, int first, int n, int rows, int cols,
                        double learning_rate, const oat_hsv_params *p, int t_front, int t_mid, int pipelined,
                        oat_detection *out, double stage_s[3]);

/* ------------------------------------------------- posifilt kalman -------- */

/* `oat posifilt kalman` (src/positionfilter/KalmanFilter2D.cpp:95-210): constant-velocity
 * 4-state filter per position stream; see kalman.c for the restated quirks. */
typedef struct {
    double dt;            /* 0.02  --dt                         */
    double timeout;       /* 0     --timeout, seconds           */
    double sigma_accel;   /* 5.0   --sigma-accel                */
    double sigma_noise;   /* 0.0   --sigma-noise                */
} oat_kalman_params;

typedef struct {
    int position_valid, velocity_valid;
    double x, y, vx, vy;
} oat_kalman_out;

typedef struct oat_kalman oat_kalman;
void oat_kalman_default_params(oat_kalman_params *p);
oat_kalman *oat_kalman_create(const oat_kalman_params *p);
void oat_kalman_destroy(oat_kalman *k);
/* KalmanFilter2D::filter(position): in = posidet's (position_valid, position) */
void oat_kalman_filter(oat_kalman *k, int position_valid, double x, double y, oat_kalman_out *out);

/* posifilt homography (HomographyTransform2D.cpp:62-107): cv::perspectiveTransform of the position (and, with its
 * offsets zeroed, of the velocity) through the row-major 3x3 matrix h; in place. */
void oat_homography_filter(const double h[9], int position_valid, double *x, double *y, int velocity_valid,
                           double *vx, double *vy);

#ifdef __cplusplus
}
#endif
#endif
