/*
 * contours.c -- ORACLE (test infrastructure): the blob stage of posidet.
 *
 *   oat_find_contours_external   DetectorFunc.cpp:41-43
 *        cv::findContours(frame, contours, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)
 *        = OpenCV 3.1.0 imgproc/contours.cpp: cvStartFindContours (zero the
 *          1-pixel image frame, binarise), cvFindNextContour (raster scan,
 *          Suzuki-Abe outer-border starts, the `lnbd` "am I inside an already
 *          traced border" test that implements RETR_EXTERNAL), icvFetchContour
 *          (8-connected border following with +2 / -126 marks and
 *          CHAIN_APPROX_SIMPLE vertex emission), and the reversed sibling list.
 *        followed by DetectorFunc.cpp:50  cv::moments(Mat(contour))
 *        = imgproc/moments.cpp contourMoments (Green's theorem, doubles).
 *   oat_sift_contours            DetectorFunc.cpp:31-66 (oat::siftContours)
 *   oat_sift_cracks              the order-free formulation of SURVEY.md 7.1,
 *                                which is what the HIP kernels implement.
 *   oat_detect_hsv / _thresh     HSVDetector.cpp:142-173 / SimpleThreshold.cpp:114-134
 *
 * PARITY UNPINNED by the reference; pinned by hand-derived known answers
 * (tests/golden/contours.json) and by the sequential-vs-order-free cross-check.
 */
#include "oat_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* -------------------------------------------- sequential border following -- */

typedef struct { int *v; int n, cap; } ivec;

static void ivec_push2(ivec *a, int x, int y)
{
    if (a->n + 2 > a->cap) {
        a->cap = a->cap ? a->cap * 2 : 256;
        a->v = (int *)realloc(a->v, (size_t)a->cap * sizeof(int));
    }
    a->v[a->n++] = x; a->v[a->n++] = y;
}

/* icvCodeDeltas: chain code s -> (dx,dy); 0=E 1=NE 2=N 3=NW 4=W 5=SW 6=S 7=SE */
static const int code_dx[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };
static const int code_dy[8] = { 0, -1, -1, -1, 0, 1, 1, 1 };

/* icvFetchContour for an OUTER border (s_end = 4), CHAIN_APPROX_SIMPLE.
 * ptr points at the start pixel inside the signed-char image. */
static void fetch_contour(signed char *ptr, int step, int ptx, int pty, ivec *pts)
{
    const signed char nbd = 2;
    int deltas[16];
    signed char *i0 = ptr, *i1, *i3, *i4 = 0;
    int prev_s = -1, s, s_end;

    /* CV_INIT_3X3_DELTAS(deltas, step, 1) */
    deltas[0] = 1; deltas[1] = -step + 1; deltas[2] = -step; deltas[3] = -step - 1;
    deltas[4] = -1; deltas[5] = step - 1; deltas[6] = step; deltas[7] = step + 1;
    memcpy(deltas + 8, deltas, 8 * sizeof(deltas[0]));

    s_end = s = 4;   /* not a hole */
    do {
        s = (s - 1) & 7;
        i1 = i0 + deltas[s];
        if (*i1 != 0)
            break;
    } while (s != s_end);

    if (s == s_end) {            /* single pixel domain */
        *i0 = (signed char)(nbd | -128);
        ivec_push2(pts, ptx, pty);
        return;
    }

    i3 = i0;
    prev_s = s ^ 4;

    for (;;) {
        s_end = s;
        for (;;) {
            i4 = i3 + deltas[++s];
            if (*i4 != 0)
                break;
        }
        s &= 7;

        /* check "right" bound */
        if ((unsigned)(s - 1) < (unsigned)s_end)
            *i3 = (signed char)(nbd | -128);
        else if (*i3 == 1)
            *i3 = nbd;

        if (s != prev_s) {
            ivec_push2(pts, ptx, pty);
            prev_s = s;
        }
        ptx += code_dx[s];
        pty += code_dy[s];

        if (i4 == i0 && i3 == i1)
            break;

        i3 = i4;
        s = (s + 4) & 7;
    }
}

/* contourMoments (imgproc/moments.cpp), integer points, orders 0 and 1 only. */
static void contour_moments(const int *pts, int lpt, oat_contour *c)
{
    double a00 = 0, a10 = 0, a01 = 0;
    c->a00 = c->a10 = c->a01 = 0;
    c->m00 = c->m10 = c->m01 = 0;
    if (lpt == 0) return;
    double xi_1 = pts[2 * (lpt - 1)], yi_1 = pts[2 * (lpt - 1) + 1];
    for (int i = 0; i < lpt; i++) {
        double xi = pts[2 * i], yi = pts[2 * i + 1];
        double dxy = xi_1 * yi - xi * yi_1;
        double xii_1 = xi_1 + xi;
        double yii_1 = yi_1 + yi;
        a00 += dxy;
        a10 += dxy * xii_1;
        a01 += dxy * yii_1;
        xi_1 = xi; yi_1 = yi;
    }
    c->a00 = a00; c->a10 = a10; c->a01 = a01;
    if (fabs(a00) > FLT_EPSILON) {
        double db1_2, db1_6;
        if (a00 > 0) { db1_2 = 0.5; db1_6 = 0.16666666666666666666666666666667; }
        else { db1_2 = -0.5; db1_6 = -0.16666666666666666666666666666667; }
        c->m00 = a00 * db1_2;
        c->m10 = a10 * db1_6;
        c->m01 = a01 * db1_6;
    }
}

oat_contours *oat_find_contours_external(uint8_t *image, int rows, int cols)
{
    oat_contours *cs = (oat_contours *)calloc(1, sizeof(*cs));
    signed char *img0 = (signed char *)image;
    const int step = cols;
    ivec pts = { 0, 0, 0 };
    int ccap = 0;

    if (rows < 1 || cols < 1) return cs;

    /* cvStartFindContours: make zero borders, then threshold to 0/1 */
    memset(img0, 0, (size_t)cols);
    memset(img0 + (size_t)step * (rows - 1), 0, (size_t)cols);
    for (int y = 1; y < rows - 1; y++) {
        img0[(size_t)y * step] = 0;
        img0[(size_t)y * step + cols - 1] = 0;
    }
    for (size_t i = 0, n = (size_t)rows * cols; i < n; i++)
        img0[i] = image[i] ? 1 : 0;

    /* cvFindNextContour, mode == CV_RETR_EXTERNAL (0), repeated to exhaustion */
    const int width = cols - 1, height = rows - 1;
    signed char *img = img0 + step;
    int lnbd_x = 0, lnbd_y = 1;
    int x = 1, y = 1;
    int prev = 0;   /* img[x-1] at (1,1) is the zeroed border column */

    for (; y < height; y++, img += step) {
        int p = 0;
        for (; x < width; x++) {
            for (; x < width && (p = img[x]) == prev; x++)
                ;
            if (x >= width)
                break;
            {
                int is_hole = 0;
                if (!(prev == 0 && p == 1)) {      /* if not external contour */
                    /* check hole */
                    if (p != 0 || prev < 1)
                        goto resume_scan;
                    if (prev & -2)
                        lnbd_x = x - 1;
                    is_hole = 1;
                }
                if (is_hole || img0[(size_t)lnbd_y * step + lnbd_x] > 0)
                    goto resume_scan;

                lnbd_x = x;
                {
                    if (cs->count == ccap) {
                        ccap = ccap ? ccap * 2 : 64;
                        cs->c = (oat_contour *)realloc(cs->c, (size_t)ccap * sizeof(oat_contour));
                    }
                    oat_contour *c = &cs->c[cs->count++];
                    c->start_x = x; c->start_y = y;
                    c->first_point = pts.n / 2;
                    fetch_contour(img + x, step, x, y, &pts);
                    c->npoints = pts.n / 2 - c->first_point;
                }
                p = img[x];   /* the scan resumes with prev = the mark just written */
            resume_scan:
                prev = p;
                if (prev & -2)
                    lnbd_x = x;
            }
        }
        lnbd_x = 0;
        lnbd_y = y + 1;
        x = 1;
        prev = 0;
    }

    cs->points = pts.v;
    cs->npoints_total = pts.n / 2;
    for (int i = 0; i < cs->count; i++)
        contour_moments(cs->points + 2 * cs->c[i].first_point, cs->c[i].npoints, &cs->c[i]);

    /* The contour tree links each new sibling at the head, so the sequence the
     * C++ wrapper walks is reverse discovery order. */
    for (int i = 0, j = cs->count - 1; i < j; i++, j--) {
        oat_contour t = cs->c[i]; cs->c[i] = cs->c[j]; cs->c[j] = t;
    }
    return cs;
}

void oat_contours_free(oat_contours *cs)
{
    if (!cs) return;
    free(cs->c); free(cs->points); free(cs);
}

/* oat::siftContours, DetectorFunc.cpp:31-66 */
void oat_sift_contours(uint8_t *thr, int rows, int cols, double min_area, double max_area,
                       oat_detection *out)
{
    oat_contours *cs = oat_find_contours_external(thr, rows, cols);
    double object_area = 0;
    memset(out, 0, sizeof(*out));
    out->valid = 0;
    out->first_pixel = -1;
    for (int i = 0; i < cs->count; i++) {
        const oat_contour *c = &cs->c[i];
        double contour_area = c->m00;
        if (contour_area >= min_area && contour_area < max_area && contour_area > object_area) {
            out->x = c->m10 / contour_area;
            out->y = c->m01 / contour_area;
            out->valid = 1;
            object_area = contour_area;
            out->a00 = (int64_t)c->a00; out->a10 = (int64_t)c->a10; out->a01 = (int64_t)c->a01;
            out->first_pixel = c->start_y * cols + c->start_x;
        }
    }
    out->area = object_area;
    oat_contours_free(cs);
}

/* ------------------------------------------------- order-free formulation -- */

/* Labels: fg 8-connected components (label = raster index of first pixel),
 * "outside" = 4-connected background reachable from the (zeroed) image frame.
 * For each fg pixel p and each 4-neighbour that is OUTSIDE background, one
 * directed polygon edge p->q is emitted at most:
 *   side    bg pixel   B (checked first)  A (checked second)
 *   left    (x-1,y)    (x-1,y+1)          (x,  y+1)
 *   bottom  (x,y+1)    (x+1,y+1)          (x+1,y)
 *   right   (x+1,y)    (x+1,y-1)          (x,  y-1)
 *   top     (x,y-1)    (x-1,y-1)          (x-1,y)
 * q = B if B is fg, else A if A is fg, else no edge.  Per edge
 * d = x0*y1 - x1*y0; a00 += d; a10 += d*(x0+x1); a01 += d*(y0+y1)  (int64). */
void oat_sift_cracks(const uint8_t *thr, int rows, int cols, double min_area, double max_area,
                     oat_detection *out)
{
    size_t n = (size_t)rows * cols;
    memset(out, 0, sizeof(*out));
    out->first_pixel = -1;
    if (rows < 3 || cols < 3) { return; }

    uint8_t *fg = (uint8_t *)calloc(n, 1);
    uint8_t *outside = (uint8_t *)calloc(n, 1);
    int32_t *label = (int32_t *)malloc(n * sizeof(int32_t));
    int32_t *stack = (int32_t *)malloc(n * sizeof(int32_t));

    for (int y = 1; y < rows - 1; y++)
        for (int x = 1; x < cols - 1; x++)
            fg[(size_t)y * cols + x] = thr[(size_t)y * cols + x] ? 1 : 0;

    /* outside background: flood from (0,0) (the whole frame ring is bg and 4-connected) */
    int sp = 0;
    stack[sp++] = 0; outside[0] = 1;
    while (sp) {
        int32_t i = stack[--sp];
        int x = i % cols, y = i / cols;
        static const int dx4[4] = { 1, -1, 0, 0 }, dy4[4] = { 0, 0, 1, -1 };
        for (int k = 0; k < 4; k++) {
            int xx = x + dx4[k], yy = y + dy4[k];
            if (xx < 0 || yy < 0 || xx >= cols || yy >= rows) continue;
            int32_t j = yy * cols + xx;
            if (!fg[j] && !outside[j]) { outside[j] = 1; stack[sp++] = j; }
        }
    }

    /* fg labels, raster order so label == first pixel index */
    for (size_t i = 0; i < n; i++) label[i] = -1;
    /* sums are kept in a table indexed by first pixel (sparse) */
    int64_t *a00 = (int64_t *)calloc(n, sizeof(int64_t));
    int64_t *a10 = (int64_t *)calloc(n, sizeof(int64_t));
    int64_t *a01 = (int64_t *)calloc(n, sizeof(int64_t));
    for (size_t i0 = 0; i0 < n; i0++) {
        if (!fg[i0] || label[i0] >= 0) continue;
        sp = 0; stack[sp++] = (int32_t)i0; label[i0] = (int32_t)i0;
        while (sp) {
            int32_t i = stack[--sp];
            int x = i % cols, y = i / cols;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    if (!dx && !dy) continue;
                    int32_t j = (y + dy) * cols + (x + dx);   /* interior pixels only: in range */
                    if (fg[j] && label[j] < 0) { label[j] = (int32_t)i0; stack[sp++] = j; }
                }
        }
    }

#define FG(xx, yy) (fg[(size_t)(yy) * cols + (xx)])
#define OUT(xx, yy) (outside[(size_t)(yy) * cols + (xx)])
    for (int y = 1; y < rows - 1; y++)
        for (int x = 1; x < cols - 1; x++) {
            if (!FG(x, y)) continue;
            int32_t L = label[(size_t)y * cols + x];
            /* {bgx,bgy, Bx,By, Ax,Ay} relative */
            static const int T[4][6] = {
                { -1, 0, -1, 1, 0, 1 },    /* left   */
                { 0, 1, 1, 1, 1, 0 },      /* bottom */
                { 1, 0, 1, -1, 0, -1 },    /* right  */
                { 0, -1, -1, -1, -1, 0 },  /* top    */
            };
            for (int k = 0; k < 4; k++) {
                if (!OUT(x + T[k][0], y + T[k][1])) continue;
                int qx, qy;
                if (FG(x + T[k][2], y + T[k][3])) { qx = x + T[k][2]; qy = y + T[k][3]; }
                else if (FG(x + T[k][4], y + T[k][5])) { qx = x + T[k][4]; qy = y + T[k][5]; }
                else continue;
                int64_t d = (int64_t)x * qy - (int64_t)qx * y;
                a00[L] += d; a10[L] += d * (x + qx); a01[L] += d * (y + qy);
            }
        }
#undef FG
#undef OUT

    /* selection: the reference walks the list in reverse discovery order with a
     * strict '>' -- equivalently: descending first-pixel index. */
    double best = 0;
    for (size_t ii = n; ii-- > 0;) {
        if (label[ii] != (int32_t)ii) continue;
        if (a00[ii] == 0) continue;               /* fabs(a00) > FLT_EPSILON fails */
        double s2 = a00[ii] > 0 ? 0.5 : -0.5;
        double s6 = a00[ii] > 0 ? 0.16666666666666666666666666666667 : -0.16666666666666666666666666666667;
        double m00 = (double)a00[ii] * s2;
        if (m00 >= min_area && m00 < max_area && m00 > best) {
            double m10 = (double)a10[ii] * s6, m01 = (double)a01[ii] * s6;
            out->x = m10 / m00; out->y = m01 / m00; out->valid = 1; best = m00;
            out->a00 = a00[ii]; out->a10 = a10[ii]; out->a01 = a01[ii];
            out->first_pixel = (int32_t)ii;
        }
    }
    out->area = best;
    free(fg); free(outside); free(label); free(stack); free(a00); free(a10); free(a01);
}

/* ------------------------------------------------------ posidet diff -------- */

struct oat_diff {
    int rows, cols, thresh, blur, last_set;
    double min_area, max_area;
    uint8_t *last;
};

oat_diff *oat_diff_create(int rows, int cols, int diff_threshold, int blur, double min_area, double max_area)
{
    oat_diff *d = (oat_diff *)calloc(1, sizeof(*d));
    d->rows = rows; d->cols = cols; d->thresh = diff_threshold; d->blur = blur;
    d->min_area = min_area; d->max_area = max_area;
    d->last = (uint8_t *)malloc((size_t)rows * cols);
    return d;
}

void oat_diff_destroy(oat_diff *d)
{
    if (!d) return;
    free(d->last); free(d);
}

/* DifferenceDetector::applyThreshold (DifferenceDetector.cpp:154-173) + siftContours (:109-113) */
void oat_diff_detect(oat_diff *d, const uint8_t *grey, uint8_t *thr_out, oat_detection *out)
{
    size_t n = (size_t)d->rows * d->cols;
    uint8_t *thr = (uint8_t *)malloc(n);
    if (d->last_set) {
        for (size_t i = 0; i < n; i++) {
            int a = grey[i] > d->last[i] ? grey[i] - d->last[i] : d->last[i] - grey[i];   /* cv::absdiff */
            thr[i] = a > d->thresh ? 255 : 0;                                               /* THRESH_BINARY */
        }
        if (d->blur > 0) oat_blur_box(thr, thr, d->rows, d->cols, d->blur);
    } else {
        memcpy(thr, grey, n);            /* first frame: threshold_frame_ = frame.clone() */
        d->last_set = 1;
    }
    memcpy(d->last, grey, n);
    if (thr_out) memcpy(thr_out, thr, n);
    oat_sift_contours(thr, d->rows, d->cols, d->min_area, d->max_area, out);
    free(thr);
}

/* ------------------------------------------------------- detector chains -- */

void oat_hsv_default_params(oat_hsv_params *p)
{
    p->h_lo = 0; p->h_hi = 256; p->s_lo = 0; p->s_hi = 256; p->v_lo = 0; p->v_hi = 256;
    p->erode = 0; p->dilate = 10;
    p->min_area = 0.0; p->max_area = DBL_MAX;
}

static void morph_and_sift(uint8_t *thr, int rows, int cols, const oat_hsv_params *p,
                           uint8_t *thr_out, oat_detection *out)
{
    if (p->erode > 0) oat_erode_rect(thr, thr, rows, cols, p->erode);
    if (p->dilate > 0) oat_dilate_rect(thr, thr, rows, cols, p->dilate);
    if (thr_out) memcpy(thr_out, thr, (size_t)rows * cols);
    oat_sift_contours(thr, rows, cols, p->min_area, p->max_area, out);
}

void oat_detect_hsv(const uint8_t *hsv, int rows, int cols, const oat_hsv_params *p,
                    uint8_t *thr_out, oat_detection *out)
{
    size_t n = (size_t)rows * cols;
    uint8_t *thr = (uint8_t *)malloc(n);
    int lo[3] = { p->h_lo, p->s_lo, p->v_lo }, hi[3] = { p->h_hi, p->s_hi, p->v_hi };
    oat_inrange3(hsv, n, lo, hi, thr);
    morph_and_sift(thr, rows, cols, p, thr_out, out);
    free(thr);
}

void oat_detect_thresh(const uint8_t *grey, int rows, int cols, const oat_hsv_params *p,
                       uint8_t *thr_out, oat_detection *out)
{
    size_t n = (size_t)rows * cols;
    uint8_t *thr = (uint8_t *)malloc(n);
    oat_inrange1(grey, n, p->h_lo, p->h_hi, thr);
    morph_and_sift(thr, rows, cols, p, thr_out, out);
    free(thr);
}

/* Row-parallel helper for the element-wise / stencil stages of the CPU baseline (what OpenCV's
 * parallel_for_ does for cvtColor / inRange / morphology).  Results do not depend on nthreads. */
typedef struct {
    int stage;                 /* 0: bgr2hsv + inRange, 1: erode/dilate row pass, 2: column pass */
    const uint8_t *src; uint8_t *dst; uint8_t *aux;
    int rows, cols, k, is_erode, y0, y1;
    const int *lo, *hi;
} chain_job;

__attribute__((optimize("O3"))) static void *chain_worker(void *arg)
{
    chain_job *j = (chain_job *)arg;
    const int cols = j->cols;
    if (j->stage == 0) {
        size_t off = (size_t)j->y0 * cols, n = (size_t)(j->y1 - j->y0) * cols;
        if (j->k == 1) {          /* GREY chain: framefilt mog -> posidet thresh (no colour conversion) */
            oat_inrange1(j->src + off, n, j->lo[0], j->hi[0], j->dst + off);
            return NULL;
        }
        oat_bgr2hsv(j->src + off * 3, j->aux + off * 3, n);
        oat_inrange3(j->aux + off * 3, n, j->lo, j->hi, j->dst + off);
        return NULL;
    }
    /* Rectangular erosion / dilation, separable: window [x - a, x - a + k - 1] (a = k / 2, not reflected), samples outside
     * the image = the operation's identity (255 for min, 0 for max) -- cv::erode / cv::dilate's default border.  Written as
     * whole-row min / max over shifted rows: the same numbers as a per-pixel window loop, in a form the compiler vectorises
     * (r04: the per-pixel form with its border tests inside made this stage, not MOG2, the slowest of the CPU baseline). */
    const int k = j->k, a = k / 2, is_erode = j->is_erode;
    const uint8_t border = is_erode ? 255 : 0;
    for (int y = j->y0; y < j->y1; y++) {
        uint8_t *restrict d = j->dst + (size_t)y * cols;
        memset(d, border, (size_t)cols);
        for (int t = 0; t < k; t++) {
            const int off = t - a;
            if (j->stage == 1) {                       /* row pass: d[x] = op(d[x], src[y][x + off]) where x + off is inside */
                const uint8_t *restrict srow = j->src + (size_t)y * cols;
                const int x0 = off < 0 ? -off : 0, x1 = off > 0 ? cols - off : cols;
                if (is_erode) { for (int x = x0; x < x1; x++) { const uint8_t v = srow[x + off]; d[x] = v < d[x] ? v : d[x]; } }
                else          { for (int x = x0; x < x1; x++) { const uint8_t v = srow[x + off]; d[x] = v > d[x] ? v : d[x]; } }
            } else {                                   /* column pass: d = op(d, src[y + off]) where the row exists */
                const int yy = y + off;
                if (yy < 0 || yy >= j->rows) continue;
                const uint8_t *restrict srow = j->src + (size_t)yy * cols;
                if (is_erode) { for (int x = 0; x < cols; x++) d[x] = srow[x] < d[x] ? srow[x] : d[x]; }
                else          { for (int x = 0; x < cols; x++) d[x] = srow[x] > d[x] ? srow[x] : d[x]; }
            }
        }
    }
    return NULL;
}

static void chain_parallel(chain_job proto, int nthreads)
{
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 512) nthreads = 512;
    if (nthreads > proto.rows) nthreads = proto.rows;
    chain_job jobs[512];
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = proto;
        jobs[t].y0 = (int)((long)proto.rows * t / nthreads);
        jobs[t].y1 = (int)((long)proto.rows * (t + 1) / nthreads);
    }
    oat_pool_run(chain_worker, jobs, sizeof jobs[0], nthreads);    /* persistent workers (pool.c) */
}

static void morph_mt(uint8_t *img, uint8_t *tmp, int rows, int cols, int k, int is_erode, int nthreads)
{
    if (k <= 1) return;
    chain_job j = { 1, img, tmp, NULL, rows, cols, k, is_erode, 0, 0, NULL, NULL };
    chain_parallel(j, nthreads);
    j.stage = 2; j.src = tmp; j.dst = img;
    chain_parallel(j, nthreads);
}

void oat_chain_step(oat_mog2 *m, uint8_t *frame, int rows, int cols, double learning_rate,
                    const oat_hsv_params *p, uint8_t *scratch, uint8_t *thr_out,
                    oat_detection *out, int nthreads)
{
    oat_chain_step_from(m, NULL, frame, rows, cols, learning_rate, p, scratch, thr_out, out, nthreads);
}

/* The three stages of the chain as Oat runs them -- three component processes (pipeline.c runs them concurrently):
 *   front  = framefilt mog                         oat_mog2_filter_from (mog2.c)
 *   middle = framefilt col -C HSV + posidet hsv up to the morphology (or the GREY window of posidet thresh)
 *   back   = siftContours                          oat_sift_contours
 * middle: work = the filtered frame (rows*cols*channels), hsv = rows*cols*3 scratch, thr = rows*cols out,
 * tmp = rows*cols morphology temporary. */
void oat_chain_middle(const uint8_t *work, int channels, int rows, int cols, const oat_hsv_params *p, uint8_t *hsv,
                      uint8_t *thr, uint8_t *tmp, int nthreads)
{
    int lo[3] = { p->h_lo, p->s_lo, p->v_lo }, hi[3] = { p->h_hi, p->s_hi, p->v_hi };
    chain_job j = { 0, (uint8_t *)work, thr, hsv, rows, cols, channels, 0, 0, 0, lo, hi };   /* k carries the channel count for stage 0 */
    chain_parallel(j, nthreads);
    if (p->erode > 0) morph_mt(thr, tmp, rows, cols, p->erode, 1, nthreads);
    if (p->dilate > 0) morph_mt(thr, tmp, rows, cols, p->dilate, 0, nthreads);
}

/* src != NULL: the caller's frame stays untouched; `work` (rows*cols*channels bytes) receives the filtered copy */
void oat_chain_step_from(oat_mog2 *m, const uint8_t *src, uint8_t *work, int rows, int cols, double learning_rate,
                         const oat_hsv_params *p, uint8_t *scratch, uint8_t *thr_out,
                         oat_detection *out, int nthreads)
{
    size_t n = (size_t)rows * cols;
    uint8_t *mask = scratch;            /* n   (reused as the morphology temporary) */
    uint8_t *hsv = scratch + n;         /* 3n  */
    uint8_t *thr = scratch + 4 * n;     /* n   */
    oat_mog2_filter_from(m, src, work, mask, learning_rate, nthreads);
    oat_chain_middle(work, oat_mog2_channels(m), rows, cols, p, hsv, thr, mask, nthreads);
    if (thr_out) memcpy(thr_out, thr, n);
    oat_sift_contours(thr, rows, cols, p->min_area, p->max_area, out);
}
