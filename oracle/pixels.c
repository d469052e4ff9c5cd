/*
 * pixels.c -- ORACLE (test infrastructure): the element-wise and stencil
 * stages between background subtraction and contour analysis.
 *
 *   oat_bgr2hsv     ColorConvert.cpp:101-107  cv::cvtColor(.., COLOR_BGR2HSV)
 *                   = OpenCV 3.1.0 imgproc/color.cpp RGB2HSV_b (hrange 180)
 *   oat_inrange3/1  HSVDetector.cpp:146-149 / SimpleThreshold.cpp:171-174
 *                   = core/arithm.cpp cv::inRange with scalar bounds on 8U
 *   oat_erode_rect  HSVDetector.cpp:152-153 (+ :253-262 structuring element)
 *   oat_dilate_rect HSVDetector.cpp:155-156 (+ :264-273)
 *                   = imgproc/morph.cpp, MORPH_RECT k x k, anchor (k/2,k/2),
 *                     BORDER_CONSTANT, morphologyDefaultBorderValue()
 *
 * PARITY UNPINNED by the reference (no tests for this path); pinned by the
 * hand-derived known answers in tests/golden/.
 */
#include "oat_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------- BGR2HSV ---- */

static int sdiv_table[256];
static int hdiv_table180[256];
static int hsv_tables_ready = 0;

static void hsv_tables_init(void)
{
    const int hsv_shift = 12;
    sdiv_table[0] = hdiv_table180[0] = 0;
    for (int i = 1; i < 256; i++) {
        /* saturate_cast<int>(double) == cvRound == round-half-to-even */
        sdiv_table[i] = (int)lrint((255 << hsv_shift) / (1. * i));
        hdiv_table180[i] = (int)lrint((180 << hsv_shift) / (6. * i));
    }
    hsv_tables_ready = 1;
}

void oat_bgr2hsv(const uint8_t *src, uint8_t *dst, size_t n)
{
    const int hsv_shift = 12;
    const int hr = 180;
    if (!hsv_tables_ready) hsv_tables_init();
    for (size_t i = 0; i < n; i++, src += 3, dst += 3) {
        int b = src[0], g = src[1], r = src[2];
        int h, s, v = b;
        int vmin = b, diff;
        int vr, vg;

        if (g > v) v = g;
        if (r > v) v = r;
        if (g < vmin) vmin = g;
        if (r < vmin) vmin = r;

        diff = v - vmin;
        vr = v == r ? -1 : 0;
        vg = v == g ? -1 : 0;

        s = (diff * sdiv_table[v] + (1 << (hsv_shift - 1))) >> hsv_shift;
        h = (vr & (g - b)) +
            (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
        h = (h * hdiv_table180[diff] + (1 << (hsv_shift - 1))) >> hsv_shift;
        h += h < 0 ? hr : 0;

        dst[0] = (uint8_t)(h < 0 ? 0 : h > 255 ? 255 : h);   /* saturate_cast<uchar> */
        dst[1] = (uint8_t)s;
        dst[2] = (uint8_t)v;
    }
}

/* ------------------------------------------------------------- inRange ---- */

/* cv::inRange, scalar bounds, 8U source: the bounds are converted to int; a
 * channel with lo > hi, lo > 255 or hi < 0 can never match; otherwise bounds
 * saturate to [0,255] and both ends are inclusive. */
static void inrange_bounds(int lo, int hi, int *l, int *h)
{
    if (lo > hi || lo > 255 || hi < 0) { *l = 1; *h = 0; return; }
    *l = lo < 0 ? 0 : lo;
    *h = hi > 255 ? 255 : hi;
}

void oat_inrange3(const uint8_t *src, size_t n, const int lo[3], const int hi[3], uint8_t *dst)
{
    int l[3], h[3];
    for (int c = 0; c < 3; c++) inrange_bounds(lo[c], hi[c], &l[c], &h[c]);
    for (size_t i = 0; i < n; i++, src += 3) {
        int ok = src[0] >= l[0] && src[0] <= h[0] &&
                 src[1] >= l[1] && src[1] <= h[1] &&
                 src[2] >= l[2] && src[2] <= h[2];
        dst[i] = ok ? 255 : 0;
    }
}

void oat_inrange1(const uint8_t *src, size_t n, int lo, int hi, uint8_t *dst)
{
    int l, h;
    inrange_bounds(lo, hi, &l, &h);
    for (size_t i = 0; i < n; i++)
        dst[i] = (src[i] >= l && src[i] <= h) ? 255 : 0;
}

/* ---------------------------------------------------------- morphology ---- */

/* One k x k rectangular min (erode) or max (dilate).  Window of output (x,y)
 * is x' in [x-a, x-a+k-1], y' likewise, a = k/2 (integer division): OpenCV
 * does not reflect the element for dilation.  Samples outside the image take
 * the constant border: 255 for erode, 0 for dilate. */
static void morph_rect(const uint8_t *src, uint8_t *dst, int rows, int cols, int k, int is_erode)
{
    size_t n = (size_t)rows * cols;
    if (k <= 1) { if (dst != src) memmove(dst, src, n); return; }
    const int a = k / 2;
    const uint8_t border = is_erode ? 255 : 0;
    uint8_t *tmp = (uint8_t *)malloc(n);
    /* row pass */
    for (int y = 0; y < rows; y++) {
        const uint8_t *s = src + (size_t)y * cols;
        uint8_t *t = tmp + (size_t)y * cols;
        for (int x = 0; x < cols; x++) {
            uint8_t acc = border;
            int first = 1;
            for (int j = 0; j < k; j++) {
                int xx = x - a + j;
                uint8_t v = (xx < 0 || xx >= cols) ? border : s[xx];
                if (first) { acc = v; first = 0; }
                else if (is_erode ? (v < acc) : (v > acc)) acc = v;
            }
            t[x] = acc;
        }
    }
    /* column pass */
    for (int y = 0; y < rows; y++) {
        uint8_t *d = dst + (size_t)y * cols;
        for (int x = 0; x < cols; x++) {
            uint8_t acc = border;
            int first = 1;
            for (int j = 0; j < k; j++) {
                int yy = y - a + j;
                uint8_t v = (yy < 0 || yy >= rows) ? border : tmp[(size_t)yy * cols + x];
                if (first) { acc = v; first = 0; }
                else if (is_erode ? (v < acc) : (v > acc)) acc = v;
            }
            d[x] = acc;
        }
    }
    free(tmp);
}

void oat_erode_rect(const uint8_t *src, uint8_t *dst, int rows, int cols, int k)
{
    morph_rect(src, dst, rows, cols, k, 1);
}

void oat_dilate_rect(const uint8_t *src, uint8_t *dst, int rows, int cols, int k)
{
    morph_rect(src, dst, rows, cols, k, 0);
}

/* ---------------------------------------------------------------- blur ---- */

static int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

void oat_blur_box(const uint8_t *src, uint8_t *dst, int rows, int cols, int k)
{
    size_t n = (size_t)rows * cols;
    if (k <= 1) { if (dst != src) memmove(dst, src, n); return; }
    const int a = k / 2;
    const double scale = 1.0 / ((double)k * k);
    int *rowsum = (int *)malloc(n * sizeof(int));
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            int s = 0;
            for (int j = 0; j < k; j++) s += src[(size_t)y * cols + reflect101(x - a + j, cols)];
            rowsum[(size_t)y * cols + x] = s;
        }
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            int s = 0;
            for (int j = 0; j < k; j++) s += rowsum[(size_t)reflect101(y - a + j, rows) * cols + x];
            long r = lrint(s * scale);                 /* saturate_cast<uchar>(cvRound(.)) */
            dst[(size_t)y * cols + x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    free(rowsum);
}

/* ------------------------------------------------ sibling frame filters ---- */

struct oat_bsub {
    int rows, cols, ch, set;
    double alpha;
    uint8_t *bg;
    float *bg_f;
};

oat_bsub *oat_bsub_create(int rows, int cols, int channels, double alpha)
{
    oat_bsub *b = (oat_bsub *)calloc(1, sizeof(*b));
    size_t n = (size_t)rows * cols * channels;
    b->rows = rows; b->cols = cols; b->ch = channels; b->alpha = alpha;
    b->bg = (uint8_t *)malloc(n);
    b->bg_f = (float *)malloc(n * sizeof(float));
    return b;
}

void oat_bsub_destroy(oat_bsub *b)
{
    if (!b) return;
    free(b->bg); free(b->bg_f); free(b);
}

void oat_bsub_filter(oat_bsub *b, uint8_t *frame)
{
    size_t n = (size_t)b->rows * b->cols * b->ch;
    if (!b->set) {                                  /* setBackgroundImage (BackgroundSubtractor.cpp:79-85) */
        memcpy(b->bg, frame, n);
        for (size_t i = 0; i < n; i++) b->bg_f[i] = (float)frame[i];
        b->set = 1;
    }
    if (b->alpha > 0.0) {
        const float a = (float)b->alpha, bb = 1 - a;    /* accW_<uchar,float>: AT a = (AT)alpha, b = 1 - a */
        for (size_t i = 0; i < n; i++) {
            b->bg_f[i] = frame[i] * a + b->bg_f[i] * bb;
            long r = lrintf(b->bg_f[i]);                /* convertTo(CV_8U): saturate_cast<uchar>(cvRound) */
            b->bg[i] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
        }
    }
    for (size_t i = 0; i < n; i++)                      /* frame - background, saturating */
        frame[i] = frame[i] > b->bg[i] ? (uint8_t)(frame[i] - b->bg[i]) : 0;
}

void oat_bgr2grey(const uint8_t *bgr, uint8_t *grey, size_t n)
{
    for (size_t i = 0; i < n; i++, bgr += 3)
        grey[i] = (uint8_t)((1868 * bgr[0] + 9617 * bgr[1] + 4899 * bgr[2] + (1 << 13)) >> 14);
}

void oat_thresh_filter(uint8_t *frame, size_t n, int channels, int i_min, int i_max)
{
    int l, h;
    inrange_bounds(i_min, i_max, &l, &h);
    for (size_t i = 0; i < n; i++) {
        int g = channels == 3
            ? ((1868 * frame[3 * i] + 9617 * frame[3 * i + 1] + 4899 * frame[3 * i + 2] + (1 << 13)) >> 14)
            : frame[i];
        if (!(g >= l && g <= h))
            for (int c = 0; c < channels; c++) frame[i * channels + c] = 0;
    }
}
