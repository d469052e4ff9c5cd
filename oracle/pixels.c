/*
 * pixels.c -- ORACLE (test infrastructure): the element-wise and stencil
 * stages between background subtraction and contour analysis.
 *
 *   oat_bgr2hsv     ColorConvert.cpp:101-107  cv::cvtColor(.., COLOR_BGR2HSV)
 *                   = OpenCV 3.1.0 imgproc/color.cpp RGB2HSV_b (hrange 180)
 *   oat_bgr2grey    ColorConvert.cpp:101-107 with Color.h:49 (BGR -> GREY / BINARY)
 *                   = RGB2Gray<uchar>, 14-bit fixed-point luma table
 *   oat_grey2bgr    Color.h:47-48 (GREY / BINARY -> BGR) = Gray2RGB<uchar>
 *   oat_hsv2bgr     Color.h:50 (HSV -> BGR) = HSV2RGB_b over HSV2RGB_f, hrange 180
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

/* ---------------------------------------------- the other cvtColor codes ---- */

/* cv::cvtColor(.., COLOR_BGR2GRAY) on CV_8UC3, OpenCV 3.1.0 imgproc/color.cpp
 * RGB2Gray<uchar>: a 3 x 256 table of the running sums b += B2Y, g += G2Y,
 * r += R2Y with R2Y 4899, G2Y 9617, B2Y 1868 (yuv_shift 14), the rounding
 * half (1 << 13) folded into the red column; grey = (tab[b] + tab[g+256] +
 * tab[r+512]) >> 14.  (The IPP route of 3.1.0's cvtColor serves CV_32F only.) */
void oat_bgr2grey(const uint8_t *src, uint8_t *dst, size_t n)
{
    enum { yuv_shift = 14, R2Y = 4899, G2Y = 9617, B2Y = 1868 };
    int tab[768];
    int b = 0, g = 0, r = 1 << (yuv_shift - 1);
    for (int i = 0; i < 256; i++, b += B2Y, g += G2Y, r += R2Y) {
        tab[i] = b;
        tab[i + 256] = g;
        tab[i + 512] = r;
    }
    for (size_t i = 0; i < n; i++, src += 3)
        dst[i] = (uint8_t)((tab[src[0]] + tab[src[1] + 256] + tab[src[2] + 512]) >> yuv_shift);
}

/* cv::cvtColor(.., COLOR_GRAY2BGR): Gray2RGB<uchar>, every channel = grey. */
void oat_grey2bgr(const uint8_t *src, uint8_t *dst, size_t n)
{
    for (size_t i = 0; i < n; i++, dst += 3) dst[0] = dst[1] = dst[2] = src[i];
}

/* cv::cvtColor(.., COLOR_HSV2BGR) on CV_8UC3, OpenCV 3.1.0: HSV2RGB_b turns a
 * pixel into floats (h, s * (1.f/255.f), v * (1.f/255.f)), runs HSV2RGB_f with
 * hscale = 6.f / 180 and stores saturate_cast<uchar>(c * 255.f) (= cvRound,
 * round half to even).  float arithmetic throughout, every operation rounded
 * on its own (-ffp-contract=off).  (OpenCV >= 3.4 replaced this by a fixed-point
 * routine whose results differ by one level here and there; Oat pins 3.1.0.) */
void oat_hsv2bgr(const uint8_t *src, uint8_t *dst, size_t n)
{
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    const float hscale = 6.f / 180.f;
    for (size_t i = 0; i < n; i++, src += 3, dst += 3) {
        float h = src[0], s = src[1] * (1.f / 255.f), v = src[2] * (1.f / 255.f);
        float b, g, r;
        if (s == 0)
            b = g = r = v;
        else {
            float tab[4];
            int sector;
            h *= hscale;
            if (h < 0)
                do h += 6; while (h < 0);
            else if (h >= 6)
                do h -= 6; while (h >= 6);
            sector = (int)floorf(h);
            h -= sector;
            if ((unsigned)sector >= 6u) {
                sector = 0;
                h = 0.f;
            }
            tab[0] = v;
            tab[1] = v * (1.f - s);
            tab[2] = v * (1.f - s * h);
            tab[3] = v * (1.f - s * (1.f - h));
            b = tab[sector_data[sector][0]];
            g = tab[sector_data[sector][1]];
            r = tab[sector_data[sector][2]];
        }
        const float c[3] = {b * 255.f, g * 255.f, r * 255.f};
        for (int k = 0; k < 3; k++) {
            long q = lrintf(c[k]);           /* cvRound: nearest, ties to even */
            dst[k] = (uint8_t)(q < 0 ? 0 : q > 255 ? 255 : q);
        }
    }
}

/* oat::color_conv_table (Color.h:45-51) behind ColorConvert::filter
 * (ColorConvert.cpp:101-107).  Colours are oat::PixelColor values (BINARY 0,
 * GREY 1, BGR 2, HSV 3).  Returns bytes per output pixel, -1 = "nothing to be
 * done" (ColorConvert.cpp:79-85 throws), -2 = "not possible" (Color.h:92-93). */
int oat_cvt_color(int from, int to, const uint8_t *src, uint8_t *dst, size_t n)
{
    if (from < 0 || from > 3 || to < 0 || to > 3) return -2;
    if (from <= 1) {
        if (to <= 1) return -1;
        if (to == 3) return -2;
        oat_grey2bgr(src, dst, n);
        return 3;
    }
    if (from == 2) {
        if (to == 2) return -1;
        if (to == 3) { oat_bgr2hsv(src, dst, n); return 3; }
        oat_bgr2grey(src, dst, n);
        return 1;
    }
    if (to == 3) return -1;
    if (to <= 1) return -2;
    oat_hsv2bgr(src, dst, n);
    return 3;
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
