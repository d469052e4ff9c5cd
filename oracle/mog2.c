/*
 * mog2.c -- ORACLE (test infrastructure): Zivkovic Gaussian-mixture background
 * subtraction exactly as cv::BackgroundSubtractorMOG2 (OpenCV 3.1.0,
 * modules/video/src/bgfg_gaussmix2.cpp: BackgroundSubtractorMOG2Impl::apply,
 * MOG2Invoker::operator(), detectShadowGMM) computes it on the CPU.
 *
 * Stands in for:  BackgroundSubtractorMOG.cpp:82-83 (creation, all defaults)
 *                 BackgroundSubtractorMOG.cpp:124   (apply)
 *                 BackgroundSubtractorMOG.cpp:125   (frame.setTo(0, mask == 0))
 *
 * OpenCV is not in /root/reference; the algorithm is restated from the
 * published source.  PARITY UNPINNED by the reference.
 *
 * Mode count.  MOG2Invoker keeps TWO counters, `int nmodes = modesUsed[x], nNewModes = nmodes;`:
 * the prune test decrements nmodes (which also is the bound of the running mode loop and of the
 * renormalisation loop), and right behind the renormalisation comes `nmodes = nNewModes;` -- so a
 * pruned mode keeps its slot with weight 0 (it can still be matched, and is then revived with
 * weight alphaT - alphaT*CT), modesUsed[x] never decreases, and a new mode replaces the LAST slot
 * once all nmixtures slots have been used.  That is how the 2.4.x, 3.x and 4.x sources read as far
 * as they can be recalled without a copy at hand (no OpenCV in this image nor on the GPU box:
 * profiles/r02_opencv_probe.txt), and it is the default here (restore_nmodes = 1).  Round 1 had
 * restated the loop without the second counter; that reading is kept as restore_nmodes = 0 so
 * that whoever holds the 3.1.0 source can flip one switch (oracle, C ABI and kernel all take it)
 * instead of re-deriving anything.  The two readings differ only once a weight has decayed below
 * 2*alphaT*CT/(1-alphaT), i.e. never at Oat's default learning rate 0.
 *
 * Build with -ffp-contract=off: the x86-64 baseline OpenCV build has no FMA,
 * so every multiply and add below rounds separately, in this order.
 */
#include "oat_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float weight; float variance; } gmm_t;   /* OpenCV's struct GMM */

struct oat_mog2 {
    int rows, cols, ch;
    oat_mog2_params p;
    int nframes;
    gmm_t *gmm;          /* rows*cols*nmixtures, AoS like OpenCV's bgmodel      */
    float *mean;         /* rows*cols*nmixtures*ch                              */
    uint8_t *modes_used; /* rows*cols                                           */
    int pristine;        /* arrays still as calloc left them: all zero, pages not yet touched */
};

void oat_mog2_default_params(oat_mog2_params *p)
{
    /* bgfg_gaussmix2.cpp defaults: defaultHistory2=500, defaultVarThreshold2=16,
     * defaultNMixtures2=5, defaultBackgroundRatio2=0.9, defaultVarThresholdGen2=9,
     * defaultVarInit2=15, defaultVarMax2=5*15, defaultVarMin2=4, defaultfCT2=0.05,
     * defaultnShadowDetection2=127, defaultfTau=0.5; createBackgroundSubtractorMOG2
     * default detectShadows=true. */
    p->history = 500;
    p->nmixtures = 5;
    p->var_threshold = 16.0f;
    p->background_ratio = 0.9f;
    p->var_threshold_gen = 3.0f * 3.0f;
    p->var_init = 15.0f;
    p->var_min = 4.0f;
    p->var_max = 5.0f * 15.0f;
    p->ct = 0.05f;
    p->detect_shadows = 1;
    p->shadow_value = 127;
    p->tau = 0.5f;
    p->restore_nmodes = 1;
}

oat_mog2 *oat_mog2_create(int rows, int cols, int channels, const oat_mog2_params *p)
{
    oat_mog2 *m = (oat_mog2 *)calloc(1, sizeof(*m));
    if (!m) return NULL;
    m->rows = rows; m->cols = cols; m->ch = channels;
    if (p) m->p = *p; else oat_mog2_default_params(&m->p);
    size_t n = (size_t)rows * cols;
    m->gmm = (gmm_t *)calloc(n * m->p.nmixtures, sizeof(gmm_t));
    m->mean = (float *)calloc(n * m->p.nmixtures * channels, sizeof(float));
    m->modes_used = (uint8_t *)calloc(n, 1);
    m->nframes = 0;
    m->pristine = 1;
    return m;
}

void oat_mog2_destroy(oat_mog2 *m)
{
    if (!m) return;
    free(m->gmm); free(m->mean); free(m->modes_used); free(m);
}

int oat_mog2_nframes(const oat_mog2 *m) { return m->nframes; }
int oat_mog2_channels(const oat_mog2 *m) { return m->ch; }
const uint8_t *oat_mog2_modes_used(const oat_mog2 *m) { return m->modes_used; }

void oat_mog2_get_state(const oat_mog2 *m, float *weight, float *variance, float *mean)
{
    size_t n = (size_t)m->rows * m->cols * m->p.nmixtures;
    for (size_t i = 0; i < n; i++) {
        if (weight) weight[i] = m->gmm[i].weight;
        if (variance) variance[i] = m->gmm[i].variance;
    }
    if (mean) memcpy(mean, m->mean, n * m->ch * sizeof(float));
}

/* Test plumbing (no OpenCV counterpart): continue from a model that was exported elsewhere -- bench.py ages a
 * model on the GPU, hands it over and lets the oracle check the frames that follow. */
void oat_mog2_set_state(oat_mog2 *m, const uint8_t *modes_used, const float *weight, const float *variance,
                        const float *mean, int nframes)
{
    size_t npx = (size_t)m->rows * m->cols, n = npx * m->p.nmixtures;
    memcpy(m->modes_used, modes_used, npx);
    for (size_t i = 0; i < n; i++) {
        m->gmm[i].weight = weight[i];
        m->gmm[i].variance = variance[i];
    }
    memcpy(m->mean, mean, n * m->ch * sizeof(float));
    m->nframes = nframes;
    m->pristine = 0;
}

/* detectShadowGMM (bgfg_gaussmix2.cpp) */
static int detect_shadow_gmm(const float *data, int nchannels, int nmodes,
                             const gmm_t *gmm, const float *mean,
                             float Tb, float TB, float tau)
{
    float tWeight = 0;
    for (int mode = 0; mode < nmodes; mode++, mean += nchannels) {
        gmm_t g = gmm[mode];
        float numerator = 0.0f;
        float denominator = 0.0f;
        for (int c = 0; c < nchannels; c++) {
            numerator += data[c] * mean[c];
            denominator += mean[c] * mean[c];
        }
        /* no division by zero allowed */
        if (denominator == 0)
            return 0;
        /* if tau < a < 1 then also check the colour distortion */
        if (numerator <= denominator && numerator >= tau * denominator) {
            float a = numerator / denominator;
            float dist2a = 0.0f;
            for (int c = 0; c < nchannels; c++) {
                float dD = a * mean[c] - data[c];
                dist2a += dD * dD;
            }
            if (dist2a < Tb * g.variance * a * a)
                return 1;
        }
        tWeight += g.weight;
        if (tWeight > TB)
            return 0;
    }
    return 0;
}

/* MOG2Invoker::operator()(Range) for rows [y0,y1) */
static void mog2_rows(oat_mog2 *m, const uint8_t *image, uint8_t *maskimg,
                      float alphaT, float prune, int y0, int y1)
{
    const int ncols = m->cols, nchannels = m->ch, nmixtures = m->p.nmixtures;
    const float Tb = m->p.var_threshold, TB = m->p.background_ratio, Tg = m->p.var_threshold_gen;
    const float varInit = m->p.var_init, varMin = m->p.var_min, varMax = m->p.var_max;
    const float tau = m->p.tau;
    const int detectShadows = m->p.detect_shadows;
    const uint8_t shadowVal = m->p.shadow_value;
    const float alpha1 = 1.f - alphaT;
    float dData[4];
    float px[4];

    for (int y = y0; y < y1; y++) {
        const uint8_t *src = image + (size_t)y * ncols * nchannels;
        float *mean = m->mean + (size_t)ncols * nmixtures * nchannels * y;
        gmm_t *gmm = m->gmm + (size_t)ncols * nmixtures * y;
        uint8_t *modesUsed = m->modes_used + (size_t)ncols * y;
        uint8_t *mask = maskimg + (size_t)ncols * y;

        for (int x = 0; x < ncols; x++, src += nchannels, gmm += nmixtures, mean += nmixtures * nchannels) {
            /* src->row(y).convertTo(CV_32F) */
            for (int c = 0; c < nchannels; c++) px[c] = (float)src[c];
            const float *data = px;

            int background = 0;
            int fitsPDF = 0;
            int nmodes = modesUsed[x], nNewModes = nmodes;
            float totalWeight = 0.f;
            float *mean_m = mean;

            for (int mode = 0; mode < nmodes; mode++, mean_m += nchannels) {
                float weight = alpha1 * gmm[mode].weight + prune;
                int swap_count = 0;
                if (!fitsPDF) {
                    float var = gmm[mode].variance;
                    float dist2;
                    if (nchannels == 3) {
                        dData[0] = mean_m[0] - data[0];
                        dData[1] = mean_m[1] - data[1];
                        dData[2] = mean_m[2] - data[2];
                        dist2 = dData[0] * dData[0] + dData[1] * dData[1] + dData[2] * dData[2];
                    } else {
                        dist2 = 0.f;
                        for (int c = 0; c < nchannels; c++) {
                            dData[c] = mean_m[c] - data[c];
                            dist2 += dData[c] * dData[c];
                        }
                    }
                    /* background? - Tb - usually larger than Tg */
                    if (totalWeight < TB && dist2 < Tb * var)
                        background = 1;
                    /* check fit */
                    if (dist2 < Tg * var) {
                        fitsPDF = 1;
                        weight += alphaT;
                        float k = alphaT / weight;
                        for (int c = 0; c < nchannels; c++)
                            mean_m[c] -= k * dData[c];
                        float varnew = var + k * (dist2 - var);
                        /* OpenCV's macros, operand order kept: MAX(a,b) ((a) < (b) ? (b) : (a)), MIN(a,b) ((a) > (b) ? (b) : (a)).
                           A NaN varnew (k = 0/0: a pruned slot re-matched at learning rate 0) compares false and STAYS NaN.
                           [OCV-mem] recalled, not read; rounds 1-4 had the comparisons the other way round, which clamps a NaN
                           to varMin.  Unreachable from stock Oat (one fixed rate a process). */
                        varnew = varnew < varMin ? varMin : varnew;   /* MAX(varnew,varMin) */
                        varnew = varnew > varMax ? varMax : varnew;   /* MIN(varnew,varMax) */
                        gmm[mode].variance = varnew;
                        /* sort: only the matched mode may have to move up */
                        for (int i = mode; i > 0; i--) {
                            if (weight < gmm[i - 1].weight)
                                break;
                            swap_count++;
                            gmm_t tg = gmm[i]; gmm[i] = gmm[i - 1]; gmm[i - 1] = tg;
                            for (int c = 0; c < nchannels; c++) {
                                float t = mean[i * nchannels + c];
                                mean[i * nchannels + c] = mean[(i - 1) * nchannels + c];
                                mean[(i - 1) * nchannels + c] = t;
                            }
                        }
                    }
                }
                /* check prune */
                if (weight < -prune) {
                    weight = 0.0f;
                    nmodes--;
                }
                gmm[mode - swap_count].weight = weight;
                totalWeight += weight;
            }

            /* renormalise weights */
            totalWeight = 1.f / totalWeight;
            for (int mode = 0; mode < nmodes; mode++)
                gmm[mode].weight *= totalWeight;

            if (m->p.restore_nmodes)
                nmodes = nNewModes;

            /* make a new mode if needed */
            if (!fitsPDF && alphaT > 0.f) {
                int mode = nmodes == nmixtures ? nmixtures - 1 : nmodes++;
                if (nmodes == 1)
                    gmm[mode].weight = 1.f;
                else {
                    gmm[mode].weight = alphaT;
                    for (int i = 0; i < nmodes - 1; i++)
                        gmm[i].weight *= alpha1;
                }
                for (int c = 0; c < nchannels; c++)
                    mean[mode * nchannels + c] = data[c];
                gmm[mode].variance = varInit;
                for (int i = nmodes - 1; i > 0; i--) {
                    if (alphaT < gmm[i - 1].weight)
                        break;
                    gmm_t tg = gmm[i]; gmm[i] = gmm[i - 1]; gmm[i - 1] = tg;
                    for (int c = 0; c < nchannels; c++) {
                        float t = mean[i * nchannels + c];
                        mean[i * nchannels + c] = mean[(i - 1) * nchannels + c];
                        mean[(i - 1) * nchannels + c] = t;
                    }
                }
            }

            modesUsed[x] = (uint8_t)nmodes;
            mask[x] = background ? 0 :
                (detectShadows && detect_shadow_gmm(data, nchannels, nmodes, gmm, mean, Tb, TB, tau)) ?
                shadowVal : 255;
        }
    }
}

/* BackgroundSubtractorMOG2Impl::apply prologue: (re)initialise, ++nframes,
 * effective learning rate; returns alphaT / prune. */
static void mog2_begin(oat_mog2 *m, double learningRate, float *alphaT, float *prune)
{
    int needToInitialize = m->nframes == 0 || learningRate >= 1;
    if (needToInitialize) {
        /* (a model fresh from calloc is zero already; leaving its pages untouched lets the row workers of the first
         * frame fault them in on their own NUMA nodes -- what a tuned CPU implementation would arrange) */
        if (!m->pristine) {
            size_t n = (size_t)m->rows * m->cols;
            memset(m->gmm, 0, n * m->p.nmixtures * sizeof(gmm_t));
            memset(m->mean, 0, n * m->p.nmixtures * m->ch * sizeof(float));
            memset(m->modes_used, 0, n);
        }
        m->nframes = 0;
    }
    m->pristine = 0;
    ++m->nframes;
    int lim = 2 * m->nframes < m->p.history ? 2 * m->nframes : m->p.history;
    learningRate = (learningRate >= 0 && m->nframes > 1) ? learningRate : 1. / lim;
    *alphaT = (float)learningRate;
    *prune = (float)(-learningRate * m->p.ct);   /* product formed in double */
}

void oat_mog2_apply(oat_mog2 *m, const uint8_t *image, uint8_t *mask, double learningRate)
{
    float alphaT, prune;
    mog2_begin(m, learningRate, &alphaT, &prune);
    mog2_rows(m, image, mask, alphaT, prune, 0, m->rows);
}

static void set_to_zero_where_mask0(uint8_t *frame, const uint8_t *mask, size_t n, int ch)
{
    for (size_t i = 0; i < n; i++)
        if (mask[i] == 0)
            for (int c = 0; c < ch; c++) frame[i * ch + c] = 0;
}

void oat_mog2_filter(oat_mog2 *m, uint8_t *frame, uint8_t *mask, double learningRate)
{
    oat_mog2_apply(m, frame, mask, learningRate);
    set_to_zero_where_mask0(frame, mask, (size_t)m->rows * m->cols, m->ch);
}

typedef struct {
    oat_mog2 *m; uint8_t *frame; uint8_t *mask; float alphaT, prune; int y0, y1;
    const uint8_t *copy_from;      /* not NULL: the caller's (read-only) frame; the worker copies its rows into `frame` first */
} mog2_job;

static void *mog2_worker(void *arg)
{
    mog2_job *j = (mog2_job *)arg;
    if (j->copy_from) {
        size_t o = (size_t)j->y0 * j->m->cols * j->m->ch, nb = (size_t)(j->y1 - j->y0) * j->m->cols * j->m->ch;
        memcpy(j->frame + o, j->copy_from + o, nb);
    }
    mog2_rows(j->m, j->frame, j->mask, j->alphaT, j->prune, j->y0, j->y1);
    size_t off = (size_t)j->y0 * j->m->cols;
    set_to_zero_where_mask0(j->frame + off * j->m->ch, j->mask + off,
                            (size_t)(j->y1 - j->y0) * j->m->cols, j->m->ch);
    return NULL;
}

void oat_mog2_filter_mt(oat_mog2 *m, uint8_t *frame, uint8_t *mask, double learningRate, int nthreads)
{
    oat_mog2_filter_from(m, NULL, frame, mask, learningRate, nthreads);
}

/* The same with the input left untouched: `src` (may be NULL = filter `frame` in place) is copied into `frame` row block
 * by row block inside the workers (FrameFilter::process copies the shared frame before it filters it, FrameFilter.cpp:73-80) */
void oat_mog2_filter_from(oat_mog2 *m, const uint8_t *src, uint8_t *frame, uint8_t *mask, double learningRate, int nthreads)
{
    if (nthreads <= 1) {
        if (src) memcpy(frame, src, (size_t)m->rows * m->cols * m->ch);
        oat_mog2_filter(m, frame, mask, learningRate);
        return;
    }
    if (nthreads > 512) nthreads = 512;
    if (nthreads > m->rows) nthreads = m->rows;
    float alphaT, prune;
    mog2_begin(m, learningRate, &alphaT, &prune);
    mog2_job jobs[512];
    int rows = m->rows;
    for (int t = 0; t < nthreads; t++) {
        jobs[t].m = m; jobs[t].frame = frame; jobs[t].mask = mask;
        jobs[t].alphaT = alphaT; jobs[t].prune = prune; jobs[t].copy_from = src;
        jobs[t].y0 = (int)((long)rows * t / nthreads);
        jobs[t].y1 = (int)((long)rows * (t + 1) / nthreads);
    }
    oat_pool_run(mog2_worker, jobs, sizeof jobs[0], nthreads);     /* persistent workers (pool.c) */
}
