/* pipeline.c -- the CPU chain over a frame sequence with the reference's stage pipelining (TEST INFRASTRUCTURE, like
 * the rest of oracle/: bench.py's cpu_baseline leg and tests/ only).
 *
 * Oat is one process per component: `framefilt mog`, `framefilt col -C HSV` and `posidet hsv` run CONCURRENTLY on
 * consecutive frames, coupled by one shared-memory slot per sink (lib/shmemdf/Sink.h:93-116, Source.h:187-232) that
 * every consumer copies out of before it works (FrameFilter.cpp:61-80, PositionDetector.cpp:60-86).  A CPU baseline
 * that runs the stages back to back per frame (oat_chain_step) therefore understates the reference's design
 * (VERDICT r03 missing-6): its throughput is 1 / sum(stage), the reference's 1 / max(stage).
 *
 * Here: three stage threads -- front (MOG2 + setTo, its own row workers), middle (BGR2HSV, inRange, erode, dilate, its
 * own row workers), back (findContours + moments + selection, one thread as in OpenCV) -- with TWO frame buffers
 * between neighbours: the sink's slot and the consumer's internal copy, without charging the CPU the two memcpys the
 * reference pays per hand-over.  Results are bit-identical to the sequential chain (tests/test_oracle_golden.py). */
#define _GNU_SOURCE
#include "oat_oracle.h"

#include <pthread.h>
#include <sched.h>
#include <semaphore.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* a few pools kept for reuse: their workers are detached threads that sleep between runs (pool.c never retires them) */
static pthread_mutex_t cache_mu = PTHREAD_MUTEX_INITIALIZER;
static oat_pool *cache[64];
static int n_cached = 0;
static oat_pool *pool_get(void)
{
    oat_pool *p = NULL;
    pthread_mutex_lock(&cache_mu);
    if (n_cached > 0) p = cache[--n_cached];
    pthread_mutex_unlock(&cache_mu);
    return p ? p : oat_pool_create();
}
static void pool_put(oat_pool *p)
{
    pthread_mutex_lock(&cache_mu);
    if (p && n_cached < 64) cache[n_cached++] = p;
    pthread_mutex_unlock(&cache_mu);
}

typedef struct {
    oat_mog2 *m;
    const uint8_t *const *frames;
    int nfile, first, n, rows, cols, channels, t_front, t_mid;
    double lr;
    const oat_hsv_params *p;
    uint8_t *work[2], *thr[2], *mask, *hsv, *tmp;
    sem_t work_free, work_full, thr_free, thr_full;
    double busy[3];
    oat_pool *pool_front, *pool_mid;
} pipe_t;

static void wait_sem(sem_t *s) { while (sem_wait(s) != 0) {} }

static void *front_thread(void *arg)
{
    pipe_t *q = (pipe_t *)arg;
    oat_pool_make_current(q->pool_front);
    for (int i = 0; i < q->n; i++) {
        wait_sem(&q->work_free);
        const double t0 = now_s();
        oat_mog2_filter_from(q->m, q->frames[(q->first + i) % q->nfile], q->work[i & 1], q->mask, q->lr, q->t_front);
        q->busy[0] += now_s() - t0;
        sem_post(&q->work_full);
    }
    return NULL;
}

static void *middle_thread(void *arg)
{
    pipe_t *q = (pipe_t *)arg;
    oat_pool_make_current(q->pool_mid);
    for (int i = 0; i < q->n; i++) {
        wait_sem(&q->work_full);
        wait_sem(&q->thr_free);
        const double t0 = now_s();
        oat_chain_middle(q->work[i & 1], q->channels, q->rows, q->cols, q->p, q->hsv, q->thr[i & 1], q->tmp, q->t_mid);
        q->busy[1] += now_s() - t0;
        sem_post(&q->work_free);
        sem_post(&q->thr_full);
    }
    return NULL;
}

double oat_pipeline_run(oat_mog2 *m, const uint8_t *const *frames, int nfile, int first, int n, int rows, int cols,
                        double learning_rate, const oat_hsv_params *p, int t_front, int t_mid, int pipelined,
                        oat_detection *out, double stage_s[3])
{
    const size_t npx = (size_t)rows * cols;
    const int ch = oat_mog2_channels(m);
    pipe_t q;
    memset(&q, 0, sizeof q);
    q.m = m; q.frames = frames; q.nfile = nfile; q.first = first; q.n = n; q.rows = rows; q.cols = cols; q.channels = ch;
    q.t_front = t_front < 1 ? 1 : t_front; q.t_mid = t_mid < 1 ? 1 : t_mid; q.lr = learning_rate; q.p = p;
    uint8_t *mem = malloc(npx * (size_t)(2 * ch + 2 + 1 + 3 + 1));
    if (!mem) return -1.0;
    q.work[0] = mem; q.work[1] = mem + npx * ch;
    q.thr[0] = mem + 2 * npx * ch; q.thr[1] = q.thr[0] + npx;
    q.mask = q.thr[1] + npx; q.hsv = q.mask + npx; q.tmp = q.hsv + 3 * npx;
    oat_detection scratch_out;
    double t_wall;
    if (!pipelined) {
        const double t0 = now_s();
        for (int i = 0; i < n; i++) {
            double a = now_s();
            oat_mog2_filter_from(m, frames[(first + i) % nfile], q.work[0], q.mask, learning_rate, q.t_front);
            double b = now_s();
            oat_chain_middle(q.work[0], ch, rows, cols, p, q.hsv, q.thr[0], q.tmp, q.t_front);
            double c = now_s();
            oat_sift_contours(q.thr[0], rows, cols, p->min_area, p->max_area, out ? &out[i] : &scratch_out);
            double d = now_s();
            q.busy[0] += b - a; q.busy[1] += c - b; q.busy[2] += d - c;
        }
        t_wall = now_s() - t0;
    } else {
        sem_init(&q.work_free, 0, 2); sem_init(&q.work_full, 0, 0);
        sem_init(&q.thr_free, 0, 2); sem_init(&q.thr_full, 0, 0);
        q.pool_front = pool_get(); q.pool_mid = pool_get();
        pthread_t ta, tb;
        pthread_attr_t at;                       /* the stage threads may run anywhere, like the pools' workers (pool.c) */
        pthread_attr_init(&at);
        {
            cpu_set_t all;
            CPU_ZERO(&all);
            const long ncpu = sysconf(_SC_NPROCESSORS_CONF);
            for (long c = 0; c < ncpu && c < CPU_SETSIZE; c++) CPU_SET((int)c, &all);
            pthread_attr_setaffinity_np(&at, sizeof all, &all);
        }
        const double t0 = now_s();
        pthread_create(&ta, &at, front_thread, &q);
        pthread_create(&tb, &at, middle_thread, &q);
        pthread_attr_destroy(&at);
        for (int i = 0; i < n; i++) {
            wait_sem(&q.thr_full);
            const double a = now_s();
            oat_sift_contours(q.thr[i & 1], rows, cols, p->min_area, p->max_area, out ? &out[i] : &scratch_out);
            q.busy[2] += now_s() - a;
            sem_post(&q.thr_free);
        }
        t_wall = now_s() - t0;
        pthread_join(ta, NULL);
        pthread_join(tb, NULL);
        pool_put(q.pool_front); pool_put(q.pool_mid);
        sem_destroy(&q.work_free); sem_destroy(&q.work_full); sem_destroy(&q.thr_free); sem_destroy(&q.thr_full);
    }
    if (stage_s) { stage_s[0] = q.busy[0]; stage_s[1] = q.busy[1]; stage_s[2] = q.busy[2]; }
    free(mem);
    return t_wall;
}
