/* pool.c -- persistent worker pools for the row-parallel stages of the CPU baseline (TEST INFRASTRUCTURE, like the
 * rest of oracle/: only tests/, __graft_entry__.smoke() and bench.py's parity gates / cpu_baseline leg load it).
 *
 * OpenCV runs MOG2, cvtColor, inRange and the morphology through parallel_for_, whose threads live as long as the
 * process; Oat adds one process per component.  Round 2's port created and joined its threads for every stage of
 * every frame (six rounds a frame), which capped the useful thread count near 32 of the GPU box's 256 hardware
 * threads.  Here the workers are created once; a stage wakes exactly the workers it has jobs
 * for, each through a semaphore of its own (one condition variable for all of them was a thundering herd: every
 * worker ever created woke for every stage), runs job 0 on the calling thread and waits for a completion count.
 * Results do not depend on the number of workers.
 *
 * r04: a pool is an OBJECT.  oat_pool_run() uses the calling thread's current pool (oat_pool_make_current), the
 * process-wide default one otherwise -- so the stage threads of pipeline.c, which stand for Oat's concurrent
 * component processes (framefilt mog | framefilt col + posidet front | findContours), each drive row workers of
 * their own at the same time, as three OpenCV processes would. */
#define _GNU_SOURCE
#include "oat_oracle.h"

#include <pthread.h>
#include <sched.h>
#include <semaphore.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <unistd.h>

#define OAT_POOL_MAX 512

struct oat_pool {
    pthread_mutex_t run_mu;                  /* one oat_pool_run at a time per pool */
    sem_t wake[OAT_POOL_MAX];                /* worker i sleeps on wake[i] */
    sem_t done;
    int n_workers;                           /* workers 1 .. n_workers exist (0 is the caller) */
    atomic_int remaining;
    void *(*cur_fn)(void *);
    char *cur_jobs;
    size_t cur_stride;
    int cpu_offset;                          /* where this pool's workers start in the machine's spread order */
};

typedef struct { struct oat_pool *pool; int id; } worker_arg;

/* cpus of the machine in an order that spreads consecutive workers over the NUMA nodes: node 0's first cpu, node 1's first,
 * ..., node 0's second, ...; read once from /sys/devices/system/node/node<N>/cpulist.  -1 when nothing could be read. */
#include <stdio.h>
static int spread_cpus[4096];
static int n_spread = -1;
static pthread_once_t spread_once = PTHREAD_ONCE_INIT;
static void spread_init(void)
{
    static int per_node[64][1024];
    int cnt[64] = {0}, nodes = 0;
    for (int nd = 0; nd < 64; nd++) {
        char path[96], buf[4096];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", nd);
        FILE *f = fopen(path, "r");
        if (!f) break;
        if (fgets(buf, sizeof buf, f)) {
            for (char *q = buf; *q && *q != '\n';) {
                char *end;
                long lo = strtol(q, &end, 10), hi = lo;
                if (end == q) break;
                if (*end == '-') { q = end + 1; hi = strtol(q, &end, 10); }
                for (long c = lo; c <= hi && cnt[nd] < 1024; c++) per_node[nd][cnt[nd]++] = (int)c;
                q = (*end == ',') ? end + 1 : end;
            }
        }
        fclose(f);
        nodes = nd + 1;
    }
    int n = 0;
    for (int k = 0; k < 1024; k++)
        for (int nd = 0; nd < nodes; nd++)
            if (k < cnt[nd] && n < 4096) spread_cpus[n++] = per_node[nd][k];
    n_spread = n;
}
static int cpu_spread(int i)
{
    pthread_once(&spread_once, spread_init);
    return n_spread > 0 ? spread_cpus[i % n_spread] : -1;
}

static struct oat_pool default_pool = { .run_mu = PTHREAD_MUTEX_INITIALIZER };
static pthread_once_t default_once = PTHREAD_ONCE_INIT;
static void default_init(void) { sem_init(&default_pool.done, 0, 0); }
static __thread struct oat_pool *current_pool = NULL;

static void *pool_worker(void *arg)
{
    worker_arg *wa = (worker_arg *)arg;
    struct oat_pool *p = wa->pool;
    const int id = wa->id;
    free(wa);
    for (;;) {
        while (sem_wait(&p->wake[id]) != 0) {}
        p->cur_fn(p->cur_jobs + (size_t)id * p->cur_stride);   /* (published before the sem_post that woke us) */
        if (atomic_fetch_sub(&p->remaining, 1) == 1) sem_post(&p->done);
    }
    return NULL;
}

int oat_pool_max(void) { return OAT_POOL_MAX; }

/* where the DEFAULT pool's workers start in the machine's spread order: several processes of one job on one host (bench.py
 * --gpus N: every rank runs its own gates) keep their workers beside each other.  Call before the first oat_pool_run. */
void oat_pool_set_cpu_offset(int offset) { default_pool.cpu_offset = offset < 0 ? 0 : offset; }

oat_pool *oat_pool_create(void)
{
    struct oat_pool *p = calloc(1, sizeof *p);
    if (!p) return NULL;
    pthread_mutex_init(&p->run_mu, NULL);
    sem_init(&p->done, 0, 0);
    static atomic_int next_offset = 64;      /* (the default pool starts at 0; concurrent pools -- pipeline.c's stages, several
                                              * streams at once -- sit beside each other, not on top) */
    p->cpu_offset = atomic_fetch_add(&next_offset, 64);
    return p;
}

/* the calling thread's oat_pool_run()s (and so every row-parallel oracle stage it calls) use `p` from now on; NULL =
 * the process-wide default pool.  Pools are never destroyed (their workers are detached and sleep). */
void oat_pool_make_current(oat_pool *p) { current_pool = p; }

/* fn(jobs + i * stride) for i in [0, njobs), in parallel (job 0 on the calling thread); returns when all are done */
void oat_pool_run(void *(*fn)(void *), void *jobs, size_t stride, int njobs)
{
    if (njobs <= 0) return;
    if (njobs == 1) { fn(jobs); return; }
    if (njobs > OAT_POOL_MAX) njobs = OAT_POOL_MAX;            /* (callers cap their job arrays at 512 too) */
    struct oat_pool *p = current_pool;
    if (!p) { pthread_once(&default_once, default_init); p = &default_pool; }
    pthread_mutex_lock(&p->run_mu);
    while (p->n_workers < njobs - 1) {                         /* grow on demand; workers are never retired */
        const int id = p->n_workers + 1;
        pthread_attr_t at;
        pthread_t th;
        worker_arg *wa = malloc(sizeof *wa);
        if (!wa) break;
        wa->pool = p; wa->id = id;
        sem_init(&p->wake[id], 0, 0);
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        {   /* Worker `id` of every pool lives on ONE cpu of the machine's spread order (cpu_spread: NUMA nodes in turn,
             * each node's cpus in the order the kernel lists them -- distinct cores before their SMT siblings), whatever
             * the creating thread is pinned to (bench.py keeps its driving thread on the GPU's NUMA node; the CPU
             * baseline is the whole host's).  A worker always takes the same block of rows of a model, and a model's
             * pages are first touched by the worker that owns them (mog2.c): rows stay in their worker's node's memory.
             * Measured (profiles/r05c): one 4K stream is unchanged by the pinning (105-117 fps unpinned, 105-114 pinned;
             * 128 threads stay slower than 32 either way -- the serial contour pass and the per-stage wake-ups bound it),
             * eight 1080p streams at once gain (279-300 -> 377 fps): their pools no longer share cores. */
            cpu_set_t one;
            CPU_ZERO(&one);
            const int cpu = cpu_spread(id + p->cpu_offset);
            if (cpu >= 0) {
                CPU_SET(cpu, &one);
            } else {
                const long ncpu = sysconf(_SC_NPROCESSORS_CONF);
                for (long c = 0; c < ncpu && c < CPU_SETSIZE; c++) CPU_SET((int)c, &one);
            }
            pthread_attr_setaffinity_np(&at, sizeof one, &one);
        }
        int rc = pthread_create(&th, &at, pool_worker, wa);
        if (rc != 0) {                                         /* (that cpu is not ours -- a cpuset: any cpu will do) */
            cpu_set_t all;
            CPU_ZERO(&all);
            const long ncpu = sysconf(_SC_NPROCESSORS_CONF);
            for (long c = 0; c < ncpu && c < CPU_SETSIZE; c++) CPU_SET((int)c, &all);
            pthread_attr_setaffinity_np(&at, sizeof all, &all);
            rc = pthread_create(&th, &at, pool_worker, wa);
        }
        pthread_attr_destroy(&at);
        if (rc != 0) { free(wa); break; }
        p->n_workers = id;
    }
    const int par = p->n_workers + 1 < njobs ? p->n_workers + 1 : njobs;     /* jobs that get a thread of their own */
    p->cur_fn = fn; p->cur_jobs = (char *)jobs; p->cur_stride = stride;
    atomic_store(&p->remaining, par - 1);
    for (int i = 1; i < par; i++) sem_post(&p->wake[i]);
    fn(jobs);
    for (int i = par; i < njobs; i++) fn((char *)jobs + (size_t)i * stride);   /* (threads could not be created) */
    if (par > 1) while (sem_wait(&p->done) != 0) {}
    pthread_mutex_unlock(&p->run_mu);
}
