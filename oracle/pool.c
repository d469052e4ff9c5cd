/* pool.c -- a persistent worker pool for the row-parallel stages of the CPU baseline (TEST INFRASTRUCTURE, like the
 * rest of oracle/: only tests/, __graft_entry__.smoke() and bench.py's parity gates / cpu_baseline leg load it).
 *
 * OpenCV runs MOG2, cvtColor, inRange and the morphology through parallel_for_, whose threads live as long as the
 * process; Oat adds one process per component.  Round 2's port created and joined its threads for every stage of
 * every frame (six rounds a frame), which capped the useful thread count near 32 of the GPU box's 256 hardware
 * threads.  Here the workers are created once; a stage wakes exactly the workers it has jobs
 * for, each through a semaphore of its own (one condition variable for all of them was a thundering herd: every
 * worker ever created woke for every stage), runs job 0 on the calling thread and waits for a completion count.
 * Results do not depend on the number of workers. */
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

static pthread_mutex_t run_mu = PTHREAD_MUTEX_INITIALIZER;     /* one oat_pool_run at a time */
static sem_t wake[OAT_POOL_MAX];                               /* worker i sleeps on wake[i] */
static sem_t done;
static int n_workers = 0;                                      /* workers 1 .. n_workers exist (0 is the caller) */
static int done_ready = 0;
static atomic_int remaining;
static void *(*cur_fn)(void *);
static char *cur_jobs;
static size_t cur_stride;

static void *pool_worker(void *arg)
{
    const int id = (int)(intptr_t)arg;
    for (;;) {
        while (sem_wait(&wake[id]) != 0) {}
        cur_fn(cur_jobs + (size_t)id * cur_stride);            /* (published before the sem_post that woke us) */
        if (atomic_fetch_sub(&remaining, 1) == 1) sem_post(&done);
    }
    return NULL;
}

int oat_pool_max(void) { return OAT_POOL_MAX; }

/* fn(jobs + i * stride) for i in [0, njobs), in parallel (job 0 on the calling thread); returns when all are done */
void oat_pool_run(void *(*fn)(void *), void *jobs, size_t stride, int njobs)
{
    if (njobs <= 0) return;
    if (njobs == 1) { fn(jobs); return; }
    if (njobs > OAT_POOL_MAX) njobs = OAT_POOL_MAX;            /* (callers cap their job arrays at 512 too) */
    pthread_mutex_lock(&run_mu);
    if (!done_ready) { sem_init(&done, 0, 0); done_ready = 1; }
    while (n_workers < njobs - 1) {                            /* grow on demand; workers are never retired */
        const int id = n_workers + 1;
        pthread_attr_t at;
        pthread_t th;
        sem_init(&wake[id], 0, 0);
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        {   /* the workers may run on every CPU of the machine, whatever the creating thread is pinned to (bench.py
             * keeps its driving thread on the GPU's NUMA node; the CPU baseline is the whole host's) */
            cpu_set_t all;
            CPU_ZERO(&all);
            const long ncpu = sysconf(_SC_NPROCESSORS_CONF);
            for (long c = 0; c < ncpu && c < CPU_SETSIZE; c++) CPU_SET((int)c, &all);
            pthread_attr_setaffinity_np(&at, sizeof all, &all);
        }
        const int rc = pthread_create(&th, &at, pool_worker, (void *)(intptr_t)id);
        pthread_attr_destroy(&at);
        if (rc != 0) break;
        n_workers = id;
    }
    const int par = n_workers + 1 < njobs ? n_workers + 1 : njobs;     /* jobs that get a thread of their own */
    cur_fn = fn; cur_jobs = (char *)jobs; cur_stride = stride;
    atomic_store(&remaining, par - 1);
    for (int i = 1; i < par; i++) sem_post(&wake[i]);
    fn(jobs);
    for (int i = par; i < njobs; i++) fn((char *)jobs + (size_t)i * stride);   /* (threads could not be created) */
    if (par > 1) while (sem_wait(&done) != 0) {}
    pthread_mutex_unlock(&run_mu);
}
