// bench_native.hip -- the pipelined track loop driven from a plain C++ process over the C ABI
// (no Python, no torch): what a native caller of liboatgpu.so sees per step.
//   hipcc --offload-arch=gfx950 -O2 -w tools/bench_native.hip -Iinclude -Loat_amd/lib -loatgpu \
//         -Wl,-rpath,'$ORIGIN/../../oat_amd/lib' -o build/bin/bench_native
//   build/bin/bench_native ROWS COLS STREAMS STEPS [ERODE DILATE [CONTEXTS]]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "oatgpu.h"

// textured background (static per pixel + small per-frame noise) and one blue disc per stream
// that moves fast enough never to be absorbed into the MOG2 background
__global__ void k_synth(unsigned char *out, int rows, int cols, int frame, int ns)
{
    const long long n = (long long)rows * cols * ns;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(p / ((long long)rows * cols));
        const int q = (int)(p - (long long)s * rows * cols);
        const int y = q / cols, x = q - y * cols;
        unsigned h = (unsigned)q * 2654435761u + (unsigned)s * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        unsigned n2 = ((unsigned)q + 977u * (unsigned)frame) * 3266489917u;
        n2 ^= n2 >> 16;
        int b = 60 + (int)(h & 63) + (int)(n2 & 3), g = 70 + (int)((h >> 8) & 63) + (int)((n2 >> 4) & 3),
            r = 80 + (int)((h >> 16) & 63) + (int)((n2 >> 8) & 3);
        const int R = rows / 18;
        const int cx = (int)((cols / 2) + (cols / 3) * __cosf(0.37f * frame + s)),
                  cy = (int)((rows / 2) + (rows / 3) * __sinf(0.53f * frame + 2 * s));
        if ((x - cx) * (x - cx) + (y - cy) * (y - cy) <= R * R) { b = 220; g = 120; r = 30; }
        out[p * 3] = (unsigned char)b; out[p * 3 + 1] = (unsigned char)g; out[p * 3 + 2] = (unsigned char)r;
    }
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s ROWS COLS STREAMS STEPS [ERODE DILATE]\n", argv[0]); return 2; }
    const int rows = atoi(argv[1]), cols = atoi(argv[2]), ns = atoi(argv[3]), steps = atoi(argv[4]);
    oatgpu_config cfg;
    oatgpu_default_config(&cfg);
    cfg.rows = rows; cfg.cols = cols; cfg.n_streams = ns; cfg.ring_depth = 8;
    cfg.h_lo = 100; cfg.h_hi = 125; cfg.s_lo = 150; cfg.s_hi = 256; cfg.v_lo = 100; cfg.v_hi = 256;
    cfg.erode = argc > 5 ? atoi(argv[5]) : 3;
    cfg.dilate = argc > 6 ? atoi(argv[6]) : 7;
    cfg.min_area = 20.0; cfg.max_area = 1e7;
    const int nctx = argc > 7 ? atoi(argv[7]) : 1;     // contexts driven round-robin from this thread
    std::vector<oatgpu_ctx *> ctxs;
    for (int i = 0; i < nctx; ++i) {
        oatgpu_ctx *ci = oatgpu_create(&cfg);
        if (!ci) { fprintf(stderr, "create: %s\n", oatgpu_last_error(nullptr)); return 1; }
        ctxs.push_back(ci);
    }
    oatgpu_ctx *c = ctxs[0];

    const int pool_n = 48;
    const size_t fbytes = (size_t)ns * rows * cols * 3;
    std::vector<unsigned char *> pool(pool_n);
    for (int f = 0; f < pool_n; ++f) {
        if (hipMalloc(&pool[f], fbytes) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
        hipLaunchKernelGGL(k_synth, dim3(4096), dim3(256), 0, 0, pool[f], rows, cols, f, ns);
    }
    hipDeviceSynchronize();

    std::vector<oatgpu_position> pos(ns);
    const double lr = 0.01;
    long found = 0;
    double t_enqueue = 0, t_collect = 0;       // host time inside the two calls
    auto loop = [&](int n, bool count) {
        int rc = 0;
        for (int i = 0; i < n && !rc; ++i) {
            for (oatgpu_ctx *ci : ctxs) {
                if (oatgpu_track_outstanding(ci) == cfg.ring_depth) {
                    const double x = now_us();
                    rc = oatgpu_track_collect(ci, pos.data());
                    t_collect += now_us() - x;
                    if (count && ci == c) for (auto &p : pos) found += p.valid;
                }
                const double x = now_us();
                if (!rc) rc = oatgpu_track_enqueue_dev(ci, pool[i % pool_n], lr);
                t_enqueue += now_us() - x;
            }
        }
        for (oatgpu_ctx *ci : ctxs)
            while (!rc && oatgpu_track_outstanding(ci)) {
                rc = oatgpu_track_collect(ci, pos.data());
                if (count && ci == c) for (auto &p : pos) found += p.valid;
            }
        if (rc) { fprintf(stderr, "track: %s\n", oatgpu_last_error(c)); exit(1); }
    };
    loop(200, false);
    for (oatgpu_ctx *ci : ctxs) oatgpu_synchronize(ci);
    oatgpu_profile_enable(c, 8);
    t_enqueue = t_collect = 0;
    const double t0 = now_us();
    if (getenv("BENCH_SEQUENCE")) {        // the same loop inside the library
        std::vector<const void *> seq(steps);
        for (int i = 0; i < steps; ++i) seq[i] = pool[i % pool_n];
        std::vector<oatgpu_position> all((size_t)steps * ns);
        if (oatgpu_track_sequence_dev(c, seq.data(), steps, lr, all.data())) { fprintf(stderr, "%s\n", oatgpu_last_error(c)); return 1; }
        for (auto &p : all) found += p.valid;
    } else {
        loop(steps, true);
    }
    for (oatgpu_ctx *ci : ctxs) oatgpu_synchronize(ci);
    const double t1 = now_us();
    oatgpu_profile pr;
    oatgpu_profile_read(c, &pr);
    const double us = (t1 - t0) / steps;
    printf("{\"rows\": %d, \"cols\": %d, \"streams\": %d, \"steps\": %d, \"us_per_step\": %.2f, \"fps\": %.1f, "
           "\"found\": %ld, \"expected\": %ld, \"k1_us\": %.2f, \"blob_us\": %.2f, \"host_enqueue_us\": %.2f, \"host_collect_us\": %.2f}\n",
           rows, cols, ns * nctx, steps, us, ns * nctx * 1e6 / us, found, (long)steps * ns,
           pr.steps ? 1e3 * (pr.mog_ms / pr.steps - pr.event_pair_ms) : 0.0, pr.steps ? 1e3 * pr.blob_ms / pr.steps : 0.0,
           t_enqueue / steps, t_collect / steps);
    for (oatgpu_ctx *ci : ctxs) oatgpu_destroy(ci);
    return 0;
}
