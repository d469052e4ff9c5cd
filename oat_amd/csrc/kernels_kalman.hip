// kernels_kalman.hip -- `oat posifilt kalman` on the batch of positions the blob stage just produced.
//
// Reference: oat::KalmanFilter2D::filter / initializeFilter / initializeStaticMatracies
// (src/positionfilter/KalmanFilter2D.cpp:95-210) over cv::KalmanFilter(4, 2, 0, CV_64F)
// (OpenCV 3.1 modules/video/src/kalman.cpp, [OCV-mem]).  One lane per camera stream, all fp64,
// operation order identical to oracle/kalman.c (built with -ffp-contract=off like the rest).
// The reference's observable quirks are kept: 6.0-filled report before the first track, the
// every-sample timeout test (default --timeout 0 never tracks), correction with the stale
// measurement on samples without a detection, predicted (not corrected) state reported.
//
// Ordering: consecutive frames of a stream run their back halves on different HIP streams, so the
// filter update takes a per-stream ticket: the launch of frame n waits until the state's ticket
// equals n.  State and ticket are only touched with agent-scope atomics (coherent across XCDs
// without cache write-backs); the ticket is bumped after the state stores have drained.
#include "oatgpu_internal.h"

namespace oatgpu {

namespace {

__device__ __forceinline__ double ld(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// d (n x m) = a (n x 4) * b [+ c]; b is (4 x m), or (m x 4) when transposed; k runs 0..3 in order
__device__ void mul4(const double *a, int n, const double *b, int m, bool b_transposed, const double *c, double *d)
{
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int q = 0; q < 4; ++q) s += a[i * 4 + q] * (b_transposed ? b[j * 4 + q] : b[q * m + j]);
            d[i * m + j] = c ? s + c[i * m + j] : s;
        }
}

struct Filter {     // registers / scratch copy of one stream's KalmanState
    double statePre[4], statePost[4], Ppre[16], Ppost[16], meas[2], reported[4];
    int found, missing, aliased;
};

__device__ void static_matrices(const KalmanLaunch &k, double *A, double *Q, double *R)
{
    const double dt = k.dt, sa = k.sig_accel;
    for (int i = 0; i < 16; ++i) { A[i] = 0.0; Q[i] = 0.0; }
    for (int i = 0; i < 4; ++i) A[i * 5] = 1.0;
    A[0 * 4 + 1] = dt;
    A[2 * 4 + 3] = dt;
    Q[0 * 4 + 0] = sa * sa * (dt * dt * dt * dt) / 4.0;
    Q[0 * 4 + 1] = sa * sa * (dt * dt * dt) / 2.0;
    Q[1 * 4 + 0] = sa * sa * (dt * dt * dt) / 2.0;
    Q[1 * 4 + 1] = sa * sa * (dt * dt);
    Q[2 * 4 + 2] = sa * sa * (dt * dt * dt * dt) / 4.0;
    Q[2 * 4 + 3] = sa * sa * (dt * dt * dt) / 2.0;
    Q[3 * 4 + 2] = sa * sa * (dt * dt * dt) / 2.0;
    Q[3 * 4 + 3] = sa * sa * (dt * dt);
    R[0] = R[3] = k.sig_noise * k.sig_noise;
    R[1] = R[2] = 0.0;
}

// KalmanFilter2D::filter, KalmanFilter2D.cpp:95-141
__device__ void filter_step(Filter &f, const KalmanLaunch &k, bool valid, double x, double y)
{
    const double H[8] = {1, 0, 0, 0, 0, 0, 1, 0};
    if (valid) {
        f.meas[0] = x; f.meas[1] = y;
        f.missing = 0;
        if (!f.found) {                                         // initializeFilter, :143-164
            for (int i = 0; i < 16; ++i) f.Ppre[i] = 0.0;
            for (int i = 0; i < 4; ++i) f.Ppre[i * 5] = 1000.0;
            f.statePre[0] = x; f.statePre[1] = 0.0; f.statePre[2] = y; f.statePre[3] = 0.0;
            for (int i = 0; i < 4; ++i) f.statePost[i] = f.statePre[i];
        }
        f.found = 1;
    } else {
        f.missing++;
    }
    if (f.missing >= k.threshold) f.found = 0;
    if (!f.found) return;

    double A[16], Q[16], R[4], t1[16], t2[8], t3[4], t4[8], t5[2], hx[2];
    static_matrices(k, A, Q, R);
    // cv::KalmanFilter::predict
    mul4(A, 4, f.statePost, 1, false, nullptr, f.statePre);
    mul4(A, 4, f.Ppost, 4, false, nullptr, t1);
    mul4(t1, 4, A, 4, true, Q, f.Ppre);
    for (int i = 0; i < 4; ++i) f.statePost[i] = f.statePre[i];
    for (int i = 0; i < 16; ++i) f.Ppost[i] = f.Ppre[i];
    f.aliased = 1;
    // cv::KalmanFilter::correct (2x2 system in closed form, see oracle/kalman.c)
    mul4(H, 2, f.Ppre, 4, false, nullptr, t2);
    mul4(t2, 2, H, 2, true, R, t3);
    const double det = t3[0] * t3[3] - t3[1] * t3[2];
    for (int j = 0; j < 4; ++j) {
        t4[j] = (t3[3] * t2[j] - t3[1] * t2[4 + j]) / det;
        t4[4 + j] = (t3[0] * t2[4 + j] - t3[2] * t2[j]) / det;
    }
    mul4(H, 2, f.statePre, 1, false, nullptr, hx);
    t5[0] = f.meas[0] - hx[0];
    t5[1] = f.meas[1] - hx[1];
    for (int i = 0; i < 4; ++i) {
        const double g0 = t4[i], g1 = t4[4 + i];
        f.statePost[i] = f.statePre[i] + (g0 * t5[0] + g1 * t5[1]);
        for (int j = 0; j < 4; ++j) f.Ppost[i * 4 + j] = f.Ppre[i * 4 + j] - (g0 * t2[j] + g1 * t2[4 + j]);
    }
}

}  // namespace

__global__ __launch_bounds__(64) void k_kalman(KalmanLaunch k, ResultRec *results, int n_streams)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    KalmanState *ks = k.state + s;
    while (__hip_atomic_load(&ks->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != k.ticket)
        __builtin_amdgcn_s_sleep(4);

    Filter f;
    for (int i = 0; i < 4; ++i) { f.statePre[i] = ld(&ks->statePre[i]); f.statePost[i] = ld(&ks->statePost[i]); f.reported[i] = ld(&ks->reported[i]); }
    for (int i = 0; i < 16; ++i) { f.Ppre[i] = ld(&ks->Ppre[i]); f.Ppost[i] = ld(&ks->Ppost[i]); }
    f.meas[0] = ld(&ks->meas[0]); f.meas[1] = ld(&ks->meas[1]);
    f.found = __hip_atomic_load(&ks->found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f.missing = __hip_atomic_load(&ks->missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f.aliased = __hip_atomic_load(&ks->aliased, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // posidet's centroid, exactly as the host epilogue derives it from the integer sums
    ResultRec &r = results[s];
    double x = 0.0, y = 0.0;
    const bool valid = r.valid != 0;
    if (valid) {
        const double a00 = (double)r.a00, a10 = (double)r.a10, a01 = (double)r.a01;
        const double db1_2 = a00 > 0 ? 0.5 : -0.5;
        const double db1_6 = a00 > 0 ? 0.16666666666666666666666666666667 : -0.16666666666666666666666666666667;
        const double m00 = a00 * db1_2;
        x = (a10 * db1_6) / m00;
        y = (a01 * db1_6) / m00;
    }
    filter_step(f, k, valid, x, y);

    const double *rep = f.aliased ? f.statePre : f.reported;
    r.kx = rep[0]; r.kvx = rep[1]; r.ky = rep[2]; r.kvy = rep[3];
    r.kal_valid = f.found;

    for (int i = 0; i < 4; ++i) { st(&ks->statePre[i], f.statePre[i]); st(&ks->statePost[i], f.statePost[i]); }
    for (int i = 0; i < 16; ++i) { st(&ks->Ppre[i], f.Ppre[i]); st(&ks->Ppost[i], f.Ppost[i]); }
    st(&ks->meas[0], f.meas[0]); st(&ks->meas[1], f.meas[1]);
    __hip_atomic_store(&ks->found, f.found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ks->missing, f.missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ks->aliased, f.aliased, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);                    // state stores have reached memory ...
    __hip_atomic_store(&ks->ticket, k.ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... before the next frame may read them
}

__global__ void k_kalman_reset(KalmanState *state, int n_streams, unsigned ticket)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    KalmanState &k = state[s];
    for (int i = 0; i < 4; ++i) { k.statePre[i] = 0.0; k.statePost[i] = 0.0; k.reported[i] = 6.0; }
    for (int i = 0; i < 16; ++i) { k.Ppre[i] = 0.0; k.Ppost[i] = 0.0; }
    k.meas[0] = k.meas[1] = 6.0;
    k.found = 0; k.missing = 0; k.aliased = 0;
    k.ticket = ticket;
}

void launch_kalman(const KalmanLaunch &k, ResultRec *results, int n_streams, hipStream_t st)
{
    hipLaunchKernelGGL(k_kalman, dim3((n_streams + 63) / 64), dim3(64), 0, st, k, results, n_streams);
}

void launch_kalman_reset(KalmanState *state, int n_streams, unsigned ticket, hipStream_t st)
{
    hipLaunchKernelGGL(k_kalman_reset, dim3((n_streams + 63) / 64), dim3(64), 0, st, state, n_streams, ticket);
}

}  // namespace oatgpu
