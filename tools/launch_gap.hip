// launch_gap.hip -- measurement aid: what does one in-order stream cost per kernel / per event on
// this box?  Build: hipcc --offload-arch=gfx950 -O2 tools/launch_gap.hip -o build/bin/launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_spin(float *p, int iters)
{
    float v = p[threadIdx.x & 63];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[0] = v;
}

struct Big { float *p; int iters; char pad[300]; };
__global__ void k_big(Big a)
{
    float v = a.p[threadIdx.x & 63] + a.pad[threadIdx.x & 255];
    for (int i = 0; i < a.iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) a.p[0] = v;
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;    // kernel length knob
    const int N = 4000;
    float *d;
    hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    hipStream_t A, B[2];
    hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
    int lo, hi;
    hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (auto &b : B) hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi);
    std::vector<hipEvent_t> ev(16), ev2(16);
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (auto &e : ev2) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    hipEvent_t t0, t1;
    hipEventCreate(&t0);
    hipEventCreate(&t1);

    // kernel duration alone
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters);
    hipStreamSynchronize(A);
    hipEventRecord(t0, A);
    hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters);
    hipEventRecord(t1, A);
    hipStreamSynchronize(A);
    float ms;
    hipEventElapsedTime(&ms, t0, t1);
    printf("kernel alone (event pair): %.2f us\n", ms * 1e3);

    auto bench = [&](const char *name, auto body) {
        hipDeviceSynchronize();
        double h = 0;
        const double a = now_us();
        for (int i = 0; i < N; ++i) { const double x = now_us(); body(i); h += now_us() - x; }
        hipDeviceSynchronize();
        const double b = now_us();
        printf("%-58s %7.2f us/iter  (host in calls %6.2f)\n", name, (b - a) / N, h / N);
    };
    bench("1 kernel", [&](int) { hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters); });
    Big big{};
    big.p = d; big.iters = iters;
    bench("1 kernel, 312-byte kernarg", [&](int) { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, A, big); });
    bench("1 kernel, 312-byte kernarg, 2025 blocks", [&](int) { hipLaunchKernelGGL(k_big, dim3(2025), dim3(256), 0, A, big); });
    bench("1 kernel + hipSetDevice + hipGetLastError", [&](int) {
        hipSetDevice(0);
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters);
        (void)hipGetLastError();
    });
    bench("alternate streams A,B0,A,B1 (1 kernel each)", [&](int i) {
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, (i & 1) ? B[(i >> 1) & 1] : A, d, iters);
    });
    bench("eventQuery on a finished event", [&](int) { (void)hipEventQuery(ev[0]); });
    bench("streamWaitEvent(B0, finished event)", [&](int) { hipStreamWaitEvent(B[0], ev[0], 0); });
    bench("1 kernel + eventRecord", [&](int i) {
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters);
        hipEventRecord(ev[i & 15], A);
    });
    bench("1 kernel + eventRecord + sync(event i-8)", [&](int i) {
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters);
        hipEventRecord(ev[i & 15], A);
        if (i >= 8) hipEventSynchronize(ev[(i - 8) & 15]);
    });
    bench("4 kernels", [&](int) {
        for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters / 4);
    });
    bench("A: kernel+rec ; B[i&1]: wait + 3 kernels + rec ; sync(i-8)", [&](int i) {
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters);
        hipEventRecord(ev[i & 15], A);
        hipStream_t b = B[i & 1];
        hipStreamWaitEvent(b, ev[i & 15], 0);
        for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, b, d, iters);
        hipEventRecord(ev2[i & 15], b);
        if (i >= 8) hipEventSynchronize(ev2[(i - 8) & 15]);
    });
    bench("same, 1 kernel of 3x length on B", [&](int i) {
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters);
        hipEventRecord(ev[i & 15], A);
        hipStream_t b = B[i & 1];
        hipStreamWaitEvent(b, ev[i & 15], 0);
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, b, d, iters * 3);
        hipEventRecord(ev2[i & 15], b);
        if (i >= 8) hipEventSynchronize(ev2[(i - 8) & 15]);
    });
    bench("everything on A: 4 kernels + rec ; sync(i-8)", [&](int i) {
        for (int j = 0; j < 4; ++j) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, A, d, iters);
        hipEventRecord(ev2[i & 15], A);
        if (i >= 8) hipEventSynchronize(ev2[(i - 8) & 15]);
    });
    return 0;
}
