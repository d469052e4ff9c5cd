// Does WHERE the model lies matter?  (tools/placement_probe.hip)  The dense K1 leg runs at 294, 315 or 350-360 us from
// process to process on the same binary.  Here ONE process allocates the 4K model again and again -- each time behind
// a growing pile of odd-sized allocations that are kept -- and times the bare access pattern of K1 (five weight planes
// and five record planes read and rewritten in place, one pixel per lane) and a plain in-place copy of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void planes_rec(float *st, uint8_t *nm, size_t P)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    float w[5];
    float4 r[5];
    const int n = nm[p];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float *wb = st + (size_t)k * 5 * P;
        w[k] = wb[p];
        r[k] = *(const float4 *)(wb + P + 4 * p);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float *wb = st + (size_t)k * 5 * P;
        wb[p] = w[k] + 1.0f;
        *(float4 *)(wb + P + 4 * p) = make_float4(r[k].x + 1.f, r[k].y + 1.f, r[k].z + 1.f, r[k].w + 1.f);
    }
    nm[p] = (uint8_t)n;
}
__global__ __launch_bounds__(256) void copy_inplace(uint4 *a, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { uint4 v = a[i]; v.x += 1; a[i] = v; }
}

static double time_us(hipEvent_t e0, hipEvent_t e1, int reps, void (*f)(void *), void *ctx)
{
    for (int i = 0; i < 5; ++i) f(ctx);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f(ctx);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}
struct Ctx { float *st; uint8_t *nm; size_t P; };
static void run_planes(void *c) { Ctx *x = (Ctx *)c; hipLaunchKernelGGL(planes_rec, dim3((unsigned)(x->P / 256)), dim3(256), 0, 0, x->st, x->nm, x->P); }
static void run_copy(void *c) { Ctx *x = (Ctx *)c; const size_t n = x->P * 25 / 4; hipLaunchKernelGGL(copy_inplace, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint4 *)x->st, n); }

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 12;
    const size_t P = (size_t)3840 * 2160;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    srand(12345);
    std::vector<void *> pads;
    printf("%-4s %-18s %12s %12s %14s\n", "try", "model address", "planes us", "copy us", "pad kept MB");
    size_t padsum = 0;
    for (int r = 0; r < rounds; ++r) {
        Ctx c{nullptr, nullptr, P};
        CK(hipMalloc(&c.st, P * 25 * 4)); CK(hipMalloc(&c.nm, P));
        CK(hipMemset(c.st, 0, P * 25 * 4)); CK(hipMemset(c.nm, 5, P));
        for (int i = 0; i < 100; ++i) run_copy(&c);           // clocks
        const double tp = time_us(e0, e1, 30, run_planes, &c), tc = time_us(e0, e1, 30, run_copy, &c);
        const double tp2 = time_us(e0, e1, 30, run_planes, &c);
        printf("%-4d %-18p %12.1f %12.1f %14.1f   (planes again: %.1f)\n", r, (void *)c.st, tp, tc, padsum / 1048576.0, tp2);
        CK(hipFree(c.st)); CK(hipFree(c.nm));
        const size_t pad = ((size_t)(rand() % 300) + 1) * 1048576 + (size_t)(rand() % 256) * 4096;
        void *q; CK(hipMalloc(&q, pad)); CK(hipMemset(q, 1, pad)); pads.push_back(q); padsum += pad;
    }
    return 0;
}
