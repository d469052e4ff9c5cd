// tools/k1_lab.hip -- access-pattern lab for K1 (measurement aid only; not part of the product).
//
// K1 (k_mog_fused) reads and rewrites 25 fp32 planes + one u8 plane per pixel.  This program times
// the bare ACCESS PATTERNS it could use, on a 4K-sized model (Palloc = 3840*2160), with a trivial
// update (v + 1) instead of the mixture arithmetic, so that pattern and arithmetic can be told apart:
//
//   copy16        plain 16-byte grid-stride copy (the "achievable" reference), several grids, +- nontemporal
//   dword<1|2>    K1's pattern today: 1 px/lane, one 4-byte load/store per plane, loads in 1 or 2
//                 dependent phases (2 = mode counter + mode 0 first, the rest behind a data-dependent test)
//   vec<2|4>      2 or 4 px/lane in registers (8/16-byte accesses)
//   lds16         block-cooperative: every plane slice of the block's 256 px is ONE 16-byte-per-lane
//                 LDS-DMA instruction (global_load_lds_dwordx4), compute at 1 px/lane out of LDS,
//                 16-byte stores back from LDS
//
//   hipcc --offload-arch=gfx950 -O3 tools/k1_lab.hip -o build/bin/k1_lab && build/bin/k1_lab
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NPL = 25;

template <bool NT>
__global__ __launch_bounds__(256) void copy16(const uint4 *src, uint4 *dst, size_t n)
{
    typedef unsigned nv4 __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) {
            nv4 v = __builtin_nontemporal_load((const nv4 *)src + i);
            __builtin_nontemporal_store(v, (nv4 *)dst + i);
        } else {
            dst[i] = src[i];
        }
    }
}

// 4 independent 16-byte accesses in flight per thread
template <bool NT>
__global__ __launch_bounds__(256) void copy16x4(const uint4 *src, uint4 *dst, size_t n)
{
    typedef unsigned nv4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        nv4 a, b, c, d;
        if (NT) {
            a = __builtin_nontemporal_load((const nv4 *)src + i);
            b = __builtin_nontemporal_load((const nv4 *)src + i + stride);
            c = __builtin_nontemporal_load((const nv4 *)src + i + 2 * stride);
            d = __builtin_nontemporal_load((const nv4 *)src + i + 3 * stride);
            __builtin_nontemporal_store(a, (nv4 *)dst + i);
            __builtin_nontemporal_store(b, (nv4 *)dst + i + stride);
            __builtin_nontemporal_store(c, (nv4 *)dst + i + 2 * stride);
            __builtin_nontemporal_store(d, (nv4 *)dst + i + 3 * stride);
        } else {
            a = *((const nv4 *)src + i); b = *((const nv4 *)src + i + stride);
            c = *((const nv4 *)src + i + 2 * stride); d = *((const nv4 *)src + i + 3 * stride);
            *((nv4 *)dst + i) = a; *((nv4 *)dst + i + stride) = b;
            *((nv4 *)dst + i + 2 * stride) = c; *((nv4 *)dst + i + 3 * stride) = d;
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// K1's pattern: 1 px per lane, 4-byte accesses, in place.
template <int PHASES, int WAVES>
__global__ __launch_bounds__(256, WAVES) void planes_dword(float *st, uint8_t *nm, size_t PS)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    float v[NPL];
    const int n = nm[p];
    const int first[5] = {0, 5, 10, 11, 12};
#pragma unroll
    for (int i = 0; i < 5; ++i) v[first[i]] = st[(size_t)first[i] * PS + p];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        if (k == 0 || k == 5 || k == 10 || k == 11 || k == 12) continue;
        if (PHASES == 2) { v[k] = 0.f; if (n > 1) v[k] = st[(size_t)k * PS + p]; }
        else v[k] = st[(size_t)k * PS + p];
    }
#pragma unroll
    for (int k = 0; k < NPL; ++k) st[(size_t)k * PS + p] = v[k] + 1.0f;
    nm[p] = (uint8_t)n;
}

template <int VEC> struct V;
template <> struct V<2> { typedef float2 T; };
template <> struct V<4> { typedef float4 T; };

template <int VEC, int WAVES>
__global__ __launch_bounds__(256, WAVES) void planes_vec(float *st, uint8_t *nm, size_t PS)
{
    typedef typename V<VEC>::T T;
    const size_t p = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    T v[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) v[k] = *(const T *)(st + (size_t)k * PS + p);
    uint8_t nb[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) nb[j] = nm[p + j];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        float *f = (float *)&v[k];
#pragma unroll
        for (int j = 0; j < VEC; ++j) f[j] += 1.0f;
        *(T *)(st + (size_t)k * PS + p) = v[k];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) nm[p + j] = nb[j];
}

// Block-cooperative LDS staging.  Tile = 256 px; plane slice = 1 KiB = one wave-wide 16-B/lane op.
// TILES tiles per block, processed one after the other (no double buffering here).
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
template <int TILES, int WAVES>
__global__ __launch_bounds__(256, WAVES) void planes_lds16(float *st, uint8_t *nm, size_t PS)
{
    __shared__ float tile[NPL * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = 0; t < TILES; ++t) {
        const size_t base = ((size_t)blockIdx.x * TILES + t) * 256;
        for (int k = wave; k < NPL; k += 4)
            __builtin_amdgcn_global_load_lds((glb_void *)(st + (size_t)k * PS + base + lane * 4),
                                             (lds_void *)(tile + k * 256), 16, 0, 0);
        const int n = nm[base + threadIdx.x];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float v[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) v[k] = tile[k * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < NPL; ++k) tile[k * 256 + threadIdx.x] = v[k] + 1.0f;
        __syncthreads();
        for (int k = wave; k < NPL; k += 4)
            *(float4 *)(st + (size_t)k * PS + base + lane * 4) = *(const float4 *)(tile + k * 256 + lane * 4);
        nm[base + threadIdx.x] = (uint8_t)n;
        if (TILES > 1) __syncthreads();
    }
}

// Same, but the loads go through registers (global_load_dwordx4 -> ds_write_b128) instead of LDS-DMA.
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void planes_lds16_reg(float *st, uint8_t *nm, size_t PS)
{
    __shared__ float tile[NPL * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * 256;
    float4 r[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int k = wave + 4 * i;
        if (k < NPL) r[i] = *(const float4 *)(st + (size_t)k * PS + base + lane * 4);
    }
    const int n = nm[base + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int k = wave + 4 * i;
        if (k < NPL) *(float4 *)(tile + k * 256 + lane * 4) = r[i];
    }
    __syncthreads();
    float v[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) v[k] = tile[k * 256 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < NPL; ++k) tile[k * 256 + threadIdx.x] = v[k] + 1.0f;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int k = wave + 4 * i;
        if (k < NPL) *(float4 *)(st + (size_t)k * PS + base + lane * 4) = *(const float4 *)(tile + k * 256 + lane * 4);
    }
    nm[base + threadIdx.x] = (uint8_t)n;
}


// Generic 1-px-per-lane pattern: NP planes, planar (plane stride PS) or tiled (TILE px of every plane contiguous),
// optional nontemporal accesses.  P pixels in total.
template <int NP, int TILE, bool NT>
__global__ __launch_bounds__(256) void planes_gen(float *st, size_t PS)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    float v[NP];
    float *base = TILE ? st + (p / TILE) * ((size_t)NP * TILE) + (p % TILE) : st + p;
    const size_t stride = TILE ? (size_t)TILE : PS;
#pragma unroll
    for (int k = 0; k < NP; ++k) v[k] = NT ? __builtin_nontemporal_load(base + k * stride) : base[k * stride];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (NT) __builtin_nontemporal_store(v[k] + 1.0f, base + k * stride);
        else base[k * stride] = v[k] + 1.0f;
    }
}

// full-grid copy, E 16-byte elements per thread (E KiB per wave), consecutive waves adjacent
template <int E, bool NT>
__global__ __launch_bounds__(256) void copy16e(const uint4 *src, uint4 *dst, size_t n)
{
    typedef unsigned nv4 __attribute__((ext_vector_type(4)));
    const size_t i0 = ((size_t)blockIdx.x * E) * 256 + threadIdx.x;
    nv4 v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) if (i0 + e * 256 < n) v[e] = NT ? __builtin_nontemporal_load((const nv4 *)src + i0 + e * 256) : *((const nv4 *)src + i0 + e * 256);
#pragma unroll
    for (int e = 0; e < E; ++e) if (i0 + e * 256 < n) { if (NT) __builtin_nontemporal_store(v[e], (nv4 *)dst + i0 + e * 256); else *((nv4 *)dst + i0 + e * 256) = v[e]; }
}

// full-grid 4-byte copy (one dword per lane): is access width itself a limit?
template <int E, bool NT>
__global__ __launch_bounds__(256) void copy4e(const float *src, float *dst, size_t n)
{
    const size_t i0 = ((size_t)blockIdx.x * E) * 256 + threadIdx.x;
    float v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = NT ? __builtin_nontemporal_load(src + i0 + e * 256) : src[i0 + e * 256];
#pragma unroll
    for (int e = 0; e < E; ++e) { if (NT) __builtin_nontemporal_store(v[e], dst + i0 + e * 256); else dst[i0 + e * 256] = v[e]; }
}


#include <algorithm>
#include <functional>
#include <string>

// The record layout K1 uses since round 2: per mode a weight plane (4 B/px) and a {variance, mean[3]} record plane
// (16 B/px), 1 px per lane, in place; PHASES = 2: mode 0 first, the other four behind a data-dependent test.
template <int PHASES>
__global__ __launch_bounds__(256) void planes_rec(float *st, uint8_t *nm, size_t P)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    float w[5];
    float4 r[5];
    const int n = nm[p];
    w[0] = st[p];
    r[0] = *(const float4 *)(st + P + 4 * p);
#pragma unroll
    for (int k = 1; k < 5; ++k) {
        float *wb = st + (size_t)k * 5 * P;
        if (PHASES == 2) { w[k] = 0.f; r[k] = make_float4(0, 0, 0, 0); if (n > 1 && r[0].x >= 0.f) { w[k] = wb[p]; r[k] = *(const float4 *)(wb + P + 4 * p); } }
        else { w[k] = wb[p]; r[k] = *(const float4 *)(wb + P + 4 * p); }
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float *wb = st + (size_t)k * 5 * P;
        wb[p] = w[k] + 1.0f;
        *(float4 *)(wb + P + 4 * p) = make_float4(r[k].x + 1.f, r[k].y + 1.f, r[k].z + 1.f, r[k].w + 1.f);
    }
    nm[p] = (uint8_t)n;
}

struct Variant { std::string name; std::function<void()> launch; double bytes; std::vector<double> ms; };

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    const int reps = argc > 2 ? atoi(argv[2]) : 40;
    const size_t P = (size_t)3840 * 2160;                     // multiple of 1024
    const size_t PS = P;
    float *st; uint8_t *nm; uint4 *dst;
    CK(hipMalloc(&st, NPL * PS * 4)); CK(hipMalloc(&nm, P)); CK(hipMalloc(&dst, NPL * PS * 4));
    CK(hipMemset(st, 0, NPL * PS * 4)); CK(hipMemset(nm, 5, P)); CK(hipMemset(dst, 0, NPL * PS * 4));
    const double bytes_planes = (double)P * (NPL * 8 + 2);
    const size_t n16 = NPL * PS * 4 / 16;
    const double bytes_copy = (double)n16 * 32;
    const int nblk = (int)(P / 256);
    std::vector<Variant> V;
    auto add = [&](const char *name, std::function<void()> f, double bytes) { V.push_back({name, f, bytes, {}}); };

    add("copy16 grid-stride 2048 blocks", [&] { hipLaunchKernelGGL(copy16<false>, dim3(2048), dim3(256), 0, 0, (const uint4 *)st, dst, n16); }, bytes_copy);
    add("copy16 grid-stride 2048 blocks nt", [&] { hipLaunchKernelGGL(copy16<true>, dim3(2048), dim3(256), 0, 0, (const uint4 *)st, dst, n16); }, bytes_copy);
#define CE(E, NT, label) add(label, [&] { hipLaunchKernelGGL((copy16e<E, NT>), dim3((unsigned)((n16 + 256 * E - 1) / (256 * E))), dim3(256), 0, 0, (const uint4 *)st, dst, n16); }, bytes_copy);
    CE(1, false, "copy16 full grid 1/thread")
    CE(1, true, "copy16 full grid 1/thread nt")
    CE(4, false, "copy16 full grid 4/thread")
    CE(4, true, "copy16 full grid 4/thread nt")
    add("copy16 full grid IN PLACE", [&] { hipLaunchKernelGGL((copy16e<1, false>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const uint4 *)st, (uint4 *)st, n16); }, bytes_copy);
    add("copy16 full grid IN PLACE nt", [&] { hipLaunchKernelGGL((copy16e<1, true>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, (const uint4 *)st, (uint4 *)st, n16); }, bytes_copy);
#define C4(E, NT, label) add(label, [&] { hipLaunchKernelGGL((copy4e<E, NT>), dim3((unsigned)(n16 * 4 / (256 * E))), dim3(256), 0, 0, (const float *)st, (float *)dst, n16 * 4); }, bytes_copy);
    C4(25, false, "copy4 full grid 25/thread")
    C4(25, true, "copy4 full grid 25/thread nt")
    add("planes dword 1 phase (K1 today)", [&] { hipLaunchKernelGGL((planes_dword<1, 1>), dim3(nblk), dim3(256), 0, 0, st, nm, PS); }, bytes_planes);
    add("planes dword 2 phases (K1 today)", [&] { hipLaunchKernelGGL((planes_dword<2, 1>), dim3(nblk), dim3(256), 0, 0, st, nm, PS); }, bytes_planes);
    add("planes rec layout 1 phase", [&] { hipLaunchKernelGGL((planes_rec<1>), dim3(nblk), dim3(256), 0, 0, st, nm, PS); }, bytes_planes);
    add("planes rec layout 2 phases", [&] { hipLaunchKernelGGL((planes_rec<2>), dim3(nblk), dim3(256), 0, 0, st, nm, PS); }, bytes_planes);
    add("planes vec2 (regs)", [&] { hipLaunchKernelGGL((planes_vec<2, 1>), dim3(nblk / 2), dim3(256), 0, 0, st, nm, PS); }, bytes_planes);
    add("planes vec4 (regs)", [&] { hipLaunchKernelGGL((planes_vec<4, 1>), dim3(nblk / 4), dim3(256), 0, 0, st, nm, PS); }, bytes_planes);
    add("planes lds16 dma", [&] { hipLaunchKernelGGL((planes_lds16<1, 1>), dim3(nblk), dim3(256), 0, 0, st, nm, PS); }, bytes_planes);
#define GEN(NP, TILE, NT, label) { const size_t Pn = (P * NPL / NP) / 1024 * 1024; \
        add(label, [=] { hipLaunchKernelGGL((planes_gen<NP, TILE, NT>), dim3((unsigned)(Pn / 256)), dim3(256), 0, 0, st, Pn); }, (double)Pn * NP * 8); }
    GEN(2, 0, false, "gen  2 planes planar")
    GEN(5, 0, false, "gen  5 planes planar")
    GEN(5, 0, true, "gen  5 planes planar nt")
    GEN(25, 0, false, "gen 25 planes planar")
    GEN(25, 0, true, "gen 25 planes planar nt")
    GEN(25, 256, false, "gen 25 planes tiled 256")
    GEN(25, 256, true, "gen 25 planes tiled 256 nt")
    GEN(25, 1024, false, "gen 25 planes tiled 1024")
    GEN(25, 1024, true, "gen 25 planes tiled 1024 nt")

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // warm the device up (clocks) before anything is timed
    for (int i = 0; i < 400; ++i) V[2].launch();
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (auto &v : V) {
            v.launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) v.launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / reps);
        }
    printf("%-36s %9s %9s %9s   %8s  %s\n", "variant (4K model, 1.7 GB moved)", "min us", "median us", "max us", "GB/s med", "of 8 TB/s");
    for (auto &v : V) {
        std::sort(v.ms.begin(), v.ms.end());
        const double med = v.ms[v.ms.size() / 2];
        printf("%-36s %9.1f %9.1f %9.1f   %8.0f  %.3f\n", v.name.c_str(), v.ms.front() * 1e3, med * 1e3, v.ms.back() * 1e3,
               v.bytes / (med * 1e-3) / 1e9, v.bytes / (med * 1e-3) / 1e9 / 8000.0);
    }
    return 0;
}
