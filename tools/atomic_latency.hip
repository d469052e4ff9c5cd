// Latency of ONE dependent memory operation as the back half's union-find issues them (tools/atomic_latency.hip):
// a single lane chases a permutation through a small table with (a) plain loads, (b) relaxed atomic loads and
// (c) atomic fetch_min, at workgroup and at agent scope.  Prints ns per operation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

template <int MODE>
__global__ void chase(int *tab, int n, int steps, long long *out, int big)
{
    if (threadIdx.x != 0) return;
    int i = (int)blockIdx.x * 4099 % n;       // (several workgroups = several XCDs chase the same table)
    const long long t0 = wall_clock64();
    for (int k = 0; k < steps; ++k) {
        if (MODE == 0) i = ((volatile int *)tab)[i];
        if (MODE == 1) i = __hip_atomic_load(tab + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) i = __hip_atomic_load(tab + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 3) i = __hip_atomic_fetch_min(tab + i, big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 4) i = __hip_atomic_fetch_min(tab + i, big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 5) i = __hip_atomic_fetch_min(tab + i, big, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const long long t1 = wall_clock64();
    if (blockIdx.x == 0) out[0] = t1 - t0;
    out[1 + blockIdx.x] = i;
}

int main()
{
    const int n = 1 << 16, steps = 20000;       // 256 KiB table: L2 resident
    std::vector<int> perm(n), tab(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937 rng(1);
    std::shuffle(perm.begin() + 1, perm.end(), rng);
    for (int k = 0; k < n; ++k) tab[perm[k]] = perm[(k + 1) % n];      // one cycle through all entries
    int *d; long long *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 8 * 65);
    hipMemcpy(d, tab.data(), n * 4, hipMemcpyHostToDevice);
    int rate = 0;
    hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);   // kHz
    const char *names[] = {"plain (volatile) load", "atomic load, workgroup scope", "atomic load, agent scope",
                           "atomic fetch_min, workgroup scope", "atomic fetch_min, agent scope", "atomic fetch_min, system scope"};
    for (int nb : {1, 8, 64})
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 6; ++m) {
            switch (m) {
            case 0: hipLaunchKernelGGL(chase<0>, nb, 64, 0, 0, d, n, steps, o, 0x7fffffff); break;
            case 1: hipLaunchKernelGGL(chase<1>, nb, 64, 0, 0, d, n, steps, o, 0x7fffffff); break;
            case 2: hipLaunchKernelGGL(chase<2>, nb, 64, 0, 0, d, n, steps, o, 0x7fffffff); break;
            case 3: hipLaunchKernelGGL(chase<3>, nb, 64, 0, 0, d, n, steps, o, 0x7fffffff); break;
            case 4: hipLaunchKernelGGL(chase<4>, nb, 64, 0, 0, d, n, steps, o, 0x7fffffff); break;
            case 5: hipLaunchKernelGGL(chase<5>, nb, 64, 0, 0, d, n, steps, o, 0x7fffffff); break;
            }
            long long h[2];
            hipMemcpy(h, o, 16, hipMemcpyDeviceToHost);
            if (rep) printf("%2d workgroup(s)  %-36s %8.1f ns per dependent operation\n", nb, names[m], (double)h[0] / rate * 1e6 / steps);
        }
    return 0;
}
