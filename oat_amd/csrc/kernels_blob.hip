// kernels_blob.hip -- K2..K5: the back half of posidet on BIT-PACKED masks.
//
//   cv::erode / cv::dilate, MORPH_RECT k x k     (HSVDetector.cpp:152-156)
//   cv::findContours(RETR_EXTERNAL, ...)          (DetectorFunc.cpp:41-43)
//   cv::moments(contour)                          (DetectorFunc.cpp:50)
//   the largest-area-in-window selection          (DetectorFunc.cpp:45-65)
//
// The reference follows borders sequentially.  Here the same numbers come from
// an order-free formulation (proved equal to the sequential one by the CPU
// tests, tests/test_oracle_contours_crosscheck.py):
//   * a mask row is a string of 64-bit words (1 bit/pixel, 1080p row = 30 words);
//     morphology is shifts and AND/OR on words;
//   * connected components are found over horizontal RUNS, not pixels: the
//     union-find node of a run is the raster index of its first pixel, so the
//     parent table is touched only at run heads.  Foreground runs merge
//     8-connected, background runs 4-connected, in the same pass;
//   * the root of every component is its smallest head = its first pixel in
//     raster order, which is exactly the reference's tie-break key; the
//     background component of pixel 0 (root 0) is "outside" (the image frame is
//     zeroed first, as OpenCV 3.1 does), anything else is a hole;
//   * each foreground pixel emits at most one directed polygon edge per side
//     facing OUTSIDE background; exact int64 Green sums per root.
#include "oatgpu_internal.h"

namespace oatgpu {

__device__ __forceinline__ int msb64(u64 v) { return 63 - __clzll((long long)v); }
__device__ __forceinline__ int lsb64(u64 v) { return __ffsll((long long)v) - 1; }
__device__ __forceinline__ u64 low_mask_incl(int i) { return (i >= 63) ? ~0ull : ((2ull << i) - 1ull); }

// bits of word w that are real pixels (x < W)
__device__ __forceinline__ u64 valid_bits(const Geom &g, int w)
{
    const int rem = g.W - w * 64;
    return rem >= 64 ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
}

// ------------------------------------------------------------- morphology ----
// One thread = one output word.  Window of output bit x is [x-a, x-a+k-1] in
// both directions (a = k/2, not reflected); outside the image reads 1 for
// erosion and 0 for dilation (morphologyDefaultBorderValue).  k <= 63.
__global__ __launch_bounds__(256) void k_morph(Geom g, const u64 *src_all, u64 *dst_all, int k, int is_erode,
                                               int first_stream)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.H * g.words) return;
    const int s = first_stream + blockIdx.y;
    const int y = t / g.words, w = t - y * g.words;
    const u64 *src = src_all + (size_t)s * (g.Palloc >> 6);
    u64 *dst = dst_all + (size_t)s * (g.Palloc >> 6);
    const int a = k / 2;
    const u64 ident = is_erode ? ~0ull : 0ull;
    const u64 vlast = valid_bits(g, g.words - 1);

    u64 ap = ident, ac = ident, an = ident;
    for (int j = 0; j < k; ++j) {
        const int yy = y - a + j;
        if (yy < 0 || yy >= g.H) continue;          // border rows are the identity
        const u64 *row = src + (size_t)yy * g.words;
        u64 pw = w > 0 ? row[w - 1] : ident;
        u64 cw = row[w];
        u64 nw = (w + 1 < g.words) ? row[w + 1] : ident;
        if (is_erode) {                              // padding bits beyond x = W-1 read as 1
            if (w == g.words - 1) cw |= ~vlast;
            if (w + 1 == g.words - 1) nw |= ~vlast;
            ap &= pw; ac &= cw; an &= nw;
        } else {
            ap |= pw; ac |= cw; an |= nw;
        }
    }
    u64 out = ident;
    for (int j = 0; j < k; ++j) {
        const int o = j - a;                         // out bit i takes in bit i+o
        u64 v;
        if (o == 0) v = ac;
        else if (o > 0) v = (ac >> o) | (an << (64 - o));
        else v = (ac << (-o)) | (ap >> (64 + o));
        out = is_erode ? (out & v) : (out | v);
    }
    dst[(size_t)y * g.words + w] = out & valid_bits(g, w);
}

void launch_morph(const Geom &g, const u64 *src, u64 *dst, int k, bool is_erode, int first_stream,
                  int n_streams, hipStream_t st)
{
    const int n = g.H * g.words;
    hipLaunchKernelGGL(k_morph, dim3((n + 255) / 256, n_streams), dim3(256), 0, st, g, src, dst, k,
                       is_erode ? 1 : 0, first_stream);
}

// dilation of word (y, w) straight from the pre-dilation mask (same window rule as k_morph)
__device__ __forceinline__ u64 dilate_word(const Geom &g, const u64 *src, int y, int w, int k)
{
    const int a = k / 2;
    u64 ap = 0, ac = 0, an = 0;
    for (int j = 0; j < k; ++j) {
        const int yy = y - a + j;
        if (yy < 0 || yy >= g.H) continue;
        const u64 *row = src + (size_t)yy * g.words;
        if (w > 0) ap |= row[w - 1];
        ac |= row[w];
        if (w + 1 < g.words) an |= row[w + 1];
    }
    u64 out = 0;
    for (int j = 0; j < k; ++j) {
        const int o = j - a;
        out |= (o == 0) ? ac : (o > 0) ? ((ac >> o) | (an << (64 - o))) : ((ac << (-o)) | (ap >> (64 + o)));
    }
    return out & valid_bits(g, w);
}

// erosion of word (y, w) straight from the raw mask (same window rule as k_morph)
__device__ __forceinline__ u64 erode_word(const Geom &g, const u64 *src, int y, int w, int k)
{
    const int a = k / 2;
    const u64 vlast = valid_bits(g, g.words - 1);
    u64 ap = ~0ull, ac = ~0ull, an = ~0ull;
    for (int j = 0; j < k; ++j) {
        const int yy = y - a + j;
        if (yy < 0 || yy >= g.H) continue;
        const u64 *row = src + (size_t)yy * g.words;
        u64 cw = row[w];
        u64 nw = (w + 1 < g.words) ? row[w + 1] : ~0ull;
        if (w == g.words - 1) cw |= ~vlast;          // padding bits beyond x = W-1 read as 1
        if (w + 1 == g.words - 1) nw |= ~vlast;
        if (w > 0) ap &= row[w - 1];
        ac &= cw;
        an &= nw;
    }
    u64 out = ~0ull;
    for (int j = 0; j < k; ++j) {
        const int o = j - a;
        out &= (o == 0) ? ac : (o > 0) ? ((ac >> o) | (an << (64 - o))) : ((ac << (-o)) | (ap >> (64 + o)));
    }
    return out & valid_bits(g, w);
}

// dilation of word (rr, w) out of a block-local stack of eroded rows in LDS: row j of the
// window is er[(rr + j) * words + ...] (rows outside the image were stored as zeros)
__device__ __forceinline__ u64 dilate_word_lds(const Geom &g, const u64 *er, int rr, int w, int k)
{
    const int a = k / 2;
    u64 ap = 0, ac = 0, an = 0;
    for (int j = 0; j < k; ++j) {
        const u64 *row = er + (size_t)(rr + j) * g.words;
        if (w > 0) ap |= row[w - 1];
        ac |= row[w];
        if (w + 1 < g.words) an |= row[w + 1];
    }
    u64 out = 0;
    for (int j = 0; j < k; ++j) {
        const int o = j - a;
        out |= (o == 0) ? ac : (o > 0) ? ((ac >> o) | (an << (64 - o))) : ((ac << (-o)) | (ap >> (64 + o)));
    }
    return out & valid_bits(g, w);
}

// ------------------------------------------------------------- row scan ------
// One wavefront per image row at a time, one lane per mask word (rows wider than 4096 px
// loop in chunks of 64 words).  Applies the erosion (ERODE: the workgroup first
// erodes the ROWS + dil_k - 1 rows its dilation windows touch into LDS) and the
// dilation (dil_k > 1) on the fly, writes the final mask (image frame zeroed,
// as cvStartFindContours does in OpenCV 3.1), the run-start bits T, the start
// x of the run entering every word, and initialises the union-find at run heads.
//
// Workgroup shape: four waves, a row each.  Other shapes were measured beside the one-wave per-pixel launch of the early
// order (r07, profiles/r07a_rowscan_shape_ab.txt, r07e_rowscan_shape_ab.txt, r07h_timeline.txt): ONE wave a workgroup taking
// 4 / 2 / 1 rows in turn -- no earlier start, a longer chain per wave: gpu_total 243-263 -> 368-432 us; 8 / 16 waves a
// workgroup -- the whole launch runs in 16-21 us once it is in, but waits 60 us and more for room on one compute unit
// (it gets in when the per-pixel launch drains): no gain either.  The in-kernel clocks (tools/rowscan_probe.py, -DOATGPU_RS_TIMING)
// say where the 55-60 us of this kernel beside the per-pixel kernel go: a workgroup RUNS 7 us (9 p90), the 540 workgroups of a 4K
// frame START over 35 us -- the dispatcher hands this queue ~15 workgroups a microsecond while the per-pixel launch streams
// 1 300 a microsecond through the other.
// (Rows a workgroup: 8 / 16 instead of 4 -- fewer workgroups beside K1, a longer loop in each -- is a trade, not a gain: 4K
// 19.0 k -> 19.3 k / 19.6 k fps and one 1080p stream 56.2 k -> 56.8 k / 58.9 k, because K1 is disturbed less, but a frame's
// result comes 40 / 83 us later (gpu_total 242 -> 282 / 325 us) and one frame at a time takes 127 -> 131 / 141 us:
// profiles/r07y_rowscan_rows_per_workgroup_ab.txt.  Oat is a real-time tracker: four rows.)
constexpr int kRsWaves = 4, kRsRows = 4;

#ifdef OATGPU_RS_TIMING             // measurement builds only (make variant DEFS=-DOATGPU_RS_TIMING, tools/rowscan_probe.py)
constexpr unsigned kRsTkRing = 1u << 16;
__device__ long long g_rs_tk[kRsTkRing * 5u];       // {first instruction, last instruction, row group, tag} of a row group (100 MHz wall clock)
__device__ unsigned g_rs_tk_n;
__device__ const unsigned long long *g_rs_k1_end;   // where the per-pixel launches stamp the end of their last workgroups
void oatgpu_debug_rs_set_k1_end(const unsigned long long *p)
{
    static const unsigned long long *cur = nullptr;
    if (p != cur) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rs_k1_end), &p, sizeof p); cur = p; }
}
extern "C" __attribute__((visibility("default"))) int oatgpu_debug_rs_timing(long long *out, int max_rows)
{
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_rs_tk_n), sizeof n) != hipSuccess) return -1;
    const unsigned rows = n < kRsTkRing ? n : kRsTkRing;
    const unsigned take = rows < (unsigned)max_rows ? rows : (unsigned)max_rows;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rs_tk), (size_t)take * 5u * sizeof(long long)) != hipSuccess) return -1;
    return (int)take;
}
#endif
// blockIdx.z selects one of TWO frames (source mask, scratch set): the two frames of a two-frame step are scanned by ONE
// launch (launch_blob_pair: a launch is ~4 us of host time, and small frames are bound by exactly that).
template <bool ERODE>
__global__ __launch_bounds__(64 * kRsWaves) void k_rowscan(Geom g, const u64 *src0, const u64 *src1, int ero_k, int dil_k, BlobBuffers b0,
                                                        BlobBuffers b1, int first_stream, int clear_lds_ok, unsigned tag)
{
    extern __shared__ u64 er[];
    constexpr int WAVES = kRsWaves, ROWS = kRsRows;
    const bool second = blockIdx.z != 0;
    const BlobBuffers &b = second ? b1 : b0;
    const u64 *src_all = second ? src1 : src0;
    const int grp = blockIdx.x;                  // row group of this workgroup
    {
#ifdef OATGPU_RS_TIMING
    const long long rs_t0 = wall_clock64();
    const long long rs_k1_end_seen = g_rs_k1_end ? (long long)__hip_atomic_load(g_rs_k1_end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ll;
#endif
    constexpr int NT = 64 * WAVES;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int s = first_stream + blockIdx.y;
    const u64 *src_img = src_all + (size_t)s * (g.Palloc >> 6);
    const int dk = dil_k > 1 ? dil_k : 1;
    if (ERODE) {
        const int r0 = grp * ROWS - dk / 2;
        const int n = (ROWS + dk - 1) * g.words;
        // Masks are mostly empty: if no bit is set in any row the erosion windows of this workgroup touch, the
        // eroded rows are zero (an image row never erodes to more than it holds) -- one pass over the source
        // words instead of ero_k x 3 loads per eroded word (4K, 7 x 7: 960 loads instead of 12 600 per workgroup).
        const int ya = max(r0 - ero_k / 2, 0), yb = min(r0 + (ROWS + dk - 1) - ero_k / 2 + ero_k - 1, g.H);   // source rows [ya, yb)
        u64 seen = 0ull;
        for (int i = ya * g.words + (int)threadIdx.x; i < yb * g.words; i += NT) seen |= src_img[i];
        const bool any = WAVES == 1 ? __ballot(seen != 0ull) != 0ull : __syncthreads_or(seen != 0ull) != 0;
        if (any) {
            for (int i = threadIdx.x; i < n; i += NT) {
                const int rr = i / g.words, w = i - rr * g.words;
                const int yy = r0 + rr;
                er[i] = (yy < 0 || yy >= g.H) ? 0ull : erode_word(g, src_img, yy, w, ero_k);
            }
        } else {
            for (int i = threadIdx.x; i < n; i += NT) er[i] = 0ull;
        }
        __syncthreads();
    }
    int *parent = b.parent + (size_t)s * g.Palloc;
    long long *acc = b.acc + (size_t)s * g.Palloc * 3;
    const int lastx = g.W - 1;

#pragma unroll 1
    for (int rr = wave; rr < ROWS; rr += WAVES) {
    const int y = grp * ROWS + rr;
    if (y >= g.H) break;
    const size_t woff = (size_t)s * (g.Palloc >> 6) + (size_t)y * g.words;
    u64 *morph = b.morph + woff;
    u64 *fin = b.fin + woff;
    u64 *trans = b.trans + woff;
    int *carry = b.carry + (size_t)s * g.H * g.words + (size_t)y * g.words;

    if (y == 0 && lane == 0) { b.nroots[s] = 0u; if (clear_lds_ok) b.lds_ok[s] = 0u; }

    const bool frame_row = (y == 0) || (y == g.H - 1);
    int chunk_carry = 0;        // start x of the last run seen so far
    u64 chunk_prevbit = 0;      // pixel value just left of this chunk
    unsigned short *wpre = b.wpre + ((size_t)s * g.H + y) * g.words;
    unsigned runs_before = 0;   // run starts of this row in earlier chunks
    bool row_fg = false;
    u64 keepT = 0ull, keepF = 0ull;   // rows of <= 64 words (<= 4096 px): the lane's T and F stay in registers for the loop below

    for (int c0 = 0; c0 < g.words; c0 += 64) {
        const int w = c0 + lane;
        const bool active = w < g.words;
        u64 F = 0;
        if (active && ERODE) {
            F = dilate_word_lds(g, er, rr, w, dk);   // dk == 1: the eroded word itself
            morph[w] = F;                            // the reference's threshold_frame_ (parity tap)
        } else if (active && dil_k > 1) {
            F = dilate_word(g, src_img, y, w, dil_k);
            morph[w] = F;
        } else if (active) {
            F = src_img[(size_t)y * g.words + w];
        }
        if (frame_row) F = 0;
        if (active && !frame_row) {
            F &= valid_bits(g, w);
            if (w == 0) F &= ~1ull;
            if (w == (lastx >> 6)) F &= ~(1ull << (lastx & 63));
        }
        // pixel left of bit 0: previous lane's bit 63
        u64 up = __shfl_up(F, 1);
        u64 prevbit = (lane == 0) ? chunk_prevbit : (up >> 63);
        u64 T = F ^ ((F << 1) | prevbit);
        if (w == 0) T |= 1ull;                       // a run starts at x = 0 by definition
        T &= valid_bits(g, w);
        if (!active) T = 0;

        const u64 nz = __ballot(T != 0);
        const u64 lower = nz & ((1ull << lane) - 1ull);
        const int pl = lower ? msb64(lower) : 0;
        const u64 Tp = __shfl(T, pl);
        const int cin = lower ? ((c0 + pl) * 64 + msb64(Tp)) : chunk_carry;

        // run starts in the words before this one (k_blob_lds places a row's runs with it)
        const unsigned cntT = (unsigned)__popcll(T);
        unsigned incl = cntT;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (active) {
            fin[w] = F;
            trans[w] = T;
            carry[w] = cin;
            wpre[w] = (unsigned short)min(runs_before + incl - cntT, 65535u);
        }
        keepT = T; keepF = F;
        runs_before += __shfl(incl, 63);
        row_fg = row_fg || __ballot(F != 0ull) != 0ull;
        // carry state into the next chunk (all lanes agree)
        if (nz) {
            const int hl = msb64(nz);
            const u64 Th = __shfl(T, hl);
            chunk_carry = (c0 + hl) * 64 + msb64(Th);
        }
        chunk_prevbit = __shfl(F, 63) >> 63;
    }

    // Union-find initialisation at run heads.  Column 0 and column W-1 are background in every
    // row (zeroed frame) and vertically 4-connected, so the first and the last run of every row
    // belong to the OUTSIDE component: hang them under root 0 right away -- on ordinary frames
    // this removes the H-long merge chain of full-width background runs.  Foreground run heads
    // get their Green accumulators cleared here.
    if (lane == 0) b.rowinfo[(size_t)s * g.H + y] = row_fg ? (int)runs_before : 0;
    const int last_start = chunk_carry;          // start x of the row's last run
    for (int c0 = 0; c0 < g.words; c0 += 64) {
        const int w = c0 + lane;
        if (w >= g.words) continue;
        // (written by this very lane above; re-read only when the row has more than one chunk -- the stores are write-
        // through, the loads were a global round trip of their own)
        u64 tt = g.words <= 64 ? keepT : trans[w];
        const u64 F = g.words <= 64 ? keepF : fin[w];
        while (tt) {
            const int i = lsb64(tt);
            tt &= tt - 1;
            const int sx = w * 64 + i;
            const int head = y * g.Wp + sx;
            parent[head] = (sx == 0 || sx == last_start) ? 0 : head;
            if ((F >> i) & 1ull) { acc[(size_t)head * 3] = 0; acc[(size_t)head * 3 + 1] = 0; acc[(size_t)head * 3 + 2] = 0; }
        }
    }
    }
#ifdef OATGPU_RS_TIMING
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        const unsigned slot = atomicAdd(&g_rs_tk_n, 1u) & (kRsTkRing - 1u);
        g_rs_tk[slot * 5u] = rs_t0; g_rs_tk[slot * 5u + 1] = wall_clock64(); g_rs_tk[slot * 5u + 2] = grp; g_rs_tk[slot * 5u + 3] = tag;
        g_rs_tk[slot * 5u + 4] = rs_k1_end_seen;
    }
#else
    (void)tag;
#endif
    }
}

size_t rowscan_lds_bytes(const Geom &g, int dil_k)
{
    return (size_t)(kRsRows + (dil_k > 1 ? dil_k : 1) - 1) * g.words * sizeof(u64);
}

// src / b: nf (1 or 2) frames' source masks and scratch sets
static void launch_rowscan(const Geom &g, const u64 *const *src, int ero_k, int dil_k, const BlobBuffers *b, int nf, int first_stream,
                           int n_streams, int clear, hipStream_t st, unsigned tag = 0u)
{
    const int k = nf > 1 ? 1 : 0;
    const dim3 grid((g.H + kRsRows - 1) / kRsRows, n_streams, nf), block(64 * kRsWaves);
    if (ero_k > 1)
        hipLaunchKernelGGL(k_rowscan<true>, grid, block, rowscan_lds_bytes(g, dil_k), st, g, src[0], src[k], ero_k, dil_k, b[0], b[k],
                           first_stream, clear, tag);
    else
        hipLaunchKernelGGL(k_rowscan<false>, grid, block, 0, st, g, src[0], src[k], 0, dil_k, b[0], b[k], first_stream, clear, tag);
}

// ------------------------------------------------------------ union-find -----
// All accesses to parent[] inside the merge / flatten kernels are agent-scope
// relaxed atomics: correct for any placement of workgroups on XCDs (per-XCD
// L2s and per-CU L1s are not coherent for plain accesses).
__device__ __forceinline__ int uf_ld(int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_min(int *p, int v)
{
    return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int uf_find(int *parent, int i)
{
    int cur = i;
    int p = uf_ld(parent + cur);
    while (p != cur) {
        const int gp = uf_ld(parent + p);
        if (gp != p) uf_min(parent + cur, gp);       // path halving (monotone, safe under races)
        cur = p;
        p = gp;
    }
    return cur;
}

__device__ __forceinline__ void uf_union(int *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }   // a > b: hang a under b
        const int old = uf_min(parent + a, b);
        if (old == a) return;                           // a was still a root
        a = old;                                        // lost a race: keep merging old with b
    }
}

// head (first pixel, padded index) of the run that contains pixel (x, y)
__device__ __forceinline__ int run_head(const Geom &g, const u64 *trans, const int *carry, int y, int x)
{
    const int w = x >> 6, i = x & 63;
    const u64 t = trans[(size_t)y * g.words + w] & low_mask_incl(i);
    const int sx = t ? (w * 64 + msb64(t)) : carry[(size_t)y * g.words + w];
    return y * g.Wp + sx;
}
// same, with this word's T / carry already in registers
__device__ __forceinline__ int run_head_w(const Geom &g, u64 Tw, int cw, int y, int w, int i)
{
    const u64 t = Tw & low_mask_incl(i);
    const int sx = t ? (w * 64 + msb64(t)) : cw;
    return y * g.Wp + sx;
}

// One thread = one word of row y (y >= 1): unite runs of row y with the runs
// of row y-1 they touch.  Background: 4-connected.  Foreground: 8-connected.
__global__ __launch_bounds__(256) void k_merge(Geom g, BlobBuffers b, int first_stream)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (g.H - 1) * g.words) return;
    const int s = first_stream + blockIdx.y;
    if (b.lds_ok[s]) return;                       // k_blob_lds took this frame
    const int y = 1 + t / g.words, w = t % g.words;
    const size_t soff = (size_t)s * (g.Palloc >> 6);
    const u64 *fin = b.fin + soff;
    const u64 *trans = b.trans + soff;
    const int *carry = b.carry + (size_t)s * g.H * g.words;
    int *parent = b.parent + (size_t)s * g.Palloc;

    const size_t rc = (size_t)y * g.words, ru = (size_t)(y - 1) * g.words;
    const u64 cur = fin[rc + w], up = fin[ru + w];
    const u64 curP = w > 0 ? fin[rc + w - 1] : 0ull;
    const u64 upP = w > 0 ? fin[ru + w - 1] : 0ull;
    const u64 upN = (w + 1 < g.words) ? fin[ru + w + 1] : 0ull;
    const u64 valid = valid_bits(g, w);
    const u64 Tc = trans[rc + w], Tu = trans[ru + w];
    const int cc = carry[rc + w], cu = carry[ru + w];

    // --- background, vertical contacts; one union per contact run ---
    u64 cb = ~cur & ~up & valid;
    if (w > 0 && (cb & 1ull) && (((~curP & ~upP) >> 63) & 1ull))
        cb &= cb + 1ull;                     // run continues from the previous word: its thread did it
    while (cb) {
        const int i = lsb64(cb);
        uf_union(parent, run_head_w(g, Tc, cc, y, w, i), run_head_w(g, Tu, cu, y - 1, w, i));
        cb &= cb + (cb & (~cb + 1ull));      // clear the lowest run of ones
    }
    if (cur == 0ull) return;

    // --- foreground, vertical contacts ---
    u64 cv = cur & up;
    if (w > 0 && (cv & 1ull) && (((curP & upP) >> 63) & 1ull))
        cv &= cv + 1ull;
    while (cv) {
        const int i = lsb64(cv);
        uf_union(parent, run_head_w(g, Tc, cc, y, w, i), run_head_w(g, Tu, cu, y - 1, w, i));
        cv &= cv + (cv & (~cv + 1ull));
    }
    // --- foreground, diagonal contacts not implied by a vertical one ---
    u64 dl = cur & ((up << 1) | (upP >> 63)) & ~up;   // up pixel at x-1
    while (dl) {
        const int i = lsb64(dl);
        dl &= dl - 1;
        uf_union(parent, run_head_w(g, Tc, cc, y, w, i), run_head(g, trans, carry, y - 1, w * 64 + i - 1));
    }
    u64 dr = cur & ((up >> 1) | (upN << 63)) & ~up;   // up pixel at x+1
    while (dr) {
        const int i = lsb64(dr);
        dr &= dr - 1;
        uf_union(parent, run_head_w(g, Tc, cc, y, w, i), run_head(g, trans, carry, y - 1, w * 64 + i + 1));
    }
}

// read-only walk to the root (after the merge kernel has completed: plain loads)
__device__ __forceinline__ int uf_root(const int *parent, int i)
{
    int p = parent[i];
    while (p != i) { i = p; p = parent[i]; }
    return i;
}

// K5: contour sums + selection in ONE kernel.
// Phase 1, one thread = one word of an interior row:
//   * every foreground run head that is a root announces itself in the stream's root list;
//   * Green sums over the directed edges of foreground pixels that face OUTSIDE background
//     (root 0), accumulated per root with int64 atomics.  Roots are found by walking parent[]
//     (no flatten pass); consecutive border pixels of one edge look at the same runs, so the last
//     (run head -> root) pair per lookup class is memoised and a straight edge costs one walk.
// Phase 2, the last workgroup of the stream to arrive (arrival counter; everything it reads that
// this kernel wrote -- root list, sums -- is written and read with agent-scope atomics, each wave
// drains vmcnt before arriving): every root offers (|a00| << 32 | first pixel); the maximum is
// the largest area with ties going to the later first pixel -- the reference walks its reversed
// contour list with a strict '>' -- and becomes the stream's result record in host-mapped memory.
constexpr int kGreenChunks = 4;
constexpr int kGreenBlock = 1024;          // big workgroups: every workgroup costs one same-address arrival atomic (~12 ns each, serialised)
__global__ __launch_bounds__(kGreenBlock) void k_green_select(Geom g, BlobBuffers b, double min_area, double max_area,
                                                      ResultRec *results, int first_stream)
{
    __shared__ int is_last;
    __shared__ unsigned long long red[kGreenBlock];
    // kGreenChunks threads per mask word, each owning 64 / kGreenChunks of its bits: the per-bit loops below are
    // serial, and a word on a horizontal edge of a blob has up to 64 border pixels -- with one thread per word
    // that thread alone ran 10+ us at 1080p while a frame without a blob took 5 (profiles/r02_backhalf_latency_*.md).
    const int tt = blockIdx.x * blockDim.x + threadIdx.x;
    const int t = tt / kGreenChunks;
    const u64 chunk_bits = (kGreenChunks == 1 ? ~0ull : ((1ull << (64 / kGreenChunks)) - 1ull)) << ((tt % kGreenChunks) * (64 / kGreenChunks));
    const int s = first_stream + blockIdx.y;
    if (b.lds_ok[s]) return;                       // k_blob_lds took this frame (uniform over the workgroup)
    const size_t soff = (size_t)s * (g.Palloc >> 6);
    const u64 *fin = b.fin + soff;
    const u64 *trans = b.trans + soff;
    const int *carry = b.carry + (size_t)s * g.H * g.words;
    const int *parent = b.parent + (size_t)s * g.Palloc;
    long long *acc = b.acc + (size_t)s * g.Palloc * 3;
    int *roots = b.roots + (size_t)s * (g.Palloc >> 1);

    u64 cur = 0ull;
    int y = 0, w = 0;
    if (t < (g.H - 2) * g.words) {
        y = 1 + t / g.words; w = t % g.words;
        cur = fin[(size_t)y * g.words + w];
    }
    if ((cur & chunk_bits) != 0ull) {
        const size_t rc = (size_t)y * g.words, ru = rc - g.words, rd = rc + g.words;
        const u64 Tc = trans[rc + w];
        // ---- roots announce themselves ----
        u64 heads = Tc & cur & chunk_bits;
        while (heads) {
            const int i = lsb64(heads);
            heads &= heads - 1;
            const int h = y * g.Wp + w * 64 + i;
            if (parent[h] == h) {
                const unsigned idx = __hip_atomic_fetch_add(&b.nroots[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(roots + idx, h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- Green sums ----
        const bool hp = w > 0, hn = w + 1 < g.words;
        const u64 U = fin[ru + w], D = fin[rd + w];
        const u64 curP = hp ? fin[rc + w - 1] : 0ull, curN = hn ? fin[rc + w + 1] : 0ull;
        const u64 upP = hp ? fin[ru + w - 1] : 0ull, upN = hn ? fin[ru + w + 1] : 0ull;
        const u64 dnP = hp ? fin[rd + w - 1] : 0ull, dnN = hn ? fin[rd + w + 1] : 0ull;
        const u64 L = (cur << 1) | (curP >> 63), R = (cur >> 1) | (curN << 63);
        const u64 UL = (U << 1) | (upP >> 63), UR = (U >> 1) | (upN << 63);
        const u64 DL = (D << 1) | (dnP >> 63), DR = (D >> 1) | (dnN << 63);

        u64 cand = cur & ~(L & R & U & D) & chunk_bits;
        const u64 Tu = trans[ru + w], Td = trans[rd + w];
        const int cc = carry[rc + w], cu = carry[ru + w], cd = carry[rd + w];

        struct Memo { int head, root; };
        Memo mo{-1, 0}, ml{-1, 0}, mb{-1, 0}, mr{-1, 0}, mt{-1, 0};
        auto root_of = [&](Memo &m, int head) -> int {
            if (head != m.head) { m.head = head; m.root = uf_root(parent, head); }
            return m.root;
        };
        auto flush = [&](int label, long long s00, long long s10, long long s01) {
            if (label >= 0 && (s00 | s10 | s01)) {
                atomicAdd((unsigned long long *)&acc[(size_t)label * 3], (unsigned long long)s00);
                atomicAdd((unsigned long long *)&acc[(size_t)label * 3 + 1], (unsigned long long)s10);
                atomicAdd((unsigned long long *)&acc[(size_t)label * 3 + 2], (unsigned long long)s01);
            }
        };

        int label = -1;
        long long s00 = 0, s10 = 0, s01 = 0;
        while (cand) {
            const int i = lsb64(cand);
            cand &= cand - 1;
            const int x = w * 64 + i;
            const u64 bit = 1ull << i;
            const int lab = root_of(mo, run_head_w(g, Tc, cc, y, w, i));
            if (lab != label) { flush(label, s00, s10, s01); label = lab; s00 = s10 = s01 = 0; }
            // side: neighbour bg pixel, then B (diagonal, first) and A (straight, second)
            int qx, qy;
            bool e;
#define OAT_EDGE()                                                                        \
            if (e) {                                                                      \
                const int d = x * qy - qx * y;                                            \
                s00 += d; s10 += (long long)d * (x + qx); s01 += (long long)d * (y + qy); \
            }
            if (!(L & bit)) {                                                             // left
                const int h = i > 0 ? run_head_w(g, Tc, cc, y, w, i - 1) : run_head(g, trans, carry, y, x - 1);
                if (root_of(ml, h) == 0) {
                    e = true;
                    if (DL & bit) { qx = x - 1; qy = y + 1; } else if (D & bit) { qx = x; qy = y + 1; } else e = false;
                    OAT_EDGE()
                }
            }
            if (!(D & bit) && root_of(mb, run_head_w(g, Td, cd, y + 1, w, i)) == 0) {    // bottom
                e = true;
                if (DR & bit) { qx = x + 1; qy = y + 1; } else if (R & bit) { qx = x + 1; qy = y; } else e = false;
                OAT_EDGE()
            }
            if (!(R & bit)) {                                                             // right
                const int h = i < 63 ? run_head_w(g, Tc, cc, y, w, i + 1) : run_head(g, trans, carry, y, x + 1);
                if (root_of(mr, h) == 0) {
                    e = true;
                    if (UR & bit) { qx = x + 1; qy = y - 1; } else if (U & bit) { qx = x; qy = y - 1; } else e = false;
                    OAT_EDGE()
                }
            }
            if (!(U & bit) && root_of(mt, run_head_w(g, Tu, cu, y - 1, w, i)) == 0) {    // top
                e = true;
                if (UL & bit) { qx = x - 1; qy = y - 1; } else if (L & bit) { qx = x - 1; qy = y; } else e = false;
                OAT_EDGE()
            }
#undef OAT_EDGE
        }
        flush(label, s00, s10, s01);
    }

    // ---- arrival; the last workgroup of this stream selects ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(&b.done[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (prev == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!is_last) return;

    const unsigned nr = __hip_atomic_load(&b.nroots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long best = 0ull;
    for (unsigned i = threadIdx.x; i < nr; i += blockDim.x) {
        const int h = __hip_atomic_load(roots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const long long a00 = __hip_atomic_load(&acc[(size_t)h * 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a00 == 0) continue;
        const u64 mag = (u64)(a00 < 0 ? -a00 : a00);
        const double area = (double)mag * 0.5;            // m00 = a00 * (+-0.5), exact
        if (area >= min_area && area < max_area) {
            const unsigned long long key = (mag << 32) | (u64)(unsigned)h;
            best = key > best ? key : best;
        }
    }
    red[threadIdx.x] = best;
    __syncthreads();
    for (int st = kGreenBlock / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st && red[threadIdx.x + st] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const u64 key = red[0];
        ResultRec r{};
        r.first_pixel = -1;
        if (key) {
            const int h = (int)(key & 0xffffffffull);
            r.a00 = __hip_atomic_load(&acc[(size_t)h * 3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r.a10 = __hip_atomic_load(&acc[(size_t)h * 3 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r.a01 = __hip_atomic_load(&acc[(size_t)h * 3 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int yy = h / g.Wp, xx = h - yy * g.Wp;
            r.first_pixel = yy * g.W + xx;
            r.valid = 1;
        }
        results[s] = r;
        __hip_atomic_store(&b.done[s], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------ K4'/K5': one workgroup, in LDS ----
// Masks are mostly empty: after morphology a frame holds a few blobs, i.e. a few hundred "dirty" rows with three to
// five runs each.  Then ONE workgroup per stream does what k_merge + k_green_select do, on a run list in LDS: every
// dependent step of the union-find is an LDS access (~40 ns) instead of an L2 atomic (~270 ns, tools/atomic_latency.hip),
// and the three launches with their drain/fill become one.  k_rowscan supplies, per row, whether it holds foreground and
// how many runs (rowinfo), and per word the run starts before it (wpre).  Runs of a row alternate background /
// foreground starting with background (column 0 is zeroed), so run k of a row is foreground iff k is odd.
//   nodes: 0 = OUTSIDE; 1..R = the runs of the dirty rows in raster order (so the smallest node of a component is its
//   first pixel, the reference's tie-break key -- as in the global kernels).
//   A background run is OUTSIDE if it is the first or the last of its row, or if the row above or below is not dirty
//   (an all-background row is outside as a whole: both its ends are); otherwise it joins the background runs it
//   overlaps in the row above (4-connected).  Foreground runs join the foreground runs of the row above they touch
//   (8-connected: overlap widened by one pixel).
// Falls back (lds_ok = 0: k_merge + k_green_select take the frame) when the frame has more than kLdsRows dirty rows,
// kLdsRuns runs or kLdsRoots foreground components.
constexpr int kLdsRows = 1024, kLdsRuns = 3072, kLdsRoots = 768;      // 58 KB of LDS
// Threads of the one workgroup.  Beside the per-pixel kernel a launch of this kernel lasts 81-96 us at 4K in a kernel
// trace, against 21-27 us alone.  What the difference is (r03, tools/lds_phase_probe.py with a -DOATGPU_LDS_TIMING build:
// the kernel stamps the 100 MHz wall clock between its phases, profiles/r03j_backhalf_slot_wait.txt): the workgroup
// itself RUNS 26.3 us beside the per-pixel kernel and 25.5 us alone, phase for phase the same -- the other 55 us of
// the trace's duration pass before its first instruction: 16 waves with 61 KB of LDS need four free wave slots and
// 288 registers on every SIMD of ONE compute unit, the per-pixel kernel holds 8 waves / all 512 registers per SIMD
// everywhere and refills every slot a retiring 4-wave workgroup frees, and stream priority does not make the
// dispatcher hold slots back: the workgroup gets in when the running per-pixel launch drains (half a launch = 53 us
// on average).  (Earlier in r03 the duration was read as execution time and blamed on memory latency; the in-kernel
// clock refutes that, and so did removing this kernel's loads.)  Smaller workgroups do get in earlier but run longer:
// 512 / 256 threads 106 / 111 us and 16-21 % off the one-1080p-stream frame rate; s_setprio 3: no change.  Keeping the
// per-pixel kernel's stream off 4 / 8 / 16 compute units with a CU mask (hipExtStreamCreateWithCUMask) brings this
// kernel to 72 / 72 / 38 us and costs the per-pixel kernel 10 % whatever the number (107 -> 118 us per two-frame 4K
// launch, 17.5 k -> 15.9 k fps): not adopted.  A camera-bound pipeline never sees the wait: no later frame's
// per-pixel kernel is running when a frame's back half starts.
constexpr int kLdsBlock = 1024;
constexpr int kLdsTrip = 8;      // words of the run-start image a thread has in flight per trip of phase B

#ifdef OATGPU_LDS_TIMING           // measurement builds only (make variant DEFS=-DOATGPU_LDS_TIMING, tools/lds_phase_probe.py)
constexpr unsigned kLdsTkRing = 4096u;
__device__ long long g_lds_tk[kLdsTkRing * 10u];
__device__ unsigned g_lds_tk_n;
extern "C" __attribute__((visibility("default"))) int oatgpu_debug_lds_timing(long long *out, int max_rows)
{
    unsigned n = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_lds_tk_n), sizeof n) != hipSuccess) return -1;
    const unsigned rows = n < kLdsTkRing ? n : kLdsTkRing;
    const unsigned take = rows < (unsigned)max_rows ? rows : (unsigned)max_rows;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lds_tk), (size_t)take * 10u * sizeof(long long)) != hipSuccess) return -1;
    return (int)take;
}
#endif

__device__ __forceinline__ int lds_find(int *par, int i)
{
    int cur = i, p = par[cur];
    while (p != cur) {
        const int gp = par[p];
        if (gp != p) atomicMin(&par[cur], gp);
        cur = p;
        p = gp;
    }
    return cur;
}
__device__ __forceinline__ void lds_union(int *par, int a, int b)
{
    for (;;) {
        a = lds_find(par, a);
        b = lds_find(par, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&par[a], b);
        if (old == a) return;
        a = old;
    }
}

// Early dispatch (r04).  Beside the per-pixel kernel this kernel's one big workgroup used to WAIT ~55 us for 16 free wave
// slots on one compute unit (above) -- after its row scan had finished, i.e. on the frame's critical path.  With
// wait_ticket != 0 the launch sits on a HIP stream of its own and is submitted together with the frame's other kernels,
// long before the row scan runs: the workgroup takes its slots whenever the device has them (at the latest when the
// running per-pixel launch drains, one launch before it is needed), parks -- one lane polls b.ready[s] between
// s_sleeps, the other waves sit in the barrier and issue nothing; 16 of the chip's 8 192 wave slots -- and starts the
// moment k_rowscan's last workgroup publishes the ticket.  Should the ticket not come within kWaitTicks (a tool that
// serialises kernels, e.g. a counter-collecting profiler, makes the row scan wait for THIS kernel) the frame is
// declined like one that is too busy: the global kernels take it.
constexpr long long kWaitTicks = 10000000ll;         // 100 ms of the 100 MHz wall clock
template <typename T> __device__ __forceinline__ T ld_pub(const T *p)
{
    return __hip_atomic_load(const_cast<T *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// blockIdx.z selects one of TWO frames (scratch set, result record, ticket): the two frames of a two-frame step park
// their workgroups with ONE launch (launch_blob_tail2) -- two launches on one stream would run one after the other.
__global__ __launch_bounds__(kLdsBlock) void k_blob_lds(Geom g, BlobBuffers b0, BlobBuffers b1, double min_area, double max_area,
                                                        ResultRec *results0, ResultRec *results1, int first_stream, int spec,
                                                        unsigned ticket0, unsigned ticket1)
{
    __shared__ int wait_failed;
    const bool second = blockIdx.z != 0;
    const BlobBuffers &b = second ? b1 : b0;
    ResultRec *const results = second ? results1 : results0;
    const unsigned wait_ticket = second ? ticket1 : ticket0;
    __shared__ unsigned short rows[kLdsRows + 1];        // dirty row r -> image row
    __shared__ unsigned rptr[kLdsRows + 2];              // dirty row r -> its first node
    __shared__ unsigned short rstart[kLdsRuns + 2];      // node -> start x
    __shared__ unsigned short rrow[kLdsRuns + 2];        // node -> dirty row
    __shared__ int par[kLdsRuns + 2];
    __shared__ unsigned short rootslot[kLdsRuns + 2];
    __shared__ int rootnode[kLdsRoots];
    __shared__ unsigned long long acc[kLdsRoots * 3];
    __shared__ unsigned scan_d[kLdsBlock / 64], scan_r[kLdsBlock / 64];
    __shared__ unsigned nroots_s;
    __shared__ unsigned short fglist[kLdsRuns / 2 + 2];  // the foreground nodes, in any order (phase E walks these)
    __shared__ unsigned long long red[kLdsBlock / 64];

    const int s = first_stream + blockIdx.y;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t soff = (size_t)s * (g.Palloc >> 6);
    const u64 *fin = b.fin + soff;
    const u64 *trans = b.trans + soff;
    const unsigned short *wpre = b.wpre + (size_t)s * g.H * g.words;
    const int *rowinfo = b.rowinfo + (size_t)s * g.H;

#ifdef OATGPU_LDS_TIMING
    long long tk[16]; int tn = 0;
#define TK() tk[tn++] = wall_clock64()
#else
#define TK()
#endif
    if (wait_ticket) {
        if (t == 0) {
            int bad = 0;
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(&b.ready[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != wait_ticket) {
                // (somebody serialises kernel dispatches and a workgroup of this context has already waited its 100 ms in vain:
                // the host learns of it only when it collects that frame -- the frames parked behind it must not pay again)
                if (__hip_atomic_load(b.nopark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = kPathNoPark; break; }
                __builtin_amdgcn_s_sleep(16);
                if (wall_clock64() - t0 > kWaitTicks) {
                    bad = kPathTimeout;
                    __hip_atomic_store(b.nopark, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            wait_failed = bad;
        }
        __syncthreads();
        if (wait_failed) {
            if (t == 0) {
                b.lds_ok[s] = 0u;
                if (spec) { ResultRec r{}; r.first_pixel = -1; r.valid = kNeedsGlobal; r.path = wait_failed; results[s] = r; }
            }
            return;
        }
        // No acquire fence here: it is an L2 invalidate per wave, 16 per workgroup, on an XCD whose L2 the per-pixel kernel is
        // streaming through (measured: 4K per-pixel launch 104 -> 108 us, 16 x 1080p 403 -> 491 us).  Instead every load
        // of the row scan's arrays below is an agent-scope load (ld_pub): it is served by the memory side, where the row
        // scan's write-through stores are, whatever this XCD's L2 or this unit's L1 hold of an earlier frame.
    }
    TK();
    // ---- A: dirty rows and their run counts, in order (each thread owns a stretch of consecutive rows) ----
    const int per = (g.H + kLdsBlock - 1) / kLdsBlock;
    const int y0 = t * per, y1 = min(y0 + per, g.H);
    unsigned d = 0, rn = 0;
    // (r03: the row counts stay in registers for the second pass below when a thread owns <= 4 rows -- frames up to 4096
    // rows: every dependent global round trip of this kernel costs 4-5 us beside the per-pixel kernel, section header)
    int ric[4] = {0, 0, 0, 0};
    if (per <= 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int y = y0 + q; if (y < y1) ric[q] = ld_pub(rowinfo + y); }
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ric[q]) { d++; rn += (unsigned)ric[q]; }
    } else {
        for (int y = y0; y < y1; ++y) { const int ri = ld_pub(rowinfo + y); if (ri) { d++; rn += (unsigned)ri; } }
    }
    unsigned di = d, ri_ = rn;                          // inclusive scans over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned vd = __shfl_up(di, o), vr = __shfl_up(ri_, o);
        if (lane >= o) { di += vd; ri_ += vr; }
    }
    if (lane == 63) { scan_d[wave] = di; scan_r[wave] = ri_; }
    if (t == 0) nroots_s = 0;
    __syncthreads();
    unsigned dbase = di - d, rbase = ri_ - rn, D = 0, R = 0;
    for (int i = 0; i < kLdsBlock / 64; ++i) {
        if (i < wave) { dbase += scan_d[i]; rbase += scan_r[i]; }
        D += scan_d[i]; R += scan_r[i];
    }
    if (D > (unsigned)kLdsRows || R > (unsigned)kLdsRuns) {          // too busy a frame for this path
        if (t == 0) {
            b.lds_ok[s] = 0u;
            if (spec) { ResultRec r{}; r.first_pixel = -1; r.valid = kNeedsGlobal; results[s] = r; }
        }
        return;
    }
    if (D == 0) {                                                     // nothing in the frame
        if (t == 0) {
            ResultRec r{};
            r.first_pixel = -1;
            r.path = 1;
            results[s] = r;
            b.lds_ok[s] = 1u;
        }
        return;
    }
    if (per <= 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ri = ric[q];
            if (ri) { rows[dbase] = (unsigned short)(y0 + q); rptr[dbase] = 1u + rbase; dbase++; rbase += (unsigned)ri; }
        }
    } else {
        for (int y = y0; y < y1; ++y) {
            const int ri = ld_pub(rowinfo + y);
            if (ri) { rows[dbase] = (unsigned short)y; rptr[dbase] = 1u + rbase; dbase++; rbase += (unsigned)ri; }
        }
    }
    if (t == 0) { rptr[D] = 1u + R; par[0] = 0; }
    __syncthreads();

    TK();
    // ---- B: the run list ----  (kLdsTrip words per thread and trip: the loads of a trip are in flight together)
    {
        const unsigned total = D * (unsigned)g.words;
        for (unsigned i0 = t; i0 < total; i0 += (unsigned)kLdsTrip * kLdsBlock) {
            u64 T[kLdsTrip];
            unsigned rr[kLdsTrip], pre[kLdsTrip];
            size_t gi[kLdsTrip];
#pragma unroll
            for (int u = 0; u < kLdsTrip; ++u) {
                const unsigned i = i0 + (unsigned)u * kLdsBlock, ic = min(i, total - 1u);   // (no branch round the load)
                const unsigned r = ic / (unsigned)g.words, w = ic - r * (unsigned)g.words;
                rr[u] = r;
                gi[u] = (size_t)rows[r] * g.words + w;
                T[u] = ld_pub(trans + gi[u]);
                pre[u] = ld_pub(wpre + gi[u]);
                if (i >= total) T[u] = 0ull;
            }
#pragma unroll
            for (int u = 0; u < kLdsTrip; ++u) {
                u64 Tu = T[u];
                if (!Tu) continue;
                const unsigned r = rr[u], w = (unsigned)(gi[u] - (size_t)rows[r] * g.words);
                unsigned node = rptr[r] + pre[u];
                while (Tu) {
                    const int bit = lsb64(Tu);
                    Tu &= Tu - 1;
                    rstart[node] = (unsigned short)(w * 64u + (unsigned)bit);
                    rrow[node] = (unsigned short)r;
                    par[node] = (int)node;
                    // (every row holds f foreground and f + 1 background runs: rows before r hold (rptr[r] - 1 - r) / 2
                    // foreground runs -- the list index needs no counter)
                    const unsigned kk = node - rptr[r];
                    if (kk & 1u) fglist[((rptr[r] - 1u - r) >> 1) + (kk >> 1)] = (unsigned short)node;
                    node++;
                }
            }
        }
    }
    __syncthreads();

    TK();
    // ---- C: unions ----
    for (unsigned i = 1 + t; i <= R; i += kLdsBlock) {
        const unsigned r = rrow[i], k = i - rptr[r];
        const int y = rows[r];
        const int sx = rstart[i], ex = (i + 1 < rptr[r + 1]) ? (int)rstart[i + 1] - 1 : g.W - 1;
        const bool adj_up = r > 0 && rows[r - 1] == y - 1, adj_dn = r + 1 < D && rows[r + 1] == y + 1;
        if (!(k & 1u)) {
            if (k == 0 || i + 1 == rptr[r + 1] || !adj_up || !adj_dn) lds_union(par, (int)i, 0);
            if (adj_up) {
                const unsigned je = rptr[r];
                for (unsigned j = rptr[r - 1]; j < je; j += 2) {
                    const int sj = rstart[j];
                    if (sj > ex) break;
                    const int ej = (j + 1 < je) ? (int)rstart[j + 1] - 1 : g.W - 1;
                    if (ej >= sx) lds_union(par, (int)i, (int)j);
                }
            }
        } else if (adj_up) {
            const unsigned je = rptr[r];
            for (unsigned j = rptr[r - 1] + 1; j < je; j += 2) {
                const int sj = rstart[j];
                if (sj > ex + 1) break;
                const int ej = (j + 1 < je) ? (int)rstart[j + 1] - 1 : g.W - 1;
                if (ej >= sx - 1) lds_union(par, (int)i, (int)j);
            }
        }
    }
    __syncthreads();

    TK();
    // ---- D: flatten; foreground roots get an accumulator slot ----
    int myroot[(kLdsRuns + kLdsBlock) / kLdsBlock];
#pragma unroll
    for (int q = 0; q < (kLdsRuns + kLdsBlock) / kLdsBlock; ++q) {
        const unsigned i = 1 + t + q * kLdsBlock;
        myroot[q] = i <= R ? lds_find(par, (int)i) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < (kLdsRuns + kLdsBlock) / kLdsBlock; ++q) {
        const unsigned i = 1 + t + q * kLdsBlock;
        if (i <= R) {
            par[i] = myroot[q];
            if (myroot[q] == (int)i && ((i - rptr[rrow[i]]) & 1u)) {
                const unsigned slot = atomicAdd(&nroots_s, 1u);
                if (slot < (unsigned)kLdsRoots) {
                    rootslot[i] = (unsigned short)slot;
                    rootnode[slot] = (int)i;
                    acc[slot * 3] = 0ull; acc[slot * 3 + 1] = 0ull; acc[slot * 3 + 2] = 0ull;
                }
            }
        }
    }
    __syncthreads();
    const unsigned NR = nroots_s;
    if (NR > (unsigned)kLdsRoots) {
        if (t == 0) {
            b.lds_ok[s] = 0u;
            if (spec) { ResultRec r{}; r.first_pixel = -1; r.valid = kNeedsGlobal; results[s] = r; }
        }
        return;
    }

    TK();
    // ---- E: Green sums over the edges facing OUTSIDE background: 4 to 16 threads per foreground run, each taking its
    // share of the bits of every word (the per-bit loop is serial, and the top and bottom rows of a blob
    // are all border) ----
    const unsigned NF = (R - D) >> 1;
    // threads per run: 16, 8 or 4 -- as many as let all runs go in one pass (the per-bit loop below is serial and
    // ~300 instructions long; a thread takes 64 / lpr bits of every word)
    const int lpr_log = NF * 16u <= (unsigned)kLdsBlock ? 4 : (NF * 8u <= (unsigned)kLdsBlock ? 3 : 2);
    const int lpr = 1 << lpr_log, nbits = 64 >> lpr_log, sub = t & (lpr - 1);
    const int sh = nbits * sub;
    const u64 qbits = ((nbits == 64 ? ~0ull : ((1ull << nbits) - 1ull))) << sh;
    const unsigned qmask = (1u << nbits) - 1u;
    for (unsigned i0 = (unsigned)(t >> lpr_log); i0 < NF; i0 += (unsigned)(kLdsBlock >> lpr_log)) {
        const unsigned i = fglist[i0];
        const unsigned r = rrow[i];
        const int y = rows[r];
        const int sx = rstart[i], ex = (int)rstart[i + 1] - 1;       // a foreground run is never the last of its row
        const bool adj_up = r > 0 && rows[r - 1] == y - 1, adj_dn = r + 1 < D && rows[r + 1] == y + 1;
        const bool out_l = par[i - 1] == 0, out_r = par[i + 1] == 0;
        // run of row rr (a dirty neighbour row) that holds pixel x, memoised: [lo, hi] = its extent, root = its root
        struct Memo { int lo, hi, root; unsigned at; };
        Memo mu{1, 0, 0, adj_up ? rptr[r - 1] : 0u}, md{1, 0, 0, adj_dn ? rptr[r + 1] : 0u};
        auto root_at = [&](Memo &m, unsigned rr, int x) -> int {
            if (x >= m.lo && x <= m.hi) return m.root;
            unsigned j = (x > m.hi) ? m.at : rptr[rr];                // runs are visited left to right
            const unsigned je = rptr[rr + 1];
            while (j + 1 < je && (int)rstart[j + 1] <= x) ++j;
            m.at = j; m.lo = rstart[j]; m.hi = (j + 1 < je) ? (int)rstart[j + 1] - 1 : g.W - 1; m.root = par[j];
            return m.root;
        };
        long long s00 = 0, s10 = 0, s01 = 0;
        const size_t rc = (size_t)y * g.words, ru = rc - g.words, rd = rc + g.words;
#ifdef OATGPU_LDS_TIMING
        if (i0 == (unsigned)(t >> lpr_log)) { TK(); }
#endif
        // state of a neighbour row over [x0 - 1, x0 + 64] from the run list: 1 all foreground, 0 all OUTSIDE background,
        // 2 all background of a hole, -1 mixed (or the word is at the frame's edge)
        auto row_state = [&](Memo &m, unsigned rr, bool adj, int x0) -> int {
            if (!adj) return 0;                                       // the row is not in the list: no foreground, all outside
            const int rt = root_at(m, rr, x0);
            if (m.lo > x0 - 1 || m.hi < x0 + 64) return -1;
            if ((m.at - rptr[rr]) & 1u) return 1;
            return rt == 0 ? 0 : 2;
        };
        for (int w = sx >> 6; w <= (ex >> 6); ++w) {
            // A word strictly inside the run is all ones, and so are its left and right neighbours: its only edges are
            // straight top / bottom edges where the row above / below is outside background over the whole word -- known
            // from the run list in LDS, no load (r04: a run of n words cost n dependent round trips of loads per thread;
            // the full-frame blob of Oat's default window: 59 us of the kernel's 112).
            if (sx < w * 64 && ex > w * 64 + 63) {
                const int su = row_state(mu, r - 1, adj_up, w * 64), sd = row_state(md, r + 1, adj_dn, w * 64);
                if (su >= 0 && sd >= 0) {
                    if (su == 0 || sd == 0) {
                        const long long n = nbits, yy = y;
                        const long long sumx = n * (w * 64 + sh) + n * (n - 1) / 2;      // x over this thread's bits
                        if (su == 0) { s00 += n * yy; s10 += yy * (2 * sumx - n); s01 += 2 * yy * yy * n; }
                        if (sd == 0) { s00 -= n * yy; s10 -= yy * (2 * sumx + n); s01 -= 2 * yy * yy * n; }
                    }
                    continue;
                }
            }
            const int lo = max(sx - w * 64, 0), hi = min(ex - w * 64, 63);
            const u64 runbits = (hi == 63 ? ~0ull : ((1ull << (hi + 1)) - 1ull)) & ~((1ull << lo) - 1ull) & qbits;
            // the threads of a run need the same nine words: thread 0 of them fetches the run's row, thread 1
            // the row above, thread 2 the row below, and they pass them round (a quarter of the requests).
            // (r03: rebuilding the nine words from the run list in LDS instead -- no global load in this phase -- was
            // parity-green and no faster: 18.4 against 16.3 us alone at 1080p, 84 against 82 us beside the per-pixel
            // kernel at 4K, gpurun_out/bh1 -> profiles/r03j_backhalf_slot_wait.txt.  The loads are not what this
            // kernel waits for: see kLdsBlock.)
            const int role = sub;
            const size_t rbase = role == 0 ? rc : (role == 1 ? ru : rd);
            const bool hp = w > 0, hn = w + 1 < g.words;
            u64 m0 = 0ull, mP = 0ull, mN = 0ull;
            if (role < 3) {
                m0 = ld_pub(fin + rbase + w);
                mP = hp ? ld_pub(fin + rbase + w - 1) : 0ull;
                mN = hn ? ld_pub(fin + rbase + w + 1) : 0ull;
            }
            const int quad = lane & ~(lpr - 1);
            const u64 cur = __shfl(m0, quad), curP = __shfl(mP, quad), curN = __shfl(mN, quad);
            const u64 U = __shfl(m0, quad + 1), upP = __shfl(mP, quad + 1), upN = __shfl(mN, quad + 1);
            const u64 Dn = __shfl(m0, quad + 2), dnP = __shfl(mP, quad + 2), dnN = __shfl(mN, quad + 2);
            if (!runbits) continue;                                  // nothing of the run in this thread's bits
            // this thread's 16 bits of the nine masks as 32-bit values: the per-bit loop below is what the kernel's
            // time goes into (one compute unit runs it), and on 64-bit masks it was twice as many instructions
            const u64 L64 = (cur << 1) | (curP >> 63), R64 = (cur >> 1) | (curN << 63);
            const u64 UL64 = (U << 1) | (upP >> 63), UR64 = (U >> 1) | (upN << 63);
            const u64 DL64 = (Dn << 1) | (dnP >> 63), DR64 = (Dn >> 1) | (dnN << 63);
#define Q16(v) ((unsigned)((v) >> sh) & qmask)
            const unsigned cq = Q16(cur & runbits), Lq = Q16(L64), Rq = Q16(R64), Uq = Q16(U), Dq = Q16(Dn);
            const unsigned ULq = Q16(UL64), URq = Q16(UR64), DLq = Q16(DL64), DRq = Q16(DR64);
#undef Q16
            unsigned cand = cq & ~(Lq & Rq & Uq & Dq);
            const int xbase = w * 64 + sh;
            // Straight horizontal edges in closed form (r04).  A border pixel whose row above holds no foreground at all (the
            // row is not in the run list: everything there is OUTSIDE) and whose left neighbour is set contributes the top edge
            // (x, y) -> (x - 1, y): d = y, and the sums over a mask of such pixels need their count and the sum of their x --
            // four popcounts -- instead of one trip of the serial loop below each; the bottom edge (x, y) -> (x + 1, y)
            // likewise.  The top and the bottom row of a large blob are ALL such pixels: with Oat's default all-pass window
            // (one full-frame contour, the reference's own benchmark case) a thread walked up to 256 of them, 168 us a
            // 1 MP frame (profiles/r04x_posidet_default_trace.md).  Same int64 terms, same sums.
            unsigned ptop = 0u, pbot = 0u;
            if (!adj_up) ptop = cq & ~Uq & ~ULq & Lq;
            if (!adj_dn) pbot = cq & ~Dq & ~DRq & Rq;
            if (ptop | pbot) {
                auto sum_x = [&](unsigned m) -> long long {         // sum of x over the set bits of m
                    const int sb = __popc(m & 0xAAAAu) + 2 * __popc(m & 0xCCCCu) + 4 * __popc(m & 0xF0F0u) + 8 * __popc(m & 0xFF00u);
                    return (long long)__popc(m) * xbase + sb;
                };
                const long long nt = __popc(ptop), nb = __popc(pbot), yy = y;
                s00 += (nt - nb) * yy;
                s10 += yy * (2 * sum_x(ptop) - nt) - yy * (2 * sum_x(pbot) + nb);
                s01 += 2 * yy * yy * (nt - nb);
                // a pixel goes through the loop only for the sides that are left
                cand &= (~Lq | ~Rq | (~Uq & ~ptop) | (~Dq & ~pbot));
            }
            while (cand) {
                const int bi = __ffs((int)cand) - 1;
                cand &= cand - 1u;
                const int x = xbase + bi;
                const unsigned bit = 1u << bi;
                int qx, qy;
                bool e;
#define OAT_EDGE()  /* (qx, qy) is a neighbour of (x, y): |d| <= x + y < 2^15 (frames up to 16383 x 16383 take this path), the products fit 32 bits */ \
                if (e) {                                                                      \
                    const int dd = x * qy - qx * y;                                           \
                    s00 += dd; s10 += dd * (x + qx); s01 += dd * (y + qy);                    \
                }
                if (!(Lq & bit) && out_l) {                                                   // left
                    e = true;
                    if (DLq & bit) { qx = x - 1; qy = y + 1; } else if (Dq & bit) { qx = x; qy = y + 1; } else e = false;
                    OAT_EDGE()
                }
                if (!((Dq | pbot) & bit) && (!adj_dn || root_at(md, r + 1, x) == 0)) {        // bottom
                    e = true;
                    if (DRq & bit) { qx = x + 1; qy = y + 1; } else if (Rq & bit) { qx = x + 1; qy = y; } else e = false;
                    OAT_EDGE()
                }
                if (!(Rq & bit) && out_r) {                                                   // right
                    e = true;
                    if (URq & bit) { qx = x + 1; qy = y - 1; } else if (Uq & bit) { qx = x; qy = y - 1; } else e = false;
                    OAT_EDGE()
                }
                if (!((Uq | ptop) & bit) && (!adj_up || root_at(mu, r - 1, x) == 0)) {        // top
                    e = true;
                    if (ULq & bit) { qx = x - 1; qy = y - 1; } else if (Lq & bit) { qx = x - 1; qy = y; } else e = false;
                    OAT_EDGE()
                }
#undef OAT_EDGE
            }
        }
        if (s00 | s10 | s01) {
            const unsigned slot = rootslot[par[i]];
            atomicAdd(&acc[slot * 3], (unsigned long long)s00);
            atomicAdd(&acc[slot * 3 + 1], (unsigned long long)s10);
            atomicAdd(&acc[slot * 3 + 2], (unsigned long long)s01);
        }
    }
#ifdef OATGPU_LDS_TIMING
    TK();
#endif
    __syncthreads();

    TK();
    // ---- F: the largest area in the window; ties go to the later first pixel (k_green_select's rule) ----
    unsigned long long best = 0ull;
    for (unsigned q = t; q < NR; q += kLdsBlock) {
        const long long a00 = (long long)acc[q * 3];
        if (a00 == 0) continue;
        const u64 mag = (u64)(a00 < 0 ? -a00 : a00);
        const double area = (double)mag * 0.5;
        if (area >= min_area && area < max_area) {
            const int node = rootnode[q];
            const unsigned h = (unsigned)rows[rrow[node]] * (unsigned)g.Wp + (unsigned)rstart[node];
            const unsigned long long key = (mag << 32) | (u64)h;
            best = key > best ? key : best;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long v = __shfl_xor(best, o);
        best = v > best ? v : best;
    }
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (t == 0) {
        u64 key = 0ull;
        for (int i = 0; i < kLdsBlock / 64; ++i) key = red[i] > key ? red[i] : key;
        ResultRec r{};
        r.first_pixel = -1;
        r.path = 1;
        if (key) {
            const unsigned h = (unsigned)(key & 0xffffffffull);
            // the root's slot: find it again (few roots)
            for (unsigned q = 0; q < NR; ++q) {
                const int node = rootnode[q];
                if ((unsigned)rows[rrow[node]] * (unsigned)g.Wp + (unsigned)rstart[node] == h) {
                    r.a00 = (long long)acc[q * 3]; r.a10 = (long long)acc[q * 3 + 1]; r.a01 = (long long)acc[q * 3 + 2];
                    break;
                }
            }
            const int yy = (int)(h / (unsigned)g.Wp), xx = (int)(h - (unsigned)yy * (unsigned)g.Wp);
            r.first_pixel = yy * g.W + xx;
            r.valid = 1;
        }
        results[s] = r;
        b.lds_ok[s] = 1u;
#ifdef OATGPU_LDS_TIMING
        TK();
        // (a ring in device memory, read by oatgpu_debug_lds_timing: a printf here takes 400 us and turns "beside the
        // per-pixel kernel" into "alone")
        const unsigned slot = atomicAdd(&g_lds_tk_n, 1u) & (kLdsTkRing - 1u);
        for (int q = 0; q < 9; ++q) g_lds_tk[slot * 10u + q] = tk[q];
        g_lds_tk[slot * 10u + 9] = (long long)D << 32 | R;
#endif
    }
}


void launch_blob(const Geom &g, const BlobBuffers &b, const u64 *src_bits, int ero_k, int dil_k, double min_area,
                 double max_area, ResultRec *results, int first_stream, int n_streams, hipStream_t st, int mode)
{
    const bool lds_able = g.H > 2 && g.H <= 16383 && g.W <= 16383;
    if (mode == kBlobSpec && !lds_able) mode = kBlobFull;
    const int clear = mode == kBlobGlobal || !lds_able;          // nobody else resets lds_ok then
    launch_rowscan(g, &src_bits, ero_k, dil_k, &b, 1, first_stream, n_streams, clear, st);
    if (lds_able && mode != kBlobGlobal)
        hipLaunchKernelGGL(k_blob_lds, dim3(1, n_streams), dim3(kLdsBlock), 0, st, g, b, b, min_area, max_area, results, results,
                           first_stream, mode == kBlobSpec ? 1 : 0, 0u, 0u);
    if (mode == kBlobSpec) return;
    if (g.H > 1)
        hipLaunchKernelGGL(k_merge, dim3(((g.H - 1) * g.words + 255) / 256, n_streams), dim3(256), 0, st, g, b,
                           first_stream);
    // (grid of at least one workgroup even for H <= 2: the last-arriver logic writes the result)
    const int nw = (g.H > 2 ? (g.H - 2) * g.words : 1) * kGreenChunks;
    hipLaunchKernelGGL(k_green_select, dim3((nw + kGreenBlock - 1) / kGreenBlock, n_streams), dim3(kGreenBlock), 0, st, g, b,
                       min_area, max_area, results, first_stream);
}

// The frame's ticket comes from a kernel of its own behind the row scan: the stream's order and the row scan's
// end-of-kernel release make its output visible to agent-scope loads, one lane per stream publishes.  (The in-kernel
// form -- k_rowscan storing what k_blob_lds reads write-through and its last-arriving workgroup publishing the ticket --
// saved that launch and cost the per-pixel kernel 3-4 %: profiles/r04o_*, r04p_*.)
__global__ void k_publish_ticket(unsigned *ready, int first_stream, unsigned ticket)
{
    __hip_atomic_store(&ready[first_stream + threadIdx.x], ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_rowscan_signal(const Geom &g, const BlobBuffers &b, const u64 *src_bits, int ero_k, int dil_k, int first_stream,
                           int n_streams, unsigned ticket, hipStream_t st)
{
    launch_rowscan(g, &src_bits, ero_k, dil_k, &b, 1, first_stream, n_streams, 0, st, ticket);
    for (int s0 = 0; s0 < n_streams; s0 += 1024)
        hipLaunchKernelGGL(k_publish_ticket, dim3(1), dim3(n_streams - s0 < 1024 ? n_streams - s0 : 1024), 0, st, b.ready,
                           first_stream + s0, ticket);
}

// The back halves of the TWO frames of a step as two launches instead of four (speculative mode: row scan + the LDS
// kernel; a declined frame comes back marked kNeedsGlobal): frame i reads src[i] and scratch set b[i], writes results[i].
void launch_blob_pair(const Geom &g, const BlobBuffers *b, const u64 *const *src, int ero_k, int dil_k, double min_area,
                      double max_area, ResultRec *const *results, int n_streams, hipStream_t st)
{
    launch_rowscan(g, src, ero_k, dil_k, b, 2, 0, n_streams, 0, st);
    hipLaunchKernelGGL(k_blob_lds, dim3(1, n_streams, 2), dim3(kLdsBlock), 0, st, g, b[0], b[1], min_area, max_area, results[0],
                       results[1], 0, 1, 0u, 0u);
}

// The early blob workgroups of the nf (1 or 2) frames of a step in ONE launch: frame i reads scratch set b[i], writes
// results[i] and waits for ticket[i].  Speculative mode only (a declined frame comes back marked kNeedsGlobal).
void launch_blob_tail2(const Geom &g, const BlobBuffers *b, double min_area, double max_area, ResultRec *const *results,
                       int n_streams, const unsigned *ticket, int nf, hipStream_t st)
{
    const int k = nf > 1 ? 1 : 0;
    hipLaunchKernelGGL(k_blob_lds, dim3(1, n_streams, nf), dim3(kLdsBlock), 0, st, g, b[0], b[k], min_area, max_area, results[0],
                       results[k], 0, 1, ticket[0], ticket[k]);
}

}  // namespace oatgpu
