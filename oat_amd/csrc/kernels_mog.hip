// kernels_mog.hip -- K1: the fused per-pixel front half of the hot path for gfx950.
//
//   cv::BackgroundSubtractorMOG2::apply      (BackgroundSubtractorMOG.cpp:124)
//   frame.setTo(0, mask == 0)                (BackgroundSubtractorMOG.cpp:125)
//   cv::cvtColor(.., COLOR_BGR2HSV)          (ColorConvert.cpp:104)
//   cv::inRange(hsv, lo, hi)                 (HSVDetector.cpp:146-149)
//
// in ONE pass over HBM: 3 B/px of BGR in, the 101 B/px Gaussian-mixture model
// read and written in place, 1 bit/px of threshold mask out.  Bandwidth bound
// (205 algorithmic B/px); no MFMA.  Design notes:
//   * one lane = one pixel, one wavefront = 64 consecutive pixels = one mask word:
//     __ballot(thr) IS the word -- the threshold image never exists as bytes.
//   * the per-pixel update keeps OpenCV's operation order exactly (compiled
//     with -ffp-contract=off: every mul/add rounds on its own, like the
//     reference's x86-64 build); the mixture lives in registers with fully
//     unrolled, statically indexed mode loops (no scratch).
//   * BGR->HSV uses RGB2HSV_b's integer quotients, taken on the spot by foreground lanes only
//     (a zeroed pixel is HSV (0,0,0): one uniform window test); the stand-alone conversion
//     kernel builds the two tables once per block in LDS.
//   * what nothing on the fused path can observe is not computed (detectShadowGMM: mog2_finish),
//     what did not change is not stored (a frozen model is read-only: FROZEN), and what is
//     only needed late is loaded late (kernel arguments: KRELOAD).
//   * the model is sparse in practice and the kernel moves only what the arithmetic can
//     depend on, in TWO load phases: (1) counter byte + mode 0 + the pixel, for every lane;
//     then the loop's mode-0 iteration runs, which tells whether the pixel matched mode 0 as
//     background; (2) of slots 1..n-1: the weight of LIVE slots (non-zero weight, hinted in
//     the counter byte), and variance/mean only on lanes that did NOT match ("needy": only
//     they can test, revive or overwrite a later mode).  Stores go
//     out only for planes whose bits changed.  The state in HBM stays bit-identical to
//     updating all of it (state-parity tests).
#include "oatgpu_internal.h"
#include <hip/hip_ext.h>
#include <algorithm>
#ifdef OATGPU_MEASURE
#include <stdlib.h>
#endif

namespace oatgpu {

// The mixture of one pixel in registers.  TUP (r03): a mode's {variance, mean[3]} as ONE ext-vector -- a contiguous
// register tuple, the shape the 16-byte record loads and stores move: no repacking moves behind a load (and so no
// `s_waitcnt vmcnt(0)` inside each exec-masked load region: the record loads of a full lane are in flight together),
// none in front of a store.  Measured (profiles/r03b_k1_ab.txt, same box): everyday 4K model 143 -> 120 us per
// two-frame launch, 16 x 1080p 558 -> 477 us, one 1080p stream 39.0 -> 34.6 us -- with MORE vector instructions per
// wave (515 against 504): the kernel is bound by its chain of dependent memory round trips, not by its instruction
// count.  The streaming-load instantiations (dense models) keep scalar registers: with every lane loading every
// record, the tuple form ran the dense 4K launch at 324-329 us against 293-295 us at 8 waves/SIMD and 296-298 us held
// at 7 waves by unused LDS -- no gain over the scalar form's 291-296 us (profiles/r03b_k1_ab.txt).  GREY
// keeps scalars too (8-byte records; its tuple form is not written).
template <int CH, bool TUP> struct PxModel;
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CH>
struct PxModel<CH, false> {
    float w[kMaxMix];
    float v[kMaxMix];
    float m[kMaxMix][3];
};
template <>
struct PxModel<3, true> {
    float w[kMaxMix];
    f32x4 r[kMaxMix];            // {variance, mean[0..2]}
};
#define OATGPU_TUP(CH, NTLD) ((CH) == 3 && !(NTLD))
template <int CH> __device__ __forceinline__ float rv(const PxModel<CH, false> &s, int k) { return s.v[k]; }
template <int CH> __device__ __forceinline__ float rv(const PxModel<CH, true> &s, int k) { return s.r[k][0]; }
template <int CH> __device__ __forceinline__ void set_rv(PxModel<CH, false> &s, int k, float x) { s.v[k] = x; }
template <int CH> __device__ __forceinline__ void set_rv(PxModel<CH, true> &s, int k, float x) { s.r[k][0] = x; }
template <int C, int CH> __device__ __forceinline__ float rm(const PxModel<CH, false> &s, int k) { return C < CH ? s.m[k][C] : 0.f; }
template <int C, int CH> __device__ __forceinline__ float rm(const PxModel<CH, true> &s, int k) { return s.r[k][1 + C]; }
template <int C, int CH> __device__ __forceinline__ void set_rm(PxModel<CH, false> &s, int k, float x) { if (C < CH) s.m[k][C] = x; }
template <int C, int CH> __device__ __forceinline__ void set_rm(PxModel<CH, true> &s, int k, float x) { s.r[k][1 + C] = x; }

// running state of MOG2Invoker's per-pixel mode loop (OpenCV 3.1.0 bgfg_gaussmix2.cpp)
struct PxLoop {
    bool background, fits;
    int nmodes;        // the reference's `nmodes`: loop bound, shrinks when a mode is pruned
    float total;
};

template <int CH>
__device__ __forceinline__ void swap_up(PxModel<CH, false> &s, int i, unsigned &dvm)   // exchange modes i and i-1
{
    dvm |= (3u << (i - 1));
    // v_swap_b32 exchanges two VGPRs in one instruction (hipcc writes three moves for a swap through a temporary:
    // 15 instead of 5 vector instructions per exchanged mode -- the kernel is VALU-bound with two frames a launch,
    // and on a dense model every bubbling loop runs in every wave)
    // Measured (profiles/r02g_two_frames_ab.txt): everyday 4K model 130.9 -> 127.0 us per two-frame launch, 16 x 1080p
    // 523 -> 507 us; dense model unchanged once its instantiation is out of scratch.  (Round 2 believed the audited
    // two-frame instantiation NEEDED this form to be right; it only moved the register allocation away from st_rec's
    // wide-store hazard -- DESIGN.md 3b.  The asm operands pin registers, which is why the GREY two-frame
    // instantiations are compiled for 7 waves/SIMD: at 8 they went into scratch.)
    asm volatile("v_swap_b32 %0, %1" : "+v"(s.w[i]), "+v"(s.w[i - 1]));
    asm volatile("v_swap_b32 %0, %1" : "+v"(s.v[i]), "+v"(s.v[i - 1]));
#pragma unroll
    for (int c = 0; c < CH; ++c) asm volatile("v_swap_b32 %0, %1" : "+v"(s.m[i][c]), "+v"(s.m[i - 1][c]));
}
template <int CH>
__device__ __forceinline__ void swap_up(PxModel<CH, true> &s, int i, unsigned &dvm)
{
    dvm |= (3u << (i - 1));
    asm volatile("v_swap_b32 %0, %1" : "+v"(s.w[i]), "+v"(s.w[i - 1]));
#pragma unroll
    for (int c = 0; c < 4; ++c) {        // (elements of a register tuple: through temporaries the allocator coalesces)
        float p = s.r[i][c], q = s.r[i - 1][c];
        asm volatile("v_swap_b32 %0, %1" : "+v"(p), "+v"(q));
        s.r[i][c] = p; s.r[i - 1][c] = q;
    }
}

// a / b where the quotient needs none of the rescaling the compiler's IEEE division carries (v_div_scale x 2, v_div_fmas's
// scale, v_div_fixup): the same reciprocal and the same five fused steps in the same order, hence the same bits, whenever b
// and 1 / b are normal, |exponent(a) - exponent(b)| < 96 and a > 2^-103 -- the conditions under which v_div_scale_f32 leaves
// both operands alone and v_div_fixup_f32 passes the quotient through -- and whenever a == 0 and b is finite and positive
// (every step is then exactly 0).  8 vector instructions instead of 11; 13 such divisions a frame sit in the unrolled mode
// blocks, each issued by every wave that holds a single lane needing it.  Measured (profiles/r05d_fastdiv_ab.txt): dense 4K
// model 282 -> 270 us per two-frame launch, everyday one-frame launch 77.6 -> 76.4 us; the everyday two-frame launch does
// not move (19 of its 461 vector instructions per wave go, it is not bound by them alone).
// The launcher (launch_mog_fused) guarantees the operands: a rate is 0 or in [2^-40, 1], a pruning threshold at least
// 2^-60 and a model's weights 0 or in [2^-62, 4] (a model this library evolved; an imported one is checked) -- any other
// launch runs the instantiations that keep the compiler's division (FD = false: the audit ones).
template <bool FD>
__device__ __forceinline__ float div_inrange(float a, float b)
{
    if (!FD) return a / b;
    float r = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    float q = a * r;
    float t = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(t, r, q);
    t = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(t, r, q);
}
// alphaT / weight at a fit site (FROZEN == 1: reached only by lanes off the no-op path below -- the compiler's division)
template <int FROZEN>
__device__ __forceinline__ float rate_over_weight(float alphaT, float weight)
{
    return div_inrange<FROZEN == 0>(alphaT, weight);
}

// Iteration MODE of the mode loop on a register resident mixture.
// dvm: bit k set when mode k's variance/mean registers were written.
// CH = 3 (BGR) or 1 (GREY: the reference's generic-channel loops, which start their sums at 0.f --
// 0.f + d*d == d*d exactly, so the single-channel expressions below are the same numbers).
template <int CH, int MODE, bool TUP, int FROZEN = 0>        // FROZEN: 0 a launch that learns, 1 every rate of the launch is 0, 2 ask frz
__device__ __forceinline__ void mog2_mode(PxModel<CH, TUP> &s, PxLoop &c, float x0, float x1, float x2, const MogParams &P,
                                          float alphaT, float alpha1, float prune, unsigned &dvm, bool frz = false)
{
    if (MODE < c.nmodes) {                // nmodes shrinks when a mode is pruned, as in the reference loop
        float weight = alpha1 * s.w[MODE] + prune;
        bool fit_here = false;
        if (!c.fits) {
            const float var = rv(s, MODE);
            const float d0 = rm<0>(s, MODE) - x0;
            const float d1 = CH == 3 ? rm<1>(s, MODE) - x1 : 0.f;
            const float d2 = CH == 3 ? rm<2>(s, MODE) - x2 : 0.f;
            const float dist2 = CH == 3 ? d0 * d0 + d1 * d1 + d2 * d2 : d0 * d0;
            if (c.total < P.TB && dist2 < P.Tb * var) c.background = true;
            if (dist2 < P.Tg * var) {
                c.fits = true;
                fit_here = true;
                weight += alphaT;
                // FROZEN == 1 -- every rate of the launch is 0, over a PLAIN model (the launcher's guarantee: means finite,
                // below 2^20 and not -0.f, variances inside [varMin, varMax]; what a run of this kernel leaves, what an import is
                // checked for): with a positive finite weight k = 0 / weight is +0, k * d is a zero, mean - zero is the mean,
                // var + 0 * (dist2 - var) is var and the clamp leaves it -- the update is the identity, bit for bit, and is not
                // computed (11 + 4 vector instructions a fit site that found out, every frame, that nothing had changed:
                // two-frame 4K launch at rate 0 84.9 -> 79.9 us, 22.2 k -> 23.4 k fps, profiles/r05t_frozen_identity_update_ab.txt).
                // A lane whose weight is anything else takes the update.
                bool upd = true;
                if (FROZEN == 1) upd = !__builtin_amdgcn_classf(weight, 0x100 | 0x80);       // not (+normal | +denormal)
                if (upd) {
                const float k = rate_over_weight<FROZEN>(alphaT, weight);
                const float o0 = rm<0>(s, MODE), o1 = CH == 3 ? rm<1>(s, MODE) : 0.f, o2 = CH == 3 ? rm<2>(s, MODE) : 0.f;
                const float n0 = o0 - k * d0, n1 = CH == 3 ? o1 - k * d1 : 0.f, n2 = CH == 3 ? o2 - k * d2 : 0.f;
                set_rm<0>(s, MODE, n0);
                if (CH == 3) { set_rm<1>(s, MODE, n1); set_rm<2>(s, MODE, n2); }
                float varnew = var + k * (dist2 - var);
                // MAX(varnew, varMin) then MIN(.., varMax) with OpenCV's macros -- MAX(a,b) ((a) < (b) ? (b) : (a)),
                // MIN(a,b) ((a) > (b) ? (b) : (a)): a comparison with a NaN is false, so a NaN variance (k = 0 / 0: a pruned slot
                // matched again at learning rate 0) STAYS NaN.  v_max_f32 / v_min_f32 would return the bound instead (IEEE maxNum);
                // gfx950's v_maximum3_f32 / v_minimum3_f32 propagate the NaN like the macros, at the same one instruction each.
                // [OCV-mem]: the macro definitions are recalled, not read (no OpenCV here) -- tests/test_opencv_crosscheck.py
                // test_mog2_nan_variance_clamp settles it where a cv2 exists; the oracle (oracle/mog2.c) follows the same reading.
                varnew = __builtin_elementwise_maximum(varnew, P.varMin);
                varnew = __builtin_elementwise_minimum(varnew, P.varMax);
                set_rv(s, MODE, varnew);
                // The record goes back to memory when it was written.  At learning rate 0 -- Oat's default, a frozen model --
                // k is 0 and the update leaves every bit as it was (unless the variance sat outside its clamp or the model
                // holds non-finite values): only a record whose bits DID change is marked, so a frozen model is read-only
                // (16 of the 43 B/px a two-frame launch moved at rate 0 were such stores).  FROZEN: the instantiations the
                // launcher picks when every rate of the launch is 0 -- as a run-time test of the rate the compiler turned it
                // into ~10 more vector instructions at every fit site of the launches that learn.
                // (2: the traffic-audit instantiations, which must count what the product kernel of the same launch moves)
                bool dirty = true;
                // (2: the audit follows the LAUNCHER's choice -- frz = MogLaunch::audit_frozen -- not this frame's rate: on a dense
                // model, or when only one frame of a pair is at rate 0, the product kernel stores every fitted record)
                if (FROZEN == 1 || (FROZEN == 2 && frz))
                    dirty = __float_as_uint(n0) != __float_as_uint(o0) || __float_as_uint(n1) != __float_as_uint(o1) ||
                            __float_as_uint(n2) != __float_as_uint(o2) || __float_as_uint(varnew) != __float_as_uint(var);
                if (dirty) dvm |= (1u << MODE);
                }
                // The reference bubbles the OLD weight up and then stores the new one
                // into the final slot; carrying the new weight along is the same state.
                s.w[MODE] = weight;
                bool moving = true;
#pragma unroll
                for (int i = MODE; i > 0; --i) {
                    if (moving) {
                        if (weight < s.w[i - 1]) moving = false;
                        else swap_up<CH>(s, i, dvm);
                    }
                }
            }
        }
        if (!fit_here) {
            // (a matched mode has weight >= alpha*(1-CT) > -prune for CT < 0.5 -- enforced by
            // oatgpu_create -- so it is never pruned)
            if (weight < -prune) { weight = 0.f; c.nmodes--; }
            s.w[MODE] = weight;
        }
        c.total += weight;
    }
}

// Everything behind the mode loop: renormalisation, `nmodes = nNewModes;`, the new mode, the mask.
// nentry: mode count at entry (the reference's nNewModes).  Returns the foreground-mask value
// {0, shadowVal, 255}; nmodes_out: the count to store; wchg: weights may differ from what was loaded
// (false only when alpha == 0 and the renormalisation was by exactly 1).
// shadow_matters (wave-uniform): whether anything downstream can tell shadowVal from 255.  detectShadowGMM only picks
// the VALUE of a foreground pixel's mask entry; the model does not depend on it, and Oat keeps shadow and foreground
// pixels alike (`frame.setTo(0, mask == 0)`, BackgroundSubtractorMOG.cpp:125).  Unless the caller asked for the mask
// bytes (oatgpu_mog_apply / oatgpu_mog_filter's mask tap) or shadowVal is 0 -- a shadow pixel would then BE background --
// the test is dead code: the fused path returns 255 for every foreground pixel without walking the modes.  (On a dense
// model a few lanes of most waves leave the background and every wave walked detectShadowGMM's five modes for them:
// ~500 of the streaming-load instantiation's 1 266 vector instructions a wave, profiles/r03n_dead_shadow_test.txt.)
template <int CH, bool TUP, int FROZEN>
__device__ __forceinline__ int mog2_finish(PxModel<CH, TUP> &s, PxLoop &c, int nentry, int &nmodes_out, float x0, float x1,
                                           float x2, const MogParams &P, float alphaT, float alpha1, unsigned &dvm,
                                           bool &wchg, bool shadow_matters)
{
    int nmodes = c.nmodes;
    // renormalise
    const float inv = div_inrange<FROZEN != 2>(1.f, c.total);
    wchg = (alphaT > 0.f) || (inv != 1.f);
#pragma unroll
    for (int mode = 0; mode < kMaxMix; ++mode)
        if (mode < nmodes) s.w[mode] *= inv;

    // `nmodes = nNewModes;`: pruned modes keep their slot (weight 0), the count never shrinks
    if (P.restoreCount) nmodes = nentry;

    // new mode
    if (!c.fits && alphaT > 0.f) {
        const int mode = (nmodes == P.nmix) ? P.nmix - 1 : nmodes++;
        const bool first = (nmodes == 1);
#pragma unroll
        for (int i = 0; i < kMaxMix; ++i) {
            if (!first && i < nmodes - 1) s.w[i] *= alpha1;
            if (i == mode) {
                dvm |= (1u << i);
                s.w[i] = first ? 1.f : alphaT;
                set_rv(s, i, P.varInit);
                set_rm<0>(s, i, x0);
                if (CH == 3) { set_rm<1>(s, i, x1); set_rm<2>(s, i, x2); }
            }
        }
        bool moving = true;
#pragma unroll
        for (int i = kMaxMix - 1; i > 0; --i) {
            if (moving && i <= nmodes - 1) {
                if (alphaT < s.w[i - 1]) moving = false;
                else swap_up<CH>(s, i, dvm);
            }
        }
    }
    nmodes_out = nmodes;

    if (c.background) return 0;
    int mask = 255;
    if (P.detectShadows && shadow_matters) {
        // detectShadowGMM
        float tW = 0.f;
        bool done = false;
#pragma unroll
        for (int mode = 0; mode < kMaxMix; ++mode) {
            if (!done && mode < nmodes) {
                const float m0 = rm<0>(s, mode), m1 = CH == 3 ? rm<1>(s, mode) : 0.f, m2 = CH == 3 ? rm<2>(s, mode) : 0.f;
                const float num = CH == 3 ? x0 * m0 + x1 * m1 + x2 * m2 : x0 * m0;
                const float den = CH == 3 ? m0 * m0 + m1 * m1 + m2 * m2 : m0 * m0;
                if (den == 0.f) {
                    done = true;
                } else {
                    if (num <= den && num >= P.tau * den) {
                        const float a = num / den;
                        const float e0 = a * m0 - x0, e1 = a * m1 - x1, e2 = a * m2 - x2;
                        const float dist2a = CH == 3 ? e0 * e0 + e1 * e1 + e2 * e2 : e0 * e0;
                        if (dist2a < P.Tb * rv(s, mode) * a * a) { mask = P.shadowVal; done = true; }
                    }
                    if (!done) {
                        tW += s.w[mode];
                        if (tW > P.TB) done = true;
                    }
                }
            }
        }
    }
    return mask;
}

// RGB2HSV_b tables: sdiv[i] = cvRound((255<<12)/i), hdiv[i] = cvRound((180<<12)/(6 i)).
// Neither quotient ever lands on .5 (255<<12 = 2^12*255, 180<<12/6 = 2^13*15), so
// round-half-even == floor(q + 1/2) == floor((2n + i) / (2i)).  The quotient is taken in fp32:
// numerator < 2^22 and denominator <= 510 are exact floats, and the exact quotient is at least 1/(2i) away from
// the next integer, so floor() of an fp32 quotient that is good to a few ulp is exact (hsv_sdiv / hsv_hdiv below;
// checked against the integer form for all 255 entries in tests/test_abi_exports.py).
__device__ __forceinline__ int hsv_sdiv(int i);
__device__ __forceinline__ int hsv_hdiv(int i);
__device__ __forceinline__ void hsv_tables_init(int *sdiv, int *hdiv)
{
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        sdiv[i] = hsv_sdiv(i);
        hdiv[i] = hsv_hdiv(i);
    }
}

__device__ __forceinline__ void bgr2hsv_px(int b, int g, int r, const int *sdiv, const int *hdiv,
                                           int &h, int &s, int &v)
{
    v = max(b, max(g, r));
    const int vmin = min(b, min(g, r));
    const int diff = v - vmin;
    s = (diff * sdiv[v] + (1 << 11)) >> 12;
    int hh = (v == r) ? (g - b) : (v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff);
    hh = (hh * hdiv[diff] + (1 << 11)) >> 12;   // arithmetic shift, as the reference
    hh += hh < 0 ? 180 : 0;
    h = hh;
}

// The same table entries computed on the spot (hsv_tables_init's exact fp32 quotients): K1 needs two entries per
// FOREGROUND pixel, and building both 256-entry tables per 256-pixel workgroup costs every pixel two quotients
// plus LDS traffic and a barrier -- the inline form costs foreground pixels the same and background pixels nothing.
// (r05) the quotient as numerator * v_rcp_f32(denominator), 2 vector instructions instead of the 11 of an IEEE division:
// v_rcp_f32 is good to 1 ulp and the product rounds once more, so the result is within 1.5 * 2^-23 of the exact quotient
// q <= 2^20 / i + 1/2, i.e. off by < 0.19 / i, while q lies at least 1 / (2i) away from the next integer (above): floor()
// still lands on the same integer.  tests/test_abi_exports.py checks every entry with the reciprocal off by one ulp either
// way, test_bgr2hsv_exhaustive_256cubed runs all 2^24 colours through tables built with these very functions.
__device__ __forceinline__ int hsv_sdiv(int i) { return i ? (int)floorf((float)(2 * (255 << 12) + i) * __builtin_amdgcn_rcpf((float)(2 * i))) : 0; }
__device__ __forceinline__ int hsv_hdiv(int i) { return i ? (int)floorf((float)(2 * ((180 << 12) / 6) + i) * __builtin_amdgcn_rcpf((float)(2 * i))) : 0; }
__device__ __forceinline__ void bgr2hsv_inline(int b, int g, int r, int &h, int &s, int &v)
{
    v = max(b, max(g, r));
    const int vmin = min(b, min(g, r));
    const int diff = v - vmin;
    s = (diff * hsv_sdiv(v) + (1 << 11)) >> 12;
    int hh = (v == r) ? (g - b) : (v == g) ? (b - r + 2 * diff) : (r - g + 4 * diff);
    hh = (hh * hsv_hdiv(diff) + (1 << 11)) >> 12;   // arithmetic shift, as the reference
    hh += hh < 0 ? 180 : 0;
    h = hh;
}

__device__ __forceinline__ bool in_range3(int a, int b, int c, const RangeParams &rp)
{
    return a >= rp.lo[0] && a <= rp.hi[0] && b >= rp.lo[1] && b <= rp.hi[1] &&
           c >= rp.lo[2] && c <= rp.hi[2];
}

// Traffic audit (oatgpu_traffic_audit): the SAME kernel with every load / store predicate also counted --
// bytes the lanes ask for ("useful": what the arithmetic can depend on) and the 32-byte sectors / 64-byte
// half lines those requests touch (what the memory system has to move at least).  A separate template
// instantiation: the product kernel carries none of it.
template <bool AUDIT>
struct Audit {
    unsigned lane_rd = 0, lane_wr = 0;          // bytes requested by this lane
    unsigned s32_rd = 0, s32_wr = 0, s64_rd = 0, s64_wr = 0;   // wave totals (identical in all lanes)
    // one 4-byte access per lane, lanes consecutive in memory: sector j = lanes 8j..8j+7
    __device__ __forceinline__ void dword(bool m, bool wr)
    {
        if (!AUDIT) return;
        const u64 bal = __ballot(m);
        u64 b = bal; b |= b >> 4; b |= b >> 2; b |= b >> 1; b &= 0x0101010101010101ull;
        u64 h = bal; h |= h >> 8; h |= h >> 4; h |= h >> 2; h |= h >> 1; h &= 0x0001000100010001ull;
        const unsigned n32 = 32u * (unsigned)__popcll(b), n64 = 64u * (unsigned)__popcll(h);
        if (wr) { lane_wr += m ? 4u : 0u; s32_wr += n32; s64_wr += n64; }
        else { lane_rd += m ? 4u : 0u; s32_rd += n32; s64_rd += n64; }
    }
    // NB-byte records of consecutive lanes (NB = 8 or 16): 32 / NB lanes per sector, 64 / NB per half line
    template <int NB>
    __device__ __forceinline__ void rec(bool m, bool wr)
    {
        if (!AUDIT) return;
        const u64 bal = __ballot(m);
        u64 b = bal, h = bal;
        if (NB == 16) { b |= b >> 1; b &= 0x5555555555555555ull; h |= h >> 2; h |= h >> 1; h &= 0x1111111111111111ull; }
        else { b |= b >> 2; b |= b >> 1; b &= 0x1111111111111111ull; h |= h >> 4; h |= h >> 2; h |= h >> 1; h &= 0x0101010101010101ull; }
        const unsigned n32 = 32u * (unsigned)__popcll(b), n64 = 64u * (unsigned)__popcll(h);
        if (wr) { lane_wr += m ? (unsigned)NB : 0u; s32_wr += n32; s64_wr += n64; }
        else { lane_rd += m ? (unsigned)NB : 0u; s32_rd += n32; s64_rd += n64; }
    }
    // one byte per lane (the counter plane): the wave's 64 bytes are two sectors / one half line
    __device__ __forceinline__ void byte(bool m, bool wr)
    {
        if (!AUDIT) return;
        const u64 bal = __ballot(m);
        const unsigned n32 = 32u * ((bal & 0xffffffffull ? 1u : 0u) + (bal >> 32 ? 1u : 0u)), n64 = bal ? 64u : 0u;
        if (wr) { lane_wr += m ? 1u : 0u; s32_wr += n32; s64_wr += n64; }
        else { lane_rd += m ? 1u : 0u; s32_rd += n32; s64_rd += n64; }
    }
    // nb bytes per lane of a packed run (the frame): 64*nb contiguous bytes per wave
    __device__ __forceinline__ void run(bool m, unsigned nb, bool wr)
    {
        if (!AUDIT) return;
        const unsigned lanes = (unsigned)__popcll(__ballot(m)), bytes = lanes * nb;
        const unsigned n32 = (bytes + 31u) / 32u * 32u, n64 = (bytes + 63u) / 64u * 64u;
        if (wr) { lane_wr += m ? nb : 0u; s32_wr += n32; s64_wr += n64; }
        else { lane_rd += m ? nb : 0u; s32_rd += n32; s64_rd += n64; }
    }
    __device__ __forceinline__ void flush(unsigned long long *out, int lane, bool valid_px)
    {
        if (!AUDIT) return;
        unsigned rd = lane_rd, wr = lane_wr;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { rd += __shfl_xor(rd, o); wr += __shfl_xor(wr, o); }
        const unsigned px = (unsigned)__popcll(__ballot(valid_px));
        if (lane == 0) {
            atomicAdd(out + 0, (unsigned long long)px);
            atomicAdd(out + 1, (unsigned long long)rd); atomicAdd(out + 2, (unsigned long long)wr);
            atomicAdd(out + 3, (unsigned long long)s32_rd); atomicAdd(out + 4, (unsigned long long)s32_wr);
            atomicAdd(out + 5, (unsigned long long)s64_rd); atomicAdd(out + 6, (unsigned long long)s64_wr);
        }
    }
};

// Measured and rejected in round 2 (tools/rejected/): (i) compacting the lanes that need phase 2
// through an LDS queue of the workgroup so that three of four waves retire after one round trip -- 135-163 us
// instead of 115 us at 4K; (ii) the same across two kernels (K1a defers <= 8 such lanes per wave through a lane
// mask, K1b takes them compacted) -- K1a alone ran no faster than the whole one-kernel form (92 us): the kernel is
// not bound by its second round trip but by the memory system (5.2 TB/s of real traffic on a device whose plain
// copy reaches 6.6), and K1b added 50 us (profiles/r02_k1_split_rejected_kernel_trace.md).
//
// Two frames a launch (NF == 2, "temporal fusion").  The model update is a recurrence per pixel; when the caller has
// two consecutive frames of a stream in hand (the pipelined entry points with a ring of results, a recorded
// sequence), ONE pass over the model takes both: the mixture stays in registers between the frames, so the 44 B/px an
// everyday model moves per frame (202 B/px a dense one) are moved once per TWO frames and only the 3 B/px of the
// second frame are added.  Same arithmetic in the same order -- the model after the launch and both threshold
// images are bit-identical to two single-frame launches (state-parity tests run both forms).  What the second
// frame needs beyond the first's registers: a lane that matched mode 0 as background in frame 1 never loaded the
// records of its later slots; if it is a "full" lane in frame 2 it loads them then (they are untouched: such a lane
// changes slot 0 only).  Weights are always current in registers (a slot not loaded was dead = 0 and stays 0).
// Waves per SIMD the two-frame instantiations are compiled for.  The streaming-load one (dense models) needs 72
// registers to stay out of scratch: at 8 waves (64 registers, 20-40 bytes of scratch a lane) the dense 4K launch took
// 343-351 us, at 7 waves 285-293 us, at 6 waves 303-308 us (gpurun_out/ab_nt.txt).
#define OATGPU_AUDIT_WAVES 4
#define OATGPU_NT2_WAVES 7
#define OATGPU_F2_WAVES 8
// ... and the one-frame instantiations (r04, profiles/r04a_k1_one_frame_waves_spills_ab.txt, r04c_k1_dense_one_vs_two_frames_ab.txt).
// Streaming loads (dense models): compiled for "7 waves" the instantiation has 90 scalar registers and NO spill (13 at 8) at
// the same 60 vector registers -- the hardware still runs it 8 waves a SIMD (96 scalar registers a wave is what 8 waves
// leave) -- and the same launch time within the noise: 271.1-271.4 against 271.8-272.6 us on one box, 320-326 against
// 315-323 us on another.  Held down to 7 / 6 / 5 waves a SIMD by unused LDS: 272 / 273-311 / 280-304 us, 312-322 / 318-321 us --
// occupancy is not what separates it from its two-frame sibling (286-294 us on that second box) either.
// Default policy (everyday models): compiled for 7 waves the 93 + 6 scalar registers DO cost the eighth wave (112 allocated):
// 76.2 -> 81.5 us at 4K, and 88.3 us at 6 waves -- the everyday launch lives on its occupancy; left at 8 with its 5 spills.
#define OATGPU_NT1_WAVES 7
#define OATGPU_F1_WAVES 8
template <int CH, bool AUDIT, bool NTLD, int NF>
constexpr int k1_waves()
{
    return AUDIT ? OATGPU_AUDIT_WAVES
         : NF == 2 ? ((NTLD || CH == 1) ? OATGPU_NT2_WAVES : OATGPU_F2_WAVES)
         : NTLD ? OATGPU_NT1_WAVES : OATGPU_F1_WAVES;
}
// Late arguments (r03).  The kernel's by-value arguments sit in scalar registers from the first instruction on, and at 8
// waves per SIMD a wave has 80 of them (800 per SIMD / 8, less the trap handler's 16): the two-frame BGR instantiation
// spilled 31 scalars to lanes of a vector register -- 61 v_writelane / v_readlane on a kernel bound by its vector
// instruction count.  What is only needed late (the second frame's rates and parameters, the inRange window, the
// threshold-word pointers) is therefore read where it is needed, by scalar loads from the kernarg segment through a
// pointer the compiler cannot see through (KRELOAD): SMEM instead of VALU work, and short live ranges.
typedef const MogLaunch __attribute__((address_space(4))) *KArgs;
typedef const char __attribute__((address_space(4))) *KBytes;
constexpr unsigned kMogLaunchArgOffset = (unsigned)((sizeof(Geom) + alignof(MogLaunch) - 1) / alignof(MogLaunch) * alignof(MogLaunch));
#define KRELOAD(ka) asm volatile("" : "+s"(ka))
__device__ __forceinline__ MogParams karg_mp(KArgs ka)
{
    MogParams P;
    P.Tb = ka->mp.Tb; P.TB = ka->mp.TB; P.Tg = ka->mp.Tg; P.varInit = ka->mp.varInit; P.varMin = ka->mp.varMin;
    P.varMax = ka->mp.varMax; P.tau = ka->mp.tau; P.nmix = ka->mp.nmix; P.detectShadows = ka->mp.detectShadows;
    P.shadowVal = ka->mp.shadowVal; P.restoreCount = ka->mp.restoreCount;
    return P;
}
__device__ __forceinline__ RangeParams karg_rp(KArgs ka)
{
    RangeParams r;
    r.lo[0] = ka->rp.lo[0]; r.lo[1] = ka->rp.lo[1]; r.lo[2] = ka->rp.lo[2];
    r.hi[0] = ka->rp.hi[0]; r.hi[1] = ka->rp.hi[1]; r.hi[2] = ka->rp.hi[2];
    return r;
}

// WG: threads a workgroup, 256 or 64.  The kernel has no LDS and no barrier, a wave is all that belongs together.  A
// 4-wave workgroup takes one wave slot on each SIMD of a compute unit and gets in only when all four have one: slots stand
// idle behind the slowest sibling (6.5 of 8 slots a SIMD are filled on average over a 4K launch, SQ_WAVE_CYCLES).  One wave a
// workgroup refills every slot at once: 4K two-frame launch 102.3 -> 98.9 us, 16 x 1080p 401 -> 374 us, one frame a launch
// 77.4 -> 75.2 us (profiles/r05f_k1_workgroup_size_ab.txt; 128 threads: 101.5 us, 512: 114.6 us) -- and starves the back
// half, whose 16-wave workgroup now finds four free slots on every SIMD of a compute unit only when a launch drains
// (k_blob_lds 110 -> 150 us beside K1 at 4K, 410 -> 750 us at 16 x 1080p): the pipeline gains only where the blob
// workgroup is dispatched early and parked, or where K1 is all that matters (a dense model): oatgpu_api.hip, launch_jobs.
template <int CH, bool AUDIT, bool NTLD, int NF, bool FROZEN = false, int WG = 256>
__global__ __launch_bounds__(WG, (k1_waves<CH, AUDIT, NTLD, NF>())) void k_mog_fused(Geom g, MogLaunch a, int first_stream)
{
    static_assert(WG == 256 || (WG == 64 && !AUDIT), "one wave a workgroup: the product instantiations only");
    static_assert(!FROZEN || (!AUDIT && !NTLD), "the frozen-model instantiations exist for the default-policy product kernels only");
    constexpr int kFrozenMode = AUDIT ? 2 : FROZEN ? 1 : 0;
    const bool audit_frz = AUDIT && a.audit_frozen != 0;
    // The audited two-frame instantiation exists for BGR only (GREY audits count one-frame launches: the library does
    // not pair GREY frames while an audit is on).  Round 2's "instantiation the compiler is touchy about" was the
    // wide-store data hazard of st_rec below, root-caused in round 3.
    static_assert(NF == 1 || !AUDIT || CH == 3, "the audited two-frame instantiation exists for BGR only");

    Audit<AUDIT> au;
    // (late arguments: the product two-frame BGR instantiation only -- the others are not bound by their instruction count
    // or were left as compiled)
    constexpr bool kLate = !NTLD && !AUDIT && CH == 3;
    KArgs ka = (KArgs)((KBytes)__builtin_amdgcn_kernarg_segment_ptr() + kMogLaunchArgOffset);
    // Nothing here spends vector instructions on what the scalar unit or the address path can do (the kernel was
    // 79 % VALU-busy at 370 VALU instructions per wave before, profiles/r02_k1_sq_counters_before.md): every plane
    // access is `buffer resource + one shared lane offset + scalar plane offset`, the row of a wave comes from a
    // scalar multiply-high, and the HSV quotients are taken only by foreground lanes.
    const int s = first_stream + blockIdx.y;
    const int lane = threadIdx.x & 63;
    const unsigned nwords = (unsigned)(g.Palloc >> 6);
    unsigned widx;                                // mask word (= 64-pixel tile) this lane works in
    unsigned lpos;                                // ... and its bit in it
    widx = blockIdx.x * (unsigned)(WG / 64) + (threadIdx.x >> 6);
    lpos = (unsigned)lane;
    const unsigned p = widx * kWavePx + lpos;
    const bool active = (int)(widx * kWavePx) < g.P;      // false: whole wave beyond the image (tail block)

    const size_t npx = (size_t)g.H * g.W;
    const uint8_t *frame = a.frames + (size_t)s * npx * CH;
    float *sbase = a.state + (size_t)s * mog_stream_floats(g.Palloc);       // uniform
    // Model access through ONE buffer resource over the stream's model: address = base + lane offset (a VGPR shared
    // by all modes: p*4 for the weights, p*16 for the {variance, mean} records) + mode offset (a scalar) -- no
    // vector instruction per access.  (As 64-bit flat addresses hipcc spent two VALU per access: ~100 of the 370
    // VALU instructions per wave.)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(sbase, 0, (int)(mog_stream_floats(g.Palloc) * 4), 0x00020000);
    const unsigned PA4 = (unsigned)g.Palloc * 4u;
    const unsigned voff_w = p * 4u, voff_r = p * 4u * (1 + CH);     // a stream's model stays below 4 GiB (oatgpu_create)
#define SW(k) ((unsigned)(k) * ((2 + CH) * PA4 + 8u * (unsigned)kPlanePad))
#define SR(k) (SW(k) + PA4 + 4u * (unsigned)kPlanePad)
// Cache policy of slots 1..4 (r02, tools/pmc_ab.sh on one box, 4K):
//   stores nontemporal, always: nothing changes on an everyday model (100.7 us, 52.3 B/px moved -- as without), a
//     dense model gains 2.6 %;
//   loads nontemporal too (NTLD, chosen per launch): a dense model gains another 6 % (sustained 339 -> 318 us:
//     everything streams, nothing is worth a cache line), an everyday model LOSES 5 % and moves 56 instead of 52 B/px
//     (the second mode of a flickering pixel and the weights of live slots ARE re-read next frame).  The host picks
//     NTLD from the model's density, which a sampling kernel measures every 64 frames (k_density_probe).
//   Mode 0 keeps the default policy either way.
// r03 (profiles/r03b_k1_ab.txt section 11, interleaved A/B on one box): on an EVERYDAY model mode 0 -- the bulk of the
// traffic, touched once per launch -- is loaded and stored with the streaming policy too, which leaves the caches to what the
// second round trip of a wave asks for (weights and records of live slots): 4K 114.2 -> 107.5 us per two-frame launch,
// 16 x 1080p 464.6 -> 425.3 us.  On a dense model (NTLD) it is the other way round, as r02 found: mode 0 default, slots 1..4
// streaming (mode 0 streaming there: 302.9 -> 305.6 us).  Loads only: worse (114.4 us); stores of slots 1..4 with the default
// policy: worse.
#define OATGPU_M0_LD ((CH == 3 && !NTLD) ? 2 : 0)
#define OATGPU_M0_ST ((CH == 3 && !NTLD) ? 2 : 0)
#define OATGPU_K_ST 2
// (the frames' bytes and the counter bytes with the streaming policy: no change either way, profiles/r03b_k1_ab.txt section 12)
#define LDW(k) __builtin_bit_cast(float, (k) == 0 ? __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_w, SW(k), OATGPU_M0_LD) : NTLD ? __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_w, SW(k), 2) : __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff_w, SW(k), 0))
#define STW(k, v) do { if ((k) >= 1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(v)), rsrc, voff_w, SW(k), OATGPU_K_ST); \
                       else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)(v)), rsrc, voff_w, SW(k), OATGPU_M0_ST); } while (0)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    // Records are LOADED through `uniform pointer + 32-bit lane offset` global loads.  Measured alternatives (r02,
    // gpurun ab12-ab17): 16-byte loads through the buffer resource -- 8 us slower at 4K on the sparse model, 3 %
    // faster on the dense one; branchless loads (disabled lanes present an out-of-range offset, the buffer unit
    // returns 0) -- no serialised waits any more, but 104 -> 135 us sparse: a vector memory instruction costs its
    // issue even when no lane touches memory (six all-out-of-range loads per wave: +15 us).
    // (NB: __builtin_bit_cast(float, q.y) on a vector ELEMENT reads element 0 in ROCm 7.2's clang -- use
    // __uint_as_float(q.y); that, not the builtin, made the 16-byte buffer load look broken earlier this round.)
    constexpr bool TUP = OATGPU_TUP(CH, NTLD);
    PxModel<CH, TUP> pm;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    auto ld_rec = [&](int k) {                                      // {variance, mean[CH]} of mode k
        const char *rp = (const char *)(sbase + mog_vm_off(g.Palloc, CH, k)) + voff_r;
        if constexpr (TUP) {
            // through the buffer resource: address = lane offset (a register the stores use anyway) + a scalar -- no vector
            // instruction per load (r02 measured this form slower in the scalar-register kernel, which was not yet bound by
            // its instruction count)
            if (k == 0) pm.r[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_r, SR(k), OATGPU_M0_LD));
            else pm.r[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff_r, SR(k), 0));
        } else if constexpr (CH == 3) {
            if (NTLD && k >= 1) {
                // (through the buffer resource as well: 292-298 us against 293-301 us on the dense 4K launch -- no change)
                const f32x4 q = __builtin_nontemporal_load((const f32x4 *)rp);
                pm.v[k] = q.x; pm.m[k][0] = q.y; pm.m[k][1] = q.z; pm.m[k][2] = q.w;
            } else {
                const float4 q = *(const float4 *)rp;
                pm.v[k] = q.x; pm.m[k][0] = q.y; pm.m[k][1] = q.z; pm.m[k][2] = q.w;
            }
        } else {
            const float2 q = *(const float2 *)rp;
            pm.v[k] = q.x; pm.m[k][0] = q.y;
        }
    };
    auto st_rec = [&](int k) {
        if constexpr (CH == 3) {
            u32x4 q;
            if constexpr (TUP) {
                q = __builtin_bit_cast(u32x4, pm.r[k]);
            } else {
                q.x = __builtin_bit_cast(unsigned, pm.v[k]); q.y = __builtin_bit_cast(unsigned, pm.m[k][0]);
                q.z = __builtin_bit_cast(unsigned, pm.m[k][1]); q.w = __builtin_bit_cast(unsigned, pm.m[k][2]);
            }
            if (k >= 1) __builtin_amdgcn_raw_buffer_store_b128(q, rsrc, voff_r, SR(k), OATGPU_K_ST);
            else __builtin_amdgcn_raw_buffer_store_b128(q, rsrc, voff_r, SR(k), OATGPU_M0_ST);
            // gfx950 wide-store data hazard (tools/store_hazard_repro.hip): a store of more than 64
            // bits reads its data VGPRs for a few cycles after issue, and hipcc (ROCm 7.2) -- which takes a buffer
            // store with an SGPR soffset to be exempt -- may overwrite one of them in the very next instruction.
            // That is what made the audited two-frame instantiation store `16` (its byte counter's increment) as the
            // mean / variance of lanes 12..15 of every 16 in round 2.  Two wait states with the data registers still
            // live behind every record store; tools/isa_hazard_check.py verifies the shipped binary.
            asm volatile("s_nop 1" :: "v"(q));
        } else {
            u32x2 q;
            q.x = __builtin_bit_cast(unsigned, rv(pm, k)); q.y = __builtin_bit_cast(unsigned, rm<0>(pm, k));
            __builtin_amdgcn_raw_buffer_store_b64(q, rsrc, voff_r, SR(k), 0);
        }
    };
    auto pin_rec = [&](int k, bool with_w) {       // an empty asm that uses the registers: pins the wait for their loads here
        if (with_w) asm volatile("" : "+v"(pm.w[k]));
        if constexpr (TUP) asm volatile("" : "+v"(pm.r[k]));
        else if constexpr (CH == 3) asm volatile("" : "+v"(pm.v[k]), "+v"(pm.m[k][0]), "+v"(pm.m[k][1]), "+v"(pm.m[k][2]));
        else asm volatile("" : "+v"(pm.v[k]), "+v"(pm.m[k][0]));
    };
    uint8_t *nmbase = a.nmodes + (size_t)s * g.Palloc;
#ifdef OATGPU_CUT                    // measurement only (tools: make variant DEFS=-DOATGPU_CUT=n): end the kernel at cut n with everything
                                     // computed so far kept alive through one store -- SQ_INSTS_VALU of the variants gives the
                                     // dynamic instruction count of every region (results are invalid)
#define CUT(n) do { if (OATGPU_CUT == (n)) { float acc_ = 0.f; _Pragma("unroll") for (int k_ = 0; k_ < kMaxMix; ++k_) \
        acc_ += pm.w[k_] + rv(pm, k_) + rm<0>(pm, k_) + rm<1>(pm, k_) + rm<2>(pm, k_); \
        ((float *)a.nmodes)[0] = acc_ + (float)(cut_extra_); return; } } while (0)
    int cut_extra_ = 0;
#else
#define CUT(n) do { } while (0)
#endif
    const unsigned coff = p;
    // a word's 64 pixels lie in one row (Wp is a multiple of 64): row = widx / words, by multiply-high
    const unsigned y = g.words == 1 ? widx : (unsigned)(((u64)widx * g.words_magic) >> 32);
    const unsigned x = (widx - y * (unsigned)g.words) * 64u + lpos;
    const bool valid = active && (x < (unsigned)g.W);
    const unsigned fi = (y * (unsigned)g.W + x) * CH;                        // frames stay below 4 GiB per stream
    const bool zero_in = kLate && in_range3(0, 0, 0, a.rp);      // a zeroed pixel is HSV (0,0,0): one uniform window test
#define AU_DW(m, wr) au.dword((m), (wr))
#define AU_REC(m, wr) au.template rec<4 * (1 + CH)>((m), (wr))
#define AU_B(m, wr) au.byte((m), (wr))

    // ---- phase 1: counter byte, mode 0, the pixel -- for every lane, nothing depends on anything ----
    // Everything starts at 0: a slot that is not loaded is dead, and its weight 0 takes part in the arithmetic.
    // (Leaving the {variance, mean} registers of slots >= 1 undefined instead -- read only by lanes that loaded or wrote
    // them -- saves 16 v_mov a launch: with scalar registers it cost 16 bytes of scratch and 14 %; with register tuples
    // it is free of scratch, parity-green, and changes the launch time by nothing (110.6 against 110.6 us): not adopted,
    // profiles/r03b_k1_ab.txt.)
#pragma unroll
    for (int k = 0; k < kMaxMix; ++k) {
        pm.w[k] = 0.f;
        set_rv(pm, k, 0.f); set_rm<0>(pm, k, 0.f); set_rm<1>(pm, k, 0.f); set_rm<2>(pm, k, 0.f);
    }
    int cnt = 0;
    int b = 0, gg = 0, r = 0;
    // (no `active` guard: the planes are allocated for Palloc pixels, a multiple of the block's 256)
    // (two-frame launches are never fresh and never write the masked frame or the mask bytes -- MogLaunch's contract:
    // their instantiations do not carry those arguments in scalar registers)
    // (kSingle also covers the streaming-load two-frame instantiation, left as it was: without these arguments the
    // allocator put it 12 bytes into scratch at its 72 registers)
    constexpr bool kSingle = NF == 1 || NTLD;
    const bool fresh = kSingle && a.fresh;
    const bool shadow_matters = (kSingle && a.out_mask != nullptr) || a.mp.shadowVal == 0;       // (uniform; mog2_finish)
    if (!fresh) {
        cnt = nmbase[coff];
        pm.w[0] = LDW(0);
        ld_rec(0);
        AU_B(active, false);
        AU_DW(active, false);
        AU_REC(active, false);
    }
    // no branch around the pixel loads either (lanes beyond the image read pixel 0 and ignore it)
    {
        const unsigned fj = valid ? fi : 0u;
        b = frame[fj];
        if (CH == 3) { gg = frame[fj + 1]; r = frame[fj + 2]; }
        au.run(valid, CH, false);
    }
    unsigned px2 = 0;                             // NF == 2: the second frame's pixel, packed b | g << 8 | r << 16
    if (NF == 2) {
        const uint8_t *frame2 = a.frames2 + (size_t)s * npx * CH;
        const unsigned fj = valid ? fi : 0u;
        px2 = frame2[fj];
        if (CH == 3) px2 |= (unsigned)frame2[fj + 1] << 8 | (unsigned)frame2[fj + 2] << 16;
        au.run(valid, CH, false);
    }
    if (!active) return;

    // `framefilt mask` placed before mog (FrameMasker.cpp:71-75: frame.setTo(0, roi_mask == 0)):
    // one bit per pixel, one word per wave.
    if (a.roi_bits) {
        const u64 rw = a.roi_bits[(size_t)s * nwords + widx];
        if (!((rw >> lpos) & 1ull)) { b = 0; gg = 0; r = 0; px2 = 0; }
    }

    const int nold = cnt & kCountMask;
    const float x0 = (float)b, x1 = (float)gg, x2 = (float)r;
    PxLoop lp{false, false, nold, 0.f};
    unsigned dvm = 0;           // modes whose variance/mean changed
    bool wchg = false;          // weights changed
    if (valid) mog2_mode<CH, 0, TUP, kFrozenMode>(pm, lp, x0, x1, x2, a.mp, a.alphaT, a.alpha1, a.prune, dvm, audit_frz);
    // A pixel that matched mode 0 as background never looks at another mode's variance/mean again this
    // frame (no fit test once fits is set, no shadow test on background, no new mode): what is left for
    // slots >= 1 is the weight decay of the live ones.  Everybody else is "full".
    const bool full = valid && !(lp.fits && lp.background);
#ifdef OATGPU_CUT
    cut_extra_ = (int)lp.fits + (int)lp.background + dvm;
#endif
    CUT(1);                          // phase 1 + mode 0 of frame 1

    const bool work = valid;

    // Two frames a launch: which lanes will be full in FRAME 2 although they are not in frame 1 -- known already (r03).
    // A lane that is not full has just matched its mode 0: that mode took its update above, it cannot move (mode 0 has
    // nowhere to bubble to), and nothing else of the lane's variances and means changes in frame 1.  Frame 2's mode-0
    // test reads exactly those registers (and `total` = 0), so its outcome is what this computes -- bit for bit, the
    // same expressions mog2_mode<0> evaluates a frame later.  Their records are then fetched in THIS phase 2 together
    // with frame 1's, and frame 2 has no memory round trip of its own: a wave's chain of dependent round trips is
    // three no longer, but two.
    // (Not in the streaming-load instantiations: on a dense model every lane is full in frame 1 anyway.)
    bool want2 = false;
    constexpr bool kEarly2 = NF == 2 && !NTLD;
    if (kEarly2) {
        const float z0 = (float)(px2 & 255u), z1 = (float)((px2 >> 8) & 255u), z2 = (float)(px2 >> 16);
        const float var = rv(pm, 0);
        const float e0 = rm<0>(pm, 0) - z0;
        const float e1 = CH == 3 ? rm<1>(pm, 0) - z1 : 0.f;
        const float e2 = CH == 3 ? rm<2>(pm, 0) - z2 : 0.f;
        const float dist2 = CH == 3 ? e0 * e0 + e1 * e1 + e2 * e2 : e0 * e0;
        const bool bg2 = 0.f < a.mp.TB && dist2 < a.mp.Tb * var;
        const bool fit2 = dist2 < a.mp.Tg * var;
        want2 = valid && !full && !(fit2 && bg2);
    }

    // ---- phase 2: slots 1..n-1 ----
    {
#pragma unroll
        for (int k = 1; k < kMaxMix; ++k) {
            const bool have = valid && k < nold;
            const bool lw = have && (cnt & (1 << (kLiveShift + k))) != 0;
            if (lw) pm.w[k] = LDW(k);
            const bool lr = have && (full || want2);
            if (lr) ld_rec(k);
            AU_DW(lw, false);
            AU_REC(lr, false);
        }
    }

    // Pin the completion of the phase-2 loads HERE, on every control-flow path: hipcc counts VMEM
    // operations conservatively across the exec-masked regions above, and without this the stores at
    // the end each carried an `s_waitcnt vmcnt(3)` -- gfx9 counts stores in vmcnt too, so a wave never
    // had more than four stores in flight.
#pragma unroll
    for (int k = 1; k < kMaxMix; ++k)
        pin_rec(k, true);

    // (r03, measured and rejected, profiles/r03b_k1_ab.txt: (i) skipping iteration k >= 1 on lanes that have fitted and
    // hold no live slot at k or behind it -- exact under `nmodes = nNewModes;` -- does not pay: the running count already
    // ends the walk two dead slots in, and the guards cost more vector instructions than they save -- also as ONE
    // wave-uniform branch round iterations 2..4 (+8 instructions a wave, no time gained); (ii) loading the
    // record of a live slot 1 also on lanes that matched, so that frame 2 seldom needs a round trip of its own: 122.1
    // against 121.8 us; (iv) warming the L2 for a later wave of the same XCD (the counter bytes of the word 2 048 / 8 192 /
    // 16 384 ahead ride with phase 1; behind the last load wait, byte loads touch that word's live slot-1 weight and record
    // sectors into a register nobody reads): 105.7 -> 112.5 us whatever the distance; (iii) a wave taking 2 or 4 mask words one after the other and carrying the NEXT word's counter
    // bytes along, so that everything the counter byte decides -- weights and records of live slots -- is loaded with
    // phase 1 instead of a round trip later: 118 us -> 172 us (2 words, 32 B of scratch), 213 us (4 words, 48 B),
    // 139 us (4 words at 7 waves/SIMD, 8 B) -- and the restructured source cost the streaming-load instantiation 20 B of
    // scratch and 15 % (298 -> 343 us): reverted.)
    CUT(2);                          // + phase 2 loads
    int mask = 0, nnew = nold;
    if (work) {
        mog2_mode<CH, 1, TUP, kFrozenMode>(pm, lp, x0, x1, x2, a.mp, a.alphaT, a.alpha1, a.prune, dvm, audit_frz);
        mog2_mode<CH, 2, TUP, kFrozenMode>(pm, lp, x0, x1, x2, a.mp, a.alphaT, a.alpha1, a.prune, dvm, audit_frz);
        mog2_mode<CH, 3, TUP, kFrozenMode>(pm, lp, x0, x1, x2, a.mp, a.alphaT, a.alpha1, a.prune, dvm, audit_frz);
        mog2_mode<CH, 4, TUP, kFrozenMode>(pm, lp, x0, x1, x2, a.mp, a.alphaT, a.alpha1, a.prune, dvm, audit_frz);
#ifdef OATGPU_CUT
        cut_extra_ = (int)lp.fits + (int)lp.background + dvm + lp.nmodes + (int)lp.total;
        CUT(3);                      // + modes 1..4 of frame 1
#endif
        mask = mog2_finish<CH, TUP, kFrozenMode>(pm, lp, nold, nnew, x0, x1, x2, a.mp, a.alphaT, a.alpha1, dvm, wchg, shadow_matters);
    }
#ifdef OATGPU_CUT
    cut_extra_ = mask + nnew + dvm + (int)wchg;
#endif
    CUT(4);                          // + finish of frame 1

    // frame.setTo(0, mask == 0): shadows (127) stay foreground
    if (mask == 0) { b = 0; gg = 0; r = 0; }
    if (kSingle && work && a.out_mask) a.out_mask[(size_t)(s - a.out_base) * npx + (size_t)y * g.W + x] = (uint8_t)mask;
    if (kSingle && work && a.out_bgr) {
        uint8_t *o = a.out_bgr + (size_t)(s - a.out_base) * npx * CH + fi;
        o[0] = (uint8_t)b;
        if (CH == 3) { o[1] = (uint8_t)gg; o[2] = (uint8_t)r; }
    }
    // A zeroed pixel is HSV (0,0,0) -- one uniform window test; only foreground lanes take the quotients,
    // and a wave without any skips the whole conversion.
    bool thr;
    if (CH == 3) {
        if (kLate) {
            thr = work && zero_in;
            if (mask != 0) {
                KArgs kb = ka;                               // (a copy: `ka` itself must stay uniform across this divergent region)
                KRELOAD(kb);
                const RangeParams rp = karg_rp(kb);
                int hh, ss, vv;
                bgr2hsv_inline(b, gg, r, hh, ss, vv);
                thr = work && in_range3(hh, ss, vv, rp);
            }
        } else {
            thr = work && in_range3(0, 0, 0, a.rp);
            if (mask != 0) {
                int hh, ss, vv;
                bgr2hsv_inline(b, gg, r, hh, ss, vv);
                thr = work && in_range3(hh, ss, vv, a.rp);
            }
        }
    } else {                                     // GREY chain: framefilt mog -> posidet thresh
        thr = work && b >= a.rp.lo[0] && b <= a.rp.hi[0];
    }

    bool full_any = full;
    int nlive = max(nold, nnew);
    if (NF == 2) {
        // ---- the second frame, on the registers the first one left ----
        const u64 word1 = __ballot(thr);
        if (kLate) KRELOAD(ka);
        u64 *const thr1 = kLate ? ka->thr_bits : a.thr_bits;
        if (thr1 && lane == 0) thr1[(size_t)s * nwords + widx] = word1;
        au.run(thr1 && lane == 0, 8, true);
        const MogParams mp2 = kLate ? karg_mp(ka) : a.mp;
        const float aT2 = kLate ? ka->alphaT2 : a.alphaT2, a12 = kLate ? ka->alpha12 : a.alpha12, pr2 = kLate ? ka->prune2 : a.prune2;
        b = (int)(px2 & 255u); gg = (int)((px2 >> 8) & 255u); r = (int)(px2 >> 16);
        const int nold2 = nnew;
        const float y0 = (float)b, y1 = (float)gg, y2 = (float)r;
        PxLoop lq{false, false, nold2, 0.f};
        bool wchg2 = false;
        if (valid) mog2_mode<CH, 0, TUP, kFrozenMode>(pm, lq, y0, y1, y2, mp2, aT2, a12, pr2, dvm, audit_frz);
        const bool full2 = valid && !(lq.fits && lq.background);
        if (!kEarly2) {
            // records this lane has not seen yet: it was not full in frame 1 (so its slots >= 1 are as in memory)
            const bool need2 = full2 && !full;
#pragma unroll
            for (int k = 1; k < kMaxMix; ++k) {
                const bool l2 = need2 && k < nold2;
                if (l2) ld_rec(k);
                AU_REC(l2, false);
            }
#pragma unroll
            for (int k = 1; k < kMaxMix; ++k)
                pin_rec(k, false);
        }
        int mask2 = 0, nnew2 = nold2;
        if (work) {
            mog2_mode<CH, 1, TUP, kFrozenMode>(pm, lq, y0, y1, y2, mp2, aT2, a12, pr2, dvm, audit_frz);
            mog2_mode<CH, 2, TUP, kFrozenMode>(pm, lq, y0, y1, y2, mp2, aT2, a12, pr2, dvm, audit_frz);
            mog2_mode<CH, 3, TUP, kFrozenMode>(pm, lq, y0, y1, y2, mp2, aT2, a12, pr2, dvm, audit_frz);
            mog2_mode<CH, 4, TUP, kFrozenMode>(pm, lq, y0, y1, y2, mp2, aT2, a12, pr2, dvm, audit_frz);
            mask2 = mog2_finish<CH, TUP, kFrozenMode>(pm, lq, nold2, nnew2, y0, y1, y2, mp2, aT2, a12, dvm, wchg2, shadow_matters);
        }
        if (mask2 == 0) { b = 0; gg = 0; r = 0; }          // frame.setTo(0, mask == 0)
        if (CH == 3) {
            if (kLate) {
                thr = work && zero_in;
                if (mask2 != 0) {
                    KArgs kb = ka;
                    KRELOAD(kb);
                    const RangeParams rp = karg_rp(kb);
                    int hh, ss, vv;
                    bgr2hsv_inline(b, gg, r, hh, ss, vv);
                    thr = work && in_range3(hh, ss, vv, rp);
                }
            } else {
                thr = work && in_range3(0, 0, 0, a.rp);
                if (mask2 != 0) {
                    int hh, ss, vv;
                    bgr2hsv_inline(b, gg, r, hh, ss, vv);
                    thr = work && in_range3(hh, ss, vv, a.rp);
                }
            }
        } else {
            thr = work && b >= a.rp.lo[0] && b <= a.rp.hi[0];
        }
        wchg = wchg || wchg2;
        full_any = full || full2;
        nlive = max(nlive, nnew2);
        nnew = nnew2;
    }

#ifdef OATGPU_CUT
    cut_extra_ = nnew + dvm + (int)wchg + (int)thr + nlive + (int)full_any;
#endif
    CUT(5);                          // + HSV / inRange of frame 1 and all of frame 2
    // ---- store back only what changed (values not stored are bit-identical in HBM) ----
    // Weights of slot k >= 1 can only have changed on a full lane (anything goes there) or where the
    // slot was live (decay / renormalisation); a dead slot on a matched lane was 0 and still is.
    int newcnt = nnew;
#pragma unroll
    for (int k = 0; k < kMaxMix; ++k) {
        const bool was_live = k == 0 || full_any || ((cnt >> (kLiveShift + k)) & 1);
        const bool sw = work && wchg && k < nlive && was_live, svm = (dvm >> k) & 1u;
        if (sw) STW(k, pm.w[k]);
        if (svm) st_rec(k);
        AU_DW(sw, true);
        AU_REC(svm, true);
        if (k >= 1 && k < nnew && __float_as_uint(pm.w[k]) != 0u) newcnt |= 1 << (kLiveShift + k);   // (bits: an imported -0.f is live)
    }
    const bool sc = work && (newcnt != cnt || fresh);
    if (sc) nmbase[coff] = (uint8_t)newcnt;
    AU_B(sc, true);

    const u64 word = __ballot(thr);
    if (kLate) KRELOAD(ka);
    u64 *const thr_out = kLate ? (NF == 2 ? ka->thr_bits2 : ka->thr_bits) : NF == 2 ? a.thr_bits2 : a.thr_bits;
    if (thr_out && lane == 0) thr_out[(size_t)s * nwords + widx] = word;
    au.run(thr_out && lane == 0, 8, true);
    au.flush(a.audit, lane, valid);
#ifdef OATGPU_RS_TIMING             // measurement builds (tools/rowscan_probe.py): when did the launch's LAST workgroups finish?  
    if (!AUDIT && a.rs_end && lane == 0 && blockIdx.x + 512u >= gridDim.x)
        __hip_atomic_fetch_max(a.rs_end, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
#undef CUT
#undef LDW
#undef STW
#undef SW
#undef SR
#undef AU_DW
#undef AU_REC
#undef AU_B
}

// ---- achievable-bandwidth probes: the best plain streaming kernels found on this chip ----
// (tools/k1_lab.hip, profiles/r02_k1_lab.txt: one 16-byte element per thread on a FULL grid with the
// streaming cache policy copies at 6.6 TB/s; the usual 2048-block grid-stride loop reaches 5.0)
typedef unsigned nv4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_read(const uint4 *src, size_t n, unsigned *sink)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const nv4 v = __builtin_nontemporal_load((const nv4 *)src + i);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x9e3779b9u) *sink = v.x;      // keeps the load alive, practically never taken
}
__global__ __launch_bounds__(256) void k_stream_copy(const uint4 *src, uint4 *dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    __builtin_nontemporal_store(__builtin_nontemporal_load((const nv4 *)src + i), (nv4 *)dst + i);
}
void launch_stream_read(const void *src, size_t n16, unsigned *sink, hipStream_t st)
{
    hipLaunchKernelGGL(k_stream_read, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (const uint4 *)src, n16, sink);
}
void launch_stream_copy(const void *src, void *dst, size_t n16, hipStream_t st)
{
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, (const uint4 *)src, (uint4 *)dst, n16);
}

// One camera's frame out of page-locked HOST memory (a registered shared-memory segment: every load is a PCIe read) into
// its staging slot, by a kernel instead of a DMA copy (oatgpu_set_stage_copy): a launch costs the host ~5 us where a
// hipMemcpyAsync costs ~20 us of set-up per 6 MB frame.  A small grid -- the waves do nothing but wait for the link, and
// the per-pixel kernel wants the slots: 4 x 16 bytes in flight per lane, 128 workgroups = 2 MB in flight.
__global__ __launch_bounds__(256) void k_stage_copy(const nv4 *src, nv4 *dst, size_t n16, const uint8_t *src_tail, uint8_t *dst_tail, int tail)
{
    const size_t stride = (size_t)gridDim.x * 1024;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n16; i += stride) {
        nv4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + (size_t)u * 256 < n16) v[u] = __builtin_nontemporal_load(src + i + (size_t)u * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i + (size_t)u * 256 < n16) __builtin_nontemporal_store(v[u], dst + i + (size_t)u * 256);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}
void launch_stage_copy(const void *src_dev_visible, void *dst, size_t bytes, hipStream_t st)
{
    const size_t n16 = bytes / 16;
    const int tail = (int)(bytes - n16 * 16);
    const unsigned blocks = (unsigned)std::min<size_t>(128, (n16 + 1023) / 1024 ? (n16 + 1023) / 1024 : 1);
    hipLaunchKernelGGL(k_stage_copy, dim3(blocks), dim3(256), 0, st, (const nv4 *)src_dev_visible, (nv4 *)dst, n16,
                       (const uint8_t *)src_dev_visible + n16 * 16, (uint8_t *)dst + n16 * 16, tail);
}

__global__ void k_nop() {}
void launch_nop(hipStream_t st) { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, st); }

template <int CH, bool AUDIT, bool NTLD, int NF, bool FROZEN, int WG>
static void launch_mog_wg(const Geom &g, const MogLaunch &a, int first_stream, int n_streams, hipStream_t st, hipEvent_t stop)
{
    const dim3 grid(g.Palloc / ((WG / 64) * kWavePx), n_streams);
    unsigned lds = 0;
#ifdef OATGPU_MEASURE                // (A/B builds only) OATGPU_K1_LDS=bytes: unused dynamic LDS per workgroup, which holds the
                                     // occupancy down -- 21 000 B = 7 workgroups a CU = 7 waves a SIMD, 26 000 = 6, 32 000 = 5
    static const unsigned env_lds = getenv("OATGPU_K1_LDS") ? (unsigned)atoi(getenv("OATGPU_K1_LDS")) : 0u;
    lds = env_lds;
#endif
    // stop != nullptr: the event rides on the dispatch packet's own completion signal (hipExtLaunchKernelGGL) -- no marker
    // packet of its own between this launch and the next one on the stream
    if (stop) hipExtLaunchKernelGGL((k_mog_fused<CH, AUDIT, NTLD, NF, FROZEN, WG>), grid, dim3(WG), lds, st, nullptr, stop, 0, g, a, first_stream);
    else hipLaunchKernelGGL((k_mog_fused<CH, AUDIT, NTLD, NF, FROZEN, WG>), grid, dim3(WG), lds, st, g, a, first_stream);
}
template <int CH, bool AUDIT, bool NTLD, int NF, bool FROZEN = false>
static void launch_mog_ch(const Geom &g, const MogLaunch &a, int first_stream, int n_streams, hipStream_t st, hipEvent_t stop, int wg)
{
    if constexpr (!AUDIT) {
        if (wg == 64) { launch_mog_wg<CH, AUDIT, NTLD, NF, FROZEN, 64>(g, a, first_stream, n_streams, st, stop); return; }
    }
    launch_mog_wg<CH, AUDIT, NTLD, NF, FROZEN, 256>(g, a, first_stream, n_streams, st, stop);
}

// div_inrange's operands (kernels above): a rate that is 0 or in [2^-40, 1] ...
static bool rate_in_range(float aT, float prune)
{
    if (aT == 0.f) return true;
    return aT >= 0x1p-40f && aT <= 1.f && -prune >= 0x1p-60f && -prune <= 0.5f * aT;       // (NaN fails every comparison)
}

static void launch_mog_pick(const Geom &g, const MogLaunch &a, int first_stream, int n_streams, hipStream_t st, hipEvent_t stop, int wg)
{
    if (a.frames2) {                     // two frames a launch (never fresh; audited for BGR only: the caller's business)
        if (a.audit) {
            launch_mog_ch<3, true, false, 2>(g, a, first_stream, n_streams, st, stop, wg);
        } else if (a.nt_loads) {
            if (a.channels == 1) launch_mog_ch<1, false, true, 2>(g, a, first_stream, n_streams, st, stop, wg);
            else launch_mog_ch<3, false, true, 2>(g, a, first_stream, n_streams, st, stop, wg);
        } else if (a.audit_frozen) {                           // a frozen model (Oat's default rate): mog2_mode, FROZEN
            if (a.channels == 1) launch_mog_ch<1, false, false, 2, true>(g, a, first_stream, n_streams, st, stop, wg);
            else launch_mog_ch<3, false, false, 2, true>(g, a, first_stream, n_streams, st, stop, wg);
        } else {
            if (a.channels == 1) launch_mog_ch<1, false, false, 2>(g, a, first_stream, n_streams, st, stop, wg);
            else launch_mog_ch<3, false, false, 2>(g, a, first_stream, n_streams, st, stop, wg);
        }
    } else if (a.audit) {                // (the audit counts bytes, not cache behaviour: default-policy loads)
        if (a.channels == 1) launch_mog_ch<1, true, false, 1>(g, a, first_stream, n_streams, st, stop, wg);
        else launch_mog_ch<3, true, false, 1>(g, a, first_stream, n_streams, st, stop, wg);
    } else if (a.nt_loads) {
        if (a.channels == 1) launch_mog_ch<1, false, true, 1>(g, a, first_stream, n_streams, st, stop, wg);
        else launch_mog_ch<3, false, true, 1>(g, a, first_stream, n_streams, st, stop, wg);
    } else if (a.audit_frozen) {
        if (a.channels == 1) launch_mog_ch<1, false, false, 1, true>(g, a, first_stream, n_streams, st, stop, wg);
        else launch_mog_ch<3, false, false, 1, true>(g, a, first_stream, n_streams, st, stop, wg);
    } else {
        if (a.channels == 1) launch_mog_ch<1, false, false, 1>(g, a, first_stream, n_streams, st, stop, wg);
        else launch_mog_ch<3, false, false, 1>(g, a, first_stream, n_streams, st, stop, wg);
    }
}

void launch_mog_fused(const Geom &g, const MogLaunch &a_in, int first_stream, int n_streams, hipStream_t st, hipEvent_t stop,
                      const MogLaunchOpts &o)
{
    MogLaunch a = a_in;
    // what the product path below would pick for this launch (the audit instantiations count that kernel's stores)
    // (frozen_ok: the configuration lets a plain model stay plain -- varMin <= varInit <= varMax; otherwise a launch at rate 0
    // runs the instantiations that learn, which are exact at any rate)
    a.audit_frozen = (o.frozen_ok && !a.nt_loads && a.alphaT == 0.f && (a.frames2 ? a.alphaT2 == 0.f : !a.fresh)) ? 1 : 0;
    // The product instantiations divide by div_inrange.  A launch whose operands it does not cover -- a rate below 2^-40 or
    // above 1, a pruning threshold below 2^-60 (complexity-reduction constant next to 0), a model that was imported with
    // weights no run of this kernel produces -- goes, one frame a launch, through the instantiations that keep the
    // compiler's division: the audit ones, counting into a sink of the context's.  Slower, and exact.
    const bool in_range = !o.wild_model && rate_in_range(a.alphaT, a.prune) && (!a.frames2 || rate_in_range(a.alphaT2, a.prune2));
    if (!a.audit && !in_range) {
        a.audit = o.wild_sink;
        if (a.frames2) {
            MogLaunch a1 = a, a2 = a;
            a1.frames2 = nullptr; a1.thr_bits2 = nullptr;
            a1.audit_frozen = (o.frozen_ok && !a1.nt_loads && a1.alphaT == 0.f && !a1.fresh) ? 1 : 0;
            a2.frames = a.frames2; a2.thr_bits = a.thr_bits2; a2.alphaT = a.alphaT2; a2.alpha1 = a.alpha12; a2.prune = a.prune2;
            a2.frames2 = nullptr; a2.thr_bits2 = nullptr;
            a2.audit_frozen = (o.frozen_ok && !a2.nt_loads && a2.alphaT == 0.f) ? 1 : 0;
            launch_mog_pick(g, a1, first_stream, n_streams, st, nullptr, 256);
            launch_mog_pick(g, a2, first_stream, n_streams, st, stop, 256);
            return;
        }
    }
    launch_mog_pick(g, a, first_stream, n_streams, st, stop, o.wg);
}

// Model density for the cache-policy choice: ONE workgroup samples 16 384 counter bytes of a coarse lattice over
// all streams and stores {sum of LIVE modes (1 + live hints), samples} to `out` (host-mapped; the host reads it
// whenever it next launches a probe -- a hint, never waited for; plain stores, no atomics on host memory).
__global__ __launch_bounds__(1024) void k_density_probe(const uint8_t *nmodes, size_t total, size_t stride, unsigned *out)
{
    __shared__ unsigned red[2][16];
    unsigned live = 0, n = 0;
    for (size_t j = threadIdx.x; j * stride < total; j += 1024) {
        const unsigned c = nmodes[j * stride];
        if (c & kCountMask) { live += 1u + (unsigned)__popc((c >> (kLiveShift + 1)) & 0xfu); n += 1; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { live += __shfl_xor(live, o); n += __shfl_xor(n, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = live; red[1][threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned a = 0, b = 0;
        for (int i = 0; i < 16; ++i) { a += red[0][i]; b += red[1][i]; }
        out[0] = a;
        out[1] = b;
    }
}
void launch_density_probe(const uint8_t *nmodes, size_t total, unsigned *out, hipStream_t st)
{
    const size_t samples = 16384;
    const size_t stride = total > samples ? total / samples : 1;
    hipLaunchKernelGGL(k_density_probe, dim3(1), dim3(1024), 0, st, nmodes, total, stride, out);
}

// ------------------------------------------------------------ small kernels --

__global__ __launch_bounds__(256) void k_bgr2hsv(const uint8_t *bgr, uint8_t *hsv, size_t npx)
{
    __shared__ int sdiv[256];
    __shared__ int hdiv[256];
    hsv_tables_init(sdiv, hdiv);
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        int h, s, v;
        bgr2hsv_px(bgr[3 * i], bgr[3 * i + 1], bgr[3 * i + 2], sdiv, hdiv, h, s, v);
        hsv[3 * i] = (uint8_t)h; hsv[3 * i + 1] = (uint8_t)s; hsv[3 * i + 2] = (uint8_t)v;
    }
}

void launch_bgr2hsv(const uint8_t *bgr, uint8_t *hsv, size_t npx, hipStream_t st)
{
    int blocks = (int)((npx + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_bgr2hsv, dim3(blocks), dim3(256), 0, st, bgr, hsv, npx);
}

// ---- the other entries of oat::color_conv_table (Color.h:45-51) behind ColorConvert::filter (ColorConvert.cpp:101-107) ----
// Four pixels a thread, every access a dword (4 px x 3 B = three dwords; 4 grey px = one); the last npx % 4 pixels
// are done byte-wise by the thread behind the last full group.

// COLOR_BGR2GRAY on 8U: RGB2Gray<uchar>'s table sums = (1868 B + 9617 G + 4899 R + 2^13) >> 14
__device__ __forceinline__ unsigned grey_of(unsigned b, unsigned g, unsigned r)
{
    return (1868u * b + 9617u * g + 4899u * r + (1u << 13)) >> 14;
}

// COLOR_HSV2BGR on 8U, OpenCV 3.1.0: HSV2RGB_b over HSV2RGB_f(hrange 180) -- fp32, every operation rounded on its own
// (-ffp-contract=off), cvRound (nearest even) + saturation on the way out.  h <= 255 -> h * (6/180) < 8.5: the
// reference's "do h -= 6 while (h >= 6)" runs once at most and the sector is always one of 0..5.
__device__ __forceinline__ void hsv2bgr_px(unsigned hh, unsigned ss, unsigned vv, unsigned &b, unsigned &g, unsigned &r)
{
    float h = (float)hh;
    const float s = (float)ss * (1.f / 255.f), v = (float)vv * (1.f / 255.f);
    float fb, fg, fr;
    if (s == 0.f) fb = fg = fr = v;
    else {
        h *= 6.f / 180.f;
        if (h >= 6.f) h -= 6.f;
        const int sector = (int)floorf(h);
        h -= (float)sector;
        const float t0 = v, t1 = v * (1.f - s), t2 = v * (1.f - s * h), t3 = v * (1.f - s * (1.f - h));
        switch (sector) {                       // sector_data[][3] of HSV2RGB_f, (b, g, r)
        case 0:  fb = t1; fg = t3; fr = t0; break;
        case 1:  fb = t1; fg = t0; fr = t2; break;
        case 2:  fb = t3; fg = t0; fr = t1; break;
        case 3:  fb = t0; fg = t2; fr = t1; break;
        case 4:  fb = t0; fg = t1; fr = t3; break;
        default: fb = t2; fg = t1; fr = t0; break;
        }
    }
    b = (unsigned)min(max(__float2int_rn(fb * 255.f), 0), 255);
    g = (unsigned)min(max(__float2int_rn(fg * 255.f), 0), 255);
    r = (unsigned)min(max(__float2int_rn(fr * 255.f), 0), 255);
}

template <int kCode>   // 0: BGR -> GREY, 1: GREY -> BGR, 2: HSV -> BGR
__global__ __launch_bounds__(256) void k_cvt_color(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, size_t npx)
{
    const size_t groups = npx >> 2;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < groups) {
        if (kCode == 1) {
            const unsigned q = ((const unsigned *)in)[t];
            const unsigned p0 = q & 255u, p1 = (q >> 8) & 255u, p2 = (q >> 16) & 255u, p3 = q >> 24;
            unsigned *o = (unsigned *)out + 3 * t;
            o[0] = p0 * 0x010101u | p1 << 24;
            o[1] = p1 * 0x0101u | p2 * 0x01010000u;
            o[2] = p2 | p3 * 0x01010100u;
        } else {
            const unsigned *w = (const unsigned *)in + 3 * t;
            const unsigned w0 = w[0], w1 = w[1], w2 = w[2];
            // channel c of pixel k is byte 3k + c of the twelve
            const unsigned a0 = w0 & 255u, a1 = (w0 >> 8) & 255u, a2 = (w0 >> 16) & 255u;
            const unsigned b0 = w0 >> 24, b1 = w1 & 255u, b2 = (w1 >> 8) & 255u;
            const unsigned c0 = (w1 >> 16) & 255u, c1 = w1 >> 24, c2 = w2 & 255u;
            const unsigned d0 = (w2 >> 8) & 255u, d1 = (w2 >> 16) & 255u, d2 = w2 >> 24;
            if (kCode == 0) {
                ((unsigned *)out)[t] = grey_of(a0, a1, a2) | grey_of(b0, b1, b2) << 8 | grey_of(c0, c1, c2) << 16 |
                                       grey_of(d0, d1, d2) << 24;
            } else {
                unsigned pb[4], pg[4], pr[4];
                hsv2bgr_px(a0, a1, a2, pb[0], pg[0], pr[0]);
                hsv2bgr_px(b0, b1, b2, pb[1], pg[1], pr[1]);
                hsv2bgr_px(c0, c1, c2, pb[2], pg[2], pr[2]);
                hsv2bgr_px(d0, d1, d2, pb[3], pg[3], pr[3]);
                unsigned *o = (unsigned *)out + 3 * t;
                o[0] = pb[0] | pg[0] << 8 | pr[0] << 16 | pb[1] << 24;
                o[1] = pg[1] | pr[1] << 8 | pb[2] << 16 | pg[2] << 24;
                o[2] = pr[2] | pb[3] << 8 | pg[3] << 16 | pr[3] << 24;
            }
        }
    } else if (t == groups) {
        for (size_t i = groups << 2; i < npx; ++i) {
            if (kCode == 0) out[i] = (uint8_t)grey_of(in[3 * i], in[3 * i + 1], in[3 * i + 2]);
            else if (kCode == 1) out[3 * i] = out[3 * i + 1] = out[3 * i + 2] = in[i];
            else {
                unsigned b, g, r;
                hsv2bgr_px(in[3 * i], in[3 * i + 1], in[3 * i + 2], b, g, r);
                out[3 * i] = (uint8_t)b; out[3 * i + 1] = (uint8_t)g; out[3 * i + 2] = (uint8_t)r;
            }
        }
    }
}

void launch_cvt_color(int code, const uint8_t *in, uint8_t *out, size_t npx, hipStream_t st)
{
    const unsigned blocks = (unsigned)(((npx >> 2) + 1 + 255) / 256);
    if (code == 0) hipLaunchKernelGGL(k_cvt_color<0>, dim3(blocks), dim3(256), 0, st, in, out, npx);
    else if (code == 1) hipLaunchKernelGGL(k_cvt_color<1>, dim3(blocks), dim3(256), 0, st, in, out, npx);
    else hipLaunchKernelGGL(k_cvt_color<2>, dim3(blocks), dim3(256), 0, st, in, out, npx);
}

// one wave = one mask word (64 pixels of one row)
__global__ __launch_bounds__(256) void k_inrange_bits(Geom g, const uint8_t *frame, int channels,
                                                      RangeParams rp, u64 *bits)
{
    const int lane = threadIdx.x & 63;
    const int word = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (word >= (g.Palloc >> 6)) return;
    const int p = word * 64 + lane;
    const int y = p / g.Wp, x = p - y * g.Wp;
    const bool valid = p < g.P && x < g.W;
    bool thr = false;
    if (valid) {
        const size_t i = (size_t)y * g.W + x;
        if (channels == 3) thr = in_range3(frame[3 * i], frame[3 * i + 1], frame[3 * i + 2], rp);
        else { const int v = frame[i]; thr = v >= rp.lo[0] && v <= rp.hi[0]; }
    }
    const u64 w = __ballot(thr);
    if (lane == 0) bits[word] = w;
}

void launch_inrange_bits(const Geom &g, const uint8_t *frame, int channels, const RangeParams &rp,
                         u64 *bits, hipStream_t st)
{
    const int nwords = g.Palloc >> 6;
    hipLaunchKernelGGL(k_inrange_bits, dim3((nwords + 3) / 4), dim3(256), 0, st, g, frame, channels, rp, bits);
}

// framefilt bsub: first frame -> background; alpha > 0: cv::accumulateWeighted in fp32
// (src*a + bg*b, a = (float)alpha, b = 1 - a) and convertTo(CV_8U) (round half even, saturate);
// then frame - background, saturating (BackgroundSubtractor.cpp:87-100).
__global__ __launch_bounds__(256) void k_bsub(const uint8_t *in, uint8_t *out, uint8_t *bg, float *bg_f, size_t n,
                                              float a, float b, int first, int learn)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int v = in[i];
        float f = first ? (float)v : bg_f[i];
        int g = first ? v : bg[i];
        if (learn) {
            f = v * a + f * b;
            g = min(255, max(0, __float2int_rn(f)));
        }
        if (first || learn) { bg_f[i] = f; bg[i] = (uint8_t)g; }
        out[i] = (uint8_t)(v > g ? v - g : 0);
    }
}

void launch_bsub(const uint8_t *in, uint8_t *out, uint8_t *bg, float *bg_f, size_t n, float a, float b, int first,
                 int learn, hipStream_t st)
{
    int blocks = (int)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_bsub, dim3(blocks), dim3(256), 0, st, in, out, bg, bg_f, n, a, b, first, learn);
}

// framefilt mask: frame.setTo(0, roi_mask == 0) (FrameMasker.cpp:71-75) with the 1-bit ROI plane
__global__ __launch_bounds__(256) void k_apply_roi(Geom g, const uint8_t *in, uint8_t *out, int ch, const u64 *roi)
{
    const size_t npx = (size_t)g.H * g.W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / g.W), x = (int)(i - (size_t)y * g.W);
        const int p = y * g.Wp + x;
        const bool keep = (roi[p >> 6] >> (p & 63)) & 1ull;
        for (int c = 0; c < ch; ++c) out[i * ch + c] = keep ? in[i * ch + c] : 0;
    }
}
void launch_apply_roi(const Geom &g, const uint8_t *in, uint8_t *out, int channels, const u64 *roi, hipStream_t st)
{
    const size_t npx = (size_t)g.H * g.W;
    int blocks = (int)((npx + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_apply_roi, dim3(blocks), dim3(256), 0, st, g, in, out, channels, roi);
}

// framefilt thresh: RGB2Gray<uchar> ((1868 B + 9617 G + 4899 R + 8192) >> 14), inRange, setTo(0)
__global__ __launch_bounds__(256) void k_thresh_filter(const uint8_t *in, uint8_t *out, size_t npx, int ch, int lo,
                                                       int hi)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        if (ch == 3) {
            const int b = in[3 * i], g = in[3 * i + 1], r = in[3 * i + 2];
            const int y = (1868 * b + 9617 * g + 4899 * r + (1 << 13)) >> 14;
            const bool keep = y >= lo && y <= hi;
            out[3 * i] = keep ? b : 0; out[3 * i + 1] = keep ? g : 0; out[3 * i + 2] = keep ? r : 0;
        } else {
            const int y = in[i];
            out[i] = (y >= lo && y <= hi) ? y : 0;
        }
    }
}

void launch_thresh_filter(const uint8_t *in, uint8_t *out, size_t npx, int channels, int lo, int hi, hipStream_t st)
{
    int blocks = (int)((npx + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_thresh_filter, dim3(blocks), dim3(256), 0, st, in, out, npx, channels, lo, hi);
}

// posidet diff: cv::absdiff + cv::threshold(THRESH_BINARY) of a GREY frame against the previous one
// (DifferenceDetector.cpp:156-161), one wave = one mask word; also refreshes the previous frame.
__global__ __launch_bounds__(256) void k_absdiff_bits(Geom g, const uint8_t *frame, uint8_t *last, int thr,
                                                      int have_last, u64 *bits)
{
    const int lane = threadIdx.x & 63;
    const int word = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (word >= (g.Palloc >> 6)) return;
    const int p = word * 64 + lane;
    const int y = p / g.Wp, x = p - y * g.Wp;
    const bool valid = p < g.P && x < g.W;
    bool on = false;
    if (valid) {
        const size_t i = (size_t)y * g.W + x;
        const int v = frame[i];
        if (have_last) { const int l = last[i]; on = (v > l ? v - l : l - v) > thr; }
        else on = v != 0;                       // first frame: threshold_frame_ = frame.clone()
        last[i] = (uint8_t)v;
    }
    const u64 w = __ballot(on);
    if (lane == 0) bits[word] = w;
}

void launch_absdiff_bits(const Geom &g, const uint8_t *frame, uint8_t *last, int thr, int have_last, u64 *bits,
                         hipStream_t st)
{
    const int nwords = g.Palloc >> 6;
    hipLaunchKernelGGL(k_absdiff_bits, dim3((nwords + 3) / 4), dim3(256), 0, st, g, frame, last, thr, have_last, bits);
}

__global__ __launch_bounds__(256) void k_unpack_bits(Geom g, const u64 *bits, uint8_t *out)
{
    const size_t npx = (size_t)g.H * g.W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / g.W), x = (int)(i - (size_t)y * g.W);
        const int p = y * g.Wp + x;
        out[i] = ((bits[p >> 6] >> (p & 63)) & 1ull) ? 255 : 0;
    }
}

void launch_pack_bits(const Geom &g, const uint8_t *in, u64 *bits, hipStream_t st)
{
    RangeParams rp;
    rp.lo[0] = 1; rp.hi[0] = 255; rp.lo[1] = rp.lo[2] = 0; rp.hi[1] = rp.hi[2] = 255;
    launch_inrange_bits(g, in, 1, rp, bits, st);     // nonzero == in [1,255]
}

void launch_unpack_bits(const Geom &g, const u64 *bits, uint8_t *out, hipStream_t st)
{
    size_t npx = (size_t)g.H * g.W;
    int blocks = (int)((npx + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_unpack_bits, dim3(blocks), dim3(256), 0, st, g, bits, out);
}

// ---- model checkpoint: device layout <-> OpenCV's logical AoS order ----
__device__ __forceinline__ float *w_elem(const Geom &g, float *sbase, int ch, int k, int p)
{
    return sbase + mog_w_off(g.Palloc, ch, k) + p;
}
__device__ __forceinline__ float *rec_elem(const Geom &g, float *sbase, int ch, int k, int p, int field)   // 0: variance, 1 + c: mean[c]
{
    return sbase + mog_vm_off(g.Palloc, ch, k) + (size_t)p * (1 + ch) + field;
}
__device__ __forceinline__ uint8_t *count_elem(const Geom &g, float *sbase, uint8_t *nmodes, int p)
{
    (void)g; (void)sbase;
    return nmodes + p;
}

__global__ __launch_bounds__(256) void k_state_export(Geom g, float *state, uint8_t *nmodes, int nmix, int ch,
                                                      uint8_t *modes_used, float *weight, float *variance,
                                                      float *mean)
{
    const size_t npx = (size_t)g.H * g.W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / g.W), x = (int)(i - (size_t)y * g.W);
        const int p = y * g.Wp + x;
        modes_used[i] = *count_elem(g, state, nmodes, p) & kCountMask;     // the live hints stay inside
        for (int k = 0; k < nmix; ++k) {
            weight[i * nmix + k] = *w_elem(g, state, ch, k, p);
            variance[i * nmix + k] = *rec_elem(g, state, ch, k, p, 0);
            for (int c = 0; c < ch; ++c)
                mean[(i * nmix + k) * ch + c] = *rec_elem(g, state, ch, k, p, 1 + c);
        }
    }
}

__global__ __launch_bounds__(256) void k_state_import(Geom g, float *state, uint8_t *nmodes, int nmix, int ch,
                                                      const uint8_t *modes_used, const float *weight,
                                                      const float *variance, const float *mean)
{
    const size_t npx = (size_t)g.H * g.W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / g.W), x = (int)(i - (size_t)y * g.W);
        const int p = y * g.Wp + x;
        int cnt = modes_used[i];
        for (int k = 1; k < nmix && k < (int)modes_used[i]; ++k)            // live hints of slots 1..n-1
            if (__float_as_uint(weight[i * nmix + k]) != 0u) cnt |= 1 << (kLiveShift + k);     // bits, not value: -0.f must be stored back as it came
        *count_elem(g, state, nmodes, p) = (uint8_t)cnt;
        for (int k = 0; k < nmix; ++k) {
            *w_elem(g, state, ch, k, p) = weight[i * nmix + k];
            *rec_elem(g, state, ch, k, p, 0) = variance[i * nmix + k];
            for (int c = 0; c < ch; ++c)
                *rec_elem(g, state, ch, k, p, 1 + c) = mean[(i * nmix + k) * ch + c];
        }
    }
}

void launch_state_export(const Geom &g, float *state, uint8_t *nmodes, int nmix, int channels,
                         uint8_t *modes_used, float *weight, float *variance, float *mean, hipStream_t st)
{
    hipLaunchKernelGGL(k_state_export, dim3(2048), dim3(256), 0, st, g, state, nmodes, nmix, channels, modes_used, weight,
                       variance, mean);
}

void launch_state_import(const Geom &g, float *state, uint8_t *nmodes, int nmix, int channels,
                         const uint8_t *modes_used, const float *weight, const float *variance, const float *mean,
                         hipStream_t st)
{
    hipLaunchKernelGGL(k_state_import, dim3(2048), dim3(256), 0, st, g, state, nmodes, nmix, channels, modes_used, weight,
                       variance, mean);
}

}  // namespace oatgpu
