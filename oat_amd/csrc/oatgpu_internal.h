// oatgpu_internal.h -- device-side layout contract shared by the kernel TUs and
// the C-ABI TU.  Not installed; the public surface is include/oatgpu.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oatgpu {

typedef unsigned long long u64;

// ---------------------------------------------------------------------------
// Geometry.  Every per-pixel device array of one camera stream lives in a
// PADDED index space: row pitch Wp = 64*ceil(W/64) pixels, so a 64-bit mask
// word never straddles two image rows.  p = y*Wp + x.  (For 640/1920/3840
// wide frames Wp == W and p is the plain raster index.)
// Arrays are allocated for Palloc = 1024*ceil(H*Wp/1024) entries so that the
// MOG kernel's 1024-pixel blocks need no tail handling on state planes.
// ---------------------------------------------------------------------------
struct Geom {
    int H, W, Wp;        // rows, cols, padded pitch
    int words;           // Wp / 64 mask words per row
    int P;               // H * Wp
    int Palloc;          // P rounded up to 1024
    int n_streams;
};

// MOG2 model layout in HBM (per stream): 25 fp32 planes + one u8 plane.
//   plane 0..4   weight[k]
//   plane 5..9   variance[k]
//   plane 10+3k+c  mean[k][c]
// Inside a plane the 256 pixels one wavefront owns are stored lane-interleaved:
//   pixel p = base + 64*j + lane   (base multiple of 64*kPX, j in 0..kPX-1)
//   slot    = base + kPX*lane + j
// so one vector load per lane fetches that lane's kPX pixels (16 B/lane for
// kPX = 4, 1 KiB per wave instruction) and pixels {base+64j .. base+64j+63} --
// one mask word -- sit in component j of the 64 lanes.
constexpr int kMogPlanes = 25;
constexpr int kMaxMix = 5;
#ifndef OATGPU_PX
#define OATGPU_PX 1
#endif
// Pixels per lane of the MOG kernel (1, 2 or 4).  Measured on MI355X at 4K: 4 px/lane (16-byte
// loads, 165 VGPRs, 3 waves/SIMD) 138-145 us; 2 px/lane (108 VGPRs, 4 waves) 144 us; 1 px/lane (62 VGPRs,
// 8 waves/SIMD, 4-byte loads) 113-123 us -- the kernel is latency-bound on its dependent load phases, so
// occupancy beats load width.
constexpr int kPX = OATGPU_PX;
constexpr int kWavePx = 64 * kPX;         // pixels one wavefront owns

__host__ __device__ inline int mog_slot(int p)
{
    int base = p - (p % kWavePx), r = p % kWavePx;
    return base + kPX * (r & 63) + (r >> 6);
}

// Where the planes live.  TILED (OATGPU_TILED=1, an A/B option; measured 3-4 % slower on MI355X): everything one wavefront needs is ONE contiguous
// ~26 KB record -- 25 fp32 planes of kWavePx entries, then kWavePx mode counters (bytes) --
// so a wave's 26 vector loads hit consecutive DRAM pages and one TLB entry instead of 26
// streams that are 33 MB apart.  PLANAR (default): plane-major arrays of Palloc
// entries, counters in a separate array.
#ifndef OATGPU_TILED
#define OATGPU_TILED 0
#endif
#if OATGPU_TILED
constexpr int kTileFloats = kMogPlanes * kWavePx + kWavePx / 4;
__host__ __device__ inline size_t mog_stream_floats(int Palloc) { return (size_t)(Palloc / kWavePx) * kTileFloats; }
// float offset (within a stream) of `plane` for the wave tile that starts at pixel `base`
__host__ __device__ inline size_t mog_plane_off(int Palloc, int plane, int base)
{
    (void)Palloc;
    return (size_t)(base / kWavePx) * kTileFloats + (size_t)plane * kWavePx;
}
__host__ __device__ inline size_t mog_plane_stride(int Palloc) { (void)Palloc; return kWavePx; }
// byte offset (within a stream's float array) of the mode counters of that tile
__host__ __device__ inline size_t mog_count_off(int Palloc, int base)
{
    (void)Palloc;
    return ((size_t)(base / kWavePx) * kTileFloats + (size_t)kMogPlanes * kWavePx) * 4;
}
#else
__host__ __device__ inline size_t mog_stream_floats(int Palloc) { return (size_t)kMogPlanes * Palloc; }
__host__ __device__ inline size_t mog_plane_off(int Palloc, int plane, int base) { return (size_t)plane * Palloc + base; }
__host__ __device__ inline size_t mog_plane_stride(int Palloc) { return Palloc; }
#endif

struct MogParams {
    float Tb, TB, Tg, varInit, varMin, varMax, tau;
    int nmix;
    int detectShadows;
    int shadowVal;
};

// inRange bounds after cv::inRange's normalisation: lo > hi encodes "empty".
struct RangeParams {
    int lo[3], hi[3];
};

struct MogLaunch {
    const uint8_t *frames;   // [n][H*W*channels] packed BGR (3) or GREY (1)
    int channels;
    float *state;            // [n][mog_stream_floats(Palloc)]
    uint8_t *nmodes;         // planar layout only: [n][Palloc] counters (same lane-interleaved slots)
    u64 *thr_bits;           // [n][Palloc/64] or nullptr
    uint8_t *out_bgr;        // [n][H*W*3] masked frame or nullptr
    uint8_t *out_mask;       // [n][H*W] {0,127,255} or nullptr
    int out_base;            // out_bgr / out_mask are indexed by (stream - out_base)
    const u64 *roi_bits;     // [n][Palloc/64] region-of-interest bits (framefilt mask fused in) or nullptr
    float alphaT, alpha1, prune;
    int fresh;               // 1: model is (re)initialised this frame -> no modes
    MogParams mp;
    RangeParams rp;
};

// --- kernels_mog.hip ---
void launch_mog_fused(const Geom &g, const MogLaunch &a, int first_stream, int n_streams, hipStream_t st);
// plain streaming kernels for the achievable-bandwidth measurement (n16 = number of 16-byte elements)
void launch_stream_read(const void *src, size_t n16, unsigned *sink, hipStream_t st);
void launch_stream_copy(const void *src, void *dst, size_t n16, hipStream_t st);
void launch_nop(hipStream_t st);   // one empty wave: calibrates what an event pair adds around a launch
void launch_bgr2hsv(const uint8_t *bgr, uint8_t *hsv, size_t npx, hipStream_t st);
// 3-channel (HSV) or 1-channel (grey) inRange of ONE frame into a bit mask.
void launch_inrange_bits(const Geom &g, const uint8_t *frame, int channels, const RangeParams &rp,
                         u64 *bits, hipStream_t st);
// framefilt bsub (BackgroundSubtractor.cpp:87-100) on n = rows*cols*channels bytes of one stream
void launch_bsub(const uint8_t *in, uint8_t *out, uint8_t *bg, float *bg_f, size_t n, float a, float b, int first,
                 int learn, hipStream_t st);
// framefilt thresh (Threshold.cpp:67-81): BGR->grey (channels 3) -> inRange -> setTo(0)
void launch_thresh_filter(const uint8_t *in, uint8_t *out, size_t npx, int channels, int lo, int hi, hipStream_t st);
// posidet diff front end of ONE frame: bits = |frame - last| > thr (or frame != 0 when !have_last); last = frame
void launch_absdiff_bits(const Geom &g, const uint8_t *frame, uint8_t *last, int thr, int have_last, u64 *bits,
                         hipStream_t st);
void launch_unpack_bits(const Geom &g, const u64 *bits, uint8_t *out, hipStream_t st);
// rows*cols bytes (nonzero = keep) -> bit mask of one stream
void launch_pack_bits(const Geom &g, const uint8_t *in, u64 *bits, hipStream_t st);
// model checkpoint: logical (OpenCV AoS) <-> device planes
// state / nmodes point at ONE stream's model
void launch_state_export(const Geom &g, float *state, uint8_t *nmodes, int nmix, int channels,
                         uint8_t *modes_used, float *weight, float *variance, float *mean, hipStream_t st);
void launch_state_import(const Geom &g, float *state, uint8_t *nmodes, int nmix, int channels,
                         const uint8_t *modes_used, const float *weight, const float *variance,
                         const float *mean, hipStream_t st);

// --- kernels_blob.hip ---
struct BlobBuffers {
    u64 *thr;        // [2][n][Palloc/64] inRange output, double-buffered across frames
    u64 *tmp;        // [n][Palloc/64] morphology ping
    u64 *morph;      // [n][Palloc/64] after erode/dilate
    u64 *fin;        // [n][Palloc/64] after 1-px frame zeroing
    u64 *trans;      // [n][Palloc/64] run-start (transition) bits
    int *carry;      // [n][H*words]   start x of the run entering each word
    int *parent;     // [n][Palloc]    union-find over run heads (sparse)
    long long *acc;  // [n][Palloc][3] Green sums per root (sparse)
    unsigned *done;  // [n]            workgroup arrival counter of k_green_select
    int *roots;      // [n][Palloc/2]  foreground roots of the current frame
    unsigned *nroots;// [n]
};
struct ResultRec {   // device-side result, one per stream per step
    long long a00, a10, a01;
    int first_pixel;
    int valid;
    // written by k_kalman when the position filter is on
    int kal_valid;
    int pad_;
    double kx, ky, kvx, kvy;
};

// `posifilt kalman` state of one camera stream (KalmanFilter2D.h:53-82 + cv::KalmanFilter members)
struct KalmanState {
    double statePre[4], statePost[4], Ppre[16], Ppost[16];
    double meas[2];          // kf_meas_ (keeps the stale measurement across missed detections)
    double reported[4];      // kf_predicted_state_ before it starts sharing statePre's buffer
    int found, missing, aliased;
    unsigned ticket;         // index of the next frame allowed to update this filter
};
struct KalmanLaunch {
    KalmanState *state;      // [n_streams]
    double dt, sig_accel, sig_noise;
    int threshold;           // not_found_count_threshold_ = (int)(timeout / dt)
    unsigned ticket;         // this frame's index
};
void launch_kalman(const KalmanLaunch &k, ResultRec *results, int n_streams, hipStream_t st);
void launch_kalman_reset(KalmanState *state, int n_streams, unsigned ticket, hipStream_t st);

void launch_morph(const Geom &g, const u64 *src, u64 *dst, int k, bool is_erode, int first_stream,
                  int n_streams, hipStream_t st);
// src_bits: mask before morphology; ero_k > 1 / dil_k > 1 fuse the erosion / dilation into the row
// scan (the erosion through LDS: rowscan_lds_bytes() must stay under kRowscanLdsMax, otherwise the
// caller erodes with launch_morph first and passes ero_k = 0).
// results: device-visible (host-mapped) array indexed by stream.
constexpr size_t kRowscanLdsMax = 64 * 1024;
size_t rowscan_lds_bytes(const Geom &g, int dil_k);
void launch_blob(const Geom &g, const BlobBuffers &b, const u64 *src_bits, int ero_k, int dil_k, double min_area,
                 double max_area, ResultRec *results, int first_stream, int n_streams, hipStream_t st);

}  // namespace oatgpu
