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
    unsigned words_magic; // ceil(2^32 / words): row of mask word i = (i * words_magic) >> 32, exact for every word of a frame
};

// MOG2 model in HBM (per stream): 25 fp32 planes + one u8 plane (mode counters), 101 B/px.
// One lane of K1 owns one pixel (p = y*Wp + x), a wavefront 64 consecutive pixels = one mask word:
// every plane access of a wave is one coalesced 256-byte line pair and __ballot(thr) IS the word.
// (Round 1 measured 2 and 4 pixels per lane with 8/16-byte accesses: 108/165 VGPRs, 4/3 waves per
// SIMD, slower; tools/k1_lab.hip shows the bare access pattern gains <= 3 % from wider accesses.)
//
// The counter byte of a pixel:  bits 0-2  modesUsed (what the reference keeps, 0..5)
//                               bits 3-6  LIVE hints: bit 2+k set <=> slot k (k = 1..4) holds a
//                                         weight that is not exactly 0
// The hints are this kernel's own bookkeeping (exported state has bits 0-2 only).  With the
// reference's `nmodes = nNewModes;` a pruned mode keeps its slot for ever with weight 0, so after a
// while most pixels count 5 modes of which 1-2 are alive; a slot whose weight is 0 can only matter
// to a pixel that has not matched an earlier mode (it may be re-matched and revived).  K1 therefore
// loads, of slots >= 1, the weights of LIVE slots only, and variance/mean only for pixels that did
// not match mode 0 as background ("needy" lanes) -- see k_mog_fused.
constexpr int kMogPlanes = 25;
constexpr int kMaxMix = 5;
constexpr int kWavePx = 64;               // pixels one wavefront owns
constexpr int kCountMask = 7;
constexpr int kLiveShift = 2;             // live bit of slot k is bit kLiveShift + k (k >= 1)

// Where the model lives.  Per mode k: a WEIGHT plane (fp32[Palloc]) and a plane of {variance, mean[channels]}
// RECORDS (fp32[Palloc][1 + channels]: 16 bytes per pixel for BGR, 8 for GREY), mode after mode; the counter bytes
// in an array of their own.  Weights stand alone because they are what changes every frame on every live mode;
// variance and mean travel together (a mode's fit test, update and swap always touch all of them): one 16-byte
// access per lane -- 1 KiB per wave instruction for mode 0, which every pixel reads and rewrites every frame --
// and a lane that needs a later mode touches ONE half-sector of it instead of a sector in each of four planes.
// (Round 1 and the first half of round 2 kept 25 scalar planes; profiles/r02_k1_layout_ab.txt also has the
// measurements of 256-pixel tile records and of the nontemporal cache policy, both rejected.)
constexpr size_t kPlanePad = 0;     // floats between consecutive planes (DRAM channel alignment of the ten streams was an A/B knob: no gain)
__host__ __device__ inline size_t mog_stream_floats(int Palloc) { return (size_t)kMogPlanes * Palloc + 10 * kPlanePad; }
__host__ __device__ inline size_t mog_w_off(int Palloc, int ch, int k) { return (size_t)k * ((2 + ch) * (size_t)Palloc + 2 * kPlanePad); }
__host__ __device__ inline size_t mog_vm_off(int Palloc, int ch, int k) { return mog_w_off(Palloc, ch, k) + Palloc + kPlanePad; }

struct MogParams {
    float Tb, TB, Tg, varInit, varMin, varMax, tau;
    int nmix;
    int detectShadows;
    int shadowVal;
    int restoreCount;        // MOG2Invoker's `nmodes = nNewModes;` (oatgpu_config.mog_restore_nmodes)
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
    // Temporal fusion (kernels_mog.hip "Two frames a launch"): frames2 != nullptr -> the launch advances every
    // stream by TWO frames -- `frames` then `frames2` -- on ONE pass over the model; thr_bits2 takes the second
    // frame's threshold words, the *2 rates are the second frame's.  Never with fresh or out_bgr / out_mask; with audit
    // for BGR only.
    const uint8_t *frames2;
    u64 *thr_bits2;
    float alphaT2, alpha12, prune2;
    int nt_loads;            // 1: slots 1..4 are LOADED with the streaming cache policy too (dense models; kernels_mog.hip)
    unsigned long long *audit;   // nullptr, or 8 device counters: the traffic-audit instantiation runs (oatgpu_traffic_audit)
    MogParams mp;
    RangeParams rp;
    int audit_frozen;            // (last: the product instantiations never read it, and their argument offsets stay as measured)
                                 // audited launches: 1 when the PRODUCT launcher would have picked a frozen-model instantiation
                                 // for this launch (launch_mog_fused: every rate 0, default-policy loads, not fresh) -- the audit
                                 // then counts a fitted record's store only where its bits changed, as that kernel stores
#ifdef OATGPU_RS_TIMING
    unsigned long long *rs_end;  // measurement builds: the launch's last workgroups stamp the wall clock here (tools/rowscan_probe.py)
#endif
};

// --- kernels_mog.hip ---
// stop: an event that becomes the launch's own completion (nullptr: none)
// wg: threads a workgroup, 256 or 64 (one wave a workgroup: kernels_mog.hip, k_mog_fused); wild_model: a stream of the launch
// holds an imported model that is not PLAIN -- weights the kernel's in-range division does not cover, means or variances on
// which a rate-0 update is not the identity; wild_sink: 8 device counters the
// launches that keep the compiler's division count into (the audit instantiations; never nullptr in a context's launches)
struct MogLaunchOpts {
    int wg = 256;
    bool wild_model = false;
    bool frozen_ok = true;       // varMin <= varInit <= varMax: the frozen-model instantiations may skip the identity update of a fit site
    unsigned long long *wild_sink = nullptr;
};
void launch_mog_fused(const Geom &g, const MogLaunch &a, int first_stream, int n_streams, hipStream_t st, hipEvent_t stop,
                      const MogLaunchOpts &o);
// plain streaming kernels for the achievable-bandwidth measurement (n16 = number of 16-byte elements)
void launch_stream_read(const void *src, size_t n16, unsigned *sink, hipStream_t st);
void launch_stream_copy(const void *src, void *dst, size_t n16, hipStream_t st);
void launch_density_probe(const uint8_t *nmodes, size_t total, unsigned *out, hipStream_t st);   // out = {live modes, samples}
void launch_nop(hipStream_t st);   // one empty wave: calibrates what an event pair adds around a launch
// bytes from device-visible (page-locked, mapped) host memory to device memory by a kernel; both 16-byte aligned
void launch_stage_copy(const void *src_dev_visible, void *dst, size_t bytes, hipStream_t st);
void launch_bgr2hsv(const uint8_t *bgr, uint8_t *hsv, size_t npx, hipStream_t st);
// code 0: BGR -> GREY, 1: GREY -> BGR, 2: HSV -> BGR (Color.h:45-51); in/out 4-byte aligned
void launch_cvt_color(int code, const uint8_t *in, uint8_t *out, size_t npx, hipStream_t st);
// 3-channel (HSV) or 1-channel (grey) inRange of ONE frame into a bit mask.
void launch_inrange_bits(const Geom &g, const uint8_t *frame, int channels, const RangeParams &rp,
                         u64 *bits, hipStream_t st);
// framefilt bsub (BackgroundSubtractor.cpp:87-100) on n = rows*cols*channels bytes of one stream
void launch_bsub(const uint8_t *in, uint8_t *out, uint8_t *bg, float *bg_f, size_t n, float a, float b, int first,
                 int learn, hipStream_t st);
// framefilt mask (FrameMasker.cpp:71-75) with ONE stream's ROI bit plane
void launch_apply_roi(const Geom &g, const uint8_t *in, uint8_t *out, int channels, const u64 *roi, hipStream_t st);
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
    // the single-workgroup LDS path (k_blob_lds): written by k_rowscan
    unsigned short *wpre;  // [n][H*words] run starts of the row in the words before this one
    int *rowinfo;          // [n][H]       0: no foreground in the row; else its number of runs
    unsigned *lds_ok;      // [n]          1: k_blob_lds wrote this frame's result, k_merge / k_green_select stand down
    // the early blob workgroup (kernels_blob.hip "Early dispatch"): a one-lane kernel behind the row scan publishes the frame's
    // ticket, the k_blob_lds workgroup that was dispatched ahead of it waits for exactly that ticket
    unsigned *ready;       // [n]          ticket of the latest frame whose row scan is complete (k_publish_ticket)
    unsigned *nopark;      // [1], one per CONTEXT (every scratch set points at it): set by the first parked workgroup that gave up
                           //              waiting -- the parked workgroups already in flight behind it decline at once instead of
                           //              waiting their own 100 ms each (kPathNoPark)
};
struct ResultRec {   // device-side result, one per stream per step
    long long a00, a10, a01;
    int first_pixel;
    int valid;
    // written by k_kalman when the position filter is on
    int kal_valid;
    int path;            // who wrote the record: 1 = k_blob_lds, 0 = k_green_select (host: when to speculate, below);
                         // kPathTimeout beside valid == kNeedsGlobal: the parked k_blob_lds workgroup gave up waiting for its ticket
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
#ifdef OATGPU_RS_TIMING
void oatgpu_debug_rs_set_k1_end(const unsigned long long *p);     // measurement builds (kernels_blob.hip)
#endif
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
// mode: kBlobFull  = row scan + k_blob_lds + k_merge + k_green_select (the last two stand down when the LDS
//                     kernel took the frame);
//       kBlobSpec  = row scan + k_blob_lds only: a frame too busy for it comes back with valid == kNeedsGlobal and
//                     the caller runs kBlobGlobal on the same threshold bits before it hands the result out;
//       kBlobGlobal = row scan + k_merge + k_green_select.
enum { kBlobFull = 0, kBlobSpec = 1, kBlobGlobal = 2 };
constexpr int kNeedsGlobal = -2;
constexpr int kPathTimeout = 2;
constexpr int kPathNoPark = 3;     // beside valid == kNeedsGlobal: declined without waiting, an earlier workgroup of the context had timed out
void launch_blob(const Geom &g, const BlobBuffers &b, const u64 *src_bits, int ero_k, int dil_k, double min_area,
                 double max_area, ResultRec *results, int first_stream, int n_streams, hipStream_t st, int mode = kBlobFull);
// The same in two halves on two HIP streams (kernels_blob.hip "Early dispatch"): the row scan with a one-lane kernel behind
// it that publishes `ticket`, and everything behind the row scan, whose k_blob_lds workgroup may be dispatched long before
// the row scan has run and waits for `ticket` on the device.  ticket != 0.
void launch_rowscan_signal(const Geom &g, const BlobBuffers &b, const u64 *src_bits, int ero_k, int dil_k, int first_stream,
                           int n_streams, unsigned ticket, hipStream_t st);
// both frames of a two-frame step: ONE row-scan launch and ONE k_blob_lds launch (speculative mode), lds-able geometries only
void launch_blob_pair(const Geom &g, const BlobBuffers *b, const u64 *const *src, int ero_k, int dil_k, double min_area,
                      double max_area, ResultRec *const *results, int n_streams, hipStream_t st);
// the blob workgroups of the nf (1 or 2) frames of a step in ONE launch (speculative mode): arrays of nf scratch sets,
// result records and tickets
void launch_blob_tail2(const Geom &g, const BlobBuffers *b, double min_area, double max_area, ResultRec *const *results,
                       int n_streams, const unsigned *ticket, int nf, hipStream_t st_tail);

}  // namespace oatgpu
