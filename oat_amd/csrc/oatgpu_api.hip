// oatgpu_api.hip -- the C ABI of include/oatgpu.h over the gfx950 kernels.
//
// Host-side orchestration only: geometry, the MOG2 learning-rate schedule
// (BackgroundSubtractorMOG2Impl::apply prologue), kernel sequencing on one HIP
// stream, the pinned result ring, and contourMoments' double-precision
// epilogue.  No CPU fallback for any pixel work: every entry point either runs
// the HIP kernels or fails with an error code.
#include "../../include/oatgpu.h"
#include "oatgpu_internal.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <unistd.h>

using namespace oatgpu;

static thread_local std::string g_last_error;

// Measurement switches (OATGPU_EXPT / SERIAL / NB / PRIVATE_STREAMS / COPY_PAD / GRAPH) exist only in
// A/B builds made with -DOATGPU_MEASURE (`make variant`); the product library never reads them, so
// nothing in the environment can change what it computes or skip work (tests/test_gpu_parity.py::
// test_measurement_env_is_inert_in_the_product_library).
static inline const char *measure_env(const char *name)
{
#ifdef OATGPU_MEASURE
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

struct ProfStep { hipEvent_t e[5]; int frames = 1; };   // frames: what the K1 launch between e[0] and e[1] covered
struct Rate { float alphaT, alpha1, prune; int fresh; };   // A: K1 begin/end; B: back-half begin, after erode, end

struct oatgpu_ctx {
    oatgpu_config cfg;
    Geom g;
    hipStream_t stream = nullptr;   // stream A: uploads + the fused per-pixel kernel
    bool own_stream = false;
    bool private_streams = false;   // OATGPU_PRIVATE_STREAMS: streams of its own instead of the device's shared ones
    bool have_shared = false;
    static constexpr int kNB = 4;
    static constexpr int kSets = 5;              // scratch sets of the back half: up to four for frames in flight + one for repairs
    int nb = 3;                                  // back-half streams / scratch sets in use (OATGPU_NB: 1..4)
    bool b_used[kNB] = {};          // B streams this context has launched on (the only ones it ever has to drain)
    hipStream_t stream_b[kNB] = {}; // streams B0..B2: morphology + blob analysis, frame t on B[t % nb],
                                                  // overlapped with later frames' per-pixel kernels and each other
    hipEvent_t ev_k1[kNB] = {};    // the K1 whose back half runs on B[q] has finished (its threshold bits are ready)
    unsigned long long enq_total = 0, col_total = 0;  // pipelined frames enqueued / collected so far
    // cache policy of K1's slot-1..4 loads (kernels_mog.hip): streaming on dense models.  Two host-mapped
    // {live modes, samples} pairs filled alternately by k_density_probe; read one probe late, never waited for.
    unsigned *dens_host = nullptr, *dens_dev = nullptr;
    unsigned long long dens_probes = 0;
    bool nt_loads = false;
    // back half: frames go through row scan + k_blob_lds ONLY (kBlobSpec) while that keeps working; a frame too busy
    // for the LDS kernel comes back marked, oatgpu_track_collect then runs the global kernels on its threshold bits
    // (still in its ring slot) before handing the result out, and the full launch sequence is used until
    // kSpecAfter frames in a row were taken by the LDS kernel again.
    bool lds_spec = true;
    int lds_streak = 0;
    std::vector<char> slot_spec, slot_q;      // per ring slot: launched speculatively? which scratch set / B stream?
    std::vector<int> slot_ev;                 // per ring slot: the slot whose ring event covers this slot's result (itself,
                                              // or the second frame of its launch when both back halves share a B stream)
    int ring_slots = 0;                               // internal ring size (= ring_depth); one threshold-bit buffer per slot
    bool serial = false;
    hipStream_t stream_c = nullptr;  // H2D copies of oatgpu_track_enqueue (created on first use)
    uint8_t *frames_ring = nullptr;  // [ring_slots][n_streams*rows*cols*channels] staging for host frames
    std::vector<hipEvent_t> copy_ev; // [ring_slots] frames of this slot have arrived
    std::vector<hipEvent_t> copy_ev_s;   // [ring_slots][n_streams] ... and every single one of them (oatgpu_track_input_consumed_stream)
    hipStream_t stream_c2 = nullptr;     // second copy stream of the camera-by-camera path (shared per device)
    hipEvent_t ev_c2 = nullptr;          // ... and its "all my copies of this set are done"
    std::vector<char> staged;            // [n_streams] oatgpu_track_stage: frame of the NEXT set already on its way
    int staged_count = 0, stage_slot = -1;
    bool per_stream_copy_ev = false;     // recorded from the first per-stream call on
    bool last_copy_per_stream = false;   // the latest enqueue recorded them
    int last_copy_slot = -1;         // slot of the most recent oatgpu_track_enqueue (host frames)
    KalmanLaunch kal{};              // kal.state == nullptr: position filter off
    bool kal_on = false;
    unsigned kal_ticket = 0;         // ticket of the next enqueued frame
    int expt = 0;
    // Early dispatch of the blob workgroup (kernels_blob.hip): the row scans of a device-frame step go down B0 / B1 by frame
    // parity, as on the plain path; the k_blob_lds workgroups of the step's frames are submitted with them as ONE launch on
    // B2 and wait on the device for their row scans' tickets.  Scratch sets 0 .. 3 by frame index mod 4; repairs of declined
    // frames use set 4 on B2.
    int early_blob = -1;             // -1: by shape -- at most THREE streams, 4 MP a step and more, where the per-pixel kernel then runs with one wave
                                     // a workgroup (k_mog_fused, WG): 4K 18.5 k -> 19.2 k fps, result 245 -> 250 us behind its frame
                                     // (profiles/r05g_wg64_early_blob_ab.txt).  With 256-thread workgroups the parked workgroup costs
                                     // the per-pixel kernel 4 % (one at 4K) to 20 % (32 of them, 16 x 1080p) and several streams gain
                                     // nothing either way (r05h): off there.  0 / 1: oatgpu_set_early_blob
    bool stage_kernel = false;       // oatgpu_set_stage_copy(1): oatgpu_track_stage copies with a kernel reading the host frame in place
    int k1_stop_event = -1;          // the step's "K1 done" event rides on the last K1 launch's own completion signal (no marker
                                     // packet behind it on stream A): -1 by step size (>= 4 MP: +1..2.5 % at 4K; small steps are
                                     // bound by the host's calls, and hipExtLaunchKernel costs more of those: one 1080p stream -3 %),
                                     // 0 / 1 forced (measurement builds: OATGPU_K1_STOP_EVENT)
    size_t early_min_px = 4000000;   // pixels a step from which the early order is considered (measurement builds: OATGPU_EARLY_MIN_PX)
    int k1_wg_force = 0;             // oatgpu_set_k1_workgroup (64 | 256): the per-pixel kernel's workgroup size whatever the path
    unsigned *nopark = nullptr;      // device word: a parked blob workgroup of this context has timed out (BlobBuffers::nopark)
    unsigned long long early_timeouts = 0;   // frames whose parked blob workgroup gave up waiting for its row scan (kWaitTicks):
                                     // something serialises kernel dispatches -- the context then stops parking (early_off)
    bool early_off = false;
    int last_k1_wg = 0;              // what the latest pipelined step used (oatgpu_last_step_shape)
    bool last_step_early = false;
    int last_early = -1;             // path of the previous step (-1: none yet; 0 plain, 1 early, 2 paired): a switch drains the B streams first
    bool have_set[kSets] = {};       // scratch sets allocated (all of them in oatgpu_create when nb >= 3)
    int lone_plain = 1;              // a frame launched with NOTHING else outstanding (a camera-paced caller) takes the plain order even where the early
                                     // order is the default: no ticket kernel, no parked workgroup to release -- one frame at a time 133.5 -> 129.3 us at
                                     // 4K, 89.6 -> 85.9 us for two 1080p streams, frame rate unchanged (profiles/r07t_lone_frame_plain_order_ab.txt) --
                                     // and its back half goes to stream A itself (inline_back in launch_jobs: 4K 131 -> 122 us, one 1080p stream
                                     // 66 -> 58.5 us, profiles/r07in_lone_frame_inline_ab.txt; measurement builds: OATGPU_LONE_PLAIN=0)
    int pair_back = 1;               // two-frame steps outside the early order: ONE row-scan launch and ONE blob launch for both frames
                                     // (measurement builds: OATGPU_PAIR_BACK=0)
    unsigned pair_steps = 0;         // paired steps so far (their parity picks the B stream and the two scratch sets)
    unsigned bh_ticket[kSets] = {};
    unsigned early_frames = 0;       // frames that took the early path (their parity picks scratch set and row-scan stream)
    hipEvent_t ev_blob[kSets] = {};  // scratch set q: its latest reader is done
    bool ev_blob_valid[kSets] = {};
    std::vector<char> slot_st;       // per ring slot: B stream its result event was recorded on / its repair goes to
    bool use_graph = false;                           // back half replayed from a captured hipGraph per slot
    std::vector<char> slot_filtered;                  // [ring_slots] was the position filter applied to this slot?
    std::vector<hipGraphExec_t> back_graph;           // [ring_slots], built lazily, dropped on set_detector
    int last_q = 0;
    std::string err;
    // Deferred completion of the single-stage calls (oatgpu_set_deferred): the call returns when its INPUT has been read
    // (the H2D copy is done), the result stays on the device until oatgpu_fetch_frame / oatgpu_fetch_position
    bool deferred = false;
    hipEvent_t ev_h2d = nullptr;
    int defer_kind = 0;              // 0 nothing waiting, 1 a frame (defer_dev, defer_bytes), 2 a position (defer_s)
    const uint8_t *defer_dev = nullptr;
    size_t defer_bytes = 0;
    int defer_s = 0;
    bool broken = false;             // a launch of the pipelined path failed half-way: the model and its rate schedule are ahead
                                     // of the results -- every later pipelined call is refused (oatgpu.h "Errors")
    // Temporal fusion (kernels_mog.hip "Two frames a launch"): with fuse == 2 a pipelined enqueue only REGISTERS its
    // frame (ring slot, counters); the kernels go out when the next frame is enqueued -- K1 once for both -- or when
    // somebody needs the frame's result or the model (collect / ready of that very frame, every synchronous entry
    // point, destroy), then for the one frame alone.  Results, their order and the model are those of one launch a frame.
    struct FrameJob { const void *frames = nullptr; double lr = 0.0; hipEvent_t ready = nullptr; int slot = 0; };
    // `posifilt homography` behind the detector / the position filter (oatgpu_set_homography): applied where the
    // centroid is finished, on the host, in the reference's double arithmetic
    bool homo_on = false;
    double homo[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int fuse = 0;                                     // 0: default -- pair where the library owns the frame's lifetime
                                                      // (host frames, track_sequence_dev), not for oatgpu_track_enqueue_dev
    bool in_sequence = false;                         // inside oatgpu_track_sequence_dev: its frames are all in hand
    hipEvent_t ev_in = nullptr;                       // stream A has read every frame handed over so far (input_consumed)
    bool dev_unconsumed = false;                      // device frames were enqueued since the last input_consumed
    std::vector<char> slot_repair;                    // per ring slot: the global kernels are redoing this frame (ready/quiesce)
    bool pend_valid = false;
    FrameJob pend;
    unsigned long long launched_total = 0;            // frames whose kernels are out (<= enq_total)

    // device memory
    float *state = nullptr;
    uint8_t *nmodes = nullptr;
    uint8_t *frames = nullptr;     // staging [n][H*W*3]
    uint8_t *aux_a = nullptr;      // [H*W*3]
    uint8_t *aux_b = nullptr;      // [H*W*3]
    uint8_t *bsub_bg = nullptr;    // [n][H*W*ch] framefilt bsub background (u8) and its fp32 accumulator
    float *bsub_f = nullptr;
    std::vector<char> bsub_have;
    uint8_t *diff_last = nullptr;  // [n][H*W] previous GREY frame of posidet diff, allocated on first use
    std::vector<char> diff_have;   // per camera stream
    u64 *roi = nullptr;            // [n][Palloc/64] ROI bits, allocated on first oatgpu_set_roi_mask
    BlobBuffers bb[kSets]{};          // scratch sets (bb[0].thr holds the ring's threshold buffers), all made with the context
    const u64 *last_morph = nullptr;
    const u64 *last_fin = nullptr;
    ResultRec *res_host = nullptr; // [ring_depth+1][n] pinned + mapped (last slot: single-stage calls)
    ResultRec *res_dev = nullptr;  // device alias of res_host: kernels store results straight to the host
    std::vector<hipEvent_t> ring_ev;
    int ring_count = 0;

    std::vector<int> nframes;      // per camera stream
    std::vector<Rate> rates_scratch;

    // profiling
    bool prof = false;
    int prof_every = 1;            // sample every Nth step (event records cost ~4 us of host time each)
    unsigned long long prof_tick = 0;
    std::vector<ProfStep> prof_steps;
    size_t prof_used = 0;
    oatgpu_profile prof_sum{};
    int prof_drop_streak = 0;      // consecutive samples prof_fold took for host stalls
    double event_pair_ms = 0.0;

    // traffic audit (oatgpu_traffic_audit)
    unsigned long long *audit_dev = nullptr;   // 8 counters
#ifdef OATGPU_RS_TIMING
    unsigned long long *rs_k1_end = nullptr;   // (leaked at destroy: measurement builds only)
#endif
    unsigned long long *wild_sink = nullptr;   // 8 counters nobody reads: launches outside div_inrange's operands (kernels_mog.hip)
    std::vector<char> wild_model;              // [n_streams] 1: an imported model with weights no run of the kernel produces
    bool audit_on = false;
    long long audit_launches = 0;
};

static int fail(oatgpu_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (c) c->err = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((c), OATGPU_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

extern "C" int oatgpu_abi_version(void) { return OATGPU_ABI_VERSION; }

extern "C" int oatgpu_default_config(oatgpu_config *c)
{
    if (!c) return OATGPU_E_INVALID;
    memset(c, 0, sizeof *c);
    c->device = 0; c->n_streams = 1; c->rows = 0; c->cols = 0; c->ring_depth = 4; c->channels = 3;
    // cv::createBackgroundSubtractorMOG2() defaults (bgfg_gaussmix2.cpp)
    c->history = 500; c->nmixtures = 5; c->var_threshold = 16.f; c->background_ratio = 0.9f;
    c->var_threshold_gen = 9.f; c->var_init = 15.f; c->var_min = 4.f; c->var_max = 75.f;
    c->ct = 0.05f; c->tau = 0.5f; c->detect_shadows = 1; c->shadow_value = 127;
    // HSVDetector.h:77-94, HSVDetector.cpp:42-43
    c->h_lo = 0; c->h_hi = 256; c->s_lo = 0; c->s_hi = 256; c->v_lo = 0; c->v_hi = 256;
    c->erode = 0; c->dilate = 10; c->min_area = 0.0; c->max_area = DBL_MAX;
    c->diff_threshold = 10; c->blur = 2;       // DifferenceDetector.h:74, DifferenceDetector.cpp:41
    c->mog_restore_nmodes = 1;                 // bgfg_gaussmix2.cpp MOG2Invoker: nmodes = nNewModes
    return OATGPU_OK;
}

static int check_detector(oatgpu_ctx *c, const oatgpu_config &k)
{
    const int v[6] = { k.h_lo, k.h_hi, k.s_lo, k.s_hi, k.v_lo, k.v_hi };
    for (int i = 0; i < 6; ++i)
        if (v[i] < 0 || v[i] > 256)   // HSVDetector.cpp:87,98,109
            return fail(c, OATGPU_E_INVALID, "threshold values should be between 0 and 256");
    if (k.erode < 0 || k.dilate < 0) return fail(c, OATGPU_E_INVALID, "erode/dilate must be >= 0");
    if (k.erode > 63 || k.dilate > 63)
        return fail(c, OATGPU_E_INVALID, "erode/dilate kernel sizes above 63 are not supported");
    if (k.blur < 0 || k.blur > 22)
        return fail(c, OATGPU_E_INVALID, "blur must be in 0..22 (above that a box blur is no longer a dilation)");
    if (!(k.min_area < k.max_area))   // HSVDetector.cpp:135
        return fail(c, OATGPU_E_INVALID, "Max area should be larger than min area.");
    return OATGPU_OK;
}

// cv::inRange's scalar-bound normalisation for 8U sources
static void norm_range(int lo, int hi, int &l, int &h)
{
    if (lo > hi || lo > 255 || hi < 0) { l = 1; h = 0; return; }
    l = lo < 0 ? 0 : lo;
    h = hi > 255 ? 255 : hi;
}
static RangeParams range_of(const oatgpu_config &k)
{
    RangeParams r;
    norm_range(k.h_lo, k.h_hi, r.lo[0], r.hi[0]);
    norm_range(k.s_lo, k.s_hi, r.lo[1], r.hi[1]);
    norm_range(k.v_lo, k.v_hi, r.lo[2], r.hi[2]);
    return r;
}
static MogParams mogparams_of(const oatgpu_config &k)
{
    MogParams m;
    m.Tb = k.var_threshold; m.TB = k.background_ratio; m.Tg = k.var_threshold_gen;
    m.varInit = k.var_init; m.varMin = k.var_min; m.varMax = k.var_max; m.tau = k.tau;
    m.nmix = k.nmixtures; m.detectShadows = k.detect_shadows; m.shadowVal = k.shadow_value;
    m.restoreCount = k.mog_restore_nmodes ? 1 : 0;
    return m;
}

// The HIP streams are shared by all contexts of a device in this process, and they are created
// together, in a fixed order.  Measured on MI355X / ROCm 7.2: the runtime
// spreads the streams of a process over FOUR hardware queues in the order they are created (the
// null stream counts from its first use; priorities and GPU_MAX_HW_QUEUES make no difference), and
// two streams on one hardware queue execute strictly one after the other.  A fifth stream therefore
// lands on the first one's queue: with private streams the second context's B0 shared a queue with
// the first context's A, and every K1 waited for the other context's back halves (18.9k fps for two
// contexts, against 38.3k for one context holding both cameras).  So: A, B0, B1, B2 are created
// back to back (any four consecutive streams sit on four different queues) and shared by all
// contexts; the host-frame copy stream is created right behind them (the host-frame path leaves B2
// unused, which keeps it at four active streams).
namespace {
struct DeviceStreams {
    int refs = 0, nb = 0;
    hipStream_t a = nullptr, b[oatgpu_ctx::kNB] = {}, copy = nullptr, copy2 = nullptr;
    std::vector<hipStream_t> padding;
};
std::mutex g_streams_mutex;
std::map<int, DeviceStreams> g_streams;      // by device ordinal

hipStream_t create_b_stream()
{
    // the B streams carry short latency-bound kernels that must slip in between the big
    // bandwidth-bound launches of stream A: highest priority
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = greatest = 0; }
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest) != hipSuccess) st = nullptr;
    return st;
}

bool acquire_streams(int device, int nb, hipStream_t *a, hipStream_t *b)
{
    std::lock_guard<std::mutex> lk(g_streams_mutex);
    DeviceStreams &d = g_streams[device];
    if (!d.a) {
        // (measurement builds: OATGPU_A_RESERVE=n keeps stream A -- the per-pixel kernel -- off the last n compute units)
        int reserve = 0;
        if (const char *e = measure_env("OATGPU_A_RESERVE")) reserve = atoi(e);
        if (reserve > 0) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, device) != hipSuccess) return false;
            const int ncu = prop.multiProcessorCount;
            std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
            for (int i = 0; i < ncu - reserve; ++i) mask[(size_t)i >> 5] |= 1u << (i & 31);
            if (hipExtStreamCreateWithCUMask(&d.a, (uint32_t)mask.size(), mask.data()) != hipSuccess) { d.a = nullptr; return false; }
        } else
        if (hipStreamCreateWithFlags(&d.a, hipStreamNonBlocking) != hipSuccess) { d.a = nullptr; return false; }
        for (int q = 0; q < oatgpu_ctx::kNB - 1; ++q)            // A + 3 B streams = the four queues
            if (!(d.b[q] = create_b_stream())) return false;
        d.nb = oatgpu_ctx::kNB - 1;
        // ... and, in the same breath, the host-frame copy stream.  Where it lands relative to A
        // decides who yields while host frames are copied: right behind the B streams (no padding)
        // the copies win and K1 waits for them, which is the better trade on a PCIe-bound path
        // (7.8k fps at 1080p with K1 stretched to 62 us, against 5.6-7.0k fps with K1 at 26 us when
        // OATGPU_COPY_PAD=1..3 idle streams are put in front of it).
        int pad = 0;
        if (const char *e = measure_env("OATGPU_COPY_PAD")) pad = atoi(e);
        for (int i = 0; i < pad; ++i) {
            hipStream_t p = nullptr;
            if (hipStreamCreateWithFlags(&p, hipStreamNonBlocking) == hipSuccess) d.padding.push_back(p);
        }
        if (hipStreamCreateWithFlags(&d.copy, hipStreamNonBlocking) != hipSuccess) d.copy = nullptr;
    }
    for (int q = d.nb; q < nb; ++q) {                             // OATGPU_NB=4: measurement only
        if (!(d.b[q] = create_b_stream())) return false;
        d.nb = q + 1;
    }
    d.refs++;
    *a = d.a;
    for (int q = 0; q < nb; ++q) b[q] = d.b[q];
    return true;
}
// A second copy stream (created on first use: the camera-by-camera path alternates between the two, so that one copy's
// set-up and completion signalling hide behind the other's transfer)
hipStream_t acquire_copy_stream2(int device)
{
    std::lock_guard<std::mutex> lk(g_streams_mutex);
    DeviceStreams &d = g_streams[device];
    if (!d.copy2 && hipStreamCreateWithFlags(&d.copy2, hipStreamNonBlocking) != hipSuccess) d.copy2 = nullptr;
    return d.copy2;
}
hipStream_t acquire_copy_stream(int device)
{
    std::lock_guard<std::mutex> lk(g_streams_mutex);
    return g_streams[device].copy;
}
void release_streams(int device)
{
    // The device's streams are NOT destroyed with the last context: the runtime places streams on its four hardware
    // queues in creation order, so a later context that had to create its streams afresh may find A and a B stream
    // on one queue (bench.py's fourth leg in a row ran 1080p at 15.8 k instead of 27.9 k fps).  They live as long as
    // the process; the runtime reclaims them at exit.
    std::lock_guard<std::mutex> lk(g_streams_mutex);
    auto it = g_streams.find(device);
    if (it != g_streams.end() && it->second.refs > 0) --it->second.refs;
}
}  // namespace

namespace {
struct DevBuf {             // scoped device allocation
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
};
}  // namespace

static void free_all(oatgpu_ctx *c)
{
    if (!c) return;
    hipFree(c->bsub_bg); hipFree(c->bsub_f); hipFree(c->diff_last); hipFree(c->roi); hipFree(c->state); hipFree(c->nmodes); hipFree(c->frames); hipFree(c->aux_a); hipFree(c->aux_b);
    hipFree(c->bb[0].thr);
    hipFree(c->nopark);
    hipFree(c->kal.state);
    hipFree(c->audit_dev);
    hipFree(c->wild_sink);
    hipFree(c->frames_ring);
    if (c->ev_in) hipEventDestroy(c->ev_in);
    if (c->ev_h2d) hipEventDestroy(c->ev_h2d);
    if (c->ev_c2) hipEventDestroy(c->ev_c2);
    for (auto e : c->copy_ev) hipEventDestroy(e);
    for (auto e : c->copy_ev_s) hipEventDestroy(e);
    for (auto &b : c->bb) {
        hipFree(b.tmp); hipFree(b.morph); hipFree(b.fin); hipFree(b.trans);
        hipFree(b.carry); hipFree(b.parent); hipFree(b.acc); hipFree(b.done);
        hipFree(b.roots); hipFree(b.nroots); hipFree(b.wpre); hipFree(b.rowinfo); hipFree(b.lds_ok);
        hipFree(b.ready);
    }
    if (c->res_host) hipHostFree(c->res_host);
    if (c->dens_host) hipHostFree(c->dens_host);
    for (int q = 0; q < oatgpu_ctx::kNB; ++q) if (c->ev_k1[q]) hipEventDestroy(c->ev_k1[q]);
    for (int q = 0; q < oatgpu_ctx::kSets; ++q) if (c->ev_blob[q]) hipEventDestroy(c->ev_blob[q]);
    for (auto ge : c->back_graph) if (ge) hipGraphExecDestroy(ge);
    for (auto e : c->ring_ev) hipEventDestroy(e);
    for (auto &p : c->prof_steps) for (auto e : p.e) hipEventDestroy(e);
    if (c->private_streams) {
        for (auto sb : c->stream_b) if (sb) hipStreamDestroy(sb);
        if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
        if (c->stream_c) hipStreamDestroy(c->stream_c);
    } else if (c->have_shared) {
        release_streams(c->cfg.device);
    }
    delete c;
}

// Opening the device.  Several processes that open one freshly booted device in the same instant (the ranks of a job, the
// components of a pipeline started by one script) can see the first runtime calls fail transiently -- no device yet, the
// device busy being initialised by a sibling.  Bounded retry with back-off (50 ms doubling, ~1.5 s in all); an ordinal that
// does not exist is not transient.  Retries are counted (oatgpu_device_open_retries) so that a caller can report them
// instead of hiding them.
static std::atomic<int> g_open_retries{0};
static hipError_t open_device(int device, int *ndev)
{
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 6; ++attempt) {
        *ndev = 0;
        e = hipGetDeviceCount(ndev);
        if (e == hipSuccess && *ndev > 0) {
            if (device < 0 || device >= *ndev) return hipErrorInvalidDevice;
            e = hipSetDevice(device);
            if (e == hipSuccess) e = hipFree(nullptr);             // forces the context into being here, not in the first hipMalloc
            if (e == hipSuccess) return hipSuccess;
        } else if (e == hipSuccess) {
            e = hipErrorNoDevice;
        }
        (void)hipGetLastError();
        // (a box WITHOUT a device says so at the first call -- hipErrorNoDevice, or a count of 0 -- and is not kept
        // waiting 1.5 s for one on every oatgpu_create; what the retry absorbs is a device that exists and is busy being
        // initialised by a sibling process: any other error, or "no device" after a first attempt that saw one)
        if (attempt == 0 && e == hipErrorNoDevice) break;
        if (attempt == 5) break;
        g_open_retries++;
        usleep(50000u << attempt);
    }
    return e;
}
extern "C" int oatgpu_device_open_retries(void) { return g_open_retries.load(); }

// One scratch set of the back half (BlobBuffers; about 30 bytes a pixel and stream), its counters zeroed.  ALL sets a context
// can come to use are made with it (oatgpu_create): the three of the plain order, the fourth of the early and
// the paired layout, and the repair set -- 150 bytes a pixel and stream in all.  An allocation in the middle of a pipelined
// step would stall a camera-paced caller for as long as hipMalloc takes, and its failure would come after the model had
// moved.  launch_jobs keeps the on-first-use path only for measurement builds that shrink nb.
static bool alloc_scratch_set(oatgpu_ctx *c, int q)
{
    const Geom &g = c->g;
    const size_t n = c->cfg.n_streams, PA = g.Palloc, NW = PA / 64;
    BlobBuffers &b = c->bb[q];
    bool ok = true;
    auto A = [&](void **p, size_t bytes) { if (ok && hipMalloc(p, bytes) != hipSuccess) ok = false; };
    b.thr = c->bb[0].thr;
    b.nopark = c->nopark;
    A((void **)&b.tmp, n * NW * 8);
    A((void **)&b.morph, n * NW * 8);
    A((void **)&b.fin, n * NW * 8);
    A((void **)&b.trans, n * NW * 8);
    A((void **)&b.carry, n * (size_t)g.H * g.words * sizeof(int));
    A((void **)&b.parent, n * PA * sizeof(int));
    A((void **)&b.acc, n * PA * 3 * sizeof(long long));
    A((void **)&b.done, n * sizeof(unsigned));
    A((void **)&b.roots, n * (PA / 2) * sizeof(int));
    A((void **)&b.nroots, n * sizeof(unsigned));
    A((void **)&b.wpre, n * (size_t)g.H * g.words * sizeof(unsigned short));
    A((void **)&b.rowinfo, n * (size_t)g.H * sizeof(int));
    A((void **)&b.lds_ok, n * sizeof(unsigned));
    A((void **)&b.ready, n * sizeof(unsigned));
    for (unsigned *p : {b.done, b.nroots, b.lds_ok, b.ready})
        if (ok && hipMemsetAsync(p, 0, n * sizeof(unsigned), c->stream) != hipSuccess) ok = false;
    return ok;
}

extern "C" oatgpu_ctx *oatgpu_create(const oatgpu_config *cfg)
{
    if (!cfg) { fail(nullptr, OATGPU_E_INVALID, "null config"); return nullptr; }
    if (cfg->rows < 1 || cfg->cols < 1 || cfg->n_streams < 1 || cfg->ring_depth < 1) {
        fail(nullptr, OATGPU_E_INVALID, "rows, cols, n_streams and ring_depth must be >= 1");
        return nullptr;
    }
    if (cfg->channels != 1 && cfg->channels != 3) {
        fail(nullptr, OATGPU_E_INVALID, "channels must be 3 (BGR) or 1 (GREY)");
        return nullptr;
    }
    if (cfg->nmixtures < 1 || cfg->nmixtures > kMaxMix) {
        fail(nullptr, OATGPU_E_INVALID, "nmixtures must be in 1..5");
        return nullptr;
    }
    if (!(cfg->ct >= 0.f && cfg->ct < 0.5f)) {      // a matched mode must never be prunable (kernels_mog.hip mog2_mode)
        fail(nullptr, OATGPU_E_INVALID, "ct (complexity reduction) must be in [0, 0.5)");
        return nullptr;
    }
    if ((long long)cfg->rows * (((long long)cfg->cols + 63) / 64 * 64) > 0x7fff0000ll) {
        fail(nullptr, OATGPU_E_INVALID, "frame too large");
        return nullptr;
    }
    if (check_detector(nullptr, *cfg) != OATGPU_OK) return nullptr;

    int ndev = 0;
    const hipError_t oe = open_device(cfg->device, &ndev);
    if (oe == hipErrorInvalidDevice && ndev > 0) {
        fail(nullptr, OATGPU_E_INVALID, "device %d out of range (have %d)", cfg->device, ndev);
        return nullptr;
    }
    if (oe != hipSuccess) {
        if (ndev < 1) fail(nullptr, OATGPU_E_NODEVICE, "no HIP device available");
        else fail(nullptr, OATGPU_E_HIP, "hipSetDevice(%d) failed: %s", cfg->device, hipGetErrorString(oe));
        return nullptr;
    }

    oatgpu_ctx *c = new oatgpu_ctx();
    c->cfg = *cfg;
    Geom &g = c->g;
    g.H = cfg->rows; g.W = cfg->cols; g.Wp = (cfg->cols + 63) / 64 * 64; g.words = g.Wp / 64;
    g.P = g.H * g.Wp; g.Palloc = (g.P + 1023) / 1024 * 1024; g.n_streams = cfg->n_streams;
    g.words_magic = (unsigned)((0x100000000ull + (unsigned)g.words - 1) / (unsigned)g.words);
    // (i * magic) >> 32 == i / words needs i * (magic * words - 2^32) < 2^32; the error term is < words
    if ((unsigned long long)(g.Palloc / 64) * (unsigned long long)g.words >= 0x100000000ull ||
        (unsigned long long)mog_stream_floats(g.Palloc) * 4ull >= 0x100000000ull) {     // 32-bit plane offsets in K1
        fail(nullptr, OATGPU_E_INVALID, "frame too large");
        delete c;
        return nullptr;
    }
    const size_t n = cfg->n_streams, npx = (size_t)g.H * g.W, PA = g.Palloc, NW = PA / 64;
    c->nframes.assign(n, 0);
    c->wild_model.assign(n, 0);
    c->diff_have.assign(n, 0);
    c->bsub_have.assign(n, 0);

    bool ok = true;
    auto A = [&](void **p, size_t bytes) { if (ok && hipMalloc(p, bytes) != hipSuccess) ok = false; };
    if (const char *e = measure_env("OATGPU_NB")) { const int v = atoi(e); if (v >= 1 && v <= oatgpu_ctx::kNB) c->nb = v; }
    c->private_streams = measure_env("OATGPU_PRIVATE_STREAMS") != nullptr;     // measurement aid: the old layout
    if (c->private_streams) {
        ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
        c->own_stream = ok;
        for (int q = 0; q < c->nb && ok; ++q) ok = (c->stream_b[q] = create_b_stream()) != nullptr;
    } else {
        ok = acquire_streams(cfg->device, c->nb, &c->stream, c->stream_b);
        c->have_shared = ok;
    }
    c->expt = measure_env("OATGPU_EXPT") ? atoi(measure_env("OATGPU_EXPT")) : 0;   // measurement aid (bit mask)
    c->serial = measure_env("OATGPU_SERIAL") != nullptr;   // measurement aid: run the back half on stream A
    // Replaying the back half from a captured hipGraph is implemented but measured 0-5 % SLOWER than
    // plain launches on MI355X / ROCm 7.2: opt-in only.
    c->use_graph = measure_env("OATGPU_GRAPH") != nullptr && !c->serial;
    c->ring_slots = cfg->ring_depth;                  // slot = threshold buffer, slot % nb = scratch set / stream
    for (int q = 0; q < c->nb && ok; ++q) {
        ok = hipEventCreateWithFlags(&c->ev_k1[q], hipEventDisableTiming | ((c->expt & 2) ? 0 : hipEventDisableSystemFence)) == hipSuccess;
    }
    for (int q = 0; q < oatgpu_ctx::kSets && ok; ++q)
        ok = hipEventCreateWithFlags(&c->ev_blob[q], hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
    if (const char *e = measure_env("OATGPU_EARLY_BLOB")) c->early_blob = atoi(e);
    if (const char *e = measure_env("OATGPU_EARLY_MIN_PX")) c->early_min_px = (size_t)atoll(e);
    if (const char *e = measure_env("OATGPU_LONE_PLAIN")) c->lone_plain = atoi(e) != 0;
    if (const char *e = measure_env("OATGPU_PAIR_BACK")) c->pair_back = atoi(e) != 0;
    if (const char *e = measure_env("OATGPU_K1_WG")) { const int v = atoi(e); if (v == 64 || v == 256) c->k1_wg_force = v; }


    if (const char *e = measure_env("OATGPU_K1_STOP_EVENT")) c->k1_stop_event = atoi(e) != 0 ? 1 : 0;

    A((void **)&c->state, n * mog_stream_floats(g.Palloc) * sizeof(float));
    A((void **)&c->nmodes, n * PA);
    A((void **)&c->frames, n * npx * cfg->channels);
    A((void **)&c->aux_a, npx * 3);
    A((void **)&c->aux_b, npx * 3);
    // One threshold-bit buffer per ring slot: a slot is only reused after its result was collected,
    // i.e. after the back half that read its buffer has finished -- stream A never waits for a B stream.
    A((void **)&c->bb[0].thr, (size_t)c->ring_slots * n * NW * 8);
    A((void **)&c->nopark, sizeof(unsigned));
    if (ok && hipMemsetAsync(c->nopark, 0, sizeof(unsigned), c->stream) != hipSuccess) ok = false;
    const int nsets = c->nb >= 3 ? oatgpu_ctx::kSets : c->nb;      // (an out-of-memory shows HERE, as OATGPU_E_NOMEM from oatgpu_create)
    for (int q = 0; q < nsets && ok; ++q) ok = alloc_scratch_set(c, q);
    for (int q = 0; q < nsets && ok; ++q) c->have_set[q] = true;
    const size_t slots = (size_t)c->ring_slots + 1;
    if (ok && hipHostMalloc((void **)&c->res_host, slots * n * sizeof(ResultRec), hipHostMallocMapped) != hipSuccess)
        ok = false;
    if (ok && hipHostGetDevicePointer((void **)&c->res_dev, c->res_host, 0) != hipSuccess) ok = false;
    if (ok && hipHostMalloc((void **)&c->dens_host, 4 * sizeof(unsigned), hipHostMallocMapped) != hipSuccess) ok = false;
    if (ok && hipHostGetDevicePointer((void **)&c->dens_dev, c->dens_host, 0) != hipSuccess) ok = false;
    if (ok) memset(c->dens_host, 0, 4 * sizeof(unsigned));
    if (ok && hipMalloc((void **)&c->wild_sink, 8 * sizeof(unsigned long long)) != hipSuccess) ok = false;
    if (ok && hipMemset(c->wild_sink, 0, 8 * sizeof(unsigned long long)) != hipSuccess) ok = false;
    if (ok) {
        c->ring_ev.resize(c->ring_slots);
        c->back_graph.assign(c->ring_slots, nullptr);
        c->slot_filtered.assign(c->ring_slots, 0);
        c->slot_spec.assign(c->ring_slots + 1, 0);
        c->slot_q.assign(c->ring_slots + 1, 0);
        c->slot_st.assign(c->ring_slots + 1, 0);
        c->slot_repair.assign(c->ring_slots + 1, 0);
        c->slot_ev.resize(c->ring_slots);
        for (int i = 0; i < c->ring_slots; ++i) c->slot_ev[i] = i;
        for (auto &e : c->ring_ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming | ((c->expt & 8) ? hipEventDisableSystemFence : 0)) != hipSuccess) ok = false;
    }
    // the model's mode counters start at zero; everything else is written before it is read
    if (ok && hipMemsetAsync(c->nmodes, 0, n * PA, c->stream) != hipSuccess) ok = false;
    if (ok && hipMemsetAsync(c->state, 0, n * mog_stream_floats(g.Palloc) * sizeof(float), c->stream) != hipSuccess) ok = false;
    if (ok && hipMemsetAsync(c->bb[0].thr, 0, (size_t)c->ring_slots * n * NW * 8, c->stream) != hipSuccess) ok = false;
    if (ok && hipStreamSynchronize(c->stream) != hipSuccess) ok = false;
    if (!ok) {
        fail(nullptr, OATGPU_E_NOMEM, "device allocation failed: %s", hipGetErrorString(hipGetLastError()));
        free_all(c);
        return nullptr;
    }
    c->last_morph = c->bb[0].thr;
    c->last_fin = c->bb[0].fin;
    return c;
}

extern "C" void oatgpu_destroy(oatgpu_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->cfg.device);
    c->pend_valid = false;            // a registered frame nobody collected: nothing to launch for
    if (c->stream) hipStreamSynchronize(c->stream);
    for (int q = 0; q < oatgpu_ctx::kNB; ++q) if (c->stream_b[q] && c->b_used[q]) hipStreamSynchronize(c->stream_b[q]);
    free_all(c);
}

extern "C" const char *oatgpu_last_error(const oatgpu_ctx *c)
{
    return c ? c->err.c_str() : g_last_error.c_str();
}

extern "C" int oatgpu_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" int oatgpu_device_numa_node(int32_t device)
{
    char bus[32] = {0};
    if (device < 0 || device >= oatgpu_device_count()) return -1;             // (never hand HIP a bad ordinal: its error is sticky)
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char *p = bus; *p; ++p) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');      // sysfs spells it in lower case
    char path[96];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    return node;
}

extern "C" int oatgpu_host_register(void *ptr, size_t bytes)
{
    if (!ptr || !bytes) return fail(nullptr, OATGPU_E_INVALID, "null argument");
    hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) return fail(nullptr, OATGPU_E_HIP, "hipHostRegister failed: %s", hipGetErrorString(e));
    return OATGPU_OK;
}
extern "C" int oatgpu_host_unregister(void *ptr)
{
    if (!ptr) return OATGPU_E_INVALID;
    return hipHostUnregister(ptr) == hipSuccess ? OATGPU_OK : OATGPU_E_HIP;
}
extern "C" void *oatgpu_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void oatgpu_host_free(void *ptr) { if (ptr) hipHostFree(ptr); }

static int flush_pending(oatgpu_ctx *c);
extern "C" int oatgpu_set_stream(oatgpu_ctx *c, void *s)
{
    if (!c) return OATGPU_E_INVALID;
    { const int frc = flush_pending(c); if (frc) return frc; }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int q = 0; q < oatgpu_ctx::kNB; ++q) if (c->stream_b[q] && c->b_used[q]) HIPCHK(c, hipStreamSynchronize(c->stream_b[q]));
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s;
    c->own_stream = false;
    return OATGPU_OK;
}
extern "C" void *oatgpu_get_stream(oatgpu_ctx *c) { return c ? (void *)c->stream : nullptr; }
static int quiesce(oatgpu_ctx *c);
extern "C" int oatgpu_synchronize(oatgpu_ctx *c)
{
    if (!c) return OATGPU_E_INVALID;
    return quiesce(c);
}
extern "C" int oatgpu_set_stage_copy(oatgpu_ctx *c, int32_t mode)
{
    if (!c || (mode != 0 && mode != 1)) return fail(c, OATGPU_E_INVALID, "stage copy mode must be 0 (DMA) or 1 (kernel)");
    c->stage_kernel = mode == 1;
    return OATGPU_OK;
}

extern "C" int oatgpu_set_early_blob(oatgpu_ctx *c, int32_t on)
{
    if (!c) return OATGPU_E_INVALID;
    c->early_blob = on < 0 ? -1 : on != 0;
    return OATGPU_OK;
}

#ifdef OATGPU_MEASURE
// measurement builds only (tools/dense_placement_probe.py): where the context's model, counters and frames landed
extern "C" __attribute__((visibility("default"))) int oatgpu_debug_addresses(oatgpu_ctx *c, unsigned long long *out4)
{
    if (!c || !out4) return OATGPU_E_INVALID;
    out4[0] = (unsigned long long)(uintptr_t)c->state; out4[1] = (unsigned long long)(uintptr_t)c->nmodes;
    out4[2] = (unsigned long long)(uintptr_t)c->bb[0].thr; out4[3] = (unsigned long long)(mog_stream_floats(c->g.Palloc) * 4);
    return OATGPU_OK;
}
#endif

extern "C" int oatgpu_set_k1_workgroup(oatgpu_ctx *c, int32_t threads)
{
    if (!c || (threads != 0 && threads != 64 && threads != 256)) return fail(c, OATGPU_E_INVALID, "k1 workgroup must be 0 (by path), 64 or 256");
    c->k1_wg_force = threads;
    return OATGPU_OK;
}

extern "C" int oatgpu_last_step_shape(const oatgpu_ctx *c, int32_t *k1_workgroup, int32_t *early_blob)
{
    if (!c) return OATGPU_E_INVALID;
    if (k1_workgroup) *k1_workgroup = c->last_k1_wg;
    if (early_blob) *early_blob = c->last_step_early ? 1 : 0;
    return OATGPU_OK;
}

extern "C" int64_t oatgpu_early_blob_timeouts(const oatgpu_ctx *c) { return c ? (int64_t)c->early_timeouts : 0; }

extern "C" int oatgpu_set_fusion(oatgpu_ctx *c, int32_t frames_per_launch)
{
    if (!c) return OATGPU_E_INVALID;
    if (frames_per_launch != 1 && frames_per_launch != 2) return fail(c, OATGPU_E_INVALID, "frames_per_launch must be 1 or 2");
    const int rc = quiesce(c);
    if (rc) return rc;
    c->fuse = frames_per_launch;
    return OATGPU_OK;
}

static int quiesce(oatgpu_ctx *c);
static int flush_pending(oatgpu_ctx *c);

extern "C" int oatgpu_set_detector(oatgpu_ctx *c, int32_t h_lo, int32_t h_hi, int32_t s_lo, int32_t s_hi,
                                   int32_t v_lo, int32_t v_hi, int32_t erode, int32_t dilate,
                                   double min_area, double max_area)
{
    if (!c) return OATGPU_E_INVALID;
    oatgpu_config k = c->cfg;
    k.h_lo = h_lo; k.h_hi = h_hi; k.s_lo = s_lo; k.s_hi = s_hi; k.v_lo = v_lo; k.v_hi = v_hi;
    k.erode = erode; k.dilate = dilate; k.min_area = min_area; k.max_area = max_area;
    int rc = check_detector(c, k);
    if (rc) return rc;
    rc = quiesce(c);
    if (rc) return rc;
    c->cfg = k;
    for (auto &ge : c->back_graph) if (ge) { hipGraphExecDestroy(ge); ge = nullptr; }   // parameters are baked in
    return OATGPU_OK;
}

// ---- BackgroundSubtractorMOG2Impl::apply prologue for one camera stream ----
static Rate mog_begin(oatgpu_ctx *c, int s, double learningRate)
{
    Rate r;
    int &nf = c->nframes[s];
    const bool needToInitialize = nf == 0 || learningRate >= 1;
    r.fresh = needToInitialize ? 1 : 0;
    if (needToInitialize) { nf = 0; c->wild_model[(size_t)s] = 0; }       // (the launch builds the stream's model anew: plain)
    ++nf;
    const int lim = 2 * nf < c->cfg.history ? 2 * nf : c->cfg.history;
    learningRate = (learningRate >= 0 && nf > 1) ? learningRate : 1. / lim;
    r.alphaT = (float)learningRate;
    r.alpha1 = 1.f - r.alphaT;
    r.prune = (float)(-learningRate * c->cfg.ct);   // product in double, as the reference
    return r;
}

static u64 *thr_buf(oatgpu_ctx *c, int parity)
{
    return c->bb[0].thr + (size_t)parity * c->cfg.n_streams * (c->g.Palloc >> 6);
}

// The single-stage calls are synchronous and share scratch with the pipelined path:
// wait until both HIP streams have drained.
static int flush_pending(oatgpu_ctx *c);

static int repair_outstanding(oatgpu_ctx *c);

static int quiesce(oatgpu_ctx *c)
{
    if (c->defer_kind) return fail(c, OATGPU_E_INVALID, "a deferred result is waiting: fetch it first (oatgpu_fetch_frame / oatgpu_fetch_position)");
    const int frc = flush_pending(c);
    if (frc) return frc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int q = 0; q < oatgpu_ctx::kNB; ++q) if (c->stream_b[q] && c->b_used[q]) HIPCHK(c, hipStreamSynchronize(c->stream_b[q]));
    // Outstanding frames whose speculative back half declined them are redone NOW, while their threshold bits are still
    // in their ring slots: the single-stage calls that follow a quiesce() write theirs into slot 0's buffer.
    return repair_outstanding(c);
}

static MogLaunch mog_launch_base(oatgpu_ctx *c, const uint8_t *frames, const Rate &r)
{
    MogLaunch a{};
    a.frames = frames; a.channels = c->cfg.channels; a.state = c->state; a.nmodes = c->nmodes; a.thr_bits = thr_buf(c, 0);
    a.out_bgr = nullptr; a.out_mask = nullptr; a.out_base = 0; a.roi_bits = c->roi;
    a.alphaT = r.alphaT; a.alpha1 = r.alpha1; a.prune = r.prune; a.fresh = r.fresh; a.nt_loads = c->nt_loads ? 1 : 0;
    a.mp = mogparams_of(c->cfg);
    a.rp = range_of(c->cfg);
    a.audit = c->audit_on ? c->audit_dev : nullptr;
#ifdef OATGPU_RS_TIMING             // measurement builds: the per-pixel launch's last workgroups stamp the wall clock here (kernels_mog.hip)
    if (!c->rs_k1_end && hipMalloc((void **)&c->rs_k1_end, 64) == hipSuccess) hipMemset(c->rs_k1_end, 0, 64);
    a.rs_end = c->rs_k1_end;
    oatgpu::oatgpu_debug_rs_set_k1_end(c->rs_k1_end);
#endif
    if (c->audit_on) c->audit_launches++;
    return a;
}

static MogLaunchOpts mog_launch_opts(oatgpu_ctx *c, int s0, int s1, int wg)
{
    MogLaunchOpts o;
    o.wg = wg;
    o.wild_sink = c->wild_sink;
    o.frozen_ok = c->cfg.var_min <= c->cfg.var_init && c->cfg.var_init <= c->cfg.var_max && c->cfg.var_min > 0.f;
    for (int s = s0; s < s1; ++s) o.wild_model = o.wild_model || c->wild_model[(size_t)s] != 0;
    return o;
}

static int check_stream_ix(oatgpu_ctx *c, int s)
{
    if (!c) return OATGPU_E_INVALID;
    if (s < 0 || s >= c->cfg.n_streams) return fail(c, OATGPU_E_INVALID, "stream index %d out of range", s);
    return OATGPU_OK;
}

// the input of a single-stage call on its way to the device; deferred: with an event behind it
static int stage_in(oatgpu_ctx *c, void *dst, const void *src, size_t bytes)
{
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    if (c->deferred) {
        if (!c->ev_h2d) HIPCHK(c, hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming | hipEventDisableSystemFence));
        HIPCHK(c, hipEventRecord(c->ev_h2d, c->stream));
    }
    return OATGPU_OK;
}
// the output frame of a single-stage call: copied out and waited for -- or, deferred, left on the device once the INPUT
// has been read (FrameFilter.cpp:73-80: the reference posts its SOURCE right after its memcpy)
static int finish_frame(oatgpu_ctx *c, uint8_t *out, const uint8_t *dev, size_t bytes)
{
    if (c->deferred) {
        HIPCHK(c, hipEventSynchronize(c->ev_h2d));
        c->defer_kind = 1; c->defer_dev = dev; c->defer_bytes = bytes;       // (only once the wait succeeded: a failed call owes no fetch)
        return OATGPU_OK;
    }
    HIPCHK(c, hipMemcpyAsync(out, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OATGPU_OK;
}

extern "C" int oatgpu_set_deferred(oatgpu_ctx *c, int32_t on)
{
    if (!c) return OATGPU_E_INVALID;
    if (c->defer_kind) return fail(c, OATGPU_E_INVALID, "a deferred result is waiting: fetch it first");
    c->deferred = on != 0;
    return OATGPU_OK;
}

extern "C" int oatgpu_fetch_frame(oatgpu_ctx *c, uint8_t *out)
{
    if (!c || !out) return fail(c, OATGPU_E_INVALID, "null argument");
    if (c->defer_kind != 1) return fail(c, OATGPU_E_INVALID, "no deferred frame is waiting");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMemcpyAsync(out, c->defer_dev, c->defer_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->defer_kind = 0;                  // (cleared once the frame has arrived: a failed fetch can be repeated)
    return OATGPU_OK;
}

static int mog_single(oatgpu_ctx *c, int s, const uint8_t *bgr_in, uint8_t *mask_out, uint8_t *bgr_out, double lr)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!bgr_in) return fail(c, OATGPU_E_INVALID, "null frame");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const size_t npx = (size_t)c->g.H * c->g.W;
    const size_t ch = c->cfg.channels;
    uint8_t *slot = c->frames + (size_t)s * npx * ch;
    rc = stage_in(c, slot, bgr_in, npx * ch);
    if (rc) return rc;
    const Rate r = mog_begin(c, s, lr);
    MogLaunch a = mog_launch_base(c, c->frames, r);
    a.out_base = s;
    if (mask_out) a.out_mask = c->aux_a;
    if (bgr_out) a.out_bgr = c->aux_b;
    launch_mog_fused(c->g, a, s, 1, c->stream, nullptr, mog_launch_opts(c, s, s + 1, 256));
    HIPCHK(c, hipGetLastError());
    if (bgr_out && !mask_out) return finish_frame(c, bgr_out, c->aux_b, npx * ch);       // (deferrable: the filter form)
    if (mask_out) HIPCHK(c, hipMemcpyAsync(mask_out, c->aux_a, npx, hipMemcpyDeviceToHost, c->stream));
    if (bgr_out) HIPCHK(c, hipMemcpyAsync(bgr_out, c->aux_b, npx * ch, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OATGPU_OK;
}

extern "C" int oatgpu_set_kalman(oatgpu_ctx *c, int32_t enable, double dt, double timeout, double sigma_accel,
                                 double sigma_noise)
{
    if (!c) return OATGPU_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->ring_count) return fail(c, OATGPU_E_INVALID, "set_kalman while enqueued results are outstanding");
    int rc = quiesce(c);
    if (rc) return rc;
    if (!enable) { c->kal_on = false; return OATGPU_OK; }
    if (!(dt > 0) || !(timeout >= 0) || !(sigma_accel >= 0) || !(sigma_noise >= 0) || !(timeout / dt < 2147483647.0))
        return fail(c, OATGPU_E_INVALID, "kalman: dt must be > 0, timeout / sigma-accel / sigma-noise >= 0");
    if (!c->kal.state) HIPCHK(c, hipMalloc((void **)&c->kal.state, (size_t)c->cfg.n_streams * sizeof(KalmanState)));
    c->kal.dt = dt; c->kal.sig_accel = sigma_accel; c->kal.sig_noise = sigma_noise;
    c->kal.threshold = (int)(timeout / dt);                     // KalmanFilter2D.cpp:74-76
    launch_kalman_reset(c->kal.state, c->cfg.n_streams, c->kal_ticket, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->kal_on = true;
    return OATGPU_OK;
}

// HomographyTransform2D::filter (HomographyTransform2D.cpp:62-107) on one result: cv::perspectiveTransform of a
// CV_64FC2 point (OpenCV 3.1.0 core/matmul.cpp perspectiveTransform_<double>: w = x m6 + y m7 + m8; |w| > FLT_EPSILON ->
// multiply by 1/w, else (0, 0)); the velocity through the same matrix with its offsets zeroed (:79-89).
static void perspective_point(const double *m, double &px, double &py)
{
    const double x = px, y = py;
    double w = x * m[6] + y * m[7] + m[8];
    if (fabs(w) > (double)FLT_EPSILON) {
        w = 1. / w;
        px = (x * m[0] + y * m[1] + m[2]) * w;
        py = (x * m[3] + y * m[4] + m[5]) * w;
    } else {
        px = py = 0;
    }
}
static void apply_homography(const oatgpu_ctx *c, oatgpu_position *o)
{
    if (o->valid) perspective_point(c->homo, o->x, o->y);
    if (o->velocity_valid) {
        double v[9];
        for (int i = 0; i < 9; ++i) v[i] = c->homo[i];
        v[2] = 0.0; v[5] = 0.0;
        perspective_point(v, o->vx, o->vy);
    }
}

extern "C" int oatgpu_set_homography(oatgpu_ctx *c, int32_t enable, const double *h9)
{
    if (!c) return OATGPU_E_INVALID;
    if (enable && !h9) return fail(c, OATGPU_E_INVALID, "null homography");
    if (c->ring_count) return fail(c, OATGPU_E_INVALID, "set_homography while enqueued results are outstanding");
    c->homo_on = enable != 0;
    if (enable) for (int i = 0; i < 9; ++i) c->homo[i] = h9[i];
    return OATGPU_OK;
}

extern "C" int oatgpu_set_roi_mask(oatgpu_ctx *c, int32_t s, const uint8_t *roi_mask)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const Geom &g = c->g;
    const size_t NW = g.Palloc >> 6, n = c->cfg.n_streams, npx = (size_t)g.H * g.W;
    if (!c->roi) {
        if (!roi_mask) return OATGPU_OK;                       // nothing to clear
        HIPCHK(c, hipMalloc((void **)&c->roi, n * NW * 8));
        HIPCHK(c, hipMemsetAsync(c->roi, 0xff, n * NW * 8, c->stream));   // other streams keep everything
    }
    if (!roi_mask) {
        HIPCHK(c, hipMemsetAsync(c->roi + (size_t)s * NW, 0xff, NW * 8, c->stream));
    } else {
        HIPCHK(c, hipMemcpyAsync(c->aux_a, roi_mask, npx, hipMemcpyHostToDevice, c->stream));
        launch_pack_bits(g, c->aux_a, c->roi + (size_t)s * NW, c->stream);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OATGPU_OK;
}

extern "C" int oatgpu_mog_apply(oatgpu_ctx *c, int32_t s, const uint8_t *bgr_in, uint8_t *fgmask_out, double lr)
{
    if (!fgmask_out) return fail(c, OATGPU_E_INVALID, "null mask buffer");
    return mog_single(c, s, bgr_in, fgmask_out, nullptr, lr);
}

extern "C" int oatgpu_mog_filter(oatgpu_ctx *c, int32_t s, const uint8_t *bgr_in, uint8_t *bgr_out, double lr)
{
    if (!bgr_out) return fail(c, OATGPU_E_INVALID, "null output frame");
    return mog_single(c, s, bgr_in, nullptr, bgr_out, lr);
}

extern "C" int oatgpu_bsub_filter(oatgpu_ctx *c, int32_t s, const uint8_t *in, uint8_t *out, double alpha)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!in || !out) return fail(c, OATGPU_E_INVALID, "null argument");
    if (!(alpha >= 0.0 && alpha <= 1.0)) return fail(c, OATGPU_E_INVALID, "adaptation-coeff must be in [0,1]");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const size_t nb = (size_t)c->g.H * c->g.W * c->cfg.channels;
    if (!c->bsub_bg) {
        HIPCHK(c, hipMalloc((void **)&c->bsub_bg, (size_t)c->cfg.n_streams * nb));
        HIPCHK(c, hipMalloc((void **)&c->bsub_f, (size_t)c->cfg.n_streams * nb * sizeof(float)));
    }
    if (c->bsub_have[s] == 2 && alpha > 0.0)        // the reference's accumulateWeighted asserts on its empty fp32 image
        return fail(c, OATGPU_E_INVALID, "a background image from a file cannot adapt (adaptation-coeff must be 0)");
    rc = stage_in(c, c->aux_a, in, nb);
    if (rc) return rc;
    const float a = (float)alpha, b = 1 - a;              // accW_: AT a = (AT)alpha, b = 1 - a
    launch_bsub(c->aux_a, c->aux_b, c->bsub_bg + (size_t)s * nb, c->bsub_f + (size_t)s * nb, nb, a, b,
                c->bsub_have[s] ? 0 : 1, alpha > 0.0 ? 1 : 0, c->stream);
    HIPCHK(c, hipGetLastError());
    if (!c->bsub_have[s]) c->bsub_have[s] = 1;
    return finish_frame(c, out, c->aux_b, nb);
}

extern "C" int oatgpu_bsub_set_background(oatgpu_ctx *c, int32_t s, const uint8_t *image)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!image) return fail(c, OATGPU_E_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const size_t nb = (size_t)c->g.H * c->g.W * c->cfg.channels;
    if (!c->bsub_bg) {
        HIPCHK(c, hipMalloc((void **)&c->bsub_bg, (size_t)c->cfg.n_streams * nb));
        HIPCHK(c, hipMalloc((void **)&c->bsub_f, (size_t)c->cfg.n_streams * nb * sizeof(float)));
    }
    HIPCHK(c, hipMemcpyAsync(c->bsub_bg + (size_t)s * nb, image, nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->bsub_have[s] = 2;          // from a file: the fp32 accumulator does not exist (BackgroundSubtractor.cpp:63-71)
    return OATGPU_OK;
}

extern "C" int oatgpu_mask_filter(oatgpu_ctx *c, int32_t s, const uint8_t *in, uint8_t *out)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!in || !out) return fail(c, OATGPU_E_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const size_t nb = (size_t)c->g.H * c->g.W * c->cfg.channels;
    if (!c->roi) {                                   // no mask set: FrameMasker::filter leaves the frame alone
        if (c->deferred) {                           // (through the device, so that the fetch finds it)
            rc = stage_in(c, c->aux_b, in, nb);
            return rc ? rc : finish_frame(c, out, c->aux_b, nb);
        }
        if (out != in) memcpy(out, in, nb);
        return OATGPU_OK;
    }
    rc = stage_in(c, c->aux_a, in, nb);
    if (rc) return rc;
    launch_apply_roi(c->g, c->aux_a, c->aux_b, c->cfg.channels, c->roi + (size_t)s * (c->g.Palloc >> 6), c->stream);
    HIPCHK(c, hipGetLastError());
    return finish_frame(c, out, c->aux_b, nb);
}

extern "C" int oatgpu_thresh_filter(oatgpu_ctx *c, const uint8_t *in, uint8_t *out, int32_t i_min, int32_t i_max)
{
    if (!c || !in || !out) return fail(c, OATGPU_E_INVALID, "null argument");
    if (i_min < 0 || i_min > 256 || i_max < 0 || i_max > 256)      // Threshold.cpp:62-63
        return fail(c, OATGPU_E_INVALID, "Values of intensity should be between 0 and 256.");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = quiesce(c);
    if (rc) return rc;
    const size_t npx = (size_t)c->g.H * c->g.W, nb = npx * c->cfg.channels;
    int lo, hi;
    norm_range(i_min, i_max, lo, hi);
    rc = stage_in(c, c->aux_a, in, nb);
    if (rc) return rc;
    launch_thresh_filter(c->aux_a, c->aux_b, npx, c->cfg.channels, lo, hi, c->stream);
    HIPCHK(c, hipGetLastError());
    return finish_frame(c, out, c->aux_b, nb);
}

extern "C" int oatgpu_bgr2hsv(oatgpu_ctx *c, const uint8_t *bgr_in, uint8_t *hsv_out)
{
    if (!c || !bgr_in || !hsv_out) return fail(c, OATGPU_E_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->defer_kind) return fail(c, OATGPU_E_INVALID, "a deferred result is waiting: fetch it first");
    const size_t npx = (size_t)c->g.H * c->g.W;
    { const int rc = stage_in(c, c->aux_a, bgr_in, npx * 3); if (rc) return rc; }
    launch_bgr2hsv(c->aux_a, c->aux_b, npx, c->stream);
    HIPCHK(c, hipGetLastError());
    return finish_frame(c, hsv_out, c->aux_b, npx * 3);
}

// ColorConvert::filter for any pair of oat::PixelColor values: oat::color_conv_table (Color.h:45-51) picks the
// cvtColor code, color_conv_code (Color.h:88-95) refuses the impossible pairs and ColorConvert::connectToNode
// (ColorConvert.cpp:79-85) the ones with nothing to do -- same texts here.
extern "C" int oatgpu_cvt_color(oatgpu_ctx *c, int32_t from_color, int32_t to_color, const uint8_t *in, uint8_t *out)
{
    if (!c || !in || !out) return fail(c, OATGPU_E_INVALID, "null argument");
    static const char *const names[4] = {"BINARY", "GREY", "BGR", "HSV"};
    if (from_color < 0 || from_color > 3 || to_color < 0 || to_color > 3) return fail(c, OATGPU_E_INVALID, "Invalid color.");
    //                       to: BINARY GREY BGR HSV        -1 nothing to do, -2 not possible, 3 = BGR -> HSV
    static const int table[4][4] = {{-1, -1, 1, -2},      // from BINARY
                                    {-1, -1, 1, -2},      // from GREY
                                    {0, 0, -1, 3},        // from BGR
                                    {-2, -2, 2, -1}};     // from HSV
    const int code = table[from_color][to_color];
    if (code == -2) return fail(c, OATGPU_E_INVALID, "Requested color conversion is not possible.");
    if (code == -1)
        return fail(c, OATGPU_E_INVALID, "Nothing to be done for %s to %s conversion.", names[from_color], names[to_color]);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t npx = (size_t)c->g.H * c->g.W;
    const size_t nin = npx * (from_color >= 2 ? 3 : 1), nout = npx * (to_color >= 2 ? 3 : 1);
    if (c->defer_kind) return fail(c, OATGPU_E_INVALID, "a deferred result is waiting: fetch it first");
    { const int rc = stage_in(c, c->aux_a, in, nin); if (rc) return rc; }
    if (code == 3) launch_bgr2hsv(c->aux_a, c->aux_b, npx, c->stream);
    else launch_cvt_color(code, c->aux_a, c->aux_b, npx, c->stream);
    HIPCHK(c, hipGetLastError());
    return finish_frame(c, out, c->aux_b, nout);
}

// contourMoments' epilogue (imgproc/moments.cpp) + siftContours' centroid
// (DetectorFunc.cpp:54-62) on the exact integer sums.
static void to_position(const ResultRec &r, oatgpu_position *o)
{
    memset(o, 0, sizeof *o);
    o->first_pixel = -1;
    if (!r.valid) return;
    const double a00 = (double)r.a00, a10 = (double)r.a10, a01 = (double)r.a01;
    double db1_2, db1_6;
    if (a00 > 0) { db1_2 = 0.5; db1_6 = 0.16666666666666666666666666666667; }
    else { db1_2 = -0.5; db1_6 = -0.16666666666666666666666666666667; }
    const double m00 = a00 * db1_2, m10 = a10 * db1_6, m01 = a01 * db1_6;
    o->valid = 1;
    o->first_pixel = r.first_pixel;
    o->x = m10 / m00;
    o->y = m01 / m00;
    o->area = m00;
    o->a00 = r.a00; o->a10 = r.a10; o->a01 = r.a01;
    o->raw_valid = 1; o->raw_x = o->x; o->raw_y = o->y;
}

// posifilt kalman's view of the token (KalmanFilter2D.cpp:123-137): position/velocity = predicted state
static void apply_kalman(const ResultRec &r, oatgpu_position *o)
{
    o->valid = r.kal_valid;
    o->velocity_valid = r.kal_valid;
    o->x = r.kx; o->y = r.ky; o->vx = r.kvx; o->vy = r.kvy;
}

// erode -> (dilate fused into the row scan) -> blob for camera streams [s0, s0+n), reading
// the threshold bits `thr`; results land in host-mapped slot `slot`.  All on HIP stream st.
static int back_half(oatgpu_ctx *c, BlobBuffers &bb, const u64 *thr, int s0, int n, int slot, hipStream_t st,
                     hipEvent_t ev_mid, int erode_k = -1, int dilate_k = -1, int mode = kBlobFull)
{
    const Geom &g = c->g;
    const u64 *src = thr;
    if (erode_k < 0) erode_k = c->cfg.erode;
    if (dilate_k < 0) dilate_k = c->cfg.dilate;
    const int dil = dilate_k > 1 ? dilate_k : 0;
    int ero = erode_k > 1 ? erode_k : 0;
    if (ero && rowscan_lds_bytes(g, dil) > kRowscanLdsMax) {     // very wide rows x large dilation
        launch_morph(g, src, bb.tmp, ero, true, s0, n, st);
        src = bb.tmp;
        ero = 0;
    }
    if (ev_mid) HIPCHK(c, hipEventRecord(ev_mid, st));
    c->last_morph = (dil || ero) ? bb.morph : src;
    c->last_fin = bb.fin;
    ResultRec *rd = c->res_dev + (size_t)slot * c->cfg.n_streams;
    launch_blob(g, bb, src, ero, dil, c->cfg.min_area, c->cfg.max_area, rd, s0, n, st, mode);
    HIPCHK(c, hipGetLastError());
    return OATGPU_OK;
}

static int detect_single(oatgpu_ctx *c, int s, const uint8_t *in, int channels, oatgpu_position *out)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!in || (!out && !c->deferred)) return fail(c, OATGPU_E_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const Geom &g = c->g;
    const size_t npx = (size_t)g.H * g.W;
    rc = stage_in(c, c->aux_a, in, npx * channels);
    if (rc) return rc;
    RangeParams rp = range_of(c->cfg);
    launch_inrange_bits(g, c->aux_a, channels, rp, thr_buf(c, 0) + (size_t)s * (g.Palloc >> 6), c->stream);
    const int slot = c->ring_slots;       // the extra slot
    rc = back_half(c, c->bb[0], thr_buf(c, 0), s, 1, slot, c->stream, nullptr);
    if (rc) return rc;
    c->last_q = 0;
    if (c->deferred) {                    // the frame has been read; the position stays with the device (oatgpu_fetch_position)
        HIPCHK(c, hipEventSynchronize(c->ev_h2d));
        c->defer_kind = 2; c->defer_s = s;
        return OATGPU_OK;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    to_position(c->res_host[(size_t)slot * c->cfg.n_streams + s], out);
    return OATGPU_OK;
}

extern "C" int oatgpu_detect_diff(oatgpu_ctx *c, int32_t s, const uint8_t *grey_in, oatgpu_position *out)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!grey_in || (!out && !c->deferred)) return fail(c, OATGPU_E_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const Geom &g = c->g;
    const size_t npx = (size_t)g.H * g.W;
    if (!c->diff_last) HIPCHK(c, hipMalloc((void **)&c->diff_last, (size_t)c->cfg.n_streams * npx));
    rc = stage_in(c, c->aux_a, grey_in, npx);
    if (rc) return rc;
    const int have = c->diff_have[s];
    launch_absdiff_bits(g, c->aux_a, c->diff_last + (size_t)s * npx, c->cfg.diff_threshold, have,
                        thr_buf(c, 0) + (size_t)s * (g.Palloc >> 6), c->stream);
    c->diff_have[s] = 1;
    const int slot = c->ring_slots;
    // the first frame is analysed unblurred (DifferenceDetector.cpp:167-171); afterwards blur == dilation
    rc = back_half(c, c->bb[0], thr_buf(c, 0), s, 1, slot, c->stream, nullptr, 0, have ? c->cfg.blur : 0);
    if (rc) return rc;
    c->last_q = 0;
    if (c->deferred) {
        HIPCHK(c, hipEventSynchronize(c->ev_h2d));
        c->defer_kind = 2; c->defer_s = s;
        return OATGPU_OK;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    to_position(c->res_host[(size_t)slot * c->cfg.n_streams + s], out);
    return OATGPU_OK;
}

extern "C" int oatgpu_detect_hsv(oatgpu_ctx *c, int32_t s, const uint8_t *hsv_in, oatgpu_position *out)
{
    return detect_single(c, s, hsv_in, 3, out);
}
extern "C" int oatgpu_detect_thresh(oatgpu_ctx *c, int32_t s, const uint8_t *grey_in, oatgpu_position *out)
{
    return detect_single(c, s, grey_in, 1, out);
}

extern "C" int oatgpu_fetch_position(oatgpu_ctx *c, oatgpu_position *out)
{
    if (!c || !out) return fail(c, OATGPU_E_INVALID, "null argument");
    if (c->defer_kind != 2) return fail(c, OATGPU_E_INVALID, "no deferred position is waiting");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->defer_kind = 0;
    to_position(c->res_host[(size_t)c->ring_slots * c->cfg.n_streams + c->defer_s], out);
    return OATGPU_OK;
}

// ------------------------------------------------------------ fused path ----

static void prof_fold(oatgpu_ctx *c)
{
    if (!c->prof_used) return;
    hipStreamSynchronize(c->stream);
    for (int q = 0; q < oatgpu_ctx::kNB; ++q) if (c->stream_b[q] && c->b_used[q]) hipStreamSynchronize(c->stream_b[q]);
    for (size_t i = 0; i < c->prof_used; ++i) {
        float a = 0, b = 0, d = 0, t = 0;
        ProfStep &p = c->prof_steps[i];
        hipEventElapsedTime(&a, p.e[0], p.e[1]);   // fused per-pixel kernel (stream A)
        hipEventElapsedTime(&b, p.e[2], p.e[3]);   // erode (stream B)
        hipEventElapsedTime(&d, p.e[3], p.e[4]);   // dilate + labelling + sums + selection (stream B)
        hipEventElapsedTime(&t, p.e[0], p.e[4]);   // latency of the frame through both streams
        // (a host thread descheduled between the first event's record and the launch call puts its absence INTO the event
        // pair: one 10 ms sample among 145 turned a 100 us average into 238 us, profiles/r07z run of the driver's arguments.  A
        // sample eight times the average of those taken so far is the host's, not the kernel's: dropped.)
        // (the drops are COUNTED -- oatgpu_profile.dropped -- and bounded: four in a row are no host hiccup but a
        // change of regime, e.g. one-frame steps on a sparse model followed by two-frame steps on a dense one; the sample is
        // then taken and the streak starts anew, so the profile cannot freeze on its first eight samples.)
        // ... and a host stall is a millisecond or more: on a 70 x 200 frame, whose launch takes 5 us, ordinary jitter is "eight
        // times the average" (a dropped sample failed test_two_frames_a_launch_... once in this round's runs)
        const double mean_ms = c->prof_sum.steps ? c->prof_sum.mog_ms / (double)c->prof_sum.steps : 0.0;
        if (c->prof_sum.steps >= 8 && a > 8.0 * mean_ms && a > mean_ms + 1.0 && c->prof_drop_streak < 4) {
            c->prof_sum.dropped += 1;
            c->prof_drop_streak++;
            continue;
        }
        c->prof_drop_streak = 0;
        c->prof_sum.steps += 1;
        c->prof_sum.mog_frames += p.frames;
        c->prof_sum.mog_ms += a; c->prof_sum.morph_ms += b; c->prof_sum.blob_ms += d; c->prof_sum.total_ms += t;
    }
    c->prof_used = 0;
}

// Capture the back half of ring slot `slot` (scratch set slot % nb, result record of that slot)
// into an executable graph: the dependent launches become one hipGraphLaunch of host work.
static hipGraphExec_t capture_back_half(oatgpu_ctx *c, int slot, hipStream_t B)
{
    const int q = slot % c->nb;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (hipStreamBeginCapture(B, hipStreamCaptureModeRelaxed) != hipSuccess) return nullptr;
    const int rc = back_half(c, c->bb[q], thr_buf(c, slot), 0, c->cfg.n_streams, slot, B, nullptr);
    const hipError_t e = hipStreamEndCapture(B, &graph);
    if (rc != OATGPU_OK || e != hipSuccess || !graph) { if (graph) hipGraphDestroy(graph); return nullptr; }
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
    hipGraphDestroy(graph);
    return exec;
}

static int enqueue_frames(oatgpu_ctx *c, const void *frames_dev, double lr, hipEvent_t frames_ready);

extern "C" int oatgpu_track_enqueue_dev(oatgpu_ctx *c, const void *frames_dev, double lr)
{
    const int rc = enqueue_frames(c, frames_dev, lr, nullptr);
    if (!rc) c->dev_unconsumed = true;
    return rc;
}

// device staging ring + copy stream of the host-frame path (first use)
static int ensure_host_ring(oatgpu_ctx *c)
{
    if (c->frames_ring) return OATGPU_OK;
    const size_t sb = (size_t)c->g.H * c->g.W * c->cfg.channels * c->cfg.n_streams;
    if (c->private_streams) HIPCHK(c, hipStreamCreateWithFlags(&c->stream_c, hipStreamNonBlocking));
    else c->stream_c = acquire_copy_stream(c->cfg.device);
    if (!c->stream_c) return fail(c, OATGPU_E_HIP, "could not create the copy stream");
    HIPCHK(c, hipMalloc((void **)&c->frames_ring, sb * c->ring_slots));
    c->copy_ev.resize(c->ring_slots);
    for (auto &e : c->copy_ev) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    return OATGPU_OK;
}
static int ensure_stream_events(oatgpu_ctx *c)
{
    if (!c->copy_ev_s.empty()) return OATGPU_OK;
    c->copy_ev_s.resize((size_t)c->ring_slots * c->cfg.n_streams);
    for (auto &e : c->copy_ev_s) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    return OATGPU_OK;
}

// The host-frame path camera by camera: oatgpu_track_stage starts ONE camera's H2D copy into the next frame set's
// staging slot; when all n_streams are on their way oatgpu_track_enqueue_staged registers the set exactly as
// oatgpu_track_enqueue would have.
extern "C" int oatgpu_track_stage(oatgpu_ctx *c, int32_t stream_ix, const uint8_t *frame_host)
{
    if (!c || !frame_host) return fail(c, OATGPU_E_INVALID, "null argument");
    const int n = c->cfg.n_streams;
    if (stream_ix < 0 || stream_ix >= n) return fail(c, OATGPU_E_INVALID, "stream index %d out of range", stream_ix);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->broken) return fail(c, OATGPU_E_HIP, "context unusable after a failed launch (destroy it): %s", c->err.c_str());
    if (c->defer_kind) return fail(c, OATGPU_E_INVALID, "a deferred result is waiting: fetch it first (oatgpu_fetch_frame / oatgpu_fetch_position)");
    if (c->staged_count == 0) {
        if (c->ring_count == c->cfg.ring_depth) return fail(c, OATGPU_E_RING_FULL, "result ring full: collect first");
        { const int rc = ensure_host_ring(c); if (rc) return rc; }
        { const int rc = ensure_stream_events(c); if (rc) return rc; }
        c->staged.assign((size_t)n, 0);
        c->stage_slot = (int)(c->enq_total % (unsigned long long)c->ring_slots);
    }
    if (c->staged[(size_t)stream_ix]) return fail(c, OATGPU_E_INVALID, "stream %d is already staged for this frame set", stream_ix);
    const size_t fb = (size_t)c->g.H * c->g.W * c->cfg.channels;
    uint8_t *dst = c->frames_ring + (size_t)c->stage_slot * fb * n;
    if (!c->stream_c2 && n > 1 && !c->private_streams) {
        hipStream_t s2 = acquire_copy_stream2(c->cfg.device);
        // (the event first: a context that shows a second copy stream always has the event enqueue_staged records on it)
        if (s2 && !c->ev_c2) HIPCHK(c, hipEventCreateWithFlags(&c->ev_c2, hipEventDisableTiming | hipEventDisableSystemFence));
        c->stream_c2 = s2;
    }
    hipStream_t cs = (c->stream_c2 && (c->staged_count & 1)) ? c->stream_c2 : c->stream_c;
    bool copied = false;
    if (c->stage_kernel && ((uintptr_t)frame_host & 15u) == 0 && ((uintptr_t)(dst + (size_t)stream_ix * fb) & 15u) == 0) {
        void *dsrc = nullptr;                   // page-locked and mapped (oatgpu_host_register / oatgpu_host_alloc)?
        if (hipHostGetDevicePointer(&dsrc, (void *)frame_host, 0) == hipSuccess && dsrc) {
            launch_stage_copy(dsrc, dst + (size_t)stream_ix * fb, fb, cs);
            HIPCHK(c, hipGetLastError());
            copied = true;
        } else {
            (void)hipGetLastError();            // ordinary memory: the DMA path below bounces it
        }
    }
    if (!copied) HIPCHK(c, hipMemcpyAsync(dst + (size_t)stream_ix * fb, frame_host, fb, hipMemcpyHostToDevice, cs));
    HIPCHK(c, hipEventRecord(c->copy_ev_s[(size_t)c->stage_slot * n + stream_ix], cs));
    // Submit NOW.  ROCm 7.2 keeps a stream's queued copy back until something flushes the stream -- measured with the
    // tracker's own clock (oat-track-hip --timing, profiles/r04g_pipeline_ncam_before_flush.txt): every camera's copy started
    // only when the loop began to WAIT for it (its hipEventQuery), so the 8 x 112 us of link time of a round ran strictly
    // one after the other with the host's 300 us of per-round work instead of under it.  A stream query is the flush.
    (void)hipStreamQuery(cs);
    c->staged[(size_t)stream_ix] = 1;
    c->staged_count++;
    return OATGPU_OK;
}

extern "C" int oatgpu_track_enqueue_staged(oatgpu_ctx *c, double lr)
{
    if (!c) return OATGPU_E_INVALID;
    const int n = c->cfg.n_streams;
    if (c->staged_count != n) return fail(c, OATGPU_E_INVALID, "%d of %d streams staged", c->staged_count, n);
    if (c->broken) return fail(c, OATGPU_E_HIP, "context unusable after a failed launch (destroy it): %s", c->err.c_str());
    if (c->defer_kind) return fail(c, OATGPU_E_INVALID, "a deferred result is waiting: fetch it first (oatgpu_fetch_frame / oatgpu_fetch_position)");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const int slot = c->stage_slot;
    const size_t sb = (size_t)c->g.H * c->g.W * c->cfg.channels * n;
    if (c->stream_c2) {                                   // the set is complete when BOTH copy streams are through
        HIPCHK(c, hipEventRecord(c->ev_c2, c->stream_c2));
        HIPCHK(c, hipStreamWaitEvent(c->stream_c, c->ev_c2, 0));
    }
    HIPCHK(c, hipEventRecord(c->copy_ev[slot], c->stream_c));
    c->last_copy_slot = slot;
    c->last_copy_per_stream = true;
    c->per_stream_copy_ev = true;
    c->staged_count = 0;
    const int rc = enqueue_frames(c, c->frames_ring + (size_t)slot * sb, lr, c->copy_ev[slot]);
    return rc;
}

// A partly staged frame set is given up (a camera ended in the middle of a round, an error after the first
// oatgpu_track_stage): the copies already on their way are waited for -- they write into the staging slot, which the
// next set will reuse -- and the set is forgotten; nothing was registered, no result is owed for it.
extern "C" int oatgpu_track_stage_abort(oatgpu_ctx *c)
{
    if (!c) return OATGPU_E_INVALID;
    if (c->staged_count == 0) return OATGPU_OK;
    const int n = c->cfg.n_streams;
    hipError_t worst = hipSetDevice(c->cfg.device);
    for (int s = 0; s < n && c->stage_slot >= 0; ++s)
        if (c->staged[(size_t)s]) {
            const hipError_t e = hipEventSynchronize(c->copy_ev_s[(size_t)c->stage_slot * n + s]);
            if (e != hipSuccess) worst = e;
        }
    c->staged.assign((size_t)n, 0);
    c->staged_count = 0;
    c->stage_slot = -1;
    if (worst != hipSuccess) return fail(c, OATGPU_E_HIP, "oatgpu_track_stage_abort: %s", hipGetErrorString(worst));
    return OATGPU_OK;
}

extern "C" int oatgpu_track_enqueue(oatgpu_ctx *c, const uint8_t *const *frames_host, int32_t n, double lr)
{
    if (!c || !frames_host) return fail(c, OATGPU_E_INVALID, "null argument");
    if (n != c->cfg.n_streams) return fail(c, OATGPU_E_INVALID, "expected %d frames, got %d", c->cfg.n_streams, n);
    // (refused BEFORE a copy out of the caller's frames is queued -- enqueue_frames checks the same again, but by
    // then stream C would be reading buffers of a call that fails, and input_consumed would wait on a set nobody registered)
    if (c->broken) return fail(c, OATGPU_E_HIP, "context unusable after a failed launch (destroy it): %s", c->err.c_str());
    if (c->defer_kind) return fail(c, OATGPU_E_INVALID, "a deferred result is waiting: fetch it first (oatgpu_fetch_frame / oatgpu_fetch_position)");
    if (c->ring_count == c->cfg.ring_depth) return fail(c, OATGPU_E_RING_FULL, "result ring full: collect first");
    for (int s = 0; s < n; ++s) if (!frames_host[s]) return fail(c, OATGPU_E_INVALID, "null frame %d", s);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t fb = (size_t)c->g.H * c->g.W * c->cfg.channels, sb = fb * n;
    if (c->staged_count) return fail(c, OATGPU_E_INVALID, "a frame set is being staged (oatgpu_track_stage): finish it with oatgpu_track_enqueue_staged");
    { const int rc = ensure_host_ring(c); if (rc) return rc; }
    // the slot's staging buffer was last read by the K1 of the frame collected from this slot
    const int slot = (int)(c->enq_total % (unsigned long long)c->ring_slots);
    uint8_t *dst = c->frames_ring + (size_t)slot * sb;
    if (c->per_stream_copy_ev) { const int rc = ensure_stream_events(c); if (rc) return rc; }
    for (int s = 0; s < n; ++s) {
        HIPCHK(c, hipMemcpyAsync(dst + (size_t)s * fb, frames_host[s], fb, hipMemcpyHostToDevice, c->stream_c));
        if (c->per_stream_copy_ev && s + 1 < n) HIPCHK(c, hipEventRecord(c->copy_ev_s[(size_t)slot * n + s], c->stream_c));
    }
    HIPCHK(c, hipEventRecord(c->copy_ev[slot], c->stream_c));
    (void)hipStreamQuery(c->stream_c);                    // submit the copies now (see oatgpu_track_stage)
    c->last_copy_slot = slot;
    c->last_copy_per_stream = c->per_stream_copy_ev;
    return enqueue_frames(c, dst, lr, c->copy_ev[slot]);
}

// The kernels of one frame (nj == 1) or of two consecutive frames (nj == 2: K1 once for both where the streams'
// learning-rate schedules allow, then each frame's back half on its own B stream).
static int launch_jobs(oatgpu_ctx *c, const oatgpu_ctx::FrameJob *j, int nj, bool lone = false)
{
    const int n = c->cfg.n_streams;
    hipStream_t A = c->stream;

    ProfStep *ps = nullptr;
    if (c->prof && (c->prof_tick++ % (unsigned long long)c->prof_every) == 0) {
        if (c->prof_used == c->prof_steps.size()) {
            if (c->prof_steps.size() >= 1024) prof_fold(c);
            else {
                ProfStep p;
                for (auto &e : p.e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableSystemFence));   // timing, device-scope fence
                c->prof_steps.push_back(p);
            }
        }
        ps = &c->prof_steps[c->prof_used++];
        ps->frames = nj;
    }

    // Stream A: the fused per-pixel kernel of THIS step may start while the B streams are still
    // analysing earlier frames' masks.  It writes the threshold buffer of its own ring slot, whose
    // previous reader finished before that slot's result was collected: nothing to wait for.
    for (int i = 0; i < nj; ++i)
        if (j[i].ready) HIPCHK(c, hipStreamWaitEvent(A, j[i].ready, 0));
    if (ps) HIPCHK(c, hipEventRecord(ps->e[0], A));

    // every camera stream advances one frame per job; launches are batched while the streams share a
    // learning-rate schedule (they do unless the single-stage calls were used unevenly)
    std::vector<Rate> &rates = c->rates_scratch;
    rates.resize((size_t)n * nj);
    for (int i = 0; i < nj; ++i)
        for (int s = 0; s < n; ++s) rates[(size_t)i * n + s] = mog_begin(c, s, j[i].lr);
    auto same = [&](int s1, int s0) {
        for (int i = 0; i < nj; ++i)
            if (memcmp(&rates[(size_t)i * n + s1], &rates[(size_t)i * n + s0], sizeof(Rate)) != 0) return false;
        return true;
    };
    // Small frames are bound by the host's launch calls: the two back halves of a two-frame step
    // then go down ONE B stream behind one wait, and one ring event -- recorded behind the second -- covers both
    // results (8 runtime calls a step instead of 10): 3 x 320x240 78 k -> 108 k fps.  From about a megapixel a step on
    // the back halves are long enough to want a stream each (one 1080p stream: 45.8 k fps apart, 38.9 k fps together).
    const bool share_b = nj == 2 && (size_t)n * (size_t)c->g.P <= ((size_t)1 << 20) && !(c->expt & 1) && !c->use_graph && !c->serial;
    // Early dispatch of the blob workgroup: device frames (the copy streams of the host-frame path share hardware queues
    // with B streams, and a parked workgroup would hold the copies behind it up), steps of 4 MP and more (below that the
    // per-pixel launches are short, the wait for wave slots with them, and the step is bound by the host's launch calls,
    // of which this path makes two more: one 1080p stream 50 k -> 37 k fps), three B streams, a frame geometry the LDS
    // kernel takes
    // ... and only while frames go through the LDS kernel alone (kBlobSpec): a step in the full launch sequence -- the position
    // filter is on, or a frame was declined a moment ago -- takes the plain order (the switch drains the B streams)
    const bool early_wanted = c->early_blob < 0 ? n <= 3 : c->early_blob != 0;        // (r06a / r06e: 2 x 1080p 64.8 k -> 70.3 k fps, 3 x: 69.3 k -> 72-74 k, 4 x: 73.0 k -> 65-72 k)
    const bool early = early_wanted && !(lone && c->lone_plain) && !c->early_off && c->lds_spec && !c->kal_on && !j[0].ready && !share_b && c->nb >= 3 && !c->serial && !c->use_graph && !(c->expt & 1) &&
                       c->g.H > 2 && c->g.H <= 16383 && c->g.W <= 16383 && (size_t)n * (size_t)c->g.P >= c->early_min_px;
    // Threads a K1 workgroup (kernels_mog.hip, k_mog_fused): one wave a workgroup keeps every wave slot filled (K1 -3.5 % on
    // an everyday 4K model, -5.5 % on a dense one) and starves the back half's workgroups of slots.  Taken where that
    // does not come back as a lower frame rate: steps whose blob workgroup is already resident (early), and dense models
    // (streaming-load launches: K1 is 5-10 x the back half, 4K 6 670 -> 7 110 fps, profiles/r05n_dense_wg64_ab.txt; a result
    // is then ready ~230 us later, one K1 launch, because the blob workgroup gets in when the launch drains).
    const int k1_wg = c->k1_wg_force ? c->k1_wg_force : (early || c->nt_loads || (lone && c->lone_plain && early_wanted && (size_t)n * (size_t)c->g.P >= c->early_min_px)) ? 64 : 256;
    c->last_k1_wg = k1_wg;
    c->last_step_early = early;
    // ONE "K1 done" event for the step: both frames' back halves wait for the same launch (a second record would be
    // a second marker packet between two K1s on stream A)
    // A LONE frame (nothing else outstanding: a camera-paced caller) has its back half queued on stream A itself, right behind
    // its per-pixel kernel: no event, no stream wait, and none of the ~10 us a dependency across two hardware queues takes
    // (the gap between the two kernels in a trace of synchronous steps, profiles/r07_sync_step_timeline.txt).  Whatever is
    // launched later waits for ITS per-pixel kernel, which queues behind this back half on A: the scratch set is safe.
    const bool inline_back = lone && c->lone_plain && nj == 1 && !early && !c->use_graph && !c->serial && !(c->expt & 1);
    hipEvent_t k1_done = nullptr;
    if (!(c->expt & 1) && !inline_back) {
        const int nbe0 = (j[0].ready && c->nb > 2 && !c->use_graph) ? 2 : c->nb;
        k1_done = c->ev_k1[j[0].slot % nbe0];
    }
    bool k1_done_recorded = false;
    int s0 = 0;
    while (s0 < n) {
        int s1 = s0 + 1;
        while (s1 < n && same(s1, s0)) ++s1;
        const bool pair = nj == 2 && !rates[s0].fresh && !rates[(size_t)n + s0].fresh && !(c->audit_on && c->cfg.channels != 3);
        for (int i = 0; i < (pair ? 1 : nj); ++i) {
            MogLaunch a = mog_launch_base(c, (const uint8_t *)j[i].frames, rates[(size_t)i * n + s0]);
            a.thr_bits = thr_buf(c, j[i].slot);
            if (pair) {
                const Rate &r2 = rates[(size_t)n + s0];
                a.frames2 = (const uint8_t *)j[1].frames;
                a.thr_bits2 = thr_buf(c, j[1].slot);
                a.alphaT2 = r2.alphaT; a.alpha12 = r2.alpha1; a.prune2 = r2.prune;
            }
            const bool last = s1 == n && i + 1 == (pair ? 1 : nj);
            const bool ride_on = c->k1_stop_event < 0 ? (size_t)n * (size_t)c->g.P >= (size_t)4000000 : c->k1_stop_event != 0;
            const bool ride = last && ride_on && k1_done && !ps;
            launch_mog_fused(c->g, a, s0, s1 - s0, A, ride ? k1_done : nullptr, mog_launch_opts(c, s0, s1, k1_wg));
            k1_done_recorded = k1_done_recorded || ride;
        }
        s0 = s1;
    }
    HIPCHK(c, hipGetLastError());
    if (ps) HIPCHK(c, hipEventRecord(ps->e[1], A));
    for (int i = 0; i < nj; ++i) {
        // model density -> cache policy of the next launches: probes at frames 8, 16, 32, then every 64th; each
        // probe's numbers are read when the NEXT one is launched (it finished long ago: the ring is a few deep)
        const unsigned long long t = c->launched_total + (unsigned long long)i;
        if ((t >= 8 && t < 64 && (t & (t - 1)) == 0) || (t >= 64 && (t & 63) == 0)) {
            const int ds = (int)(c->dens_probes & 1);
            const unsigned *prev = c->dens_host + 2 * (ds ^ 1);
            if (c->dens_probes && prev[1]) c->nt_loads = 2ull * prev[0] >= 5ull * prev[1];      // mean live modes >= 2.5
            launch_density_probe(c->nmodes, (size_t)n * c->g.Palloc, c->dens_dev + 2 * ds, A);
            c->dens_probes++;
        }
    }
    c->launched_total += (unsigned long long)nj;

    if (k1_done && !k1_done_recorded) HIPCHK(c, hipEventRecord(k1_done, A));
    const int ero_cfg = c->cfg.erode > 1 ? c->cfg.erode : 0, dil_cfg = c->cfg.dilate > 1 ? c->cfg.dilate : 0;
    // The paired back half: both frames of a two-frame step go through ONE row-scan launch and ONE k_blob_lds launch (grid z =
    // frame) on ONE B stream behind one wait, one ring event covers both results -- 6 runtime calls a step instead of 10.
    // Small frames are bound by exactly those calls (one 1080p stream: 30 us of per-pixel kernel a step under ~40 us of
    // HIP calls).  Steps alternate between B0 / B1 with two scratch sets each, so consecutive steps' back halves overlap.
    // Speculative mode only (row scan + LDS kernel; a declined frame is repaired from its threshold bits in the repair set, below).
    const bool paired = nj == 2 && !early && c->pair_back && c->lds_spec && !c->kal_on && !c->serial && !c->use_graph && !(c->expt & 1) &&
                        c->nb >= 3 && c->g.H > 2 && c->g.H <= 16383 && c->g.W <= 16383 &&
                        !(ero_cfg && rowscan_lds_bytes(c->g, dil_cfg) > kRowscanLdsMax);
    const int path = early ? 1 : paired ? 2 : 0;
    if (c->last_early >= 0 && c->last_early != path)          // the paths use the scratch sets from different streams
        for (int q = 0; q < oatgpu_ctx::kNB; ++q) if (c->stream_b[q] && c->b_used[q]) HIPCHK(c, hipStreamSynchronize(c->stream_b[q]));
    c->last_early = path;
    // Scratch sets 3 and 4 (made with the context; the loop below allocates only in measurement builds that shrank nb -- and then
    // BEFORE anything of the step is recorded as launched would be too late anyway: those builds accept the stall): 3 for the
    // early and the paired layout; 4 = THE REPAIR SET -- every frame that
    // goes through the LDS kernel alone (kBlobSpec), on whichever path, is repaired in set 4 on B2 if it is declined: each layout
    // ties its sets to streams in its own way (plain: set q on stream q; paired: sets 2p, 2p + 1 on stream p; early: sets 0..3 on
    // B0 / B1 by parity), a switch of layouts drains the streams, but a repair comes LATER -- at the frame's collect -- and must
    // not write a set that the layout of the day uses from another stream.
    const bool spec_plain = !early && !paired && c->lds_spec && !c->kal_on && !c->use_graph && !(c->expt & 1) && c->nb >= 3;
    for (int q = 3; q < oatgpu_ctx::kSets; ++q) {
        if (c->have_set[q] || !(q == 3 ? (early || paired) : (early || paired || spec_plain))) continue;
        if (!alloc_scratch_set(c, q)) return fail(c, OATGPU_E_NOMEM, "device allocation failed: %s", hipGetErrorString(hipGetLastError()));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->have_set[q] = true;
    }
    if (paired) {
        const Geom &g = c->g;
        const int p = (int)(c->pair_steps++ & 1u);
        hipStream_t B = c->stream_b[p];
        c->b_used[p] = true;
        HIPCHK(c, hipStreamWaitEvent(B, k1_done, 0));
        if (ps) { HIPCHK(c, hipEventRecord(ps->e[2], B)); HIPCHK(c, hipEventRecord(ps->e[3], B)); }
        const BlobBuffers bbs[2] = {c->bb[2 * p], c->bb[2 * p + 1]};
        const u64 *srcs[2] = {thr_buf(c, j[0].slot), thr_buf(c, j[1].slot)};
        ResultRec *res[2] = {c->res_dev + (size_t)j[0].slot * n, c->res_dev + (size_t)j[1].slot * n};
        launch_blob_pair(g, bbs, srcs, ero_cfg, dil_cfg, c->cfg.min_area, c->cfg.max_area, res, n, B);
        HIPCHK(c, hipGetLastError());
        if (ps) HIPCHK(c, hipEventRecord(ps->e[4], B));
        for (int i = 0; i < 2; ++i) {
            const int slot = j[i].slot;
            c->slot_spec[slot] = 1;
            // A repair redoes the frame in scratch set 4 on B2, like the early order's -- NOT in the frame's own set on its step's
            // stream: this layout ties set 2p + i to stream p, the plain order ties set q to stream q, and a repair launched
            // after a switch to the plain order (the decline itself causes one) would write set 1 from B0 while a plain frame
            // uses it from B1 (found by tools/fuzz.py --seed 11, configuration 301: profiles/r07_fuzz_1500.txt).  Set 4 is
            // touched by repairs only, and only from B2.
            c->slot_q[slot] = 4;
            c->slot_st[slot] = 2;
            c->slot_filtered[slot] = 0;
            c->slot_ev[slot] = j[1].slot;                 // one ring event behind the step's blob launch covers both results
        }
        c->last_morph = (dil_cfg || ero_cfg) ? bbs[1].morph : srcs[1];
        c->last_fin = bbs[1].fin;
        c->last_q = j[1].slot;
        HIPCHK(c, hipEventRecord(c->ring_ev[j[1].slot], B));
        return OATGPU_OK;
    }
    if (early) {
        const Geom &g = c->g;
        hipStream_t C = c->stream_b[2];
        c->b_used[0] = c->b_used[1] = c->b_used[2] = true;
        BlobBuffers bbs[2];
        ResultRec *res[2];
        unsigned tk[2] = {0, 0};
        int qs[2] = {0, 0};
        for (int i = 0; i < nj; ++i) {
            // scratch set by FRAME index mod 4, row-scan stream by its parity (not by ring slot: with an odd ring depth two
            // consecutive frames can sit in slots of the same parity, and the two frames of a step must not share a set).
            // FOUR sets (r07; two until then): with two, a frame's row scan had to wait for the blob workgroup of the frame two
            // before it -- the back half was a serial chain of row scan (~60 us beside the per-pixel kernel) + blob analysis
            // (~30 us) per step, as long as the per-pixel kernel's own period: the row scan started 37 us behind its per-pixel
            // launch, a result was ready ~125 us behind it (profiles/r07f_timeline_rowscan_shapes.txt).  With four, a step's row
            // scans depend on nothing but their own per-pixel launch.
            const int slot = j[i].slot, q = (int)(c->early_frames++ & 3u);
            qs[i] = q;
            hipStream_t R = c->stream_b[q & 1];
            ProfStep *pb = i == 0 ? ps : nullptr;
            BlobBuffers &bb = c->bb[q];
            // ---- B0 / B1: the frame's row scan, behind the step's per-pixel kernel and the last reader of scratch set q ----
            HIPCHK(c, hipStreamWaitEvent(R, k1_done, 0));
            if (c->ev_blob_valid[q]) HIPCHK(c, hipStreamWaitEvent(R, c->ev_blob[q], 0));
            if (pb) HIPCHK(c, hipEventRecord(pb->e[2], R));
            const u64 *src = thr_buf(c, slot);
            const int dil = c->cfg.dilate > 1 ? c->cfg.dilate : 0;
            int ero = c->cfg.erode > 1 ? c->cfg.erode : 0;
            if (ero && rowscan_lds_bytes(g, dil) > kRowscanLdsMax) {     // very wide rows x large dilation
                launch_morph(g, src, bb.tmp, ero, true, 0, n, R);
                src = bb.tmp;
                ero = 0;
            }
            if (pb) HIPCHK(c, hipEventRecord(pb->e[3], R));
            c->last_morph = (dil || ero) ? bb.morph : src;
            c->last_fin = bb.fin;
            unsigned ticket = ++c->bh_ticket[q];
            if (!ticket) ticket = ++c->bh_ticket[q];
            launch_rowscan_signal(g, bb, src, ero, dil, 0, n, ticket, R);
            bbs[i] = bb; res[i] = c->res_dev + (size_t)slot * n; tk[i] = ticket;
            c->slot_spec[slot] = 1;
            c->slot_q[slot] = 4;                         // a repair redoes the frame in scratch set 4 ...
            c->slot_st[slot] = 2;                        // ... on B2, behind the blob launches
            c->slot_filtered[slot] = 0;
        }
        // ---- B2: the blob workgroups of the step's frames, ONE launch, dispatched now, started by their row scans' tickets ----
        launch_blob_tail2(g, bbs, c->cfg.min_area, c->cfg.max_area, res, n, tk, nj, C);
        HIPCHK(c, hipGetLastError());
        if (ps) HIPCHK(c, hipEventRecord(ps->e[4], C));
        for (int i = 0; i < nj; ++i) {
            const int slot = j[i].slot, q = qs[i];
            c->slot_ev[slot] = j[nj - 1].slot;           // one ring event behind the step's blob launch covers both results
            HIPCHK(c, hipEventRecord(c->ev_blob[q], C));
            c->ev_blob_valid[q] = true;
            c->last_q = slot;
        }
        HIPCHK(c, hipEventRecord(c->ring_ev[j[nj - 1].slot], C));
    }
    for (int i = 0; i < nj && !early; ++i) {
        const int slot = j[i].slot;
        // scratch set / B stream of this frame.  Host frames arrive over the copy stream: use one B stream
        // fewer then: the copy stream sits on the last B stream's hardware queue (see acquire_streams)
        const int nbe = (j[i].ready && c->nb > 2 && !c->use_graph) ? 2 : c->nb;
        const int q = (share_b ? j[0].slot : slot) % nbe;
        const int k = slot;                              // threshold-bit buffer of this frame
        hipStream_t B = (c->serial || inline_back) ? c->stream : c->stream_b[q];
        if (!inline_back) c->b_used[q] = true;
        ProfStep *pb = i == 0 ? ps : nullptr;           // the back half of the step's first frame is the sampled one
        if (c->expt & 1) {                               // K1 only: how fast can stream A go on its own?
            c->slot_ev[slot] = slot;
            HIPCHK(c, hipEventRecord(c->ring_ev[slot], A));
            continue;
        }
        // Stream B[q]: morphology + blob analysis of this frame.
        if (!(share_b && i == 1) && !inline_back) HIPCHK(c, hipStreamWaitEvent(B, k1_done, 0));
        if (c->use_graph && !c->back_graph[slot]) {
            c->back_graph[slot] = capture_back_half(c, slot, B);
            if (!c->back_graph[slot]) c->use_graph = false;          // capture unsupported: plain launches
        }
        if (c->use_graph && c->back_graph[slot]) {
            if (pb) { HIPCHK(c, hipEventRecord(pb->e[2], B)); HIPCHK(c, hipEventRecord(pb->e[3], B)); }
            HIPCHK(c, hipGraphLaunch(c->back_graph[slot], B));
            const int dil = c->cfg.dilate > 1 ? c->cfg.dilate : 0;
            const bool ero = c->cfg.erode > 1, fused = ero && rowscan_lds_bytes(c->g, dil) <= kRowscanLdsMax;
            c->last_morph = (dil || fused) ? c->bb[q].morph : (ero ? c->bb[q].tmp : thr_buf(c, k));
            c->last_fin = c->bb[q].fin;
        } else {
            if (pb) HIPCHK(c, hipEventRecord(pb->e[2], B));
            const int mode = (c->lds_spec && !c->kal_on) ? kBlobSpec : kBlobFull;   // (the position filter is sequential: no repairs behind it)
            c->slot_spec[slot] = mode == kBlobSpec;
            c->slot_q[slot] = (char)(spec_plain ? 4 : q);          // (a declined frame is redone in the repair set on B2, above)
            c->slot_st[slot] = (char)(spec_plain ? 2 : q);
            int rc = back_half(c, c->bb[q], thr_buf(c, k), 0, n, slot, B, pb ? pb->e[3] : nullptr, -1, -1, mode);
            if (rc) return rc;
        }
        if (c->kal_on) {
            KalmanLaunch kl = c->kal;
            kl.ticket = c->kal_ticket++;
            launch_kalman(kl, c->res_dev + (size_t)slot * n, n, B);
            HIPCHK(c, hipGetLastError());
        }
        c->slot_filtered[slot] = c->kal_on;
        if (pb) HIPCHK(c, hipEventRecord(pb->e[4], B));
        c->slot_ev[slot] = slot;
        if (share_b && i == 0) c->slot_ev[slot] = j[1].slot;       // the second frame's event, recorded below, covers it
        else HIPCHK(c, hipEventRecord(c->ring_ev[slot], B));
        c->last_q = k;
    }
    return OATGPU_OK;
}

// launch what oatgpu_track_enqueue[_dev] only registered
static int flush_pending(oatgpu_ctx *c)
{
    if (!c->pend_valid) return OATGPU_OK;
    c->pend_valid = false;
    int rc = hipSetDevice(c->cfg.device) == hipSuccess ? OATGPU_OK : fail(c, OATGPU_E_HIP, "hipSetDevice(%d) failed", c->cfg.device);
    if (!rc) rc = launch_jobs(c, &c->pend, 1, c->ring_count == 1);        // (lone: nothing else is outstanding)
    if (rc) {
        // the same rule as in enqueue_frames: mog_begin has advanced the frame counts and the rate schedule, the model may
        // have moved -- fatal for the context.  The registered frame is no longer outstanding: without this a later collect
        // would find its ring event as an earlier frame left it and hand out that frame's record as this one's.
        c->broken = true;
        c->enq_total--;
        c->ring_count--;
        c->err += " (the frame registered by the previous enqueue was dropped)";
    }
    return rc;
}

static int enqueue_frames(oatgpu_ctx *c, const void *frames_dev, double lr, hipEvent_t frames_ready)
{
    if (!c || !frames_dev) return fail(c, OATGPU_E_INVALID, "null argument");
    // (a set being staged owns the next ring slot: nothing else may be enqueued until oatgpu_track_enqueue_staged took it)
    if (c->staged_count) return fail(c, OATGPU_E_INVALID, "a frame set is being staged (oatgpu_track_stage): finish it with oatgpu_track_enqueue_staged");
    if (c->broken) return fail(c, OATGPU_E_HIP, "context unusable after a failed launch (destroy it): %s", c->err.c_str());
    // (a deferred detector's kernels may still be running on stream A with threshold slot 0 and scratch set 0: the pipelined
    // path would write both from its own streams)
    if (c->defer_kind) return fail(c, OATGPU_E_INVALID, "a deferred result is waiting: fetch it first (oatgpu_fetch_frame / oatgpu_fetch_position)");
    if (c->ring_count == c->cfg.ring_depth) return fail(c, OATGPU_E_RING_FULL, "result ring full: collect first");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    oatgpu_ctx::FrameJob cur;
    cur.frames = frames_dev; cur.lr = lr; cur.ready = frames_ready;
    cur.slot = (int)(c->enq_total % (unsigned long long)c->ring_slots);
    // two frames a launch: only where a second frame can be outstanding, and not under the measurement switches
    const bool want_pair = c->fuse == 2 || (c->fuse == 0 && (frames_ready != nullptr || c->in_sequence));
    const bool may_fuse = want_pair && c->cfg.ring_depth >= 2 && !c->use_graph && !c->serial && !c->expt && !(c->audit_on && c->cfg.channels != 3);
    int rc = OATGPU_OK;
    if (c->pend_valid) {
        const oatgpu_ctx::FrameJob two[2] = {c->pend, cur};
        c->pend_valid = false;
        rc = launch_jobs(c, two, 2);
        if (rc) {                         // the registered frame went down with this one: it is no longer outstanding
            c->enq_total--;
            c->ring_count--;
            c->err += " (the frame registered by the previous enqueue was dropped with this one)";
        }
    } else if (may_fuse) {
        c->pend = cur;
        c->pend_valid = true;
    } else {
        rc = launch_jobs(c, &cur, 1, c->ring_count == 0);
    }
    if (rc) {
        // launch_jobs had advanced the streams' frame counts and rate schedules (mog_begin) and may have updated the
        // model before it failed: results from here on would silently differ from the reference's.  Fatal for the
        // context (sticky): results already outstanding can still be collected, nothing new is accepted.
        c->broken = true;
        return rc;
    }
    c->enq_total++;
    c->ring_count++;
    return OATGPU_OK;
}

// Speculation and repair (see lds_spec).  A frame the single-workgroup LDS kernel declined comes back marked; the global
// kernels then redo it from its threshold bits, which stay in its ring slot until its result was collected.
static bool slot_needs_global(const oatgpu_ctx *c, int slot)
{
    if (!c->slot_spec[slot]) return false;
    const ResultRec *r = c->res_host + (size_t)slot * c->cfg.n_streams;
    for (int s = 0; s < c->cfg.n_streams; ++s) if (r[s].valid == kNeedsGlobal) return true;
    return false;
}
// ... and did its parked workgroup give up WAITING (path == kPathTimeout) rather than find the frame too busy?
static bool slot_timed_out(const oatgpu_ctx *c, int slot)
{
    const ResultRec *r = c->res_host + (size_t)slot * c->cfg.n_streams;
    for (int s = 0; s < c->cfg.n_streams; ++s) if (r[s].valid == kNeedsGlobal && r[s].path == kPathTimeout) return true;
    return false;
}

// ... on the frame's scratch set's stream, behind whatever later frame that is busy with; the slot's ring event is
// recorded again behind them.  Not waited for here.
static int launch_repair(oatgpu_ctx *c, int slot)
{
    if (slot_timed_out(c, slot)) {
        // The blob workgroup was resident and its row scan never ran beside it: a tool serialises kernel dispatches (a
        // counter-collecting profiler, a debug layer) or too many contexts share the hardware queues.  Correct either way
        // (the global kernels redo the frame), but every such step costs 100 ms: stop parking for this context and say so.
        // (counted once: the frames that were parked in the same instant -- the second frame of a two-frame step, the steps a deep
        // ring had launched before this one was collected -- belong to the same episode; they saw the device flag the first one
        // set (BlobBuffers::nopark) and declined without waiting, or gave up in the very same 100 ms)
        if (!c->early_off) {
            c->early_timeouts++;
            c->early_off = true;
            g_last_error = c->err = "early blob dispatch switched off for this context: a parked blob workgroup waited 100 ms for a row "
                                    "scan that was not dispatched beside it (kernel dispatches are being serialised); results are unaffected";
        }
    }
    const int q = c->slot_q[slot];
    hipStream_t B = c->serial ? c->stream : c->stream_b[(int)c->slot_st[slot]];
    c->b_used[(int)c->slot_st[slot]] = true;
    const int rc = back_half(c, c->bb[q], thr_buf(c, slot), 0, c->cfg.n_streams, slot, B, nullptr, -1, -1, kBlobGlobal);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(c->ring_ev[slot], B));
    c->slot_ev[slot] = slot;
    c->slot_repair[slot] = 1;
    c->slot_spec[slot] = 0;
    c->lds_spec = false;
    c->lds_streak = 0;
    return OATGPU_OK;
}

// quiesce(): every outstanding frame is done; redo the declined ones and wait for them
static int repair_outstanding(oatgpu_ctx *c)
{
    bool any = false;
    for (unsigned long long t = c->col_total; t < c->col_total + (unsigned long long)c->ring_count; ++t) {
        const int slot = (int)(t % (unsigned long long)c->ring_slots);
        if (c->slot_repair[slot] || !slot_needs_global(c, slot)) continue;
        const int rc = launch_repair(c, slot);
        if (rc) return rc;
        any = true;
    }
    if (any)
        for (int q = 0; q < oatgpu_ctx::kNB; ++q) if (c->stream_b[q] && c->b_used[q]) HIPCHK(c, hipStreamSynchronize(c->stream_b[q]));
    if (any && c->serial) HIPCHK(c, hipStreamSynchronize(c->stream));
    return OATGPU_OK;
}

// A copy of one frame takes ~0.1 ms; hipEventSynchronize puts the thread to sleep and its wake-up costs 30-50 us -- a
// third of the wait, once per camera and round.  Poll for up to 2 ms, then sleep.
static hipError_t wait_short(hipEvent_t e)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(e);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipEventSynchronize(e);
    }
}

extern "C" int oatgpu_track_collect(oatgpu_ctx *c, oatgpu_position *out)
{
    if (!c || !out) return fail(c, OATGPU_E_INVALID, "null argument");
    if (c->ring_count == 0) return fail(c, OATGPU_E_RING_EMPTY, "nothing outstanding");
    const int slot = (int)(c->col_total % (unsigned long long)c->ring_slots);
    if (c->pend_valid && c->pend.slot == slot) {          // the frame wanted is still only registered: launch it alone
        const int frc = flush_pending(c);
        if (frc) return frc;
    }
    // (polling here, as wait_short does for the copies, made the one-camera latency WORSE: p50 207 -> 257 us free-running)
    HIPCHK(c, hipEventSynchronize(c->ring_ev[c->slot_ev[slot]]));
    const ResultRec *r = c->res_host + (size_t)slot * c->cfg.n_streams;
    if (!c->slot_repair[slot] && slot_needs_global(c, slot)) {
        const int rc = launch_repair(c, slot);
        if (rc) return rc;
    }
    if (c->slot_repair[slot]) {                            // (launched here, by oatgpu_track_ready or by a quiesce())
        HIPCHK(c, hipEventSynchronize(c->ring_ev[slot]));
        c->slot_repair[slot] = 0;
    } else {
        constexpr int kSpecAfter = 16;
        bool all_lds = true;
        for (int s = 0; s < c->cfg.n_streams; ++s) if (r[s].path != 1) all_lds = false;
        if (all_lds) {
            if (++c->lds_streak >= kSpecAfter) c->lds_spec = true;
        } else {
            c->lds_streak = 0;
            c->lds_spec = false;
        }
    }
    for (int s = 0; s < c->cfg.n_streams; ++s) {
        to_position(r[s], &out[s]);
        if (c->slot_filtered[slot]) apply_kalman(r[s], &out[s]);
        if (c->homo_on) apply_homography(c, &out[s]);
    }
    c->col_total++;
    c->ring_count--;
    return OATGPU_OK;
}

extern "C" int oatgpu_track_input_consumed(oatgpu_ctx *c)
{
    if (!c) return OATGPU_E_INVALID;
    if (c->dev_unconsumed) {             // device frames: the per-pixel kernels that read them must have finished
        HIPCHK(c, hipSetDevice(c->cfg.device));
        const int frc = flush_pending(c);
        if (frc) return frc;
        if (!c->ev_in) HIPCHK(c, hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        HIPCHK(c, hipEventRecord(c->ev_in, c->stream));
        HIPCHK(c, hipEventSynchronize(c->ev_in));
        c->dev_unconsumed = false;
    }
    if (c->last_copy_slot < 0) return OATGPU_OK;
    HIPCHK(c, hipEventSynchronize(c->copy_ev[c->last_copy_slot]));
    return OATGPU_OK;
}

extern "C" int oatgpu_track_input_consumed_stream(oatgpu_ctx *c, int32_t stream_ix)
{
    if (!c) return OATGPU_E_INVALID;
    if (stream_ix < 0 || stream_ix >= c->cfg.n_streams) return fail(c, OATGPU_E_INVALID, "stream index %d out of range", stream_ix);
    if (c->staged_count) {                                // a set being staged: that camera's own copy
        if (!c->staged[(size_t)stream_ix]) return fail(c, OATGPU_E_INVALID, "stream %d is not staged", stream_ix);
        HIPCHK(c, wait_short(c->copy_ev_s[(size_t)c->stage_slot * c->cfg.n_streams + stream_ix]));
        return OATGPU_OK;
    }
    c->per_stream_copy_ev = true;                         // from the next enqueue on
    // device frames, a set enqueued before the switch, the last stream of a set: the whole set's rule
    if (c->dev_unconsumed || c->last_copy_slot < 0 || !c->last_copy_per_stream || stream_ix + 1 == c->cfg.n_streams)
        return oatgpu_track_input_consumed(c);
    HIPCHK(c, wait_short(c->copy_ev_s[(size_t)c->last_copy_slot * c->cfg.n_streams + stream_ix]));
    return OATGPU_OK;
}

extern "C" int oatgpu_track_ready(oatgpu_ctx *c)
{
    if (!c) return OATGPU_E_INVALID;
    if (c->ring_count == 0) return 0;
    const int slot = (int)(c->col_total % (unsigned long long)c->ring_slots);
    if (c->pend_valid && c->pend.slot == slot) {          // somebody is waiting for this very frame: send it off
        const int frc = flush_pending(c);
        if (frc) return frc;
    }
    const hipError_t e = hipEventQuery(c->ring_ev[c->slot_repair[slot] ? slot : c->slot_ev[slot]]);
    if (e == hipErrorNotReady) return 0;
    if (e != hipSuccess) return fail(c, OATGPU_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
    if (!c->slot_repair[slot] && slot_needs_global(c, slot)) {   // declined by the LDS kernel: collect would block on the
        const int rc = launch_repair(c, slot);                    // global kernels -- start them, not ready yet
        return rc ? rc : 0;
    }
    return 1;
}

static int sequence_dev(oatgpu_ctx *c, const void *const *frames_dev, int32_t n_frames, double lr, oatgpu_position *out,
                        double *done_s, double *enq_s)
{
    if (!c || !frames_dev || !out || n_frames < 0) return fail(c, OATGPU_E_INVALID, "null argument");
    if (c->ring_count) return fail(c, OATGPU_E_INVALID, "track_sequence while enqueued results are outstanding");
    const int n = c->cfg.n_streams;
    const auto t0 = std::chrono::steady_clock::now();
    int got = 0, rc = OATGPU_OK;
    auto collect_one = [&]() {
        const int r = oatgpu_track_collect(c, out + (size_t)got * n);
        if (!r && done_s) done_s[got] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        ++got;
        return r;
    };
    c->in_sequence = true;               // every frame of the sequence is in hand: pairing is safe (oatgpu_set_fusion)
    for (int t = 0; t < n_frames && !rc; ++t) {
        if (c->ring_count == c->cfg.ring_depth) rc = collect_one();
        if (!rc && enq_s) enq_s[t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!rc) rc = oatgpu_track_enqueue_dev(c, frames_dev[t], lr);
    }
    c->in_sequence = false;
    while (!rc && c->ring_count) rc = collect_one();
    c->dev_unconsumed = false;           // all collected: every frame was read
    return rc;
}

extern "C" int oatgpu_track_sequence_dev_timed(oatgpu_ctx *c, const void *const *frames_dev, int32_t n_frames, double lr,
                                               oatgpu_position *out, double *done_s)
{
    return sequence_dev(c, frames_dev, n_frames, lr, out, done_s, nullptr);
}

extern "C" int oatgpu_track_sequence_dev_latency(oatgpu_ctx *c, const void *const *frames_dev, int32_t n_frames, double lr,
                                                 oatgpu_position *out, double *done_s, double *enq_s)
{
    return sequence_dev(c, frames_dev, n_frames, lr, out, done_s, enq_s);
}

extern "C" int oatgpu_track_sequence_dev(oatgpu_ctx *c, const void *const *frames_dev, int32_t n_frames, double lr,
                                         oatgpu_position *out)
{
    return sequence_dev(c, frames_dev, n_frames, lr, out, nullptr, nullptr);
}

extern "C" int oatgpu_track_outstanding(const oatgpu_ctx *c) { return c ? c->ring_count : 0; }

extern "C" int oatgpu_track_batch_dev(oatgpu_ctx *c, const void *frames_dev, double lr, oatgpu_position *out)
{
    if (!c) return OATGPU_E_INVALID;
    if (c->ring_count) return fail(c, OATGPU_E_INVALID, "track_batch while enqueued results are outstanding");
    int rc = oatgpu_track_enqueue_dev(c, frames_dev, lr);
    if (rc) return rc;
    return oatgpu_track_collect(c, out);
}

extern "C" int oatgpu_track_batch(oatgpu_ctx *c, const uint8_t *const *frames_host, int32_t n, double lr,
                                  oatgpu_position *out)
{
    if (!c || !frames_host || !out) return fail(c, OATGPU_E_INVALID, "null argument");
    if (n != c->cfg.n_streams) return fail(c, OATGPU_E_INVALID, "expected %d frames, got %d", c->cfg.n_streams, n);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const size_t fb = (size_t)c->g.H * c->g.W * c->cfg.channels;
    for (int s = 0; s < n; ++s) {
        if (!frames_host[s]) return fail(c, OATGPU_E_INVALID, "null frame %d", s);
        HIPCHK(c, hipMemcpyAsync(c->frames + (size_t)s * fb, frames_host[s], fb, hipMemcpyHostToDevice, c->stream));
    }
    return oatgpu_track_batch_dev(c, c->frames, lr, out);
}

// ------------------------------------------------------------------ taps ----

extern "C" int oatgpu_read_mask(oatgpu_ctx *c, int32_t s, int32_t which, uint8_t *out)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!out) return fail(c, OATGPU_E_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const u64 *last_thr = thr_buf(c, c->last_q);
    const u64 *base = which == OATGPU_TAP_THRESHOLD ? last_thr
                    : which == OATGPU_TAP_MORPH ? c->last_morph
                    : which == OATGPU_TAP_FINAL ? c->last_fin : nullptr;
    if (!base) return fail(c, OATGPU_E_INVALID, "unknown tap %d", which);
    const size_t npx = (size_t)c->g.H * c->g.W;
    launch_unpack_bits(c->g, base + (size_t)s * (c->g.Palloc >> 6), c->aux_b, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->aux_b, npx, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OATGPU_OK;
}

extern "C" int oatgpu_mog_get_state(oatgpu_ctx *c, int32_t s, uint8_t *modes_used, float *weight, float *variance,
                                    float *mean, int32_t *nframes)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const Geom &g = c->g;
    const size_t npx = (size_t)g.H * g.W, k = c->cfg.nmixtures, mb = 4 * (size_t)c->cfg.channels;
    DevBuf b_mu, b_w, b_v, b_m;          // freed on every return path
    HIPCHK(c, b_mu.alloc(npx));
    HIPCHK(c, b_w.alloc(npx * k * 4));
    HIPCHK(c, b_v.alloc(npx * k * 4));
    HIPCHK(c, b_m.alloc(npx * k * mb));
    uint8_t *d_mu = (uint8_t *)b_mu.p; float *d_w = (float *)b_w.p, *d_v = (float *)b_v.p, *d_m = (float *)b_m.p;
    launch_state_export(g, c->state + (size_t)s * mog_stream_floats(g.Palloc), c->nmodes + (size_t)s * g.Palloc, (int)k,
                        c->cfg.channels, d_mu, d_w, d_v, d_m, c->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && modes_used) e = hipMemcpyAsync(modes_used, d_mu, npx, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && weight) e = hipMemcpyAsync(weight, d_w, npx * k * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && variance) e = hipMemcpyAsync(variance, d_v, npx * k * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && mean) e = hipMemcpyAsync(mean, d_m, npx * k * mb, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, OATGPU_E_HIP, "state export failed: %s", hipGetErrorString(e));
    if (nframes) *nframes = c->nframes[s];
    return OATGPU_OK;
}

extern "C" int oatgpu_mog_set_state(oatgpu_ctx *c, int32_t s, const uint8_t *modes_used, const float *weight,
                                    const float *variance, const float *mean, int32_t nframes)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!modes_used || !weight || !variance || !mean || nframes < 0)
        return fail(c, OATGPU_E_INVALID, "null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    rc = quiesce(c);
    if (rc) return rc;
    const Geom &g = c->g;
    const size_t npx = (size_t)g.H * g.W, k = c->cfg.nmixtures, mb = 4 * (size_t)c->cfg.channels;
    DevBuf b_mu, b_w, b_v, b_m;          // freed on every return path
    HIPCHK(c, b_mu.alloc(npx));
    HIPCHK(c, b_w.alloc(npx * k * 4));
    HIPCHK(c, b_v.alloc(npx * k * 4));
    HIPCHK(c, b_m.alloc(npx * k * mb));
    uint8_t *d_mu = (uint8_t *)b_mu.p; float *d_w = (float *)b_w.p, *d_v = (float *)b_v.p, *d_m = (float *)b_m.p;
    hipError_t e = hipMemcpyAsync(d_mu, modes_used, npx, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_w, weight, npx * k * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_v, variance, npx * k * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_m, mean, npx * k * mb, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_state_import(g, c->state + (size_t)s * mog_stream_floats(g.Palloc), c->nmodes + (size_t)s * g.Palloc, (int)k,
                            c->cfg.channels, d_mu, d_w, d_v, d_m, c->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, OATGPU_E_HIP, "state import failed: %s", hipGetErrorString(e));
    c->nframes[s] = nframes;
    // The product kernels rely on what a run of the kernel leaves in a model (kernels_mog.hip): a weight that is 0 or in
    // [2^-62, 4] (div_inrange), means that are finite, below 2^20 and not -0.f and variances inside [var_min, var_max] (at
    // rate 0 the update of a fitted mode is then the identity and is not computed).  A model holding anything else
    // (hand-made, NaN, a weight of 1e-30, a variance outside the clamp) is taken as it is and advanced by the
    // instantiations that keep the compiler's division and compute every update.
    bool wild = false;
    const size_t chn = (size_t)c->cfg.channels;
    for (size_t p = 0; p < npx && !wild; ++p) {
        const size_t used = modes_used[p] < k ? modes_used[p] : k;
        for (size_t m = 0; m < used && !wild; ++m) {
            const float w = weight[p * k + m], v = variance[p * k + m];
            if (!(w == 0.f || (w >= 0x1p-62f && w <= 4.f))) wild = true;
            if (!(v >= c->cfg.var_min && v <= c->cfg.var_max)) wild = true;
            for (size_t ch = 0; ch < chn; ++ch) {
                const float mu = mean[(p * k + m) * chn + ch];
                uint32_t bits;
                memcpy(&bits, &mu, sizeof bits);
                if (!(mu > -0x1p20f && mu < 0x1p20f) || bits == 0x80000000u) wild = true;
            }
        }
    }
    c->wild_model[(size_t)s] = wild ? 1 : 0;
    return OATGPU_OK;
}

// ------------------------------------------------------- model checkpoint ----
// File: 64-byte header, then modes_used u8[npx], weight f32[npx*k], variance f32[npx*k],
// mean f32[npx*k*channels] in the logical layout of oatgpu_mog_get_state; entries of unused modes
// are written as zeros so that equal models give equal files.
namespace {
struct CkptHeader {
    char magic[8];          // "OATMOG2\0"
    uint32_t version;       // 1
    uint32_t rows, cols, channels, nmixtures;
    int32_t nframes;
    uint64_t payload_bytes;
    uint8_t reserved[24];
};
static_assert(sizeof(CkptHeader) == 64, "checkpoint header layout");
const char kCkptMagic[8] = {'O', 'A', 'T', 'M', 'O', 'G', '2', 0};
}  // namespace

extern "C" int oatgpu_mog_save(oatgpu_ctx *c, int32_t s, const char *path)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!path || !*path) return fail(c, OATGPU_E_INVALID, "null path");
    const size_t npx = (size_t)c->g.H * c->g.W, k = c->cfg.nmixtures, ch = c->cfg.channels;
    std::vector<uint8_t> mu(npx);
    std::vector<float> w(npx * k), v(npx * k), m(npx * k * ch);
    int32_t nf = 0;
    rc = oatgpu_mog_get_state(c, s, mu.data(), w.data(), v.data(), m.data(), &nf);
    if (rc) return rc;
    for (size_t p = 0; p < npx; ++p)
        for (size_t j = mu[p]; j < k; ++j) {
            w[p * k + j] = 0.f; v[p * k + j] = 0.f;
            for (size_t q = 0; q < ch; ++q) m[(p * k + j) * ch + q] = 0.f;
        }
    CkptHeader h{};
    memcpy(h.magic, kCkptMagic, 8);
    h.version = 1; h.rows = c->g.H; h.cols = c->g.W; h.channels = (uint32_t)ch; h.nmixtures = (uint32_t)k;
    h.nframes = nf;
    h.payload_bytes = npx + (w.size() + v.size() + m.size()) * sizeof(float);
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return fail(c, OATGPU_E_INVALID, "cannot open '%s' for writing", tmp.c_str());
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(mu.data(), 1, npx, f) == npx &&
              fwrite(w.data(), 4, w.size(), f) == w.size() && fwrite(v.data(), 4, v.size(), f) == v.size() &&
              fwrite(m.data(), 4, m.size(), f) == m.size();
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path) == 0;      // readers never see a half-written file
    if (!ok) { remove(tmp.c_str()); return fail(c, OATGPU_E_INVALID, "writing '%s' failed", path); }
    return OATGPU_OK;
}

extern "C" int oatgpu_mog_load(oatgpu_ctx *c, int32_t s, const char *path)
{
    int rc = check_stream_ix(c, s);
    if (rc) return rc;
    if (!path || !*path) return fail(c, OATGPU_E_INVALID, "null path");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(c, OATGPU_E_INVALID, "cannot open '%s'", path);
    CkptHeader h{};
    const size_t npx = (size_t)c->g.H * c->g.W, k = c->cfg.nmixtures, ch = c->cfg.channels;
    if (fread(&h, sizeof h, 1, f) != 1 || memcmp(h.magic, kCkptMagic, 8) != 0 || h.version != 1) {
        fclose(f);
        return fail(c, OATGPU_E_INVALID, "'%s' is not a MOG2 model checkpoint", path);
    }
    if (h.rows != (uint32_t)c->g.H || h.cols != (uint32_t)c->g.W || h.channels != ch || h.nmixtures != k ||
        h.nframes < 0 || h.payload_bytes != npx + (2 * npx * k + npx * k * ch) * sizeof(float)) {
        fclose(f);
        return fail(c, OATGPU_E_INVALID, "checkpoint '%s' is %ux%ux%u with %u mixtures; this context is %dx%dx%d with %d",
                    path, h.rows, h.cols, h.channels, h.nmixtures, c->g.H, c->g.W, (int)ch, (int)k);
    }
    std::vector<uint8_t> mu(npx);
    std::vector<float> w(npx * k), v(npx * k), m(npx * k * ch);
    const bool ok = fread(mu.data(), 1, npx, f) == npx && fread(w.data(), 4, w.size(), f) == w.size() &&
                    fread(v.data(), 4, v.size(), f) == v.size() && fread(m.data(), 4, m.size(), f) == m.size();
    fclose(f);
    if (!ok) return fail(c, OATGPU_E_INVALID, "checkpoint '%s' is truncated", path);
    for (size_t p = 0; p < npx; ++p)
        if (mu[p] > k) return fail(c, OATGPU_E_INVALID, "checkpoint '%s' is corrupt (mode count %u)", path, mu[p]);
    return oatgpu_mog_set_state(c, s, mu.data(), w.data(), v.data(), m.data(), h.nframes);
}

// ---------------------------------------------------------- measurement ----

extern "C" int oatgpu_measure_hbm(oatgpu_ctx *c, size_t bytes, int32_t reps, double *read_gbps, double *copy_gbps)
{
    if (!c || bytes < (1u << 20) || reps < 1) return fail(c, OATGPU_E_INVALID, "bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = quiesce(c);
    if (rc) return rc;
    const size_t n16 = bytes / 16;
    void *a = nullptr, *b = nullptr;
    unsigned *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc(&a, n16 * 16);
    if (e == hipSuccess) e = hipMalloc(&b, n16 * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&sink, 4);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0x5a, n16 * 16, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, n16 * 16, c->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    double best_r = 0, best_c = 0;
    for (int i = 0; i < reps + 1 && e == hipSuccess; ++i) {          // first iteration warms up
        float ms = 0;
        hipEventRecord(e0, c->stream);
        launch_stream_read(a, n16, sink, c->stream);
        hipEventRecord(e1, c->stream);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && i && ms > 0)
            best_r = std::max(best_r, (double)(n16 * 16) / (ms * 1e-3) / 1e9);
        hipEventRecord(e0, c->stream);
        launch_stream_copy(a, b, n16, c->stream);
        hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && i && ms > 0)
            best_c = std::max(best_c, (double)(2 * n16 * 16) / (ms * 1e-3) / 1e9);
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(a); hipFree(b); hipFree(sink);
    if (e != hipSuccess) return fail(c, OATGPU_E_HIP, "bandwidth probe failed: %s", hipGetErrorString(e));
    if (read_gbps) *read_gbps = best_r;
    if (copy_gbps) *copy_gbps = best_c;
    return OATGPU_OK;
}

extern "C" int oatgpu_traffic_audit(oatgpu_ctx *c, int32_t on)
{
    if (!c) return OATGPU_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = quiesce(c);
    if (rc) return rc;
    if (on) {
        if (!c->audit_dev) HIPCHK(c, hipMalloc((void **)&c->audit_dev, 8 * sizeof(unsigned long long)));
        HIPCHK(c, hipMemsetAsync(c->audit_dev, 0, 8 * sizeof(unsigned long long), c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->audit_launches = 0;
    }
    c->audit_on = on != 0;
    return OATGPU_OK;
}
extern "C" int oatgpu_traffic_read(oatgpu_ctx *c, oatgpu_traffic *out)
{
    if (!c || !out) return OATGPU_E_INVALID;
    memset(out, 0, sizeof *out);
    if (!c->audit_dev) return OATGPU_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = quiesce(c);
    if (rc) return rc;
    unsigned long long h[8];
    HIPCHK(c, hipMemcpy(h, c->audit_dev, sizeof h, hipMemcpyDeviceToHost));
    out->launches = c->audit_launches;
    out->pixels = (int64_t)h[0];
    out->lane_bytes_read = (int64_t)h[1]; out->lane_bytes_written = (int64_t)h[2];
    out->sector32_bytes_read = (int64_t)h[3]; out->sector32_bytes_written = (int64_t)h[4];
    out->sector64_bytes_read = (int64_t)h[5]; out->sector64_bytes_written = (int64_t)h[6];
    return OATGPU_OK;
}

extern "C" int oatgpu_profile_enable(oatgpu_ctx *c, int32_t on)
{
    if (!c) return OATGPU_E_INVALID;
    { const int frc = flush_pending(c); if (frc) return frc; }
    if (!on) prof_fold(c);
    if (on && !c->prof) {
        // calibrate: what an event pair measures on stream A around a launch that does nothing
        // (event processing + dispatch latency; the empty wave itself runs ~1 us)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipEvent_t e0, e1;
        HIPCHK(c, hipEventCreateWithFlags(&e0, hipEventDisableSystemFence));
        HIPCHK(c, hipEventCreateWithFlags(&e1, hipEventDisableSystemFence));
        float best = 1e30f;
        for (int i = 0; i < 16; ++i) {
            HIPCHK(c, hipEventRecord(e0, c->stream));
            launch_nop(c->stream);
            HIPCHK(c, hipEventRecord(e1, c->stream));
            HIPCHK(c, hipEventSynchronize(e1));
            float ms = 0;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best) best = ms;
        }
        hipEventDestroy(e0); hipEventDestroy(e1);
        c->event_pair_ms = best < 1e29f ? best : 0.0;
    }
    c->prof = on != 0;
    c->prof_every = on > 1 ? on : 1;
    c->prof_tick = 0;
    return OATGPU_OK;
}
extern "C" int oatgpu_profile_read(oatgpu_ctx *c, oatgpu_profile *out)
{
    if (!c || !out) return OATGPU_E_INVALID;
    { const int frc = flush_pending(c); if (frc) return frc; }
    prof_fold(c);
    *out = c->prof_sum;
    out->event_pair_ms = c->event_pair_ms;
    return OATGPU_OK;
}
extern "C" int oatgpu_profile_reset(oatgpu_ctx *c)
{
    if (!c) return OATGPU_E_INVALID;
    { const int frc = flush_pending(c); if (frc) return frc; }
    prof_fold(c);
    c->prof_sum = oatgpu_profile{};
    c->prof_drop_streak = 0;
    return OATGPU_OK;
}
