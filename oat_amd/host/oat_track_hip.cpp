// oat-track-hip SOURCE[,SOURCE...] SINK[,SINK...] [CONFIGURATION]
//
// The whole chain  framefilt mog -> framefilt col -C HSV -> posidet hsv  (optionally -> posifilt kalman)
// for N camera streams in ONE process, ONE device context and ONE fused launch per stage and step:
// N BGR oat::Frame SOURCEs in, N oat::Position2D SINKs out -- SOURCE i feeds SINK i, one token out per
// token in, in order, carrying its frame's Sample (PositionDetector.cpp:80).  N = 1 is the drop-in for a
// single pipeline; N > 1 is the reference's multi-camera shape (examples/two-gige/two-gige.sh:7-8: one
// component instance per camera) batched into one launch, which is what BASELINE configs 3 and 4 ask for.
//
// The loop is the reference's (FrameFilter.cpp:59-98, PositionDetector.cpp:58-99) with its three phases
// pipelined through the library's result ring instead of run back to back:
//
//   wait on every SOURCE                              PositionDetector.cpp:63-75
//   oatgpu_track_enqueue(frames in shared memory)     the H2D copies read the (page-locked) shm frames
//   oatgpu_track_input_consumed_stream(i), post SOURCE i   the reference posts right after its memcpy (:78-86);
//                                                     here, camera by camera, right after the DMA out of its segment
//   collect + publish finished results                :88-96, SINK i: wait, *shared = position, post
//
// A result is published as soon as it is ready whenever the camera the loop is waiting for has no frame yet (minimum
// latency, camera-bound pipelines); with frames waiting the loop goes on staging and finished results leave SINK by
// SINK between the copies (the link never idles for a consumer's hand-shake), at the latest when the ring is full.  Options are the union of the three stock components'
// (-a is mog's adaptation coefficient; the detector's area is --area).
//
// --gpu-index D0,D1,...  shards the cameras over several devices from THIS process (BASELINE configs 3/4 at the
// drop-in boundary: 64 x 1080p as 8 per GPU, 8 x 4K as one per GPU): the SOURCE list is cut into
// contiguous blocks -- camera s goes to shard s / ceil(S/N), as oat_amd/dist.py partitions streams over ranks --
// and every shard is a batched tracker of its own (own device context, own thread, own loop): streams are
// independent, so nothing crosses between shards.  The same device may be named twice (two contexts on one GPU).
//
// --ingest-root D0 (with --gpu-index D0,D1,...): every camera's frame is ingested on device D0 and travels to the device that
// owns its stream over RCCL send/recv (scatter_tracker.hpp; north_star's "RCCL over xGMI only for the trivial stream-to-rank
// scatter") -- one process, one thread, the same partition.  Without it every shard ingests its own cameras (the realistic
// camera topology).
//
// --thresh [lo,hi] selects the GREY chain instead:  framefilt mog -> posidet thresh  on SOURCEs that carry GREY
// frames (a mono camera or `framefilt col -C GREY`; SimpleThreshold.cpp:46 requires them), the one-channel model
// and the intensity window of SimpleThreshold.cpp:171-174 in the same fused launches.
#include "component.hpp"
#include "scatter_tracker.hpp"
#include <chrono>
#include <deque>
#include <fstream>
#include <sched.h>
#include <thread>
#include <unistd.h>

using namespace oat;

static std::vector<std::string> split_list(const std::string &s)
{
    std::vector<std::string> out;
    size_t a = 0;
    while (a <= s.size()) {
        const size_t b = s.find(',', a);
        const std::string t = s.substr(a, b == std::string::npos ? std::string::npos : b - a);
        if (t.empty()) throw std::runtime_error("empty address in list '" + s + "'");
        out.push_back(t);
        if (b == std::string::npos) break;
        a = b + 1;
    }
    return out;
}

// Keep the calling thread on the CPUs of the NUMA node device `dev` hangs off (oatgpu_device_numa_node): its launch calls,
// its reads of the shared-memory frames and the staging copies stay on that socket.  Best effort; false if nothing was done.
static bool pin_thread_to_device_node(int dev)
{
    const int node = oatgpu_device_numa_node(dev);
    if (node < 0) return false;
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string list;
    if (!f || !std::getline(f, list)) return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    size_t a = 0;
    int n = 0;
    while (a < list.size()) {
        size_t b = list.find(',', a);
        if (b == std::string::npos) b = list.size();
        const std::string part = list.substr(a, b - a);
        const size_t dash = part.find('-');
        const int lo = atoi(part.c_str()), hi = dash == std::string::npos ? lo : atoi(part.c_str() + dash + 1);
        for (int c = lo; c <= hi && c < CPU_SETSIZE; ++c) { CPU_SET(c, &set); ++n; }
        a = b + 1;
    }
    return n > 0 && sched_setaffinity(0, sizeof set, &set) == 0;
}

class BatchedTracker : public Component {
public:
    // stream_base / n_total: where this shard's cameras sit in the command line's SOURCE list (model file names)
    BatchedTracker(const std::vector<std::string> &sources, const std::vector<std::string> &sinks, int stream_base = 0,
                   int n_total = -1)
        : source_addresses_(sources), sink_addresses_(sinks), n_((int)sources.size()), stream_base_(stream_base),
          n_total_(n_total < 0 ? (int)sources.size() : n_total)
    {
        if (sources.size() != sinks.size()) throw std::runtime_error("need as many SINKs as SOURCEs");
        oatgpu_default_config(&cfg_);
        name_ = "track[" + sources[0] + (n_ > 1 ? ",..(" + std::to_string(n_) + ")" : "") + "->" + sinks[0] + (n_ > 1 ? ",.." : "") + "]";
        frame_sources_ = std::vector<Source<Frame>>(n_);
        position_sinks_ = std::vector<Sink<Position2D>>(n_);
        src_pins_ = std::vector<ShmRegistration>(n_);
    }
    std::string name() const override { return name_; }
    oatgpu_config cfg_;
    double learning_coeff_{0.0};
    bool kalman_{false};            // --kalman: `posifilt kalman` fused behind the detector
    double dt_{0.02}, timeout_{0.0}, sig_accel_{5.0}, sig_noise_{0.0};   // KalmanFilter2D.h:56-60
    std::string model_file_;        // --model-file: resume the MOG2 model(s) from / checkpoint to this file
    std::string mask_file_;         // --mask: `framefilt mask` fused in front of mog (FrameMasker.cpp:45-75)
    bool grey_{false};              // --thresh: GREY frames, mog -> posidet thresh
    int stage_copy_{0};             // --stage-copy kernel: oatgpu_set_stage_copy(1)
    bool timing_{false};            // --timing: where the loop's wall clock goes, printed at exit (stderr)
    bool homography_on_{false};     // --homography: `posifilt homography` behind the detector / the position filter
    double homography_[9]{1, 0, 0, 0, 1, 0, 0, 0, 1};
    ~BatchedTracker() override
    {
        if (!model_file_.empty() && gpu_.ctx)
            for (int s = 0; s < n_; ++s)
                if (oatgpu_mog_save(gpu_.ctx, s, model_path(s).c_str()) != OATGPU_OK)
                    std::cerr << name() << ": " << oatgpu_last_error(gpu_.ctx) << std::endl;
    }

    // --timing: seconds spent waiting for SOURCEs, in oatgpu_track_stage, waiting for a camera's copy, posting SOURCEs,
    // registering the set, and collecting + publishing results; rounds counted
    double t_wait_{0}, t_stage_{0}, t_consumed_{0}, t_post_{0}, t_enqueue_{0}, t_publish_{0}, t_r16_{0}, t_last_{0};
    unsigned long long rounds_{0};
    void print_timing() const
    {
        if (!timing_ || !rounds_) return;
        const double r = 1e6 / (double)rounds_;
        // (the rate between the end of round 16 and the end of the last round: process start-up, the first frames' page
        // registration and the models' initialisation are the harness's, not the loop's)
        const double steady = rounds_ > 16 && t_last_ > t_r16_ ? (double)(rounds_ - 16) * n_ / (t_last_ - t_r16_) : 0.0;
        std::fprintf(stderr, "%s: %llu rounds x %d cameras; per round (us): source wait %.1f, stage calls %.1f, copy wait %.1f, "
                             "source post %.1f, enqueue %.1f, collect+publish %.1f; steady %.1f fps aggregate\n", name().c_str(),
                     rounds_, n_, t_wait_ * r, t_stage_ * r, t_consumed_ * r, t_post_ * r, t_enqueue_ * r, t_publish_ * r, steady);
    }

protected:
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    std::string model_path(int s) const { return n_total_ == 1 ? model_file_ : model_file_ + "." + std::to_string(stream_base_ + s); }

    // PositionDetector.cpp:40-56, for every stream
    bool connectToNode() override
    {
        // A tracker that fails while connecting (a camera with another geometry, no device memory, an unreadable mask
        // or model file) still BINDS its SINKs on the way out, so that its destructor sets them END (Sink.h:73-91) and
        // the consumers of this shard's cameras end instead of waiting for a sink that will never appear -- with
        // several shards in one process the others keep running, where the reference's one-camera process would
        // simply have exited (framefilter/main.cpp:278-295).
        try {
            return connect_all();
        } catch (...) {
            for (int s = 0; s < n_; ++s) {
                if ((size_t)s < shared_positions_.size()) continue;        // bound before the failure
                try { position_sinks_[s].bind(sink_addresses_[s], sink_addresses_[s]); } catch (...) {}
            }
            throw;
        }
    }

    bool connect_all()
    {
        for (int s = 0; s < n_; ++s) frame_sources_[s].touch(source_addresses_[s]);
        FrameParams p0{};
        for (int s = 0; s < n_; ++s) {
            if (frame_sources_[s].connect(grey_ ? PIX_GREY : PIX_BGR) != SourceState::CONNECTED) return false;
            const FrameParams p = frame_sources_[s].parameters();
            if (s == 0) p0 = p;
            else if (p.rows != p0.rows || p.cols != p0.cols)
                throw std::runtime_error("all SOURCEs of one batched tracker must have the same frame geometry");
        }
        cfg_.rows = (int)p0.rows; cfg_.cols = (int)p0.cols; cfg_.n_streams = n_;
        cfg_.channels = grey_ ? 1 : 3;
        if (cfg_.ring_depth < 2) cfg_.ring_depth = 2;
        gpu_.create(cfg_);
        if (!model_file_.empty())
            for (int s = 0; s < n_; ++s)
                if (access(model_path(s).c_str(), R_OK) == 0) gpu_.check(oatgpu_mog_load(gpu_.ctx, s, model_path(s).c_str()));
        if (!mask_file_.empty()) {
            const GreyImage m = read_pnm_grey(mask_file_);               // FrameMasker.cpp:45-62: imread(.., GRAYSCALE)
            if (m.rows != p0.rows || m.cols != p0.cols) throw std::runtime_error("Mask image and frame source image do not have equal sizes");   // FrameMasker.cpp:77-81
            for (int s = 0; s < n_; ++s) gpu_.check(oatgpu_set_roi_mask(gpu_.ctx, s, m.px.data()));
        }
        if (kalman_) gpu_.check(oatgpu_set_kalman(gpu_.ctx, 1, dt_, timeout_, sig_accel_, sig_noise_));
        if (stage_copy_) gpu_.check(oatgpu_set_stage_copy(gpu_.ctx, stage_copy_));
        if (homography_on_) gpu_.check(oatgpu_set_homography(gpu_.ctx, 1, homography_));
        for (int s = 0; s < n_; ++s) {
            position_sinks_[s].bind(sink_addresses_[s], sink_addresses_[s]);
            shared_positions_.push_back(position_sinks_[s].retrieve());
        }
        frame_ptrs_.resize(n_);
        results_.resize(n_);
        return true;
    }

    // Results leave set by set, SINK by SINK (PositionDetector.cpp:88-96 per camera): publish_some() hands out up to
    // max_sinks tokens of the oldest outstanding result set -- with only_if_ready it never waits for the device -- so that
    // the staging loop below can publish BETWEEN its copies: a SINK's wait() is a round trip to the consumer process
    // (~30 us each; 8 cameras: 275 us a round), and done behind the round it was time the PCIe link sat idle
    // (r04, `--timing`: 8 x 1080p 6.2 k -> see DESIGN.md section 6).  Order per SINK is the frames' order.
    void publish_sink(int s)
    {
        const oatgpu_position &r = results_[s];
        Position2D pos("");
        pos.set_sample(pub_samples_[s]);                             // PositionDetector.cpp:80
        pos.position_valid = r.valid != 0;
        if (kalman_) {                                               // KalmanFilter2D.cpp:123-137
            pos.position.x = r.x; pos.position.y = r.y;
            pos.velocity.x = r.vx; pos.velocity.y = r.vy;
            pos.velocity_valid = r.velocity_valid != 0;
        } else if (r.valid) {                                        // DetectorFunc.cpp:46,58-60: x/y only when found
            pos.position.x = r.x; pos.position.y = r.y;
        }
        if (homography_on_) pos.setCoordSystem(DistanceUnit::WORLD, homography_);   // HomographyTransform2D.cpp:102
        position_sinks_[s].wait();
        *shared_positions_[s] = pos;
        position_sinks_[s].post();
    }
    void publish_some(int max_sinks, bool only_if_ready)
    {
        if (!collected_) {
            if (pending_.empty()) return;
            if (only_if_ready && oatgpu_track_ready(gpu_.ctx) != 1) return;
            gpu_.check(oatgpu_track_collect(gpu_.ctx, results_.data()));
            pub_samples_ = std::move(pending_.front());
            pending_.pop_front();
            collected_ = true;
            pub_cursor_ = 0;
        }
        for (; pub_cursor_ < n_ && max_sinks > 0; ++pub_cursor_, --max_sinks) publish_sink(pub_cursor_);
        if (pub_cursor_ == n_) collected_ = false;
    }
    // the set in progress to its end, or else the whole next set (waits for the device if it must)
    void publish() { publish_some(n_, false); }
    bool results_owed() const { return collected_ || !pending_.empty(); }

    int process() override
    {
        // ---- a frame from every camera (PositionDetector.cpp:63-75), camera by camera: as soon as camera s has delivered,
        // its H2D copy starts (oatgpu_track_stage), and camera s - 1, whose copy has meanwhile left its segment, is
        // posted (PositionDetector.cpp:78-86 per camera) -- the n frames cross one PCIe link one after the other, and
        // the upstream writers refill their segments while the later cameras are still being waited for and copied
        // (8 x 1080p: a round is the link's time, not the slowest writer's memcpy + the link's time) ----
        std::vector<Sample> samples(n_);
        double t0 = timing_ ? now_s() : 0.0, t1;
#define OAT_LAP(acc) do { if (timing_) { t1 = now_s(); acc += t1 - t0; t0 = t1; } } while (0)
        for (int s = 0; s < n_; ++s) {
            // Nothing to stage yet (this camera has not delivered): results that are owed leave NOW -- the reference publishes
            // a position as soon as it has one (PositionDetector.cpp:88-96); a frame that was only registered for a
            // two-frame launch goes out alone (oatgpu_track_collect), which is right when the cameras are slower than
            // the device.  With a frame waiting the loop goes on staging and the results leave between the copies.
            while (!quit && results_owed() && !frame_sources_[s].token_waiting()) publish_some(1, collected_);
            OAT_LAP(t_publish_);
            const NodeState st = frame_sources_[s].wait();
            OAT_LAP(t_wait_);
            if (st == NodeState::END) {
                // (frames of this round already staged are dropped with the round: their sources get their post, the
                // set is never registered; frames of earlier rounds still get their tokens)
                for (int q = (s > 0 ? s - 1 : 0); q < s; ++q) {
                    gpu_.check(oatgpu_track_input_consumed_stream(gpu_.ctx, q));
                    frame_sources_[q].post();
                }
                gpu_.check(oatgpu_track_stage_abort(gpu_.ctx));        // the partly staged set is given up, nothing is owed for it
                while (results_owed() && !quit) publish();
                print_timing();
                return 1;
            }
            const Frame &shm = *frame_sources_[s].retrieve();
            src_pins_[s].pin(shm);
            samples[s] = shm.sample();
            gpu_.check(oatgpu_track_stage(gpu_.ctx, s, shm.data()));
            OAT_LAP(t_stage_);
            // while this camera's copy runs: one token of a FINISHED result set to its SINK (never the newest set: asking for
            // a frame that is only registered would launch it alone and end the two-frames-a-launch pairing)
            if (collected_ || pending_.size() >= 2) { publish_some(1, true); OAT_LAP(t_publish_); }
            if (s > 0) {
                gpu_.check(oatgpu_track_input_consumed_stream(gpu_.ctx, s - 1));
                OAT_LAP(t_consumed_);
                frame_sources_[s - 1].post();
                OAT_LAP(t_post_);
            }
        }
        gpu_.check(oatgpu_track_input_consumed_stream(gpu_.ctx, n_ - 1));
        OAT_LAP(t_consumed_);
        frame_sources_[n_ - 1].post();
        OAT_LAP(t_post_);
        gpu_.check(oatgpu_track_enqueue_staged(gpu_.ctx, learning_coeff_));
        OAT_LAP(t_enqueue_);
        pending_.push_back(std::move(samples));
        ++rounds_;
        if (timing_) { t_last_ = now_s(); if (rounds_ == 16) t_r16_ = t_last_; }

        // ---- a free ring slot for the next round; everything else leaves while the loop waits for a camera (top of the
        // loop: at once when no frame is waiting -- minimum latency) or between the next round's copies (frames waiting) ----
        while ((int)pending_.size() == cfg_.ring_depth && !quit) publish();
        OAT_LAP(t_publish_);
#undef OAT_LAP
        return 0;
    }

    std::string name_;
    std::vector<std::string> source_addresses_, sink_addresses_;
    int n_, stream_base_, n_total_;
    std::vector<Source<Frame>> frame_sources_;
    std::vector<Sink<Position2D>> position_sinks_;
    std::vector<Position2D *> shared_positions_;
    std::vector<ShmRegistration> src_pins_;    // declared after the sources: unregistered before the segments are unmapped
    std::vector<const uint8_t *> frame_ptrs_;
    std::vector<oatgpu_position> results_;
    std::deque<std::vector<Sample>> pending_;  // Samples of the frames whose results are still on the device
    std::vector<Sample> pub_samples_;          // ... and of the result set that is being handed out (publish_some)
    bool collected_{false};
    int pub_cursor_{0};
    GpuCtx gpu_;
};

int main(int argc, char **argv)
{
    try {
        Options o = Options::parse(argc, argv,
            {{"a", "adaptation-coeff"}, {"H", "h-thresh"}, {"S", "s-thresh"}, {"V", "v-thresh"}, {"e", "erode"},
             {"d", "dilate"}, {"T", "timeout"}, {"n", "sigma-noise"}, {"f", "mask"}, {"h", "help"}, {"v", "version"}},
            {"help", "version", "kalman", "timing", "print-partition"});
        if (o.has("version")) { std::cout << "oat-track-hip (MI355X drop-in, liboatgpu ABI " << oatgpu_abi_version() << ")\n"; return 0; }
        if (o.has("help") || o.positional.size() != 2) {
            std::cout << "Usage: oat-track-hip SOURCE[,SOURCE..] SINK[,SINK..] [-a coeff] [-H [lo,hi]] [-S ..] [-V ..] [-e n] [-d n] [--area [min,max]]\n"
                         "       [--gpu-index N | N0,N1,..] [--ring D] [--model-file FILE] [-f|--mask FILE.pgm] [--stage-copy dma|kernel]\n"
                         "       [--thresh [lo,hi]]   GREY SOURCEs: framefilt mog -> posidet thresh instead of the HSV chain\n"
                         "       [--homography [h11,h12,...,h33]]   posifilt homography fused in (positions in world units)\n"
                         "       [--kalman [--dt s] [-T|--timeout s] [--sigma-accel a] [-n|--sigma-noise n]]   (posifilt kalman fused in)\n"
                         "N SOURCEs / N SINKs: N cameras batched into one device pass per frame; SOURCE i feeds SINK i.\n"
                         "--gpu-index N0,N1,..: the cameras are split into contiguous blocks, one per listed device (own context and thread).\n"
                         "--ingest-root D0 (with --gpu-index D0,D1,..): all frames are ingested on device D0 and scattered to their devices\n"
                         "       over RCCL send/recv (one process, one thread); --timing prints bytes per peer and ms per step.\n"
                         "--print-partition: print which cameras go to which device (both forms) and exit; no device is touched.\n";
            return o.has("help") ? 0 : -1;
        }
        o.apply_config({"adaptation-coeff", "h-thresh", "s-thresh", "v-thresh", "erode", "dilate", "area", "model-file",
                        "kalman", "dt", "timeout", "sigma-accel", "sigma-noise", "gpu-index", "ring", "mask", "thresh", "homography", "stage-copy", "timing",
                        "ingest-root", "print-partition"}, {"kalman", "timing", "print-partition"});
        const std::vector<std::string> sources = split_list(o.positional[0]), sinks = split_list(o.positional[1]);
        if (sources.size() != sinks.size()) throw std::runtime_error("need as many SINKs as SOURCEs");
        // --gpu-index N | N0,N1,...: one shard of the SOURCE list per listed device (contiguous blocks)
        std::vector<int> devices;
        for (const std::string &d : split_list(o.has("gpu-index") ? o.kv["gpu-index"] : std::string("0"))) {
            char *end = nullptr;
            const long v = strtol(d.c_str(), &end, 10);
            if (!end || *end || v < 0 || v > 1023) throw std::runtime_error("--gpu-index: expected N or N0,N1,... (device ordinals)");
            devices.push_back((int)v);
        }
        if (o.has("print-partition")) {
            // which cameras go to which device (camera s -> shard s / ceil(S / N), contiguous blocks, for life) --
            // printed without touching a device, for both launch forms; tests/test_host_pipeline.py holds it to oat_amd/dist.py
            const int S = (int)sources.size(), N = (int)devices.size(), per = (S + N - 1) / N;
            for (int k = 0; k < N && k * per < S; ++k)
                std::cout << "shard " << k << " device " << devices[k] << " cameras " << k * per << " " << std::min(S, (k + 1) * per)
                          << (o.has("ingest-root") ? (devices[k] == (int)o.num("ingest-root", 0, 0, 1023) ? " root" : " peer") : "") << "\n";
            return 0;
        }
        if (o.has("ingest-root")) {                                       // the stream-to-rank scatter over RCCL (scatter_tracker.hpp)
            for (const char *k : {"kalman", "thresh", "mask", "model-file", "homography", "stage-copy"})
                if (o.has(k)) throw std::runtime_error(std::string("--ingest-root does not take --") + k);
            ScatterTracker t(sources, sinks, devices, (int)o.num("ingest-root", 0, 0, 1023));
            t.learning_coeff_ = o.num("adaptation-coeff", 0.0, 0.0, 1.0);
            double a, b;
            if (o.arr2("h-thresh", a, b)) { t.cfg_.h_lo = (int)a; t.cfg_.h_hi = (int)b; }
            if (o.arr2("s-thresh", a, b)) { t.cfg_.s_lo = (int)a; t.cfg_.s_hi = (int)b; }
            if (o.arr2("v-thresh", a, b)) { t.cfg_.v_lo = (int)a; t.cfg_.v_hi = (int)b; }
            if (o.has("erode")) t.cfg_.erode = (int)o.num("erode", 0, 0, 1e6);
            if (o.has("dilate")) t.cfg_.dilate = (int)o.num("dilate", 0, 0, 1e6);
            if (o.arr2("area", a, b)) { t.cfg_.min_area = a; t.cfg_.max_area = b; }
            t.cfg_.ring_depth = (int)o.num("ring", 2, 2, 64);
            t.timing_ = o.has("timing");
            return t.run();
        }
        const int S = (int)sources.size(), N = (int)devices.size(), per = (S + N - 1) / N;
        std::vector<std::unique_ptr<BatchedTracker>> shards;
        for (int k = 0; k < N && k * per < S; ++k) {
            const int s0 = k * per, s1 = std::min(S, s0 + per);
            auto t = std::make_unique<BatchedTracker>(std::vector<std::string>(sources.begin() + s0, sources.begin() + s1),
                                                      std::vector<std::string>(sinks.begin() + s0, sinks.begin() + s1), s0, S);
            t->learning_coeff_ = o.num("adaptation-coeff", 0.0, 0.0, 1.0);
            double a, b;
            if (o.arr2("h-thresh", a, b)) { t->cfg_.h_lo = (int)a; t->cfg_.h_hi = (int)b; }
            if (o.arr2("s-thresh", a, b)) { t->cfg_.s_lo = (int)a; t->cfg_.s_hi = (int)b; }
            if (o.arr2("v-thresh", a, b)) { t->cfg_.v_lo = (int)a; t->cfg_.v_hi = (int)b; }
            if (o.arr2("thresh", a, b)) {                                 // SimpleThreshold.cpp:86-97
                if (a < 0 || a > 256 || b < 0 || b > 256) throw std::runtime_error("Values of thresh should be between 0 and 256.");
                t->grey_ = true;
                t->cfg_.h_lo = (int)a; t->cfg_.h_hi = (int)b;              // the one-channel window lives in the h slot
            }
            if (o.has("erode")) t->cfg_.erode = (int)o.num("erode", 0, 0, 1e6);
            if (o.has("dilate")) t->cfg_.dilate = (int)o.num("dilate", 0, 0, 1e6);
            if (o.arr2("area", a, b)) { t->cfg_.min_area = a; t->cfg_.max_area = b; }
            t->cfg_.device = devices[k];
            t->cfg_.ring_depth = (int)o.num("ring", 2, 1, 64);
            if (o.has("model-file")) t->model_file_ = o.kv["model-file"];
            if (o.has("mask")) t->mask_file_ = o.kv["mask"];
            if (o.has("stage-copy")) {
                if (o.kv["stage-copy"] == "kernel") t->stage_copy_ = 1;
                else if (o.kv["stage-copy"] != "dma") throw std::runtime_error("--stage-copy: expected dma or kernel");
            }
            t->timing_ = o.has("timing");
            t->kalman_ = o.has("kalman");
            t->homography_on_ = o.arr9("homography", t->homography_);
            t->dt_ = o.num("dt", 0.02, 0, 1e9);                        // KalmanFilter2D.cpp:69-85 (lower bound 0)
            t->timeout_ = o.num("timeout", 0.0, 0, 1e18);
            t->sig_accel_ = o.num("sigma-accel", 5.0, 0, 1e18);
            t->sig_noise_ = o.num("sigma-noise", 0.0, 0, 1e18);
            shards.push_back(std::move(t));
        }
        if (shards.size() == 1) return shards[0]->run();
        // one thread per shard: a device context belongs to the thread that drives it; SIGINT reaches every loop
        // through `quit` (the waits are 10 ms slices), END of a shard's SOURCEs ends that shard only
        // A shard that ends -- END of its SOURCEs, or a failure -- is DESTROYED at once, by its own thread: its SINKs go
        // END and its SOURCE slots are released while the other shards keep running, exactly what the exit of the
        // reference's one-camera process does for its consumers and producers (lib/shmemdf/Sink.h:73-91,
        // Source.h:90-112); the process exit code still reports the failure.
        std::vector<int> rc(shards.size(), 0);
        std::vector<std::thread> th;
        for (size_t k = 0; k < shards.size(); ++k)
            th.emplace_back([&, k] {
                pin_thread_to_device_node(shards[k]->cfg_.device);      // the shard's thread next to its GPU
                rc[k] = shards[k]->run();
                shards[k].reset();
            });
        for (auto &t : th) t.join();
        for (int r : rc) if (r) return r;
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "oat-track-hip: " << e.what() << std::endl;
        return -1;
    }
}
