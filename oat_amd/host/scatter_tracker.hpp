// scatter_tracker.hpp -- `oat-track-hip --ingest-root D0 --gpu-index D0,D1,..`: the stream-to-rank scatter behind the C++ boundary.
//
// SURVEY.md 8e / north_star: "RCCL over xGMI only for the trivial stream-to-rank scatter".  The per-rank-ingest launcher
// (oat-track-hip --gpu-index alone: every device's thread reads its own cameras over its own PCIe link) is the realistic
// camera topology; THIS form is the one where every frame originates on ONE device -- a capture host whose frames land on
// GPU D0 -- and travels to the device that owns its stream over xGMI:
//
//   ONE process, ONE thread, ncclCommInitAll over the listed devices (one communicator per device);
//   per step, slot = step mod 2:
//     ingest    every camera's frame out of its shared-memory SOURCE into the root's staging slot (H2D, root device), SOURCEs posted
//               (PositionDetector.cpp:63-86 per camera)
//     gate      oatgpu_track_input_consumed on every shard: the per-pixel kernels that read this slot two steps ago are through
//     scatter   ncclGroupStart; ncclSend x (N - 1) on the root's communicator / ncclRecv on each peer's; ncclGroupEnd -- on a
//               transfer stream per device (xGMI is point-to-point: the root's links carry the blocks side by side; RCCL has no
//               scatter primitive).  The root's own block stays where the ingest put it.
//     consume   every shard's context stream waits for its transfer's event (no host wait), oatgpu_track_enqueue_dev
//     publish   the PREVIOUS step's results, SINK by SINK (PositionDetector.cpp:88-96), while this step computes
//
// The partition is the launcher's and dist.py's: camera s lives on shard s / ceil(S / N) for life.  What FrameScatterPipe does in
// oat_amd/dist.py:118-161 over torch.distributed, without Python.  The reference has no counterpart (its only ingest is the
// shared-memory SOURCE, lib/shmemdf/Source.h:114-232).  UNVERIFIED for N > 1: the builder has one GPU at a time; the world of one
// runs end to end against the oracle (tests/test_host_pipeline.py), N > 1 is a code path the driver's node exercises
// (`--timing` prints bytes per peer and milliseconds per step).
#pragma once

#include "component.hpp"

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <array>
#include <chrono>
#include <deque>

namespace oat {

#define OAT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)
#define OAT_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) throw std::runtime_error(std::string(#expr) + ": " + ncclGetErrorString(r_)); } while (0)

class ScatterTracker : public Component {
public:
    ScatterTracker(const std::vector<std::string> &sources, const std::vector<std::string> &sinks, const std::vector<int> &devices,
                   int root_device)
        : source_addresses_(sources), sink_addresses_(sinks), devices_(devices), S_((int)sources.size())
    {
        if (sources.size() != sinks.size()) throw std::runtime_error("need as many SINKs as SOURCEs");
        per_ = (S_ + (int)devices.size() - 1) / (int)devices.size();
        while ((int)devices_.size() > 1 && ((int)devices_.size() - 1) * per_ >= S_) devices_.pop_back();     // more devices than blocks
        N_ = (int)devices_.size();
        root_ = -1;
        for (int k = 0; k < N_; ++k) if (devices_[k] == root_device && root_ < 0) root_ = k;
        if (root_ < 0) throw std::runtime_error("--ingest-root: the root device must be one of --gpu-index");
        for (int k = 0; k < N_; ++k)
            for (int q = 0; q < k; ++q)
                if (devices_[k] == devices_[q]) throw std::runtime_error("--ingest-root: every device of --gpu-index once (one RCCL rank per device)");
        oatgpu_default_config(&cfg_);
        name_ = "track-scatter[" + sources[0] + (S_ > 1 ? ",..(" + std::to_string(S_) + ")" : "") + " root " + std::to_string(root_device) +
                " -> " + std::to_string(N_) + " device(s)]";
        frame_sources_ = std::vector<Source<Frame>>(S_);
        position_sinks_ = std::vector<Sink<Position2D>>(S_);
        src_pins_ = std::vector<ShmRegistration>(S_);
    }
    ~ScatterTracker() override
    {
        for (auto &g : gpu_) g.reset();                       // contexts first: they drain their streams
        for (int k = 0; k < (int)comm_.size(); ++k) if (comm_[k]) ncclCommDestroy(comm_[k]);
        for (int k = 0; k < (int)xfer_.size(); ++k) {
            (void)hipSetDevice(devices_[k]);
            for (int q = 0; q < 2; ++q) {
                if (ev_x_[k][q]) (void)hipEventDestroy(ev_x_[k][q]);
                if (k != root_ && block_[k][q]) (void)hipFree(block_[k][q]);
            }
            if (xfer_[k]) (void)hipStreamDestroy(xfer_[k]);
        }
        if (root_ >= 0 && root_ < (int)devices_.size()) {
            (void)hipSetDevice(devices_[root_]);
            for (int q = 0; q < 2; ++q) { if (stage_[q]) (void)hipFree(stage_[q]); if (ev_in_[q]) (void)hipEventDestroy(ev_in_[q]); }
            if (copy_) (void)hipStreamDestroy(copy_);
        }
    }
    std::string name() const override { return name_; }
    oatgpu_config cfg_;
    double learning_coeff_{0.0};
    bool timing_{false};

    void print_timing() const
    {
        if (!timing_ || !steps_) return;
        int v = 0;
        ncclGetVersion(&v);
        std::fprintf(stderr, "%s: %llu steps, %d cameras over %d device(s) (block of %d), RCCL %d: %zu bytes per peer and step; per step (ms): "
                             "ingest %.3f, gate %.3f, scatter calls %.3f, enqueue %.3f, collect+publish %.3f; %.1f fps aggregate\n",
                     name().c_str(), steps_, S_, N_, per_, v, (size_t)per_ * frame_bytes_, t_ingest_ * 1e3 / steps_, t_gate_ * 1e3 / steps_,
                     t_scatter_ * 1e3 / steps_, t_enqueue_ * 1e3 / steps_, t_publish_ * 1e3 / steps_,
                     steps_ > 8 && t_last_ > t_first_ ? (double)(steps_ - 8) * S_ / (t_last_ - t_first_) : 0.0);
    }

protected:
    static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    int block_begin(int k) const { return k * per_; }
    int block_len(int k) const { return std::max(0, std::min(S_, (k + 1) * per_) - k * per_); }

    bool connectToNode() override
    {
        try {
            return connect_all();
        } catch (...) {                                        // SINKs that never came up still go END (oat_track_hip.cpp, connectToNode)
            for (int s = 0; s < S_; ++s) {
                if ((size_t)s < shared_positions_.size()) continue;
                try { position_sinks_[s].bind(sink_addresses_[s], sink_addresses_[s]); } catch (...) {}
            }
            throw;
        }
    }

    bool connect_all()
    {
        for (int s = 0; s < S_; ++s) frame_sources_[s].touch(source_addresses_[s]);
        FrameParams p0{};
        for (int s = 0; s < S_; ++s) {
            if (frame_sources_[s].connect(PIX_BGR) != SourceState::CONNECTED) return false;
            const FrameParams p = frame_sources_[s].parameters();
            if (s == 0) p0 = p;
            else if (p.rows != p0.rows || p.cols != p0.cols)
                throw std::runtime_error("all SOURCEs of one scatter tracker must have the same frame geometry");
        }
        frame_bytes_ = (size_t)p0.rows * p0.cols * 3;
        // ---- the communicators: one rank per listed device, this process owns them all ----
        comm_.assign(N_, nullptr);
        OAT_NCCL(ncclCommInitAll(comm_.data(), N_, devices_.data()));
        xfer_.assign(N_, nullptr);
        ev_x_.assign(N_, {nullptr, nullptr});
        block_.assign(N_, {nullptr, nullptr});
        for (int k = 0; k < N_; ++k) {
            OAT_HIP(hipSetDevice(devices_[k]));
            OAT_HIP(hipStreamCreateWithFlags(&xfer_[k], hipStreamNonBlocking));
            for (int q = 0; q < 2; ++q) OAT_HIP(hipEventCreateWithFlags(&ev_x_[k][q], hipEventDisableTiming));
        }
        OAT_HIP(hipSetDevice(devices_[root_]));
        OAT_HIP(hipStreamCreateWithFlags(&copy_, hipStreamNonBlocking));
        for (int q = 0; q < 2; ++q) {
            OAT_HIP(hipMalloc((void **)&stage_[q], (size_t)S_ * frame_bytes_));
            OAT_HIP(hipEventCreateWithFlags(&ev_in_[q], hipEventDisableTiming));
        }
        for (int k = 0; k < N_; ++k) {
            if (k == root_) { for (int q = 0; q < 2; ++q) block_[k][q] = stage_[q] + (size_t)block_begin(k) * frame_bytes_; continue; }
            OAT_HIP(hipSetDevice(devices_[k]));
            for (int q = 0; q < 2; ++q) OAT_HIP(hipMalloc((void **)&block_[k][q], (size_t)block_len(k) * frame_bytes_));
        }
        // ---- one batched context per shard (its streams' models live there for life) ----
        gpu_.clear();
        for (int k = 0; k < N_; ++k) {
            oatgpu_config c = cfg_;
            c.rows = (int)p0.rows; c.cols = (int)p0.cols; c.n_streams = block_len(k); c.channels = 3; c.device = devices_[k];
            if (c.ring_depth < 2) c.ring_depth = 2;
            gpu_.push_back(std::make_unique<GpuCtx>());
            gpu_.back()->create(c);
        }
        for (int s = 0; s < S_; ++s) {
            position_sinks_[s].bind(sink_addresses_[s], sink_addresses_[s]);
            shared_positions_.push_back(position_sinks_[s].retrieve());
        }
        results_.resize(S_);
        return true;
    }

    void publish_set()
    {
        if (pending_.empty()) return;
        for (int k = 0; k < N_; ++k) gpu_[k]->check(oatgpu_track_collect(gpu_[k]->ctx, results_.data() + block_begin(k)));
        const std::vector<Sample> samples = std::move(pending_.front());
        pending_.pop_front();
        for (int s = 0; s < S_; ++s) {                          // PositionDetector.cpp:80, 88-96; DetectorFunc.cpp:46,58-60
            const oatgpu_position &r = results_[s];
            Position2D pos("");
            pos.set_sample(samples[s]);
            pos.position_valid = r.valid != 0;
            if (r.valid) { pos.position.x = r.x; pos.position.y = r.y; }
            position_sinks_[s].wait();
            *shared_positions_[s] = pos;
            position_sinks_[s].post();
        }
    }

    int process() override
    {
        const int q = (int)(steps_ & 1ull);
        double t0 = timing_ ? now_s() : 0.0, t1;
#define OAT_LAP(acc) do { if (timing_) { t1 = now_s(); acc += t1 - t0; t0 = t1; } } while (0)
        // ---- ingest: every camera's frame into the root's staging slot q.  The slot was last read by step - 2's sends and by
        // the root shard's per-pixel kernel of step - 2: both are behind the gate of step - 1 and the event wait below. ----
        OAT_HIP(hipSetDevice(devices_[root_]));
        if (steps_ >= 2) OAT_HIP(hipEventSynchronize(ev_x_[root_][q]));
        std::vector<Sample> samples(S_);
        for (int s = 0; s < S_; ++s) {
            const NodeState st = frame_sources_[s].wait();
            if (st == NodeState::END) {
                OAT_HIP(hipStreamSynchronize(copy_));           // copies of this round already queued read SOURCE segments
                for (int p = 0; p < s; ++p) frame_sources_[p].post();
                while (!pending_.empty() && !quit) publish_set();
                print_timing();
                return 1;
            }
            const Frame &shm = *frame_sources_[s].retrieve();
            src_pins_[s].pin(shm);
            samples[s] = shm.sample();
            OAT_HIP(hipMemcpyAsync(stage_[q] + (size_t)s * frame_bytes_, shm.data(), frame_bytes_, hipMemcpyHostToDevice, copy_));
            (void)hipStreamQuery(copy_);             // submit NOW: ROCm 7.2 holds a queued copy back until something flushes its stream (oatgpu_track_stage)
        }
        OAT_HIP(hipEventRecord(ev_in_[q], copy_));
        // (publishing the previous step's positions HERE, under the copies, was measured and is slower: 8 x 1080p 6.55 k -> 5.8-6.1 k
        // fps -- the SINK hand-shakes and the copy engine's completion handling share this one thread; they leave behind the enqueue)
        OAT_HIP(hipEventSynchronize(ev_in_[q]));                 // the frames have left their segments ...
        for (int s = 0; s < S_; ++s) frame_sources_[s].post();  // ... the cameras may refill them (PositionDetector.cpp:78-86)
        OAT_LAP(t_ingest_);
        // ---- gate: the kernels that read the peers' slot q (step - 2) are through (oatgpu_track_input_consumed) ----
        for (int k = 0; k < N_; ++k) gpu_[k]->check(oatgpu_track_input_consumed(gpu_[k]->ctx));
        OAT_LAP(t_gate_);
        // ---- scatter: block k of the root's slot -> device k, all peers in one group ----
        // (the gate's library calls left the LAST shard's device current: every call below names its device first -- the
        // single-process form of RCCL's group calls is `set device i; ncclXxx(.., comm[i], stream[i])`)
        OAT_HIP(hipSetDevice(devices_[root_]));
        OAT_HIP(hipStreamWaitEvent(xfer_[root_], ev_in_[q], 0));
        if (N_ > 1) {
            OAT_NCCL(ncclGroupStart());
            for (int k = 0; k < N_; ++k) {
                if (k == root_ || !block_len(k)) continue;
                const size_t bytes = (size_t)block_len(k) * frame_bytes_;
                OAT_HIP(hipSetDevice(devices_[root_]));
                OAT_NCCL(ncclSend(stage_[q] + (size_t)block_begin(k) * frame_bytes_, bytes, ncclUint8, k, comm_[root_], xfer_[root_]));
                OAT_HIP(hipSetDevice(devices_[k]));
                OAT_NCCL(ncclRecv(block_[k][q], bytes, ncclUint8, root_, comm_[k], xfer_[k]));
            }
            OAT_NCCL(ncclGroupEnd());
        }
        for (int k = 0; k < N_; ++k) {
            OAT_HIP(hipSetDevice(devices_[k]));
            OAT_HIP(hipEventRecord(ev_x_[k][q], xfer_[k]));
        }
        OAT_LAP(t_scatter_);
        // ---- consume: every shard's stream behind ITS transfer (an event, never the host), then the fused chain ----
        for (int k = 0; k < N_; ++k) {
            if (!block_len(k)) continue;
            OAT_HIP(hipSetDevice(devices_[k]));
            OAT_HIP(hipStreamWaitEvent((hipStream_t)oatgpu_get_stream(gpu_[k]->ctx), ev_x_[k][q], 0));
            gpu_[k]->check(oatgpu_track_enqueue_dev(gpu_[k]->ctx, block_[k][q], learning_coeff_));
        }
        OAT_LAP(t_enqueue_);
        pending_.push_back(std::move(samples));
        ++steps_;
        if (timing_) { t_last_ = now_s(); if (steps_ == 8) t_first_ = t_last_; }
        // ---- publish: the previous step's positions leave while this one computes; a lone step's at once when no frame waits ----
        while (pending_.size() >= 2 && !quit) publish_set();
        if (!pending_.empty() && !frame_sources_[0].token_waiting()) publish_set();
        OAT_LAP(t_publish_);
#undef OAT_LAP
        return 0;
    }

    std::string name_;
    std::vector<std::string> source_addresses_, sink_addresses_;
    std::vector<int> devices_;
    int S_, N_{0}, per_{0}, root_{-1};
    size_t frame_bytes_{0};
    std::vector<Source<Frame>> frame_sources_;
    std::vector<Sink<Position2D>> position_sinks_;
    std::vector<Position2D *> shared_positions_;
    std::vector<ShmRegistration> src_pins_;
    std::vector<oatgpu_position> results_;
    std::deque<std::vector<Sample>> pending_;
    std::vector<ncclComm_t> comm_;
    std::vector<hipStream_t> xfer_;
    std::vector<std::array<hipEvent_t, 2>> ev_x_;
    std::vector<std::array<uint8_t *, 2>> block_;
    uint8_t *stage_[2]{nullptr, nullptr};
    hipEvent_t ev_in_[2]{nullptr, nullptr};
    hipStream_t copy_{nullptr};
    std::vector<std::unique_ptr<GpuCtx>> gpu_;
    unsigned long long steps_{0};
    double t_ingest_{0}, t_gate_{0}, t_scatter_{0}, t_enqueue_{0}, t_publish_{0}, t_first_{0}, t_last_{0};
};

}  // namespace oat
