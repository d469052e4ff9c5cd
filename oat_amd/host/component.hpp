// component.hpp -- the callers either side of the hot path, mirrored from jonnew/Oat:
//
//   oat::Component::run / runComponent   lib/base/Component.cpp:36-76   (SIGINT -> quit, END -> exit)
//   oat::FrameFilter::connectToNode/process    src/framefilter/FrameFilter.cpp:37-98
//   oat::PositionDetector::connectToNode/process  src/positiondetector/PositionDetector.cpp:40-99
//
// The virtual filter() / detectPosition() are implemented over the C ABI of liboatgpu.so.
#pragma once

#include "shmemdf.hpp"
#include "../../include/oatgpu.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>

namespace oat {

volatile sig_atomic_t quit = 0;

inline void sigHandler(int) { quit = 1; }

class Component {
public:
    virtual ~Component() {}
    virtual std::string name() const = 0;
    // lib/base/Component.cpp:50-76
    int run()
    {
        std::signal(SIGINT, sigHandler);
        std::signal(SIGTERM, sigHandler);
        try {
            if (!connectToNode()) return 0;
            int eos = 0;
            while (!eos && !quit) eos = process();
        } catch (const std::exception &e) {
            std::cerr << name() << ": " << e.what() << std::endl;   // whoError(name, what)
            return -1;
        }
        std::cout << name() << ": Exiting." << std::endl;
        return 0;
    }

protected:
    virtual bool connectToNode() = 0;
    virtual int process() = 0;
};

struct GpuCtx {
    oatgpu_ctx *ctx{nullptr};
    ~GpuCtx() { if (ctx) oatgpu_destroy(ctx); }
    void create(const oatgpu_config &cfg)
    {
        ctx = oatgpu_create(&cfg);
        if (!ctx) throw std::runtime_error(std::string("oatgpu_create: ") + oatgpu_last_error(nullptr));
    }
    void check(int rc) const
    {
        if (rc != OATGPU_OK) throw std::runtime_error(std::string("oatgpu: ") + oatgpu_last_error(ctx));
    }
};

// A page-locked frame the GPU copies into / out of by DMA (oatgpu_host_alloc), and the registration
// of an existing shared-memory frame (oatgpu_host_register).  Both fall back to ordinary memory if
// pinning fails (e.g. a memlock limit): the copies then go through the runtime's bounce buffers.
class PinnedFrame {
public:
    ~PinnedFrame() { if (mem_) oatgpu_host_free(mem_); }
    Frame &get(size_t rows, size_t cols, PixelColor color)
    {
        const size_t bytes = rows * cols * color_bytes(color);
        if (bytes != bytes_) {
            if (mem_) oatgpu_host_free(mem_);
            mem_ = oatgpu_host_alloc(bytes);
            bytes_ = bytes;
            frame_ = mem_ ? Frame(rows, cols, color, mem_, &sample_) : Frame(rows, cols, color);
        }
        frame_.set_color(color);
        return frame_;
    }
private:
    void *mem_{nullptr};
    size_t bytes_{0};
    Sample sample_;
    Frame frame_;
};
struct ShmRegistration {
    void *p{nullptr};
    ~ShmRegistration() { if (p) oatgpu_host_unregister(p); }
    void pin(const Frame &f) { if (!p && oatgpu_host_register((void *)f.data(), f.bytes()) == OATGPU_OK) p = (void *)f.data(); }
};

// src/framefilter/FrameFilter.h:36-86
class FrameFilter : public Component {
public:
    FrameFilter(const std::string &source, const std::string &sink)
        : name_("framefilt[" + source + "->" + sink + "]"), frame_source_address_(source), frame_sink_address_(sink) {}
    std::string name() const override { return name_; }

protected:
    virtual void filter(Frame &frame) = 0;                  // FrameFilter.h:64
    // GPU fast path: filter straight out of the (registered) shared-memory frame into `out`; the
    // default is the reference's copy-then-filter.  Returning false selects the reference order
    // (copy, post, filter) for this component.
    virtual bool filter_from_shm(const Frame &, Frame &) { return false; }
    virtual PixelColor sink_color(PixelColor in) const { return in; }
    virtual void configure_for(const FrameParams &) {}
    // The device context of a GPU filter whose filter_from_shm() may complete DEFERRED (oatgpu_set_deferred): process()
    // then posts the SOURCE as soon as the frame has left shared memory and has the result copied straight into the SINK's
    // shared frame (oatgpu_fetch_frame) -- no staging frame, no memcpy into the sink.  nullptr: the plain order below.
    virtual oatgpu_ctx *deferred_ctx() { return nullptr; }

    // FrameFilter.cpp:37-57
    bool connectToNode() override
    {
        frame_source_.touch(frame_source_address_);
        if (frame_source_.connect() != SourceState::CONNECTED) return false;
        auto p = frame_source_.parameters();
        configure_for(p);
        const PixelColor out = sink_color(p.color);
        const size_t bytes = p.rows * p.cols * color_bytes(out);
        frame_sink_.bind(frame_sink_address_, bytes);
        shared_frame_ = frame_sink_.retrieve(p.rows, p.cols, color_cvtype(out), out);
        return true;
    }
    // FrameFilter.cpp:59-98.  The reference copies the shared frame out, posts, then filters the
    // copy.  The GPU components instead DMA the (page-locked) shared frame to the device INSIDE the
    // read critical section -- shorter than the reference's memcpy -- and receive the result in a
    // page-locked internal frame; token order and count are unchanged.
    int process() override
    {
        if (frame_source_.wait() == NodeState::END) return 1;
        const Frame &shm = *frame_source_.retrieve();
        src_pin_.pin(shm);
        const PixelColor out_color = shared_frame_.color();
        if (oatgpu_ctx *dc = deferred_ctx()) {
            // FrameFilter.cpp:59-98 with the GPU in the middle: the frame is DMA'd out of the source's segment (the read
            // critical section ends there, as the reference's ends behind its memcpy), the kernels run while this
            // component waits for its sink, and the result is DMA'd straight into the sink's segment inside the write
            // critical section (where the reference memcpys).  One token out per token in, in order, with its Sample.
            if (!deferred_on_) {
                if (oatgpu_set_deferred(dc, 1) != OATGPU_OK) throw std::runtime_error(std::string("oatgpu: ") + oatgpu_last_error(dc));
                deferred_on_ = true;
            }
            Frame nowhere(shm.rows(), shm.cols(), out_color, shared_frame_.data(), &deferred_sample_);   // (never written: deferred)
            if (!filter_from_shm(shm, nowhere)) throw std::runtime_error("deferred filter refused the frame");
            deferred_sample_ = shm.sample();
            frame_source_.post();
            frame_sink_.wait();
            sink_pin_.pin(shared_frame_);
            if (oatgpu_fetch_frame(dc, shared_frame_.data()) != OATGPU_OK)
                throw std::runtime_error(std::string("oatgpu: ") + oatgpu_last_error(dc));
            shared_frame_.sample() = deferred_sample_;
            frame_sink_.post();
            return 0;
        }
        Frame &internal_frame = internal_.get(shm.rows(), shm.cols(), out_color);
        if (filter_from_shm(shm, internal_frame)) {
            internal_frame.sample() = shm.sample();
            frame_source_.post();
        } else {
            internal_frame.set_color(shm.color());
            frame_source_.copyTo(internal_frame);
            frame_source_.post();
            filter(internal_frame);
        }

        frame_sink_.wait();
        internal_frame.copyTo(shared_frame_);
        frame_sink_.post();
        return 0;
    }

    std::string name_, frame_source_address_, frame_sink_address_;
    Source<Frame> frame_source_;
    Sink<Frame> frame_sink_;
    Frame shared_frame_;
    PinnedFrame internal_;
    ShmRegistration src_pin_;      // declared after the source: unregistered before the segment is unmapped
    ShmRegistration sink_pin_;     // ... and the sink's frame (the deferred path copies into it by DMA)
    Sample deferred_sample_;
    bool deferred_on_{false};
};

// src/positiondetector/PositionDetector.h:43-90
class PositionDetector : public Component {
public:
    PositionDetector(const std::string &source, const std::string &sink)
        : name_("posidet[" + source + "->" + sink + "]"), frame_source_address_(source), position_sink_address_(sink) {}
    std::string name() const override { return name_; }

protected:
    virtual void detectPosition(Frame &frame, Position2D &position) = 0;   // PositionDetector.h:65
    // GPU fast path: detect straight out of the (registered) shared-memory frame, see FrameFilter.
    virtual bool detect_from_shm(const Frame &, Position2D &) { return false; }
    // deferred detectors (oatgpu_set_deferred): detect_from_shm returned when the frame had been read; the position comes here,
    // AFTER the source was posted
    virtual void detect_finish(Position2D &) {}
    virtual void configure_for(const FrameParams &) {}
    PixelColor required_color_{PIX_BGR};

    // PositionDetector.cpp:40-56
    bool connectToNode() override
    {
        frame_source_.touch(frame_source_address_);
        if (frame_source_.connect(required_color_) != SourceState::CONNECTED) return false;
        configure_for(frame_source_.parameters());
        position_sink_.bind(position_sink_address_, position_sink_address_);
        shared_position_ = position_sink_.retrieve();
        return true;
    }
    // PositionDetector.cpp:58-99
    int process() override
    {
        Position2D internal_pos("");
        if (frame_source_.wait() == NodeState::END) return 1;
        const Frame &shm = *frame_source_.retrieve();
        src_pin_.pin(shm);
        internal_pos.set_sample(shm.sample());                 // PositionDetector.cpp:80
        if (detect_from_shm(shm, internal_pos)) {
            frame_source_.post();
            detect_finish(internal_pos);
        } else {
            Frame internal_frame;
            frame_source_.copyTo(internal_frame);
            frame_source_.post();
            detectPosition(internal_frame, internal_pos);
        }

        position_sink_.wait();
        *shared_position_ = internal_pos;
        position_sink_.post();
        return 0;
    }

    std::string name_, frame_source_address_, position_sink_address_;
    Source<Frame> frame_source_;
    Sink<Position2D> position_sink_;
    Position2D *shared_position_{nullptr};
    ShmRegistration src_pin_;
};

// ---- image files for `framefilt mask -f` (FrameMasker.cpp:45-62) and `framefilt bsub -f` (BackgroundSubtractor.cpp:
// 52-68).  The reference calls cv::imread; there is no image decoder here, so the binaries read the
// Netpbm family -- P5/P2 (grey), P6/P3 (colour), maxval <= 255 -- which every image tool writes. ----
struct PnmImage {
    size_t rows{0}, cols{0};
    int channels{0};                 // 1, or 3 in B,G,R order (what imread(.., IMREAD_COLOR) returns)
    std::vector<uint8_t> px;
};
inline PnmImage read_pnm(const std::string &path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("File \"" + path + "\" could not be read.");      // FrameMasker.cpp:57
    auto token = [&]() {
        std::string t;
        int ch;
        while ((ch = in.get()) != EOF) {
            if (ch == '#') { while ((ch = in.get()) != EOF && ch != '\n') {} continue; }
            if (isspace(ch)) { if (!t.empty()) break; continue; }
            t.push_back((char)ch);
        }
        return t;
    };
    const std::string magic = token();
    if (magic != "P5" && magic != "P2" && magic != "P6" && magic != "P3")
        throw std::runtime_error("'" + path + "' is not a PGM/PPM image (only the Netpbm formats P2, P3, P5, P6 are read here)");
    PnmImage im;
    im.cols = strtoul(token().c_str(), nullptr, 10);
    im.rows = strtoul(token().c_str(), nullptr, 10);
    const unsigned long maxval = strtoul(token().c_str(), nullptr, 10);    // consumes exactly one whitespace byte after it
    if (!im.rows || !im.cols || maxval < 1 || maxval > 255) throw std::runtime_error("'" + path + "': unsupported PNM header");
    im.channels = (magic == "P6" || magic == "P3") ? 3 : 1;
    const size_t n = im.rows * im.cols * im.channels;
    im.px.resize(n);
    if (magic == "P5" || magic == "P6") {
        in.read((char *)im.px.data(), (std::streamsize)n);
        if ((size_t)in.gcount() != n) throw std::runtime_error("'" + path + "' is truncated");
    } else {
        for (size_t i = 0; i < n; ++i) {
            const std::string t = token();
            if (t.empty()) throw std::runtime_error("'" + path + "' is truncated");
            im.px[i] = (uint8_t)strtoul(t.c_str(), nullptr, 10);
        }
    }
    if (maxval != 255) for (auto &v : im.px) v = (uint8_t)((v * 255u + maxval / 2) / maxval);
    if (im.channels == 3) for (size_t i = 0; i < n; i += 3) std::swap(im.px[i], im.px[i + 2]);    // RGB file -> BGR
    return im;
}
struct GreyImage { size_t rows{0}, cols{0}; std::vector<uint8_t> px; };
// imread(path, IMREAD_GRAYSCALE): colour files go through the same fixed-point BGR->grey as cvtColor
// ((1868 B + 9617 G + 4899 R + 8192) >> 14, the constants of Threshold.cpp's path as well)
inline GreyImage read_pnm_grey(const std::string &path)
{
    PnmImage im = read_pnm(path);
    GreyImage g;
    g.rows = im.rows; g.cols = im.cols;
    if (im.channels == 1) { g.px = std::move(im.px); return g; }
    g.px.resize(im.rows * im.cols);
    for (size_t i = 0; i < g.px.size(); ++i)
        g.px[i] = (uint8_t)((1868 * im.px[3 * i] + 9617 * im.px[3 * i + 1] + 4899 * im.px[3 * i + 2] + (1 << 13)) >> 14);
    return g;
}

// ---- option parsing: "TYPE SOURCE SINK [--key value | -k value | --flag]" with the reference's
// names (SURVEY.md 8a "Option surface"); array values are TOML literals like "[0,256]". ----
struct Options {
    std::vector<std::string> positional;
    std::map<std::string, std::string> kv;
    std::string config_file, config_key;      // -c FILE KEY (Configurable.h:46-49)

    // `-c FILE KEY`: table [KEY] of a TOML file supplies options the command line did not
    // (TOMLSanitize.h:73-98 getConfigTable, :102-118 checkKeys, :169-196 getValue -- the command
    // line wins).  The reader covers the TOML the reference's configurations use: comments,
    // [table] headers, `key = value` with numbers, booleans, quoted strings and (possibly
    // multi-line) arrays of those; array values are handed on as their literal text, which is what
    // arr2() parses.
    void apply_config(const std::vector<std::string> &valid_keys, const std::vector<std::string> &flags = {})
    {
        if (config_file.empty()) return;
        std::ifstream in(config_file);
        if (!in) throw std::runtime_error("Could not open configuration file '" + config_file + "'.");
        auto trim = [](std::string t) {
            const size_t a = t.find_first_not_of(" \t\r\n"), b = t.find_last_not_of(" \t\r\n");
            return a == std::string::npos ? std::string() : t.substr(a, b - a + 1);
        };
        auto strip_comment = [](const std::string &l) {
            char q = 0;
            for (size_t i = 0; i < l.size(); ++i) {
                if (q) { if (l[i] == q) q = 0; }
                else if (l[i] == '"' || l[i] == '\'') q = l[i];
                else if (l[i] == '#') return l.substr(0, i);
            }
            return l;
        };
        std::string line, table;
        bool found = false;
        int lineno = 0;
        while (std::getline(in, line)) {
            ++lineno;
            line = trim(strip_comment(line));
            if (line.empty()) continue;
            if (line.front() == '[' && line.find('=') == std::string::npos) {
                if (line.back() != ']') throw std::runtime_error(config_file + ":" + std::to_string(lineno) + ": malformed table header");
                table = trim(line.substr(1, line.size() - 2));
                if (table == config_key) found = true;
                continue;
            }
            const size_t eq = line.find('=');
            if (eq == std::string::npos) throw std::runtime_error(config_file + ":" + std::to_string(lineno) + ": expected key = value");
            std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
            if (key.size() > 1 && (key.front() == '"' || key.front() == '\'')) key = key.substr(1, key.size() - 2);
            auto depth = [](const std::string &v) {
                int d = 0; char q = 0;
                for (char ch : v) {
                    if (q) { if (ch == q) q = 0; }
                    else if (ch == '"' || ch == '\'') q = ch;
                    else if (ch == '[') ++d;
                    else if (ch == ']') --d;
                }
                return d;
            };
            while (depth(val) > 0 && std::getline(in, line)) { ++lineno; val += " " + trim(strip_comment(line)); }
            if (val.empty() || depth(val) != 0) throw std::runtime_error(config_file + ":" + std::to_string(lineno) + ": malformed value for '" + key + "'");
            if (table != config_key) continue;
            bool known = false;
            for (auto &k : valid_keys) if (k == key) known = true;
            if (!known) throw std::runtime_error("Unknown configuration key '" + key + "'.");      // TOMLSanitize.h:114
            if (kv.count(key)) continue;                                                           // command line wins
            bool is_flag = false;
            for (auto &f : flags) if (f == key) is_flag = true;
            if (is_flag) {
                if (val == "true") kv[key] = "true";
                else if (val != "false") throw std::runtime_error("'" + key + "' must be a TOML value of type bool.");
                continue;
            }
            if (val.size() > 1 && (val.front() == '"' || val.front() == '\'')) val = val.substr(1, val.size() - 2);
            kv[key] = val;
        }
        if (!found) throw std::runtime_error("No configuration table named '" + config_key + "' was provided in the configuration file '" + config_file + "'.");
    }

    static Options parse(int argc, char **argv, const std::map<std::string, std::string> &short2long,
                         const std::vector<std::string> &flags = {})
    {
        Options o;
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            bool is_num = a.size() > 1 && a[0] == '-' && (isdigit((unsigned char)a[1]) || a[1] == '.');
            if (a.size() > 1 && a[0] == '-' && !is_num) {
                std::string key;
                if (a == "-c") key = "config";
                else if (a[1] == '-') key = a.substr(2);
                else {
                    auto it = short2long.find(a.substr(1));
                    if (it == short2long.end()) throw std::runtime_error("unrecognised option '" + a + "'");
                    key = it->second;
                }
                if (key == "config") {                                // multitoken: FILE KEY
                    if (i + 2 >= argc || argv[i + 1][0] == '-' || argv[i + 2][0] == '-')
                        throw std::runtime_error("Configuration must be supplied as file key pair.");   // TOMLSanitize.h:82
                    o.config_file = argv[++i];
                    o.config_key = argv[++i];
                    continue;
                }
                bool is_flag = false;
                for (auto &f : flags) if (f == key) is_flag = true;
                if (is_flag) { o.kv[key] = "true"; continue; }
                if (i + 1 >= argc) throw std::runtime_error("option '" + a + "' needs a value");
                o.kv[key] = argv[++i];
            } else {
                o.positional.push_back(a);
            }
        }
        return o;
    }
    bool has(const std::string &k) const { return kv.count(k) != 0; }
    double num(const std::string &k, double def, double lo, double hi) const
    {
        if (!has(k)) return def;
        char *end = nullptr;
        double v = strtod(kv.at(k).c_str(), &end);
        if (end == kv.at(k).c_str()) throw std::runtime_error("'" + k + "' must be numeric");
        if (v < lo || v > hi) throw std::runtime_error("'" + k + "' out of range");   // TOMLSanitize.h:218-277
        return v;
    }
    // "[a,b]" -> 2 numbers (TOMLSanitize.h:279-344 getArray<T,2>)
    bool arr2(const std::string &k, double &a, double &b) const
    {
        if (!has(k)) return false;
        const std::string &s = kv.at(k);
        if (sscanf(s.c_str(), " [ %lf , %lf ]", &a, &b) != 2) throw std::runtime_error("'" + k + "' must be a 2-element array, e.g. [0,256]");
        return true;
    }
    // "[h11,...,h33]" -> 9 numbers (getArray<double, 9>, HomographyTransform2D.cpp:48-58)
    bool arr9(const std::string &k, double *h) const
    {
        if (!has(k)) return false;
        const std::string &s = kv.at(k);
        char tail = 0;
        if (sscanf(s.c_str(), " [ %lf , %lf , %lf , %lf , %lf , %lf , %lf , %lf , %lf %c", h, h + 1, h + 2, h + 3, h + 4, h + 5,
                   h + 6, h + 7, h + 8, &tail) != 10 || tail != ']')
            throw std::runtime_error("'" + k + "' must be a 9-element array, [h11,h12,...,h33]");
        return true;
    }
};

}  // namespace oat
