// oat-posidet-hip TYPE SOURCE SINK [CONFIGURATION]
//   TYPE  hsv     replacement of `oat posidet hsv`    (src/positiondetector/HSVDetector.cpp)
//         thresh  replacement of `oat posidet thresh` (src/positiondetector/SimpleThreshold.cpp)
//         diff    replacement of `oat posidet diff`   (src/positiondetector/DifferenceDetector.cpp)
// Same positional arguments and option names as src/positiondetector/main.cpp:85-271.
#include "component.hpp"

#include <cfloat>

using namespace oat;

enum class Kind { HSV, THRESH, DIFF };

class GpuDetector : public PositionDetector {
public:
    GpuDetector(const std::string &src, const std::string &snk, Kind kind) : PositionDetector(src, snk), kind_(kind)
    {
        oatgpu_default_config(&cfg_);
        if (kind == Kind::HSV) { required_color_ = PIX_HSV; cfg_.erode = 0; cfg_.dilate = 10; }   // HSVDetector.cpp:42-46
        else { required_color_ = PIX_GREY; cfg_.erode = 0; cfg_.dilate = 0; }   // SimpleThreshold.cpp:42-46, DifferenceDetector.cpp:41-44
    }
    oatgpu_config cfg_;

protected:
    void configure_for(const FrameParams &p) override
    {
        cfg_.rows = (int)p.rows; cfg_.cols = (int)p.cols; cfg_.n_streams = 1;
        gpu_.create(cfg_);
    }
    // HSVDetector.cpp:142-173 / SimpleThreshold.cpp:114-134
    // Deferred (oatgpu_set_deferred): the call returns when the frame has left shared memory, PositionDetector::process posts
    // the source, and detect_finish() waits for the kernels and takes the position (PositionDetector.cpp:78-96).
    bool detect_from_shm(const Frame &frame, Position2D &) override
    {
        if (!deferred_on_) { gpu_.check(oatgpu_set_deferred(gpu_.ctx, 1)); deferred_on_ = true; }
        gpu_.check(kind_ == Kind::HSV ? oatgpu_detect_hsv(gpu_.ctx, 0, frame.data(), nullptr)
                   : kind_ == Kind::THRESH ? oatgpu_detect_thresh(gpu_.ctx, 0, frame.data(), nullptr)
                                           : oatgpu_detect_diff(gpu_.ctx, 0, frame.data(), nullptr));
        return true;
    }
    void detect_finish(Position2D &position) override
    {
        oatgpu_position r;
        gpu_.check(oatgpu_fetch_position(gpu_.ctx, &r));
        position.position_valid = r.valid != 0;                   // DetectorFunc.cpp:46,58-60
        if (r.valid) { position.position.x = r.x; position.position.y = r.y; }
    }
    void detectPosition(Frame &frame, Position2D &position) override
    {
        oatgpu_position r;
        gpu_.check(kind_ == Kind::HSV ? oatgpu_detect_hsv(gpu_.ctx, 0, frame.data(), &r)
                   : kind_ == Kind::THRESH ? oatgpu_detect_thresh(gpu_.ctx, 0, frame.data(), &r)
                                           : oatgpu_detect_diff(gpu_.ctx, 0, frame.data(), &r));
        position.position_valid = r.valid != 0;                   // DetectorFunc.cpp:46,58-60
        if (r.valid) { position.position.x = r.x; position.position.y = r.y; }
    }
    Kind kind_;
    bool deferred_on_{false};
    GpuCtx gpu_;
};

static void usage()
{
    std::cout << "Usage: oat-posidet-hip TYPE SOURCE SINK [CONFIGURATION]\nTYPE\n  hsv | thresh | diff\n"
                 "diff:   -d diff-threshold (default 10)  -b blur (default 2, <= 22)  -a [min,max] area\n"
                 "hsv:    -H/-S/-V [min,max] in [0,256]  -e erode  -d dilate (default 10)  -a [min,max] area\n"
                 "thresh: -T [min,max]  -e erode  -d dilate  -a [min,max] area\n"
                 "all:    --gpu-index N  HIP device ordinal (default 0)\n";
}

int main(int argc, char **argv)
{
    try {
        // -d is "dilate" for hsv/thresh and "diff-threshold" for diff (DifferenceDetector.cpp:49-50)
        const bool is_diff = argc > 1 && std::string(argv[1]) == "diff";
        Options o = Options::parse(argc, argv,
            {{"H", "h-thresh"}, {"S", "s-thresh"}, {"V", "v-thresh"}, {"T", "thresh"}, {"e", "erode"},
             {"d", is_diff ? "diff-threshold" : "dilate"}, {"b", "blur"}, {"a", "area"}, {"t", "tune"}, {"h", "help"}, {"v", "version"}},
            {"help", "version", "tune"});
        if (o.has("version")) { std::cout << "oat-posidet-hip (MI355X drop-in, liboatgpu ABI " << oatgpu_abi_version() << ")\n"; return 0; }
        if (o.has("help") || o.positional.size() != 3) { usage(); return o.has("help") ? 0 : -1; }
        const std::string type = o.positional[0];
        if (type != "hsv" && type != "thresh" && type != "diff") throw std::runtime_error("Selected TYPE is invalid.");
        // option names per TYPE: HSVDetector.cpp:49-75, SimpleThreshold.cpp:49-69, DifferenceDetector.cpp:41-62
        if (type == "hsv") o.apply_config({"h-thresh", "s-thresh", "v-thresh", "erode", "dilate", "area", "tune", "gpu-index"}, {"tune"});
        else if (type == "thresh") o.apply_config({"thresh", "erode", "dilate", "area", "tune", "gpu-index"}, {"tune"});
        else o.apply_config({"diff-threshold", "blur", "area", "tune", "gpu-index"}, {"tune"});
        if (o.has("tune")) throw std::runtime_error("--tune needs a GUI and is not available in the hip detector");
        auto d = std::make_unique<GpuDetector>(o.positional[1], o.positional[2],
                                               type == "hsv" ? Kind::HSV : type == "thresh" ? Kind::THRESH : Kind::DIFF);
        d->cfg_.device = (int)o.num("gpu-index", 0, 0, 64);
        double a, b;
        auto range = [](double x, double y, const char *what) {
            if (x < 0 || x > 256 || y < 0 || y > 256) throw std::runtime_error(std::string("Values of ") + what + " should be between 0 and 256.");
        };
        if (type == "hsv") {
            if (o.arr2("h-thresh", a, b)) { range(a, b, "h-thresh"); d->cfg_.h_lo = (int)a; d->cfg_.h_hi = (int)b; }
            if (o.arr2("s-thresh", a, b)) { range(a, b, "s-thresh"); d->cfg_.s_lo = (int)a; d->cfg_.s_hi = (int)b; }
            if (o.arr2("v-thresh", a, b)) { range(a, b, "v-thresh"); d->cfg_.v_lo = (int)a; d->cfg_.v_hi = (int)b; }
        } else if (o.arr2("thresh", a, b)) { range(a, b, "thresh"); d->cfg_.h_lo = (int)a; d->cfg_.h_hi = (int)b; }
        if (o.has("diff-threshold")) d->cfg_.diff_threshold = (int)o.num("diff-threshold", 10, 0, 1e6);
        if (o.has("blur")) d->cfg_.blur = (int)o.num("blur", 2, 0, 1e6);
        if (o.has("erode")) d->cfg_.erode = (int)o.num("erode", 0, 0, 1e6);
        if (o.has("dilate")) d->cfg_.dilate = (int)o.num("dilate", 0, 0, 1e6);
        if (o.arr2("area", a, b)) {
            if (a >= b) throw std::runtime_error("Max area should be larger than min area.");   // HSVDetector.cpp:135
            d->cfg_.min_area = a; d->cfg_.max_area = b;
        }
        return d->run();
    } catch (const std::exception &e) {
        std::cerr << "oat-posidet-hip: " << e.what() << std::endl;
        return -1;
    }
}
