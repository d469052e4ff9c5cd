// oat-framefilt-hip TYPE SOURCE SINK [CONFIGURATION]
//   TYPE  mog   MI355X replacement of `oat framefilt mog`  (src/framefilter/BackgroundSubtractorMOG.cpp)
//         col   MI355X replacement of `oat framefilt col`  (src/framefilter/ColorConvert.cpp), BGR->HSV only
//         bsub  MI355X replacement of `oat framefilt bsub` (src/framefilter/BackgroundSubtractor.cpp)
//         thresh MI355X replacement of `oat framefilt thresh` (src/framefilter/Threshold.cpp)
// Drop-in: same positional arguments, same option names (src/framefilter/main.cpp:91-296).
#include "component.hpp"
#include <unistd.h>

using namespace oat;

class BackgroundSubtractorMOG : public FrameFilter {
public:
    using FrameFilter::FrameFilter;
    double learning_coeff_{0.0};    // BackgroundSubtractorMOG.h:76
    int gpu_index_{0};
    std::string model_file_;        // --model-file: resume the MOG2 model from / checkpoint it to this file
    ~BackgroundSubtractorMOG() override
    {
        if (!model_file_.empty() && gpu_.ctx && oatgpu_mog_save(gpu_.ctx, 0, model_file_.c_str()) != OATGPU_OK)
            std::cerr << name() << ": " << oatgpu_last_error(gpu_.ctx) << std::endl;
    }

protected:
    void configure_for(const FrameParams &p) override
    {
        if (p.color != PIX_BGR) throw std::runtime_error("framefilt mog (hip) needs BGR frames");
        oatgpu_config cfg;
        oatgpu_default_config(&cfg);
        cfg.device = gpu_index_; cfg.rows = (int)p.rows; cfg.cols = (int)p.cols; cfg.n_streams = 1;
        gpu_.create(cfg);
        if (!model_file_.empty() && access(model_file_.c_str(), R_OK) == 0)
            gpu_.check(oatgpu_mog_load(gpu_.ctx, 0, model_file_.c_str()));
    }
    // BackgroundSubtractorMOG.cpp:114-127 (CPU-branch semantics: MOG2, shadows kept)
    void filter(Frame &frame) override
    {
        gpu_.check(oatgpu_mog_filter(gpu_.ctx, 0, frame.data(), frame.data(), learning_coeff_));
    }
    bool filter_from_shm(const Frame &in, Frame &out) override
    {
        gpu_.check(oatgpu_mog_filter(gpu_.ctx, 0, in.data(), out.data(), learning_coeff_));
        return true;
    }
    GpuCtx gpu_;
};

class ColorConvert : public FrameFilter {
public:
    using FrameFilter::FrameFilter;
    PixelColor color_{PIX_HSV};

protected:
    PixelColor sink_color(PixelColor in) const override
    {
        if (in != PIX_BGR || color_ != PIX_HSV)   // color_conv_table, Color.h:46-51: only BGR->HSV is on the hot path
            throw std::runtime_error("framefilt col (hip) converts BGR to HSV only");
        return PIX_HSV;
    }
    void configure_for(const FrameParams &p) override
    {
        oatgpu_config cfg;
        oatgpu_default_config(&cfg);
        cfg.rows = (int)p.rows; cfg.cols = (int)p.cols;
        gpu_.create(cfg);
    }
    // ColorConvert.cpp:101-107
    void filter(Frame &frame) override
    {
        gpu_.check(oatgpu_bgr2hsv(gpu_.ctx, frame.data(), frame.data()));
        frame.set_color(PIX_HSV);
    }
    bool filter_from_shm(const Frame &in, Frame &out) override
    {
        gpu_.check(oatgpu_bgr2hsv(gpu_.ctx, in.data(), out.data()));
        out.set_color(PIX_HSV);
        return true;
    }
    GpuCtx gpu_;
};

class BackgroundSubtractor : public FrameFilter {
public:
    using FrameFilter::FrameFilter;
    double alpha_{0.0};             // BackgroundSubtractor.h

protected:
    void configure_for(const FrameParams &p) override
    {
        oatgpu_config cfg;
        oatgpu_default_config(&cfg);
        cfg.rows = (int)p.rows; cfg.cols = (int)p.cols; cfg.channels = color_bytes(p.color);
        gpu_.create(cfg);
    }
    // BackgroundSubtractor.cpp:87-100 (a -f background image needs imread and is not supported)
    void filter(Frame &frame) override
    {
        gpu_.check(oatgpu_bsub_filter(gpu_.ctx, 0, frame.data(), frame.data(), alpha_));
    }
    bool filter_from_shm(const Frame &in, Frame &out) override
    {
        gpu_.check(oatgpu_bsub_filter(gpu_.ctx, 0, in.data(), out.data(), alpha_));
        return true;
    }
    GpuCtx gpu_;
};

class Threshold : public FrameFilter {
public:
    using FrameFilter::FrameFilter;
    int i_min_{0}, i_max_{256};     // Threshold.h

protected:
    void configure_for(const FrameParams &p) override
    {
        oatgpu_config cfg;
        oatgpu_default_config(&cfg);
        cfg.rows = (int)p.rows; cfg.cols = (int)p.cols; cfg.channels = color_bytes(p.color);
        gpu_.create(cfg);
    }
    // Threshold.cpp:67-81
    void filter(Frame &frame) override
    {
        gpu_.check(oatgpu_thresh_filter(gpu_.ctx, frame.data(), frame.data(), i_min_, i_max_));
    }
    bool filter_from_shm(const Frame &in, Frame &out) override
    {
        gpu_.check(oatgpu_thresh_filter(gpu_.ctx, in.data(), out.data(), i_min_, i_max_));
        return true;
    }
    GpuCtx gpu_;
};

static void usage()
{
    std::cout << "Usage: oat-framefilt-hip TYPE SOURCE SINK [CONFIGURATION]\n"
                 "TYPE\n  mog: MOG2 background segmentation on an MI355X\n  col: BGR -> HSV colour conversion on an MI355X\n"
                 "mog:  -a, --adaptation-coeff  0..1, default 0 (no adaptation)\n      --gpu-index          HIP device ordinal\n"
                 "      --model-file FILE    resume the background model from FILE if it exists; checkpoint it there on exit\n"
                 "col:  -C, --color             HSV\n"
                 "bsub: -a, --adaptation-coeff  0..1, default 0 (static background = first frame)\n"
                 "thresh: -I, --intensity       [min,max] in [0,256]\n";
}

int main(int argc, char **argv)
{
    try {
        Options o = Options::parse(argc, argv, {{"a", "adaptation-coeff"}, {"C", "color"}, {"I", "intensity"}, {"h", "help"}, {"v", "version"}}, {"help", "version"});
        if (o.has("version")) { std::cout << "oat-framefilt-hip (MI355X drop-in, liboatgpu ABI " << oatgpu_abi_version() << ")\n"; return 0; }
        if (o.has("help") || o.positional.size() != 3) { usage(); return o.has("help") ? 0 : -1; }
        const std::string type = o.positional[0];
        // option names per TYPE: BackgroundSubtractorMOG.cpp:51-67, ColorConvert.cpp:45-63,
        // BackgroundSubtractor.cpp:42-60, Threshold.cpp:40-54
        if (type == "mog") o.apply_config({"adaptation-coeff", "gpu-index", "model-file"});
        else if (type == "col") o.apply_config({"color"});
        else if (type == "bsub") o.apply_config({"adaptation-coeff", "background"});
        else if (type == "thresh") o.apply_config({"intensity"});
        std::unique_ptr<Component> comp;
        if (type == "mog") {
            auto f = std::make_unique<BackgroundSubtractorMOG>(o.positional[1], o.positional[2]);
            f->learning_coeff_ = o.num("adaptation-coeff", 0.0, 0.0, 1.0);     // BackgroundSubtractorMOG.cpp:86-88
            f->gpu_index_ = (int)o.num("gpu-index", 0, 0, 64);
            if (o.has("model-file")) f->model_file_ = o.kv["model-file"];
            comp = std::move(f);
        } else if (type == "col") {
            auto f = std::make_unique<ColorConvert>(o.positional[1], o.positional[2]);
            if (o.has("color") && o.kv["color"] != "HSV") throw std::runtime_error("only -C HSV is supported");
            comp = std::move(f);
        } else if (type == "bsub") {
            auto f = std::make_unique<BackgroundSubtractor>(o.positional[1], o.positional[2]);
            f->alpha_ = o.num("adaptation-coeff", 0.0, 0.0, 1.0);                // BackgroundSubtractor.cpp:75
            if (o.has("background")) throw std::runtime_error("--background needs an image reader: not supported");
            comp = std::move(f);
        } else if (type == "thresh") {
            auto f = std::make_unique<Threshold>(o.positional[1], o.positional[2]);
            double a, b;
            if (o.arr2("intensity", a, b)) {
                if (a < 0 || a > 256 || b < 0 || b > 256)
                    throw std::runtime_error("Values of intensity should be between 0 and 256.");   // Threshold.cpp:62-63
                f->i_min_ = (int)a; f->i_max_ = (int)b;
            }
            comp = std::move(f);
        } else {
            throw std::runtime_error("Selected TYPE is invalid.");
        }
        return comp->run();
    } catch (const std::exception &e) {
        std::cerr << "oat-framefilt-hip: " << e.what() << std::endl;
        return -1;
    }
}
