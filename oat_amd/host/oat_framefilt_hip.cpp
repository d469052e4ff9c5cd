// oat-framefilt-hip TYPE SOURCE SINK [CONFIGURATION]
//   TYPE  mog   MI355X replacement of `oat framefilt mog`  (src/framefilter/BackgroundSubtractorMOG.cpp)
//         col   MI355X replacement of `oat framefilt col`  (src/framefilter/ColorConvert.cpp): every pair of oat::color_conv_table
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
        // cv::BackgroundSubtractorMOG2::apply takes any 8U frame of one or three channels and the reference
        // passes on whatever colour its source carries (FrameFilter.cpp:37-57): GREY frames of a
        // `framefilt col -C GREY` or a mono camera run the one-channel model
        oatgpu_config cfg;
        oatgpu_default_config(&cfg);
        cfg.device = gpu_index_; cfg.rows = (int)p.rows; cfg.cols = (int)p.cols; cfg.n_streams = 1;
        cfg.channels = color_bytes(p.color);
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
    oatgpu_ctx *deferred_ctx() override { return gpu_.ctx; }
    GpuCtx gpu_;
};

class ColorConvert : public FrameFilter {
public:
    using FrameFilter::FrameFilter;
    PixelColor color_{PIX_BGR};     // ColorConvert.h: color_ {PIX_BGR}; --color is required (ColorConvert.cpp:58-63)
    int gpu_index_{0};

protected:
    // oat::color_conv_code (Color.h:88-95) + ColorConvert::connectToNode (ColorConvert.cpp:76-85): the table
    // lives behind oatgpu_cvt_color; here only its two refusals, raised at connect time as the reference does
    PixelColor sink_color(PixelColor in) const override
    {
        static const int table[4][4] = {{-1, -1, 0, -2}, {-1, -1, 0, -2}, {0, 0, -1, 0}, {-2, -2, 0, -1}};
        const int code = table[in][color_];
        if (code == -2) throw std::runtime_error("Requested color conversion is not possible.");
        if (code == -1)
            throw std::runtime_error(std::string("Nothing to be done for ") + color_str(in) + " to " + color_str(color_) + " conversion.");
        return color_;
    }
    void configure_for(const FrameParams &p) override
    {
        oatgpu_config cfg;
        oatgpu_default_config(&cfg);
        cfg.device = gpu_index_; cfg.rows = (int)p.rows; cfg.cols = (int)p.cols;
        gpu_.create(cfg);
        from_ = p.color;
    }
    // ColorConvert.cpp:101-107.  Not in place: the two sides may differ in bytes per pixel.
    void filter(Frame &frame) override
    {
        Frame out(frame.rows(), frame.cols(), color_);
        gpu_.check(oatgpu_cvt_color(gpu_.ctx, (int)from_, (int)color_, frame.data(), out.data()));
        out.sample() = frame.sample();
        frame = std::move(out);
    }
    bool filter_from_shm(const Frame &in, Frame &out) override
    {
        gpu_.check(oatgpu_cvt_color(gpu_.ctx, (int)in.color(), (int)color_, in.data(), out.data()));
        out.set_color(color_);
        return true;
    }
    oatgpu_ctx *deferred_ctx() override { return gpu_.ctx; }
    GpuCtx gpu_;
    PixelColor from_{PIX_BGR};
};

class BackgroundSubtractor : public FrameFilter {
public:
    using FrameFilter::FrameFilter;
    double alpha_{0.0};             // BackgroundSubtractor.h
    int gpu_index_{0};
    std::string background_file_;   // -f: PGM/PPM instead of cv::imread's formats

protected:
    void configure_for(const FrameParams &p) override
    {
        oatgpu_config cfg;
        oatgpu_default_config(&cfg);
        cfg.device = gpu_index_; cfg.rows = (int)p.rows; cfg.cols = (int)p.cols; cfg.channels = color_bytes(p.color);
        gpu_.create(cfg);
        if (!background_file_.empty()) {                 // BackgroundSubtractor.cpp:63-71: imread(.., COLOR)
            PnmImage im = read_pnm(background_file_);
            if (im.channels == 1 && cfg.channels == 3) {    // IMREAD_COLOR turns a grey file into 3 equal channels
                std::vector<uint8_t> c3(im.px.size() * 3);
                for (size_t i = 0; i < im.px.size(); ++i) c3[3 * i] = c3[3 * i + 1] = c3[3 * i + 2] = im.px[i];
                im.px.swap(c3);
                im.channels = 3;
            }
            if (im.rows != p.rows || im.cols != p.cols || im.channels != cfg.channels)
                throw std::runtime_error("background image and SOURCE frames differ in size or channels");   // cv::subtract would throw
            gpu_.check(oatgpu_bsub_set_background(gpu_.ctx, 0, im.px.data()));
        }
    }
    // BackgroundSubtractor.cpp:87-100
    void filter(Frame &frame) override
    {
        gpu_.check(oatgpu_bsub_filter(gpu_.ctx, 0, frame.data(), frame.data(), alpha_));
    }
    bool filter_from_shm(const Frame &in, Frame &out) override
    {
        gpu_.check(oatgpu_bsub_filter(gpu_.ctx, 0, in.data(), out.data(), alpha_));
        return true;
    }
    oatgpu_ctx *deferred_ctx() override { return gpu_.ctx; }
    GpuCtx gpu_;
};

// src/framefilter/FrameMasker.cpp: frame.setTo(0, roi_mask == 0); without -f the frame passes unchanged
class FrameMasker : public FrameFilter {
public:
    using FrameFilter::FrameFilter;
    int gpu_index_{0};
    std::string mask_file_;

protected:
    void configure_for(const FrameParams &p) override
    {
        oatgpu_config cfg;
        oatgpu_default_config(&cfg);
        cfg.device = gpu_index_; cfg.rows = (int)p.rows; cfg.cols = (int)p.cols; cfg.channels = color_bytes(p.color);
        if (cfg.channels != 1 && cfg.channels != 3) throw std::runtime_error("framefilt mask (hip) needs 1- or 3-byte pixels");
        gpu_.create(cfg);
        if (!mask_file_.empty()) {
            const GreyImage m = read_pnm_grey(mask_file_);            // FrameMasker.cpp:52-60
            if (m.rows != p.rows || m.cols != p.cols)
                throw std::runtime_error("Mask image and frame source image do not have equal sizes");
            gpu_.check(oatgpu_set_roi_mask(gpu_.ctx, 0, m.px.data()));
        }
    }
    void filter(Frame &frame) override { gpu_.check(oatgpu_mask_filter(gpu_.ctx, 0, frame.data(), frame.data())); }
    bool filter_from_shm(const Frame &in, Frame &out) override
    {
        gpu_.check(oatgpu_mask_filter(gpu_.ctx, 0, in.data(), out.data()));
        return true;
    }
    oatgpu_ctx *deferred_ctx() override { return gpu_.ctx; }
    GpuCtx gpu_;
};

class Threshold : public FrameFilter {
public:
    using FrameFilter::FrameFilter;
    int i_min_{0}, i_max_{256};     // Threshold.h
    int gpu_index_{0};

protected:
    void configure_for(const FrameParams &p) override
    {
        oatgpu_config cfg;
        oatgpu_default_config(&cfg);
        cfg.device = gpu_index_; cfg.rows = (int)p.rows; cfg.cols = (int)p.cols; cfg.channels = color_bytes(p.color);
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
    oatgpu_ctx *deferred_ctx() override { return gpu_.ctx; }
    GpuCtx gpu_;
};

static void usage()
{
    std::cout << "Usage: oat-framefilt-hip TYPE SOURCE SINK [CONFIGURATION]\n"
                 "TYPE\n  mog: MOG2 background segmentation on an MI355X\n  col: colour conversion (BGR -> HSV | GREY, GREY -> BGR, HSV -> BGR) on an MI355X\n"
                 "  bsub | thresh | mask: the other per-pixel filters of oat-framefilt\n"
                 "all:  --gpu-index N           HIP device ordinal (default 0)\n"
                 "mog:  -a, --adaptation-coeff  0..1, default 0 (no adaptation)\n"
                 "      --model-file FILE    resume the background model from FILE if it exists; checkpoint it there on exit\n"
                 "col:  -C, --color             colour to convert to: GREY | BGR | HSV (required)\n"
                 "bsub: -a, --adaptation-coeff  0..1, default 0 (static background = first frame)\n"
                 "      -f, --background FILE   PGM/PPM background image instead of the first frame\n"
                 "mask: -f, --mask FILE         PGM/PPM: pixels where it is 0 are set to 0\n"
                 "thresh: -I, --intensity       [min,max] in [0,256]\n";
}

int main(int argc, char **argv)
{
    try {
        // -f is "background" for bsub and "mask" for mask (BackgroundSubtractor.cpp:52, FrameMasker.cpp:45)
        const bool is_mask = argc > 1 && std::string(argv[1]) == "mask";
        Options o = Options::parse(argc, argv, {{"a", "adaptation-coeff"}, {"C", "color"}, {"I", "intensity"},
                                                {"f", is_mask ? "mask" : "background"}, {"h", "help"}, {"v", "version"}}, {"help", "version"});
        if (o.has("version")) { std::cout << "oat-framefilt-hip (MI355X drop-in, liboatgpu ABI " << oatgpu_abi_version() << ")\n"; return 0; }
        if (o.has("help") || o.positional.size() != 3) { usage(); return o.has("help") ? 0 : -1; }
        const std::string type = o.positional[0];
        // option names per TYPE: BackgroundSubtractorMOG.cpp:51-67, ColorConvert.cpp:45-63,
        // BackgroundSubtractor.cpp:42-60, Threshold.cpp:40-54
        if (type == "mog") o.apply_config({"adaptation-coeff", "gpu-index", "model-file"});
        else if (type == "col") o.apply_config({"color", "gpu-index"});
        else if (type == "bsub") o.apply_config({"adaptation-coeff", "background", "gpu-index"});
        else if (type == "thresh") o.apply_config({"intensity", "gpu-index"});
        else if (type == "mask") o.apply_config({"mask", "gpu-index"});
        const int gpu_index = (int)o.num("gpu-index", 0, 0, 64);
        std::unique_ptr<Component> comp;
        if (type == "mog") {
            auto f = std::make_unique<BackgroundSubtractorMOG>(o.positional[1], o.positional[2]);
            f->learning_coeff_ = o.num("adaptation-coeff", 0.0, 0.0, 1.0);     // BackgroundSubtractorMOG.cpp:86-88
            f->gpu_index_ = gpu_index;
            if (o.has("model-file")) f->model_file_ = o.kv["model-file"];
            comp = std::move(f);
        } else if (type == "col") {
            auto f = std::make_unique<ColorConvert>(o.positional[1], o.positional[2]);
            if (!o.has("color")) throw std::runtime_error("Required configuration value 'color' was not specified.");   // TOMLSanitize.h:199-200 via ColorConvert.cpp:59-60
            f->color_ = str_color(o.kv["color"]);                                 // Color.h:65-77
            f->gpu_index_ = gpu_index;
            comp = std::move(f);
        } else if (type == "bsub") {
            auto f = std::make_unique<BackgroundSubtractor>(o.positional[1], o.positional[2]);
            f->alpha_ = o.num("adaptation-coeff", 0.0, 0.0, 1.0);                // BackgroundSubtractor.cpp:75
            if (o.has("background")) f->background_file_ = o.kv["background"];
            f->gpu_index_ = gpu_index;
            comp = std::move(f);
        } else if (type == "mask") {
            auto f = std::make_unique<FrameMasker>(o.positional[1], o.positional[2]);
            if (o.has("mask")) f->mask_file_ = o.kv["mask"];
            f->gpu_index_ = gpu_index;
            comp = std::move(f);
        } else if (type == "thresh") {
            auto f = std::make_unique<Threshold>(o.positional[1], o.positional[2]);
            double a, b;
            if (o.arr2("intensity", a, b)) {
                if (a < 0 || a > 256 || b < 0 || b > 256)
                    throw std::runtime_error("Values of intensity should be between 0 and 256.");   // Threshold.cpp:62-63
                f->i_min_ = (int)a; f->i_max_ = (int)b;
            }
            f->gpu_index_ = gpu_index;
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
