// oat-track-hip SOURCE SINK [CONFIGURATION]
// The whole chain  framefilt mog -> framefilt col -C HSV -> posidet hsv  in ONE process and ONE
// fused device pass per frame: BGR oat::Frame in, oat::Position2D out (one token out per token
// in, carrying the frame's Sample -- PositionDetector.cpp:80).  Options are the union of the
// three stock components' (-a is mog's adaptation coefficient; the detector's area is --area).
#include "component.hpp"
#include <unistd.h>

using namespace oat;

class FusedTracker : public PositionDetector {
public:
    FusedTracker(const std::string &src, const std::string &snk) : PositionDetector(src, snk)
    {
        oatgpu_default_config(&cfg_);
        required_color_ = PIX_BGR;
        name_ = "track[" + src + "->" + snk + "]";
    }
    oatgpu_config cfg_;
    double learning_coeff_{0.0};
    bool kalman_{false};            // --kalman: `posifilt kalman` fused behind the detector
    double dt_{0.02}, timeout_{0.0}, sig_accel_{5.0}, sig_noise_{0.0};   // KalmanFilter2D.h:56-60
    std::string model_file_;        // --model-file: resume the MOG2 model from / checkpoint it to this file
    ~FusedTracker() override
    {
        if (!model_file_.empty() && gpu_.ctx && oatgpu_mog_save(gpu_.ctx, 0, model_file_.c_str()) != OATGPU_OK)
            std::cerr << name() << ": " << oatgpu_last_error(gpu_.ctx) << std::endl;
    }

protected:
    void configure_for(const FrameParams &p) override
    {
        cfg_.rows = (int)p.rows; cfg_.cols = (int)p.cols; cfg_.n_streams = 1;
        gpu_.create(cfg_);
        if (!model_file_.empty() && access(model_file_.c_str(), R_OK) == 0)
            gpu_.check(oatgpu_mog_load(gpu_.ctx, 0, model_file_.c_str()));
        if (kalman_) gpu_.check(oatgpu_set_kalman(gpu_.ctx, 1, dt_, timeout_, sig_accel_, sig_noise_));
    }
    bool detect_from_shm(const Frame &frame, Position2D &position) override
    {
        detectPosition(const_cast<Frame &>(frame), position);      // reads only
        return true;
    }
    void detectPosition(Frame &frame, Position2D &position) override
    {
        const uint8_t *f = frame.data();
        oatgpu_position r;
        gpu_.check(oatgpu_track_batch(gpu_.ctx, &f, 1, learning_coeff_, &r));
        position.position_valid = r.valid != 0;
        if (kalman_) {                                          // KalmanFilter2D.cpp:123-137
            position.position.x = r.x; position.position.y = r.y;
            position.velocity.x = r.vx; position.velocity.y = r.vy;
            position.velocity_valid = r.velocity_valid != 0;
        } else if (r.valid) {
            position.position.x = r.x; position.position.y = r.y;
        }
    }
    GpuCtx gpu_;
};

int main(int argc, char **argv)
{
    try {
        Options o = Options::parse(argc, argv,
            {{"a", "adaptation-coeff"}, {"H", "h-thresh"}, {"S", "s-thresh"}, {"V", "v-thresh"}, {"e", "erode"},
             {"d", "dilate"}, {"T", "timeout"}, {"n", "sigma-noise"}, {"h", "help"}, {"v", "version"}}, {"help", "version", "kalman"});
        if (o.has("version")) { std::cout << "oat-track-hip (MI355X drop-in, liboatgpu ABI " << oatgpu_abi_version() << ")\n"; return 0; }
        if (o.has("help") || o.positional.size() != 2) {
            std::cout << "Usage: oat-track-hip SOURCE SINK [-a coeff] [-H [lo,hi]] [-S ..] [-V ..] [-e n] [-d n] [--area [min,max]] [--model-file FILE]\n"
                         "       [--kalman [--dt s] [-T|--timeout s] [--sigma-accel a] [-n|--sigma-noise n]]   (posifilt kalman fused in)\n";
            return o.has("help") ? 0 : -1;
        }
        o.apply_config({"adaptation-coeff", "h-thresh", "s-thresh", "v-thresh", "erode", "dilate", "area", "model-file",
                        "kalman", "dt", "timeout", "sigma-accel", "sigma-noise"}, {"kalman"});
        auto t = std::make_unique<FusedTracker>(o.positional[0], o.positional[1]);
        t->learning_coeff_ = o.num("adaptation-coeff", 0.0, 0.0, 1.0);
        double a, b;
        if (o.arr2("h-thresh", a, b)) { t->cfg_.h_lo = (int)a; t->cfg_.h_hi = (int)b; }
        if (o.arr2("s-thresh", a, b)) { t->cfg_.s_lo = (int)a; t->cfg_.s_hi = (int)b; }
        if (o.arr2("v-thresh", a, b)) { t->cfg_.v_lo = (int)a; t->cfg_.v_hi = (int)b; }
        if (o.has("erode")) t->cfg_.erode = (int)o.num("erode", 0, 0, 1e6);
        if (o.has("dilate")) t->cfg_.dilate = (int)o.num("dilate", 0, 0, 1e6);
        if (o.arr2("area", a, b)) { t->cfg_.min_area = a; t->cfg_.max_area = b; }
        if (o.has("model-file")) t->model_file_ = o.kv["model-file"];
        t->kalman_ = o.has("kalman");
        t->dt_ = o.num("dt", 0.02, 0, 1e9);                        // KalmanFilter2D.cpp:69-85 (lower bound 0)
        t->timeout_ = o.num("timeout", 0.0, 0, 1e18);
        t->sig_accel_ = o.num("sigma-accel", 5.0, 0, 1e18);
        t->sig_noise_ = o.num("sigma-noise", 0.0, 0, 1e18);
        return t->run();
    } catch (const std::exception &e) {
        std::cerr << "oat-track-hip: " << e.what() << std::endl;
        return -1;
    }
}
