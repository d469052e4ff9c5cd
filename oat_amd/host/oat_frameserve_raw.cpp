// oat-frameserve-raw SINK -f FILE --rows R --cols C [-C BGR|GREY|HSV] [-n N] [-r fps]
// Counterpart of `oat frameserve test` (src/frameserver/TestFrame.cpp:59-128) for pipelines without
// OpenCV's imread: FILE is a headerless packed dump of one or more frames (rows*cols*channels bytes
// each).  With one frame it is published N times WITHOUT re-copying, exactly like TestFrame::process
// (:103-128); with several, frame i is copied in before the i-th post.  The Sample rate is set once
// and the sample count incremented exactly once per published frame (:98,:114).
#include "component.hpp"

#include <chrono>
#include <fstream>
#include <thread>

using namespace oat;

int main(int argc, char **argv)
{
    try {
        Options o = Options::parse(argc, argv, {{"f", "file"}, {"n", "num-frames"}, {"r", "fps"}, {"C", "color"}, {"h", "help"}}, {"help"});
        if (o.has("help") || o.positional.size() != 1 || !o.has("file") || !o.has("rows") || !o.has("cols")) {
            std::cout << "Usage: oat-frameserve-raw SINK -f FILE --rows R --cols C [-C BGR|GREY|HSV] [-n N] [-r fps]\n";
            return o.has("help") ? 0 : -1;
        }
        std::signal(SIGINT, sigHandler);
        const size_t rows = (size_t)o.num("rows", 0, 1, 1e5), cols = (size_t)o.num("cols", 0, 1, 1e5);
        PixelColor col = PIX_BGR;
        if (o.has("color")) col = o.kv["color"] == "GREY" ? PIX_GREY : o.kv["color"] == "HSV" ? PIX_HSV : PIX_BGR;
        const size_t fb = rows * cols * color_bytes(col);
        std::ifstream in(o.kv["file"], std::ios::binary);
        if (!in) throw std::runtime_error("cannot open " + o.kv["file"]);
        std::vector<uint8_t> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (buf.size() < fb) throw std::runtime_error("file is smaller than one frame");
        const size_t nfile = buf.size() / fb;
        const uint64_t n = (uint64_t)o.num("num-frames", (double)nfile, 1, 1e15);
        const double fps = o.num("fps", 0.0, 0.0, 1e9);

        Sink<Frame> sink;
        sink.bind(o.positional[0], fb);
        Frame shared = sink.retrieve(rows, cols, color_cvtype(col), col);
        memcpy(shared.data(), buf.data(), fb);
        if (fps > 0) shared.sample().set_rate_hz(fps);
        const auto period = std::chrono::duration<double>(fps > 0 ? 1.0 / fps : 0.0);
        auto tick = std::chrono::steady_clock::now();
        for (uint64_t i = 0; i < n && !quit; ++i) {
            sink.wait();
            if (nfile > 1) memcpy(shared.data(), buf.data() + (i % nfile) * fb, fb);
            shared.sample().incrementCount();
            sink.post();
            if (fps > 0) { tick += std::chrono::duration_cast<std::chrono::steady_clock::duration>(period); std::this_thread::sleep_until(tick); }
        }
        sink.wait();     // let the last token be consumed before END is signalled by ~Sink
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "oat-frameserve-raw: " << e.what() << std::endl;
        return -1;
    }
}
