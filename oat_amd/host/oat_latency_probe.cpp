// oat-latency-probe FRAME_SINK POSITION_SOURCE -f FILE --rows R --cols C [-C BGR|GREY] [-n N] [-r fps]
//
// End-to-end latency of a tracking component at the drop-in boundary: the time from a frame being POSTED to the shared
// memory node FRAME_SINK (what a camera component does, FrameServer.cpp / TestFrame.cpp:103-128) to its position
// token arriving on POSITION_SOURCE (what `oat posisock` / `oat record` see, PositionCout.cpp:53-67).  One thread serves
// frames like oat-frameserve-raw (free-running, or paced with -r), one reads position tokens like oat-posi-cout;
// tokens come back one per frame, in order, so token i belongs to frame i.  Prints one JSON line:
//   {"frames": N, "fps": .., "latency_us": {"p50": .., "p90": .., "p99": .., "max": .., "mean": ..}}
// Serving starts when a SOURCE has attached to FRAME_SINK, i.e. when the component under test is up.  (Not "when the
// position node is connected": a SOURCE that touches a node before its SINK is bound connects on the sink's FIRST
// token, Source.h:149-185 -- and the component posts its first position only after it was served a frame.)
// The first 10 tokens (model initialisation, first-launch costs) are left out of the statistics.
#include "component.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <fstream>
#include <thread>

using namespace oat;
using Clock = std::chrono::steady_clock;

int main(int argc, char **argv)
{
    try {
        Options o = Options::parse(argc, argv, {{"f", "file"}, {"n", "num-frames"}, {"r", "fps"}, {"C", "color"}, {"h", "help"}}, {"help"});
        if (o.has("help") || o.positional.size() != 2 || !o.has("file") || !o.has("rows") || !o.has("cols")) {
            std::cout << "Usage: oat-latency-probe FRAME_SINK POSITION_SOURCE -f FILE --rows R --cols C [-C BGR|GREY] [-n N] [-r fps]\n";
            return o.has("help") ? 0 : -1;
        }
        std::signal(SIGINT, sigHandler);
        const size_t rows = (size_t)o.num("rows", 0, 1, 1e5), cols = (size_t)o.num("cols", 0, 1, 1e5);
        const PixelColor col = (o.has("color") && o.kv["color"] == "GREY") ? PIX_GREY : PIX_BGR;
        const size_t fb = rows * cols * color_bytes(col);
        std::ifstream in(o.kv["file"], std::ios::binary);
        if (!in) throw std::runtime_error("cannot open " + o.kv["file"]);
        std::vector<uint8_t> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (buf.size() < fb) throw std::runtime_error("file is smaller than one frame");
        const size_t nfile = buf.size() / fb;
        const uint64_t n = (uint64_t)o.num("num-frames", 1000, 1, 1e9);
        const double fps = o.num("fps", 0.0, 0.0, 1e9);

        std::vector<Clock::time_point> posted(n), got(n);
        std::atomic<bool> connected{false};
        std::atomic<uint64_t> received{0};

        Sink<Frame> sink;
        sink.bind(o.positional[0], fb);
        Frame shared = sink.retrieve(rows, cols, color_cvtype(col), col);
        memcpy(shared.data(), buf.data(), fb);
        if (fps > 0) shared.sample().set_rate_hz(fps);

        std::thread reader([&] {
            try {
                Source<Position2D> src;
                src.touch(o.positional[1]);
                if (src.connect() != SourceState::CONNECTED) return;
                connected = true;
                for (uint64_t i = 0; i < n && !quit; ++i) {
                    if (src.wait() == NodeState::END) break;
                    got[i] = Clock::now();
                    src.post();
                    received = i + 1;
                }
            } catch (const std::exception &e) {
                std::cerr << "oat-latency-probe (reader): " << e.what() << std::endl;
                quit = 1;
            }
        });
        while (sink.source_ref_count() == 0 && !quit) std::this_thread::sleep_for(std::chrono::milliseconds(2));
        std::this_thread::sleep_for(std::chrono::milliseconds(300));      // it attaches to FRAME_SINK, creates its device context, binds its own sink
        const auto period = std::chrono::duration<double>(fps > 0 ? 1.0 / fps : 0.0);
        auto tick = Clock::now();
        const auto t_begin = tick;
        for (uint64_t i = 0; i < n && !quit; ++i) {
            sink.wait();
            if (nfile > 1) memcpy(shared.data(), buf.data() + (i % nfile) * fb, fb);
            shared.sample().incrementCount();
            posted[i] = Clock::now();
            sink.post();
            if (fps > 0) { tick += std::chrono::duration_cast<Clock::duration>(period); std::this_thread::sleep_until(tick); }
        }
        sink.wait();
        const auto give_up = Clock::now() + std::chrono::seconds(20);
        while (received < n && !quit && Clock::now() < give_up) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        const uint64_t m = received;
        const double wall = std::chrono::duration<double>(Clock::now() - t_begin).count();
        quit = 1;
        reader.join();
        std::vector<double> lat;
        double sum = 0;
        const uint64_t skip = m > 50 ? 10 : 0;
        for (uint64_t i = skip; i < m; ++i) { lat.push_back(std::chrono::duration<double, std::micro>(got[i] - posted[i]).count()); sum += lat.back(); }
        std::sort(lat.begin(), lat.end());
        auto pct = [&](double p) { return lat.empty() ? 0.0 : lat[std::min(lat.size() - 1, (size_t)(p * (double)lat.size()))]; };
        printf("{\"frames\": %llu, \"tokens\": %llu, \"fps\": %.1f, \"paced_fps\": %.1f, \"latency_us\": {\"p50\": %.1f, \"p90\": %.1f, \"p99\": %.1f, "
               "\"max\": %.1f, \"mean\": %.1f}}\n", (unsigned long long)n, (unsigned long long)m, (double)m / wall, fps, pct(0.50), pct(0.90),
               pct(0.99), lat.empty() ? 0.0 : lat.back(), lat.empty() ? 0.0 : sum / (double)lat.size());
        return m == n ? 0 : 1;
    } catch (const std::exception &e) {
        std::cerr << "oat-latency-probe: " << e.what() << std::endl;
        return -1;
    }
}
