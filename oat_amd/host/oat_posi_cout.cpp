// oat-posi-cout SOURCE [-n N]
// Minimal `oat posisock std` (src/positionsocket/PositionCout.cpp:53-67): prints one JSON object per
// position token with the reference's key set (lib/datatypes/Position2D.h:170-233): tick, usec, unit,
// pos_ok, pos_xy, vel_ok, head_ok, reg_ok.
#include "component.hpp"

using namespace oat;

int main(int argc, char **argv)
{
    try {
        Options o = Options::parse(argc, argv, {{"n", "num"}, {"h", "help"}}, {"help"});
        if (o.has("help") || o.positional.size() != 1) { std::cout << "Usage: oat-posi-cout SOURCE [-n N]\n"; return o.has("help") ? 0 : -1; }
        std::signal(SIGINT, sigHandler);
        const uint64_t n = (uint64_t)o.num("num", 1e18, 1, 1e18);
        Source<Position2D> src;
        src.touch(o.positional[0]);
        if (src.connect() != SourceState::CONNECTED) return 0;
        for (uint64_t i = 0; i < n && !quit; ++i) {
            if (src.wait() == NodeState::END) break;
            Position2D p = src.clone();
            src.post();
            printf("{\"tick\":%llu,\"usec\":%lld,\"unit\":%d,\"pos_ok\":%s", (unsigned long long)p.sample().count(),
                   (long long)p.sample().microseconds(), (int)p.unit_of_length_, p.position_valid ? "true" : "false");
            if (p.position_valid) printf(",\"pos_xy\":[%.5f,%.5f]", p.position.x, p.position.y);
            printf(",\"vel_ok\":%s,\"head_ok\":%s,\"reg_ok\":%s}\n", p.velocity_valid ? "true" : "false",
                   p.heading_valid ? "true" : "false", p.region_valid ? "true" : "false");
            fflush(stdout);
        }
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "oat-posi-cout: " << e.what() << std::endl;
        return -1;
    }
}
