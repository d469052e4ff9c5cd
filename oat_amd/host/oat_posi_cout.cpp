// oat-posi-cout SOURCE [-n N] [--npy FILE]
// Minimal `oat posisock std` (src/positionsocket/PositionCout.cpp:53-67): one JSON object per position
// token, serialised like oat::serializePosition (lib/datatypes/Position2D.h:170-233).  With --npy the
// tokens are ALSO written as the reference recorder's packed records (Position2D.cpp:24-96) into a
// numpy .npy file (structured dtype Position2D::NPY_DTYPE, shape patched at the end).
#include "component.hpp"

#include <fstream>

using namespace oat;

static std::string npy_header(uint64_t n)
{
    char shape[32];
    snprintf(shape, sizeof shape, "(%020llu,)", (unsigned long long)n);     // fixed width: patched in place
    std::string d = std::string("{'descr': ") + npy_dtype() + ", 'fortran_order': False, 'shape': " + shape + ", }";
    size_t total = 10 + d.size() + 1;
    d += std::string((64 - total % 64) % 64, ' ') + "\n";
    std::string h("\x93NUMPY\x01\x00", 8);
    uint16_t len = (uint16_t)d.size();
    h.append((const char *)&len, 2);
    return h + d;
}

int main(int argc, char **argv)
{
    try {
        Options o = Options::parse(argc, argv, {{"n", "num"}, {"h", "help"}}, {"help"});
        if (o.has("help") || o.positional.size() != 1) { std::cout << "Usage: oat-posi-cout SOURCE [-n N] [--npy FILE]\n"; return o.has("help") ? 0 : -1; }
        std::signal(SIGINT, sigHandler);
        const uint64_t n = (uint64_t)o.num("num", 1e18, 1, 1e18);
        std::ofstream npy;
        if (o.has("npy")) {
            npy.open(o.kv["npy"], std::ios::binary);
            if (!npy) throw std::runtime_error("cannot open " + o.kv["npy"]);
            npy << npy_header(0);
        }
        Source<Position2D> src;
        src.touch(o.positional[0]);
        if (src.connect() != SourceState::CONNECTED) return 0;
        uint64_t written = 0;
        for (uint64_t i = 0; i < n && !quit; ++i) {
            if (src.wait() == NodeState::END) break;
            Position2D p = src.clone();
            src.post();
            puts(serializePosition(p).c_str());
            fflush(stdout);
            if (npy.is_open()) { auto r = packPosition(p); npy.write(r.data(), r.size()); ++written; }
        }
        if (npy.is_open()) { npy.seekp(0); npy << npy_header(written); }
        return 0;
    } catch (const std::exception &e) {
        std::cerr << "oat-posi-cout: " << e.what() << std::endl;
        return -1;
    }
}
