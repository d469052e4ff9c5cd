// test_shmemdf.cpp -- the reference's transport tests restated against this tree's POSIX-shm
// Node / Sink / Source (no Catch here, so a tiny CHECK harness):
//   test/shmemdf/Node_test.cpp:28-78, Sink_test.cpp:34-185, Source_test.cpp:34-218,
//   test/shmemdf/concurrency_test.cpp:79-527 (std::async sources vs sink, blocking asserted through
//   future.wait_for after short sleeps, all on one shm address).
#include "shmemdf.hpp"

#include <chrono>
#include <cstdio>
#include <future>
#include <thread>

namespace oat {
volatile sig_atomic_t quit = 0;
namespace detail {
template <> struct TypeName<int> { static const char *get() { return "i"; } };
template <> struct TypeName<float> { static const char *get() { return "f"; } };
}
}
using namespace oat;
using namespace std::chrono_literals;

static int g_fail = 0, g_checks = 0;
#define CHECK(cond) do { ++g_checks; if (!(cond)) { ++g_fail; printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } } while (0)
#define CHECK_THROWS(expr) do { ++g_checks; bool t_ = false; try { expr; } catch (const std::exception &) { t_ = true; } if (!t_) { ++g_fail; printf("FAIL %s:%d  expected throw: %s\n", __FILE__, __LINE__, #expr); } } while (0)
#define CHECK_NOTHROW(expr) do { ++g_checks; try { expr; } catch (const std::exception &e) { ++g_fail; printf("FAIL %s:%d  threw %s: %s\n", __FILE__, __LINE__, e.what(), #expr); } } while (0)

static std::string ADDR;
static void scrub() { remove_segment(ADDR + "_node"); remove_segment(ADDR + "_obj"); }
template <typename F> static bool ready(F &f, std::chrono::milliseconds d = 5ms) { return f.wait_for(d) == std::future_status::ready; }

static void node_tests()
{
    // Node_test.cpp:28-78
#if defined(OAT_SHMEM_BOOST)
    Node *n = new Node();                              // (stock Oat's Node is constructed in place by Boost; here on the heap)
#else
    Node *n = (Node *)aligned_alloc(64, (sizeof(Node) + 127) / 64 * 64);
    n->construct();
#endif
    size_t idx = 0;
    for (size_t i = 0; i < Node::NUM_SLOTS; ++i) { CHECK(n->acquireSlot(idx) == 0); CHECK(idx == i); }
    CHECK(n->acquireSlot(idx) == -1);                 // the 11th fails
    CHECK(n->source_ref_count() == Node::NUM_SLOTS);
    for (size_t i = 0; i < Node::NUM_SLOTS; ++i) CHECK_NOTHROW(n->read_barrier(i));   // all ten usable (ref: slot 5 throws)
    for (size_t i = 0; i < Node::NUM_SLOTS; ++i) CHECK(n->releaseSlot(i) == 0);
    CHECK(n->source_ref_count() == 0);
    CHECK(n->releaseSlot(3) == 0); CHECK(n->source_ref_count() == 0);   // removing again keeps 0
    CHECK(n->releaseSlot(Node::NUM_SLOTS) == -1);
    CHECK_THROWS(n->read_barrier((size_t)-1));
    CHECK(n->acquireSlot(idx) == 0 && idx == 0);
    CHECK_THROWS(n->read_barrier(1));                 // not bound to that slot
#if defined(OAT_SHMEM_BOOST)
    delete n;
#else
    free(n);
#endif
}

static void sink_tests()
{
    scrub();
    {   // Sink_test.cpp:34-54: one sink per node
        Sink<int> s1, s2;
        CHECK_NOTHROW(s1.bind(ADDR));
        CHECK_THROWS(s2.bind(ADDR));
        CHECK_THROWS(s1.bind(ADDR));                  // Sink_test.cpp:109-116
    }
    {   // Sink_test.cpp:56-107: bind before wait/post; wait/post alternate
        Sink<int> s;
        CHECK_THROWS(s.wait()); CHECK_THROWS(s.post());
        CHECK_THROWS(s.retrieve());                   // :125-130
        s.bind(ADDR);
        CHECK_THROWS(s.post());                       // post before wait
        CHECK_NOTHROW(s.wait());
        CHECK_THROWS(s.wait());                       // wait twice
        CHECK_NOTHROW(s.post());
        int *p = nullptr;
        CHECK_NOTHROW(p = s.retrieve());              // :132-150
        *p = 42; CHECK(*s.retrieve() == 42);
    }
    {   // Sink_test.cpp:153-185
        Sink<Frame> s;
        CHECK_THROWS(s.wait()); CHECK_THROWS(s.post()); CHECK_THROWS(s.retrieve(2, 2, kCV_8UC3, PIX_BGR));
        s.bind(ADDR, 12);
        CHECK_THROWS(s.retrieve(3, 3, kCV_8UC3, PIX_BGR));     // does not fit
        Frame f = s.retrieve(2, 2, kCV_8UC3, PIX_BGR);
        CHECK(f.bytes() == 12);
    }
    // destroyed sinks without sources unlink their segments
    { Sink<int> again; CHECK_NOTHROW(again.bind(ADDR)); }
}

static void source_tests()
{
    scrub();
    {   // Source_test.cpp:34-101: ten sources connect, the 11th is refused
        Sink<int> sink; sink.bind(ADDR);
        std::vector<std::unique_ptr<Source<int>>> src;
        for (size_t i = 0; i < Node::NUM_SLOTS; ++i) {
            src.emplace_back(new Source<int>());
            src.back()->touch(ADDR);
            CHECK(src.back()->state() == SourceState::TOUCHED);
            CHECK(src.back()->connect() == SourceState::CONNECTED);
        }
        Source<int> extra; extra.touch(ADDR);
        CHECK(extra.state() == SourceState::ERR_NODEFULL);
        CHECK_THROWS(extra.connect());
    }
    {   // Source_test.cpp:103-140
        Source<int> s;
        CHECK_THROWS(s.wait()); CHECK_THROWS(s.post());
        Sink<int> sink; sink.bind(ADDR);
        s.touch(ADDR);
        CHECK_THROWS(s.touch(ADDR));
        CHECK(s.connect() == SourceState::CONNECTED);
        CHECK_THROWS(s.connect());
    }
    {   // Source_test.cpp:142-161: Source<T> only connects to Sink<T>
        Sink<int> sink; sink.bind(ADDR);
        Source<float> s; s.touch(ADDR);
        CHECK_THROWS(s.connect());
        CHECK(s.state() == SourceState::ERR_TYPEMIS);
    }
    {   // Source_test.cpp:163-218: shared object visible on both sides
        Sink<int> sink; sink.bind(ADDR);
        Source<int> s; s.touch(ADDR); s.connect();
        *sink.retrieve() = 7; CHECK(*s.retrieve() == 7); CHECK(s.clone() == 7);
        *s.retrieve() = 9; CHECK(*sink.retrieve() == 9);
    }
    {   // frames: header, colour requirement (Source.h:300-313), Sample travels with the pixels
        Sink<Frame> sink; sink.bind(ADDR, 2 * 3 * 3);
        Frame shared = sink.retrieve(2, 3, kCV_8UC3, PIX_BGR);
        for (int i = 0; i < 18; ++i) shared.data()[i] = (uint8_t)(i * 3);
        shared.sample().set_rate_hz(100.0);
        shared.sample().incrementCount();
        Source<Frame> bad; bad.touch(ADDR);
        CHECK_THROWS(bad.connect(PIX_HSV));                       // "Maybe use oat-framefilt col?"
        Source<Frame> s; s.touch(ADDR);
        CHECK(s.connect(PIX_BGR) == SourceState::CONNECTED);
        auto p = s.parameters();
        CHECK(p.rows == 2 && p.cols == 3 && p.type == kCV_8UC3 && p.color == PIX_BGR && p.bytes == 18);
        Frame copy; s.copyTo(copy);
        CHECK(copy.bytes() == 18 && copy.data()[17] == 51 && copy.sample().count() == 1 && copy.sample().period_microseconds() == 10000);
    }
    {   // the object segment is keyed by the reference's mangled type names
        CHECK(std::string(detail::TypeName<Position2D>::get()) == "N3oat10Position2DE");
        CHECK(std::string(detail::TypeName<SharedFrameHeader>::get()) == "N3oat17SharedFrameHeaderE");
    }
}

static void concurrency_tests()
{
    scrub();
    {   // concurrency_test.cpp:79-237
        Sink<int> sink; Source<int> src;
        sink.bind(ADDR); src.touch(ADDR); src.connect();
        // source first: blocks until the sink posts
        auto f = std::async(std::launch::async, [&] { src.wait(); });
        CHECK(!ready(f));
        sink.wait(); CHECK(!ready(f, 1ms));
        sink.post(); CHECK(ready(f, 200ms));
        // sink blocks until the source posts
        auto g = std::async(std::launch::async, [&] { sink.wait(); });
        CHECK(!ready(g));
        src.post(); CHECK(ready(g, 200ms));
        sink.post();
        // second source attached: sink needs BOTH posts
        Source<int> src2; src2.touch(ADDR); src2.connect();
        src.wait(); src.post();
        sink.wait(); sink.post();                       // now both must read
        auto h = std::async(std::launch::async, [&] { sink.wait(); });
        src.wait(); src.post(); CHECK(!ready(h));
        src2.wait(); src2.post(); CHECK(ready(h, 200ms));
        sink.post();
    }
    scrub();
    {   // concurrency_test.cpp:239-421: order of connect / bind is irrelevant
        Source<int> a, b; Sink<int> sink;
        a.touch(ADDR); b.touch(ADDR);
        auto fa = std::async(std::launch::async, [&] { return a.connect(); });
        auto fb = std::async(std::launch::async, [&] { return b.connect(); });
        CHECK(!ready(fa) && !ready(fb));                // sources wait for the sink to bind AND write
        sink.bind(ADDR);
        CHECK(!ready(fa, 1ms));
        sink.wait(); *sink.retrieve() = 5; sink.post();
        CHECK(ready(fa, 300ms) && ready(fb, 300ms));
        CHECK(fa.get() == SourceState::CONNECTED && fb.get() == SourceState::CONNECTED);
        // the first token was handed back ("freebie"): both can still read write #1
        a.wait(); CHECK(a.clone() == 5); a.post();
        b.wait(); CHECK(b.clone() == 5); b.post();
        auto g = std::async(std::launch::async, [&] { sink.wait(); });
        CHECK(ready(g, 200ms)); sink.post();
    }
    scrub();
    {   // concurrency_test.cpp:426-473: a source destructs while the sink is inside its critical section
        Sink<int> sink; Source<int> s0; auto s1 = std::make_unique<Source<int>>();
        sink.bind(ADDR); s0.touch(ADDR); s0.connect(); s1->touch(ADDR); s1->connect();
        sink.wait();
        auto f = std::async(std::launch::async, [&] { s0.wait(); });
        s1.reset();
        CHECK(!ready(f));
        CHECK_NOTHROW(sink.post());
        CHECK(ready(f, 200ms));
        s0.post();
    }
    scrub();
    {   // concurrency_test.cpp:475-527: sources connect while the sink is inside its critical section
        Sink<int> sink; Source<int> s0, s1;
        sink.bind(ADDR); sink.wait();
        s0.touch(ADDR); s1.touch(ADDR);
        auto f0 = std::async(std::launch::async, [&] { s0.connect(); s0.wait(); });
        auto f1 = std::async(std::launch::async, [&] { s1.connect(); s1.wait(); });
        CHECK(!ready(f0) && !ready(f1, 1ms));
        CHECK_NOTHROW(sink.post());
        CHECK(ready(f0, 200ms) && ready(f1, 200ms));
        s0.post(); s1.post();
    }
    scrub();
    {   // hardening beyond the reference: a source that detaches while it still owes a read of
        // the current token must not stall the sink
        Sink<int> sink; sink.bind(ADDR);
        auto a = std::make_unique<Source<int>>(); Source<int> b;
        a->touch(ADDR); b.touch(ADDR); a->connect(); b.connect();
        sink.wait(); sink.post();                       // token 1 owed by a and b
        b.wait(); b.post();
        auto g = std::async(std::launch::async, [&] { sink.wait(); });
        CHECK(!ready(g));                               // still waiting for `a`
        a.reset();                                      // a leaves without reading
        CHECK(ready(g, 300ms));
        sink.post();
        b.wait(); b.post();
        auto h = std::async(std::launch::async, [&] { sink.wait(); });
        CHECK(ready(h, 300ms));                         // only b is required now
        sink.post();
    }
    scrub();
    {   // END propagation (Sink.h:73-91, Source.h:207-209): a waiting source wakes when the sink dies
        auto sink = std::make_unique<Sink<int>>(); sink->bind(ADDR);
        Source<int> s; s.touch(ADDR); s.connect();
        auto f = std::async(std::launch::async, [&] { return s.wait(); });
        CHECK(!ready(f));
        sink.reset();
        CHECK(ready(f, 300ms));
        CHECK(f.get() == NodeState::END);
    }
    scrub();
}

// Position wire formats (lib/datatypes/Position2D.cpp:24-96, Position2D.h:170-233)
static void wire_tests(const char *npy_path)
{
    Position2D p("cam");
    p.sample_.set_rate_hz(100.0);
    p.sample_.incrementCount(); p.sample_.incrementCount();
    p.position_valid = true; p.position.x = 6.0; p.position.y = 3.254901960784;
    auto r = packPosition(p);
    CHECK(r.size() == kNpyDtypeBytes && r.size() == 82);
    uint64_t tick, usec; int32_t unit; double px, py;
    memcpy(&tick, &r[0], 8); memcpy(&usec, &r[8], 8); memcpy(&unit, &r[16], 4);
    memcpy(&px, &r[21], 8); memcpy(&py, &r[29], 8);
    CHECK(tick == 2 && usec == 20000 && unit == 0 && r[20] == 1 && px == 6.0 && py == 3.254901960784);
    CHECK(r[37] == 0 && r[54] == 0 && r[71] == 0);               // vel_ok, head_ok, reg_ok
    CHECK(serializePosition(p) == "{\"tick\":2,\"usec\":20000,\"unit\":0,\"pos_ok\":true,\"pos_xy\":[6.0,3.2549],"
                                  "\"vel_ok\":false,\"head_ok\":false,\"reg_ok\":false}");
    p.position_valid = false;
    CHECK(serializePosition(p) == "{\"tick\":2,\"usec\":20000,\"unit\":0,\"pos_ok\":false,\"vel_ok\":false,"
                                  "\"head_ok\":false,\"reg_ok\":false}");
    if (npy_path) {     // three records in a .npy file for numpy to read back (tests/test_host_pipeline.py)
        FILE *f = fopen(npy_path, "wb");
        std::string d = std::string("{'descr': ") + npy_dtype() + ", 'fortran_order': False, 'shape': (3,), }";
        d += std::string((64 - (10 + d.size() + 1) % 64) % 64, ' ') + "\n";
        uint16_t len = (uint16_t)d.size();
        fwrite("\x93NUMPY\x01\x00", 1, 8, f); fwrite(&len, 2, 1, f); fwrite(d.data(), 1, d.size(), f);
        for (int i = 0; i < 3; ++i) {
            p.sample_.incrementCount();
            p.position_valid = i != 1; p.position.x = 10.5 * i; p.position.y = -i;
            auto rec = packPosition(p);
            fwrite(rec.data(), 1, rec.size(), f);
        }
        fclose(f);
    }
}

int main(int argc, char **argv)
{
    wire_tests(argc > 2 ? argv[2] : nullptr);
    ADDR = argc > 1 ? argv[1] : "oat_hip_test_" + std::to_string(getpid());
    node_tests();
    sink_tests();
    source_tests();
    concurrency_tests();
    printf("%d checks, %d failures\n", g_checks, g_fail);
    return g_fail ? 1 : 0;
}
