// shmemdf_boost.hpp -- the SAME Node / Sink<T> / Source<T> interface as shmemdf.hpp, but with the BYTES of
// stock Oat: Boost.Interprocess managed_shared_memory segments holding exactly the named objects the
// reference constructs, so that binaries built with it attach to an unmodified Oat pipeline.
//
//   lib/shmemdf/Sink.h:164-207    Sink<T>::bind:     "<addr>_node" open_or_create, 1024 + sizeof(Node),
//                                                    find_or_construct<Node>(typeid(Node).name());
//                                                    "<addr>_obj" create_only, 1024 + sizeof(T),
//                                                    find_or_construct<T>(typeid(T).name())(args...)
//   lib/shmemdf/Sink.h:232-298    Sink<Frame>:       obj segment 1024 + sizeof(SharedFrameHeader) + bytes + 8;
//                                                    retrieve(): allocate(sizeof(Sample)), allocate(pixels),
//                                                    handles into the SharedFrameHeader
//   lib/shmemdf/Source.h:114-185  touch / connect:   find<T>(typeid(T).name()), nullptr -> type mismatch
//   lib/shmemdf/Source.h:315-368  Source<Frame>:     Frame over get_address_from_handle(data / sample)
//   lib/shmemdf/Node.h:41-184     Node:              member order and semaphore types below
//   lib/shmemdf/SharedFrameHeader.h:91-96            params_, data_, sample_ (oat::SharedFrameHeader of
//                                                    datatypes.hpp has that layout, static_asserted)
//
// Compile-guarded AND opt-in: this image (and the GPU box) has no Boost, so the header is inert here; where
// Boost exists, `make -C oat_amd/host SHMEM=boost` defines OAT_SHMEM_BOOST and shmemdf.hpp then selects
// oat::stock instead of oat::native (`make boostcheck` builds and runs the protocol tests on it first).  It has
// never been run against a stock oat-* process: until it has, the native transport is what the binaries use.  The object NAMES are the
// Itanium-mangled type names typeid(T).name() yields for the reference's own classes with GCC/Clang,
// spelled out because this tree's classes live in other namespaces.
#pragma once

#if defined(__has_include)
#if __has_include(<boost/interprocess/managed_shared_memory.hpp>)
#define OAT_HAVE_BOOST_INTERPROCESS 1
#endif
#endif

// ---- layout self-check (compiled everywhere, Boost or not) ----
// The reference's Node (lib/shmemdf/Node.h:170-183, with `write_barrier` declared in front of the private members,
// :143) as a plain aggregate over an arbitrary semaphore type: what offsets and size its member order implies.
// With glibc's 32-byte sem_t -- what boost::interprocess::interprocess_semaphore wraps on Linux
// (BOOST_INTERPROCESS_POSIX_SEMAPHORES) -- that is write_barrier 0, sink_state_ 32, source_slots_ 40,
// source_read_required_ 48, source_ref_count_ 56, write_number_ 64, mutex_ 72, rb0_.. 104 + 32 i, 424 bytes: the
// "<addr>_node" segment of Sink.h:171-176 is 1024 + 424 bytes.  stock::Node below is asserted against this twin
// instantiated with Boost's own semaphore type, member by member.
#include <atomic>
#include <bitset>
#include <cstddef>
#include <cstdint>
#include <semaphore.h>

namespace oat {
namespace stock_layout {
template <typename Sem>
struct NodeTwin {
    Sem write_barrier;
    std::atomic<int> sink_state_;
    std::bitset<10> source_slots_, source_read_required_;
    size_t source_ref_count_;
    uint64_t write_number_;
    Sem mutex_;
    Sem rb_[10];
};
#if defined(__linux__) && defined(__x86_64__)
static_assert(sizeof(sem_t) == 32 && sizeof(std::bitset<10>) == 8, "glibc x86-64 sem_t / libstdc++ bitset sizes");
static_assert(offsetof(NodeTwin<sem_t>, sink_state_) == 32 && offsetof(NodeTwin<sem_t>, source_slots_) == 40 &&
              offsetof(NodeTwin<sem_t>, source_read_required_) == 48 && offsetof(NodeTwin<sem_t>, source_ref_count_) == 56 &&
              offsetof(NodeTwin<sem_t>, write_number_) == 64 && offsetof(NodeTwin<sem_t>, mutex_) == 72 &&
              offsetof(NodeTwin<sem_t>, rb_) == 104 && sizeof(NodeTwin<sem_t>) == 424,
              "layout implied by lib/shmemdf/Node.h:143,170-183");
#endif
}  // namespace stock_layout
}  // namespace oat

#if defined(OAT_HAVE_BOOST_INTERPROCESS)

#include <boost/interprocess/managed_shared_memory.hpp>
#include <boost/interprocess/sync/interprocess_semaphore.hpp>
#include <boost/date_time/posix_time/posix_time.hpp>

#include <atomic>
#include <bitset>

namespace oat {
namespace stock {

namespace bip = boost::interprocess;

constexpr const char *kNodeName = "N3oat4NodeE";                       // typeid(oat::Node).name()
template <typename T> struct StockName;
template <> struct StockName<Position2D> { static const char *get() { return "N3oat10Position2DE"; } };
template <> struct StockName<SharedFrameHeader> { static const char *get() { return "N3oat17SharedFrameHeaderE"; } };
template <> struct StockName<int> { static const char *get() { return "i"; } };       // (the protocol tests: Sink<int>, Source<float>,
template <> struct StockName<float> { static const char *get() { return "f"; } };     //  as the reference's own Sink_test / Source_test)

// lib/shmemdf/Node.h:41-184 -- same members, same order, same types: this IS the object stock binaries
// find under kNodeName.  (The reference's read_barrier() switch lacks `case 5`; slot 5 works here.)
class Node {
public:
    using semaphore = bip::interprocess_semaphore;
    static constexpr size_t NUM_SLOTS{10};

    Node() { source_slots_.reset(); source_read_required_.reset(); }
    Node(const Node &) = delete;
    Node &operator=(const Node &) = delete;

    void set_sink_state(NodeState v) { sink_state_ = v; }
    NodeState sink_state() const { return sink_state_; }
    uint64_t write_number() const { return write_number_; }

    void notifySinkWriteComplete()
    {
        mutex_.wait();
        source_read_required_ = source_slots_;
        for (size_t i = 0; i < source_slots_.size(); i++)
            if (source_slots_[i]) read_barrier(i).post();
        ++write_number_;
        mutex_.post();
    }
    bool notifySourceReadComplete(size_t index)
    {
        mutex_.wait();
        source_read_required_[index] = false;
        const bool reads_finished = source_read_required_.none();
        mutex_.post();
        return reads_finished;
    }
    int acquireSlot(size_t &index)
    {
        mutex_.wait();
        if (source_slots_.all()) { mutex_.post(); return -1; }
        index = 0;
        while (source_slots_[index]) ++index;
        source_slots_[index] = true;
        source_ref_count_ = source_slots_.count();
        mutex_.post();
        return 0;
    }
    int releaseSlot(size_t index)
    {
        if (index >= source_slots_.size()) return -1;
        mutex_.wait();
        source_slots_[index] = false;
        source_ref_count_ = source_slots_.count();
        mutex_.post();
        return 0;
    }
    size_t source_ref_count() const { return source_ref_count_; }

    semaphore write_barrier{1};

    semaphore &read_barrier(size_t index)
    {
        if (index >= NUM_SLOTS || !source_slots_[index])
            throw std::runtime_error("Requested index refers to a SOURCE that is not bound to this node.");
        semaphore *rb[NUM_SLOTS] = {&rb0_, &rb1_, &rb2_, &rb3_, &rb4_, &rb5_, &rb6_, &rb7_, &rb8_, &rb9_};
        return *rb[index];
    }

private:
    std::atomic<NodeState> sink_state_{NodeState::UNDEFINED};
    std::bitset<NUM_SLOTS> source_slots_;
    std::bitset<NUM_SLOTS> source_read_required_;
    size_t source_ref_count_{0};
    uint64_t write_number_{0};
    semaphore mutex_{1};
    semaphore rb0_{0}, rb1_{0}, rb2_{0}, rb3_{0}, rb4_{0}, rb5_{0}, rb6_{0}, rb7_{0}, rb8_{0}, rb9_{0};
    friend struct NodeLayoutCheck;
};

// stock::Node member by member against the reference's layout (stock_layout::NodeTwin over Boost's semaphore type)
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Winvalid-offsetof"
struct NodeLayoutCheck {
    using Twin = stock_layout::NodeTwin<Node::semaphore>;
    static_assert(sizeof(Node) == sizeof(Twin) && alignof(Node) == alignof(Twin), "stock::Node size");
    static_assert(offsetof(Node, write_barrier) == offsetof(Twin, write_barrier), "write_barrier");
    static_assert(offsetof(Node, sink_state_) == offsetof(Twin, sink_state_), "sink_state_");
    static_assert(offsetof(Node, source_slots_) == offsetof(Twin, source_slots_), "source_slots_");
    static_assert(offsetof(Node, source_read_required_) == offsetof(Twin, source_read_required_), "source_read_required_");
    static_assert(offsetof(Node, source_ref_count_) == offsetof(Twin, source_ref_count_), "source_ref_count_");
    static_assert(offsetof(Node, write_number_) == offsetof(Twin, write_number_), "write_number_");
    static_assert(offsetof(Node, mutex_) == offsetof(Twin, mutex_), "mutex_");
    static_assert(offsetof(Node, rb0_) == offsetof(Twin, rb_) && offsetof(Node, rb9_) == offsetof(Twin, rb_) + 9 * sizeof(Node::semaphore), "read barriers");
    static_assert(sizeof(NodeState) == sizeof(int), "NodeState is an int-sized enum in the reference");
};
#pragma GCC diagnostic pop

inline bool timed_wait_10ms(bip::interprocess_semaphore &s)
{
    // Sink.h:101-109 / Source.h:197-207: timed_wait in 10 ms slices so that SIGINT is honoured
    const boost::system_time timeout = boost::get_system_time() + boost::posix_time::milliseconds(10);
    return s.timed_wait(timeout);
}

// ---------------------------------------------------------------------------- Sink --
template <typename T>
class SinkBase {
public:
    SinkBase() = default;
    virtual ~SinkBase()
    {
        if (bound_) {                                   // Sink.h:64-90
            node_->set_sink_state(NodeState::END);
            if (node_->source_ref_count() == 0) {
                bip::shared_memory_object::remove(node_address_.c_str());
                bip::shared_memory_object::remove(obj_address_.c_str());
            }
        }
    }
    void wait()
    {
        if (!bound_) throw std::runtime_error("Sink must be bound before calling wait()");
        if (did_wait_need_post_) throw std::runtime_error("wait() called when post() was required.");
        while (node_->source_ref_count() > 0 && !timed_wait_10ms(node_->write_barrier) && !quit) {}
        did_wait_need_post_ = true;
    }
    void post()
    {
        if (!bound_) throw std::runtime_error("Sink must be bound before calling post()");
        if (!did_wait_need_post_) throw std::runtime_error("post() called when wait() was required.");
        node_->notifySinkWriteComplete();
        did_wait_need_post_ = false;
    }
    size_t source_ref_count() const { return node_->source_ref_count(); }   // SOURCEs attached to this node (Node.h:139)
    uint64_t write_number() const { return node_->write_number(); }

protected:
    void bind_node(const std::string &address)
    {
        if (bound_) throw std::runtime_error("A sink can only bind a single time to a single node.");
        address_ = address;
        node_address_ = address + "_node";
        obj_address_ = address + "_obj";
        node_shmem_ = bip::managed_shared_memory(bip::open_or_create, node_address_.c_str(), 1024 + sizeof(Node));
        node_ = node_shmem_.find_or_construct<Node>(kNodeName)();
        if (node_->sink_state() != NodeState::UNDEFINED)
            throw std::runtime_error("Requested SINK address, '" + address + "', is not available.");
    }
    std::string address_, node_address_, obj_address_;
    bip::managed_shared_memory node_shmem_, obj_shmem_;
    Node *node_{nullptr};
    T *sh_object_{nullptr};
    bool bound_{false};

private:
    bool did_wait_need_post_{false};
};

template <typename T>
class Sink : public SinkBase<T> {
public:
    template <typename... Targs>
    void bind(const std::string &address, Targs... args)                     // Sink.h:164-207
    {
        this->bind_node(address);
        this->obj_shmem_ = bip::managed_shared_memory(bip::create_only, this->obj_address_.c_str(), 1024 + sizeof(T));
        this->sh_object_ = this->obj_shmem_.template find_or_construct<T>(StockName<T>::get())(args...);
        this->node_->set_sink_state(NodeState::SINK_BOUND);
        this->bound_ = true;
    }
    T *retrieve()
    {
        if (!this->bound_) throw std::runtime_error("SINK must be bound before shared object is retrieved.");
        return this->sh_object_;
    }
};

template <>
class Sink<Frame> : public SinkBase<SharedFrameHeader> {
public:
    void bind(const std::string &address, size_t bytes)                        // Sink.h:232-272
    {
        bind_node(address);
        obj_shmem_ = bip::managed_shared_memory(bip::create_only, obj_address_.c_str(),
                                                1024 + sizeof(SharedFrameHeader) + bytes + sizeof(uint64_t));
        sh_object_ = obj_shmem_.find_or_construct<SharedFrameHeader>(StockName<SharedFrameHeader>::get())();
        node_->set_sink_state(NodeState::SINK_BOUND);
        bound_ = true;
    }
    Frame retrieve(size_t rows, size_t cols, int type, PixelColor color)      // Sink.h:274-298
    {
        if (!bound_) throw std::runtime_error("SINK must be bound before shared frame is retrieved.");
        void *sample = obj_shmem_.allocate(sizeof(Sample));
        const auto sample_handle = obj_shmem_.get_handle_from_address(sample);
        void *data = obj_shmem_.allocate(rows * cols * color_bytes(color));
        const auto data_handle = obj_shmem_.get_handle_from_address(data);
        new (sample) Sample();
        sh_object_->setParameters((SharedFrameHeader::handle_t)data_handle, (SharedFrameHeader::handle_t)sample_handle,
                                  rows, cols, type, color);
        return Frame(rows, cols, color, data, sample);
    }
};

// -------------------------------------------------------------------------- Source --
template <typename T>
class SourceBase {
public:
    SourceBase() = default;
    virtual ~SourceBase()                                                       // Source.h:86-111
    {
        if (state_ >= SourceState::TOUCHED || state_ == SourceState::ERR_TYPEMIS) node_->releaseSlot(slot_index_);
        if (node_ != nullptr && node_->source_ref_count() == 0 && node_->sink_state() != NodeState::SINK_BOUND) {
            bip::shared_memory_object::remove(node_address_.c_str());
            bip::shared_memory_object::remove(obj_address_.c_str());
        }
    }
    void touch(const std::string &address)                                      // Source.h:114-147
    {
        if (state_ != SourceState::VIRGIN) throw std::runtime_error("A source can only connect a single time to a single node.");
        address_ = address;
        node_address_ = address + "_node";
        obj_address_ = address + "_obj";
        node_shmem_ = bip::managed_shared_memory(bip::open_or_create, node_address_.c_str(), 1024 + sizeof(Node));
        node_ = node_shmem_.find_or_construct<Node>(kNodeName)();
        if (node_->acquireSlot(slot_index_) < 0) { state_ = SourceState::ERR_NODEFULL; return; }
        state_ = SourceState::TOUCHED;
    }
    NodeState wait()                                                            // Source.h:187-215
    {
        if (state_ < SourceState::TOUCHED) throw std::runtime_error("Source must have touched node before calling wait()");
        if (did_wait_need_post_) throw std::runtime_error("wait() called when post() was required.");
        while (!timed_wait_10ms(node_->read_barrier(slot_index_)) && !quit) {
            if (node_->sink_state() == NodeState::END) break;
        }
        did_wait_need_post_ = true;
        return node_->sink_state();
    }
    void post()                                                                 // Source.h:217-232
    {
        if (state_ < SourceState::CONNECTED) throw std::runtime_error("source must be connected before calling post()");
        if (!did_wait_need_post_) throw std::runtime_error("post() called when wait() was required.");
        if (node_->notifySourceReadComplete(slot_index_)) node_->write_barrier.post();
        did_wait_need_post_ = false;
    }
    SourceState state() const { return state_; }
    uint64_t write_number() const { return node_->write_number(); }
    bool token_waiting()                                    // see shmemdf.hpp; try_wait + post leaves the count as it was
    {
        if (state_ < SourceState::TOUCHED) return false;
        if (node_->read_barrier(slot_index_).try_wait()) { node_->read_barrier(slot_index_).post(); return true; }
        return node_->sink_state() == NodeState::END;
    }

protected:
    SourceState connect_object()                                                // Source.h:149-185
    {
        if (state_ != SourceState::TOUCHED) throw std::runtime_error("A source can only connect() after it has touch()ed a node.");
        if (node_->sink_state() != NodeState::SINK_BOUND) {
            if (wait() != NodeState::SINK_BOUND) return SourceState::ERR_CONNECT;
            node_->read_barrier(slot_index_).post();          // the "freebie"
            did_wait_need_post_ = false;
        }
        obj_shmem_ = bip::managed_shared_memory(bip::open_only, obj_address_.c_str());
        sh_object_ = obj_shmem_.find<T>(StockName<T>::get()).first;
        if (sh_object_ == nullptr) {
            state_ = SourceState::ERR_TYPEMIS;
            throw std::runtime_error("Type mismatch: Source<T> can only connect to Node<T>.");
        }
        state_ = SourceState::CONNECTED;
        return SourceState::CONNECTED;
    }
    std::string address_, node_address_, obj_address_;
    bip::managed_shared_memory node_shmem_, obj_shmem_;
    Node *node_{nullptr};
    T *sh_object_{nullptr};
    size_t slot_index_{0};
    SourceState state_{SourceState::VIRGIN};

private:
    bool did_wait_need_post_{false};
};

template <typename T>
class Source : public SourceBase<T> {
public:
    SourceState connect() { return this->connect_object(); }
    T *retrieve() const { return this->sh_object_; }
    T clone() const { return *this->sh_object_; }
};

template <>
class Source<Frame> : public SourceBase<SharedFrameHeader> {
public:
    SourceState connect()                                                       // Source.h:315-368
    {
        auto rc = connect_object();
        if (rc != SourceState::CONNECTED) return rc;
        auto p = sh_object_->params();
        frame_ = Frame(p.rows, p.cols, p.color, obj_shmem_.get_address_from_handle(sh_object_->data()),
                       obj_shmem_.get_address_from_handle(sh_object_->sample()));
        parameters_ = p;
        parameters_.bytes = frame_.bytes();
        return rc;
    }
    SourceState connect(PixelColor color)                                       // Source.h:300-313
    {
        auto rc = connect();
        if (rc == SourceState::CONNECTED && frame_.color() != color)
            throw std::runtime_error(std::string("Component requires frame source with pixels of type ") +
                                     color_str(color) + ". Maybe use oat-framefilt col?");
        return rc;
    }
    const Frame *retrieve() const { return &frame_; }
    void copyTo(Frame &frame) const { frame_.copyTo(frame); }
    FrameParams parameters() const { return parameters_; }

private:
    Frame frame_;
    FrameParams parameters_;
};

inline bool remove_segment(const std::string &name) { return bip::shared_memory_object::remove(name.c_str()); }

}  // namespace stock
}  // namespace oat

#endif  // OAT_HAVE_BOOST_INTERPROCESS
