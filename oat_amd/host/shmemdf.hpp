// shmemdf.hpp -- Oat's shared-memory dataflow (Node / Sink<T> / Source<T>) on plain POSIX
// shm_open + mmap + process-shared sem_t, with the protocol and segment NAMES of the reference:
//
//   lib/shmemdf/Node.h:41-184    sink_state {END -1, UNDEFINED 0, SINK_BOUND 1, ERROR 2},
//                                <= 10 source slots, write_barrier (init 1), one read barrier per
//                                slot (init 0), mutex, write_number
//   lib/shmemdf/Sink.h:40-298    bind / retrieve / wait / post, END on destruction
//   lib/shmemdf/Source.h:52-368  touch / connect / wait / post / copyTo, slot release, unlink
//
// Segments are "/<addr>_node" and "/<addr>_obj" exactly as the reference names them
// (Sink.h:239-241).  The reference builds them with Boost.Interprocess managed_shared_memory;
// Boost is not available in this image, so the BYTES inside the segments are this file's own
// layout: all binaries built from this tree interoperate with each other; talking to stock Oat
// binaries needs the Boost adapter described in INTEGRATION.md.  The executable specification of
// the protocol is the reference's test/shmemdf/concurrency_test.cpp, restated in
// host/test_shmemdf.cpp.
#pragma once

#include "datatypes.hpp"

#include <atomic>
#include <cerrno>
#include <csignal>
#include <ctime>
#include <fcntl.h>
#include <semaphore.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace oat {

// set by the SIGINT handler of the component (lib/base/Component.cpp:36-48)
extern volatile sig_atomic_t quit;

enum class NodeState : int32_t { END = -1, UNDEFINED = 0, SINK_BOUND = 1, ERROR = 2 };
enum class SourceState : int { ERR_CONNECT = -3, ERR_NODEFULL = -2, ERR_TYPEMIS = -1, VIRGIN = 0, TOUCHED = 1, CONNECTED = 2 };

namespace detail {

constexpr uint32_t kMagic = 0x0A7D0F01u;

inline bool timed_wait_10ms(sem_t *s)
{
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_nsec += 10 * 1000 * 1000;
    if (ts.tv_nsec >= 1000000000L) { ts.tv_sec += 1; ts.tv_nsec -= 1000000000L; }
    while (sem_timedwait(s, &ts) != 0) {
        if (errno == EINTR) { if (quit) return false; continue; }
        return false;   // ETIMEDOUT
    }
    return true;
}

// One POSIX shm segment, mapped whole.
class Segment {
public:
    Segment() = default;
    ~Segment() { close(); }
    Segment(const Segment &) = delete;
    Segment &operator=(const Segment &) = delete;

    // returns true if this call created the segment
    bool open_or_create(const std::string &name, size_t bytes)
    {
        name_ = "/" + name;
        int fd = shm_open(name_.c_str(), O_RDWR | O_CREAT | O_EXCL, 0666);
        bool created = fd >= 0;
        if (!created) {
            if (errno != EEXIST) throw std::runtime_error("shm_open(" + name_ + ") failed");
            fd = shm_open(name_.c_str(), O_RDWR, 0666);
            if (fd < 0) throw std::runtime_error("shm_open(" + name_ + ") failed");
            // the creator may not have sized it yet
            struct stat st;
            for (int i = 0; i < 2000; ++i) {
                if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
                usleep(500);
            }
            if (fstat(fd, &st) != 0 || (size_t)st.st_size < bytes) { ::close(fd); throw std::runtime_error("shm segment " + name_ + " has the wrong size"); }
            bytes = st.st_size;
        } else if (ftruncate(fd, bytes) != 0) {
            ::close(fd);
            shm_unlink(name_.c_str());
            throw std::runtime_error("ftruncate(" + name_ + ") failed");
        }
        map(fd, bytes);
        return created;
    }
    void create_only(const std::string &name, size_t bytes)
    {
        name_ = "/" + name;
        int fd = shm_open(name_.c_str(), O_RDWR | O_CREAT | O_EXCL, 0666);
        if (fd < 0) throw std::runtime_error("shm segment " + name_ + " already exists");
        if (ftruncate(fd, bytes) != 0) { ::close(fd); shm_unlink(name_.c_str()); throw std::runtime_error("ftruncate failed"); }
        map(fd, bytes);
    }
    void open_only(const std::string &name)
    {
        name_ = "/" + name;
        int fd = shm_open(name_.c_str(), O_RDWR, 0666);
        if (fd < 0) throw std::runtime_error("shm segment " + name_ + " does not exist");
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); throw std::runtime_error("fstat failed"); }
        map(fd, st.st_size);
    }
    static bool remove(const std::string &name) { return shm_unlink(("/" + name).c_str()) == 0; }

    uint8_t *base() const { return base_; }
    size_t size() const { return size_; }
    void close()
    {
        if (base_) munmap(base_, size_);
        base_ = nullptr;
    }

private:
    void map(int fd, size_t bytes)
    {
        void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        ::close(fd);
        if (p == MAP_FAILED) throw std::runtime_error("mmap(" + name_ + ") failed");
        base_ = (uint8_t *)p;
        size_ = bytes;
    }
    std::string name_;
    uint8_t *base_{nullptr};
    size_t size_{0};
};

}  // namespace detail

// Everything below, down to the selection at the end of the file, is the NATIVE transport (POSIX shm + sem_t).
// shmemdf_boost.hpp holds the same classes over Boost.Interprocess with stock Oat's bytes (namespace stock).
namespace native {

// ----------------------------------------------------------------------------- Node --
class Node {
public:
    static constexpr size_t NUM_SLOTS{10};

    void construct()
    {
        sink_state_.store((int32_t)NodeState::UNDEFINED);
        source_slots_ = 0; source_read_required_ = 0; source_ref_count_ = 0; write_number_ = 0;
        sem_init(&write_barrier, 1, 1);     // write always occurs before read
        sem_init(&mutex_, 1, 1);
        for (auto &s : rb_) sem_init(&s, 1, 0);
        ready_.store(detail::kMagic, std::memory_order_release);
    }
    void wait_constructed() const
    {
        for (int i = 0; i < 4000 && ready_.load(std::memory_order_acquire) != detail::kMagic; ++i) usleep(500);
        if (ready_.load(std::memory_order_acquire) != detail::kMagic) throw std::runtime_error("node was never initialised");
    }

    void set_sink_state(NodeState v) { sink_state_.store((int32_t)v); }
    NodeState sink_state() const { return (NodeState)sink_state_.load(); }
    uint64_t write_number() const { return write_number_; }

    // Node.h:69-84
    void notifySinkWriteComplete()
    {
        sem_wait_retry(&mutex_);
        source_read_required_ = source_slots_;
        for (size_t i = 0; i < NUM_SLOTS; i++)
            if (source_slots_ & (1u << i)) sem_post(&rb_[i]);
        ++write_number_;
        sem_post(&mutex_);
    }
    // Node.h:87-97
    bool notifySourceReadComplete(size_t index)
    {
        sem_wait_retry(&mutex_);
        source_read_required_ &= ~(1u << index);
        bool reads_finished = source_read_required_ == 0;
        sem_post(&mutex_);
        return reads_finished;
    }
    // Node.h:102-121
    int acquireSlot(size_t &index)
    {
        sem_wait_retry(&mutex_);
        if (source_slots_ == (1u << NUM_SLOTS) - 1) { sem_post(&mutex_); return -1; }
        index = 0;
        while (source_slots_ & (1u << index)) ++index;
        source_slots_ |= (1u << index);
        source_ref_count_ = __builtin_popcount(source_slots_);
        sem_post(&mutex_);
        return 0;
    }
    // Node.h:123-135.  One deliberate hardening over the reference: a source that detaches while
    // it still owes a read of the current token no longer stalls the sink forever -- its pending
    // read is cancelled, and if it was the last one outstanding the write barrier is released.
    int releaseSlot(size_t index)
    {
        if (index >= NUM_SLOTS) return -1;
        sem_wait_retry(&mutex_);
        const uint32_t bit = 1u << index;
        const bool owed = (source_read_required_ & bit) != 0;
        source_slots_ &= ~bit;
        source_read_required_ &= ~bit;
        source_ref_count_ = __builtin_popcount(source_slots_);
        const bool release_writer = owed && source_read_required_ == 0;
        sem_post(&mutex_);
        if (release_writer) sem_post(&write_barrier);
        return 0;
    }
    size_t source_ref_count() const { return source_ref_count_; }

    // (the reference's switch has no `case 5`, Node.h:154-163: its 6th slot throws; here all ten work)
    sem_t &read_barrier(size_t index)
    {
        if (index >= NUM_SLOTS || !(source_slots_ & (1u << index)))
            throw std::runtime_error("Requested index refers to a SOURCE that is not bound to this node.");
        return rb_[index];
    }

    sem_t write_barrier;

private:
    static void sem_wait_retry(sem_t *s) { while (sem_wait(s) != 0 && errno == EINTR) {} }

    std::atomic<uint32_t> ready_;
    std::atomic<int32_t> sink_state_;
    volatile uint32_t source_slots_;
    volatile uint32_t source_read_required_;
    volatile size_t source_ref_count_;
    volatile uint64_t write_number_;
    sem_t mutex_;
    sem_t rb_[NUM_SLOTS];
};

inline bool remove_segment(const std::string &name) { return detail::Segment::remove(name); }

}  // namespace native

namespace detail {

// what sits at the start of "<addr>_obj": which token type the segment carries
struct ObjHeader {
    std::atomic<uint32_t> ready;
    char type_name[60];   // the Itanium-mangled name the reference keys its object by (Sink.h:203,267)
};
constexpr size_t kObjDataOffset = 64;
static_assert(sizeof(ObjHeader) <= kObjDataOffset, "ObjHeader");

template <typename T> struct TypeName;
template <> struct TypeName<Position2D> { static const char *get() { return "N3oat10Position2DE"; } };
template <> struct TypeName<SharedFrameHeader> { static const char *get() { return "N3oat17SharedFrameHeaderE"; } };

inline native::Node *open_node(Segment &seg, const std::string &node_address)
{
    const bool created = seg.open_or_create(node_address, 1024 + sizeof(native::Node));
    native::Node *n = (native::Node *)seg.base();
    if (created) n->construct(); else n->wait_constructed();
    return n;
}

}  // namespace detail

namespace native {

// --------------------------------------------------------------------------- SinkBase --
template <typename T>
class SinkBase {
public:
    SinkBase() = default;
    virtual ~SinkBase()
    {
        if (bound_) {
            node_->set_sink_state(NodeState::END);
            if (node_->source_ref_count() == 0) {
                detail::Segment::remove(node_address_);
                detail::Segment::remove(obj_address_);
            }
        }
    }
    // Sink.h:93-116: blocks (10 ms slices, honouring `quit`) until every attached source has read
    void wait()
    {
        if (!bound_) throw std::runtime_error("Sink must be bound before calling wait()");
        if (did_wait_need_post_) throw std::runtime_error("wait() called when post() was required.");
        while (node_->source_ref_count() > 0 && !detail::timed_wait_10ms(&node_->write_barrier) && !quit) {}
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
        node_ = detail::open_node(node_shmem_, node_address_);
        if (node_->sink_state() != NodeState::UNDEFINED)
            throw std::runtime_error("Requested SINK address, '" + address + "', is not available.");
    }
    void finish_bind()
    {
        auto *h = (detail::ObjHeader *)obj_shmem_.base();
        strncpy(h->type_name, detail::TypeName<T>::get(), sizeof(h->type_name) - 1);
        h->ready.store(detail::kMagic, std::memory_order_release);
        node_->set_sink_state(NodeState::SINK_BOUND);
        bound_ = true;
    }
    std::string address_, node_address_, obj_address_;
    detail::Segment node_shmem_, obj_shmem_;
    Node *node_{nullptr};
    T *sh_object_{nullptr};
    bool bound_{false};

private:
    bool did_wait_need_post_{false};
};

template <typename T>
class Sink : public SinkBase<T> {
public:
    // Sink.h:164-207
    template <typename... Targs>
    void bind(const std::string &address, Targs... args)
    {
        this->bind_node(address);
        this->obj_shmem_.create_only(this->obj_address_, 1024 + sizeof(T));
        this->sh_object_ = new (this->obj_shmem_.base() + detail::kObjDataOffset) T(args...);
        this->finish_bind();
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
    // Sink.h:232-272
    void bind(const std::string &address, size_t bytes)
    {
        bind_node(address);
        obj_shmem_.create_only(obj_address_, 1024 + sizeof(SharedFrameHeader) + bytes + sizeof(uint64_t));
        sh_object_ = new (obj_shmem_.base() + detail::kObjDataOffset) SharedFrameHeader();
        bytes_ = bytes;
        finish_bind();
    }
    // Sink.h:274-298: Sample and pixel blocks live in the segment, referenced by offsets
    Frame retrieve(size_t rows, size_t cols, int type, PixelColor color)
    {
        if (!bound_) throw std::runtime_error("SINK must be bound before shared frame is retrieved.");
        if (rows * cols * color_bytes(color) > bytes_) throw std::runtime_error("frame does not fit the bound sink");
        const ptrdiff_t sample_off = detail::kObjDataOffset + 64;          // after the 48-byte header
        const ptrdiff_t data_off = sample_off + 64;                        // after the 40-byte Sample
        new (obj_shmem_.base() + sample_off) Sample();
        sh_object_->setParameters(data_off, sample_off, rows, cols, type, color);
        sh_object_->params_.bytes = rows * cols * color_bytes(color);
        return Frame(rows, cols, color, obj_shmem_.base() + data_off, obj_shmem_.base() + sample_off);
    }

private:
    size_t bytes_{0};
};

// ------------------------------------------------------------------------- SourceBase --
template <typename T>
class SourceBase {
public:
    SourceBase() = default;
    virtual ~SourceBase()
    {
        if (state_ >= SourceState::TOUCHED || state_ == SourceState::ERR_TYPEMIS) node_->releaseSlot(slot_index_);
        if (node_ != nullptr && node_->source_ref_count() == 0 && node_->sink_state() != NodeState::SINK_BOUND) {
            detail::Segment::remove(node_address_);
            detail::Segment::remove(obj_address_);
        }
    }
    // Source.h:114-147
    void touch(const std::string &address)
    {
        if (state_ != SourceState::VIRGIN) throw std::runtime_error("A source can only connect a single time to a single node.");
        address_ = address;
        node_address_ = address + "_node";
        obj_address_ = address + "_obj";
        node_ = detail::open_node(node_shmem_, node_address_);
        if (node_->acquireSlot(slot_index_) < 0) { state_ = SourceState::ERR_NODEFULL; return; }
        state_ = SourceState::TOUCHED;
    }
    // Source.h:187-215
    NodeState wait()
    {
        if (state_ < SourceState::TOUCHED) throw std::runtime_error("Source must have touched node before calling wait()");
        if (did_wait_need_post_) throw std::runtime_error("wait() called when post() was required.");
        while (!detail::timed_wait_10ms(&node_->read_barrier(slot_index_)) && !quit) {
            if (node_->sink_state() == NodeState::END) break;
        }
        did_wait_need_post_ = true;
        return node_->sink_state();
    }
    // Source.h:217-232
    void post()
    {
        if (state_ < SourceState::CONNECTED) throw std::runtime_error("source must be connected before calling post()");
        if (!did_wait_need_post_) throw std::runtime_error("post() called when wait() was required.");
        if (node_->notifySourceReadComplete(slot_index_)) sem_post(&node_->write_barrier);
        did_wait_need_post_ = false;
    }
    SourceState state() const { return state_; }
    uint64_t write_number() const { return node_->write_number(); }
    // Would wait() return at once?  (A new token has been posted for this slot, or the sink has ended.)
    // Not in the reference: the batched tracker uses it to choose between taking the next frames and
    // publishing a finished result first.
    bool token_waiting()
    {
        if (state_ < SourceState::TOUCHED) return false;
        int v = 0;
        sem_getvalue(&node_->read_barrier(slot_index_), &v);
        return v > 0 || node_->sink_state() == NodeState::END;
    }

protected:
    // Source.h:149-185 (common part of both connect()s)
    SourceState connect_object()
    {
        if (state_ != SourceState::TOUCHED) throw std::runtime_error("A source can only connect() after it has touch()ed a node.");
        if (node_->sink_state() != NodeState::SINK_BOUND) {
            if (wait() != NodeState::SINK_BOUND) return SourceState::ERR_CONNECT;
            sem_post(&node_->read_barrier(slot_index_));   // the "freebie": give the first token back
            did_wait_need_post_ = false;
        }
        obj_shmem_.open_only(obj_address_);
        auto *h = (detail::ObjHeader *)obj_shmem_.base();
        if (h->ready.load(std::memory_order_acquire) != detail::kMagic ||
            strncmp(h->type_name, detail::TypeName<T>::get(), sizeof(h->type_name)) != 0) {
            state_ = SourceState::ERR_TYPEMIS;
            throw std::runtime_error("Type mismatch: Source<T> can only connect to Node<T>.");
        }
        sh_object_ = (T *)(obj_shmem_.base() + detail::kObjDataOffset);
        state_ = SourceState::CONNECTED;
        return SourceState::CONNECTED;
    }
    std::string address_, node_address_, obj_address_;
    detail::Segment node_shmem_, obj_shmem_;
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
    SourceState connect()
    {
        auto rc = connect_object();
        if (rc != SourceState::CONNECTED) return rc;
        auto p = sh_object_->params();
        frame_ = Frame(p.rows, p.cols, p.color, obj_shmem_.base() + sh_object_->data(), obj_shmem_.base() + sh_object_->sample());
        parameters_ = p;
        parameters_.bytes = frame_.bytes();
        return rc;
    }
    // Source.h:300-313
    SourceState connect(PixelColor color)
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

}  // namespace native

}  // namespace oat

// ---- which transport the components use ----
// OAT_SHMEM_BOOST (set by host/Makefile where <boost/interprocess/managed_shared_memory.hpp> exists): stock Oat's
// Boost.Interprocess bytes, i.e. these binaries attach to unmodified oat-* processes; otherwise the native one.
#if defined(OAT_SHMEM_BOOST)
#include "shmemdf_boost.hpp"
#if !defined(OAT_HAVE_BOOST_INTERPROCESS)
#error "OAT_SHMEM_BOOST was requested but <boost/interprocess/managed_shared_memory.hpp> is not available"
#endif
namespace oat { using namespace stock; }
#else
namespace oat { using namespace native; }
#endif
