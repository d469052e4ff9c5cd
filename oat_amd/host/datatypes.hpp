// datatypes.hpp -- wire types at the drop-in boundary, mirrored from jonnew/Oat without OpenCV.
//
//   oat::Sample            lib/datatypes/Sample.h:35-123        (40 bytes)
//   oat::Position2D        lib/datatypes/Position2D.h:65-156    (280 bytes)
//   oat::SharedFrameHeader lib/shmemdf/SharedFrameHeader.h:31-97 (48 bytes)
//   oat::PixelColor        lib/datatypes/Color.h:29-34
//   oat::Frame             lib/datatypes/Frame.h:41-146 (a cv::Mat header over shm pixels +
//                          Sample* + colour; here a plain view, because cv::Mat is not available)
//
// Sizes and member offsets are static_asserted against the x86-64 GCC layout of the reference
// structs, so a token written by these binaries is byte-compatible with what the
// stock components read out of their Position2D / Sample / SharedFrameHeader objects.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace oat {

// lib/datatypes/Color.h:29-37
enum PixelColor : int { PIX_BINARY = 0, PIX_GREY = 1, PIX_BGR = 2, PIX_HSV = 3 };
constexpr int kCV_8UC1 = 0;
constexpr int kCV_8UC3 = 16;
inline int color_bytes(PixelColor c) { return (c == PIX_BGR || c == PIX_HSV) ? 3 : 1; }
inline int color_cvtype(PixelColor c) { return (c == PIX_BGR || c == PIX_HSV) ? kCV_8UC3 : kCV_8UC1; }
inline const char *color_str(PixelColor c)
{
    switch (c) {
        case PIX_BINARY: return "BINARY";
        case PIX_GREY: return "GREY";
        case PIX_BGR: return "BGR";
        case PIX_HSV: return "HSV";
    }
    throw std::runtime_error("Invalid color.");
}

// Color.h:65-77
inline PixelColor str_color(const std::string &s)
{
    if (s == "BINARY") return PIX_BINARY;
    if (s == "GREY") return PIX_GREY;
    if (s == "BGR") return PIX_BGR;
    if (s == "HSV") return PIX_HSV;
    throw std::runtime_error("Invalid color.");
}

// lib/datatypes/Sample.h:35-123
class Sample {
public:
    Sample() = default;
    explicit Sample(double period_sec) : period_sec_(period_sec), rate_hz_(1.0 / period_sec) {}

    uint64_t incrementCount()
    {
        microseconds_ += period_microseconds_;
        return ++count_;
    }
    uint64_t incrementCount(int64_t usec)
    {
        microseconds_ = usec;
        return ++count_;
    }
    void set_rate_hz(double value)
    {
        rate_hz_ = value;
        period_sec_ = 1.0 / value;
        period_microseconds_ = (int64_t)(period_sec_ * 1e6);   // duration_cast truncates
    }
    uint64_t count() const { return count_; }
    int64_t microseconds() const { return microseconds_; }
    double period_sec() const { return period_sec_; }
    int64_t period_microseconds() const { return period_microseconds_; }
    double rate_hz() const { return rate_hz_; }

    uint64_t count_{0};
    int64_t microseconds_{0};
    double period_sec_{0.0};
    int64_t period_microseconds_{0};
    double rate_hz_{0.0};
};
static_assert(sizeof(Sample) == 40, "oat::Sample layout");
static_assert(offsetof(Sample, microseconds_) == 8 && offsetof(Sample, period_sec_) == 16 &&
              offsetof(Sample, period_microseconds_) == 24 && offsetof(Sample, rate_hz_) == 32, "oat::Sample offsets");

struct Point2D { double x{0}, y{0}; };

enum class DistanceUnit : int { PIXELS = 0, WORLD = 1 };

// lib/datatypes/Position2D.h:65-156
class Position2D {
public:
    explicit Position2D(const std::string &label)
    {
        strncpy(label_, label.c_str(), sizeof(label_));
        label_[sizeof(label_) - 1] = '\0';
    }
    // Copy all but label (and homography), like the reference's operator= (Position2D.h:84-105)
    Position2D &operator=(const Position2D &p)
    {
        if (this == &p) return *this;
        unit_of_length_ = p.unit_of_length_;
        sample_ = p.sample_;
        position_valid = p.position_valid;
        velocity_valid = p.velocity_valid;
        heading_valid = p.heading_valid;
        position = p.position;
        velocity = p.velocity;
        heading = p.heading;
        region_valid = p.region_valid;
        strncpy(region, p.region, sizeof(region));
        region[sizeof(region) - 1] = '\0';
        return *this;
    }
    Position2D(const Position2D &) = default;

    void set_sample(const Sample &s) { sample_ = s; }
    const Sample &sample() const { return sample_; }
    // Position2D.h:138-142
    void setCoordSystem(DistanceUnit value, const double homography[9])
    {
        unit_of_length_ = value;
        for (int i = 0; i < 9; ++i) homography_[i] = homography[i];
    }
    const char *label() const { return label_; }

    static constexpr size_t REGION_LEN{10};
    bool region_valid{false};
    char region[REGION_LEN]{0};
    bool position_valid{false};
    bool velocity_valid{false};
    bool heading_valid{false};
    Point2D position;
    Point2D velocity;
    Point2D heading;
    char label_[100]{0};
    DistanceUnit unit_of_length_{DistanceUnit::PIXELS};
    Sample sample_;
    double homography_[9]{1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0};
};
static_assert(sizeof(Position2D) == 280, "oat::Position2D layout");
static_assert(offsetof(Position2D, region) == 1 && offsetof(Position2D, position_valid) == 11 &&
              offsetof(Position2D, velocity_valid) == 12 && offsetof(Position2D, heading_valid) == 13 &&
              offsetof(Position2D, position) == 16 && offsetof(Position2D, velocity) == 32 &&
              offsetof(Position2D, heading) == 48 && offsetof(Position2D, label_) == 64 &&
              offsetof(Position2D, unit_of_length_) == 164 && offsetof(Position2D, sample_) == 168 &&
              offsetof(Position2D, homography_) == 208, "oat::Position2D offsets");

// ---- position wire formats (lib/datatypes/Position2D.cpp:24-96, Position2D.h:170-233) ----

// numpy structured dtype of one packed record, 82 bytes (Position2D::NPY_DTYPE / NPY_DTYPE_BYTES)
constexpr size_t kNpyDtypeBytes = 82;
inline const char *npy_dtype()
{
    return "[('tick', '<u8'),('usec', '<u8'),('unit', '<i4'),('pos_ok', '<i1'),('pos_xy', 'f8', (2)),"
           "('vel_ok', '<i1'),('vel_xy', 'f8', (2)),('head_ok', '<i1'),('head_xy', 'f8', (2)),"
           "('reg_ok', '<i1'),('reg', 'a10')]";
}

// oat::packPosition (Position2D.cpp:37-96): the record the reference's recorder writes to .npy files
inline std::vector<char> packPosition(const Position2D &p)
{
    std::vector<char> pack;
    pack.reserve(kNpyDtypeBytes);
    auto put = [&pack](const void *v, size_t n) { pack.insert(pack.end(), (const char *)v, (const char *)v + n); };
    const uint64_t tick = p.sample_.count(), usec = (uint64_t)p.sample_.microseconds();
    const int32_t unit = (int32_t)p.unit_of_length_;
    const char pok = p.position_valid ? 1 : 0, vok = p.velocity_valid ? 1 : 0, hok = p.heading_valid ? 1 : 0,
               rok = p.region_valid ? 1 : 0;
    put(&tick, 8); put(&usec, 8); put(&unit, 4);
    put(&pok, 1); put(&p.position.x, 8); put(&p.position.y, 8);
    put(&vok, 1); put(&p.velocity.x, 8); put(&p.velocity.y, 8);
    put(&hok, 1); put(&p.heading.x, 8); put(&p.heading.y, 8);
    put(&rok, 1); put(p.region, Position2D::REGION_LEN);
    return pack;
}

// oat::serializePosition (Position2D.h:170-233) with rapidjson's SetMaxDecimalPlaces(5) rendered by
// truncation to 5 places minus trailing zeros -- same field set and order, same "only when valid" rule.
inline std::string jsonNumber(double v)
{
    char buf[80];
    snprintf(buf, sizeof buf, "%.9f", v);            // rapidjson TRUNCATES to 5 places, it does not round
    std::string s(buf);
    size_t dot = s.find('.');
    if (dot != std::string::npos) s.erase(dot + 6);
    if (dot != std::string::npos) {
        size_t last = s.find_last_not_of('0');
        s.erase(last == dot ? dot + 2 : last + 1);       // rapidjson keeps one digit after the point
    }
    return s;
}
inline std::string serializePosition(const Position2D &p, bool verbose = false)
{
    std::string o = "{\"tick\":" + std::to_string(p.sample_.count()) + ",\"usec\":" +
                    std::to_string((unsigned long long)p.sample_.microseconds()) + ",\"unit\":" +
                    std::to_string((int)p.unit_of_length_);
    auto pair = [](const char *key, const Point2D &v) {
        return std::string(",\"") + key + "\":[" + jsonNumber(v.x) + "," + jsonNumber(v.y) + "]";
    };
    o += std::string(",\"pos_ok\":") + ((p.position_valid || verbose) ? "true" : "false");
    if (p.position_valid || verbose) o += pair("pos_xy", p.position);
    o += std::string(",\"vel_ok\":") + ((p.velocity_valid || verbose) ? "true" : "false");
    if (p.velocity_valid || verbose) o += pair("vel_xy", p.velocity);
    o += std::string(",\"head_ok\":") + (p.heading_valid ? "true" : "false");
    if (p.heading_valid || verbose) o += pair("head_xy", p.heading);
    o += std::string(",\"reg_ok\":") + (p.region_valid ? "true" : "false");
    if (p.region_valid || verbose) o += std::string(",\"reg\":\"") + p.region + "\"";
    return o + "}";
}

// lib/shmemdf/SharedFrameHeader.h:31-37
struct FrameParams {
    size_t cols{0};
    size_t rows{0};
    int type{0};
    PixelColor color{PIX_BGR};
    size_t bytes{0};
};

// lib/shmemdf/SharedFrameHeader.h:50-96.  handle_t is an offset from the segment base.
class SharedFrameHeader {
public:
    using handle_t = ptrdiff_t;
    handle_t sample() const { return sample_; }
    handle_t data() const { return data_; }
    FrameParams params() const { return params_; }
    void setParameters(handle_t data, handle_t sample, size_t rows, size_t cols, int type, PixelColor color)
    {
        data_ = data; sample_ = sample;
        params_.rows = rows; params_.cols = cols; params_.type = type; params_.color = color;
    }
    FrameParams params_;
    handle_t data_{0};
    handle_t sample_{0};
};
static_assert(sizeof(SharedFrameHeader) == 48, "oat::SharedFrameHeader layout");
static_assert(offsetof(SharedFrameHeader, data_) == 32 && offsetof(SharedFrameHeader, sample_) == 40,
              "oat::SharedFrameHeader offsets");

// lib/datatypes/Frame.h:41-146 without cv::Mat: packed pixels (no row padding) + Sample + colour.
// A Frame either owns its storage (internal copies made by components) or views shared memory.
class Frame {
public:
    Frame() = default;
    Frame(size_t rows, size_t cols, PixelColor color)           // owning
        : rows_(rows), cols_(cols), color_(color), own_(rows * cols * color_bytes(color)), own_sample_()
    {
        data_ = own_.data();
        sample_ = &own_sample_;
    }
    Frame(size_t rows, size_t cols, PixelColor color, void *data, void *sample)   // view over shm
        : rows_(rows), cols_(cols), color_(color), data_((uint8_t *)data), sample_((Sample *)sample) {}

    Frame(const Frame &) = delete;
    Frame &operator=(const Frame &) = delete;
    Frame(Frame &&o) noexcept { *this = std::move(o); }
    Frame &operator=(Frame &&o) noexcept
    {
        rows_ = o.rows_; cols_ = o.cols_; color_ = o.color_;
        const bool owned = !o.own_.empty() && o.data_ == o.own_.data();
        own_ = std::move(o.own_);
        own_sample_ = o.own_sample_;
        data_ = owned ? own_.data() : o.data_;
        sample_ = (o.sample_ == &o.own_sample_) ? &own_sample_ : o.sample_;
        return *this;
    }

    // Frame::copyTo (Frame.h:113-118): pixels, sample and colour
    void copyTo(Frame &dst) const
    {
        if (dst.data_ == nullptr || dst.bytes() != bytes()) {
            dst.rows_ = rows_; dst.cols_ = cols_;
            dst.own_.assign(bytes(), 0);
            dst.data_ = dst.own_.data();
            dst.sample_ = &dst.own_sample_;
        }
        dst.color_ = color_;
        memcpy(dst.data_, data_, bytes());
        *dst.sample_ = *sample_;
    }

    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    size_t bytes() const { return rows_ * cols_ * color_bytes(color_); }
    PixelColor color() const { return color_; }
    void set_color(PixelColor c) { color_ = c; }
    uint8_t *data() { return data_; }
    const uint8_t *data() const { return data_; }
    Sample &sample() { return *sample_; }
    const Sample &sample() const { return *sample_; }

private:
    size_t rows_{0}, cols_{0};
    PixelColor color_{PIX_BGR};
    uint8_t *data_{nullptr};
    Sample *sample_{nullptr};
    std::vector<uint8_t> own_;
    Sample own_sample_;
};

}  // namespace oat
