// libvis_min.h -- the handful of libvis types that the DirectBA / Keyframe surface of BAD SLAM
// names ("libvis core"): integer typedefs, Image<T>, PinholeCamera4f, SE3f,
// Timer and glog-style logging macros.  Minimal stand-ins with the same member names so that code
// written against the reference (its tests, bad_slam.cc) reads the same against this backend.
// Reference: libvis/src/libvis/{libvis.h,image.h,camera.h,sophus.h,timing.h,logging.h}.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "../csrc/se3_device.h"

namespace vis {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef int8_t i8;
typedef int16_t i16;
typedef int32_t i32;
typedef size_t usize;
using std::shared_ptr;
using std::vector;
using std::mutex;
using std::lock_guard;

// The reference's signatures carry cudaStream_t; here a stream is the opaque hipStream_t value
// handed through the C ABI (include/badslam_hip.h), so host code needs no HIP headers.
typedef void* hipStream_t;

// ---- logging (libvis/src/libvis/logging.h: loguru behind glog-style macros) -------------------------
enum LogSeverity { INFO = 0, WARNING = 1, ERROR = 2, FATAL = 3 };
class LogMessage {
 public:
  LogMessage(LogSeverity severity, const char* file, int line) : severity_(severity) {
    static const char* names[] = {"INFO", "WARNING", "ERROR", "FATAL"};
    stream_ << "[" << names[severity] << " " << file << ":" << line << "] ";
  }
  ~LogMessage() {
    stream_ << "\n";
    if (severity_ >= min_severity()) fputs(stream_.str().c_str(), stderr);
    if (severity_ == FATAL) abort();
  }
  std::ostream& stream() { return stream_; }
  static int& min_severity() { static int s = WARNING; return s; }

 private:
  LogSeverity severity_;
  std::ostringstream stream_;
};
#define LOG(severity) ::vis::LogMessage(::vis::severity, __FILE__, __LINE__).stream()
#define CHECK(cond) if (!(cond)) LOG(FATAL) << "Check failed: " #cond " "
#define CHECK_EQ(a, b) CHECK((a) == (b))
#define CHECK_GT(a, b) CHECK((a) > (b))
#define CHECK_GE(a, b) CHECK((a) >= (b))
#define CHECK_LE(a, b) CHECK((a) <= (b))
#define CHECK_LT(a, b) CHECK((a) < (b))

// ---- small fixed-size vectors ---------------------------------------------------------------------------
struct Vec3u8 {
  u8 v[3];
  Vec3u8() : v{0, 0, 0} {}
  Vec3u8(u8 a, u8 b, u8 c) : v{a, b, c} {}
  u8& x() { return v[0]; } u8& y() { return v[1]; } u8& z() { return v[2]; }
  const u8& x() const { return v[0]; } const u8& y() const { return v[1]; } const u8& z() const { return v[2]; }
};
struct uchar4 { u8 x, y, z, w; };
struct uchar3 { u8 x, y, z; };

// ---- Image<T> (libvis/src/libvis/image.h): constructed (width, height), indexed (x, y) -------------------
template <typename T>
class Image {
 public:
  Image() : width_(0), height_(0) {}
  Image(u32 width, u32 height) : width_(width), height_(height), data_((size_t)width * height) {}
  void SetSize(u32 width, u32 height) { width_ = width; height_ = height; data_.assign((size_t)width * height, T()); }
  u32 width() const { return width_; }
  u32 height() const { return height_; }
  u32 stride() const { return width_ * sizeof(T); }   // bytes per row
  T* data() { return data_.data(); }
  const T* data() const { return data_.data(); }
  T* row(u32 y) { return data_.data() + (size_t)y * width_; }
  const T* row(u32 y) const { return data_.data() + (size_t)y * width_; }
  T& operator()(u32 x, u32 y) { return data_[(size_t)y * width_ + x]; }
  const T& operator()(u32 x, u32 y) const { return data_[(size_t)y * width_ + x]; }
  void SetTo(const T& value) { std::fill(data_.begin(), data_.end(), value); }

 private:
  u32 width_, height_;
  std::vector<T> data_;
};

// ---- PinholeCamera4f (libvis/src/libvis/camera.h:1740): fx, fy, cx, cy, pixel-corner convention ----------
class PinholeCamera4f {
 public:
  PinholeCamera4f() : width_(0), height_(0), p_{0, 0, 0, 0} {}
  PinholeCamera4f(int width, int height, const float* parameters) : width_(width), height_(height) {
    for (int i = 0; i < 4; ++i) p_[i] = parameters[i];
  }
  int width() const { return width_; }
  int height() const { return height_; }
  const float* parameters() const { return p_; }
  // Direction (z = 1) through a pixel given in the pixel-centre / pixel-corner convention.
  void UnprojectFromPixelCenterConv(float x, float y, float out[3]) const {
    out[0] = (x - (p_[2] - 0.5f)) / p_[0]; out[1] = (y - (p_[3] - 0.5f)) / p_[1]; out[2] = 1.f;
  }
  void UnprojectFromPixelCornerConv(float x, float y, float out[3]) const {
    out[0] = (x - p_[2]) / p_[0]; out[1] = (y - p_[3]) / p_[1]; out[2] = 1.f;
  }

 private:
  int width_, height_;
  float p_[4];
};

// ---- SE3f stored like Sophus (unit quaternion x,y,z,w + translation) --------------------------------------
class SE3f {
 public:
  SE3f() : v_{0, 0, 0, 1, 0, 0, 0} {}
  explicit SE3f(const float* qxyzw_t) { for (int i = 0; i < 7; ++i) v_[i] = qxyzw_t[i]; }
  static SE3f exp(const float tangent[6]) { SE3f r; bahip::se3_exp(tangent, r.v_); return r; }
  void log(float tangent[6]) const { bahip::se3_log(v_, tangent); }
  SE3f inverse() const { SE3f r; bahip::se3_inverse(v_, r.v_); return r; }
  SE3f operator*(const SE3f& o) const { SE3f r; bahip::se3_mul(v_, o.v_, r.v_); return r; }
  void matrix3x4(float m[12]) const { bahip::se3_matrix3x4(v_, m); }
  void rotationMatrix(float r[9]) const { bahip::se3_rotation(v_, r); }
  const float* translation() const { return v_ + 4; }
  const float* data() const { return v_; }
  float* data() { return v_; }

 private:
  float v_[7];
};

// ---- Timer (libvis/src/libvis/timing.h:47) -------------------------------------------------------------------
class Timer {
 public:
  explicit Timer(const char* = "") : start_(std::chrono::steady_clock::now()) {}
  double GetTimeSinceStart() const {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - start_).count();
  }

 private:
  std::chrono::steady_clock::time_point start_;
};

}  // namespace vis
