// cuda_buffer.h -- libvis CUDABuffer<T> semantics over HIP, through the C ABI only.
// Reference: libvis/src/libvis/cuda/cuda_buffer.h:45-129 (class), cuda_buffer.cuh:44-119 (the POD
// CUDABuffer_<T>), cuda_buffer_inl.h:36-240 (pitched allocation, whole / pitched / byte-range
// transfers, Clear, SetTo).  Constructed (height, width); element (y, x) at
// (char*)address + y*pitch + x*sizeof(T).  The class name is kept so that code written against
// the reference compiles unchanged; texture creation does not exist on gfx950 (no tex2D), the
// BA kernels sample the buffer directly.
#pragma once

#include "../../include/badslam_hip.h"
#include "libvis_min.h"

namespace vis {

#define BAHIP_CHECKED_CALL(expr)                                            \
  do {                                                                      \
    if ((expr) != 0) LOG(FATAL) << "HIP backend error: " << bahip_last_error(); \
  } while (0)

template <typename T>
struct CUDABuffer_ {
  T* address_ = nullptr;
  int height_ = 0, width_ = 0;
  size_t pitch_ = 0;
  T* address() const { return address_; }
  int width() const { return width_; }
  int height() const { return height_; }
  size_t pitch() const { return pitch_; }
};

template <typename T>
class CUDABuffer {
 public:
  CUDABuffer(int height, int width) {
    void* p = nullptr;
    size_t pitch = 0;
    BAHIP_CHECKED_CALL(bahip_malloc_pitch(&p, &pitch, (size_t)width * sizeof(T), (size_t)height));
    data_.address_ = static_cast<T*>(p);
    data_.height_ = height; data_.width_ = width; data_.pitch_ = pitch;
  }
  ~CUDABuffer() { if (data_.address_) bahip_free(data_.address_); }
  CUDABuffer(const CUDABuffer&) = delete;
  CUDABuffer& operator=(const CUDABuffer&) = delete;

  int width() const { return data_.width_; }
  int height() const { return data_.height_; }
  const CUDABuffer_<T>& ToCUDA() const { return data_; }
  CUDABuffer_<T>& ToCUDA() { return data_; }

  // The *Async transfers return with the copy in flight on `stream` when the host range is page-locked (bahip_host_alloc,
  // hipHostMalloc, hipHostRegister) -- cudaMemcpy2DAsync's behaviour, libvis/src/libvis/cuda/cuda_buffer_inl.h:73-90 -- and wait for
  // it when the range is pageable, where the CUDA runtime stages an upload and completes a download before it returns (a caller of
  // the reference that passes a std::vector and reads it right after DownloadAsync relies on exactly that).
  static void FinishTransfer(hipStream_t stream, const void* host, size_t bytes) {
    if (!bahip_host_is_pinned(host, bytes)) BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
  }
  // dense host array (width*height elements)
  void UploadAsync(hipStream_t stream, const T* host) {
    BAHIP_CHECKED_CALL(bahip_memcpy_2d_async(stream, data_.address_, data_.pitch_, host, (size_t)data_.width_ * sizeof(T),
                                             (size_t)data_.width_ * sizeof(T), (size_t)data_.height_, 1));
    FinishTransfer(stream, host, (size_t)data_.width_ * sizeof(T) * (size_t)data_.height_);
  }
  void UploadAsync(hipStream_t stream, const Image<T>& image) {
    CHECK_EQ((int)image.width(), data_.width_); CHECK_EQ((int)image.height(), data_.height_);
    BAHIP_CHECKED_CALL(bahip_memcpy_2d_async(stream, data_.address_, data_.pitch_, image.data(), image.stride(),
                                             (size_t)data_.width_ * sizeof(T), (size_t)data_.height_, 1));
    FinishTransfer(stream, image.data(), image.stride() * (size_t)data_.height_);
  }
  void DownloadAsync(hipStream_t stream, T* host) const {
    BAHIP_CHECKED_CALL(bahip_memcpy_2d_async(stream, host, (size_t)data_.width_ * sizeof(T), data_.address_, data_.pitch_,
                                             (size_t)data_.width_ * sizeof(T), (size_t)data_.height_, 2));
    FinishTransfer(stream, host, (size_t)data_.width_ * sizeof(T) * (size_t)data_.height_);
  }
  void DownloadAsync(hipStream_t stream, Image<T>* image) const {
    CHECK_EQ((int)image->width(), data_.width_); CHECK_EQ((int)image->height(), data_.height_);
    BAHIP_CHECKED_CALL(bahip_memcpy_2d_async(stream, image->data(), image->stride(), data_.address_, data_.pitch_,
                                             (size_t)data_.width_ * sizeof(T), (size_t)data_.height_, 2));
    FinishTransfer(stream, image->data(), image->stride() * (size_t)data_.height_);
  }
  // byte ranges relative to the start of the allocation (used to move single surfel rows,
  // B/direct_ba.cc:469, test_geometry_optimization_geometric_residual.cc:159-161)
  void UploadPartAsync(size_t start_bytes, size_t length_bytes, hipStream_t stream, const T* host) {
    BAHIP_CHECKED_CALL(bahip_memcpy_async(stream, reinterpret_cast<char*>(data_.address_) + start_bytes, host, length_bytes, 1));
    FinishTransfer(stream, host, length_bytes);
  }
  void DownloadPartAsync(size_t start_bytes, size_t length_bytes, hipStream_t stream, T* host) const {
    BAHIP_CHECKED_CALL(bahip_memcpy_async(stream, host, reinterpret_cast<const char*>(data_.address_) + start_bytes, length_bytes, 2));
    FinishTransfer(stream, host, length_bytes);
  }
  void Clear(T value, hipStream_t stream) {
    static_assert(sizeof(T) == 1 || sizeof(T) == 2 || sizeof(T) == 4, "Clear() supports 1/2/4-byte elements");
    u32 bits = 0;
    memcpy(&bits, &value, sizeof(T));
    BAHIP_CHECKED_CALL(bahip_fill_2d(stream, data_.address_, data_.pitch_, (int)sizeof(T), bits, data_.width_, data_.height_));
  }
  void SetTo(const CUDABuffer<T>& other, hipStream_t stream) {
    CHECK_EQ(other.width(), data_.width_); CHECK_EQ(other.height(), data_.height_);
    BAHIP_CHECKED_CALL(bahip_memcpy_2d_async(stream, data_.address_, data_.pitch_, other.data_.address_, other.data_.pitch_,
                                             (size_t)data_.width_ * sizeof(T), (size_t)data_.height_, 3));
  }

 private:
  CUDABuffer_<T> data_;
};

template <typename T> using CUDABufferPtr = shared_ptr<CUDABuffer<T>>;
template <typename T> using CUDABufferConstPtr = shared_ptr<const CUDABuffer<T>>;

}  // namespace vis
