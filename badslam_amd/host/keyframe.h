// keyframe.h -- vis::Keyframe with the reference's surface (B/keyframe.h:50-237, B/keyframe.cc:35-158;
// B/ = applications/badslam/src/badslam/): owns depth / normals / radius (u16) and colour (uchar4,
// w = luma) device buffers, the pose with both directions cached on every set, activation state,
// co-visibility list and min/max depth.  The reference additionally creates a bilinear
// cudaTextureObject_t over the colour buffer (B/keyframe.cc:67-73); gfx950 has no such sampling
// path, so color_texture() returns a handle to the colour buffer itself and the BA kernels filter
// in software.  Next to the four images the keyframe owns their tiled "BA planes"
// (bahip_frame_planes, include/badslam_hip.h), derived in the constructors like normals / radii /
// luma are; code that writes into the image buffers afterwards (the reference's tests do, through
// const_cast) must call RefreshPlanes().
#pragma once

#include "cuda_buffer.h"

namespace vis {

// B/surfel_projection.cuh:129-149
struct DepthParameters {
  CUDABuffer_<float> cfactor_buffer;
  float a = 0;
  float raw_to_float_depth = 0;
  float baseline_fx = 0;
  int sparse_surfel_cell_size = 1;
};

typedef const CUDABuffer<uchar4>* hipTextureHandle_t;   // stands in for cudaTextureObject_t

inline bahip_camera ToBahipCamera(const PinholeCamera4f& c) {
  bahip_camera r;
  r.fx = c.parameters()[0]; r.fy = c.parameters()[1]; r.cx = c.parameters()[2]; r.cy = c.parameters()[3];
  r.width = c.width(); r.height = c.height();
  return r;
}
inline bahip_depth_params ToBahipDepthParams(const DepthParameters& p) {
  bahip_depth_params r;
  r.a = p.a; r.raw_to_float_depth = p.raw_to_float_depth; r.baseline_fx = p.baseline_fx;
  r.sparse_surfel_cell_size = p.sparse_surfel_cell_size;
  r.cfactor = p.cfactor_buffer.address(); r.cfactor_pitch_bytes = (uint32_t)p.cfactor_buffer.pitch();
  r.cfactor_width = p.cfactor_buffer.width(); r.cfactor_height = p.cfactor_buffer.height();
  return r;
}

// A backend context for work that is not tied to a DirectBA (keyframe preprocessing, plane packing): one per (thread, stream),
// created on first use and kept -- a context is six allocations and a synchronisation, too much to pay per keyframe.
// Never destroyed (a few KB per thread and stream; tearing contexts down from thread-exit destructors would race the HIP
// runtime's own shutdown).
inline bahip_context* UtilityContext(hipStream_t stream) {
  thread_local std::vector<std::pair<hipStream_t, bahip_context*>> cache;
  for (const auto& entry : cache)
    if (entry.first == stream) return entry.second;
  bahip_context* ctx = nullptr;
  BAHIP_CHECKED_CALL(bahip_context_create(&ctx, stream));
  cache.emplace_back(stream, ctx);
  return ctx;
}

class Keyframe {
 public:
  enum class Activation { kActive = 0, kCovisibleActive = 1, kInactive = 2 };

  // From existing device buffers (B/keyframe.cc:35-79); contents are copied.
  Keyframe(hipStream_t stream, u32 frame_index, float min_depth, float max_depth, const CUDABuffer<u16>& depth_buffer,
           const CUDABuffer<u16>& normals_buffer, const CUDABuffer<u16>& radius_buffer, const CUDABuffer<uchar4>& color_buffer,
           const SE3f& global_T_frame)
      : frame_index_(frame_index), last_active_in_ba_iteration_(-1), last_covis_in_ba_iteration_(-1), min_depth_(min_depth),
        max_depth_(max_depth), depth_buffer_(depth_buffer.height(), depth_buffer.width()),
        normals_buffer_(normals_buffer.height(), normals_buffer.width()),
        radius_buffer_(radius_buffer.height(), radius_buffer.width()), color_buffer_(color_buffer.height(), color_buffer.width()) {
    CHECK_GT(min_depth, 0.f) << "Keyframe min depth must be larger than 0 since the frustum checks do not work properly otherwise.";
    depth_buffer_.SetTo(depth_buffer, stream);
    normals_buffer_.SetTo(normals_buffer, stream);
    radius_buffer_.SetTo(radius_buffer, stream);
    color_buffer_.SetTo(color_buffer, stream);
    activation_ = Activation::kActive;
    set_global_T_frame(global_T_frame);
    RefreshPlanes(stream);
  }

  // Convenience constructor from raw depth + RGB (B/keyframe.cc:81-158): luma, normals, radii +
  // isolated pixel removal, min/max depth, in the reference's order.
  Keyframe(hipStream_t stream, u32 frame_index, const DepthParameters& depth_params, const PinholeCamera4f& depth_camera,
           const Image<u16>& depth_image, const Image<Vec3u8>& color_image, const SE3f& global_tr_frame)
      : frame_index_(frame_index), last_active_in_ba_iteration_(-1), last_covis_in_ba_iteration_(-1),
        depth_buffer_(depth_image.height(), depth_image.width()), normals_buffer_(depth_image.height(), depth_image.width()),
        radius_buffer_(depth_image.height(), depth_image.width()), color_buffer_(color_image.height(), color_image.width()) {
    bahip_context* ctx = UtilityContext(stream);
    const int W = depth_image.width(), H = depth_image.height();
    CUDABuffer<u8> rgb_buffer(color_image.height(), color_image.width() * 3);
    rgb_buffer.UploadAsync(stream, reinterpret_cast<const u8*>(color_image.data()));
    BAHIP_CHECKED_CALL(bahip_compute_brightness(ctx, rgb_buffer.ToCUDA().address(), (uint32_t)rgb_buffer.ToCUDA().pitch(),
                                                reinterpret_cast<uint8_t*>(color_buffer_.ToCUDA().address()),
                                                (uint32_t)color_buffer_.ToCUDA().pitch(), color_image.width(), color_image.height()));
    CUDABuffer<u16> depth_temp(H, W);
    depth_buffer_.UploadAsync(stream, depth_image);
    const bahip_camera cam = ToBahipCamera(depth_camera);
    const bahip_depth_params dp = ToBahipDepthParams(depth_params);
    BAHIP_CHECKED_CALL(bahip_compute_normals(ctx, &cam, &dp, depth_buffer_.ToCUDA().address(), (uint32_t)depth_buffer_.ToCUDA().pitch(),
                                             depth_temp.ToCUDA().address(), (uint32_t)depth_temp.ToCUDA().pitch(),
                                             normals_buffer_.ToCUDA().address(), (uint32_t)normals_buffer_.ToCUDA().pitch()));
    radius_buffer_.Clear(0, stream);
    BAHIP_CHECKED_CALL(bahip_compute_point_radii_and_remove_isolated_pixels(
        ctx, &cam, depth_params.raw_to_float_depth, depth_temp.ToCUDA().address(), (uint32_t)depth_temp.ToCUDA().pitch(),
        radius_buffer_.ToCUDA().address(), (uint32_t)radius_buffer_.ToCUDA().pitch(), depth_buffer_.ToCUDA().address(),
        (uint32_t)depth_buffer_.ToCUDA().pitch()));
    BAHIP_CHECKED_CALL(bahip_compute_min_max_depth(ctx, depth_temp.ToCUDA().address(), (uint32_t)depth_temp.ToCUDA().pitch(), W, H,
                                                   depth_params.raw_to_float_depth, &min_depth_, &max_depth_));
    BAHIP_CHECKED_CALL(bahip_frame_planes_create(ctx, W, H, color_image.width(), color_image.height(), &planes_));
    const bahip_frame images = ToBahipFrame();
    BAHIP_CHECKED_CALL(bahip_frame_planes_update(ctx, planes_, &images));
    set_global_T_frame(global_tr_frame);
    activation_ = Activation::kActive;
  }

  ~Keyframe() { bahip_frame_planes_destroy(planes_); }
  Keyframe(const Keyframe&) = delete;
  Keyframe& operator=(const Keyframe&) = delete;

  // Re-derives the BA planes from the current contents of the image buffers.
  void RefreshPlanes(hipStream_t stream) {
    bahip_context* ctx = UtilityContext(stream);
    if (!planes_)
      BAHIP_CHECKED_CALL(bahip_frame_planes_create(ctx, depth_buffer_.width(), depth_buffer_.height(), color_buffer_.width(),
                                                   color_buffer_.height(), &planes_));
    const bahip_frame images = ToBahipFrame();
    BAHIP_CHECKED_CALL(bahip_frame_planes_update(ctx, planes_, &images));
  }

  void SetID(int id) { id_ = id; }
  int id() const { return id_; }
  int last_active_in_ba_iteration() const { return last_active_in_ba_iteration_; }
  void SetLastActiveInBAIteration(int iteration) { last_active_in_ba_iteration_ = iteration; }
  int last_covis_in_ba_iteration() const { return last_covis_in_ba_iteration_; }
  void SetLastCovisInBAIteration(int iteration) { last_covis_in_ba_iteration_ = iteration; }
  float min_depth() const { return min_depth_; }
  float max_depth() const { return max_depth_; }
  vector<int>& co_visibility_list() { return co_visibility_list_; }
  const vector<int>& co_visibility_list() const { return co_visibility_list_; }
  Activation activation() const { return activation_; }
  void SetActivation(Activation activation) { activation_ = activation; }
  u32 frame_index() const { return frame_index_; }

  // B/keyframe.h:160-172: both directions are cached whenever the pose is set.
  void set_global_T_frame(const SE3f& global_T_frame) {
    global_T_frame_ = global_T_frame;
    frame_T_global_ = global_T_frame.inverse();
  }
  void set_frame_T_global(const SE3f& frame_T_global) {
    frame_T_global_ = frame_T_global;
    global_T_frame_ = frame_T_global.inverse();
  }
  const SE3f& global_T_frame() const { return global_T_frame_; }
  const SE3f& frame_T_global() const { return frame_T_global_; }

  const CUDABuffer<u16>& depth_buffer() const { return depth_buffer_; }
  const CUDABuffer<u16>& normals_buffer() const { return normals_buffer_; }
  const CUDABuffer<u16>& radius_buffer() const { return radius_buffer_; }
  const CUDABuffer<uchar4>& color_buffer() const { return color_buffer_; }
  hipTextureHandle_t color_texture() const { return &color_buffer_; }

  bahip_frame ToBahipFrame() const {
    bahip_frame f;
    f.depth = depth_buffer_.ToCUDA().address(); f.depth_pitch_bytes = (uint32_t)depth_buffer_.ToCUDA().pitch();
    f.normals = normals_buffer_.ToCUDA().address(); f.normals_pitch_bytes = (uint32_t)normals_buffer_.ToCUDA().pitch();
    f.radius = radius_buffer_.ToCUDA().address(); f.radius_pitch_bytes = (uint32_t)radius_buffer_.ToCUDA().pitch();
    f.color = reinterpret_cast<uint8_t*>(color_buffer_.ToCUDA().address()); f.color_pitch_bytes = (uint32_t)color_buffer_.ToCUDA().pitch();
    f.planes = planes_;
    return f;
  }

 private:
  int id_ = -1;
  u32 frame_index_;
  int last_active_in_ba_iteration_;
  int last_covis_in_ba_iteration_;
  float min_depth_ = 0, max_depth_ = 0;
  SE3f global_T_frame_, frame_T_global_;
  vector<int> co_visibility_list_;
  Activation activation_ = Activation::kActive;
  CUDABuffer<u16> depth_buffer_;
  CUDABuffer<u16> normals_buffer_;
  CUDABuffer<u16> radius_buffer_;
  CUDABuffer<uchar4> color_buffer_;
  bahip_frame_planes* planes_ = nullptr;
};

}  // namespace vis
