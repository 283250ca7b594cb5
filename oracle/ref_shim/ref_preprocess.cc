// ref_preprocess.cc -- the REFERENCE's own frame / keyframe preprocessing KERNELS and its surfel compaction, compiled for the host.
// TEST INFRASTRUCTURE (see oracle/oracle.h).  Three .cu files are included from where they lie under /root/reference
// (oracle/Makefile passes the include paths; nothing is copied):
//   B/cuda_depth_processing.cu   BilateralFilteringAndDepthCutoffCUDA (BadSlam::PreprocessFrame, B/bad_slam.cc:697-706),
//                                ComputeNormalsCUDA, ComputePointRadiiAndRemoveIsolatedPixelsCUDA, ComputeMinMaxDepthCUDA
//                                (the Keyframe constructor, B/keyframe.cc:96-144)
//   B/cuda_image_processing.cu   ComputeBrightnessCUDA (B/keyframe.cc:97-103; the Sobel and upscaling kernels of the file compile
//                                but are not on the path)
//   B/kernel_compact_surfels.cu  CompactSurfelsCUDA (PerformBASchemeEndTasks, B/direct_ba.cc:634-650)
// (B/ = applications/badslam/src/badslam/) with their own host wrappers; the grids run on the host through the stand-in launcher of
// ref_shim/libvis/cuda/cuda_auto_tuner.h, whose thread-local block state lives in ref_kernels.cc.  The entry points below take
// plain dense arrays and call the wrappers in the order the reference's callers do.
#define REF_BLOCK_COLLECTIVES 1
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <limits>
#include <vector>

#include <libvis/cuda/cuda_auto_tuner.h>

// B/kernel_compact_surfels.cu:205 checks a size with the logging library's macro
#define CHECK_GE(a, b) do { if (!((a) >= (b))) std::abort(); } while (false)

#include "badslam/cuda_depth_processing.cu"
#include "badslam/cuda_image_processing.cu"
#include "badslam/kernel_compact_surfels.cu"

using namespace vis;

namespace {

template <typename T>
CUDABuffer_<T> dense(T* data, int height, int width) { return CUDABuffer_<T>(data, height, width, (size_t)width * sizeof(T)); }

PixelCenterUnprojector unprojector_of(const float cam[4]) {   // as B/surfel_projection.h:54-71 builds it from the pixel-corner parameters
  return PixelCenterUnprojector(PixelCornerProjector(cam[0], cam[1], cam[2], cam[3]));
}

}  // namespace

extern "C" {

// B/cuda_depth_processing.cu:100-128
void ref_bilateral_filter_and_depth_cutoff(float sigma_xy, float sigma_value, float radius_factor, uint16_t max_depth, float raw_to_float_depth,
                                           uint16_t* in_depth, int width, int height, uint16_t* out_depth) {
  CUDABuffer_<u16> out = dense(out_depth, height, width);
  BilateralFilteringAndDepthCutoffCUDA(nullptr, sigma_xy, sigma_value, radius_factor, max_depth, raw_to_float_depth, dense(in_depth, height, width), &out);
}

// B/cuda_image_processing.cu:178-193
void ref_compute_brightness(uint8_t* rgb, int width, int height, uint8_t* rgba) {
  CUDABuffer_<uchar4> out = dense(reinterpret_cast<uchar4*>(rgba), height, width);
  ComputeBrightnessCUDA(nullptr, dense(reinterpret_cast<uchar3*>(rgb), height, width), &out);
}

// The depth half of the Keyframe constructor (B/keyframe.cc:111-144): normals (pixels without four valid neighbours dropped),
// radii (isolated pixels dropped), the depth range of the image the normals pass left.  cam: fx, fy, cx, cy in the pixel-corner
// convention.  radius is cleared first (the reference leaves the words of dropped pixels as allocated).
void ref_keyframe_depth_preprocessing(const float cam[4], float a, float raw_to_float_depth, float baseline_fx, int cell, float* cfactor, int cf_width,
                                      int cf_height, uint16_t* depth_image, int width, int height, uint16_t* out_depth, uint16_t* out_normals,
                                      uint16_t* out_radius, uint16_t* depth_after_normals, float* min_depth, float* max_depth) {
  DepthParameters dp;
  dp.cfactor_buffer = dense(cfactor, cf_height, cf_width);
  dp.a = a; dp.raw_to_float_depth = raw_to_float_depth; dp.baseline_fx = baseline_fx; dp.sparse_surfel_cell_size = cell;
  const PixelCenterUnprojector unprojector = unprojector_of(cam);
  const size_t pixels = (size_t)width * height;
  CUDABuffer_<u16> after_normals = dense(depth_after_normals, height, width), normals = dense(out_normals, height, width);
  ComputeNormalsCUDA(nullptr, unprojector, dp, dense(depth_image, height, width), &after_normals, &normals);
  std::memset(out_radius, 0, pixels * sizeof(uint16_t));
  CUDABuffer_<u16> radius = dense(out_radius, height, width), final_depth = dense(out_depth, height, width);
  ComputePointRadiiAndRemoveIsolatedPixelsCUDA(nullptr, unprojector, raw_to_float_depth, after_normals, &radius, &final_depth);
  float init[2] = {std::numeric_limits<float>::infinity(), 0.f}, result[2] = {0.f, 0.f};   // B/cuda_depth_processing.cc:35-44
  CUDABuffer_<float> result_buffer = dense(result, 1, 2);
  ref_thread0_last = true;   // the block's minimum / maximum is what thread 0 gets back (cub/cub.cuh)
  ComputeMinMaxDepthCUDA(nullptr, after_normals, raw_to_float_depth, dense(init, 1, 2), &result_buffer, min_depth, max_depth);
  ref_thread0_last = false;
}

// B/kernel_compact_surfels.cu:159-279 on 17 rows of `capacity` floats; active may be null.  Returns the new surfels_size.
uint32_t ref_compact_surfels(float* surfel_rows, uint32_t capacity, uint32_t surfels_size, uint32_t surfel_count, uint8_t* active) {
  CUDABuffer_<float> surfels(surfel_rows, kSurfelAttributeCount, (int)capacity, (size_t)capacity * sizeof(float));
  CUDABuffer_<u8> active_buffer(active, 1, (int)capacity, (size_t)capacity);
  void* temp_storage = nullptr;
  usize temp_storage_bytes = 0;
  u32 size = surfels_size;
  CompactSurfelsCUDA(nullptr, &temp_storage, &temp_storage_bytes, surfel_count, &size, &surfels, active ? &active_buffer : nullptr);
  return size;
}

// the stand-in's binary32 -> binary16 conversion, so that a test can compare it with another implementation over every pattern it cares about
uint16_t ref_float_to_half_bits(float value) { return __half_as_ushort(__float2half_rn(value)); }

}  // extern "C"
