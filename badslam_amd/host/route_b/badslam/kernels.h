// badslam/kernels.h -- STAND-IN for the reference's applications/badslam/src/badslam/kernels.h (B/kernels.h:94-495), used only to
// compile and test the Route-B shim (../kernels_hip.cc) without the reference tree: the declarations below restate the
// SIGNATURES of the free functions DirectBA's alternating scheme calls -- same names, same argument order and meaning as the
// reference header -- over this repository's libvis-style types (cuda_buffer.h, keyframe.h).  In the reference tree the real
// header is used and this file is not needed.  Not declared: the visualisation, image-pyramid odometry and PCG entry points
// (B/kernels.h:145-152, 190-236, 352-495), which the shim does not cover (Route B).
#pragma once

#include "../../keyframe.h"

namespace vis {

typedef hipStream_t cudaStream_t;                 // libvis/src/libvis/cuda: the reference's stream and texture handles
typedef hipTextureHandle_t cudaTextureObject_t;   // gfx950 has no texture unit: the handle names the colour CUDABuffer

struct float4_ { float x, y, z, w; };
struct CUDAMatrix3x4 {                            // B/cuda_matrix.cuh:37-141: three float4 rows
  float4_ row0, row1, row2;
  CUDAMatrix3x4() : row0{1, 0, 0, 0}, row1{0, 1, 0, 0}, row2{0, 0, 1, 0} {}
  explicit CUDAMatrix3x4(const float m[12]) : row0{m[0], m[1], m[2], m[3]}, row1{m[4], m[5], m[6], m[7]}, row2{m[8], m[9], m[10], m[11]} {}
};
struct PoseEstimationHelperBuffers {};            // B/kernels.h:47-58: scratch the HIP backend keeps inside its context
struct IntrinsicsOptimizationHelperBuffers {};    // B/kernels.h:60-89

// B/kernels.h:94-104
void DetermineSupportingSurfelsCUDA(cudaStream_t stream, const PinholeCamera4f& camera, const CUDAMatrix3x4& frame_T_global,
                                    const DepthParameters& depth_params, const CUDABuffer<u16>& depth_buffer,
                                    const CUDABuffer<u16>& normals_buffer, u32 surfels_size, CUDABuffer<float>* surfels,
                                    CUDABuffer<u32>** supporting_surfels);
// B/kernels.h:106-119
void DetermineSupportingSurfelsAndMergeSurfelsCUDA(cudaStream_t stream, float merge_dist_factor, const PinholeCamera4f& camera,
                                                   const CUDAMatrix3x4& frame_T_global, const DepthParameters& depth_params,
                                                   const CUDABuffer<u16>& depth_buffer, const CUDABuffer<u16>& normals_buffer,
                                                   u32 surfels_size, CUDABuffer<float>* surfels, CUDABuffer<u32>** supporting_surfels,
                                                   u32* surfel_count, CUDABufferPtr<u32>* deleted_count_buffer);
// B/kernels.h:121-147
void CreateSurfelsForKeyframeCUDA(cudaStream_t stream, int sparse_surfel_cell_size, bool filter_new_surfels, int min_observation_count,
                                  int keyframe_id, const vector<shared_ptr<Keyframe>>& keyframes, const PinholeCamera4f& color_camera,
                                  const PinholeCamera4f& depth_camera, const CUDAMatrix3x4& global_T_frame,
                                  const CUDAMatrix3x4& frame_T_global, const vector<CUDAMatrix3x4>& covis_T_frame,
                                  const DepthParameters& depth_params, const CUDABuffer<u16>& depth_buffer,
                                  const CUDABuffer<u16>& normals_buffer, const CUDABuffer<u16>& radius_buffer,
                                  const CUDABuffer<uchar4>& color_buffer, cudaTextureObject_t color_texture,
                                  CUDABuffer<u32>** supporting_surfels, void** new_surfels_temp_storage,
                                  usize* new_surfels_temp_storage_bytes, CUDABuffer<u8>* new_surfel_flag_vector,
                                  CUDABuffer<u32>* new_surfel_indices, u32 surfels_size, u32 surfel_count, u32* new_surfel_count,
                                  CUDABuffer<float>* surfels);
// B/kernels.h:158-176
void AccumulatePoseEstimationCoeffsCUDA(cudaStream_t stream, bool use_depth_residuals, bool use_descriptor_residuals,
                                        const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                                        const DepthParameters& depth_params, const CUDABuffer<u16>& depth_buffer,
                                        const CUDABuffer<u16>& normals_buffer, cudaTextureObject_t color_texture,
                                        const CUDAMatrix3x4& frame_T_global_estimate, u32 surfels_size,
                                        const CUDABuffer<float>& surfels, bool debug, u32* residual_count, float* residual_sum,
                                        float* H, float* b, PoseEstimationHelperBuffers* helper_buffers);
// B/kernels.h:238-245
void UpdateSurfelNormalsCUDA(cudaStream_t stream, const PinholeCamera4f& depth_camera, const DepthParameters& depth_params,
                             const vector<shared_ptr<Keyframe>>& keyframes, u32 surfels_size, const CUDABuffer<float>& surfels,
                             const CUDABuffer<u8>& active_surfels);
// B/kernels.h:247-257
void OptimizeGeometryIterationCUDA(cudaStream_t stream, bool use_depth_residuals, bool use_descriptor_residuals,
                                   const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                                   const DepthParameters& depth_params, const vector<shared_ptr<Keyframe>>& keyframes,
                                   u32 surfels_size, const CUDABuffer<float>& surfels, const CUDABuffer<u8>& active_surfels);
// B/kernels.h:259-273
void OptimizeIntrinsicsCUDA(cudaStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                            const vector<shared_ptr<Keyframe>>& keyframes, const PinholeCamera4f& color_camera,
                            const PinholeCamera4f& depth_camera, const DepthParameters& depth_params, u32 surfels_size,
                            const CUDABuffer<float>& surfels, PinholeCamera4f* out_color_camera, PinholeCamera4f* out_depth_camera,
                            float* a, CUDABufferPtr<float>* cfactor_buffer, IntrinsicsOptimizationHelperBuffers* buffers);
// B/kernels.h:275-282
void UpdateSurfelActivationCUDA(cudaStream_t stream, const PinholeCamera4f& camera, const DepthParameters& depth_params,
                                const vector<shared_ptr<Keyframe>>& keyframes, u32 surfels_size, CUDABuffer<float>* surfels,
                                CUDABuffer<u8>* active_surfels);
// B/kernels.h:284-293
void DeleteSurfelsAndUpdateRadiiCUDA(cudaStream_t stream, int min_observation_count, const PinholeCamera4f& camera,
                                     const DepthParameters& depth_params, const vector<shared_ptr<Keyframe>>& keyframes,
                                     u32* surfel_count, u32 surfels_size, CUDABuffer<float>* surfels,
                                     CUDABufferPtr<u32>* deleted_count_buffer);
// B/kernels.h:295-302
void CompactSurfelsCUDA(cudaStream_t stream, void** free_spots_temp_storage, usize* free_spots_temp_storage_bytes, u32 surfel_count,
                        u32* surfels_size, CUDABuffer_<float>* surfels, CUDABuffer_<u8>* active_surfels = nullptr);
// B/kernels.h:304-311
void AssignColorsCUDA(cudaStream_t stream, const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                      const DepthParameters& depth_params, const vector<shared_ptr<Keyframe>>& keyframes, u32 surfels_size,
                      CUDABuffer<float>* surfels);

}  // namespace vis
