// badslam/kernels.h -- STAND-IN for the reference's applications/badslam/src/badslam/kernels.h (B/kernels.h:94-495), used only to
// compile and test the Route-B shim (../kernels_hip.cc) without the reference tree: the declarations below restate the
// SIGNATURES of the free functions DirectBA's alternating scheme calls -- same names, same argument order and meaning as the
// reference header -- over this repository's libvis-style types (cuda_buffer.h, keyframe.h).  In the reference tree the real
// header is used and this file is not needed.  tests/test_cpu_route_b.py holds every declaration below against the text of the
// real header (parameter lists compared token for token) wherever /root/reference is present.  Not declared: the
// visualisation and image-pyramid odometry entry points (UpdateVisualizationBuffersCUDA, *FromImagesCUDA, Calibrate*,
// Downsample*, AssignDescriptorColorsCUDA, PCGDebugVerifyResultCUDA), which the shim does not cover (Route B).
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

// B/kernels.h:94-103
void DetermineSupportingSurfelsCUDA(cudaStream_t stream, const PinholeCamera4f& camera, const CUDAMatrix3x4& frame_T_global,
                                    const DepthParameters& depth_params, const CUDABuffer<u16>& depth_buffer,
                                    const CUDABuffer<u16>& normals_buffer, u32 surfels_size, CUDABuffer<float>* surfels,
                                    CUDABuffer<u32>** supporting_surfels);
// B/kernels.h:105-117
void DetermineSupportingSurfelsAndMergeSurfelsCUDA(cudaStream_t stream, float merge_dist_factor, const PinholeCamera4f& camera,
                                                   const CUDAMatrix3x4& frame_T_global, const DepthParameters& depth_params,
                                                   const CUDABuffer<u16>& depth_buffer, const CUDABuffer<u16>& normals_buffer,
                                                   u32 surfels_size, CUDABuffer<float>* surfels, CUDABuffer<u32>** supporting_surfels,
                                                   u32* surfel_count, CUDABufferPtr<u32>* deleted_count_buffer);
// B/kernels.h:119-145
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
// B/kernels.h:156-174
void AccumulatePoseEstimationCoeffsCUDA(cudaStream_t stream, bool use_depth_residuals, bool use_descriptor_residuals,
                                        const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                                        const DepthParameters& depth_params, const CUDABuffer<u16>& depth_buffer,
                                        const CUDABuffer<u16>& normals_buffer, cudaTextureObject_t color_texture,
                                        const CUDAMatrix3x4& frame_T_global_estimate, u32 surfels_size,
                                        const CUDABuffer<float>& surfels, bool debug, u32* residual_count, float* residual_sum,
                                        float* H, float* b, PoseEstimationHelperBuffers* helper_buffers);
// B/kernels.h:225-232
void UpdateSurfelNormalsCUDA(cudaStream_t stream, const PinholeCamera4f& depth_camera, const DepthParameters& depth_params,
                             const vector<shared_ptr<Keyframe>>& keyframes, u32 surfels_size, const CUDABuffer<float>& surfels,
                             const CUDABuffer<u8>& active_surfels);
// B/kernels.h:234-244
void OptimizeGeometryIterationCUDA(cudaStream_t stream, bool use_depth_residuals, bool use_descriptor_residuals,
                                   const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                                   const DepthParameters& depth_params, const vector<shared_ptr<Keyframe>>& keyframes,
                                   u32 surfels_size, const CUDABuffer<float>& surfels, const CUDABuffer<u8>& active_surfels);
// B/kernels.h:246-260
void OptimizeIntrinsicsCUDA(cudaStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                            const vector<shared_ptr<Keyframe>>& keyframes, const PinholeCamera4f& color_camera,
                            const PinholeCamera4f& depth_camera, const DepthParameters& depth_params, u32 surfels_size,
                            const CUDABuffer<float>& surfels, PinholeCamera4f* out_color_camera, PinholeCamera4f* out_depth_camera,
                            float* a, CUDABufferPtr<float>* cfactor_buffer, IntrinsicsOptimizationHelperBuffers* buffers);
// B/kernels.h:262-269
void UpdateSurfelActivationCUDA(cudaStream_t stream, const PinholeCamera4f& camera, const DepthParameters& depth_params,
                                const vector<shared_ptr<Keyframe>>& keyframes, u32 surfels_size, CUDABuffer<float>* surfels,
                                CUDABuffer<u8>* active_surfels);
// B/kernels.h:271-280
void DeleteSurfelsAndUpdateRadiiCUDA(cudaStream_t stream, int min_observation_count, const PinholeCamera4f& camera,
                                     const DepthParameters& depth_params, const vector<shared_ptr<Keyframe>>& keyframes,
                                     u32* surfel_count, u32 surfels_size, CUDABuffer<float>* surfels,
                                     CUDABufferPtr<u32>* deleted_count_buffer);
// B/kernels.h:292-299
void CompactSurfelsCUDA(cudaStream_t stream, void** free_spots_temp_storage, usize* free_spots_temp_storage_bytes, u32 surfel_count,
                        u32* surfels_size, CUDABuffer_<float>* surfels, CUDABuffer_<u8>* active_surfels = nullptr);
// B/kernels.h:301-308
void AssignColorsCUDA(cudaStream_t stream, const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                      const DepthParameters& depth_params, const vector<shared_ptr<Keyframe>>& keyframes, u32 surfels_size,
                      CUDABuffer<float>* surfels);

// ---- the PCG scheme (B/kernels.h:397-491).  The PODs its entry points take are those of B/surfel_projection.cuh:40-207, restated
// with the members the shim reads (the reference's versions carry device-side methods as well). ---------------------------------
typedef float PCGScalar;                          // B/kernels.cuh:62
struct PixelCornerProjector { float fx, fy, cx, cy; };                       // B/surfel_projection.cuh:40-60
struct PixelCenterUnprojector { float fx_inv, fy_inv, cx_inv, cy_inv; };     // B/surfel_projection.cuh:86-126
struct DepthToColorPixelCorner { float fx, fy, cx, cy; int width, height; }; // B/surfel_projection.cuh:180-187
struct SurfelProjectionParameters {                                           // B/surfel_projection.cuh:151-178
  CUDABuffer_<float> surfels;
  CUDABuffer_<u16> depth_buffer;
  CUDABuffer_<u16> normals_buffer;
  DepthParameters depth_params;
  PixelCornerProjector projector;
  PixelCenterUnprojector center_unprojector;
  CUDAMatrix3x4 frame_T_global;
  u32 surfels_size;
};

// B/kernels.h:397-416
void PCGInitCUDA(
    cudaStream_t stream,
    const SurfelProjectionParameters& s,
    const DepthToColorPixelCorner& depth_to_color,
    const PixelCenterUnprojector& depth_unprojector,
    const PixelCornerProjector& color_projector,
    cudaTextureObject_t color_texture,
    u32 kf_pose_unknown_index,
    u32 surfel_unknown_start_index,
    bool optimize_poses,
    bool optimize_geometry,
    bool use_depth_residuals,
    bool use_descriptor_residuals,
    bool optimize_depth_intrinsics,
    bool optimize_color_intrinsics,
    u32 depth_intrinsics_unknown_start_index,
    u32 color_intrinsics_unknown_start_index,
    CUDABuffer_<PCGScalar>* pcg_r,
    CUDABuffer_<PCGScalar>* pcg_M,
    u32 surfels_size);

// B/kernels.h:418-428
void PCGInit2CUDA(
    cudaStream_t stream,
    u32 unknown_count,
    u32 a_unknown_index,
    float a,
    const CUDABuffer_<PCGScalar>& pcg_r,
    const CUDABuffer_<PCGScalar>& pcg_M,
    CUDABuffer_<PCGScalar>* pcg_delta,
    CUDABuffer_<PCGScalar>* pcg_g,
    CUDABuffer_<PCGScalar>* pcg_p,
    CUDABuffer_<PCGScalar>* pcg_alpha_n);

// B/kernels.h:430-452
void PCGStep1CUDA(
    cudaStream_t stream,
    u32 unknown_count,
    const SurfelProjectionParameters& s,
    const DepthToColorPixelCorner& depth_to_color,
    const PixelCenterUnprojector& depth_unprojector,
    const PixelCornerProjector& color_projector,
    cudaTextureObject_t color_texture,
    u32 kf_pose_unknown_index,
    u32 surfel_unknown_start_index,
    bool optimize_poses,
    bool optimize_geometry,
    bool use_depth_residuals,
    bool use_descriptor_residuals,
    bool optimize_depth_intrinsics,
    bool optimize_color_intrinsics,
    u32 depth_intrinsics_unknown_start_index,
    u32 a_unknown_index,
    u32 color_intrinsics_unknown_start_index,
    CUDABuffer_<PCGScalar>* pcg_p,
    CUDABuffer_<PCGScalar>* pcg_g,
    CUDABuffer_<PCGScalar>* pcg_alpha_d,
    u32 surfels_size);

// B/kernels.h:454-465
void PCGStep2CUDA(
    cudaStream_t stream,
    u32 unknown_count,
    u32 a_unknown_index,
    const CUDABuffer_<PCGScalar>& pcg_r,
    const CUDABuffer_<PCGScalar>& pcg_M,
    CUDABuffer_<PCGScalar>* pcg_delta,
    CUDABuffer_<PCGScalar>* pcg_g,
    CUDABuffer_<PCGScalar>* pcg_p,
    CUDABuffer_<PCGScalar>* pcg_alpha_n,
    CUDABuffer_<PCGScalar>* pcg_alpha_d,
    CUDABuffer_<PCGScalar>* pcg_beta_n);

// B/kernels.h:467-473
void PCGStep3CUDA(
    cudaStream_t stream,
    u32 unknown_count,
    CUDABuffer_<PCGScalar>* pcg_g,
    CUDABuffer_<PCGScalar>* pcg_p,
    CUDABuffer_<PCGScalar>* pcg_alpha_n,
    CUDABuffer_<PCGScalar>* pcg_beta_n);

// B/kernels.h:483-489
void UpdateSurfelsFromPCGDeltaCUDA(
    cudaStream_t stream,
    u32 surfels_size,
    CUDABuffer_<float>* surfels,
    bool use_descriptor_residuals,
    u32 surfel_unknown_start_index,
    const CUDABuffer_<PCGScalar>& pcg_delta);

// B/kernels.h:491-495
void UpdateCFactorsFromPCGDeltaCUDA(
    cudaStream_t stream,
    CUDABuffer_<float>* cfactor_buffer,
    u32 cfactor_unknown_start_index,
    const CUDABuffer_<PCGScalar>& pcg_delta);

}  // namespace vis
