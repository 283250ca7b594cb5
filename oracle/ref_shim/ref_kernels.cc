// ref_kernels.cc -- the REFERENCE's own geometry-step and activation KERNELS, compiled for the host and run over whole scenes.
// TEST INFRASTRUCTURE (see oracle/oracle.h).  The two .cu files are included from where they lie under /root/reference
// (oracle/Makefile passes the include paths; nothing is copied):
//   B/kernel_opt_geometry.cu        ResetSurfelAccum*, AccumulateSurfelNormalOptimizationCoeffs, UpdateSurfelNormal,
//                                   AccumulateSurfelPositionOptimizationCoeffsFromDepthResidual, UpdateSurfelPosition,
//                                   AccumulateSurfelPositionAndDescriptorOptimizationCoeffs<use_depth>, UpdateSurfelPositionAndDescriptor
//   B/kernel_surfel_activation.cu   SetSurfelInactive, DetermineActiveSurfels
//   B/kernel_assign_colors.cu       ResetSurfelForColorAssignment, AccumulateColorObservations, AssignColors
//   B/kernel_delete_surfels.cu      ResetSurfelAccumForSurfelDeletionAndRadiusUpdate, CountObservationsAndFreeSpaceViolations,
//                                   MarkDeletedSurfels
//   B/kernel_supporting_surfels.cu  DetermineSupportingSurfels<merge_surfels>
//   B/kernel_opt_pose.cu            AccumulatePoseEstimationCoeffs<block_width, debug, use_depth, use_descriptors> with B/gauss_newton.cuh
//   B/kernel_pcg.cu                 PCGInit (r = -J^T W F, M = diag(J^T W J): every unknown block of the PCG scheme), PCGInit2, PCGStep1-3,
//                                   UpdateSurfelsFromPCGDelta
//   B/kernel_opt_intrinsics.cu      AccumulateIntrinsicsCoefficients<block_width, colour, depth>, ComputeIntrinsicsIntermediateMatrices,
//                                   SolveForPixelIntrinsicsUpdate (the whole intrinsics step of the alternating scheme)
//   B/kernel_create_surfels.cu      CreateSurfelsForKeyframeCUDASerializing, ..._CountNewSurfels, WriteNewSurfelIndexAndInitializeObservations,
//                                   CountObservationsForNewSurfels, FilterNewSurfels, CreateSurfelsForKeyframeCUDACreationAppend
// (B/ = applications/badslam/src/badslam/) with their own Call...CUDAKernel wrappers; the grid runs on the host through the
// stand-in CUDA_AUTO_TUNE_1D of ref_shim/libvis/cuda/cuda_auto_tuner.h.  What this file adds is the sequence of calls the
// reference's host drivers make -- B/kernel_opt_geometry.cc:80-201 (OptimizeGeometryIterationCUDA),
// B/kernel_surfel_activation.cc:38-66 (UpdateSurfelActivationCUDA), B/kernel_assign_colors.cc:38-74 (AssignColorsCUDA) and
// B/kernel_delete_surfels.cc:38-98 (DeleteSurfelsAndUpdateRadiiCUDAImpl), B/kernel_supporting_surfels.cc:38-108
// (DetermineSupportingSurfelsCUDAImpl), B/kernel_opt_pose.cc:38-96 (AccumulatePoseEstimationCoeffsCUDA), B/direct_ba_pcg.cc:276-365 (the
// unknown layout and the PCGInitCUDA loop of BundleAdjustmentPCG), B/kernel_opt_intrinsics.cc:39-104 (the accumulation of OptimizeIntrinsicsCUDA), B/direct_ba.cc:340-405 + B/kernel_create_surfels.cc:40-197 (CreateSurfelsForKeyframe) --
// over plain arrays instead of Keyframe objects, with
// the projector PODs built as B/surfel_projection.h:54-124 builds them.  The reference accumulates a surfel's sums keyframe by
// keyframe, one launch after the other; that order is kept.
#define REF_BLOCK_COLLECTIVES 1
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include <libvis/cuda/cuda_auto_tuner.h>

thread_local RefDim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local RefBlockState ref_block;
bool ref_launch_sequential = false;
bool ref_thread0_last = false;
bool ref_barrier_passes = false;
void ref_syncthreads() {
  if (!ref_barrier_passes) return;
  RefBlockState& state = ref_block;
  if (state.vote_call < (int)state.votes.size()) { ++state.vote_call; return; }
  state.pending = true;
  throw RefVotePending();
}
// the block vote of the stand-in launcher (cuda_auto_tuner.h): a resolved vote is replayed, the first unresolved one takes the
// thread's predicate and ends the thread's pass
int ref_syncthreads_or(int predicate) {
  RefBlockState& state = ref_block;
  if (state.vote_call < (int)state.votes.size()) return state.votes[state.vote_call++];
  state.pending_or |= predicate != 0 ? 1 : 0;
  state.pending = true;
  throw RefVotePending();
}

#include "badslam/kernel_opt_geometry.cu"
#include "badslam/kernel_surfel_activation.cu"
#include "badslam/kernel_assign_colors.cu"
#include "badslam/kernel_delete_surfels.cu"
#include "badslam/kernel_supporting_surfels.cu"
#include "badslam/kernel_create_surfels.cu"
#include "badslam/kernel_opt_pose.cu"
#include "badslam/kernel_pcg.cu"
#include "badslam/kernel_opt_intrinsics.cu"

using namespace vis;

extern "C" {

struct ref_ba_keyframe {
  uint16_t* depth; uint16_t* normals; uint8_t* rgba;   // dense, row-major; rgba.w = luma
  uint16_t* radius;                                    // binary16 bits of the squared point radius (B/keyframe.h:227-231)
  float frame_T_global[12];
  float global_T_frame[12];
  float global_R_frame[9];
  int32_t activation;                                  // 0 kActive, 1 kCovisibleActive, 2 kInactive (B/keyframe.h:54-67)
  int32_t pad;
};

struct ref_ba_scene {
  float depth_cam[4], color_cam[4];                    // fx, fy, cx, cy in the pixel-corner convention
  int width, height, color_width, color_height;
  float a, raw_to_float_depth, baseline_fx;
  int cell;
  float* cfactor; int cf_width, cf_height;
  float* surfel_rows; uint32_t capacity, surfels_size; // 17 rows of `capacity` floats
  uint8_t* active;                                     // kSurfelActiveFlag per surfel
  int quantize_texture_weights;
  int num_keyframes;
  const ref_ba_keyframe* keyframes;
};

}  // extern "C"

namespace {

struct Bound {
  CUDABuffer_<float> surfels;
  CUDABuffer_<u8> active;
  DepthParameters dp;
  PixelCornerProjector depth_projector, color_projector;
  PixelCenterUnprojector unprojector;
  DepthToColorPixelCorner d2c;
  explicit Bound(const ref_ba_scene* sc)
      : surfels(sc->surfel_rows, kSurfelAttributeCount, (int)sc->capacity, (size_t)sc->capacity * sizeof(float)),
        active(sc->active, 1, (int)sc->capacity, (size_t)sc->capacity),
        depth_projector(sc->depth_cam[0], sc->depth_cam[1], sc->depth_cam[2], sc->depth_cam[3]),
        color_projector(sc->color_cam[0], sc->color_cam[1], sc->color_cam[2], sc->color_cam[3]),
        unprojector(depth_projector) {                                    // B/surfel_projection.h:54-71
    dp.cfactor_buffer = CUDABuffer_<float>(sc->cfactor, sc->cf_height, sc->cf_width, (size_t)sc->cf_width * sizeof(float));
    dp.a = sc->a; dp.raw_to_float_depth = sc->raw_to_float_depth; dp.baseline_fx = sc->baseline_fx; dp.sparse_surfel_cell_size = sc->cell;
    d2c.fx = sc->color_cam[0] / sc->depth_cam[0];                         // B/surfel_projection.h:100-124
    d2c.fy = sc->color_cam[1] / sc->depth_cam[1];
    d2c.cx = -1 * sc->color_cam[0] * sc->depth_cam[2] / sc->depth_cam[0] + sc->color_cam[2];
    d2c.cy = -1 * sc->color_cam[1] * sc->depth_cam[3] / sc->depth_cam[1] + sc->color_cam[3];
    d2c.width = sc->color_width; d2c.height = sc->color_height;
  }
};

CUDAMatrix3x4 pose_of(const ref_ba_keyframe& kf) {
  CUDAMatrix3x4 F;
  F.row0 = make_float4(kf.frame_T_global[0], kf.frame_T_global[1], kf.frame_T_global[2], kf.frame_T_global[3]);
  F.row1 = make_float4(kf.frame_T_global[4], kf.frame_T_global[5], kf.frame_T_global[6], kf.frame_T_global[7]);
  F.row2 = make_float4(kf.frame_T_global[8], kf.frame_T_global[9], kf.frame_T_global[10], kf.frame_T_global[11]);
  return F;
}

// The projection parameters of one launch over keyframe `kf`, and a guard around that launch.  ProjectSurfelToImage (B/util.cuh:83-118)
// converts the projected pixel to int BEFORE testing it: CUDA's conversion saturates and turns NaN into 0, so on the GPU a surfel
// marked deleted (x = NaN: pixel (0, 0), whose depth is always invalid) or projecting beyond 2^31 (INT_MAX >= width) is simply not
// associated; the host's conversion yields INT_MIN for both, which passes every test and reads depth_buffer(py, INT_MIN).  To give
// the kernels the outcome they have on their own platform, such surfels are parked behind the keyframe's camera for the duration of
// the launch (local z = -1: rejected by the first test of the association, nothing is read or written for them) and put back, bit
// for bit, when the temporary dies at the end of the launch expression.
struct GuardedProjection {
  SurfelProjectionParameters params;
  CUDABuffer_<float> surfels;
  std::vector<uint32_t> parked;        // surfel index, then the three original words
  operator SurfelProjectionParameters() const { return params; }
  GuardedProjection(const SurfelProjectionParameters& p, const CUDABuffer_<float>& s) : params(p), surfels(s) {}
  GuardedProjection(GuardedProjection&& other) : params(other.params), surfels(other.surfels), parked(std::move(other.parked)) { other.parked.clear(); }
  GuardedProjection(const GuardedProjection&) = delete;
  ~GuardedProjection() {
    for (size_t e = 0; e + 3 < parked.size(); e += 4)
      for (int c = 0; c < 3; ++c) std::memcpy(&surfels(kSurfelX + c, parked[e]), &parked[e + 1 + c], sizeof(uint32_t));
  }
};

GuardedProjection projection_of(const ref_ba_scene* sc, const Bound& b, const ref_ba_keyframe& kf) {   // B/surfel_projection.h:69-84
  CUDABuffer_<u16> depth_buffer(kf.depth, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
  CUDABuffer_<u16> normals_buffer(kf.normals, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
  const CUDAMatrix3x4 F = pose_of(kf);
  GuardedProjection guarded(SurfelProjectionParameters(b.surfels, depth_buffer, normals_buffer, b.dp, b.depth_projector, b.unprojector, F, sc->surfels_size), b.surfels);
  // behind the camera: p with F p = (0, 0, -1), i.e. p = R^T ((0, 0, -1) - t)
  const float3 d = make_float3(-F.row0.w, -F.row1.w, -1.f - F.row2.w);
  const float safe[3] = {F.row0.x * d.x + F.row1.x * d.y + F.row2.x * d.z, F.row0.y * d.x + F.row1.y * d.y + F.row2.y * d.z,
                         F.row0.z * d.x + F.row1.z * d.y + F.row2.z * d.z};
  for (uint32_t i = 0; i < sc->surfels_size; ++i) {
    const float3 global_position = SurfelGetPosition(b.surfels, i);
    float3 local_position;
    if (!F.MultiplyIfResultZIsPositive(global_position, &local_position)) continue;   // rejected there anyway
    const float2 p = b.depth_projector.Project(local_position);
    if (p.x < 2147483648.f && p.y < 2147483648.f) continue;                           // (false for NaN too)
    guarded.parked.push_back(i);
    for (int c = 0; c < 3; ++c) {
      uint32_t word;
      std::memcpy(&word, &guarded.surfels(kSurfelX + c, i), sizeof(word));
      guarded.parked.push_back(word);
      guarded.surfels(kSurfelX + c, i) = safe[c];
    }
  }
  return guarded;
}

constexpr int kInactive = 2;

}  // namespace

extern "C" {

// flags[i] |= 1 where surfel i, projected into some keyframe, lands beyond the int range: CUDA's float -> int conversion
// saturates there (the pixel is outside the image), the host's yields INT_MIN and the reference's bounds test lets it through
// (see pixel_outside_int_range in ref_entry.cc).  The launch guard of projection_of gives the kernels CUDA's outcome for such surfels;
// this function lets a test see whether a scene has any.
void ref_flag_pairs_outside_int_range(const ref_ba_scene* sc, uint8_t* flags) {
  const Bound b(sc);
  for (int k = 0; k < sc->num_keyframes; ++k) {
    const CUDAMatrix3x4 F = pose_of(sc->keyframes[k]);
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < (long long)sc->surfels_size; ++i) {
      const float3 global_position = SurfelGetPosition(b.surfels, (unsigned int)i);
      float3 local_position;
      if (!F.MultiplyIfResultZIsPositive(global_position, &local_position)) continue;
      const float2 p = b.depth_projector.Project(local_position);
      if (!(p.x < 2147483648.f && p.y < 2147483648.f)) flags[i] |= 1;
    }
  }
}

// B/kernel_surfel_activation.cc:38-66
void ref_update_surfel_activation(const ref_ba_scene* sc) {
  if (sc->surfels_size == 0) return;
  const Bound b(sc);
  CallSetSurfelInactiveKernel(nullptr, sc->surfels_size, b.active);
  for (int k = 0; k < sc->num_keyframes; ++k) {
    if (sc->keyframes[k].activation != 0) continue;   // kActive only
    CallDetermineActiveSurfelsKernel(nullptr, projection_of(sc, b, sc->keyframes[k]), b.active);
  }
}

// B/kernel_opt_geometry.cc:80-201
void ref_optimize_geometry_iteration(const ref_ba_scene* sc, int use_depth_residuals, int use_descriptor_residuals) {
  if (sc->surfels_size == 0) return;
  const Bound b(sc);
  // --- normals (:108-134)
  CallResetSurfelAccum0to3CUDAKernel(nullptr, sc->surfels_size, b.surfels, b.active);
  for (int k = 0; k < sc->num_keyframes; ++k) {
    const ref_ba_keyframe& kf = sc->keyframes[k];
    if (kf.activation == kInactive) continue;
    CUDAMatrix3x3 R;
    R.row0 = make_float3(kf.global_R_frame[0], kf.global_R_frame[1], kf.global_R_frame[2]);
    R.row1 = make_float3(kf.global_R_frame[3], kf.global_R_frame[4], kf.global_R_frame[5]);
    R.row2 = make_float3(kf.global_R_frame[6], kf.global_R_frame[7], kf.global_R_frame[8]);
    CallAccumulateSurfelNormalOptimizationCoeffsCUDAKernel(nullptr, projection_of(sc, b, kf), R, b.active);
  }
  CallUpdateSurfelNormalCUDAKernel(nullptr, sc->surfels_size, b.surfels, b.active);

  if (!use_descriptor_residuals) {
    // --- position from the depth residual (:137-168)
    CallResetSurfelAccum0to1CUDAKernel(nullptr, sc->surfels_size, b.surfels, b.active);
    for (int k = 0; k < sc->num_keyframes; ++k) {
      const ref_ba_keyframe& kf = sc->keyframes[k];
      if (kf.activation == kInactive) continue;
      RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
      CallAccumulateSurfelPositionOptimizationCoeffsFromDepthResidualCUDAKernel(nullptr, projection_of(sc, b, kf), b.unprojector, b.d2c, sc->color_cam[0],
                                                                               sc->color_cam[1], reinterpret_cast<cudaTextureObject_t>(&tex), b.active);
    }
    CallUpdateSurfelPositionCUDAKernel(nullptr, sc->surfels_size, b.surfels, b.active);
  } else {
    // --- position and descriptors jointly (:169-199)
    CallResetSurfelAccumCUDAKernel(nullptr, sc->surfels_size, b.surfels, b.active);
    for (int k = 0; k < sc->num_keyframes; ++k) {
      const ref_ba_keyframe& kf = sc->keyframes[k];
      if (kf.activation == kInactive) continue;
      RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
      AccumulateSurfelPositionAndDescriptorOptimizationCoeffsCUDAKernel(nullptr, projection_of(sc, b, kf), b.unprojector, b.d2c, b.color_projector,
                                                                        reinterpret_cast<cudaTextureObject_t>(&tex), b.active, use_depth_residuals != 0);
    }
    CallUpdateSurfelPositionAndDescriptorCUDAKernel(nullptr, sc->surfels_size, b.surfels, b.active);
  }
}

// B/kernel_assign_colors.cc:38-74: every keyframe, whatever its activation
void ref_assign_colors(const ref_ba_scene* sc) {
  if (sc->surfels_size == 0) return;
  const Bound b(sc);
  CallResetSurfelForColorAssignmentKernel(nullptr, (int)sc->surfels_size, b.surfels);
  for (int k = 0; k < sc->num_keyframes; ++k) {
    const ref_ba_keyframe& kf = sc->keyframes[k];
    RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
    CallAccumulateColorObservationsCUDAKernel(nullptr, (int)sc->surfels_size, projection_of(sc, b, kf), b.d2c, reinterpret_cast<cudaTextureObject_t>(&tex));
  }
  CallAssignColorsCUDAKernel(nullptr, sc->surfels_size, b.surfels);
}

// B/kernel_delete_surfels.cc:38-98 with update_radii = true: every keyframe.  Returns the number of surfels the call marked
// as deleted, counted from the marks (the kernel's own counter is not meaningful under the stand-in BlockReduce, cub/cub.cuh).
uint32_t ref_delete_surfels_and_update_radii(const ref_ba_scene* sc, int min_observation_count) {
  if (sc->surfels_size == 0) return 0;
  const Bound b(sc);
  auto deleted_marks = [&]() {
    uint32_t n = 0;
    for (uint32_t i = 0; i < sc->surfels_size; ++i) n += __float_as_int(b.surfels(kSurfelX, i)) == 0x7fffffff ? 1u : 0u;
    return n;
  };
  const uint32_t before = deleted_marks();
  CallResetSurfelAccumForSurfelDeletionAndRadiusUpdateCUDAKernel(nullptr, sc->surfels_size, b.surfels, true);
  for (int k = 0; k < sc->num_keyframes; ++k) {
    const ref_ba_keyframe& kf = sc->keyframes[k];
    CUDABuffer_<u16> radius_buffer(kf.radius, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
    CallCountObservationsAndFreeSpaceViolationsCUDAKernel(nullptr, projection_of(sc, b, kf), radius_buffer, true);
  }
  u32 counter = 0;
  CUDABuffer_<u32> deleted_count_buffer(&counter, 1, 1, sizeof(u32));
  CallMarkDeletedSurfelsCUDAKernel(nullptr, min_observation_count, sc->surfels_size, b.surfels, &deleted_count_buffer, true);
  return deleted_marks() - before;
}

// B/kernel_supporting_surfels.cc:38-108 for keyframe `keyframe_index`: the three planes (width x height words each, the cell grid
// in their top-left corner) are cleared to kInvalidIndex and filled; with merge_surfels, surfels that project into an occupied
// cell and are close to its occupant in normal and position are marked deleted.  The winner of a cell is whoever gets there
// first: the launch runs sequentially, in ascending surfel order (see ref_launch_sequential).  Returns the number of surfels the
// call marked as deleted.
uint32_t ref_determine_supporting_surfels(const ref_ba_scene* sc, int keyframe_index, int merge_surfels, float merge_dist_factor, uint32_t* planes) {
  const Bound b(sc);
  const size_t plane_words = (size_t)sc->width * sc->height;
  for (size_t w = 0; w < kMergeBufferCount * plane_words; ++w) planes[w] = kInvalidIndex;
  if (sc->surfels_size == 0) return 0;
  SupportingSurfelBuffers buffers;
  for (int i = 0; i < kMergeBufferCount; ++i) buffers.b[i] = CUDABuffer_<u32>(planes + i * plane_words, sc->height, sc->width, (size_t)sc->width * sizeof(u32));
  auto deleted_marks = [&]() {
    uint32_t n = 0;
    for (uint32_t i = 0; i < sc->surfels_size; ++i) n += __float_as_int(b.surfels(kSurfelX, i)) == 0x7fffffff ? 1u : 0u;
    return n;
  };
  const uint32_t before = deleted_marks();
  u32 counter = 0;
  const float cell = (float)sc->cell;
  ref_launch_sequential = true;
  if (merge_surfels)
    CallDetermineSupportingSurfelsCUDAKernel(nullptr, true, cell * cell * merge_dist_factor * merge_dist_factor, cos_normal_compatibility_threshold,
                                             projection_of(sc, b, sc->keyframes[keyframe_index]), buffers, CUDABuffer_<u32>(&counter, 1, 1, sizeof(u32)));
  else
    CallDetermineSupportingSurfelsCUDAKernel(nullptr, false, 0, 0, projection_of(sc, b, sc->keyframes[keyframe_index]), buffers, CUDABuffer_<u32>());
  ref_launch_sequential = false;
  return deleted_marks() - before;
}

// DirectBA::CreateSurfelsForKeyframe (B/direct_ba.cc:340-405 -> B/kernel_create_surfels.cc:40-197): supporting surfels of the
// keyframe, one new surfel per free sparse cell with a valid depth (first pixel of the cell in row-major order: sequential
// launches, see CUDA_AUTO_TUNE_2D), optionally filtered by the observations in the co-visible keyframes, appended behind
// surfels_size in row-major pixel order.  covis_T_frame: n_covis 3x4 matrices = covis.frame_T_global * keyframe.global_T_frame
// (computed by the caller, as B/direct_ba.cc:359-365 does outside the .cu file).  Returns the number of surfels created; the
// caller's surfels_size is NOT advanced here.
uint32_t ref_create_surfels_for_keyframe(const ref_ba_scene* sc, int keyframe_index, int filter_new_surfels, int min_observation_count, int n_covis,
                                         const int* covis_indices, const float* covis_T_frame) {
  const Bound b(sc);
  const ref_ba_keyframe& kf = sc->keyframes[keyframe_index];
  const size_t pixels = (size_t)sc->width * sc->height;
  std::vector<uint32_t> planes(kMergeBufferCount * pixels);
  {  // DetermineSupportingSurfelsCUDA (B/direct_ba.cc:349-358); a surfel already marked deleted cannot be one (its x is NaN)
    ref_ba_scene copy = *sc;
    ref_determine_supporting_surfels(&copy, keyframe_index, 0, 0.f, planes.data());
  }
  CUDABuffer_<u16> depth_buffer(kf.depth, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
  CUDABuffer_<u16> normals_buffer(kf.normals, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
  CUDABuffer_<u16> radius_buffer(kf.radius, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
  CUDABuffer_<uchar4> color_buffer(reinterpret_cast<uchar4*>(kf.rgba), sc->color_height, sc->color_width, (size_t)sc->color_width * sizeof(uchar4));
  CUDABuffer_<u32> supporting0(planes.data(), sc->height, sc->width, (size_t)sc->width * sizeof(u32));
  std::vector<u8> flags(pixels);
  std::vector<u32> indices(pixels);
  CUDABuffer_<u8> flag_vector(flags.data(), 1, (int)pixels, pixels);
  CUDABuffer_<u32> index_vector(indices.data(), 1, (int)pixels, pixels * sizeof(u32));
  void* scan_storage = nullptr;
  usize scan_bytes = 0;

  ref_launch_sequential = true;
  CallCreateSurfelsForKeyframeCUDASerializingKernel(nullptr, sc->cell, depth_buffer, color_buffer, supporting0, flag_vector);
  ref_launch_sequential = false;
  u32 new_surfel_count = CreateSurfelsForKeyframeCUDA_CountNewSurfels(nullptr, (u32)pixels, &scan_storage, &scan_bytes, &flag_vector, &index_vector);
  if (new_surfel_count != 0 && filter_new_surfels) {
    // scratch in the accumulator rows, as the reference lays it out (B/kernel_create_surfels.cc:106-108)
    u8* base = reinterpret_cast<u8*>(b.surfels.address());
    u16* observation_vector = reinterpret_cast<u16*>(base + kSurfelAccum0 * b.surfels.pitch());
    u16* free_space_violation_vector = reinterpret_cast<u16*>(base + kSurfelAccum1 * b.surfels.pitch());
    u32* new_surfel_index_list = reinterpret_cast<u32*>(base + kSurfelAccum2 * b.surfels.pitch());
    CallWriteNewSurfelIndexAndInitializeObservationsCUDAKernel(nullptr, (u32)pixels, flag_vector, index_vector, observation_vector, free_space_violation_vector,
                                                               new_surfel_index_list);
    for (int c = 0; c < n_covis; ++c) {
      const ref_ba_keyframe& other = sc->keyframes[covis_indices[c]];
      CUDAMatrix3x4 M;
      M.row0 = make_float4(covis_T_frame[12 * c + 0], covis_T_frame[12 * c + 1], covis_T_frame[12 * c + 2], covis_T_frame[12 * c + 3]);
      M.row1 = make_float4(covis_T_frame[12 * c + 4], covis_T_frame[12 * c + 5], covis_T_frame[12 * c + 6], covis_T_frame[12 * c + 7]);
      M.row2 = make_float4(covis_T_frame[12 * c + 8], covis_T_frame[12 * c + 9], covis_T_frame[12 * c + 10], covis_T_frame[12 * c + 11]);
      // The kernel projects every candidate into the co-visible keyframe through ProjectSurfelToImage: a candidate whose pixel there lies
      // beyond the int range is outside the image on the GPU (saturating conversion) and reads depth_buffer(py, INT_MIN) on the host
      // (see GuardedProjection).  Such candidates -- one in ~1e8 pairs -- are left out of the launch: the kernel runs on the stretches
      // of the candidate list between them, which is what the GPU's early return amounts to.
      const CUDABuffer_<u16> covis_depth(other.depth, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
      const CUDABuffer_<u16> covis_normals(other.normals, sc->height, sc->width, (size_t)sc->width * sizeof(u16));
      auto beyond_int_range = [&](u32 candidate) {
        const u32 pixel = new_surfel_index_list[candidate], y = pixel / (u32)sc->width, x = pixel - y * (u32)sc->width;
        const float depth = RawToCalibratedDepth(b.dp.a, b.dp.cfactor_buffer(y / b.dp.sparse_surfel_cell_size, x / b.dp.sparse_surfel_cell_size),
                                                 b.dp.raw_to_float_depth, depth_buffer(y, x));
        float3 local_position;
        if (!M.MultiplyIfResultZIsPositive(b.unprojector.UnprojectPoint(x, y, depth), &local_position)) return false;
        const float2 p = b.depth_projector.Project(local_position);
        return !(p.x < 2147483648.f && p.y < 2147483648.f);
      };
      u32 first = 0;
      for (u32 candidate = 0; candidate <= new_surfel_count; ++candidate) {
        if (candidate < new_surfel_count && !beyond_int_range(candidate)) continue;
        if (candidate > first)
          CallCountObservationsForNewSurfelsCUDAKernel(nullptr, (int)(candidate - first), new_surfel_index_list + first, observation_vector + first,
                                                       free_space_violation_vector + first, b.dp, b.unprojector, depth_buffer, normals_buffer, M,
                                                       b.depth_projector, covis_depth, covis_normals);
        first = candidate + 1;
      }
    }
    CallFilterNewSurfelsCUDAKernel(nullptr, (u16)min_observation_count, new_surfel_count, new_surfel_index_list, observation_vector,
                                   free_space_violation_vector, flag_vector);
    new_surfel_count = CreateSurfelsForKeyframeCUDA_CountNewSurfels(nullptr, (u32)pixels, &scan_storage, &scan_bytes, &flag_vector, &index_vector);
  }
  std::free(scan_storage);
  if (new_surfel_count == 0 || sc->surfels_size + new_surfel_count > sc->capacity) return 0;
  CUDAMatrix3x4 G;
  G.row0 = make_float4(kf.global_T_frame[0], kf.global_T_frame[1], kf.global_T_frame[2], kf.global_T_frame[3]);
  G.row1 = make_float4(kf.global_T_frame[4], kf.global_T_frame[5], kf.global_T_frame[6], kf.global_T_frame[7]);
  G.row2 = make_float4(kf.global_T_frame[8], kf.global_T_frame[9], kf.global_T_frame[10], kf.global_T_frame[11]);
  RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
  CallCreateSurfelsForKeyframeCUDACreationAppendKernel(nullptr, b.unprojector, b.d2c, b.color_projector, G, pose_of(kf), b.dp, depth_buffer, normals_buffer,
                                                       radius_buffer, reinterpret_cast<cudaTextureObject_t>(&tex), flag_vector, index_vector, sc->surfels_size,
                                                       b.surfels);
  return new_surfel_count;
}

// AccumulatePoseEstimationCoeffsCUDA (B/kernel_opt_pose.cc:38-96): H (21 entries, row-major upper triangle) and b (6) of the pose
// normal equations of keyframe `keyframe_index`'s images at the pose estimate `frame_T_global` (3x4), over all surfels, by the
// reference's kernel: per-thread Jacobians, 27 block reductions per residual (B/gauss_newton.cuh:46-93; here the block total is a
// binary64 sum of the binary32 terms, cub/cub.cuh), one binary32 atomicAdd per block and entry.  Returns 0 (surfels that project
// beyond the int range or are marked deleted are handled by the launch guard of projection_of).
int ref_accumulate_pose_estimation_coeffs(const ref_ba_scene* sc, int keyframe_index, const float* frame_T_global, int use_depth_residuals,
                                          int use_descriptor_residuals, float* H, float* b) {
  const Bound bound(sc);
  ref_ba_keyframe kf = sc->keyframes[keyframe_index];
  memcpy(kf.frame_T_global, frame_T_global, sizeof(kf.frame_T_global));
  u32 residual_count = 0;
  float residual_sum = 0;
  for (int c = 0; c < 21; ++c) H[c] = 0;
  for (int c = 0; c < 6; ++c) b[c] = 0;
  const PixelCenterProjector color_center_projector(sc->color_cam[0], sc->color_cam[1], sc->color_cam[2] - 0.5f, sc->color_cam[3] - 0.5f);   // B/surfel_projection.h:50-56
  RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
  ref_thread0_last = true;   // thread 0 of a block adds the block totals (B/gauss_newton.cuh:71,89)
  CallAccumulatePoseEstimationCoeffsCUDAKernel(nullptr, /*debug*/ false, use_depth_residuals != 0, use_descriptor_residuals != 0, projection_of(sc, bound, kf),
                                               bound.d2c, color_center_projector, bound.color_projector, bound.unprojector,
                                               reinterpret_cast<cudaTextureObject_t>(&tex), CUDABuffer_<u32>(&residual_count, 1, 1, sizeof(u32)),
                                               CUDABuffer_<float>(&residual_sum, 1, 1, sizeof(float)), CUDABuffer_<float>(H, 1, 21, 21 * sizeof(float)),
                                               CUDABuffer_<float>(b, 1, 6, 6 * sizeof(float)));
  ref_thread0_last = false;
  return 0;
}

// The system the PCG scheme solves, as the reference assembles it (B/direct_ba_pcg.cc:276-365: unknown layout, then PCGInitCUDA
// once per keyframe): r = -J^T W F and M = diag(J^T W J) over the unknowns
//   6 per keyframe except the gauge keyframe | 1 or 3 per surfel (offset along the normal, descriptors) | fx^-1 fy^-1 cx^-1 cy^-1 a
//   + one cfactor per sparse cell | 4 colour intrinsics
// in that order, each block only if it is optimised.  Dense entries are block sums added with binary32 atomics (arrival order),
// surfel entries per-thread sums keyframe by keyframe.  Returns the number of unknowns (0 if r / M are too short).
uint32_t ref_pcg_assemble(const ref_ba_scene* sc, int optimize_poses, int optimize_geometry, int use_depth_residuals, int use_descriptor_residuals,
                          int optimize_depth_intrinsics, int optimize_color_intrinsics, int gauge_keyframe, float* r, float* M, uint32_t capacity) {
  const Bound bound(sc);
  constexpr u32 kInvalidUnknownIndex = 0xffffffffu;
  const int K = sc->num_keyframes;
  u32 current = 0;
  if (optimize_poses) current += 6 * (K - 1);
  u32 surfel_start = kInvalidUnknownIndex;
  if (optimize_geometry) { surfel_start = current; current += (use_descriptor_residuals ? 3 : 1) * sc->surfels_size; }
  u32 depth_intrinsics_start = kInvalidUnknownIndex;
  if (optimize_depth_intrinsics) { depth_intrinsics_start = current; current += 4 + 1 + (u32)sc->cf_width * sc->cf_height; }
  u32 color_intrinsics_start = kInvalidUnknownIndex;
  if (optimize_color_intrinsics) { color_intrinsics_start = current; current += 4; }
  const u32 unknown_count = current;
  if (unknown_count > capacity) return 0;
  memset(r, 0, sizeof(float) * unknown_count);
  memset(M, 0, sizeof(float) * unknown_count);
  CUDABuffer_<PCGScalar> pcg_r(r, 1, (int)unknown_count, sizeof(float) * unknown_count), pcg_M(M, 1, (int)unknown_count, sizeof(float) * unknown_count);
  ref_thread0_last = true;   // thread 0 of a block adds the block totals (B/kernel_pcg.cu:78-93)
  for (int k = 0; k < K; ++k) {
    const ref_ba_keyframe& kf = sc->keyframes[k];
    const u32 pose_index = k == gauge_keyframe ? kInvalidUnknownIndex : (u32)(6 * (k < gauge_keyframe ? k : k - 1));   // :329-337
    RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
    PCGInitCUDA(nullptr, projection_of(sc, bound, kf), bound.d2c, bound.unprojector, bound.color_projector, reinterpret_cast<cudaTextureObject_t>(&tex), pose_index,
                surfel_start, k == gauge_keyframe ? false : optimize_poses != 0, optimize_geometry != 0, use_depth_residuals != 0, use_descriptor_residuals != 0,
                optimize_depth_intrinsics != 0, optimize_color_intrinsics != 0, depth_intrinsics_start, color_intrinsics_start, &pcg_r, &pcg_M, sc->surfels_size);
  }
  ref_thread0_last = false;
  return unknown_count;
}

// The accumulation of OptimizeIntrinsicsCUDA (B/kernel_opt_intrinsics.cc:39-104): buffers cleared, then the accumulation kernel
// once per keyframe.  glob[34]: A (15, row-major upper triangle of the 5 x 5 block of fx^-1 fy^-1 cx^-1 cy^-1 a), b1 (5), colour
// H (10), colour b (4); cells[8 S], per sparse cell: B0..B4, D, b2, observation count -- the layout of orc_intrinsics_accumulate.
// Returns 0.
int ref_intrinsics_accumulate(const ref_ba_scene* sc, int optimize_depth_intrinsics, int optimize_color_intrinsics, float* glob, float* cells) {
  const Bound bound(sc);
  const int S = sc->cf_width * sc->cf_height;
  std::vector<u32> observation_count(S, 0);
  std::vector<float> depth_A(15, 0.f), depth_B(5 * (size_t)S, 0.f), depth_D(S, 0.f), depth_b1(5, 0.f), depth_b2(S, 0.f), color_H(10, 0.f), color_b(4, 0.f);
  ref_thread0_last = true;   // thread 0 of a block adds the block totals (B/gauss_newton.cuh:71,89)
  for (int k = 0; k < sc->num_keyframes; ++k) {
    const ref_ba_keyframe& kf = sc->keyframes[k];
    RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
    CallAccumulateIntrinsicsCoefficientsCUDAKernel(
        nullptr, optimize_color_intrinsics != 0, optimize_depth_intrinsics != 0, projection_of(sc, bound, kf), bound.d2c, bound.color_projector, bound.unprojector,
        sc->color_cam[0], sc->color_cam[1], reinterpret_cast<cudaTextureObject_t>(&tex), CUDABuffer_<u32>(observation_count.data(), 1, S, sizeof(u32) * S),
        CUDABuffer_<float>(depth_A.data(), 1, 15, sizeof(float) * 15), CUDABuffer_<float>(depth_B.data(), 5, S, sizeof(float) * S),
        CUDABuffer_<float>(depth_D.data(), 1, S, sizeof(float) * S), CUDABuffer_<float>(depth_b1.data(), 1, 5, sizeof(float) * 5),
        CUDABuffer_<float>(depth_b2.data(), 1, S, sizeof(float) * S), CUDABuffer_<float>(color_H.data(), 1, 10, sizeof(float) * 10),
        CUDABuffer_<float>(color_b.data(), 1, 4, sizeof(float) * 4));
  }
  ref_thread0_last = false;
  for (int q = 0; q < 15; ++q) glob[q] = depth_A[q];
  for (int q = 0; q < 5; ++q) glob[15 + q] = depth_b1[q];
  for (int q = 0; q < 10; ++q) glob[20 + q] = color_H[q];
  for (int q = 0; q < 4; ++q) glob[30 + q] = color_b[q];
  for (int c = 0; c < S; ++c) {
    for (int q = 0; q < 5; ++q) cells[8 * c + q] = depth_B[(size_t)q * S + c];
    cells[8 * c + 5] = depth_D[c];
    cells[8 * c + 6] = depth_b2[c];
    cells[8 * c + 7] = (float)observation_count[c];
  }
  return 0;
}

// One outer iteration of BundleAdjustmentPCG over poses and geometry (B/direct_ba_pcg.cc:172-560, without surfel creation /
// merging and without the intrinsics blocks): every surfel active, UpdateSurfelNormalsCUDA (B/kernel_opt_geometry.cc:39-77), the
// system assembled by PCGInit per keyframe, PCGInit2, then up to max_inner_iterations steps of PCGStep1 per keyframe / PCGStep2 /
// the stopping rule of :441-456 / PCGStep3, and UpdateSurfelsFromPCGDeltaCUDA.  The surfels are updated in place; the pose block of
// delta (6 (K - 1) values, gauge keyframe left out) is returned for the caller to apply T <- T * exp(delta) (:566-583, host code).
// Returns the number of inner steps taken.
int ref_pcg_outer_iteration(const ref_ba_scene* sc, int use_depth_residuals, int use_descriptor_residuals, int gauge_keyframe, int max_inner_iterations,
                            float* pose_delta) {
  const Bound bound(sc);
  const int K = sc->num_keyframes;
  memset(sc->active, kSurfelActiveFlag, sc->surfels_size);                                                            // :212-216
  CallResetSurfelAccum0to3CUDAKernel(nullptr, sc->surfels_size, bound.surfels, bound.active);                        // UpdateSurfelNormalsCUDA
  for (int k = 0; k < K; ++k) {
    const ref_ba_keyframe& kf = sc->keyframes[k];
    if (kf.activation == kInactive) continue;
    CUDAMatrix3x3 R;
    R.row0 = make_float3(kf.global_R_frame[0], kf.global_R_frame[1], kf.global_R_frame[2]);
    R.row1 = make_float3(kf.global_R_frame[3], kf.global_R_frame[4], kf.global_R_frame[5]);
    R.row2 = make_float3(kf.global_R_frame[6], kf.global_R_frame[7], kf.global_R_frame[8]);
    CallAccumulateSurfelNormalOptimizationCoeffsCUDAKernel(nullptr, projection_of(sc, bound, kf), R, bound.active);
  }
  CallUpdateSurfelNormalCUDAKernel(nullptr, sc->surfels_size, bound.surfels, bound.active);

  constexpr u32 kInvalidUnknownIndex = 0xffffffffu;
  const u32 surfel_start = 6 * (K - 1);
  const u32 unknown_count = surfel_start + (use_descriptor_residuals ? 3 : 1) * sc->surfels_size;
  std::vector<float> r(unknown_count, 0.f), M(unknown_count, 0.f), delta(unknown_count, 0.f), g(unknown_count, 0.f), p(unknown_count, 0.f);
  float alpha_n = 0, alpha_d = 0, beta_n = 0;
  const size_t row = sizeof(float) * unknown_count;
  CUDABuffer_<PCGScalar> pcg_r(r.data(), 1, (int)unknown_count, row), pcg_M(M.data(), 1, (int)unknown_count, row), pcg_delta(delta.data(), 1, (int)unknown_count, row),
      pcg_g(g.data(), 1, (int)unknown_count, row), pcg_p(p.data(), 1, (int)unknown_count, row);
  CUDABuffer_<PCGScalar> pcg_alpha_n(&alpha_n, 1, 1, sizeof(float)), pcg_alpha_d(&alpha_d, 1, 1, sizeof(float)), pcg_beta_n(&beta_n, 1, 1, sizeof(float));
  auto pose_index = [&](int k) { return k == gauge_keyframe ? kInvalidUnknownIndex : (u32)(6 * (k < gauge_keyframe ? k : k - 1)); };   // :329-337
  ref_thread0_last = true;   // thread 0 of a block adds the block totals (B/kernel_pcg.cu:78-93)
  for (int k = 0; k < K; ++k) {
    const ref_ba_keyframe& kf = sc->keyframes[k];
    RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
    PCGInitCUDA(nullptr, projection_of(sc, bound, kf), bound.d2c, bound.unprojector, bound.color_projector, reinterpret_cast<cudaTextureObject_t>(&tex), pose_index(k),
                surfel_start, k != gauge_keyframe, true, use_depth_residuals != 0, use_descriptor_residuals != 0, false, false, kInvalidUnknownIndex,
                kInvalidUnknownIndex, &pcg_r, &pcg_M, sc->surfels_size);
  }
  PCGInit2CUDA(nullptr, unknown_count, kInvalidUnknownIndex, sc->a, pcg_r, pcg_M, &pcg_delta, &pcg_g, &pcg_p, &pcg_alpha_n);
  float prev_r_norm = std::numeric_limits<float>::infinity();
  int without_improvement = 0, steps = 0;
  CUDABuffer_<PCGScalar>* an = &pcg_alpha_n;
  CUDABuffer_<PCGScalar>* bn = &pcg_beta_n;
  for (int step = 0; step < max_inner_iterations; ++step) {
    alpha_d = 0;
    if (step > 0) {
      std::swap(an, bn);                                                                                               // :388-392
      std::fill(g.begin(), g.end(), 0.f);
    }
    for (int k = 0; k < K; ++k) {
      const ref_ba_keyframe& kf = sc->keyframes[k];
      RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
      PCGStep1CUDA(nullptr, unknown_count, projection_of(sc, bound, kf), bound.d2c, bound.unprojector, bound.color_projector, reinterpret_cast<cudaTextureObject_t>(&tex),
                   pose_index(k), surfel_start, k != gauge_keyframe, true, use_depth_residuals != 0, use_descriptor_residuals != 0, false, false, kInvalidUnknownIndex,
                   kInvalidUnknownIndex, kInvalidUnknownIndex, &pcg_p, &pcg_g, &pcg_alpha_d, sc->surfels_size);
    }
    PCGStep2CUDA(nullptr, unknown_count, kInvalidUnknownIndex, pcg_r, pcg_M, &pcg_delta, &pcg_g, &pcg_p, an, &pcg_alpha_d, bn);
    ++steps;
    const float r_norm = std::sqrt(*bn->address());                                                                    // :437-456
    if (r_norm < prev_r_norm - 1e-3) without_improvement = 0;
    else if (++without_improvement >= 3) break;
    prev_r_norm = r_norm;
    if (step < max_inner_iterations - 1) PCGStep3CUDA(nullptr, unknown_count, &pcg_g, &pcg_p, an, bn);
  }
  ref_thread0_last = false;
  CUDABuffer_<float> surfels = bound.surfels;
  UpdateSurfelsFromPCGDeltaCUDA(nullptr, sc->surfels_size, &surfels, use_descriptor_residuals != 0, surfel_start, pcg_delta);   // :585-594
  for (u32 u = 0; u < surfel_start; ++u) pose_delta[u] = delta[u];
  return steps;
}

}  // extern "C"

namespace {
// x = A^-1 b for a small symmetric system given by its upper triangle, in binary64 (the reference: Eigen's
// A.cast<double>().selfadjointView<Upper>().ldlt().solve(b.cast<double>()), B/kernel_opt_intrinsics.cc:173,269).  Gaussian
// elimination with partial pivoting: for these well-conditioned-in-binary64 systems the two agree far below binary32 resolution.
template <int N>
void solve_symmetric(const float* upper, const float* rhs, float* x) {
  double A[N][N + 1];
  int index = 0;
  for (int row = 0; row < N; ++row)
    for (int col = row; col < N; ++col, ++index) A[row][col] = A[col][row] = (double)upper[index];
  for (int row = 0; row < N; ++row) A[row][N] = (double)rhs[row];
  for (int k = 0; k < N; ++k) {
    int pivot = k;
    for (int r = k + 1; r < N; ++r) if (std::fabs(A[r][k]) > std::fabs(A[pivot][k])) pivot = r;
    for (int c = 0; c <= N; ++c) std::swap(A[k][c], A[pivot][c]);
    for (int r = k + 1; r < N; ++r) {
      const double f = A[r][k] / A[k][k];
      for (int c = k; c <= N; ++c) A[r][c] -= f * A[k][c];
    }
  }
  double y[N];
  for (int k = N - 1; k >= 0; --k) {
    double v = A[k][N];
    for (int c = k + 1; c < N; ++c) v -= A[k][c] * y[c];
    y[k] = v / A[k][k];
  }
  for (int k = 0; k < N; ++k) x[k] = (float)y[k];
}
}  // namespace

extern "C" {

// OptimizeIntrinsicsCUDA (B/kernel_opt_intrinsics.cc:39-281), the intrinsics step of the alternating scheme, by the reference's
// kernels: accumulation once per keyframe, the Schur complement kernel, the 5 x 5 solve with the prior on `a` (host code of the
// reference, :120-195, restated: binary64 solve), the per-cell back-substitution kernel -- which updates sc->cfactor in place --
// and the 4 x 4 solve of the colour camera (:255-279).  Outputs: the new depth and colour cameras (fx, fy, cx, cy) and `a`.
// Returns 0.
int ref_optimize_intrinsics(const ref_ba_scene* sc, int optimize_depth_intrinsics, int optimize_color_intrinsics, float* out_depth_cam, float* out_color_cam,
                            float* out_a) {
  const Bound bound(sc);
  for (int c = 0; c < 4; ++c) { out_depth_cam[c] = sc->depth_cam[c]; out_color_cam[c] = sc->color_cam[c]; }
  *out_a = sc->a;
  const int S = sc->cf_width * sc->cf_height;
  std::vector<u32> observation_count(S, 0);
  std::vector<float> depth_A(15, 0.f), depth_B(5 * (size_t)S, 0.f), depth_D(S, 0.f), depth_b1(5, 0.f), depth_b2(S, 0.f), color_H(10, 0.f), color_b(4, 0.f);
  CUDABuffer_<u32> count_buffer(observation_count.data(), 1, S, sizeof(u32) * S);
  CUDABuffer_<float> A(depth_A.data(), 1, 15, sizeof(float) * 15), B(depth_B.data(), 5, S, sizeof(float) * S), D(depth_D.data(), 1, S, sizeof(float) * S),
      b1(depth_b1.data(), 1, 5, sizeof(float) * 5), b2(depth_b2.data(), 1, S, sizeof(float) * S), H(color_H.data(), 1, 10, sizeof(float) * 10),
      hb(color_b.data(), 1, 4, sizeof(float) * 4);
  ref_thread0_last = true;
  for (int k = 0; k < sc->num_keyframes; ++k) {
    const ref_ba_keyframe& kf = sc->keyframes[k];
    RefTexture tex = {reinterpret_cast<const uchar4*>(kf.rgba), sc->color_width, sc->color_height, (size_t)sc->color_width * 4, sc->quantize_texture_weights};
    CallAccumulateIntrinsicsCoefficientsCUDAKernel(nullptr, optimize_color_intrinsics != 0, optimize_depth_intrinsics != 0, projection_of(sc, bound, kf), bound.d2c,
                                                   bound.color_projector, bound.unprojector, sc->color_cam[0], sc->color_cam[1],
                                                   reinterpret_cast<cudaTextureObject_t>(&tex), count_buffer, A, B, D, b1, b2, H, hb);
  }
  if (optimize_depth_intrinsics) {
    CallComputeIntrinsicsIntermediateMatricesCUDAKernel(nullptr, (u32)S, A, B, D, b1, b2);                                // :118-127
    ref_thread0_last = false;
    constexpr float kAPriorWeight = 10;                                                                                    // :150-152
    depth_A[14] += kAPriorWeight * kAPriorWeight;
    depth_b1[4] += kAPriorWeight * kAPriorWeight * sc->a;
    float x1[5];
    solve_symmetric<5>(depth_A.data(), depth_b1.data(), x1);                                                              // :173
    const float new_fx = 1.0f / (bound.unprojector.fx_inv - x1[0]), new_fy = 1.0f / (bound.unprojector.fy_inv - x1[1]);    // :184-187
    out_depth_cam[0] = new_fx;
    out_depth_cam[1] = new_fy;
    out_depth_cam[2] = -(new_fx * (bound.unprojector.cx_inv - x1[2])) + 0.5f;
    out_depth_cam[3] = -(new_fy * (bound.unprojector.cy_inv - x1[3])) + 0.5f;
    *out_a = sc->a - x1[4];
    for (int c = 0; c < 5; ++c) depth_b1[c] = x1[c];                                                                       // b1 re-used for x1
    ref_barrier_passes = true;   // x1 travels through __shared__ memory across a __syncthreads() (B/kernel_opt_intrinsics.cu:384-398)
    CallSolveForPixelIntrinsicsUpdateCUDAKernel(nullptr, (u32)S, count_buffer, B, D, b1,
                                                CUDABuffer_<float>(sc->cfactor, sc->cf_height, sc->cf_width, sizeof(float) * sc->cf_width));
    ref_barrier_passes = false;
  }
  ref_thread0_last = false;
  if (optimize_color_intrinsics) {
    float x[4];
    solve_symmetric<4>(color_H.data(), color_b.data(), x);                                                                // :255-279
    for (int c = 0; c < 4; ++c) out_color_cam[c] = sc->color_cam[c] - x[c];
  }
  return 0;
}

}  // extern "C"
