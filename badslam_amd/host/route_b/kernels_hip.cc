// kernels_hip.cc -- Route B of INTEGRATION.md: the free functions of the reference's B/kernels.h (the *CUDA entry points
// DirectBA calls) implemented on top of the C ABI of the MI355X backend (include/badslam_hip.h).  A maintainer of the
// reference who keeps their own DirectBA host code drops this file in place of kernel_*.cc / kernel_*.cu.  Every function
// does the same three things: describe the caller's buffers as bahip_* structs (pointers are borrowed), bind cameras / depth
// parameters / keyframes, call the one bahip_* function that replaces the reference function, and turn a failure into
// LOG(FATAL) like the reference's CUDA_CHECK().  One backend context per host thread, re-pointed at the caller's stream.
#include <algorithm>
#include "badslam/kernels.h"

namespace vis {
namespace {

bahip_context* Ctx(cudaStream_t stream) {
  static thread_local bahip_context* ctx = nullptr;
  if (!ctx) BAHIP_CHECKED_CALL(bahip_context_create(&ctx, stream));
  BAHIP_CHECKED_CALL(bahip_context_set_stream(ctx, stream));
  return ctx;
}

bahip_surfels Surfels(const CUDABuffer<float>& surfels, const CUDABuffer<u8>* active, u32 surfels_size) {
  bahip_surfels s;
  s.data = surfels.ToCUDA().address();
  s.pitch_bytes = (uint32_t)surfels.ToCUDA().pitch();
  s.active = active ? active->ToCUDA().address() : nullptr;
  s.surfels_size = surfels_size;
  s.capacity = (uint32_t)surfels.width();
  return s;
}

void BindIntrinsics(bahip_context* ctx, const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                    const DepthParameters& depth_params) {
  const bahip_camera cc = ToBahipCamera(color_camera), dc = ToBahipCamera(depth_camera);
  const bahip_depth_params dp = ToBahipDepthParams(depth_params);
  BAHIP_CHECKED_CALL(bahip_set_intrinsics(ctx, &cc, &dc, &dp));
}

// The keyframe vector of the reference (deleted keyframes are null entries) -> the dense bound list; returns the bound
// index of `keyframe_id` (or -1).
int BindKeyframes(bahip_context* ctx, const vector<shared_ptr<Keyframe>>& keyframes, int keyframe_id = -1) {
  vector<bahip_keyframe> table;
  int bound_of_id = -1;
  for (const shared_ptr<Keyframe>& kf : keyframes) {
    if (!kf) continue;
    bahip_keyframe e;
    e.frame = kf->ToBahipFrame();
    memcpy(e.global_T_frame, kf->global_T_frame().data(), 7 * sizeof(float));
    e.activation = (int)kf->activation();
    if (kf->id() == keyframe_id) bound_of_id = (int)table.size();
    table.push_back(e);
  }
  BAHIP_CHECKED_CALL(bahip_set_keyframes(ctx, table.data(), (int)table.size()));
  return bound_of_id;
}

bahip_frame Frame(const CUDABuffer<u16>& depth, const CUDABuffer<u16>& normals, const CUDABuffer<u16>* radius,
                  const CUDABuffer<uchar4>* color) {
  bahip_frame f{};
  f.depth = depth.ToCUDA().address(); f.depth_pitch_bytes = (uint32_t)depth.ToCUDA().pitch();
  f.normals = normals.ToCUDA().address(); f.normals_pitch_bytes = (uint32_t)normals.ToCUDA().pitch();
  if (radius) { f.radius = radius->ToCUDA().address(); f.radius_pitch_bytes = (uint32_t)radius->ToCUDA().pitch(); }
  if (color) { f.color = reinterpret_cast<uint8_t*>(color->ToCUDA().address()); f.color_pitch_bytes = (uint32_t)color->ToCUDA().pitch(); }
  f.planes = nullptr;   // the library packs its tiled planes itself (a caller that keeps bahip_frame_planes passes them)
  return f;
}

void SupportingPointers(CUDABuffer<u32>** supporting_surfels, uint32_t* out[BAHIP_MERGE_BUFFER_COUNT], uint32_t* pitch) {
  for (int i = 0; i < BAHIP_MERGE_BUFFER_COUNT; ++i) out[i] = supporting_surfels[i]->ToCUDA().address();
  *pitch = (uint32_t)supporting_surfels[0]->ToCUDA().pitch();
}

}  // namespace

void DetermineSupportingSurfelsCUDA(cudaStream_t stream, const PinholeCamera4f& camera, const CUDAMatrix3x4& frame_T_global,
                                    const DepthParameters& depth_params, const CUDABuffer<u16>& depth_buffer,
                                    const CUDABuffer<u16>& normals_buffer, u32 surfels_size, CUDABuffer<float>* surfels,
                                    CUDABuffer<u32>** supporting_surfels) {
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, camera, camera, depth_params);
  const bahip_frame frame = Frame(depth_buffer, normals_buffer, nullptr, nullptr);
  const bahip_surfels s = Surfels(*surfels, nullptr, surfels_size);
  uint32_t* sup[BAHIP_MERGE_BUFFER_COUNT]; uint32_t pitch;
  SupportingPointers(supporting_surfels, sup, &pitch);
  BAHIP_CHECKED_CALL(bahip_determine_supporting_surfels(ctx, 0, 0.f, &frame, &frame_T_global.row0.x, &s, sup, pitch, nullptr));
}

void DetermineSupportingSurfelsAndMergeSurfelsCUDA(cudaStream_t stream, float merge_dist_factor, const PinholeCamera4f& camera,
                                                   const CUDAMatrix3x4& frame_T_global, const DepthParameters& depth_params,
                                                   const CUDABuffer<u16>& depth_buffer, const CUDABuffer<u16>& normals_buffer,
                                                   u32 surfels_size, CUDABuffer<float>* surfels, CUDABuffer<u32>** supporting_surfels,
                                                   u32* surfel_count, CUDABufferPtr<u32>* /*deleted_count_buffer*/) {
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, camera, camera, depth_params);
  const bahip_frame frame = Frame(depth_buffer, normals_buffer, nullptr, nullptr);
  const bahip_surfels s = Surfels(*surfels, nullptr, surfels_size);
  uint32_t* sup[BAHIP_MERGE_BUFFER_COUNT]; uint32_t pitch;
  SupportingPointers(supporting_surfels, sup, &pitch);
  uint32_t merged = 0;
  BAHIP_CHECKED_CALL(bahip_determine_supporting_surfels(ctx, 1, merge_dist_factor, &frame, &frame_T_global.row0.x, &s, sup, pitch, &merged));
  *surfel_count -= merged;
}

void CreateSurfelsForKeyframeCUDA(cudaStream_t stream, int /*sparse_surfel_cell_size: part of depth_params*/, bool filter_new_surfels,
                                  int min_observation_count, int keyframe_id, const vector<shared_ptr<Keyframe>>& keyframes,
                                  const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                                  const CUDAMatrix3x4& /*global_T_frame*/, const CUDAMatrix3x4& /*frame_T_global*/,
                                  const vector<CUDAMatrix3x4>& /*covis_T_frame: recomputed from the bound poses*/,
                                  const DepthParameters& depth_params, const CUDABuffer<u16>&, const CUDABuffer<u16>&,
                                  const CUDABuffer<u16>&, const CUDABuffer<uchar4>&, cudaTextureObject_t,
                                  CUDABuffer<u32>** supporting_surfels, void**, usize*, CUDABuffer<u8>*, CUDABuffer<u32>*,
                                  u32 surfels_size, u32 /*surfel_count*/, u32* new_surfel_count, CUDABuffer<float>* surfels) {
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, color_camera, depth_camera, depth_params);
  const int bound = BindKeyframes(ctx, keyframes, keyframe_id);
  CHECK(bound >= 0) << "keyframe " << keyframe_id << " is not in the keyframe list";
  // co-visible keyframes: ids -> bound indices (the reference passes their relative transforms; the backend derives them)
  vector<int> id_to_bound(keyframes.size(), -1), covis;
  { int b = 0; for (const auto& kf : keyframes) if (kf) id_to_bound[kf->id()] = b++; }
  for (int id : keyframes[keyframe_id]->co_visibility_list())
    if (id >= 0 && id < (int)id_to_bound.size() && id_to_bound[id] >= 0) covis.push_back(id_to_bound[id]);
  const bahip_surfels s = Surfels(*surfels, nullptr, surfels_size);
  uint32_t* sup[BAHIP_MERGE_BUFFER_COUNT]; uint32_t pitch;
  SupportingPointers(supporting_surfels, sup, &pitch);
  uint32_t created = 0;
  BAHIP_CHECKED_CALL(bahip_create_surfels_for_keyframe(ctx, bound, filter_new_surfels ? 1 : 0, min_observation_count, covis.data(),
                                                       (int)covis.size(), &s, sup, pitch, &created));
  *new_surfel_count = created;
}

void AccumulatePoseEstimationCoeffsCUDA(cudaStream_t stream, bool use_depth_residuals, bool use_descriptor_residuals,
                                        const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                                        const DepthParameters& depth_params, const CUDABuffer<u16>& depth_buffer,
                                        const CUDABuffer<u16>& normals_buffer, cudaTextureObject_t color_texture,
                                        const CUDAMatrix3x4& frame_T_global_estimate, u32 surfels_size,
                                        const CUDABuffer<float>& surfels, bool /*debug*/, u32* /*residual_count*/,
                                        float* /*residual_sum*/, float* H, float* b, PoseEstimationHelperBuffers*) {
  CHECK(use_depth_residuals || use_descriptor_residuals);   // B/kernel_opt_pose.cc:58
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, color_camera, depth_camera, depth_params);
  const bahip_frame frame = Frame(depth_buffer, normals_buffer, nullptr, color_texture);   // the handle IS the colour buffer
  const bahip_surfels s = Surfels(surfels, nullptr, surfels_size);
  BAHIP_CHECKED_CALL(bahip_accumulate_pose_estimation_coeffs(ctx, use_depth_residuals, use_descriptor_residuals, &frame,
                                                             &frame_T_global_estimate.row0.x, &s, H, b));
}

void UpdateSurfelNormalsCUDA(cudaStream_t stream, const PinholeCamera4f& depth_camera, const DepthParameters& depth_params,
                             const vector<shared_ptr<Keyframe>>& keyframes, u32 surfels_size, const CUDABuffer<float>& surfels,
                             const CUDABuffer<u8>& active_surfels) {
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, depth_camera, depth_camera, depth_params);
  BindKeyframes(ctx, keyframes);
  const bahip_surfels s = Surfels(surfels, &active_surfels, surfels_size);
  BAHIP_CHECKED_CALL(bahip_update_surfel_normals(ctx, &s));
}

void OptimizeGeometryIterationCUDA(cudaStream_t stream, bool use_depth_residuals, bool use_descriptor_residuals,
                                   const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                                   const DepthParameters& depth_params, const vector<shared_ptr<Keyframe>>& keyframes,
                                   u32 surfels_size, const CUDABuffer<float>& surfels, const CUDABuffer<u8>& active_surfels) {
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, color_camera, depth_camera, depth_params);
  BindKeyframes(ctx, keyframes);
  const bahip_surfels s = Surfels(surfels, &active_surfels, surfels_size);
  BAHIP_CHECKED_CALL(bahip_optimize_geometry_iteration(ctx, use_depth_residuals, use_descriptor_residuals, &s));
}

void OptimizeIntrinsicsCUDA(cudaStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                            const vector<shared_ptr<Keyframe>>& keyframes, const PinholeCamera4f& color_camera,
                            const PinholeCamera4f& depth_camera, const DepthParameters& depth_params, u32 surfels_size,
                            const CUDABuffer<float>& surfels, PinholeCamera4f* out_color_camera, PinholeCamera4f* out_depth_camera,
                            float* a, CUDABufferPtr<float>* /*cfactor_buffer: updated in place through depth_params*/,
                            IntrinsicsOptimizationHelperBuffers*) {
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, color_camera, depth_camera, depth_params);
  BindKeyframes(ctx, keyframes);
  const bahip_surfels s = Surfels(surfels, nullptr, surfels_size);
  bahip_camera cc, dc;
  BAHIP_CHECKED_CALL(bahip_optimize_intrinsics(ctx, optimize_depth_intrinsics, optimize_color_intrinsics, &s, &cc, &dc, a));
  *out_color_camera = PinholeCamera4f(cc.width, cc.height, &cc.fx);
  *out_depth_camera = PinholeCamera4f(dc.width, dc.height, &dc.fx);
}

void UpdateSurfelActivationCUDA(cudaStream_t stream, const PinholeCamera4f& camera, const DepthParameters& depth_params,
                                const vector<shared_ptr<Keyframe>>& keyframes, u32 surfels_size, CUDABuffer<float>* surfels,
                                CUDABuffer<u8>* active_surfels) {
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, camera, camera, depth_params);
  BindKeyframes(ctx, keyframes);
  const bahip_surfels s = Surfels(*surfels, active_surfels, surfels_size);
  BAHIP_CHECKED_CALL(bahip_update_surfel_activation(ctx, &s, surfels_size));
}

void DeleteSurfelsAndUpdateRadiiCUDA(cudaStream_t stream, int min_observation_count, const PinholeCamera4f& camera,
                                     const DepthParameters& depth_params, const vector<shared_ptr<Keyframe>>& keyframes,
                                     u32* surfel_count, u32 surfels_size, CUDABuffer<float>* surfels, CUDABufferPtr<u32>*) {
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, camera, camera, depth_params);
  BindKeyframes(ctx, keyframes);
  const bahip_surfels s = Surfels(*surfels, nullptr, surfels_size);
  uint32_t deleted = 0;
  BAHIP_CHECKED_CALL(bahip_delete_surfels_and_update_radii(ctx, min_observation_count, &s, &deleted));
  *surfel_count -= deleted;
}

void CompactSurfelsCUDA(cudaStream_t stream, void**, usize*, u32 surfel_count, u32* surfels_size, CUDABuffer_<float>* surfels,
                        CUDABuffer_<u8>* active_surfels) {
  bahip_context* ctx = Ctx(stream);
  bahip_surfels s;
  s.data = surfels->address(); s.pitch_bytes = (uint32_t)surfels->pitch();
  s.active = active_surfels ? active_surfels->address() : nullptr;
  s.surfels_size = *surfels_size; s.capacity = (uint32_t)surfels->width();
  BAHIP_CHECKED_CALL(bahip_compact_surfels(ctx, surfel_count, &s));
  *surfels_size = surfel_count;
}

void AssignColorsCUDA(cudaStream_t stream, const PinholeCamera4f& color_camera, const PinholeCamera4f& depth_camera,
                      const DepthParameters& depth_params, const vector<shared_ptr<Keyframe>>& keyframes, u32 surfels_size,
                      CUDABuffer<float>* surfels) {
  bahip_context* ctx = Ctx(stream);
  BindIntrinsics(ctx, color_camera, depth_camera, depth_params);
  BindKeyframes(ctx, keyframes);
  const bahip_surfels s = Surfels(*surfels, nullptr, surfels_size);
  BAHIP_CHECKED_CALL(bahip_assign_colors(ctx, &s));
}

// ---- the PCG scheme (B/kernels.h:397-491) on the stage entry points of the C ABI ------------------------------------------------
// The reference's driver (B/direct_ba_pcg.cc:229-646) calls PCGInitCUDA / PCGStep1CUDA once per keyframe with the keyframe's
// buffers packed in SurfelProjectionParameters; the backend keeps the dense sums of those calls in exact accumulators inside
// its context and writes them when PCGInit2CUDA / PCGStep2CUDA run (include/badslam_hip.h, "PCG solver, stage by stage").  The
// unknown layout is not an argument of every reference function: the shim remembers the one the last PCGInitCUDA /
// PCGStep1CUDA call described (the driver uses one layout per outer iteration).
namespace {
struct PcgSession {
  bahip_pcg_layout layout{};
  u32 surfels_size = 0;
  bool begun = false;      // bahip_pcg_begin ran for this outer iteration (reset by PCGInit2CUDA)
};
PcgSession& Session() { static thread_local PcgSession s; return s; }

void BindFromProjection(bahip_context* ctx, const SurfelProjectionParameters& s, const DepthToColorPixelCorner& depth_to_color,
                        const PixelCornerProjector& color_projector) {
  bahip_camera dc{s.projector.fx, s.projector.fy, s.projector.cx, s.projector.cy, s.depth_buffer.width(), s.depth_buffer.height()};
  bahip_camera cc{color_projector.fx, color_projector.fy, color_projector.cx, color_projector.cy, depth_to_color.width, depth_to_color.height};
  const bahip_depth_params dp = ToBahipDepthParams(s.depth_params);
  BAHIP_CHECKED_CALL(bahip_set_intrinsics(ctx, &cc, &dc, &dp));
}
bahip_frame FrameFromProjection(const SurfelProjectionParameters& s, cudaTextureObject_t color_texture) {
  bahip_frame f{};
  f.depth = s.depth_buffer.address(); f.depth_pitch_bytes = (uint32_t)s.depth_buffer.pitch();
  f.normals = s.normals_buffer.address(); f.normals_pitch_bytes = (uint32_t)s.normals_buffer.pitch();
  if (color_texture) { f.color = reinterpret_cast<uint8_t*>(color_texture->ToCUDA().address()); f.color_pitch_bytes = (uint32_t)color_texture->ToCUDA().pitch(); }
  return f;
}
bahip_surfels SurfelsFromProjection(const SurfelProjectionParameters& s) {
  bahip_surfels out{};
  out.data = s.surfels.address(); out.pitch_bytes = (uint32_t)s.surfels.pitch();
  out.surfels_size = s.surfels_size; out.capacity = (uint32_t)s.surfels.width();
  return out;
}
bahip_pcg_layout Layout(u32 unknown_count, bool optimize_poses, bool optimize_geometry, bool use_depth_residuals, bool use_descriptor_residuals,
                        bool optimize_depth_intrinsics, bool optimize_color_intrinsics, u32 surfel_start, u32 depth_intrinsics_start,
                        u32 color_intrinsics_start) {
  bahip_pcg_layout L{};
  L.optimize_poses = optimize_poses; L.optimize_geometry = optimize_geometry;
  L.optimize_depth_intrinsics = optimize_depth_intrinsics; L.optimize_color_intrinsics = optimize_color_intrinsics;
  L.use_depth_residuals = use_depth_residuals; L.use_descriptor_residuals = use_descriptor_residuals;
  L.unknown_count = unknown_count; L.surfel_unknown_start_index = surfel_start;
  L.depth_intrinsics_unknown_start_index = depth_intrinsics_start; L.color_intrinsics_unknown_start_index = color_intrinsics_start;
  return L;
}
}  // namespace

void PCGInitCUDA(cudaStream_t stream, const SurfelProjectionParameters& s, const DepthToColorPixelCorner& depth_to_color,
                 const PixelCenterUnprojector& /*depth_unprojector: derived from the depth camera*/, const PixelCornerProjector& color_projector,
                 cudaTextureObject_t color_texture, u32 kf_pose_unknown_index, u32 surfel_unknown_start_index, bool optimize_poses,
                 bool optimize_geometry, bool use_depth_residuals, bool use_descriptor_residuals, bool optimize_depth_intrinsics,
                 bool optimize_color_intrinsics, u32 depth_intrinsics_unknown_start_index, u32 color_intrinsics_unknown_start_index,
                 CUDABuffer_<PCGScalar>* pcg_r, CUDABuffer_<PCGScalar>* pcg_M, u32 surfels_size) {
  bahip_context* ctx = Ctx(stream);
  BindFromProjection(ctx, s, depth_to_color, color_projector);
  PcgSession& session = Session();
  if (!session.begun) {
    // Whether a keyframe's pose is an unknown is said per call (the reference passes optimize_poses = false for the gauge
    // keyframe), so the layout itself only needs the block boundaries.  The unknown count is NOT the width of the caller's
    // vectors: the reference allocates them for max_unknown_count (B/direct_ba_pcg.cc:249-268) and hands the count of this
    // outer iteration to PCGInit2CUDA only.  It follows from the block starts (the unknown ordering of :270-300: poses,
    // surfels, 4 + 1 + cfactor cells, 4): the end of the last block present.  With poses as the only unknowns nothing but the
    // width bounds it; PCGInit2CUDA then says how many of those entries exist (a smaller head is accepted there).
    const u32 cfactor_cells = (u32)s.depth_params.cfactor_buffer.width() * (u32)s.depth_params.cfactor_buffer.height();
    // (poses only: at most 65 536 keyframes' worth of exact accumulators, not 72 bytes for every entry of a vector that was
    // sized for all surfels)
    u32 unknown_count = std::min<u32>((u32)pcg_r->width(), 6u * 65536u);
    if (optimize_color_intrinsics) unknown_count = color_intrinsics_unknown_start_index + 4;
    else if (optimize_depth_intrinsics) unknown_count = depth_intrinsics_unknown_start_index + 5 + cfactor_cells;
    else if (optimize_geometry) unknown_count = surfel_unknown_start_index + (use_descriptor_residuals ? 3u : 1u) * surfels_size;
    CHECK_LE(unknown_count, (u32)pcg_r->width()) << "the PCG vectors are narrower than the unknown layout";
    session.layout = Layout(unknown_count, true, optimize_geometry, use_depth_residuals, use_descriptor_residuals, optimize_depth_intrinsics,
                            optimize_color_intrinsics, surfel_unknown_start_index, depth_intrinsics_unknown_start_index,
                            color_intrinsics_unknown_start_index);
    session.surfels_size = surfels_size;
    BAHIP_CHECKED_CALL(bahip_pcg_begin(ctx, &session.layout, surfels_size));
    session.begun = true;
  }
  if (optimize_poses) CHECK_LE(kf_pose_unknown_index + 6, session.layout.unknown_count) << "pose unknown outside the layout";
  const bahip_frame frame = FrameFromProjection(s, color_texture);
  const bahip_surfels surfels = SurfelsFromProjection(s);
  BAHIP_CHECKED_CALL(bahip_pcg_init(ctx, &session.layout, &frame, &s.frame_T_global.row0.x, kf_pose_unknown_index, optimize_poses ? 1 : 0, &surfels,
                                    pcg_r->address(), pcg_M->address()));
}

void PCGInit2CUDA(cudaStream_t stream, u32 unknown_count, u32 /*a_unknown_index: depth_intrinsics_unknown_start_index + 4*/, float a,
                  const CUDABuffer_<PCGScalar>& pcg_r, const CUDABuffer_<PCGScalar>& pcg_M, CUDABuffer_<PCGScalar>* pcg_delta,
                  CUDABuffer_<PCGScalar>* pcg_g, CUDABuffer_<PCGScalar>* pcg_p, CUDABuffer_<PCGScalar>* pcg_alpha_n) {
  bahip_context* ctx = Ctx(stream);
  PcgSession& session = Session();
  CHECK(session.begun) << "PCGInit2CUDA without a preceding PCGInitCUDA";
  CHECK_LE(unknown_count, session.layout.unknown_count) << "PCGInit2CUDA: more unknowns than the layout PCGInitCUDA described";
  if (session.layout.optimize_geometry || session.layout.optimize_depth_intrinsics || session.layout.optimize_color_intrinsics)
    CHECK_EQ(unknown_count, session.layout.unknown_count) << "PCGInit2CUDA: the unknown count differs from the block layout PCGInitCUDA described";
  BAHIP_CHECKED_CALL(bahip_pcg_init2(ctx, &session.layout, session.surfels_size, a, pcg_r.address(), pcg_M.address(), pcg_delta->address(),
                                     pcg_g->address(), pcg_p->address(), pcg_alpha_n->address()));
  session.begun = false;   // the next PCGInitCUDA starts a new outer iteration
}

void PCGStep1CUDA(cudaStream_t stream, u32 /*unknown_count*/, const SurfelProjectionParameters& s, const DepthToColorPixelCorner& depth_to_color,
                  const PixelCenterUnprojector&, const PixelCornerProjector& color_projector, cudaTextureObject_t color_texture,
                  u32 kf_pose_unknown_index, u32 /*surfel_unknown_start_index*/, bool optimize_poses, bool /*optimize_geometry*/,
                  bool /*use_depth_residuals*/, bool /*use_descriptor_residuals*/, bool /*optimize_depth_intrinsics*/,
                  bool /*optimize_color_intrinsics*/, u32 /*depth_intrinsics_unknown_start_index*/, u32 /*a_unknown_index*/,
                  u32 /*color_intrinsics_unknown_start_index*/, CUDABuffer_<PCGScalar>* pcg_p, CUDABuffer_<PCGScalar>* pcg_g,
                  CUDABuffer_<PCGScalar>* /*pcg_alpha_d: written by PCGStep2CUDA*/, u32 /*surfels_size*/) {
  bahip_context* ctx = Ctx(stream);
  BindFromProjection(ctx, s, depth_to_color, color_projector);
  const bahip_frame frame = FrameFromProjection(s, color_texture);
  const bahip_surfels surfels = SurfelsFromProjection(s);
  BAHIP_CHECKED_CALL(bahip_pcg_step1(ctx, &Session().layout, &frame, &s.frame_T_global.row0.x, kf_pose_unknown_index, optimize_poses ? 1 : 0, &surfels,
                                     pcg_p->address(), pcg_g->address()));
}

void PCGStep2CUDA(cudaStream_t stream, u32 /*unknown_count*/, u32 /*a_unknown_index*/, const CUDABuffer_<PCGScalar>& pcg_r,
                  const CUDABuffer_<PCGScalar>& pcg_M, CUDABuffer_<PCGScalar>* pcg_delta, CUDABuffer_<PCGScalar>* pcg_g,
                  CUDABuffer_<PCGScalar>* pcg_p, CUDABuffer_<PCGScalar>* pcg_alpha_n, CUDABuffer_<PCGScalar>* pcg_alpha_d,
                  CUDABuffer_<PCGScalar>* pcg_beta_n) {
  PcgSession& session = Session();
  BAHIP_CHECKED_CALL(bahip_pcg_step2(Ctx(stream), &session.layout, session.surfels_size, pcg_r.address(), pcg_M.address(), pcg_delta->address(),
                                     pcg_g->address(), pcg_p->address(), pcg_alpha_n->address(), pcg_alpha_d->address(), pcg_beta_n->address()));
}

void PCGStep3CUDA(cudaStream_t stream, u32 /*unknown_count*/, CUDABuffer_<PCGScalar>* pcg_g, CUDABuffer_<PCGScalar>* pcg_p,
                  CUDABuffer_<PCGScalar>* pcg_alpha_n, CUDABuffer_<PCGScalar>* pcg_beta_n) {
  PcgSession& session = Session();
  BAHIP_CHECKED_CALL(bahip_pcg_step3(Ctx(stream), &session.layout, session.surfels_size, pcg_g->address(), pcg_p->address(), pcg_alpha_n->address(),
                                     pcg_beta_n->address()));
}

void UpdateSurfelsFromPCGDeltaCUDA(cudaStream_t stream, u32 surfels_size, CUDABuffer_<float>* surfels, bool use_descriptor_residuals,
                                   u32 surfel_unknown_start_index, const CUDABuffer_<PCGScalar>& pcg_delta) {
  bahip_surfels s{};
  s.data = surfels->address(); s.pitch_bytes = (uint32_t)surfels->pitch(); s.surfels_size = surfels_size; s.capacity = (uint32_t)surfels->width();
  BAHIP_CHECKED_CALL(bahip_update_surfels_from_pcg_delta(Ctx(stream), &s, use_descriptor_residuals ? 1 : 0, surfel_unknown_start_index, pcg_delta.address()));
}

void UpdateCFactorsFromPCGDeltaCUDA(cudaStream_t stream, CUDABuffer_<float>* /*cfactor_buffer: the one bound with the depth parameters*/,
                                    u32 cfactor_unknown_start_index, const CUDABuffer_<PCGScalar>& pcg_delta) {
  BAHIP_CHECKED_CALL(bahip_update_cfactors_from_pcg_delta(Ctx(stream), cfactor_unknown_start_index, pcg_delta.address()));
}

}  // namespace vis
