// c_api.cc -- include/badslam_directba.h: flat C view of vis::DirectBA / vis::Keyframe.
#include "../../include/badslam_directba.h"

#include "direct_ba.h"
#include "rgbd_io.h"   // Point3fC3u8Nf

using namespace vis;

struct dba_handle {
  std::unique_ptr<DirectBA> ba;
  int width, height;
};

extern "C" {

dba_handle* dba_create(int max_surfel_count, float raw_to_float_depth, float baseline_fx, int sparse_surfel_cell_size,
                       float surfel_merge_dist_factor, int boot1, int boot2, int min_observation_count, int width, int height,
                       const float color_camera[4], const float depth_camera[4], int use_depth_residuals,
                       int use_descriptor_residuals) {
  if (bahip_device_count() <= 0) return nullptr;   // no CPU fallback
  dba_handle* h = new dba_handle();
  h->width = width; h->height = height;
  PinholeCamera4f cc(width, height, color_camera), dc(width, height, depth_camera);
  h->ba.reset(new DirectBA(max_surfel_count, raw_to_float_depth, baseline_fx, sparse_surfel_cell_size, surfel_merge_dist_factor,
                           boot1, boot2, min_observation_count, cc, dc, 0, use_depth_residuals != 0,
                           use_descriptor_residuals != 0, nullptr, SE3f()));
  return h;
}

void dba_destroy(dba_handle* h) { delete h; }

int dba_add_keyframe(dba_handle* h, void* stream, const uint16_t* depth_image, const uint8_t* rgb_image, const float pose[7]) {
  Image<u16> depth(h->width, h->height);
  memcpy(depth.data(), depth_image, sizeof(u16) * (size_t)h->width * h->height);
  Image<Vec3u8> color(h->width, h->height);
  memcpy(color.data(), rgb_image, 3 * (size_t)h->width * h->height);
  shared_ptr<Keyframe> kf(new Keyframe(stream, (u32)h->ba->keyframes().size(), h->ba->depth_params(), h->ba->depth_camera(), depth,
                                       color, SE3f(pose)));
  h->ba->AddKeyframe(kf);
  return kf->id();
}

int dba_keyframe_count(dba_handle* h) { return (int)h->ba->keyframes().size(); }

static Keyframe* get_kf(dba_handle* h, int id) {
  if (id < 0 || id >= (int)h->ba->keyframes().size()) return nullptr;
  return h->ba->keyframes()[id].get();
}

int dba_get_keyframe_pose(dba_handle* h, int id, float pose[7]) {
  Keyframe* kf = get_kf(h, id);
  if (!kf) return 1;
  memcpy(pose, kf->global_T_frame().data(), 7 * sizeof(float));
  return 0;
}
int dba_set_keyframe_pose(dba_handle* h, int id, const float pose[7]) {
  Keyframe* kf = get_kf(h, id);
  if (!kf) return 1;
  kf->set_global_T_frame(SE3f(pose));
  return 0;
}
int dba_get_keyframe_activation(dba_handle* h, int id) {
  Keyframe* kf = get_kf(h, id);
  return kf ? (int)kf->activation() : -1;
}

int dba_download_keyframe_image(dba_handle* h, void* stream, int id, int which, void* out) {
  Keyframe* kf = get_kf(h, id);
  if (!kf) return 1;
  switch (which) {
    case 0: kf->depth_buffer().DownloadAsync(stream, static_cast<u16*>(out)); break;
    case 1: kf->normals_buffer().DownloadAsync(stream, static_cast<u16*>(out)); break;
    case 2: kf->radius_buffer().DownloadAsync(stream, static_cast<u16*>(out)); break;
    case 3: kf->color_buffer().DownloadAsync(stream, static_cast<uchar4*>(out)); break;
    default: return 1;
  }
  return 0;
}
int dba_upload_keyframe_image(dba_handle* h, void* stream, int id, int which, const void* in) {
  Keyframe* kf = get_kf(h, id);
  if (!kf) return 1;
  // the reference's tests do exactly this const_cast (test_geometry_optimization_geometric_residual.cc:121-136)
  switch (which) {
    case 0: const_cast<CUDABuffer<u16>&>(kf->depth_buffer()).UploadAsync(stream, static_cast<const u16*>(in)); break;
    case 1: const_cast<CUDABuffer<u16>&>(kf->normals_buffer()).UploadAsync(stream, static_cast<const u16*>(in)); break;
    case 2: const_cast<CUDABuffer<u16>&>(kf->radius_buffer()).UploadAsync(stream, static_cast<const u16*>(in)); break;
    case 3: const_cast<CUDABuffer<uchar4>&>(kf->color_buffer()).UploadAsync(stream, static_cast<const uchar4*>(in)); break;
    default: return 1;
  }
  kf->RefreshPlanes(static_cast<hipStream_t>(stream));
  return 0;
}
int dba_keyframe_exists(dba_handle* h, int id) { return (id >= 0 && id < (int)h->ba->keyframes().size() && h->ba->keyframes()[id]) ? 1 : 0; }
int dba_merge_keyframes(dba_handle* h, void* stream, int approx_merge_count) {
  h->ba->MergeKeyframes(static_cast<hipStream_t>(stream), nullptr, (vis::usize)approx_merge_count);
  return 0;
}
int dba_export_point_count(dba_handle* h, void* stream, unsigned* count_out) {
  std::vector<vis::Point3fC3u8Nf> cloud;
  h->ba->ExportToPointCloud(static_cast<hipStream_t>(stream), &cloud);
  *count_out = (unsigned)cloud.size();
  return 0;
}
int dba_delete_keyframe(dba_handle* h, int id) {
  if (!get_kf(h, id)) return 1;
  h->ba->DeleteKeyframe(id, nullptr);
  return 0;
}

int dba_create_surfels_for_keyframe(dba_handle* h, void* stream, int filter_new_surfels, int id) {
  if (!get_kf(h, id)) return 1;
  h->ba->CreateSurfelsForKeyframe(stream, filter_new_surfels != 0, h->ba->keyframes()[id]);
  return 0;
}

int dba_estimate_frame_pose(dba_handle* h, void* stream, int id, const float init[7], float out[7]) {
  Keyframe* kf = get_kf(h, id);
  if (!kf) return 1;
  SE3f result;
  h->ba->EstimateFramePose(stream, SE3f(init), kf->depth_buffer(), kf->normals_buffer(), kf->color_texture(), &result, false);
  memcpy(out, result.data(), 7 * sizeof(float));
  return 0;
}

int dba_bundle_adjustment(dba_handle* h, void* stream, int optimize_depth_intrinsics, int optimize_color_intrinsics,
                          int do_surfel_updates, int optimize_poses, int optimize_geometry, int min_iterations, int max_iterations,
                          int use_pcg, int window_start, int window_end, int increase_ba_iteration_count, int* iterations_done,
                          int* converged, int pcg_max_inner_iterations) {
  bool conv = false;
  int done = 0;
  h->ba->BundleAdjustment(stream, optimize_depth_intrinsics != 0, optimize_color_intrinsics != 0, do_surfel_updates != 0,
                          optimize_poses != 0, optimize_geometry != 0, min_iterations, max_iterations, use_pcg != 0, window_start,
                          window_end, increase_ba_iteration_count != 0, &done, &conv, 0, nullptr, pcg_max_inner_iterations);
  if (iterations_done) *iterations_done = done;
  if (converged) *converged = conv ? 1 : 0;
  return 0;
}

uint32_t dba_surfel_count(dba_handle* h) { return h->ba->surfel_count(); }
uint32_t dba_surfels_size(dba_handle* h) { return h->ba->surfels_size(); }
int dba_set_surfel_count(dba_handle* h, uint32_t surfel_count, uint32_t surfels_size) {
  h->ba->SetSurfelCount(surfel_count, surfels_size);
  return 0;
}

int dba_sort_surfels_spatially(dba_handle* h, void* stream, float grid_cell_size) {
  h->ba->SortSurfelsSpatially(stream, grid_cell_size);
  return 0;
}

int dba_set_batched_creation(dba_handle* h, int enabled) {
  h->ba->SetBatchedCreation(enabled != 0);
  return 0;
}
int dba_set_spatial_sort_cell_size(dba_handle* h, float grid_cell_size) {
  h->ba->SetSpatialSortCellSize(grid_cell_size);
  return 0;
}
uint32_t dba_unsorted_surfels(dba_handle* h) { return h->ba->unsorted_surfels(); }

int dba_download_surfels(dba_handle* h, void* stream, int rows, uint32_t count, float* out) {
  auto s = h->ba->surfels();
  for (int r = 0; r < rows; ++r)
    s->DownloadPartAsync((size_t)r * s->ToCUDA().pitch(), (size_t)count * sizeof(float), stream, out + (size_t)r * count);
  return 0;
}
int dba_upload_surfels(dba_handle* h, void* stream, int rows, uint32_t count, const float* in) {
  auto s = h->ba->surfels();
  if ((int)count > s->width()) return 1;
  for (int r = 0; r < rows; ++r)
    s->UploadPartAsync((size_t)r * s->ToCUDA().pitch(), (size_t)count * sizeof(float), stream, in + (size_t)r * count);
  bahip_context_surfels_rearranged(h->ba->backend_context());   // (other surfels in every tile: the sweeps' run order is rebuilt)
  return 0;
}

int dba_get_cameras(dba_handle* h, float color_camera[4], float depth_camera[4], float* a) {
  memcpy(color_camera, h->ba->color_camera().parameters(), 4 * sizeof(float));
  memcpy(depth_camera, h->ba->depth_camera().parameters(), 4 * sizeof(float));
  *a = h->ba->a();
  return 0;
}
int dba_set_cameras(dba_handle* h, const float color_camera[4], const float depth_camera[4], float a) {
  h->ba->SetColorCamera(PinholeCamera4f(h->width, h->height, color_camera));
  h->ba->SetDepthCamera(PinholeCamera4f(h->width, h->height, depth_camera));
  h->ba->a() = a;
  return 0;
}
int dba_cfactor_size(dba_handle* h, int* width, int* height) {
  *width = h->ba->cfactor_buffer()->width();
  *height = h->ba->cfactor_buffer()->height();
  return 0;
}
int dba_download_cfactor(dba_handle* h, void* stream, float* out) {
  h->ba->cfactor_buffer()->DownloadAsync(stream, out);
  return 0;
}
int dba_clear_cfactor(dba_handle* h, void* stream) {
  h->ba->cfactor_buffer()->Clear(0, stream);
  return 0;
}
int dba_set_ba_iteration_counts(dba_handle* h, int count, int last) {
  h->ba->SetBAIterationCount(count);
  h->ba->SetLastBAIterationCount(last);
  return 0;
}
int dba_set_surfel_sharding(dba_handle* h, int rank, int world, uint32_t chunk) {
  h->ba->SetSurfelSharding(rank, world, chunk);
  return 0;
}
int dba_set_row_major_creation(dba_handle* h, int enabled) {
  h->ba->SetRowMajorCreation(enabled != 0);
  return 0;
}
int dba_set_fast_arithmetic(dba_handle* h, int enabled) {
  h->ba->SetFastArithmetic(enabled != 0);
  return 0;
}
int dba_set_sum_classes(dba_handle* h, int classes) {
  h->ba->SetSumClasses(classes);
  return 0;
}
int dba_set_keyframe_sharding(dba_handle* h, int rank, int world) {
  h->ba->SetKeyframeSharding(rank, world);
  return 0;
}
int dba_set_pcg_gauge_keyframe(dba_handle* h, int id) {
  h->ba->SetPCGGaugeKeyframe(id);
  return 0;
}
int dba_last_stats(dba_handle* h, int* pose_rounds, int* pose_steps, int* pcg_inner_steps) {
  if (pose_rounds) *pose_rounds = h->ba->last_pose_rounds();
  if (pose_steps) *pose_steps = h->ba->last_pose_steps();
  if (pcg_inner_steps) *pcg_inner_steps = h->ba->last_pcg_inner_steps();
  return 0;
}
bahip_context* dba_backend_context(dba_handle* h) { return h->ba->backend_context(); }
int dba_keyframe_frame(dba_handle* h, int id, bahip_frame* out) {
  Keyframe* kf = get_kf(h, id);
  if (!kf || !out) return 1;
  *out = kf->ToBahipFrame();
  return 0;
}
int dba_surfels_struct(dba_handle* h, bahip_surfels* out) {
  if (!out) return 1;
  *out = h->ba->SurfelsStruct(true);
  return 0;
}
int dba_keyframe_covisibility(dba_handle* h, int id, int* out, int capacity) {
  Keyframe* kf = get_kf(h, id);
  if (!kf) return -1;
  const auto& list = kf->co_visibility_list();
  for (int i = 0; i < (int)list.size() && i < capacity; ++i) out[i] = list[i];
  return (int)list.size();
}
int dba_bind_scene(dba_handle* h, void* stream) {
  h->ba->BindScene(static_cast<hipStream_t>(stream));
  return 0;
}

}  // extern "C"
