// direct_ba.cc -- host driver of the MI355X direct-BA backend behind the reference's DirectBA surface.
// Control flow follows B/direct_ba.cc, B/direct_ba_alternating.cc and B/direct_ba_pcg.cc
// (B/ = applications/badslam/src/badslam/ of ETH3D/badslam); all device work goes through the C ABI
// (include/badslam_hip.h).  Where the reference loops over keyframes and launches one kernel per
// keyframe, this driver makes ONE call per stage.
#include "direct_ba.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <memory>

namespace vis {

namespace {
int ToBahipActivation(Keyframe::Activation a) { return static_cast<int>(a); }
}  // namespace

DirectBA::DirectBA(int max_surfel_count, float raw_to_float_depth, float baseline_fx, int sparse_surfel_cell_size,
                   float surfel_merge_dist_factor, int min_observation_count_while_bootstrapping_1,
                   int min_observation_count_while_bootstrapping_2, int min_observation_count,
                   const PinholeCamera4f& color_camera_initial_estimate, const PinholeCamera4f& depth_camera_initial_estimate,
                   int pyramid_level_for_color, bool use_depth_residuals, bool use_descriptor_residuals, void* render_window,
                   const SE3f& global_T_anchor_frame)
    : color_camera_(color_camera_initial_estimate), pyramid_level_for_color_(pyramid_level_for_color),
      depth_camera_(depth_camera_initial_estimate), use_depth_residuals_(use_depth_residuals),
      use_descriptor_residuals_(use_descriptor_residuals),
      min_observation_count_while_bootstrapping_1_(min_observation_count_while_bootstrapping_1),
      min_observation_count_while_bootstrapping_2_(min_observation_count_while_bootstrapping_2),
      min_observation_count_(min_observation_count), surfel_merge_dist_factor_(surfel_merge_dist_factor),
      global_T_anchor_frame_(global_T_anchor_frame) {
  CHECK(render_window == nullptr) << "the render window is not part of the BA backend";
  BAHIP_CHECKED_CALL(bahip_context_create(&ctx_, nullptr));
  // B/direct_ba.cc:107-121
  depth_params_.a = 0;
  cfactor_buffer_.reset(new CUDABuffer<float>((depth_camera_.height() - 1) / sparse_surfel_cell_size + 1,
                                              (depth_camera_.width() - 1) / sparse_surfel_cell_size + 1));
  cfactor_buffer_->Clear(0, nullptr);
  BAHIP_CHECKED_CALL(bahip_stream_synchronize(nullptr));
  depth_params_.cfactor_buffer = cfactor_buffer_->ToCUDA();
  depth_params_.raw_to_float_depth = raw_to_float_depth;
  depth_params_.baseline_fx = baseline_fx;
  depth_params_.sparse_surfel_cell_size = sparse_surfel_cell_size;
  surfels_.reset(new CUDABuffer<float>(kSurfelAttributeCount, max_surfel_count));
  active_surfels_.reset(new CUDABuffer<u8>(1, max_surfel_count));
  for (int i = 0; i < kMergeBufferCount; ++i)
    supporting_surfels_[i].reset(new CUDABuffer<u32>(depth_camera_.height(), depth_camera_.width()));
}

DirectBA::~DirectBA() { bahip_context_destroy(ctx_); }

// ---- keyframe bookkeeping (B/direct_ba.cc:214-338,549-564,710-738) --------------------------------------------
void DirectBA::AddKeyframe(const shared_ptr<Keyframe>& new_keyframe) {
  new_keyframe->SetID(static_cast<int>(keyframes_.size()));
  DetermineNewKeyframeCoVisibility(new_keyframe);
  keyframes_.push_back(new_keyframe);
}

void DirectBA::DeleteKeyframe(int keyframe_index, void* /*loop_detector*/) {
  shared_ptr<Keyframe> frame_to_delete = keyframes_[keyframe_index];
  for (int covis_index : frame_to_delete->co_visibility_list()) {
    auto& list = keyframes_[covis_index]->co_visibility_list();
    auto it = std::find(list.begin(), list.end(), keyframe_index);
    if (it != list.end()) list.erase(it);
  }
  keyframes_[keyframe_index].reset();
}

void DirectBA::DetermineNewKeyframeCoVisibility(const shared_ptr<Keyframe>& new_keyframe) {
  CameraFrustum new_frustum(depth_camera_, new_keyframe->min_depth(), new_keyframe->max_depth(), new_keyframe->global_T_frame());
  for (const shared_ptr<Keyframe>& keyframe : keyframes_) {
    if (!keyframe) continue;
    CameraFrustum frustum(depth_camera_, keyframe->min_depth(), keyframe->max_depth(), keyframe->global_T_frame());
    if (new_frustum.Intersects(frustum)) {
      new_keyframe->co_visibility_list().push_back(keyframe->id());
      keyframe->co_visibility_list().push_back(new_keyframe->id());
      if (keyframe->activation() == Keyframe::Activation::kInactive) keyframe->SetActivation(Keyframe::Activation::kCovisibleActive);
    }
  }
}

void DirectBA::AssignColors(hipStream_t stream) {
  BindScene(stream);
  const bahip_surfels s = SurfelsStruct(/*with_active*/ false);
  BAHIP_CHECKED_CALL(bahip_assign_colors(ctx_, &s));
}

void DirectBA::UpdateKeyframeCoVisibility(const shared_ptr<Keyframe>& keyframe) {
  for (int covis_index : keyframe->co_visibility_list()) {
    auto& list = keyframes_[covis_index]->co_visibility_list();
    auto it = std::find(list.begin(), list.end(), keyframe->id());
    if (it != list.end()) list.erase(it);
  }
  keyframe->co_visibility_list().clear();
  CameraFrustum frustum(depth_camera_, keyframe->min_depth(), keyframe->max_depth(), keyframe->global_T_frame());
  for (const shared_ptr<Keyframe>& other : keyframes_) {
    if (!other) continue;
    CameraFrustum other_frustum(depth_camera_, other->min_depth(), other->max_depth(), other->global_T_frame());
    if (frustum.Intersects(other_frustum)) {
      keyframe->co_visibility_list().push_back(other->id());
      other->co_visibility_list().push_back(keyframe->id());
    }
  }
}

void DirectBA::DetermineCovisibleActiveKeyframes() {
  for (const shared_ptr<Keyframe>& keyframe : keyframes_) {
    if (!keyframe || keyframe->activation() != Keyframe::Activation::kActive) continue;
    for (int covisible_index : keyframe->co_visibility_list()) {
      shared_ptr<Keyframe>& other = keyframes_[covisible_index];
      if (other && other->activation() == Keyframe::Activation::kInactive) other->SetActivation(Keyframe::Activation::kCovisibleActive);
    }
  }
}

// B/direct_ba.cc:251-338: drops the keyframes that are closest to their neighbours (frees memory).
void DirectBA::MergeKeyframes(hipStream_t /*stream*/, void* loop_detector, usize approx_merge_count) {
  constexpr float kMaxAngleDifference = 0.5f * 1.57079632679f;
  constexpr float kMaxEuclideanDistance = 0.3f;
  if (keyframes_.size() <= 1) return;
  struct Candidate { float distance; usize prev_id, id, next_id; };
  vector<Candidate> distances;
  float prev_half_distance = 0;
  usize prev_keyframe_id = 0;
  for (usize id = 0; id + 1 < keyframes_.size(); ++id) {
    const shared_ptr<Keyframe>& keyframe = keyframes_[id];
    if (!keyframe) continue;
    const Keyframe* next = nullptr;
    for (usize n = id + 1; n < keyframes_.size(); ++n) if (keyframes_[n]) { next = keyframes_[n].get(); break; }
    if (!next) break;
    float Ra[9], Rb[9];
    keyframe->global_T_frame().rotationMatrix(Ra);
    next->global_T_frame().rotationMatrix(Rb);
    const float dot = Ra[2] * Rb[2] + Ra[5] * Rb[5] + Ra[8] * Rb[8];   // z axes
    const float angle_difference = acosf(dot);
    if (angle_difference > kMaxAngleDifference) continue;
    const float* ta = keyframe->global_T_frame().translation();
    const float* tb = next->global_T_frame().translation();
    const float dist = sqrtf((ta[0] - tb[0]) * (ta[0] - tb[0]) + (ta[1] - tb[1]) * (ta[1] - tb[1]) + (ta[2] - tb[2]) * (ta[2] - tb[2]));
    if (dist > kMaxEuclideanDistance) continue;
    const float next_half_distance = dist + (0.5f / 1.57079632679f) * angle_difference;
    if (id > 0) distances.push_back({prev_half_distance + next_half_distance, prev_keyframe_id, id, (usize)next->id()});
    prev_half_distance = next_half_distance;
    prev_keyframe_id = id;
  }
  const usize count = std::min(approx_merge_count, distances.size());
  std::partial_sort(distances.begin(), distances.begin() + count, distances.end(),
                    [](const Candidate& a, const Candidate& b) { return a.distance < b.distance; });
  for (usize i = 0; i < count; ++i) {
    const Candidate& m = distances[i];
    if (!keyframes_[m.prev_id] || !keyframes_[m.id] || !keyframes_[m.next_id]) continue;
    DeleteKeyframe((int)m.id, loop_detector);
    LOG(WARNING) << "Deleted keyframe with ID " << m.id;
  }
}

// ---- scene binding ---------------------------------------------------------------------------------------------
bahip_surfels DirectBA::SurfelsStruct(bool with_active) const {
  bahip_surfels s;
  s.data = surfels_->ToCUDA().address();
  s.pitch_bytes = (uint32_t)surfels_->ToCUDA().pitch();
  s.active = with_active ? active_surfels_->ToCUDA().address() : nullptr;
  s.surfels_size = surfels_size_;
  s.capacity = (uint32_t)surfels_->width();
  return s;
}

void DirectBA::BindScene(hipStream_t stream) {
  static const bool host_timing = getenv("BADSLAM_HOST_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = host_timing ? now() : 0;
  BAHIP_CHECKED_CALL(bahip_context_set_stream(ctx_, stream));
  const bahip_camera cc = ToBahipCamera(color_camera_), dc = ToBahipCamera(depth_camera_);
  const bahip_depth_params dp = ToBahipDepthParams(depth_params_);
  BAHIP_CHECKED_CALL(bahip_set_intrinsics(ctx_, &cc, &dc, &dp));
  vector<bahip_keyframe> list;
  bound_ids_.clear();
  id_to_bound_.assign(keyframes_.size(), -1);
  for (const shared_ptr<Keyframe>& keyframe : keyframes_) {
    if (!keyframe) continue;
    bahip_keyframe k;
    k.frame = keyframe->ToBahipFrame();
    memcpy(k.global_T_frame, keyframe->global_T_frame().data(), 7 * sizeof(float));
    k.activation = ToBahipActivation(keyframe->activation());
    id_to_bound_[keyframe->id()] = (int)list.size();
    bound_ids_.push_back(keyframe->id());
    list.push_back(k);
  }
  const double t1 = host_timing ? now() : 0;
  BAHIP_CHECKED_CALL(bahip_set_keyframes(ctx_, list.data(), (int)list.size()));
  const double t2 = host_timing ? now() : 0;
  // co-visibility lists over bound indices: the activation state machine of the alternating scheme runs on the device table
  vector<int> offsets(1, 0), indices;
  for (int id : bound_ids_) {
    for (int other : keyframes_[id]->co_visibility_list())
      if (other >= 0 && other < (int)id_to_bound_.size() && id_to_bound_[other] >= 0) indices.push_back(id_to_bound_[other]);
    offsets.push_back((int)indices.size());
  }
  const double t3 = host_timing ? now() : 0;
  BAHIP_CHECKED_CALL(bahip_set_covisibility(ctx_, offsets.data(), indices.data(), (int)bound_ids_.size()));
  if (host_timing)
    fprintf(stderr, "[BindScene, us] list %.0f | set_keyframes %.0f | covisibility lists (%zu entries) %.0f | set_covisibility %.0f\n", t1 - t0,
            t2 - t1, indices.size(), t3 - t2, now() - t3);
}

// ---- surfel sharding: whole-cloud phases -----------------------------------------------------------------------------
void DirectBA::SetSurfelSharding(int rank, int world, u32 chunk) {
  CHECK(world >= 1 && rank >= 0 && rank < world && chunk > 0 && chunk % 64 == 0) << "bad surfel partition";
  CHECK(world == 1 || keyframe_shard_world_ == 1) << "surfel and keyframe sharding exclude each other";
  CHECK_EQ(whole_cloud_depth_, 0);
  shard_rank_ = rank; shard_world_ = world; shard_chunk_ = chunk;
  if (world > 1 && !other_surfels_) {
    CHECK_EQ(surfels_->width() % 8, 0) << "surfel sharding needs max_surfel_count to be a multiple of 8";
    other_surfels_.reset(new CUDABuffer<float>(kSurfelAttributeCount, surfels_->width()));
    other_active_surfels_.reset(new CUDABuffer<u8>(1, surfels_->width()));
  }
}

void DirectBA::SetSumClasses(int classes) { BAHIP_CHECKED_CALL(bahip_context_set_sum_classes(ctx_, classes)); }
void DirectBA::SetFastArithmetic(bool enabled) { BAHIP_CHECKED_CALL(bahip_context_set_arithmetic(ctx_, enabled ? BAHIP_ARITHMETIC_FAST : BAHIP_ARITHMETIC_EXACT)); }
void DirectBA::SetRowMajorCreation(bool enabled) { BAHIP_CHECKED_CALL(bahip_context_set_creation_order(ctx_, enabled ? 1 : 0)); }

void DirectBA::SetKeyframeSharding(int rank, int world) {
  CHECK_EQ(shard_world_, 1) << "surfel and keyframe sharding exclude each other";
  BAHIP_CHECKED_CALL(bahip_context_set_keyframe_sharding(ctx_, rank, world));
  keyframe_shard_world_ = world;
}

void DirectBA::EnterWholeCloud(hipStream_t stream) {
  if (shard_world_ <= 1 || whole_cloud_depth_++ > 0) return;
  BAHIP_CHECKED_CALL(bahip_context_set_stream(ctx_, stream));
  const bahip_surfels shard = SurfelsStruct();
  std::swap(surfels_, other_surfels_);
  std::swap(active_surfels_, other_active_surfels_);
  bahip_surfels cloud = SurfelsStruct();
  uint32_t size = 0, count = 0;
  BAHIP_CHECKED_CALL(bahip_gather_surfel_shards(ctx_, &shard, surfel_count_, shard_rank_, shard_world_, shard_chunk_, &cloud, &size, &count));
  Lock();
  surfels_size_ = size;
  surfel_count_ = count;
  Unlock();
}

void DirectBA::LeaveWholeCloud(hipStream_t stream) {
  if (shard_world_ <= 1 || --whole_cloud_depth_ > 0) return;
  BAHIP_CHECKED_CALL(bahip_context_set_stream(ctx_, stream));
  CHECK_EQ(surfels_size_, surfel_count_) << "a whole-cloud phase must end compacted";
  const bahip_surfels cloud = SurfelsStruct();
  std::swap(surfels_, other_surfels_);
  std::swap(active_surfels_, other_active_surfels_);
  bahip_surfels shard = SurfelsStruct();
  uint32_t mine = 0;
  BAHIP_CHECKED_CALL(bahip_extract_surfel_shard(ctx_, &cloud, shard_rank_, shard_world_, shard_chunk_, &shard, &mine));
  Lock();
  surfels_size_ = mine;
  surfel_count_ = mine;
  Unlock();
}

// ---- surfel creation (B/direct_ba.cc:340-405) ---------------------------------------------------------------------
void DirectBA::CreateSurfelsForKeyframe(hipStream_t stream, bool filter_new_surfels, const shared_ptr<Keyframe>& keyframe) {
  WholeCloudScope whole_cloud(this, stream);
  // inside a batch (the BA loop's creation pass) the scene was bound once for all its keyframes: nothing the binding reads
  // (poses, activations, co-visibility lists, intrinsics) changes between the creations of one batch
  if (!creation_batch_bound_) BindScene(stream);
  vector<int> covis;
  for (int id : keyframe->co_visibility_list())
    if (id >= 0 && id < (int)id_to_bound_.size() && id_to_bound_[id] >= 0) covis.push_back(id_to_bound_[id]);
  uint32_t* sup[kMergeBufferCount];
  for (int i = 0; i < kMergeBufferCount; ++i) sup[i] = supporting_surfels_[i]->ToCUDA().address();
  const bahip_surfels s = SurfelsStruct();
  uint32_t new_surfel_count = 0;
  BAHIP_CHECKED_CALL(bahip_create_surfels_for_keyframe(ctx_, id_to_bound_[keyframe->id()], filter_new_surfels ? 1 : 0,
                                                       GetMinObservationCount(), covis.data(), (int)covis.size(), &s, sup,
                                                       (uint32_t)supporting_surfels_[0]->ToCUDA().pitch(), &new_surfel_count));
  if (bahip_context_take_capacity_exceeded(ctx_)) LOG(ERROR) << "Maximum surfel count exceeded! Retry with a higher max_surfel_count.";
  Lock();
  surfels_size_ += new_surfel_count;
  surfel_count_ += new_surfel_count;
  unsorted_surfels_ += new_surfel_count;
  Unlock();
}

// A batch of keyframes in one call of the backend (bahip_create_surfels_for_keyframes): the creations of B/direct_ba_alternating.cc:
// 389-425 in their order, the cloud's size on the device in between.
void DirectBA::CreateSurfelsForKeyframes(hipStream_t stream, bool filter_new_surfels, const vector<u32>& keyframe_ids) {
  if (keyframe_ids.empty()) return;
  if (!batched_creation_) {
    for (u32 id : keyframe_ids) CreateSurfelsForKeyframe(stream, filter_new_surfels, keyframes_[id]);
    return;
  }
  WholeCloudScope whole_cloud(this, stream);
  if (!creation_batch_bound_) BindScene(stream);
  vector<int> bound, offsets(1, 0), covis;
  for (u32 id : keyframe_ids) {
    bound.push_back(id_to_bound_[id]);
    for (int other : keyframes_[id]->co_visibility_list())
      if (other >= 0 && other < (int)id_to_bound_.size() && id_to_bound_[other] >= 0) covis.push_back(id_to_bound_[other]);
    offsets.push_back((int)covis.size());
  }
  uint32_t* sup[kMergeBufferCount];
  for (int i = 0; i < kMergeBufferCount; ++i) sup[i] = supporting_surfels_[i]->ToCUDA().address();
  const bahip_surfels s = SurfelsStruct();
  uint32_t new_surfel_count = 0;
  BAHIP_CHECKED_CALL(bahip_lifecycle_batch_set_keyframes(ctx_, bound.data(), (int)bound.size()));   // (nothing without an open batch)
  BAHIP_CHECKED_CALL(bahip_create_surfels_for_keyframes(ctx_, bound.data(), (int)bound.size(), filter_new_surfels ? 1 : 0, GetMinObservationCount(),
                                                        offsets.data(), covis.data(), &s, sup, (uint32_t)supporting_surfels_[0]->ToCUDA().pitch(),
                                                        &new_surfel_count));
  if (bahip_context_take_capacity_exceeded(ctx_)) LOG(ERROR) << "Maximum surfel count exceeded! Retry with a higher max_surfel_count.";
  Lock();
  surfels_size_ += new_surfel_count;
  surfel_count_ += new_surfel_count;
  unsorted_surfels_ += new_surfel_count;
  Unlock();
}

void DirectBA::MergeForKeyframe(const Keyframe& keyframe, bool defer_count) {
  uint32_t* sup[kMergeBufferCount];
  for (int i = 0; i < kMergeBufferCount; ++i) sup[i] = supporting_surfels_[i]->ToCUDA().address();
  const bahip_frame frame = keyframe.ToBahipFrame();
  float F[12];
  keyframe.frame_T_global().matrix3x4(F);
  const bahip_surfels s = SurfelsStruct();
  uint32_t merged = 0;
  BAHIP_CHECKED_CALL(bahip_determine_supporting_surfels(ctx_, 1, surfel_merge_dist_factor_, &frame, F, &s, sup,
                                                        (uint32_t)supporting_surfels_[0]->ToCUDA().pitch(), defer_count ? nullptr : &merged));
  surfel_count_ -= merged;
}

// Tile bounds for the per-keyframe sweeps of a batch (include/badslam_hip.h: bahip_lifecycle_batch_begin).
DirectBA::LifecycleBatch::LifecycleBatch(DirectBA* ba) : ba_(ba) {
  const bahip_surfels s = ba_->SurfelsStruct();
  BAHIP_CHECKED_CALL(bahip_lifecycle_batch_begin(ba_->ctx_, &s));
}
DirectBA::LifecycleBatch::~LifecycleBatch() { bahip_lifecycle_batch_end(ba_->ctx_); }

// The merges of a batch of keyframes (the BA loop's merge pass, the end tasks'): one lifecycle batch that knows its frames -- each
// keyframe's sweeps then run over the tiles it can see --, counts deferred to one read at the end.
void DirectBA::MergeForKeyframes(const vector<u32>& keyframe_ids) {
  vector<float> frames;
  for (u32 id : keyframe_ids) {
    if (!keyframes_[id]) continue;
    float F[12];
    keyframes_[id]->frame_T_global().matrix3x4(F);   // what MergeForKeyframe passes: the list is found by these coefficients
    frames.insert(frames.end(), F, F + 12);
  }
  if (frames.empty()) return;
  LifecycleBatch batch(this);
  BAHIP_CHECKED_CALL(bahip_lifecycle_batch_set_frames(ctx_, frames.data(), (int)(frames.size() / 12)));
  if (batched_creation_) {
    // one call for the batch: two dependent launches per keyframe instead of three (bahip_merge_surfels_for_keyframes)
    vector<bahip_frame> structs;
    for (u32 id : keyframe_ids)
      if (keyframes_[id]) structs.push_back(keyframes_[id]->ToBahipFrame());
    uint32_t* sup[kMergeBufferCount];
    for (int i = 0; i < kMergeBufferCount; ++i) sup[i] = supporting_surfels_[i]->ToCUDA().address();
    const bahip_surfels s = SurfelsStruct();
    BAHIP_CHECKED_CALL(bahip_merge_surfels_for_keyframes(ctx_, surfel_merge_dist_factor_, structs.data(), frames.data(), (int)structs.size(), &s, sup,
                                                         (uint32_t)supporting_surfels_[0]->ToCUDA().pitch(), nullptr));
  } else {
    for (u32 id : keyframe_ids)
      if (keyframes_[id]) MergeForKeyframe(*keyframes_[id], /*defer_count*/ true);
  }
  TakeDeferredMergeCount();
}

// The merges of a batch of keyframes run back to back on the stream; their total is read once, before the compaction that needs it
// (the reference reads one count per keyframe, B/direct_ba.cc:618-622 -- the buffer contents are the same either way).
void DirectBA::TakeDeferredMergeCount() {
  uint32_t merged = 0;
  BAHIP_CHECKED_CALL(bahip_take_merged_count(ctx_, &merged));
  surfel_count_ -= merged;
}

// ---- single-frame pose estimation (B/direct_ba_alternating.cc:42-283) -----------------------------------------------
void DirectBA::EstimateFramePose(hipStream_t stream, const SE3f& global_T_frame_initial_estimate, const CUDABuffer<u16>& depth_buffer,
                                 const CUDABuffer<u16>& normals_buffer, hipTextureHandle_t color_texture,
                                 SE3f* out_global_T_frame_estimate, bool /*called_within_ba*/) {
  BAHIP_CHECKED_CALL(bahip_context_set_stream(ctx_, stream));
  const bahip_camera cc = ToBahipCamera(color_camera_), dc = ToBahipCamera(depth_camera_);
  const bahip_depth_params dp = ToBahipDepthParams(depth_params_);
  BAHIP_CHECKED_CALL(bahip_set_intrinsics(ctx_, &cc, &dc, &dp));
  bahip_frame frame{};
  frame.depth = depth_buffer.ToCUDA().address(); frame.depth_pitch_bytes = (uint32_t)depth_buffer.ToCUDA().pitch();
  frame.normals = normals_buffer.ToCUDA().address(); frame.normals_pitch_bytes = (uint32_t)normals_buffer.ToCUDA().pitch();
  frame.color = reinterpret_cast<uint8_t*>(color_texture->ToCUDA().address());
  frame.color_pitch_bytes = (uint32_t)color_texture->ToCUDA().pitch();
  const bahip_surfels s = SurfelsStruct();
  int iterations = 0, converged = 0;
  BAHIP_CHECKED_CALL(bahip_estimate_frame_pose(ctx_, use_depth_residuals_, use_descriptor_residuals_, &frame,
                                               global_T_frame_initial_estimate.data(), &s, out_global_T_frame_estimate->data(),
                                               &iterations, &converged));
  if (!converged) LOG(WARNING) << "Pose estimation not converged";
}

// ---- dispatcher (B/direct_ba.cc:407-454) ---------------------------------------------------------------------------------
void DirectBA::BundleAdjustment(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                                bool do_surfel_updates, bool optimize_poses, bool optimize_geometry, int min_iterations,
                                int max_iterations, bool use_pcg, int active_keyframe_window_start,
                                int active_keyframe_window_end, bool increase_ba_iteration_count, int* iterations_done,
                                bool* converged, double time_limit, Timer* timer, int pcg_max_inner_iterations,
                                int pcg_max_keyframes, std::function<bool(int)> progress_function) {
  if (optimize_depth_intrinsics && !use_depth_residuals_) {
    LOG(WARNING) << "optimize_depth_intrinsics set to true, but use_depth_residuals_ set to false. Depth intrinsics will not be optimized.";
    optimize_depth_intrinsics = false;
  }
  if (optimize_color_intrinsics && !use_descriptor_residuals_) {
    LOG(WARNING) << "optimize_color_intrinsics set to true, but use_descriptor_residuals_ set to false. Color intrinsics will not be optimized.";
    optimize_color_intrinsics = false;
  }
  last_pose_rounds_ = last_pose_steps_ = last_pcg_inner_steps_ = 0;
  if (keyframe_shard_world_ > 1)
    CHECK(!use_pcg && !optimize_depth_intrinsics && !optimize_color_intrinsics && !do_surfel_updates && !increase_ba_iteration_count)
        << "keyframe sharding covers the alternating scheme over poses and geometry without surfel updates and end tasks "
           "(their per-surfel sums run over all keyframes in order): use surfel sharding for the rest";
  if (use_pcg) {
    BundleAdjustmentPCG(stream, optimize_depth_intrinsics, optimize_color_intrinsics, do_surfel_updates, optimize_poses,
                        optimize_geometry, min_iterations, max_iterations, pcg_max_inner_iterations, pcg_max_keyframes,
                        active_keyframe_window_start, active_keyframe_window_end, increase_ba_iteration_count, iterations_done,
                        converged, time_limit, timer, progress_function);
  } else {
    BundleAdjustmentAlternating(stream, optimize_depth_intrinsics, optimize_color_intrinsics, do_surfel_updates, optimize_poses,
                                optimize_geometry, min_iterations, max_iterations, active_keyframe_window_start,
                                active_keyframe_window_end, increase_ba_iteration_count, iterations_done, converged, time_limit,
                                timer, progress_function);
  }
}

// ---- end-of-scheme tasks (B/direct_ba.cc:566-653) -----------------------------------------------------------------------
void DirectBA::PerformBASchemeEndTasks(hipStream_t stream, bool do_surfel_updates) {
  WholeCloudScope whole_cloud(this, stream);
  BindScene(stream);
  if (do_surfel_updates) {
    vector<u32> active_this_call;
    for (shared_ptr<Keyframe>& keyframe : keyframes_)
      if (keyframe && keyframe->last_active_in_ba_iteration() == ba_iteration_count_) active_this_call.push_back(keyframe->id());
    MergeForKeyframes(active_this_call);
  }
  const bahip_surfels s = SurfelsStruct();
  uint32_t deleted = 0;
  BAHIP_CHECKED_CALL(bahip_delete_surfels_and_update_radii(ctx_, GetMinObservationCount(), &s, &deleted));
  u32 surfel_count = surfel_count_ - deleted;
  const bahip_surfels s2 = SurfelsStruct(/*with_active*/ false);   // B/direct_ba.cc:619: no active-flag buffer
  unsorted_surfels_ += surfels_size_ - surfel_count;               // the holes compaction fills with surfels from the end
  BAHIP_CHECKED_CALL(bahip_compact_surfels(ctx_, surfel_count, &s2));
  Lock();
  surfels_size_ = surfel_count;
  surfel_count_ = surfel_count;
  Unlock();
  // Ours: back into Morton order if anything was appended or moved since the last reorder (direct_ba.h: SetSpatialSortCellSize;
  // the oracle's end tasks do the same, oracle_ba.c).  Under surfel sharding this runs on the gathered cloud, on every rank alike.
  if (spatial_sort_cell_size_ > 0.f) {
    if (unsorted_surfels_ > 0 && surfels_size_ > 1) SortSurfelsSpatially(stream, spatial_sort_cell_size_);
    unsorted_surfels_ = 0;
  }
}

// Ours: the compaction inside the loop fills the holes of merged surfels with surfels from the END of the buffer (the reference's rule,
// B/kernel_compact_surfels.cu:101-157) -- after a creation + merge pass over many keyframes nearly every 64-surfel tile of the Morton-
// ordered part then holds a surfel from somewhere else, its bounding sphere covers half the scene, and the sweeps of the call's remaining
// iterations cull nothing (measured: geometry 1.0 ms instead of 0.47 at 2.2 M surfels, profiles/r6_drop_in_trace_by_kernel.csv).  So once
// the surfels out of order amount to one per tile, the buffer goes back into Morton order here instead of at the end tasks only
// (same switch: SetSpatialSortCellSize(0) leaves the reference's order alone; the oracle's loop applies the same rule, oracle_ba.c).
void DirectBA::SortAfterInLoopCompaction(hipStream_t stream) {
  if (spatial_sort_cell_size_ > 0.f && surfels_size_ > 1 && (uint64_t)unsorted_surfels_ * 64 >= (uint64_t)surfels_size_)
    SortSurfelsSpatially(stream, spatial_sort_cell_size_);
}

void DirectBA::SortSurfelsSpatially(hipStream_t stream, float grid_cell_size) {
  BAHIP_CHECKED_CALL(bahip_context_set_stream(ctx_, stream));
  const bahip_surfels s = SurfelsStruct();
  BAHIP_CHECKED_CALL(bahip_sort_surfels_spatially(ctx_, &s, grid_cell_size));
  unsorted_surfels_ = 0;
}

// ---- alternating scheme (B/direct_ba_alternating.cc:285-738) ----------------------------------------------------------------
void DirectBA::BundleAdjustmentAlternating(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                                           bool do_surfel_updates, bool optimize_poses, bool optimize_geometry, int min_iterations,
                                           int max_iterations, int active_keyframe_window_start, int active_keyframe_window_end,
                                           bool increase_ba_iteration_count, int* num_iterations_done, bool* converged,
                                           double time_limit, Timer* timer, std::function<bool(int)> progress_function) {
  if (converged) *converged = false;
  if (num_iterations_done) *num_iterations_done = 0;
  Lock();
  const int fixed_ba_iteration_count = ba_iteration_count_;
  Unlock();
  if (!increase_ba_iteration_count && fixed_ba_iteration_count != last_ba_iteration_count_) {
    last_ba_iteration_count_ = fixed_ba_iteration_count;
    PerformBASchemeEndTasks(stream, do_surfel_updates);
  }
  vector<u32> keyframes_with_new_surfels;
  const bool fixed_active_keyframe_set = active_keyframe_window_start > 0 || active_keyframe_window_end > 0;
  const bool full_window = active_keyframe_window_start == 0 && active_keyframe_window_end == (int)keyframes_.size() - 1;
  if (!full_window)
    LOG(WARNING) << "Currently, only using all keyframes in every optimization iteration will work properly.";
  BAHIP_CHECKED_CALL(bahip_memset_async(stream, active_surfels_->ToCUDA().address(), 0, surfels_size_));

  // During the loop the device keyframe table is authoritative: the pose phase updates poses AND activations there
  // (bahip_estimate_keyframe_poses_and_update_activation, bahip_propagate_covisible_activation), so the next iteration's sweeps
  // can be queued at once, and the results are copied into the Keyframe objects ("pending") while the GPU already works on
  // them.  Anything that reads the Keyframe objects applies the pending results first.
  // Deferred host bookkeeping: operations on the Keyframe objects that mirror what the device table already went through,
  // run in order under the lock.
  vector<std::function<void()>> pending;
  auto apply_pending = [&]() {
    if (pending.empty()) return;
    Lock();
    for (auto& op : pending) op();
    Unlock();
    pending.clear();
  };
  auto set_window_activation = [this, active_keyframe_window_start, active_keyframe_window_end]() {
    for (u32 i = 0; i < keyframes_.size(); ++i) {
      if (!keyframes_[i]) continue;
      keyframes_[i]->SetActivation(((int)i >= active_keyframe_window_start && (int)i <= active_keyframe_window_end)
                                       ? Keyframe::Activation::kActive : Keyframe::Activation::kInactive);
    }
    DetermineCovisibleActiveKeyframes();
  };
  bool scene_bound = false;   // does the backend context hold the current keyframes / poses / activations / intrinsics?
  // BADSLAM_HOST_TIMING=1: wall time the host spends in each part of an iteration (printed once per call; diagnostics)
  const bool host_timing = getenv("BADSLAM_HOST_TIMING") != nullptr;
  double t_phase[6] = {0, 0, 0, 0, 0, 0};
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto lap = [&](int phase) { if (host_timing) { const double t = now(); t_phase[phase] += t - t_mark; t_mark = t; } };

  // ---- the loop driven by the device (bahip_alternating_iterations): poses + geometry over a fixed surfel set, nothing the
  // host has to look at between the iterations.  All iterations are queued at once; the stopping rule of :693-701 is evaluated
  // by the last solve launch of every pose phase, and the host waits once.  Everything else takes the loop below.
  bool device_loop_done = false;
  if (host_timing)
    fprintf(stderr, "[DirectBA] loop conditions: poses %d geometry %d updates %d intr %d/%d progress %d timer %d timings %d kfworld %d full_window %d max_it %d\n",
            (int)optimize_poses, (int)optimize_geometry, (int)do_surfel_updates, (int)optimize_depth_intrinsics, (int)optimize_color_intrinsics,
            progress_function ? 1 : 0, timer ? 1 : 0, timings_stream_ ? 1 : 0, keyframe_shard_world_, (int)full_window, max_iterations);
  if (optimize_poses && optimize_geometry && !do_surfel_updates && !optimize_depth_intrinsics && !optimize_color_intrinsics &&
      !progress_function && !timer && !timings_stream_ && keyframe_shard_world_ == 1 && full_window && max_iterations > 0) {
    if (fixed_active_keyframe_set) {
      Lock();
      set_window_activation();
      Unlock();
    }
    BindScene(stream);
    if (fixed_active_keyframe_set) {
      vector<uint8_t> in_window(bound_ids_.size());
      for (usize b = 0; b < bound_ids_.size(); ++b)
        in_window[b] = (bound_ids_[b] >= active_keyframe_window_start && bound_ids_[b] <= active_keyframe_window_end) ? 1 : 0;
      BAHIP_CHECKED_CALL(bahip_set_activation_window(ctx_, in_window.data(), (int)in_window.size()));
    }
    const int K = (int)bound_ids_.size();
    vector<float> poses(7 * (size_t)std::max(K, 1));
    vector<int> activation(std::max(K, 1));
    bahip_alternating_options options;
    options.use_depth_residuals = use_depth_residuals_; options.use_descriptor_residuals = use_descriptor_residuals_;
    options.fixed_window = fixed_active_keyframe_set ? 1 : 0;
    options.activate_in_geometry = 1;                        // full window: UpdateSurfelActivation inside the geometry sweep
    options.activation_surfels_size = surfels_size_;
    options.min_iterations = min_iterations; options.max_iterations = max_iterations;
    const bahip_surfels s = SurfelsStruct();
    int handled = 0, done = 0, conv = 0, rounds = 0, steps = 0, not_converged = 0;
    const double t_before_loop = host_timing ? now() : 0;
    BAHIP_CHECKED_CALL(bahip_alternating_iterations(ctx_, &options, &s, poses.data(), activation.data(), &handled, &done, &conv, &rounds, &steps,
                                                    &not_converged));
    const double t_after_loop = host_timing ? now() : 0;
    if (host_timing) fprintf(stderr, "[DirectBA] device loop %s\n", handled ? "handled the call" : "declined");
    if (handled) {
      device_loop_done = true;
      Lock();
      for (const shared_ptr<Keyframe>& keyframe : keyframes_) {
        if (!keyframe) continue;
        const int b = id_to_bound_[keyframe->id()];
        // Only a keyframe whose pose the loop changed gets it written back: a keyframe that stayed kInactive (or converged without
        // moving a bit) keeps the SE3f object it has -- set_global_T_frame rebuilds frame_T_global as the inverse, which is not
        // bit for bit what a caller may have set through set_frame_T_global (the host loop and the reference touch only the
        // keyframes that took a Gauss-Newton step).
        if (memcmp(keyframe->global_T_frame().data(), &poses[7 * (size_t)b], 7 * sizeof(float)) != 0)
          keyframe->set_global_T_frame(SE3f(&poses[7 * (size_t)b]));
        keyframe->SetActivation(static_cast<Keyframe::Activation>(activation[b]));
      }
      // the device table is in the state after the last pose phase; an iteration that did not end the loop is followed by
      // DetermineCovisibleActiveKeyframes (:703-709)
      if (!conv) DetermineCovisibleActiveKeyframes();
      Unlock();
      if (num_iterations_done) *num_iterations_done = done;
      if (converged) *converged = conv != 0;
      last_pose_rounds_ += rounds;
      last_pose_steps_ += steps;
      if (not_converged) LOG(WARNING) << "Pose estimation not converged (" << not_converged << " estimations)";
    }
    if (host_timing)
      fprintf(stderr, "[DirectBA, us] bind + window %.0f | bahip_alternating_iterations %.0f | write-back %.0f\n", t_before_loop - t_mark,
              t_after_loop - t_before_loop, now() - t_after_loop);
  }

  for (int iteration = 0; iteration < max_iterations && !device_loop_done; ++iteration) {
    if (progress_function) {
      apply_pending();   // the callback may look at the keyframes
      if (!progress_function(iteration)) break;
    }
    if (num_iterations_done) ++*num_iterations_done;
    if (fixed_active_keyframe_set) {
      if (scene_bound) {
        // the device table takes the window activation itself; the Keyframe objects follow when the pending work is applied
        BAHIP_CHECKED_CALL(bahip_apply_activation_window(ctx_));
        pending.push_back(set_window_activation);
      } else {
        apply_pending();
        Lock();
        set_window_activation();
        Unlock();
      }
    }

    // --- surfel creation ---
    keyframes_with_new_surfels.clear();
    CHECK_EQ(surfels_size_, surfel_count_);
    const usize old_surfels_size = surfels_size_;
    if (optimize_geometry && do_surfel_updates) {
      apply_pending();
      Lock();
      for (shared_ptr<Keyframe>& keyframe : keyframes_) {
        if (!keyframe) continue;
        if (keyframe->activation() == Keyframe::Activation::kActive && keyframe->last_active_in_ba_iteration() != fixed_ba_iteration_count) {
          keyframe->SetLastActiveInBAIteration(fixed_ba_iteration_count);
          keyframes_with_new_surfels.push_back(keyframe->id());
        } else if (keyframe->activation() == Keyframe::Activation::kCovisibleActive &&
                   keyframe->last_covis_in_ba_iteration() != fixed_ba_iteration_count) {
          keyframe->SetLastCovisInBAIteration(fixed_ba_iteration_count);
        }
      }
      Unlock();
      if (!keyframes_with_new_surfels.empty()) {
        WholeCloudScope whole_cloud(this, stream);   // one gather for the whole batch
        apply_pending();
        BindScene(stream);                           // ... and one binding
        creation_batch_bound_ = true;
        LifecycleBatch batch(this);
        CreateSurfelsForKeyframes(stream, /*filter_new_surfels*/ true, keyframes_with_new_surfels);
        creation_batch_bound_ = false;
      }
      if (!keyframes_with_new_surfels.empty()) scene_bound = false;   // CreateSurfelsForKeyframe re-bound the keyframes: lists and window go again
    }

    lap(0);
    if (!scene_bound) {
      apply_pending();
      BindScene(stream);
      if (fixed_active_keyframe_set) {
        vector<uint8_t> in_window(bound_ids_.size());
        for (usize b = 0; b < bound_ids_.size(); ++b)
          in_window[b] = (bound_ids_[b] >= active_keyframe_window_start && bound_ids_[b] <= active_keyframe_window_end) ? 1 : 0;
        BAHIP_CHECKED_CALL(bahip_set_activation_window(ctx_, in_window.data(), (int)in_window.size()));
      }
      scene_bound = true;
    }
    lap(1);

    // --- surfel activation + geometry ---
    if (optimize_geometry && surfels_size_ > old_surfels_size)
      BAHIP_CHECKED_CALL(bahip_memset_async(stream, active_surfels_->ToCUDA().address() + old_surfels_size, 1, surfels_size_ - old_surfels_size));
    if (!full_window) {
      BAHIP_CHECKED_CALL(bahip_memset_async(stream, active_surfels_->ToCUDA().address(), 1, old_surfels_size));
    } else if (!optimize_geometry) {
      const bahip_surfels s = SurfelsStruct();
      BAHIP_CHECKED_CALL(bahip_update_surfel_activation(ctx_, &s, (uint32_t)old_surfels_size));
    }
    if (optimize_geometry) {
      const bahip_surfels s = SurfelsStruct();
      if (full_window) {
        // UpdateSurfelActivationCUDA + OptimizeGeometryIterationCUDA (B/direct_ba_alternating.cc:441-487) as one sweep
        BAHIP_CHECKED_CALL(bahip_update_activation_and_optimize_geometry(ctx_, use_depth_residuals_, use_descriptor_residuals_, &s,
                                                                         (uint32_t)old_surfels_size));
      } else {
        BAHIP_CHECKED_CALL(bahip_optimize_geometry_iteration(ctx_, use_depth_residuals_, use_descriptor_residuals_, &s));
      }
    }
    lap(2);
    // the previous iteration's pose results go into the Keyframe objects while the GPU runs the sweep queued above
    apply_pending();
    lap(3);

    // --- surfel merge + compaction ---
    if (do_surfel_updates && !keyframes_with_new_surfels.empty()) {
      WholeCloudScope whole_cloud(this, stream);
      MergeForKeyframes(keyframes_with_new_surfels);
      {
        const bahip_surfels s = SurfelsStruct();
        unsorted_surfels_ += surfels_size_ - surfel_count_;
        BAHIP_CHECKED_CALL(bahip_compact_surfels(ctx_, surfel_count_, &s));
        Lock();
        surfels_size_ = surfel_count_;
        Unlock();
      }
      SortAfterInLoopCompaction(stream);
    }

    // --- poses: every non-inactive keyframe, batched per Gauss-Newton round; activations updated on the device ---
    usize num_converged = 0;
    if (optimize_poses) {
      const int K = (int)bound_ids_.size();
      auto poses = std::make_shared<vector<float>>(7 * (size_t)K, 0.f);
      auto moved = std::make_shared<vector<int>>(K, 0);
      vector<int> its(K), conv(K);
      int rounds = 0, converged_bound = 0;
      const bahip_surfels s = SurfelsStruct();
      BAHIP_CHECKED_CALL(bahip_estimate_keyframe_poses_and_update_activation(ctx_, use_depth_residuals_, use_descriptor_residuals_, &s,
                                                                             poses->data(), its.data(), conv.data(), moved->data(),
                                                                             &rounds, &converged_bound));
      pending.push_back([this, poses, moved]() {
        for (const shared_ptr<Keyframe>& keyframe : keyframes_) {
          if (!keyframe || keyframe->activation() == Keyframe::Activation::kInactive) continue;
          const int b = id_to_bound_[keyframe->id()];
          keyframe->set_global_T_frame(SE3f(&(*poses)[7 * (size_t)b]));
          keyframe->SetActivation((*moved)[b] ? Keyframe::Activation::kActive : Keyframe::Activation::kInactive);
        }
      });
      lap(4);
      last_pose_rounds_ += rounds;
      num_converged = (usize)converged_bound + (keyframes_.size() - (usize)K);   // deleted keyframes count as converged
      for (const shared_ptr<Keyframe>& keyframe : keyframes_) {
        if (!keyframe || keyframe->activation() == Keyframe::Activation::kInactive) continue;
        const int b = id_to_bound_[keyframe->id()];
        last_pose_steps_ += its[b];
        if (!conv[b]) LOG(WARNING) << "Pose estimation not converged (keyframe " << keyframe->id() << ")";
      }
    }

    // --- intrinsics ---
    if (optimize_depth_intrinsics || optimize_color_intrinsics) {
      bahip_camera out_color, out_depth;
      float out_a = depth_params_.a;
      const bahip_surfels s = SurfelsStruct();
      BAHIP_CHECKED_CALL(bahip_optimize_intrinsics(ctx_, optimize_depth_intrinsics, optimize_color_intrinsics, &s, &out_color, &out_depth, &out_a));
      // under surfel sharding every rank takes the globally solved cameras, also one whose own shard is empty (else the ranks
      // would bind different intrinsics from here on and disagree on convergence tests, i.e. on the number of collectives)
      if (surfels_size_ > 0 || bahip_context_is_sharded(ctx_)) {
        Lock();
        if (optimize_color_intrinsics) color_camera_ = PinholeCamera4f(out_color.width, out_color.height, &out_color.fx);
        if (optimize_depth_intrinsics) {
          depth_camera_ = PinholeCamera4f(out_depth.width, out_depth.height, &out_depth.fx);
          depth_params_.a = out_a;
        }
        Unlock();
        // the backend context takes over the new cameras / depth parameters (the keyframe table is unaffected)
        const bahip_camera cc = ToBahipCamera(color_camera_), dc = ToBahipCamera(depth_camera_);
        const bahip_depth_params dp = ToBahipDepthParams(depth_params_);
        BAHIP_CHECKED_CALL(bahip_set_intrinsics(ctx_, &cc, &dc, &dp));
      }
      if (intrinsics_updated_callback_) { apply_pending(); intrinsics_updated_callback_(); }
    }

    if (timings_stream_) {
      *timings_stream_ << "BA_count " << fixed_ba_iteration_count << " inner_iteration " << iteration << " keyframe_count "
                       << keyframes_.size() << " surfel_count " << surfel_count_ << std::endl;
      static const char* keys[4] = {"BA_surfel_activation", "BA_geometry_optimization", "BA_pose_optimization", nullptr};
      for (int stage = 0; stage < 3; ++stage) {
        float ms = 0; int launches = 0;
        if (bahip_last_stage_time_ms(ctx_, stage, &ms, &launches) == 0 && launches > 0) *timings_stream_ << keys[stage] << " " << ms << std::endl;
      }
    }

    // --- convergence ---
    if (iteration >= min_iterations - 1 && (num_converged == keyframes_.size() || !optimize_poses)) {
      if (converged) *converged = true;
      break;
    }
    if (timer && timer->GetTimeSinceStart() > time_limit) break;
    // DetermineCovisibleActiveKeyframes: on the device table now, on the Keyframe objects when the pending results are applied
    // (with a fixed window the next iteration overwrites every activation on the device first thing, so only the Keyframe objects
    // need this step there)
    if (!fixed_active_keyframe_set) BAHIP_CHECKED_CALL(bahip_propagate_covisible_activation(ctx_));
    pending.push_back([this]() { DetermineCovisibleActiveKeyframes(); });
  }
  apply_pending();
  lap(5);
  if (host_timing)
    fprintf(stderr, "[DirectBA host timing, us] loop top %.0f | bind %.0f | geometry launch %.0f | apply pending %.0f | pose phase %.0f | rest %.0f\n",
            t_phase[0], t_phase[1], t_phase[2], t_phase[3], t_phase[4], t_phase[5]);

  if (increase_ba_iteration_count) {
    PerformBASchemeEndTasks(stream, do_surfel_updates);
    ++ba_iteration_count_;
  }
  BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
}

// ---- PCG scheme (B/direct_ba_pcg.cc:43-819) -----------------------------------------------------------------------------------
void DirectBA::BundleAdjustmentPCG(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                                   bool do_surfel_updates, bool optimize_poses, bool optimize_geometry, int min_iterations,
                                   int max_iterations, int max_inner_iterations, int max_keyframe_count,
                                   int active_keyframe_window_start, int active_keyframe_window_end, bool increase_ba_iteration_count,
                                   int* num_iterations_done, bool* converged, double time_limit, Timer* timer,
                                   std::function<bool(int)> progress_function) {
  if ((active_keyframe_window_start != -1 || active_keyframe_window_end != -1) &&
      (active_keyframe_window_start != 0 || active_keyframe_window_end != (int)keyframes_.size() - 1))
    LOG(WARNING) << "The PCG-based solver implementation does not support an active window! These parameters will be ignored.";
  if (num_iterations_done) *num_iterations_done = 0;
  if (converged) *converged = false;
  for (auto& keyframe : keyframes_) {
    if (!keyframe) {
      LOG(ERROR) << "The PCG-based solver implementation does not support having deleted keyframes yet! Aborting.";
      return;
    }
  }
  CHECK_LE((int)keyframes_.size(), max_keyframe_count);
  if (!increase_ba_iteration_count && ba_iteration_count_ != last_ba_iteration_count_) {
    last_ba_iteration_count_ = ba_iteration_count_;
    PerformBASchemeEndTasks(stream, do_surfel_updates);
  }
  vector<u32> keyframes_with_new_surfels;
  auto merge_and_compact = [&]() {
    if (keyframes_with_new_surfels.empty()) return;
    WholeCloudScope whole_cloud(this, stream);
    MergeForKeyframes(keyframes_with_new_surfels);
    if (!keyframes_with_new_surfels.empty()) {
      const bahip_surfels s = SurfelsStruct();
      unsorted_surfels_ += surfels_size_ - surfel_count_;
      BAHIP_CHECKED_CALL(bahip_compact_surfels(ctx_, surfel_count_, &s));
      surfels_size_ = surfel_count_;
      SortAfterInLoopCompaction(stream);
    }
  };

  for (int iteration = 0; iteration < max_iterations; ++iteration) {
    if (progress_function && !progress_function(iteration)) break;
    if (num_iterations_done) ++*num_iterations_done;
    keyframes_with_new_surfels.clear();
    if (optimize_geometry && do_surfel_updates) {
      bool any_new = false;
      for (const shared_ptr<Keyframe>& keyframe : keyframes_)
        any_new |= keyframe->activation() == Keyframe::Activation::kActive && keyframe->last_active_in_ba_iteration() != ba_iteration_count_;
      std::unique_ptr<WholeCloudScope> whole_cloud(any_new ? new WholeCloudScope(this, stream) : nullptr);   // one gather for the batch
      if (any_new) BindScene(stream);                                                                          // ... and one binding
      creation_batch_bound_ = any_new;
      std::unique_ptr<LifecycleBatch> batch(any_new ? new LifecycleBatch(this) : nullptr);
      for (shared_ptr<Keyframe>& keyframe : keyframes_) {
        if (keyframe->activation() == Keyframe::Activation::kActive && keyframe->last_active_in_ba_iteration() != ba_iteration_count_) {
          keyframe->SetLastActiveInBAIteration(ba_iteration_count_);
          keyframes_with_new_surfels.push_back(keyframe->id());
        } else if (keyframe->activation() == Keyframe::Activation::kCovisibleActive &&
                   keyframe->last_covis_in_ba_iteration() != ba_iteration_count_) {
          keyframe->SetLastCovisInBAIteration(ba_iteration_count_);
        }
      }
      CreateSurfelsForKeyframes(stream, /*filter_new_surfels*/ true, keyframes_with_new_surfels);   // in keyframe order, as the loop above met them
      creation_batch_bound_ = false;
    }
    BindScene(stream);
    BAHIP_CHECKED_CALL(bahip_memset_async(stream, active_surfels_->ToCUDA().address(), 1, surfels_size_));
    if (optimize_geometry) {
      const bahip_surfels s = SurfelsStruct();
      BAHIP_CHECKED_CALL(bahip_update_surfel_normals(ctx_, &s));
    }

    bahip_pcg_options opt;
    opt.optimize_poses = optimize_poses; opt.optimize_geometry = optimize_geometry;
    opt.optimize_depth_intrinsics = optimize_depth_intrinsics; opt.optimize_color_intrinsics = optimize_color_intrinsics;
    opt.use_depth_residuals = use_depth_residuals_; opt.use_descriptor_residuals = use_descriptor_residuals_;
    opt.max_inner_iterations = max_inner_iterations;
    opt.gauge_keyframe = (pcg_gauge_keyframe_ >= 0) ? pcg_gauge_keyframe_ : (rand() % (int)keyframes_.size());   // B/direct_ba_pcg.cc:328
    bahip_camera out_color, out_depth;
    float out_a = depth_params_.a;
    int inner_steps = 0, num_converged = 0;
    const bahip_surfels s = SurfelsStruct();
    BAHIP_CHECKED_CALL(bahip_pcg_iteration(ctx_, &opt, &s, &out_color, &out_depth, &out_a, &inner_steps, &num_converged));
    last_pcg_inner_steps_ += inner_steps;
    if (optimize_poses) {
      vector<float> poses(7 * keyframes_.size());
      BAHIP_CHECKED_CALL(bahip_get_keyframe_poses(ctx_, poses.data(), (int)keyframes_.size()));
      for (usize k = 0; k < keyframes_.size(); ++k) keyframes_[k]->set_global_T_frame(SE3f(&poses[7 * k]));
    }
    if (optimize_color_intrinsics) color_camera_ = PinholeCamera4f(out_color.width, out_color.height, &out_color.fx);
    if (optimize_depth_intrinsics) {
      depth_camera_ = PinholeCamera4f(out_depth.width, out_depth.height, &out_depth.fx);
      depth_params_.a = out_a;
    }
    if ((optimize_depth_intrinsics || optimize_color_intrinsics) && intrinsics_updated_callback_) intrinsics_updated_callback_();
    if (do_surfel_updates) {
      BindScene(stream);   // merges use the updated poses / intrinsics
      merge_and_compact();
    }

    if (iteration >= min_iterations - 1 && ((usize)num_converged == keyframes_.size() || !optimize_poses)) {
      if (converged) *converged = true;
      break;
    }
    if (timer && timer->GetTimeSinceStart() > time_limit) break;
  }

  if (increase_ba_iteration_count) {
    PerformBASchemeEndTasks(stream, do_surfel_updates);
    ++ba_iteration_count_;
  } else if (do_surfel_updates) {
    BindScene(stream);
    merge_and_compact();   // B/direct_ba_pcg.cc:775-812
  }
  BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
}

}  // namespace vis
