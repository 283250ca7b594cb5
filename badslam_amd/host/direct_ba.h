// direct_ba.h -- vis::DirectBA, the direct bundle-adjustment back-end of BAD SLAM, with the public
// surface of the reference class (B/direct_ba.h:65-550; B/ = applications/badslam/src/badslam/):
// same constructor arguments, same method names, argument meaning and defaults, same accessors.
// Everything below the surface is new: the scene lives in HIP buffers, and every stage is one call
// into the C ABI of the MI355X backend (include/badslam_hip.h) instead of a loop of per-keyframe
// kernel launches.  Differences forced by the platform:
//   - cudaStream_t -> hipStream_t (opaque), cudaTextureObject_t -> hipTextureHandle_t (gfx950 has
//     no texture sampling; the handle names the colour buffer),
//   - render window / OpenGL interop and loop-detector hooks are not part of the BA path
//     (OUT) and are omitted,
//   - SetRunParallel is declared but never defined in the reference (B/direct_ba.h:172): omitted.
#pragma once

#include "camera_frustum.h"
#include "keyframe.h"

namespace vis {

constexpr int kMergeBufferCount = BAHIP_MERGE_BUFFER_COUNT;       // B/kernels.cuh:51
constexpr int kSurfelAttributeCount = BAHIP_SURFEL_ATTRIBUTE_COUNT;
constexpr int kSurfelX = 0, kSurfelY = 1, kSurfelZ = 2, kSurfelNormal = 3, kSurfelRadiusSquared = 4, kSurfelColor = 5,
              kSurfelDescriptor1 = 6, kSurfelDescriptor2 = 7, kSurfelAccum0 = 8;   // B/kernels.cuh:69-88

struct Point3fC3u8Nf;   // rgbd_io.h (L/point_cloud.h: the element of Point3fC3u8NfCloud)
class DirectBA {
 public:
  DirectBA(int max_surfel_count, float raw_to_float_depth, float baseline_fx, int sparse_surfel_cell_size,
           float surfel_merge_dist_factor, int min_observation_count_while_bootstrapping_1,
           int min_observation_count_while_bootstrapping_2, int min_observation_count,
           const PinholeCamera4f& color_camera_initial_estimate, const PinholeCamera4f& depth_camera_initial_estimate,
           int pyramid_level_for_color, bool use_depth_residuals, bool use_descriptor_residuals,
           void* render_window /* must be nullptr */, const SE3f& global_T_anchor_frame);
  ~DirectBA();

  void AddKeyframe(const shared_ptr<Keyframe>& new_keyframe);
  void DeleteKeyframe(int keyframe_index, void* loop_detector = nullptr);
  void MergeKeyframes(hipStream_t stream, void* loop_detector, usize approx_merge_count = 10);
  // B/direct_ba.h:175, B/direct_ba.cc:461-547: the valid surfels as a point cloud (position, colour, normal), in index order
  void ExportToPointCloud(hipStream_t stream, vector<Point3fC3u8Nf>* cloud);
  void CreateSurfelsForKeyframe(hipStream_t stream, bool filter_new_surfels, const shared_ptr<Keyframe>& keyframe);
  void EstimateFramePose(hipStream_t stream, const SE3f& global_T_frame_initial_estimate, const CUDABuffer<u16>& depth_buffer,
                         const CUDABuffer<u16>& normals_buffer, hipTextureHandle_t color_texture,
                         SE3f* out_global_T_frame_estimate, bool called_within_ba);
  void BundleAdjustment(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool do_surfel_updates,
                        bool optimize_poses, bool optimize_geometry, int min_iterations, int max_iterations, bool use_pcg,
                        int active_keyframe_window_start, int active_keyframe_window_end, bool increase_ba_iteration_count,
                        int* iterations_done = nullptr, bool* converged = nullptr, double time_limit = 0, Timer* timer = nullptr,
                        int pcg_max_inner_iterations = 30, int pcg_max_keyframes = 2500,
                        std::function<bool(int)> progress_function = nullptr);
  void UpdateKeyframeCoVisibility(const shared_ptr<Keyframe>& keyframe);
  // B/direct_ba.h:167, B/direct_ba.cc:456-459: surfel colours := mean of their observations in all keyframes (before an export).
  void AssignColors(hipStream_t stream);

  void Lock() const { ba_thread_mutex_.lock(); }
  void Unlock() const { ba_thread_mutex_.unlock(); }
  std::mutex& Mutex() const { return ba_thread_mutex_; }

  int GetMinObservationCount() const {
    return (keyframes_.size() < 10) ? ((keyframes_.size() < 5) ? min_observation_count_while_bootstrapping_1_
                                                                : min_observation_count_while_bootstrapping_2_)
                                    : min_observation_count_;
  }

  // --- accessors (B/direct_ba.h:226-388) ---
  const vector<shared_ptr<Keyframe>>& keyframes() const { return keyframes_; }
  vector<shared_ptr<Keyframe>>* keyframes_mutable() { return &keyframes_; }
  PinholeCamera4f color_camera() const { lock_guard<mutex> lock(ba_thread_mutex_); return color_camera_; }
  PinholeCamera4f color_camera_no_lock() const { return color_camera_; }
  void SetColorCamera(const PinholeCamera4f& camera) { lock_guard<mutex> lock(ba_thread_mutex_); color_camera_ = camera; }
  int pyramid_level_for_color() const { return pyramid_level_for_color_; }
  void SetPyramidLevelForColor(int level) { pyramid_level_for_color_ = level; }
  PinholeCamera4f depth_camera() const { lock_guard<mutex> lock(ba_thread_mutex_); return depth_camera_; }
  PinholeCamera4f depth_camera_no_lock() const { return depth_camera_; }
  void SetDepthCamera(const PinholeCamera4f& camera) { lock_guard<mutex> lock(ba_thread_mutex_); depth_camera_ = camera; }
  DepthParameters depth_params() const { lock_guard<mutex> lock(ba_thread_mutex_); return depth_params_; }
  DepthParameters depth_params_no_lock() const { return depth_params_; }
  void SetDepthParams(const DepthParameters& params) { lock_guard<mutex> lock(ba_thread_mutex_); depth_params_ = params; }
  float& a() { return depth_params_.a; }
  float a() const { return depth_params_.a; }
  CUDABufferPtr<float> cfactor_buffer() { return cfactor_buffer_; }
  CUDABufferConstPtr<float> cfactor_buffer() const { return cfactor_buffer_; }
  void SetCFactorBuffer(const CUDABufferPtr<float>& cfactor_buffer) {
    cfactor_buffer_ = cfactor_buffer;
    depth_params_.cfactor_buffer = cfactor_buffer_->ToCUDA();
  }
  void IncreaseBAIterationCount() { lock_guard<mutex> lock(ba_thread_mutex_); ++ba_iteration_count_; }
  bool use_depth_residuals() const { return use_depth_residuals_; }
  void SetUseDepthResiduals(bool v) { use_depth_residuals_ = v; }
  bool use_descriptor_residuals() const { return use_descriptor_residuals_; }
  void SetUseDescriptorResiduals(bool v) { use_descriptor_residuals_ = v; }
  int sparse_surfel_cell_size() const { return depth_params_.sparse_surfel_cell_size; }
  void SetSparsificationSideFactor(int cell) { depth_params_.sparse_surfel_cell_size = cell; }
  int min_observation_count_while_bootstrapping_1() const { return min_observation_count_while_bootstrapping_1_; }
  void SetMinObservationCountWhileBootstrapping1(int c) { min_observation_count_while_bootstrapping_1_ = c; }
  int min_observation_count_while_bootstrapping_2() const { return min_observation_count_while_bootstrapping_2_; }
  void SetMinObservationCountWhileBootstrapping2(int c) { min_observation_count_while_bootstrapping_2_ = c; }
  int min_observation_count() const { return min_observation_count_; }
  void SetMinObservationCount(int c) { min_observation_count_ = c; }
  void SetIntrinsicsUpdatedCallback(const std::function<void()>& callback) { intrinsics_updated_callback_ = callback; }
  u32 surfel_count() const { lock_guard<mutex> lock(ba_thread_mutex_); return surfel_count_; }
  u32 surfels_size() const { lock_guard<mutex> lock(ba_thread_mutex_); return surfels_size_; }
  void SetSurfelCount(u32 surfel_count, u32 surfels_size) {
    surfel_count_ = surfel_count; surfels_size_ = surfels_size;
    if (unsorted_surfels_ > surfels_size) unsorted_surfels_ = surfels_size;   // (an emptied buffer has nothing out of order)
  }
  // Not in the reference.  Reorders the surfel buffer along a Morton curve over a world grid (bahip_sort_surfels_spatially):
  // surfels that an image region shows become neighbours in the buffer, which the sweeps' cache behaviour wants.  No
  // result depends on the order; call it when the surfel set has grown (after adding keyframes), not per iteration.
  void SortSurfelsSpatially(hipStream_t stream, float grid_cell_size = 0.02f);
  // Spatial order as part of the reference's own call path (round 4): PerformBASchemeEndTasks -- which compacts, i.e. moves
  // surfels, anyway (B/direct_ba.cc:619-640) -- puts the buffer back into Morton order whenever surfels were appended or moved
  // since the last reorder, so a caller that knows only B/direct_ba.h:73-388 gets the buffer the sweeps are fast on.
  // cell size 0 switches it off (the reference's surfel order stays observable); default 0.02 m.
  // Round 6: the compaction INSIDE the loop (after a merge pass) is followed by the same reorder once the surfels out of order amount
  // to one per 64-surfel tile, so that the call's remaining iterations sweep a coherent buffer (direct_ba.cc: SortAfterInLoopCompaction).
  void SetSpatialSortCellSize(float grid_cell_size) { spatial_sort_cell_size_ = grid_cell_size; }
  void SortAfterInLoopCompaction(hipStream_t stream);
  // Ours: the creations of a BA iteration and the merges of a merge pass as ONE call of the backend each (default:
  // bahip_create_surfels_for_keyframes, bahip_merge_surfels_for_keyframes) or keyframe by keyframe with the host in between (the
  // reference's shape).  Same surfels either way.
  void SetBatchedCreation(bool enabled) { batched_creation_ = enabled; }
  float spatial_sort_cell_size() const { return spatial_sort_cell_size_; }
  u32 unsorted_surfels() const { return unsorted_surfels_; }
  CUDABufferConstPtr<float> surfels() const { return surfels_; }
  CUDABufferPtr<float> surfels() { return surfels_; }
  CUDABufferPtr<u8> active_surfels() { return active_surfels_; }
  int ba_iteration_count() const { return ba_iteration_count_; }
  void SetBAIterationCount(int count) { ba_iteration_count_ = count; }
  int last_ba_iteration_count() const { return last_ba_iteration_count_; }
  void SetLastBAIterationCount(int count) { last_ba_iteration_count_ = count; }
  float surfel_merge_dist_factor() const { return surfel_merge_dist_factor_; }
  void SetSurfelMergeDistFactor(float factor) { surfel_merge_dist_factor_ = factor; }
  void SetSaveTimings(std::ostream* stream) { timings_stream_ = stream; }

  // --- additions of this backend ---
  // The gauge keyframe of the PCG scheme; the reference draws rand() % K per outer iteration
  // (B/direct_ba_pcg.cc:328).  < 0 (default) = same rand() stream.
  void SetPCGGaugeKeyframe(int keyframe_id) { pcg_gauge_keyframe_ = keyframe_id; }
  // Multi-GPU surfel sharding: sums of the per-keyframe normal equations go through this hook
  // (see include/badslam_hip.h, bahip_allreduce_fn).
  void SetAllReduce(bahip_allreduce_fn fn, void* user) { BAHIP_CHECKED_CALL(bahip_context_set_allreduce(ctx_, fn, user)); }
  // Multi-GPU surfel sharding WITH surfel updates: this object holds rank `rank`'s chunk-cyclic shard of one surfel cloud
  // (chunks of `chunk` surfels, a multiple of 64; see bahip_gather_surfel_shards).  The sweeps of an iteration work on the
  // shard; every phase of the surfel lifecycle (creation, merging, deletion, compaction) assembles the whole cloud on every
  // rank, runs unchanged, and takes the shard back out -- so a sharded BundleAdjustment with do_surfel_updates ends with the
  // bits of the unsharded one.  Needs SetAllReduce (or an RCCL communicator on the backend context) when world > 1.
  void SetSurfelSharding(int rank, int world, u32 chunk);
  // Multi-GPU KEYFRAME sharding (bahip_context_set_keyframe_sharding): this object holds ALL surfels; of the keyframes it needs
  // the images of those with (index among the non-deleted keyframes) % world == rank only (world = 1, 2, 4, or 8 after
  // SetSumClasses(8)).  Covers the
  // alternating scheme over poses and geometry -- BundleAdjustment(stream, false, false, /*do_surfel_updates*/ false, ...,
  // /*use_pcg*/ false, ..., /*increase_ba_iteration_count*/ false) -- and ends with the unsharded run's bits on every rank; the
  // intrinsics step, the PCG scheme and the surfel lifecycle (end tasks included) are refused.  Needs SetAllReduce or an RCCL
  // communicator when world > 1.
  void SetKeyframeSharding(int rank, int world);
  // The per-surfel sums of the normals / geometry passes are defined over 4 (default) or 8 interleaved keyframe classes
  // (bahip_context_set_sum_classes); keyframe sharding over 8 ranks needs 8 -- and so does the single-GPU run it is compared with.
  void SetSumClasses(int classes);
  // Ours: new surfels of a keyframe in the reference's row-major append order (B/kernel_create_surfels.cu:357-390) instead of this
  // backend's tile-major one (bahip_context_set_creation_order): the same surfels, the reference's indices -- and therefore the
  // reference's survivors when surfels merge.  Slower sweeps until the next spatial reorder (SetSpatialSortCellSize).
  void SetRowMajorCreation(bool enabled);
  // Ours: the arithmetic flavour of the sweeps (bahip_context_set_arithmetic).  false (default): every bit is the CPU oracle's; true:
  // hardware reciprocal / square root / exp, contraction, flushed denormals -- what the reference's own -use_fast_math build computes
  // with -- a few percent faster, same determinism and shard invariance, results within the reference's own tolerance.
  void SetFastArithmetic(bool enabled);
  bahip_context* backend_context() { return ctx_; }
  // Binds intrinsics + all non-null keyframes to the backend context; fills index maps between
  // keyframe ids and the dense bound list.  Public so that a caller can drive single bahip_* stages
  // on this scene (the stage-level parity tests do).
  void BindScene(hipStream_t stream);
  bahip_surfels SurfelsStruct(bool with_active = true) const;
  // Statistics of the last BundleAdjustment() call: Gauss-Newton rounds (batched over keyframes)
  // and total per-keyframe GN steps, PCG inner steps.
  int last_pose_rounds() const { return last_pose_rounds_; }
  int last_pose_steps() const { return last_pose_steps_; }
  int last_pcg_inner_steps() const { return last_pcg_inner_steps_; }

 private:
  void BundleAdjustmentAlternating(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                                   bool do_surfel_updates, bool optimize_poses, bool optimize_geometry, int min_iterations,
                                   int max_iterations, int active_keyframe_window_start, int active_keyframe_window_end,
                                   bool increase_ba_iteration_count, int* num_iterations_done, bool* converged, double time_limit,
                                   Timer* timer, std::function<bool(int)> progress_function);
  void BundleAdjustmentPCG(hipStream_t stream, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool do_surfel_updates,
                           bool optimize_poses, bool optimize_geometry, int min_iterations, int max_iterations,
                           int max_inner_iterations, int max_keyframe_count, int active_keyframe_window_start,
                           int active_keyframe_window_end, bool increase_ba_iteration_count, int* num_iterations_done,
                           bool* converged, double time_limit, Timer* timer, std::function<bool(int)> progress_function);
  void DetermineCovisibleActiveKeyframes();
  void DetermineNewKeyframeCoVisibility(const shared_ptr<Keyframe>& new_keyframe);
  void PerformBASchemeEndTasks(hipStream_t stream, bool do_surfel_updates);

  void MergeForKeyframe(const Keyframe& keyframe, bool defer_count = false);
  void TakeDeferredMergeCount();
  void MergeForKeyframes(const vector<u32>& keyframe_ids);
  void CreateSurfelsForKeyframes(hipStream_t stream, bool filter_new_surfels, const vector<u32>& keyframe_ids);
  bool batched_creation_ = true;
  class LifecycleBatch {   // RAII: bahip_lifecycle_batch_begin / _end around the creations or merges of a batch of keyframes
   public:
    explicit LifecycleBatch(DirectBA* ba);
    ~LifecycleBatch();
    LifecycleBatch(const LifecycleBatch&) = delete;
    LifecycleBatch& operator=(const LifecycleBatch&) = delete;
   private:
    DirectBA* ba_;
  };
  bool creation_batch_bound_ = false;
  // Whole-cloud phases under surfel sharding (no-ops without it); they nest, only the outermost pair moves data.
  void EnterWholeCloud(hipStream_t stream);
  void LeaveWholeCloud(hipStream_t stream);
  struct WholeCloudScope {
    WholeCloudScope(DirectBA* ba, hipStream_t stream) : ba_(ba), stream_(stream) { ba_->EnterWholeCloud(stream_); }
    ~WholeCloudScope() { ba_->LeaveWholeCloud(stream_); }
    DirectBA* ba_; hipStream_t stream_;
  };

  PinholeCamera4f color_camera_;
  int pyramid_level_for_color_;
  PinholeCamera4f depth_camera_;
  CUDABufferPtr<float> cfactor_buffer_;
  DepthParameters depth_params_;
  vector<shared_ptr<Keyframe>> keyframes_;
  u32 surfel_count_ = 0, surfels_size_ = 0;
  u32 unsorted_surfels_ = 0;            // appended, or moved by a compaction, since the buffer was last in Morton order
  float spatial_sort_cell_size_ = 0.02f;
  CUDABufferPtr<float> surfels_;
  CUDABufferPtr<u8> active_surfels_;
  int ba_iteration_count_ = 0, last_ba_iteration_count_ = -1;
  bool use_depth_residuals_, use_descriptor_residuals_;
  int min_observation_count_while_bootstrapping_1_, min_observation_count_while_bootstrapping_2_, min_observation_count_;
  float surfel_merge_dist_factor_;
  CUDABufferPtr<u32> supporting_surfels_[kMergeBufferCount];
  mutable mutex ba_thread_mutex_;
  std::ostream* timings_stream_ = nullptr;
  std::function<void()> intrinsics_updated_callback_;
  SE3f global_T_anchor_frame_;

  bahip_context* ctx_ = nullptr;
  vector<int> bound_ids_;        // bound list index -> keyframe id
  vector<int> id_to_bound_;      // keyframe id -> bound list index (-1 for deleted keyframes)
  int pcg_gauge_keyframe_ = -1;
  int shard_rank_ = 0, shard_world_ = 1, whole_cloud_depth_ = 0;
  int keyframe_shard_world_ = 1;
  u32 shard_chunk_ = 0;
  CUDABufferPtr<float> other_surfels_;      // under surfel sharding: the buffer not in use (whole cloud <-> shard)
  CUDABufferPtr<u8> other_active_surfels_;
  int last_pose_rounds_ = 0, last_pose_steps_ = 0, last_pcg_inner_steps_ = 0;
};

}  // namespace vis
