// ba_scheduler.cc -- see ba_scheduler.h.  Host-only code: every GPU operation goes through DirectBA / the bahip_* C ABI.
#include "ba_scheduler.h"

#include <algorithm>
#include <functional>

namespace vis {

BAScheduler::BAScheduler(const BASchedulerConfig& config, DirectBA* direct_ba, RGBDVideo<Vec3u8, u16>* rgbd_video, hipStream_t stream)
    : config_(config), direct_ba_(direct_ba), rgbd_video_(rgbd_video), stream_(stream) {
  if (config_.parallel_ba) RestartBAThread();   // B/bad_slam.cc:173-176
}

BAScheduler::~BAScheduler() {
  if (config_.parallel_ba) StopBAThreadAndWaitForIt();   // B/bad_slam.cc:286-288
}

void BAScheduler::SetLastFrameIndex(int frame_index) {
  lock_guard<mutex> lock(direct_ba_->Mutex());
  last_frame_index_ = frame_index;
}

SE3f BAScheduler::base_kf_global_T_frame() const {
  lock_guard<mutex> lock(direct_ba_->Mutex());
  return base_kf_global_T_frame_;
}

void BAScheduler::GetQueuedKeyframes(vector<shared_ptr<Keyframe>>* queued_keyframes, vector<SE3f>* queued_keyframes_last_kf_tr_this_kf) const {
  lock_guard<mutex> lock(direct_ba_->Mutex());
  *queued_keyframes = queued_keyframes_;
  *queued_keyframes_last_kf_tr_this_kf = queued_keyframes_last_kf_tr_this_kf_;
}

void BAScheduler::PublishKeyframePosesNoLock() {
  for (const shared_ptr<Keyframe>& keyframe : direct_ba_->keyframes()) {
    if (!keyframe || keyframe->frame_index() >= rgbd_video_->frame_count()) continue;
    rgbd_video_->depth_frame_mutable(keyframe->frame_index())->SetGlobalTFrame(keyframe->global_T_frame());
    rgbd_video_->color_frame_mutable(keyframe->frame_index())->SetGlobalTFrame(keyframe->global_T_frame());
  }
}

// B/bad_slam.cc:1126-1162 without the loop detector (loop closure is outside the BA path).
void BAScheduler::AddKeyframeToBA(hipStream_t /*stream*/, const shared_ptr<Keyframe>& new_keyframe) {
  direct_ba_->Lock();
  direct_ba_->AddKeyframe(new_keyframe);
  direct_ba_->Unlock();
}

void BAScheduler::AddKeyframe(const shared_ptr<Keyframe>& new_keyframe, const SE3f& last_kf_tr_this_kf) {
  direct_ba_->Lock();
  base_kf_ = new_keyframe.get();
  // The front-end keeps working with this "pre-BA" pose while a BA iteration is under way (B/bad_slam.cc:1003-1016).
  base_kf_global_T_frame_ = base_kf_->global_T_frame();
  last_frame_index_ = std::max<int>(last_frame_index_, (int)new_keyframe->frame_index());
  const bool nothing_yet = direct_ba_->keyframes().empty() && queued_keyframes_.empty();
  direct_ba_->Unlock();

  int keyframes_added;
  // The very first keyframe is added directly in both modes: its surfels are created right below, by this thread, and
  // DirectBA::CreateSurfelsForKeyframe addresses the keyframe through the bound keyframe list.  (The reference queues
  // it and creates surfels from the still-queued keyframe; the BA thread has no work at that point either way.)
  if (config_.parallel_ba && !nothing_yet) {
    // B/bad_slam.cc:1030-1050.  The reference records an event on the odometry stream for the BA stream to wait on;
    // the C ABI has no event object, so the odometry stream is drained here instead (the keyframe's buffers were
    // filled on it).
    BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream_));
    direct_ba_->Lock();
    queued_keyframes_.push_back(new_keyframe);
    queued_keyframes_last_kf_tr_this_kf_.push_back(last_kf_tr_this_kf);
    keyframes_added = (int)(queued_keyframes_.size() + direct_ba_->keyframes().size());
    direct_ba_->Unlock();
  } else {
    AddKeyframeToBA(stream_, new_keyframe);   // :1051-1055
    keyframes_added = (int)direct_ba_->keyframes().size();
  }

  if (!config_.estimate_poses) return;   // :1072-1075

  // B/bad_slam.cc:1078-1099
  if (keyframes_added >= 2) {
    // Without surfel updates inside BA, new keyframes get their surfels here (sequential mode) or from the BA thread
    // right after it has added the keyframe (parallel mode; the reference does it from this thread).
    if (!config_.do_surfel_updates && !config_.parallel_ba) direct_ba_->CreateSurfelsForKeyframe(stream_, true, new_keyframe);
    num_planned_ba_iterations_ += config_.max_num_ba_iterations_per_keyframe;
    // Trigger surfel updates within the next BA iteration.
    if (config_.parallel_ba) direct_ba_->IncreaseBAIterationCount();
  } else {
    direct_ba_->CreateSurfelsForKeyframe(stream_, false, new_keyframe);
    BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream_));
  }
}

// B/bad_slam.cc:214-281 (offline mode: no frame-time budget)
void BAScheduler::RunPlannedIterations(u32 frame_index) {
  if (num_planned_ba_iterations_ <= 0) return;
  ++bundle_adjustment_counter_;
  direct_ba_->Lock();
  const usize keyframes_size = direct_ba_->keyframes().size() + queued_keyframes_.size();
  last_frame_index_ = std::max<int>(last_frame_index_, (int)frame_index);
  direct_ba_->Unlock();

  // :230-245: intrinsics are optimised often while there are few keyframes, then every n-th call
  const bool optimize_depth_intrinsics =
      config_.optimize_intrinsics &&
      (keyframes_size >= 10 && (keyframes_size <= 20 || (bundle_adjustment_counter_ % config_.intrinsics_optimization_interval == 0)));
  const bool optimize_color_intrinsics = optimize_depth_intrinsics;

  if (config_.parallel_ba) {
    StartParallelIterations(num_planned_ba_iterations_, optimize_depth_intrinsics, optimize_color_intrinsics, config_.do_surfel_updates,
                            /*optimize_poses*/ true, /*optimize_geometry*/ true);
    num_planned_ba_iterations_ = 0;
  } else {
    int iterations_done = 0;
    bool converged = false;
    RunBundleAdjustment(frame_index, optimize_depth_intrinsics && config_.use_geometric_residuals,
                        optimize_color_intrinsics && config_.use_photometric_residuals, /*optimize_poses*/ true, /*optimize_geometry*/ true,
                        /*min_iterations*/ 0, num_planned_ba_iterations_,
                        /*active_keyframe_window_start*/ config_.disable_deactivation ? 0 : -1,
                        /*active_keyframe_window_end*/ config_.disable_deactivation ? ((int)direct_ba_->keyframes().size() - 1) : -1,
                        /*increase_ba_iteration_count*/ true, &iterations_done, &converged);
    num_planned_ba_iterations_ = converged ? 0 : std::max<int>(0, num_planned_ba_iterations_ - iterations_done);
  }
}

// B/bad_slam.cc:485-540
void BAScheduler::RunBundleAdjustment(u32 frame_index, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool optimize_poses,
                                      bool optimize_geometry, int min_iterations, int max_iterations, int active_keyframe_window_start,
                                      int active_keyframe_window_end, bool increase_ba_iteration_count, int* iterations_done,
                                      bool* converged, double time_limit, Timer* timer, std::function<bool(int)> progress_function) {
  vector<SE3f> original_keyframe_T_global;
  RememberKeyframePoses(direct_ba_, &original_keyframe_T_global);

  direct_ba_->BundleAdjustment(stream_, optimize_depth_intrinsics, optimize_color_intrinsics, config_.do_surfel_updates, optimize_poses,
                               optimize_geometry, min_iterations, max_iterations, config_.use_pcg, active_keyframe_window_start,
                               active_keyframe_window_end, increase_ba_iteration_count, iterations_done, converged, time_limit, timer,
                               config_.pcg_max_inner_iterations, config_.pcg_max_keyframes, progress_function);

  // Interpolate / extrapolate the pose update to non-keyframes
  PublishKeyframePosesNoLock();
  ExtrapolateAndInterpolateKeyframePoseChanges(config_.start_frame, frame_index, direct_ba_, original_keyframe_T_global, rgbd_video_);
  if (base_kf_) base_kf_global_T_frame_ = base_kf_->global_T_frame();
}

// B/bad_slam.cc:1164-1193
void BAScheduler::StartParallelIterations(int num_planned_iterations, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                                          bool do_surfel_updates, bool optimize_poses, bool optimize_geometry) {
  direct_ba_->Lock();
  ParallelBAOptions options;
  options.optimize_depth_intrinsics = optimize_depth_intrinsics;
  options.optimize_color_intrinsics = optimize_color_intrinsics;
  options.do_surfel_updates = do_surfel_updates;
  options.optimize_poses = optimize_poses;
  options.optimize_geometry = optimize_geometry;
  const int max_queued_iterations = config_.max_num_ba_iterations_per_keyframe;
  const int iterations_to_queue = std::min<int>(max_queued_iterations - (int)parallel_ba_iteration_queue_.size(), num_planned_iterations);
  for (int i = 0; i < iterations_to_queue; ++i) parallel_ba_iteration_queue_.push_back(options);
  direct_ba_->Unlock();
  zero_iterations_condition_.notify_all();
}

// B/bad_slam.cc:1195-1317
void BAScheduler::BAThreadMain() {
  hipStream_t thread_stream = nullptr;
  BAHIP_CHECKED_CALL(bahip_stream_create(&thread_stream));

  while (true) {
    std::unique_lock<std::mutex> lock(direct_ba_->Mutex());
    ba_thread_busy_ = false;
    idle_condition_.notify_all();
    while (parallel_ba_iteration_queue_.empty() && !quit_requested_) zero_iterations_condition_.wait(lock);
    if (quit_requested_) break;
    ba_thread_busy_ = true;

    const ParallelBAOptions options = parallel_ba_iteration_queue_.front();
    parallel_ba_iteration_queue_.erase(parallel_ba_iteration_queue_.begin());

    // Add any queued keyframes (the queue is read within the lock, the keyframe is added outside of it).
    bool mutex_locked = true;
    while (true) {
      if (!mutex_locked) { lock.lock(); mutex_locked = true; }
      if (queued_keyframes_.empty()) break;
      shared_ptr<Keyframe> new_keyframe = queued_keyframes_.front();
      const SE3f last_kf_tr_this_kf = queued_keyframes_last_kf_tr_this_kf_.front();
      // Convert relative to absolute pose: BA may have moved the previous keyframe since the odometry ran.
      if (!direct_ba_->keyframes().empty() && direct_ba_->keyframes().back())
        new_keyframe->set_global_T_frame(direct_ba_->keyframes().back()->global_T_frame() * last_kf_tr_this_kf);
      queued_keyframes_.erase(queued_keyframes_.begin());
      queued_keyframes_last_kf_tr_this_kf_.erase(queued_keyframes_last_kf_tr_this_kf_.begin());
      lock.unlock();
      mutex_locked = false;
      AddKeyframeToBA(thread_stream, new_keyframe);
      if (!config_.do_surfel_updates) direct_ba_->CreateSurfelsForKeyframe(thread_stream, true, new_keyframe);
    }
    lock.unlock();

    // Do a BA iteration.
    vector<SE3f> original_keyframe_T_global;
    RememberKeyframePoses(direct_ba_, &original_keyframe_T_global);
    if (config_.use_pcg)
      LOG(WARNING) << "PCG-based solving is not supported for running in parallel, using the alternating solver instead.";
    direct_ba_->BundleAdjustment(thread_stream, options.optimize_depth_intrinsics && config_.use_geometric_residuals,
                                 options.optimize_color_intrinsics && config_.use_photometric_residuals, options.do_surfel_updates,
                                 options.optimize_poses, options.optimize_geometry, /*min_iterations*/ 0, /*max_iterations*/ 1,
                                 /*use_pcg*/ false, /*active_keyframe_window_start*/ 0,
                                 /*active_keyframe_window_end*/ (int)direct_ba_->keyframes().size() - 1,
                                 /*increase_ba_iteration_count*/ false, nullptr, nullptr, 0, nullptr);

    direct_ba_->Lock();
    PublishKeyframePosesNoLock();
    ExtrapolateAndInterpolateKeyframePoseChanges(config_.start_frame, last_frame_index_, direct_ba_, original_keyframe_T_global, rgbd_video_);
    if (base_kf_) base_kf_global_T_frame_ = base_kf_->global_T_frame();
    ++parallel_iterations_done_;
    direct_ba_->Unlock();
  }

  BAHIP_CHECKED_CALL(bahip_stream_synchronize(thread_stream));
  bahip_stream_destroy(thread_stream);

  std::unique_lock<std::mutex> quit_lock(quit_mutex_);
  quit_done_ = true;
  quit_lock.unlock();
  quit_condition_.notify_all();
}

void BAScheduler::WaitForQueuedWork() {
  if (!ba_thread_) return;
  std::unique_lock<std::mutex> lock(direct_ba_->Mutex());
  while (ba_thread_busy_ || !parallel_ba_iteration_queue_.empty()) idle_condition_.wait(lock);
}

// B/bad_slam.cc:567-587
void BAScheduler::StopBAThreadAndWaitForIt() {
  if (!ba_thread_) return;
  std::unique_lock<std::mutex> lock(direct_ba_->Mutex());
  quit_requested_ = true;
  lock.unlock();
  zero_iterations_condition_.notify_all();

  std::unique_lock<std::mutex> quit_lock(quit_mutex_);
  while (!quit_done_) quit_condition_.wait(quit_lock);
  quit_lock.unlock();

  ba_thread_->join();
  ba_thread_.reset();
}

// B/bad_slam.cc:589-595
void BAScheduler::RestartBAThread() {
  StopBAThreadAndWaitForIt();
  quit_requested_ = false;
  quit_done_ = false;
  ba_thread_.reset(new std::thread(std::bind(&BAScheduler::BAThreadMain, this)));
}

}  // namespace vis
