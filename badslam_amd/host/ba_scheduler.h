// ba_scheduler.h -- the BA side of vis::BadSlam: what happens between "this frame became a keyframe"
// and "bundle adjustment moved the trajectory", restated with the reference's names (B/ = applications/badslam/src/badslam/):
//   * AddKeyframe              B/bad_slam.cc:1026-1102  (queue for the BA thread, or add directly; plan iterations)
//   * AddKeyframeToBA          B/bad_slam.cc:1126-1162  (without the loop detector, which is out of scope)
//   * RunBundleAdjustment      B/bad_slam.cc:485-540    (remember poses, BundleAdjustment, deform the trajectory)
//   * RunPlannedIterations     B/bad_slam.cc:214-281    (the tail of ProcessFrame: intrinsics heuristic, sequential or
//                                                         parallel dispatch of the planned iterations)
//   * StartParallelIterations  B/bad_slam.cc:1164-1193
//   * BAThreadMain             B/bad_slam.cc:1195-1317  (one BA iteration per queue entry on its own low-priority stream)
//   * StopBAThreadAndWaitForIt / RestartBAThread  B/bad_slam.cc:567-595
// The mutex protocol is the reference's: DirectBA::Mutex() guards the keyframe list, the iteration queue, the keyframe
// queue and the video poses; BundleAdjustment itself takes the lock around every state change (direct_ba.cc).
// The odometry front-end (pose tracking, keyframe selection, motion model) is not part of this backend: the caller
// supplies each new keyframe with its pose relative to the previous keyframe.
#pragma once

#include <condition_variable>
#include <memory>
#include <thread>

#include "rgbd_io.h"

namespace vis {

struct BASchedulerConfig {                        // the fields of B/bad_slam_config.h the BA side reads, with its defaults
  int start_frame = 0;                            // :52
  int max_num_ba_iterations_per_keyframe = 10;    // :185
  bool disable_deactivation = true;               // :194
  bool use_geometric_residuals = true;            // :198
  bool use_photometric_residuals = true;          // :203
  bool optimize_intrinsics = false;               // :207
  int intrinsics_optimization_interval = 10;      // :214
  bool do_surfel_updates = true;                  // :219
  bool parallel_ba = true;                        // :224
  bool use_pcg = false;                           // :230
  bool estimate_poses = true;                     // :237
  int pcg_max_inner_iterations = 30;              // B/bad_slam.h:132
  int pcg_max_keyframes = 2500;                   // B/bad_slam.h:133
};

class BAScheduler {
 public:
  // `stream` is the caller's ("odometry") stream; the BA thread creates its own.
  BAScheduler(const BASchedulerConfig& config, DirectBA* direct_ba, RGBDVideo<Vec3u8, u16>* rgbd_video, hipStream_t stream);
  ~BAScheduler();

  // B/bad_slam.cc:1026-1102.  `last_kf_tr_this_kf`: pose of the new keyframe relative to the previous one (what the
  // odometry estimated); in parallel mode it is applied when the BA thread takes the keyframe from the queue, on top
  // of whatever pose BA has given the previous keyframe by then.  In sequential mode the keyframe keeps its pose.
  void AddKeyframe(const shared_ptr<Keyframe>& new_keyframe, const SE3f& last_kf_tr_this_kf);

  // B/bad_slam.cc:214-281: spends the planned iterations -- hands them to the BA thread, or runs them here.
  // `frame_index`: the newest frame of the video (the trajectory deformation reaches up to it).
  void RunPlannedIterations(u32 frame_index);

  // B/bad_slam.cc:485-540
  void RunBundleAdjustment(u32 frame_index, bool optimize_depth_intrinsics, bool optimize_color_intrinsics, bool optimize_poses,
                           bool optimize_geometry, int min_iterations, int max_iterations, int active_keyframe_window_start,
                           int active_keyframe_window_end, bool increase_ba_iteration_count, int* iterations_done, bool* converged,
                           double time_limit = 0, Timer* timer = nullptr, std::function<bool(int)> progress_function = nullptr);

  // B/bad_slam.cc:1164-1193
  void StartParallelIterations(int num_planned_iterations, bool optimize_depth_intrinsics, bool optimize_color_intrinsics,
                               bool do_surfel_updates, bool optimize_poses, bool optimize_geometry);
  // B/bad_slam.cc:567-595
  void StopBAThreadAndWaitForIt();
  void RestartBAThread();

  // Not in the reference (its GUI polls): blocks until the BA thread has emptied both queues and is idle.
  void WaitForQueuedWork();

  int num_planned_ba_iterations() const { return num_planned_ba_iterations_; }
  int parallel_iterations_done() const { return parallel_iterations_done_; }
  void SetLastFrameIndex(int frame_index);       // newest frame the front-end has a pose for (B/bad_slam.h: last_frame_index_)
  Keyframe* base_kf() const { return base_kf_; }
  SE3f base_kf_global_T_frame() const;            // cached under the lock, B/bad_slam.cc:1006-1016
  void GetQueuedKeyframes(vector<shared_ptr<Keyframe>>* queued_keyframes, vector<SE3f>* queued_keyframes_last_kf_tr_this_kf) const;

 private:
  struct ParallelBAOptions {                      // B/bad_slam.h:ParallelBAOptions
    bool optimize_depth_intrinsics, optimize_color_intrinsics, do_surfel_updates, optimize_poses, optimize_geometry;
  };
  void AddKeyframeToBA(hipStream_t stream, const shared_ptr<Keyframe>& new_keyframe);
  void BAThreadMain();
  // Copies the keyframes' poses into their video frames (the reference shares one pose object between the two).
  void PublishKeyframePosesNoLock();

  BASchedulerConfig config_;
  DirectBA* direct_ba_;
  RGBDVideo<Vec3u8, u16>* rgbd_video_;
  hipStream_t stream_;

  int num_planned_ba_iterations_ = 0;
  int bundle_adjustment_counter_ = 0;
  int last_frame_index_ = 0;
  Keyframe* base_kf_ = nullptr;
  SE3f base_kf_global_T_frame_;

  // guarded by direct_ba_->Mutex()
  vector<ParallelBAOptions> parallel_ba_iteration_queue_;
  vector<shared_ptr<Keyframe>> queued_keyframes_;
  vector<SE3f> queued_keyframes_last_kf_tr_this_kf_;
  bool ba_thread_busy_ = false;
  int parallel_iterations_done_ = 0;
  bool quit_requested_ = false;
  std::condition_variable zero_iterations_condition_;
  std::condition_variable idle_condition_;

  std::mutex quit_mutex_;
  std::condition_variable quit_condition_;
  bool quit_done_ = false;
  std::unique_ptr<std::thread> ba_thread_;
};

}  // namespace vis
