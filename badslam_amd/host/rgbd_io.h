// rgbd_io.h -- the data formats either side of the BA path, restated with the reference's
// names (L/ = libvis/src/libvis/, B/ = applications/badslam/src/badslam/):
//   * RGBDVideo / ImageFrame with lazily loaded images          L/rgbd_video.h, L/image_frame.h
//   * TUM RGB-D dataset reader (associated.txt, calibration.txt with cx, cy + 0.5, trajectory with slerp
//     interpolation)                                              L/rgbd_video_io_tum_dataset.h:42-240
//   * PNG decoding (8-bit RGB / RGBA / grey, 16-bit grey; the reference links libpng, here zlib + the PNG filters)
//   * SavePoses / LoadPoses (TUM trajectory lines), SaveCalibration / LoadCalibration, ExportToPointCloud +
//     SavePointCloudAsPLY                                        B/io.cc:537-703, B/direct_ba.cc:461-547
//   * trajectory deformation of the non-keyframes after BA          B/trajectory_deformation.cc:33-130
//   * PreprocessFrame + CreateKeyframe: raw depth + RGB of a video frame -> bilateral filter, normals, radii, luma,
//     min / max depth -> vis::Keyframe                            B/bad_slam.cc:643-765, 957-1001
#pragma once

#include <string>

#include "direct_ba.h"

namespace vis {

// ---- PNG ---------------------------------------------------------------------------------------------------------------
// Decodes a non-interlaced PNG.  Colour images: colour type 2 (RGB) or 6 (RGBA, alpha dropped) or 0 (grey, replicated),
// 8 bits.  Depth images: colour type 0, 16 bits (big-endian samples), as the TUM benchmark stores them.
bool ReadPNG(const std::string& path, Image<Vec3u8>* image);
bool ReadPNG(const std::string& path, Image<u16>* image);

// ---- ImageFrame / RGBDVideo (L/image_frame.h:47-150, L/rgbd_video.h:46-120) ---------------------------------------------
template <typename T>
class ImageFrame {
 public:
  ImageFrame(const std::string& image_path, double timestamp, const std::string& timestamp_string)
      : path_(image_path), timestamp_(timestamp), timestamp_string_(timestamp_string) {}
  explicit ImageFrame(const shared_ptr<Image<T>>& image, double timestamp = 0, const std::string& timestamp_string = "")
      : timestamp_(timestamp), timestamp_string_(timestamp_string), image_(image) {}

  // Loads the image on first use (L/image_frame.h:97-109).
  shared_ptr<Image<T>> GetImage() {
    if (!image_ && !path_.empty()) {
      shared_ptr<Image<T>> loaded(new Image<T>());
      if (ReadPNG(path_, loaded.get())) image_ = loaded;
      else LOG(ERROR) << "Cannot read image: " << path_;
    }
    return image_;
  }
  void ClearImageAndDerivedData() { if (!path_.empty()) image_.reset(); }

  // both directions cached on every set (L/image_frame.h:84-94)
  void SetGlobalTFrame(const SE3f& global_T_frame) { global_T_frame_ = global_T_frame; frame_T_global_ = global_T_frame.inverse(); }
  void SetFrameTGlobal(const SE3f& frame_T_global) { frame_T_global_ = frame_T_global; global_T_frame_ = frame_T_global.inverse(); }
  const SE3f& global_T_frame() const { return global_T_frame_; }
  const SE3f& frame_T_global() const { return frame_T_global_; }
  double timestamp() const { return timestamp_; }
  const std::string& timestamp_string() const { return timestamp_string_; }
  const std::string& path() const { return path_; }

 private:
  std::string path_;
  double timestamp_;
  std::string timestamp_string_;
  shared_ptr<Image<T>> image_;
  SE3f global_T_frame_, frame_T_global_;
};
template <typename T> using ImageFramePtr = shared_ptr<ImageFrame<T>>;

template <typename ColorT, typename DepthT>
class RGBDVideo {
 public:
  usize frame_count() const { return depth_frames_.size(); }
  const ImageFramePtr<ColorT>& color_frame(usize i) const { return color_frames_[i]; }
  const ImageFramePtr<DepthT>& depth_frame(usize i) const { return depth_frames_[i]; }
  ImageFramePtr<ColorT>& color_frame_mutable(usize i) { return color_frames_[i]; }
  ImageFramePtr<DepthT>& depth_frame_mutable(usize i) { return depth_frames_[i]; }
  vector<ImageFramePtr<ColorT>>* color_frames_mutable() { return &color_frames_; }
  vector<ImageFramePtr<DepthT>>* depth_frames_mutable() { return &depth_frames_; }
  const shared_ptr<PinholeCamera4f>& color_camera() const { return color_camera_; }
  const shared_ptr<PinholeCamera4f>& depth_camera() const { return depth_camera_; }
  shared_ptr<PinholeCamera4f>* color_camera_mutable() { return &color_camera_; }
  shared_ptr<PinholeCamera4f>* depth_camera_mutable() { return &depth_camera_; }

 private:
  vector<ImageFramePtr<ColorT>> color_frames_;
  vector<ImageFramePtr<DepthT>> depth_frames_;
  shared_ptr<PinholeCamera4f> color_camera_, depth_camera_;
};

// ---- TUM RGB-D format (L/rgbd_video_io_tum_dataset.h) --------------------------------------------------------------------
bool InterpolatePose(double timestamp, const vector<double>& pose_timestamps, const vector<SE3f>& poses, SE3f* pose);   // :42-72
bool ReadTUMRGBDTrajectory(const char* path, vector<double>* pose_timestamps, vector<SE3f>* poses_global_T_frame);      // :74-118
// :120-240.  trajectory_filename may be nullptr (poses stay identity).
bool ReadTUMRGBDDatasetAssociatedAndCalibrated(const char* dataset_folder_path, const char* trajectory_filename,
                                               RGBDVideo<Vec3u8, u16>* rgbd_video);

// ---- results (B/io.cc) -----------------------------------------------------------------------------------------------------
// B/io.cc:537-568: one "timestamp tx ty tz qx qy qz qw" line per frame, poses expressed relative to `start_frame`.
bool SavePoses(const RGBDVideo<Vec3u8, u16>& rgbd_video, bool use_depth_timestamps, int start_frame, const std::string& export_poses_path);
// B/io.cc:570-632 / :635-692: <base>.depth_intrinsics.txt, <base>.color_intrinsics.txt (cx, cy - 0.5), <base>.deformation.txt
bool SaveCalibration(hipStream_t stream, DirectBA& direct_ba, const std::string& export_base_path);
bool LoadCalibration(DirectBA* direct_ba, const std::string& import_base_path);

struct Point3fC3u8Nf { float position[3]; u8 color[3]; float normal[3]; };   // L/point_cloud.h Point3fC3u8NfCloud element
// B/direct_ba.cc:461-547: valid (non-NaN) surfels only, in index order, by row downloads.
void ExportToPointCloud(hipStream_t stream, DirectBA& direct_ba, vector<Point3fC3u8Nf>* cloud);
// B/io.cc:694-703 + L/point_cloud.h:493-533 (binary little-endian PLY: x y z, red green blue, nx ny nz)
bool SavePointCloudAsPLY(hipStream_t stream, DirectBA& direct_ba, const std::string& export_path);

// ---- trajectory deformation (B/trajectory_deformation.cc:33-130) -----------------------------------------------------------
// After BA moved the keyframes, the frames between them follow: RememberKeyframePoses before BA, then every
// non-keyframe in [start_frame, end_frame] gets the pose change of its neighbouring keyframes, interpolated (translation
// linearly, rotation by slerp) by frame index, or extrapolated from the nearest keyframe at the ends.
void RememberKeyframePoses(DirectBA* dense_ba, vector<SE3f>* original_keyframe_T_global);
void ExtrapolateAndInterpolateKeyframePoseChanges(u32 start_frame, u32 end_frame, DirectBA* dense_ba,
                                                  const vector<SE3f>& original_keyframe_T_global, RGBDVideo<Vec3u8, u16>* rgbd_video);

// ---- frame -> keyframe (B/bad_slam.cc:643-765, 957-1001) -----------------------------------------------------------------
struct PreprocessConfig {            // defaults of B/bad_slam_config.h:96-122
  float max_depth = 3.0f;
  float bilateral_filter_sigma_xy = 1.5f;
  float bilateral_filter_radius_factor = 2.0f;
  float bilateral_filter_sigma_inv_depth = 0.005f;
};
// Uploads the frame's depth and colour, computes luma, filtered depth, normals, radii (isolated pixels removed) and the
// depth range, and builds the keyframe with the frame's current pose.
shared_ptr<Keyframe> CreateKeyframeFromFrame(hipStream_t stream, const PreprocessConfig& config, DirectBA& direct_ba,
                                             RGBDVideo<Vec3u8, u16>& rgbd_video, int frame_index);

// ---- binary state (B/io.cc:38-183 SaveState, :185-535 LoadState): checkpoint / resume of the BA backend -------------------
// The sections of the reference's state file that belong to the backend, with the reference's field order and types:
// "RGBDVideo (frame poses)" (B/io.cc:112-117) and "Direct BA" (:120-180: cameras, cfactor image, depth parameters,
// keyframe table, the kSurfelDataAttributeCount = 8 surfel rows, iteration counters, residual switches, observation
// thresholds, merge factor).  The front-end sections (motion model, queued keyframes, BadSlamConfig) have no counterpart
// here and are not written, so the container announces itself as "BADSLAM" with version 101: a reference build refuses it
// ("Unknown file format version") instead of misreading it.  Keyframe images are not stored (as in the reference): on
// loading, every keyframe is rebuilt from its video frame by CreateKeyframeFromFrame, with the loaded pose.
bool SaveState(hipStream_t stream, const RGBDVideo<Vec3u8, u16>& rgbd_video, DirectBA& direct_ba, const std::string& path);
// progress_function(keyframes_done, keyframe_count) may return false to abort (B/io.cc:396-403).
bool LoadState(hipStream_t stream, const PreprocessConfig& config, RGBDVideo<Vec3u8, u16>* rgbd_video, DirectBA* direct_ba,
               const std::string& path, std::function<bool(int, int)> progress_function = nullptr);

}  // namespace vis
