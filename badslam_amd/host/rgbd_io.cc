// rgbd_io.cc -- see rgbd_io.h.
#include "rgbd_io.h"

#include <zlib.h>

#include <cmath>
#include <fstream>
#include <iomanip>
#include <limits>

namespace vis {

// ---- PNG (ISO/IEC 15948): signature, IHDR, concatenated IDAT -> inflate -> per-row filters -------------------------------
namespace {

struct PngRaw {
  u32 width = 0, height = 0;
  int bit_depth = 0, color_type = 0, channels = 0;
  std::vector<u8> pixels;   // unfiltered scanlines, width * channels * bit_depth / 8 bytes per row
};

u32 be32(const u8* p) { return ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3]; }

bool DecodePNG(const std::string& path, PngRaw* out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<u8> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const u8 kSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || memcmp(file.data(), kSignature, 8) != 0) return false;
  std::vector<u8> compressed;
  size_t at = 8;
  bool have_header = false;
  int interlace = 0;
  while (at + 12 <= file.size()) {
    const u32 length = be32(&file[at]);
    const char* type = reinterpret_cast<const char*>(&file[at + 4]);
    if (at + 12 + (size_t)length > file.size()) return false;
    const u8* data = &file[at + 8];
    if (!memcmp(type, "IHDR", 4)) {
      if (length != 13) return false;
      out->width = be32(data); out->height = be32(data + 4);
      out->bit_depth = data[8]; out->color_type = data[9];
      interlace = data[12];
      have_header = true;
    } else if (!memcmp(type, "IDAT", 4)) {
      compressed.insert(compressed.end(), data, data + length);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    at += 12 + (size_t)length;
  }
  if (!have_header || interlace != 0 || out->width == 0 || out->height == 0) return false;
  switch (out->color_type) {
    case 0: out->channels = 1; break;
    case 2: out->channels = 3; break;
    case 4: out->channels = 2; break;
    case 6: out->channels = 4; break;
    default: return false;   // palette images are not used by RGB-D datasets
  }
  if (out->bit_depth != 8 && out->bit_depth != 16) return false;
  const size_t bpp = (size_t)out->channels * out->bit_depth / 8;   // bytes per pixel = filter distance
  const size_t row_bytes = (size_t)out->width * bpp;
  // A header may announce any size up to 2^31 squared; deflate expands by at most 1032:1, so more scanline bytes than
  // that cannot be in this file (and are not allocated for).
  const size_t max_raw = compressed.size() * 1032 + 64;
  if (row_bytes + 1 > max_raw / out->height) return false;
  std::vector<u8> raw((row_bytes + 1) * out->height);
  uLongf raw_size = (uLongf)raw.size();
  if (uncompress(raw.data(), &raw_size, compressed.data(), (uLong)compressed.size()) != Z_OK || raw_size != raw.size()) return false;
  out->pixels.assign(row_bytes * out->height, 0);
  for (u32 y = 0; y < out->height; ++y) {
    const u8 filter = raw[(row_bytes + 1) * y];
    const u8* src = &raw[(row_bytes + 1) * y + 1];
    u8* dst = &out->pixels[row_bytes * y];
    const u8* up = y ? dst - row_bytes : nullptr;
    for (size_t i = 0; i < row_bytes; ++i) {
      const int a = i >= bpp ? dst[i - bpp] : 0;            // left
      const int b = up ? up[i] : 0;                          // above
      const int c = (up && i >= bpp) ? up[i - bpp] : 0;      // above-left
      int predicted = 0;
      switch (filter) {
        case 0: predicted = 0; break;
        case 1: predicted = a; break;
        case 2: predicted = b; break;
        case 3: predicted = (a + b) / 2; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          predicted = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: return false;
      }
      dst[i] = (u8)(src[i] + predicted);
    }
  }
  return true;
}

}  // namespace

bool ReadPNG(const std::string& path, Image<Vec3u8>* image) {
  PngRaw png;
  if (!DecodePNG(path, &png) || png.bit_depth != 8) return false;
  image->SetSize(png.width, png.height);
  const size_t n = (size_t)png.width * png.height;
  for (size_t i = 0; i < n; ++i) {
    const u8* p = &png.pixels[i * png.channels];
    image->data()[i] = (png.channels >= 3) ? Vec3u8(p[0], p[1], p[2]) : Vec3u8(p[0], p[0], p[0]);
  }
  return true;
}

bool ReadPNG(const std::string& path, Image<u16>* image) {
  PngRaw png;
  if (!DecodePNG(path, &png) || png.color_type != 0) return false;
  image->SetSize(png.width, png.height);
  const size_t n = (size_t)png.width * png.height;
  for (size_t i = 0; i < n; ++i)
    image->data()[i] = (png.bit_depth == 16) ? (u16)(((u16)png.pixels[2 * i] << 8) | png.pixels[2 * i + 1]) : (u16)png.pixels[i];
  return true;
}

// ---- TUM RGB-D --------------------------------------------------------------------------------------------------------
namespace {
// Eigen's Quaternion::slerp (what Sophus poses are interpolated with, L/rgbd_video_io_tum_dataset.h:62-64)
void Slerp(const float* qa, const float* qb, double t, float* out) {
  double d = (double)qa[0] * qb[0] + (double)qa[1] * qb[1] + (double)qa[2] * qb[2] + (double)qa[3] * qb[3];
  const double abs_d = std::fabs(d);
  double scale0, scale1;
  if (abs_d >= 1.0 - std::numeric_limits<double>::epsilon()) {
    scale0 = 1.0 - t; scale1 = t;
  } else {
    const double theta = std::acos(abs_d), sin_theta = std::sin(theta);
    scale0 = std::sin((1.0 - t) * theta) / sin_theta;
    scale1 = std::sin(t * theta) / sin_theta;
  }
  if (d < 0) scale1 = -scale1;
  double q[4], norm = 0;
  for (int c = 0; c < 4; ++c) { q[c] = scale0 * qa[c] + scale1 * qb[c]; norm += q[c] * q[c]; }
  norm = std::sqrt(norm);
  for (int c = 0; c < 4; ++c) out[c] = (float)(q[c] / norm);
}
}  // namespace

bool InterpolatePose(double timestamp, const vector<double>& pose_timestamps, const vector<SE3f>& poses, SE3f* pose) {
  CHECK_EQ(pose_timestamps.size(), poses.size());
  CHECK_GE(pose_timestamps.size(), 2u);
  if (timestamp <= pose_timestamps[0]) { *pose = poses[0]; return true; }
  if (timestamp >= pose_timestamps.back()) { *pose = poses.back(); return true; }
  for (usize i = 0; i + 1 < pose_timestamps.size(); ++i) {
    if (timestamp >= pose_timestamps[i] && timestamp <= pose_timestamps[i + 1]) {
      const double factor = (timestamp - pose_timestamps[i]) / (pose_timestamps[i + 1] - pose_timestamps[i]);
      float v[7];
      Slerp(poses[i].data(), poses[i + 1].data(), factor, v);
      for (int c = 0; c < 3; ++c) v[4 + c] = (float)(poses[i].translation()[c] + factor * ((double)poses[i + 1].translation()[c] - poses[i].translation()[c]));
      *pose = SE3f(v);
      return true;
    }
  }
  return false;
}

bool ReadTUMRGBDTrajectory(const char* path, vector<double>* pose_timestamps, vector<SE3f>* poses_global_T_frame) {
  std::ifstream trajectory_file(path);
  if (!trajectory_file) { LOG(ERROR) << "Could not open trajectory file: " << path; return false; }
  std::string line;
  while (std::getline(trajectory_file, line)) {
    if (line.empty()) break;            // the reference stops at the first empty line
    if (line[0] == '#') continue;
    char time_string[128];
    double t[3], q[4];
    if (sscanf(line.c_str(), "%127s %lf %lf %lf %lf %lf %lf %lf", time_string, &t[0], &t[1], &t[2], &q[0], &q[1], &q[2], &q[3]) != 8) {
      LOG(ERROR) << "Cannot read poses! Line: " << line;
      return false;
    }
    const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);   // Sophus normalises the quaternion
    const float v[7] = {(float)(q[0] / qn), (float)(q[1] / qn), (float)(q[2] / qn), (float)(q[3] / qn), (float)t[0], (float)t[1], (float)t[2]};
    pose_timestamps->push_back(atof(time_string));
    poses_global_T_frame->push_back(SE3f(v));
  }
  return true;
}

bool ReadTUMRGBDDatasetAssociatedAndCalibrated(const char* dataset_folder_path, const char* trajectory_filename,
                                               RGBDVideo<Vec3u8, u16>* rgbd_video) {
  rgbd_video->color_frames_mutable()->clear();
  rgbd_video->depth_frames_mutable()->clear();
  const std::string folder(dataset_folder_path);
  std::ifstream calibration_file(folder + "/calibration.txt");
  if (!calibration_file) { LOG(ERROR) << "Could not open calibration file: " << folder << "/calibration.txt"; return false; }
  std::string line;
  std::getline(calibration_file, line);
  double fx, fy, cx, cy;
  if (sscanf(line.c_str(), "%lf %lf %lf %lf", &fx, &fy, &cx, &cy) != 4) { LOG(ERROR) << "Cannot read calibration!"; return false; }

  vector<double> pose_timestamps;
  vector<SE3f> poses_global_T_frame;
  if (trajectory_filename != nullptr && !ReadTUMRGBDTrajectory((folder + "/" + trajectory_filename).c_str(), &pose_timestamps, &poses_global_T_frame))
    return false;

  u32 width = 0, height = 0;
  std::ifstream associated_file(folder + "/associated.txt");
  if (!associated_file) { LOG(ERROR) << "Could not open associated file: " << folder << "/associated.txt"; return false; }
  while (std::getline(associated_file, line)) {
    if (line.empty() || line[0] == '#') continue;
    char rgb_time_string[128], rgb_filename[128], depth_time_string[128], depth_filename[128];
    if (sscanf(line.c_str(), "%127s %127s %127s %127s", rgb_time_string, rgb_filename, depth_time_string, depth_filename) != 4) {
      LOG(ERROR) << "Cannot read association line!";
      return false;
    }
    SE3f rgb_global_T_frame, depth_global_T_frame;
    const double rgb_timestamp = atof(rgb_time_string), depth_timestamp = atof(depth_time_string);
    if (!poses_global_T_frame.empty()) {
      if (!InterpolatePose(rgb_timestamp, pose_timestamps, poses_global_T_frame, &rgb_global_T_frame)) continue;
      if (!InterpolatePose(depth_timestamp, pose_timestamps, poses_global_T_frame, &depth_global_T_frame)) continue;
    }
    ImageFramePtr<Vec3u8> image_frame(new ImageFrame<Vec3u8>(folder + "/" + rgb_filename, rgb_timestamp, rgb_time_string));
    image_frame->SetGlobalTFrame(rgb_global_T_frame);
    rgbd_video->color_frames_mutable()->push_back(image_frame);
    ImageFramePtr<u16> depth_frame(new ImageFrame<u16>(folder + "/" + depth_filename, depth_timestamp, depth_time_string));
    depth_frame->SetGlobalTFrame(depth_global_T_frame);
    rgbd_video->depth_frames_mutable()->push_back(depth_frame);
    if (width == 0) {   // image size from the first colour image
      shared_ptr<Image<Vec3u8>> image_ptr = image_frame->GetImage();
      if (!image_ptr) { LOG(ERROR) << "Cannot load image to determine image dimensions."; return false; }
      width = image_ptr->width(); height = image_ptr->height();
      image_frame->ClearImageAndDerivedData();
    }
  }
  // calibration.txt is in the pixel-centre convention, PinholeCamera4f in the pixel-corner convention (:229-233)
  const float camera_parameters[4] = {(float)fx, (float)fy, (float)(cx + 0.5), (float)(cy + 0.5)};
  rgbd_video->color_camera_mutable()->reset(new PinholeCamera4f(width, height, camera_parameters));
  rgbd_video->depth_camera_mutable()->reset(new PinholeCamera4f(width, height, camera_parameters));
  return true;
}

// ---- results ------------------------------------------------------------------------------------------------------------
bool SavePoses(const RGBDVideo<Vec3u8, u16>& rgbd_video, bool use_depth_timestamps, int start_frame, const std::string& export_poses_path) {
  const SE3f start_frame_T_global = rgbd_video.depth_frame(start_frame)->frame_T_global();
  std::ofstream poses_file(export_poses_path, std::ios::out);
  if (!poses_file) return false;
  poses_file << std::setprecision(std::numeric_limits<double>::digits10 + 1);
  poses_file << "# Format: Each line gives one global_T_frame pose with values: tx ty tz qx qy qz qw" << std::endl;
  for (usize frame_index = 0; frame_index < rgbd_video.frame_count(); ++frame_index) {
    const SE3f global_T_frame = start_frame_T_global * rgbd_video.depth_frame(frame_index)->global_T_frame();
    const float* v = global_T_frame.data();
    poses_file << (use_depth_timestamps ? rgbd_video.depth_frame(frame_index)->timestamp_string()
                                        : rgbd_video.color_frame(frame_index)->timestamp_string())
               << " " << v[4] << " " << v[5] << " " << v[6] << " " << v[0] << " " << v[1] << " " << v[2] << " " << v[3] << std::endl;
  }
  return true;
}

bool SaveCalibration(hipStream_t stream, DirectBA& direct_ba, const std::string& export_base_path) {
  const struct { const char* suffix; PinholeCamera4f camera; } cameras[2] = {{".depth_intrinsics.txt", direct_ba.depth_camera()},
                                                                             {".color_intrinsics.txt", direct_ba.color_camera()}};
  for (const auto& c : cameras) {
    std::ofstream calibration_file(export_base_path + c.suffix, std::ios::out);
    if (!calibration_file) return false;
    calibration_file << c.camera.parameters()[0] << " " << c.camera.parameters()[1] << " " << (c.camera.parameters()[2] - 0.5) << " "
                     << (c.camera.parameters()[3] - 0.5);
  }
  CUDABufferPtr<float> cfactor_buffer = direct_ba.cfactor_buffer();
  std::ofstream deformation_file(export_base_path + ".deformation.txt", std::ios::out);
  if (!deformation_file) return false;
  deformation_file.precision(8);
  deformation_file << cfactor_buffer->width() << " " << cfactor_buffer->height() << std::endl;
  deformation_file << direct_ba.a() << std::endl;
  Image<float> cfactor_buffer_cpu(cfactor_buffer->width(), cfactor_buffer->height());
  cfactor_buffer->DownloadAsync(stream, &cfactor_buffer_cpu);
  BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
  for (u32 y = 0; y < cfactor_buffer_cpu.height(); ++y)
    for (u32 x = 0; x < cfactor_buffer_cpu.width(); ++x) deformation_file << cfactor_buffer_cpu(x, y) << std::endl;
  return true;
}

bool LoadCalibration(DirectBA* direct_ba, const std::string& import_base_path) {
  for (int which = 0; which < 2; ++which) {
    const std::string intrinsics_path = import_base_path + (which == 0 ? ".depth_intrinsics.txt" : ".color_intrinsics.txt");
    std::ifstream calibration_file(intrinsics_path, std::ios::in);
    if (!calibration_file) { LOG(ERROR) << "Cannot read file: " << intrinsics_path; return false; }
    float intrinsics[4];
    calibration_file >> intrinsics[0] >> intrinsics[1] >> intrinsics[2] >> intrinsics[3];
    intrinsics[2] += 0.5f;
    intrinsics[3] += 0.5f;
    if (which == 0) direct_ba->SetDepthCamera(PinholeCamera4f(direct_ba->depth_camera().width(), direct_ba->depth_camera().height(), intrinsics));
    else direct_ba->SetColorCamera(PinholeCamera4f(direct_ba->color_camera().width(), direct_ba->color_camera().height(), intrinsics));
  }
  const std::string deformation_path = import_base_path + ".deformation.txt";
  std::ifstream deformation_file(deformation_path, std::ios::in);
  if (!deformation_file) { LOG(ERROR) << "Cannot read file: " << deformation_path; return false; }
  int cfactor_buffer_width, cfactor_buffer_height;
  deformation_file >> cfactor_buffer_width >> cfactor_buffer_height;
  CUDABufferPtr<float> cfactor_buffer = direct_ba->cfactor_buffer();
  if (cfactor_buffer_width != cfactor_buffer->width() || cfactor_buffer_height != cfactor_buffer->height()) {
    LOG(ERROR) << "cfactor buffer size mismatch in current configuration vs. imported deformation - need to implement rescaling";
    return false;
  }
  deformation_file >> direct_ba->a();
  Image<float> cfactor_buffer_cpu(cfactor_buffer->width(), cfactor_buffer->height());
  for (u32 y = 0; y < cfactor_buffer_cpu.height(); ++y)
    for (u32 x = 0; x < cfactor_buffer_cpu.width(); ++x) deformation_file >> cfactor_buffer_cpu(x, y);
  cfactor_buffer->UploadAsync(nullptr, cfactor_buffer_cpu);
  return true;
}

// the member of B/direct_ba.h:175 (declared in direct_ba.h; the element type and the row downloads live in this file)
void DirectBA::ExportToPointCloud(hipStream_t stream, vector<Point3fC3u8Nf>* cloud) { vis::ExportToPointCloud(stream, *this, cloud); }
void ExportToPointCloud(hipStream_t stream, DirectBA& direct_ba, vector<Point3fC3u8Nf>* cloud) {
  const u32 surfels_size = direct_ba.surfels_size();
  cloud->clear();
  if (surfels_size == 0) return;
  CUDABufferPtr<float> surfels = direct_ba.surfels();
  const size_t pitch = surfels->ToCUDA().pitch();
  vector<float> rows[5];
  const int row_index[5] = {kSurfelX, kSurfelY, kSurfelZ, kSurfelColor, kSurfelNormal};
  for (int r = 0; r < 5; ++r) {
    rows[r].resize(surfels_size);
    surfels->DownloadPartAsync(row_index[r] * pitch, surfels_size * sizeof(float), stream, rows[r].data());
  }
  BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
  cloud->reserve(direct_ba.surfel_count());
  for (u32 i = 0; i < surfels_size; ++i) {
    if (std::isnan(rows[0][i])) continue;   // deleted surfel
    Point3fC3u8Nf p;
    p.position[0] = rows[0][i]; p.position[1] = rows[1][i]; p.position[2] = rows[2][i];
    u32 color, packed;
    memcpy(&color, &rows[3][i], 4);
    memcpy(&packed, &rows[4][i], 4);
    p.color[0] = color & 0xff; p.color[1] = (color >> 8) & 0xff; p.color[2] = (color >> 16) & 0xff;
    float n[3];
    for (int c = 0; c < 3; ++c) {   // TenBitSignedToFloat, B/util_nvcc_only.cuh:51-63
      const int32_t s = (int32_t)((packed >> (10 * c)) << 22) >> 22;
      n[c] = (float)s * (1.0f / 511.0f);
    }
    const float factor = 1.0f / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int c = 0; c < 3; ++c) p.normal[c] = factor * n[c];
    cloud->push_back(p);
  }
  if (cloud->size() != direct_ba.surfel_count())
    LOG(ERROR) << "surfel_count_ (" << direct_ba.surfel_count() << ") is not consistent with the actual number of valid surfels (" << cloud->size() << ")!";
}

bool SavePointCloudAsPLY(hipStream_t stream, DirectBA& direct_ba, const std::string& export_path) {
  vector<Point3fC3u8Nf> cloud;
  ExportToPointCloud(stream, direct_ba, &cloud);
  FILE* file = fopen(export_path.c_str(), "wb");
  if (!file) return false;
  std::ostringstream header;
  header << "ply\nformat binary_little_endian 1.0\nelement vertex " << cloud.size() << "\n"
         << "property float x\nproperty float y\nproperty float z\n"
         << "property uchar red\nproperty uchar green\nproperty uchar blue\n"
         << "property float nx\nproperty float ny\nproperty float nz\nend_header\n";
  const std::string header_string = header.str();
  fwrite(header_string.data(), 1, header_string.size(), file);
  for (const Point3fC3u8Nf& p : cloud) {
    fwrite(p.position, sizeof(float), 3, file);
    fwrite(p.color, 1, 3, file);
    fwrite(p.normal, sizeof(float), 3, file);
  }
  fclose(file);
  return true;
}

// ---- trajectory deformation ----------------------------------------------------------------------------------------------
void RememberKeyframePoses(DirectBA* dense_ba, vector<SE3f>* original_keyframe_T_global) {
  original_keyframe_T_global->resize(dense_ba->keyframes().size());
  for (usize keyframe_index = 0; keyframe_index < dense_ba->keyframes().size(); ++keyframe_index)
    if (dense_ba->keyframes()[keyframe_index]) original_keyframe_T_global->at(keyframe_index) = dense_ba->keyframes()[keyframe_index]->frame_T_global();
}

void ExtrapolateAndInterpolateKeyframePoseChanges(u32 start_frame, u32 end_frame, DirectBA* dense_ba,
                                                  const vector<SE3f>& original_keyframe_T_global, RGBDVideo<Vec3u8, u16>* rgbd_video) {
  end_frame = std::min<int>(end_frame, (int)rgbd_video->frame_count() - 1);
  const auto& keyframes = dense_ba->keyframes();
  usize prev_keyframe_index = 0, next_keyframe_index = 0;
  for (usize other_frame_index = start_frame; other_frame_index <= end_frame; ++other_frame_index) {
    while (next_keyframe_index < keyframes.size() && keyframes[next_keyframe_index]->frame_index() <= other_frame_index) {
      prev_keyframe_index = next_keyframe_index;
      ++next_keyframe_index;
      while (next_keyframe_index < keyframes.size() && !keyframes[next_keyframe_index]) ++next_keyframe_index;
    }
    Keyframe* prev_keyframe = keyframes[prev_keyframe_index].get();
    Keyframe* next_keyframe = (next_keyframe_index < keyframes.size()) ? keyframes[next_keyframe_index].get() : nullptr;
    if (prev_keyframe->frame_index() == other_frame_index) continue;   // a keyframe: nothing to do
    const SE3f old_global_T_other = rgbd_video->depth_frame_mutable(other_frame_index)->global_T_frame();
    const SE3f old_other_T_global = rgbd_video->depth_frame_mutable(other_frame_index)->frame_T_global();
    SE3f new_global_T_other_frame;
    if (next_keyframe == nullptr || prev_keyframe->frame_index() > other_frame_index) {   // extrapolate at the end / at the start
      const SE3f old_kf_T_other_frame = original_keyframe_T_global[prev_keyframe_index] * old_global_T_other;
      new_global_T_other_frame = prev_keyframe->global_T_frame() * old_kf_T_other_frame;
    } else {
      const SE3f from_prev = old_other_T_global * (prev_keyframe->global_T_frame() * (original_keyframe_T_global[prev_keyframe_index] * old_global_T_other));
      const SE3f from_next = old_other_T_global * (next_keyframe->global_T_frame() * (original_keyframe_T_global[next_keyframe_index] * old_global_T_other));
      const u32 prev_frame = prev_keyframe->frame_index(), next_frame = next_keyframe->frame_index();
      const float factor = (other_frame_index - prev_frame) * 1.0f / (next_frame - prev_frame);
      float v[7];
      Slerp(from_prev.data(), from_next.data(), factor, v);
      for (int c = 0; c < 3; ++c) v[4 + c] = (1 - factor) * from_prev.translation()[c] + factor * from_next.translation()[c];
      new_global_T_other_frame = old_global_T_other * SE3f(v);
    }
    rgbd_video->depth_frame_mutable(other_frame_index)->SetGlobalTFrame(new_global_T_other_frame);
    rgbd_video->color_frame_mutable(other_frame_index)->SetGlobalTFrame(new_global_T_other_frame);
  }
}

// ---- frame -> keyframe --------------------------------------------------------------------------------------------------
shared_ptr<Keyframe> CreateKeyframeFromFrame(hipStream_t stream, const PreprocessConfig& config, DirectBA& direct_ba,
                                             RGBDVideo<Vec3u8, u16>& rgbd_video, int frame_index) {
  shared_ptr<Image<u16>> depth_image = rgbd_video.depth_frame_mutable(frame_index)->GetImage();
  shared_ptr<Image<Vec3u8>> rgb_image = rgbd_video.color_frame_mutable(frame_index)->GetImage();
  CHECK(depth_image && rgb_image) << "cannot load the images of frame " << frame_index;
  const int W = depth_image->width(), H = depth_image->height();
  const PinholeCamera4f depth_camera = direct_ba.depth_camera();
  const PinholeCamera4f color_camera = direct_ba.color_camera();
  const DepthParameters depth_params = direct_ba.depth_params();
  const float raw_to_float_depth = depth_params.raw_to_float_depth;
  // the kernels address the images with the cameras' dimensions
  CHECK(W == depth_camera.width() && H == depth_camera.height())
      << "frame " << frame_index << ": depth image is " << W << " x " << H << ", the depth camera " << depth_camera.width() << " x " << depth_camera.height();
  CHECK((int)rgb_image->width() == color_camera.width() && (int)rgb_image->height() == color_camera.height())
      << "frame " << frame_index << ": colour image is " << rgb_image->width() << " x " << rgb_image->height() << ", the colour camera "
      << color_camera.width() << " x " << color_camera.height();

  bahip_context* ctx = UtilityContext(stream);
  CUDABuffer<u16> depth_buffer(H, W), filtered_A(H, W), filtered_B(H, W), normals_buffer(H, W), radius_buffer(H, W);
  CUDABuffer<u8> rgb_buffer(rgb_image->height(), rgb_image->width() * 3);
  CUDABuffer<uchar4> color_buffer(rgb_image->height(), rgb_image->width());
  depth_buffer.UploadAsync(stream, *depth_image);
  rgb_buffer.UploadAsync(stream, reinterpret_cast<const u8*>(rgb_image->data()));
  // B/bad_slam.cc:691-706
  BAHIP_CHECKED_CALL(bahip_compute_brightness(ctx, rgb_buffer.ToCUDA().address(), (uint32_t)rgb_buffer.ToCUDA().pitch(),
                                              reinterpret_cast<uint8_t*>(color_buffer.ToCUDA().address()), (uint32_t)color_buffer.ToCUDA().pitch(),
                                              rgb_image->width(), rgb_image->height()));
  BAHIP_CHECKED_CALL(bahip_bilateral_filtering_and_depth_cutoff(
      ctx, config.bilateral_filter_sigma_xy, config.bilateral_filter_sigma_inv_depth, config.bilateral_filter_radius_factor,
      (uint16_t)(config.max_depth / raw_to_float_depth), raw_to_float_depth, depth_buffer.ToCUDA().address(),
      (uint32_t)depth_buffer.ToCUDA().pitch(), filtered_A.ToCUDA().address(), (uint32_t)filtered_A.ToCUDA().pitch(), W, H));
  // :716-722, :754-760
  const bahip_camera cam = ToBahipCamera(depth_camera);
  const bahip_depth_params dp = ToBahipDepthParams(depth_params);
  BAHIP_CHECKED_CALL(bahip_compute_normals(ctx, &cam, &dp, filtered_A.ToCUDA().address(), (uint32_t)filtered_A.ToCUDA().pitch(),
                                           filtered_B.ToCUDA().address(), (uint32_t)filtered_B.ToCUDA().pitch(),
                                           normals_buffer.ToCUDA().address(), (uint32_t)normals_buffer.ToCUDA().pitch()));
  radius_buffer.Clear(0, stream);
  BAHIP_CHECKED_CALL(bahip_compute_point_radii_and_remove_isolated_pixels(
      ctx, &cam, raw_to_float_depth, filtered_B.ToCUDA().address(), (uint32_t)filtered_B.ToCUDA().pitch(), radius_buffer.ToCUDA().address(),
      (uint32_t)radius_buffer.ToCUDA().pitch(), filtered_A.ToCUDA().address(), (uint32_t)filtered_A.ToCUDA().pitch()));
  // B/bad_slam.cc:976-1001
  float keyframe_min_depth = 0, keyframe_max_depth = 0;
  BAHIP_CHECKED_CALL(bahip_compute_min_max_depth(ctx, filtered_A.ToCUDA().address(), (uint32_t)filtered_A.ToCUDA().pitch(), W, H, raw_to_float_depth,
                                                 &keyframe_min_depth, &keyframe_max_depth));
  shared_ptr<Keyframe> keyframe(new Keyframe(stream, frame_index, keyframe_min_depth, keyframe_max_depth, filtered_A, normals_buffer,
                                             radius_buffer, color_buffer, rgbd_video.depth_frame(frame_index)->global_T_frame()));
  rgbd_video.depth_frame_mutable(frame_index)->ClearImageAndDerivedData();
  rgbd_video.color_frame_mutable(frame_index)->ClearImageAndDerivedData();
  return keyframe;
}

// ---- binary state -------------------------------------------------------------------------------------------------------
namespace {
constexpr u8 kStateVersion = 101;              // the reference writes 1 (B/io.cc:66); see rgbd_io.h
constexpr int kSurfelDataAttributeCount = 8;   // B/kernels.cuh:90: the rows that are state; the accumulators are not
constexpr int kPinholeCamera4fTypeInt = 1;     // L/camera.h:289

struct StateWriter {
  FILE* file;
  bool ok = true;
  void Bytes(const void* data, size_t size) { if (size && fwrite(data, 1, size, file) != size) ok = false; }
  void Int32(int value) { const i32 v = value; Bytes(&v, sizeof(v)); }
  void U32(u32 value) { Bytes(&value, sizeof(value)); }
  void Float(float value) { Bytes(&value, sizeof(value)); }
  void Bool(bool value) { const u8 v = value ? 1 : 0; Bytes(&v, 1); }
  void Pose(const SE3f& value) { Bytes(value.data(), 7 * sizeof(float)); }
};
struct StateReader {
  FILE* file;
  bool ok = true;
  void Bytes(void* data, size_t size) { if (size && fread(data, 1, size, file) != size) { ok = false; memset(data, 0, size); } }
  int Int32() { i32 v; Bytes(&v, sizeof(v)); return v; }
  u32 U32() { u32 v; Bytes(&v, sizeof(v)); return v; }
  float Float() { float v; Bytes(&v, sizeof(v)); return v; }
  bool Bool() { u8 v; Bytes(&v, 1); return v != 0; }
  SE3f Pose() { float v[7]; Bytes(v, sizeof(v)); SE3f pose; memcpy(pose.data(), v, sizeof(v)); return pose; }
};
struct FileCloser {
  FILE* file;
  ~FileCloser() { if (file) fclose(file); }
};
}  // namespace

bool SaveState(hipStream_t stream, const RGBDVideo<Vec3u8, u16>& rgbd_video, DirectBA& direct_ba, const std::string& path) {
  FILE* file = fopen(path.c_str(), "wb");
  if (!file) return false;
  FileCloser closer{file};
  StateWriter w{file};
  w.Bytes("BADSLAM", 7);
  w.Bytes(&kStateVersion, 1);

  // RGBDVideo (frame poses), B/io.cc:112-117.  The reference shares one pose object between a keyframe and its video
  // frame; here the keyframe owns its pose, and it is the one that counts.
  vector<const Keyframe*> keyframe_of_frame(rgbd_video.frame_count(), nullptr);
  for (const shared_ptr<Keyframe>& keyframe : direct_ba.keyframes())
    if (keyframe && keyframe->frame_index() < keyframe_of_frame.size()) keyframe_of_frame[keyframe->frame_index()] = keyframe.get();
  w.U32((u32)rgbd_video.frame_count());
  for (usize i = 0; i < rgbd_video.frame_count(); ++i)
    w.Pose(keyframe_of_frame[i] ? keyframe_of_frame[i]->global_T_frame() : rgbd_video.depth_frame(i)->global_T_frame());

  // Direct BA, B/io.cc:120-180
  const PinholeCamera4f color_camera = direct_ba.color_camera(), depth_camera = direct_ba.depth_camera();
  w.Int32(kPinholeCamera4fTypeInt); w.Int32(color_camera.width()); w.Int32(color_camera.height()); w.Int32(4);
  w.Bytes(color_camera.parameters(), 4 * sizeof(float));
  w.Int32(direct_ba.pyramid_level_for_color());
  w.Int32(kPinholeCamera4fTypeInt); w.Int32(depth_camera.width()); w.Int32(depth_camera.height()); w.Int32(4);
  w.Bytes(depth_camera.parameters(), 4 * sizeof(float));

  CUDABufferPtr<float> cfactor_buffer = direct_ba.cfactor_buffer();
  Image<float> cfactor_cpu(cfactor_buffer->width(), cfactor_buffer->height());
  cfactor_buffer->DownloadAsync(stream, &cfactor_cpu);
  BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
  w.Int32(cfactor_cpu.width()); w.Int32(cfactor_cpu.height()); w.Int32(cfactor_cpu.stride());
  w.Bytes(cfactor_cpu.data(), (size_t)cfactor_cpu.height() * cfactor_cpu.stride());

  const DepthParameters depth_params = direct_ba.depth_params();
  w.Float(depth_params.a); w.Float(depth_params.raw_to_float_depth); w.Float(depth_params.baseline_fx);
  w.Int32(depth_params.sparse_surfel_cell_size);

  w.Int32((int)direct_ba.keyframes().size());
  for (const shared_ptr<Keyframe>& keyframe : direct_ba.keyframes()) {
    w.Int32(keyframe ? keyframe->id() : -1);
    if (!keyframe) continue;
    w.Int32((int)keyframe->frame_index());
    w.Int32(static_cast<int>(keyframe->activation()));
    w.Int32(keyframe->last_active_in_ba_iteration());
    w.Int32(keyframe->last_covis_in_ba_iteration());
  }

  const u32 surfels_size = direct_ba.surfels_size();
  w.Int32((int)direct_ba.surfel_count());
  w.Int32((int)surfels_size);
  CUDABufferConstPtr<float> surfels = direct_ba.surfels();
  vector<float> surfel_data(surfels_size);
  for (int row = 0; row < kSurfelDataAttributeCount; ++row) {
    if (surfels_size) {
      surfels->DownloadPartAsync((size_t)row * surfels->ToCUDA().pitch(), surfels_size * sizeof(float), stream, surfel_data.data());
      BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
    }
    w.Bytes(surfel_data.data(), surfels_size * sizeof(float));
  }

  w.Int32(direct_ba.ba_iteration_count());
  w.Int32(direct_ba.last_ba_iteration_count());
  w.Bool(direct_ba.use_depth_residuals());
  w.Bool(direct_ba.use_descriptor_residuals());
  w.Int32(direct_ba.min_observation_count_while_bootstrapping_1());
  w.Int32(direct_ba.min_observation_count_while_bootstrapping_2());
  w.Int32(direct_ba.min_observation_count());
  w.Float(direct_ba.surfel_merge_dist_factor());
  return w.ok;
}

bool LoadState(hipStream_t stream, const PreprocessConfig& config, RGBDVideo<Vec3u8, u16>* rgbd_video, DirectBA* direct_ba,
               const std::string& path, std::function<bool(int, int)> progress_function) {
  FILE* file = fopen(path.c_str(), "rb");
  if (!file) return false;
  FileCloser closer{file};
  StateReader r{file};
  char identifier[7];
  u8 version = 0;
  r.Bytes(identifier, 7);
  r.Bytes(&version, 1);
  if (!r.ok || memcmp(identifier, "BADSLAM", 7) != 0) { LOG(ERROR) << "File identifier does not match."; return false; }
  if (version != kStateVersion) { LOG(ERROR) << "Unknown file format version."; return false; }

  // Everything that can be checked is checked before the objects are touched (the reference's TODO, B/io.cc:189-190):
  // the file is parsed into host memory first.
  const u32 frame_count = r.U32();
  if (!r.ok || frame_count != rgbd_video->frame_count()) {
    LOG(ERROR) << "Loaded frame count does not match the existing frame count in the dataset.";
    return false;
  }
  vector<SE3f> frame_poses(frame_count);
  for (u32 i = 0; i < frame_count; ++i) frame_poses[i] = r.Pose();

  struct CameraRecord { int type_int, width, height, parameter_count; float parameters[4]; } cameras[2];
  int pyramid_level_for_color = 0;
  for (int which = 0; which < 2; ++which) {   // colour camera, pyramid level, depth camera
    CameraRecord& c = cameras[which];
    c.type_int = r.Int32(); c.width = r.Int32(); c.height = r.Int32(); c.parameter_count = r.Int32();
    if (!r.ok || c.type_int != kPinholeCamera4fTypeInt || c.parameter_count != 4) {
      LOG(ERROR) << "Unexpected " << (which == 0 ? "color" : "depth") << " camera type or parameter count.";
      return false;
    }
    r.Bytes(c.parameters, sizeof(c.parameters));
    if (which == 0) pyramid_level_for_color = r.Int32();
    // The images on disk and the buffers of this DirectBA have the live cameras' size: a state with other dimensions cannot
    // be applied (it would abort later, in the keyframe constructor), and parameters that are not finite are not a camera.
    const PinholeCamera4f live = which == 0 ? direct_ba->color_camera() : direct_ba->depth_camera();
    if (!r.ok || c.width != (int)live.width() || c.height != (int)live.height()) {
      LOG(ERROR) << "Stored " << (which == 0 ? "color" : "depth") << " camera size does not match the existing camera.";
      return false;
    }
    for (float v : c.parameters)
      if (!std::isfinite(v)) { LOG(ERROR) << "Non-finite camera parameter."; return false; }
    if (!(c.parameters[0] > 0.f) || !(c.parameters[1] > 0.f)) { LOG(ERROR) << "Non-positive focal length."; return false; }
  }

  const int cfactor_width = r.Int32(), cfactor_height = r.Int32(), cfactor_stride = r.Int32();
  CUDABufferPtr<float> cfactor_buffer = direct_ba->cfactor_buffer();
  if (!r.ok || cfactor_width != cfactor_buffer->width() || cfactor_height != cfactor_buffer->height()) {
    LOG(ERROR) << "cfactor_buffer size does not match.";
    return false;
  }
  if (cfactor_stride < cfactor_width * (int)sizeof(float) || cfactor_stride > 100000 * (int)sizeof(float)) {
    LOG(ERROR) << "Implausible cfactor_buffer stride, refusing to load.";
    return false;
  }
  vector<u8> cfactor_bytes((size_t)cfactor_height * cfactor_stride);
  r.Bytes(cfactor_bytes.data(), cfactor_bytes.size());
  Image<float> cfactor_cpu(cfactor_width, cfactor_height);
  for (int y = 0; y < cfactor_height; ++y) memcpy(cfactor_cpu.row(y), cfactor_bytes.data() + (size_t)y * cfactor_stride, cfactor_width * sizeof(float));

  DepthParameters depth_params = direct_ba->depth_params();
  depth_params.a = r.Float();
  depth_params.raw_to_float_depth = r.Float();
  depth_params.baseline_fx = r.Float();
  depth_params.sparse_surfel_cell_size = r.Int32();
  // The surfel grid and the cfactor image are laid out for this DirectBA's cell size; depth scales must be usable numbers.
  if (!r.ok || depth_params.sparse_surfel_cell_size != direct_ba->depth_params().sparse_surfel_cell_size) {
    LOG(ERROR) << "sparse_surfel_cell_size does not match.";
    return false;
  }
  if (!std::isfinite(depth_params.a) || !std::isfinite(depth_params.raw_to_float_depth) || !(depth_params.raw_to_float_depth > 0.f) ||
      !std::isfinite(depth_params.baseline_fx) || !(depth_params.baseline_fx > 0.f)) {
    LOG(ERROR) << "Invalid depth parameters.";
    return false;
  }

  struct KeyframeRecord { int id, frame_index, activation, last_active_in_ba_iteration, last_covis_in_ba_iteration; };
  const int keyframe_count = r.Int32();
  if (!r.ok || keyframe_count < 0 || (usize)keyframe_count > rgbd_video->frame_count()) {
    LOG(ERROR) << "More keyframes than frames in the video.";
    return false;
  }
  vector<KeyframeRecord> keyframe_records(keyframe_count);
  for (int i = 0; i < keyframe_count; ++i) {
    KeyframeRecord& k = keyframe_records[i];
    k.id = r.Int32();
    if (k.id < 0) continue;
    if (k.id != i) { LOG(ERROR) << "Unexpected keyframe id."; return false; }
    k.frame_index = r.Int32(); k.activation = r.Int32(); k.last_active_in_ba_iteration = r.Int32(); k.last_covis_in_ba_iteration = r.Int32();
    if (!r.ok || k.frame_index < 0 || (usize)k.frame_index >= rgbd_video->frame_count() || k.activation < 0 || k.activation > 2) {
      LOG(ERROR) << "Invalid keyframe record.";
      return false;
    }
  }

  const int surfel_count = r.Int32(), surfels_size = r.Int32();
  CUDABufferPtr<float> surfels = direct_ba->surfels();
  if (!r.ok || surfel_count < 0 || surfels_size < surfel_count || surfels_size > surfels->width()) {
    LOG(ERROR) << "Invalid surfel count (or more surfels than this DirectBA was allocated for).";
    return false;
  }
  vector<vector<float>> surfel_rows(kSurfelDataAttributeCount, vector<float>(surfels_size));
  for (int row = 0; row < kSurfelDataAttributeCount; ++row) r.Bytes(surfel_rows[row].data(), (size_t)surfels_size * sizeof(float));

  const int ba_iteration_count = r.Int32(), last_ba_iteration_count = r.Int32();
  const bool use_depth_residuals = r.Bool(), use_descriptor_residuals = r.Bool();
  const int min_obs_bootstrapping_1 = r.Int32(), min_obs_bootstrapping_2 = r.Int32(), min_obs = r.Int32();
  const float surfel_merge_dist_factor = r.Float();
  if (!r.ok) { LOG(ERROR) << "Unexpected end of file."; return false; }

  // ---- apply ----  All or nothing: what the keyframe rebuild below needs to be in place already (poses, calibration) is
  // remembered first and put back if the rebuild is cancelled.
  struct Snapshot {
    vector<SE3f> color_poses, depth_poses;
    vector<shared_ptr<Keyframe>> keyframes;
    PinholeCamera4f color_camera, depth_camera;
    int pyramid_level;
    DepthParameters depth_params;
    Image<float> cfactor;
  } old{{}, {}, *direct_ba->keyframes_mutable(), direct_ba->color_camera(), direct_ba->depth_camera(), direct_ba->pyramid_level_for_color(),
        direct_ba->depth_params(), Image<float>(cfactor_width, cfactor_height)};
  old.color_poses.reserve(frame_count);
  old.depth_poses.reserve(frame_count);
  for (u32 i = 0; i < frame_count; ++i) {
    old.color_poses.push_back(rgbd_video->color_frame_mutable(i)->global_T_frame());
    old.depth_poses.push_back(rgbd_video->depth_frame_mutable(i)->global_T_frame());
  }
  cfactor_buffer->DownloadAsync(stream, &old.cfactor);
  BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
  auto roll_back = [&]() {
    for (u32 i = 0; i < frame_count; ++i) {
      rgbd_video->color_frame_mutable(i)->SetGlobalTFrame(old.color_poses[i]);
      rgbd_video->depth_frame_mutable(i)->SetGlobalTFrame(old.depth_poses[i]);
    }
    *direct_ba->keyframes_mutable() = old.keyframes;
    direct_ba->SetColorCamera(old.color_camera);
    direct_ba->SetPyramidLevelForColor(old.pyramid_level);
    direct_ba->SetDepthCamera(old.depth_camera);
    cfactor_buffer->UploadAsync(stream, old.cfactor);
    BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
    direct_ba->SetDepthParams(old.depth_params);
    direct_ba->SetCFactorBuffer(cfactor_buffer);
  };
  for (u32 i = 0; i < frame_count; ++i) {
    rgbd_video->color_frame_mutable(i)->SetGlobalTFrame(frame_poses[i]);
    rgbd_video->depth_frame_mutable(i)->SetGlobalTFrame(frame_poses[i]);
  }
  direct_ba->keyframes_mutable()->clear();
  direct_ba->SetColorCamera(PinholeCamera4f(cameras[0].width, cameras[0].height, cameras[0].parameters));
  direct_ba->SetPyramidLevelForColor(pyramid_level_for_color);
  direct_ba->SetDepthCamera(PinholeCamera4f(cameras[1].width, cameras[1].height, cameras[1].parameters));
  cfactor_buffer->UploadAsync(stream, cfactor_cpu);
  BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
  direct_ba->SetDepthParams(depth_params);
  direct_ba->SetCFactorBuffer(cfactor_buffer);   // SetDepthParams overwrote the device view kept inside depth_params

  // Keyframes are rebuilt from their video frames with the loaded calibration and poses (B/io.cc:405-445).
  for (int i = 0; i < keyframe_count; ++i) {
    if (progress_function && !progress_function(i, keyframe_count)) {
      roll_back();   // cancelled: the objects are as they were before the call
      return false;
    }
    const KeyframeRecord& k = keyframe_records[i];
    if (k.id < 0) { direct_ba->keyframes_mutable()->push_back(nullptr); continue; }
    shared_ptr<Keyframe> keyframe = CreateKeyframeFromFrame(stream, config, *direct_ba, *rgbd_video, k.frame_index);
    direct_ba->AddKeyframe(keyframe);
    CHECK_EQ(keyframe->id(), k.id);
  }
  // The stored activation states are applied once all keyframes are in: AddKeyframe wakes up inactive keyframes that are
  // co-visible with the one being added, so applying them keyframe by keyframe (B/io.cc:437-440) would not reproduce the
  // saved state.
  for (int i = 0; i < keyframe_count; ++i) {
    const KeyframeRecord& k = keyframe_records[i];
    if (k.id < 0) continue;
    Keyframe* keyframe = (*direct_ba->keyframes_mutable())[i].get();
    keyframe->SetActivation(static_cast<Keyframe::Activation>(k.activation));
    keyframe->SetLastActiveInBAIteration(k.last_active_in_ba_iteration);
    keyframe->SetLastCovisInBAIteration(k.last_covis_in_ba_iteration);
  }

  direct_ba->SetSurfelCount(surfel_count, surfels_size);
  for (int row = 0; row < kSurfelDataAttributeCount && surfels_size; ++row)
    surfels->UploadPartAsync((size_t)row * surfels->ToCUDA().pitch(), (size_t)surfels_size * sizeof(float), stream, surfel_rows[row].data());
  BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));

  direct_ba->SetBAIterationCount(ba_iteration_count);
  direct_ba->SetLastBAIterationCount(last_ba_iteration_count);
  direct_ba->SetUseDepthResiduals(use_depth_residuals);
  direct_ba->SetUseDescriptorResiduals(use_descriptor_residuals);
  direct_ba->SetMinObservationCountWhileBootstrapping1(min_obs_bootstrapping_1);
  direct_ba->SetMinObservationCountWhileBootstrapping2(min_obs_bootstrapping_2);
  direct_ba->SetMinObservationCount(min_obs);
  direct_ba->SetSurfelMergeDistFactor(surfel_merge_dist_factor);
  return true;
}

}  // namespace vis
