// ba_tum.cc -- direct bundle adjustment of a TUM-format RGB-D sequence with given initial poses.
//
// What BadSlam does around DirectBA for every keyframe (B/bad_slam.cc: PreprocessFrame, CreateKeyframe, AddKeyframe,
// RunBundleAdjustment), without the odometry front-end: the initial keyframe poses come from a trajectory file of the
// dataset (as with the reference's --import_poses).  Reads <dataset>/{calibration.txt, associated.txt, <trajectory>},
// makes every <interval>-th frame a keyframe, runs <iterations> BA calls of up to 10 iterations each with surfel
// updates, lets the frames in between follow their keyframes (trajectory deformation) and writes <out>.poses.txt (TUM
// trajectory lines of all frames, relative to the first), <out>.*_intrinsics.txt,
// <out>.deformation.txt and <out>.ply.
//
//   ba_tum <dataset_dir> <trajectory_file> <out_prefix> [--interval N] [--iterations N] [--cell N] [--max_depth M]
//          [--raw_to_float_depth S] [--pcg] [--intrinsics] [--incremental | --parallel_ba] [--save_state F] [--load_state F]
//          [--ba_call_iterations N] [--baseline_fx F] [--spatial_sort_cell C] [--row_major_creation]
//          [--bilateral_sigma_xy S] [--bilateral_sigma_inv_depth S] [--bilateral_radius_factor R]
//
// --ba_call_iterations N: every BA call runs exactly N iterations (min = max = N) instead of 1 .. 10; the remaining options set
// what B/bad_slam_config.h makes configurable (and the two order switches of this backend): together they let a TUM-format copy of
// a test scene go through exactly the chain of tests/e2e_vga.py (tests/test_gpu_tum_pipeline.py holds the result against the
// reference's own kernels).
//
// --incremental / --parallel_ba feed the keyframes one at a time through vis::BAScheduler (ba_scheduler.h), the way
// BadSlam::ProcessFrame does: each keyframe arrives with its pose relative to the previous keyframe (taken from the
// trajectory file), plans max_num_ba_iterations_per_keyframe iterations, and these run either right away (sequential) or
// on the BA thread while the next keyframe is being prepared.  --save_state writes the backend state after BA;
// --load_state starts from such a file instead of creating keyframes and surfels (checkpoint / resume, rgbd_io.h).
#include <cstdio>
#include <cstring>
#include <string>

#include "ba_scheduler.h"
#include "rgbd_io.h"

using namespace vis;

// ba_tum --check-dataset <dataset_dir> <trajectory_file>: parse and decode only (no GPU): prints what was read.
static int CheckDataset(const char* dataset, const char* trajectory) {
  RGBDVideo<Vec3u8, u16> video;
  if (!ReadTUMRGBDDatasetAssociatedAndCalibrated(dataset, (trajectory && *trajectory) ? trajectory : nullptr, &video)) return 1;
  const float* c = video.depth_camera()->parameters();
  printf("frames %zu width %d height %d camera %.9g %.9g %.9g %.9g\n", (size_t)video.frame_count(), video.depth_camera()->width(),
         video.depth_camera()->height(), c[0], c[1], c[2], c[3]);
  for (usize f = 0; f < video.frame_count(); ++f) {
    shared_ptr<Image<u16>> depth = video.depth_frame_mutable(f)->GetImage();
    shared_ptr<Image<Vec3u8>> color = video.color_frame_mutable(f)->GetImage();
    if (!depth || !color) return 1;
    unsigned long long depth_sum = 0, color_sum = 0;
    for (usize i = 0; i < (usize)depth->width() * depth->height(); ++i) depth_sum += depth->data()[i];
    for (usize i = 0; i < (usize)color->width() * color->height(); ++i) color_sum += color->data()[i].v[0] + 2 * color->data()[i].v[1] + 3 * color->data()[i].v[2];
    const float* v = video.depth_frame(f)->global_T_frame().data();
    printf("frame %zu %s %s depth_sum %llu color_sum %llu pose %.9g %.9g %.9g %.9g %.9g %.9g %.9g\n", (size_t)f,
           video.color_frame(f)->timestamp_string().c_str(), video.depth_frame(f)->timestamp_string().c_str(), depth_sum, color_sum, v[0], v[1],
           v[2], v[3], v[4], v[5], v[6]);
  }
  return 0;
}

// ba_tum --decode-png <file.png> <rgb|depth> <out.raw>: decodes one PNG the way the dataset reader does (8-bit colour / grey as
// RGB triples, 8- or 16-bit grey as 16-bit depth) and writes "width height\n" + the samples (u8 x 3 or native-endian u16).
// Exit code 3 = the decoder refused the file (unsupported format or malformed), 0 = decoded.
static int DecodePng(const char* path, const char* kind, const char* out_path) {
  FILE* out = nullptr;
  if (!strcmp(kind, "rgb")) {
    Image<Vec3u8> image;
    if (!ReadPNG(path, &image)) return 3;
    if (!(out = fopen(out_path, "wb"))) return 1;
    fprintf(out, "%d %d\n", image.width(), image.height());
    fwrite(image.data(), 3, (size_t)image.width() * image.height(), out);
  } else if (!strcmp(kind, "depth")) {
    Image<u16> image;
    if (!ReadPNG(path, &image)) return 3;
    if (!(out = fopen(out_path, "wb"))) return 1;
    fprintf(out, "%d %d\n", image.width(), image.height());
    fwrite(image.data(), 2, (size_t)image.width() * image.height(), out);
  } else {
    return 2;
  }
  fclose(out);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && !strcmp(argv[1], "--check-dataset")) return CheckDataset(argv[2], argc > 3 ? argv[3] : nullptr);
  if (argc == 5 && !strcmp(argv[1], "--decode-png")) return DecodePng(argv[2], argv[3], argv[4]);
  if (argc < 4) {
    fprintf(stderr, "usage: ba_tum <dataset_dir> <trajectory_file> <out_prefix> [--interval N] [--iterations N] [--cell N] [--max_depth M]"
                    " [--raw_to_float_depth S] [--pcg] [--intrinsics]\n");
    return 2;
  }
  const std::string dataset = argv[1], trajectory = argv[2], out = argv[3];
  int interval = 1, iterations = 10, cell = 4;
  float raw_to_float_depth = 1.0f / 5000;   // TUM RGB-D depth PNGs: 5000 units per metre
  float baseline_fx = 40.f, spatial_sort_cell = -1.f;
  int ba_call_iterations = 0;
  bool row_major_creation = false;
  bool use_pcg = false, intrinsics = false, incremental = false, parallel_ba = false;
  std::string save_state, load_state;
  PreprocessConfig config;
  for (int i = 4; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--interval" && i + 1 < argc) interval = atoi(argv[++i]);
    else if (a == "--iterations" && i + 1 < argc) iterations = atoi(argv[++i]);
    else if (a == "--cell" && i + 1 < argc) cell = atoi(argv[++i]);
    else if (a == "--max_depth" && i + 1 < argc) config.max_depth = (float)atof(argv[++i]);
    else if (a == "--raw_to_float_depth" && i + 1 < argc) raw_to_float_depth = (float)atof(argv[++i]);
    else if (a == "--ba_call_iterations" && i + 1 < argc) ba_call_iterations = atoi(argv[++i]);
    else if (a == "--baseline_fx" && i + 1 < argc) baseline_fx = (float)atof(argv[++i]);
    else if (a == "--spatial_sort_cell" && i + 1 < argc) spatial_sort_cell = (float)atof(argv[++i]);
    else if (a == "--row_major_creation") row_major_creation = true;
    else if (a == "--bilateral_sigma_xy" && i + 1 < argc) config.bilateral_filter_sigma_xy = (float)atof(argv[++i]);
    else if (a == "--bilateral_sigma_inv_depth" && i + 1 < argc) config.bilateral_filter_sigma_inv_depth = (float)atof(argv[++i]);
    else if (a == "--bilateral_radius_factor" && i + 1 < argc) config.bilateral_filter_radius_factor = (float)atof(argv[++i]);
    else if (a == "--pcg") use_pcg = true;
    else if (a == "--intrinsics") intrinsics = true;
    else if (a == "--incremental") incremental = true;
    else if (a == "--parallel_ba") incremental = parallel_ba = true;
    else if (a == "--save_state" && i + 1 < argc) save_state = argv[++i];
    else if (a == "--load_state" && i + 1 < argc) load_state = argv[++i];
    else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  if (bahip_device_count() <= 0) { fprintf(stderr, "no HIP device (there is no CPU fallback)\n"); return 99; }

  RGBDVideo<Vec3u8, u16> video;
  if (!ReadTUMRGBDDatasetAssociatedAndCalibrated(dataset.c_str(), trajectory.c_str(), &video)) return 1;
  printf("read %zu frames, %d x %d\n", (size_t)video.frame_count(), video.depth_camera()->width(), video.depth_camera()->height());
  if (video.frame_count() == 0) return 1;

  hipStream_t stream = nullptr;
  BAHIP_CHECKED_CALL(bahip_stream_create(&stream));
  {
    // B/bad_slam.cc:125-142 with the defaults of B/bad_slam_config.h
    DirectBA ba(/*max_surfel_count*/ 25 * 1000 * 1000, raw_to_float_depth, baseline_fx, cell, /*surfel_merge_dist_factor*/ 0.8f,
                /*min_observation_count_while_bootstrapping_1*/ 1, /*min_observation_count_while_bootstrapping_2*/ 2, /*min_observation_count*/ 2,
                *video.color_camera(), *video.depth_camera(), /*pyramid_level_for_color*/ 0, /*use_depth_residuals*/ true,
                /*use_descriptor_residuals*/ true, nullptr, SE3f());
    if (spatial_sort_cell >= 0.f) ba.SetSpatialSortCellSize(spatial_sort_cell);
    if (row_major_creation) ba.SetRowMajorCreation(true);
    vector<SE3f> original_keyframe_T_global;
    if (incremental) {
      BASchedulerConfig scheduler_config;
      scheduler_config.parallel_ba = parallel_ba;
      scheduler_config.use_pcg = use_pcg;
      scheduler_config.optimize_intrinsics = intrinsics;
      scheduler_config.max_num_ba_iterations_per_keyframe = iterations;
      BAScheduler scheduler(scheduler_config, &ba, &video, stream);
      SE3f previous_keyframe_pose;
      for (usize f = 0; f < video.frame_count(); f += interval) {
        // what an odometry front-end would hand over: the pose relative to the previous keyframe
        const SE3f initial_pose = video.depth_frame(f)->global_T_frame();
        const SE3f last_kf_tr_this_kf = (f == 0) ? SE3f() : previous_keyframe_pose.inverse() * initial_pose;
        previous_keyframe_pose = initial_pose;
        shared_ptr<Keyframe> kf = CreateKeyframeFromFrame(stream, config, ba, video, (int)f);
        const u32 newest_frame = (u32)std::min<usize>(f + interval - 1, video.frame_count() - 1);
        scheduler.SetLastFrameIndex((int)newest_frame);
        scheduler.AddKeyframe(kf, last_kf_tr_this_kf);
        scheduler.RunPlannedIterations(newest_frame);
      }
      scheduler.WaitForQueuedWork();
      scheduler.StopBAThreadAndWaitForIt();
      printf("%zu keyframes through the scheduler (%s), %d parallel iteration(s), %u surfels\n", ba.keyframes().size(),
             parallel_ba ? "BA thread" : "sequential", scheduler.parallel_iterations_done(), ba.surfel_count());
    } else {
      if (!load_state.empty()) {
        if (!LoadState(stream, config, &video, &ba, load_state)) { fprintf(stderr, "cannot load state %s\n", load_state.c_str()); return 1; }
        printf("loaded %s: %zu keyframes, %u surfels, BA iteration count %d\n", load_state.c_str(), ba.keyframes().size(), ba.surfel_count(),
               ba.ba_iteration_count());
      } else {
        for (usize f = 0; f < video.frame_count(); f += interval) ba.AddKeyframe(CreateKeyframeFromFrame(stream, config, ba, video, (int)f));
      }
      printf("%zu keyframes\n", ba.keyframes().size());
      RememberKeyframePoses(&ba, &original_keyframe_T_global);   // B/bad_slam.cc:1230 (before the BA that moves them)
      for (int i = 0; i < iterations; ++i) {
        int done = 0;
        bool converged = false;
        ba.BundleAdjustment(stream, /*optimize_depth_intrinsics*/ intrinsics, /*optimize_color_intrinsics*/ intrinsics, /*do_surfel_updates*/ true,
                            /*optimize_poses*/ true, /*optimize_geometry*/ true, /*min_iterations*/ ba_call_iterations > 0 ? ba_call_iterations : 1,
                            /*max_iterations*/ ba_call_iterations > 0 ? ba_call_iterations : 10, use_pcg, 0,
                            (int)ba.keyframes().size() - 1, /*increase_ba_iteration_count*/ true, &done, &converged);
        printf("BA call %d: %d iteration(s)%s, %u surfels\n", i + 1, done, converged ? ", converged" : "", ba.surfel_count());
      }
    }
    // keyframe poses back into the video (the reference shares the pose object between keyframe and video frame), then the
    // frames in between follow their keyframes (B/bad_slam.cc:1259-1269; the scheduler has done that after every BA)
    for (const shared_ptr<Keyframe>& kf : ba.keyframes()) {
      if (!kf) continue;
      video.depth_frame_mutable(kf->frame_index())->SetGlobalTFrame(kf->global_T_frame());
      video.color_frame_mutable(kf->frame_index())->SetGlobalTFrame(kf->global_T_frame());
    }
    if (!incremental) ExtrapolateAndInterpolateKeyframePoseChanges(0, (u32)video.frame_count() - 1, &ba, original_keyframe_T_global, &video);
    if (!save_state.empty()) {
      if (!SaveState(stream, video, ba, save_state)) { fprintf(stderr, "cannot write state %s\n", save_state.c_str()); return 1; }
      printf("wrote %s\n", save_state.c_str());
    }
    if (!SavePoses(video, /*use_depth_timestamps*/ true, /*start_frame*/ 0, out + ".poses.txt")) return 1;
    if (!SaveCalibration(stream, ba, out)) return 1;
    if (!SavePointCloudAsPLY(stream, ba, out + ".ply")) return 1;
    printf("wrote %s.{poses.txt,depth_intrinsics.txt,color_intrinsics.txt,deformation.txt,ply}\n", out.c_str());
  }
  bahip_stream_destroy(stream);
  return 0;
}
