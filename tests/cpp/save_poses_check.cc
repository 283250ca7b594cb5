// Reads a TUM-format dataset with the reference-named reader and writes its trajectory back with SavePoses
// (badslam_amd/host/rgbd_io.h; B/io.cc:537-568), relative to <start_frame>.  No GPU involved.  Used by tests/test_cpu_tum_io.py.
#include <cstdio>
#include <cstdlib>

#include "rgbd_io.h"

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  vis::RGBDVideo<vis::Vec3u8, vis::u16> video;
  if (!vis::ReadTUMRGBDDatasetAssociatedAndCalibrated(argv[1], argv[2], &video)) return 1;
  return vis::SavePoses(video, /*use_depth_timestamps*/ true, atoi(argv[4]), argv[3]) ? 0 : 1;
}
