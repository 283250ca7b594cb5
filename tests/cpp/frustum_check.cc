// CPU check of vis::CameraFrustum::Intersects (badslam_amd/host/camera_frustum.h) against brute force: points sampled on a
// pixel x depth grid of one frustum are tested for containment in the other (by projecting into its camera), both ways.
// Prints one line per pair: "<Intersects> <brute-force hit>".  Built and run by tests/test_cpu_camera_frustum.py.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "camera_frustum.h"

using namespace vis;

struct Rig {
  PinholeCamera4f camera;
  float min_depth, max_depth;
  SE3f global_T_camera;
};

static bool Contains(const Rig& r, const float p_global[3]) {
  float M[12];
  r.global_T_camera.inverse().matrix3x4(M);
  float l[3];
  for (int i = 0; i < 3; ++i) l[i] = M[4 * i] * p_global[0] + M[4 * i + 1] * p_global[1] + M[4 * i + 2] * p_global[2] + M[4 * i + 3];
  if (l[2] < r.min_depth || l[2] > r.max_depth) return false;
  const float* c = r.camera.parameters();
  const float x = c[0] * l[0] / l[2] + c[2], y = c[1] * l[1] / l[2] + c[3];   // pixel-corner convention
  return x >= 0 && y >= 0 && x <= r.camera.width() && y <= r.camera.height();
}

static bool BruteForce(const Rig& a, const Rig& b, int n) {
  float M[12];
  a.global_T_camera.matrix3x4(M);
  for (int iz = 0; iz <= n; ++iz) {
    const float z = a.min_depth + (a.max_depth - a.min_depth) * iz / n;
    for (int iy = 0; iy <= n; ++iy) {
      for (int ix = 0; ix <= n; ++ix) {
        float d[3];
        a.camera.UnprojectFromPixelCornerConv(a.camera.width() * (float)ix / n, a.camera.height() * (float)iy / n, d);
        const float l[3] = {z * d[0], z * d[1], z * d[2]};
        float g[3];
        for (int i = 0; i < 3; ++i) g[i] = M[4 * i] * l[0] + M[4 * i + 1] * l[1] + M[4 * i + 2] * l[2] + M[4 * i + 3];
        if (Contains(b, g)) return true;
      }
    }
  }
  return false;
}

int main(int argc, char** argv) {
  const int pairs = argc > 1 ? atoi(argv[1]) : 1000;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  const float params[4] = {520.f, 525.f, 319.5f, 240.5f};
  for (int i = 0; i < pairs; ++i) {
    Rig r[2];
    for (Rig& rig : r) {
      rig.camera = PinholeCamera4f(640, 480, params);
      rig.min_depth = 0.4f + 0.3f * (u(rng) + 1);
      rig.max_depth = rig.min_depth + 0.5f + 1.5f * (u(rng) + 1);
      const float spread = (i % 3 == 0) ? 1.0f : ((i % 3 == 1) ? 3.0f : 6.0f);   // mostly overlapping ... mostly disjoint
      const float tangent[6] = {spread * u(rng), spread * u(rng), spread * u(rng), 1.5f * u(rng), 1.5f * u(rng), 1.5f * u(rng)};
      rig.global_T_camera = SE3f::exp(tangent);
    }
    const CameraFrustum fa(r[0].camera, r[0].min_depth, r[0].max_depth, r[0].global_T_camera);
    const CameraFrustum fb(r[1].camera, r[1].min_depth, r[1].max_depth, r[1].global_T_camera);
    const bool sat = fa.Intersects(fb), sat_reverse = fb.Intersects(fa);
    const bool brute = BruteForce(r[0], r[1], 24) || BruteForce(r[1], r[0], 24);
    printf("%d %d %d\n", sat ? 1 : 0, sat_reverse ? 1 : 0, brute ? 1 : 0);
  }
  return 0;
}
