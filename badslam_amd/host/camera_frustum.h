// camera_frustum.h -- keyframe frustum intersection for the co-visibility lists
// (libvis/src/libvis/camera_frustum.h:72-225, used by B/direct_ba.cc:231-249,710-738).
// Host-only, O(K^2) over keyframes: bounding-box pre-test, frustum planes vs. vertices both ways,
// then the separating-axis test over edge cross products (edges nearly parallel are skipped).
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>

#include "libvis_min.h"

namespace vis {

class CameraFrustum {
 public:
  CameraFrustum(const PinholeCamera4f& camera, float min_depth, float max_depth, const SE3f& global_T_camera) {
    float M[12];
    global_T_camera.matrix3x4(M);
    const float corners[4][2] = {{0.f, 0.f}, {(float)camera.width(), 0.f}, {0.f, (float)camera.height()},
                                 {(float)camera.width(), (float)camera.height()}};
    for (int c = 0; c < 4; ++c) {
      float d[3];
      camera.UnprojectFromPixelCornerConv(corners[c][0], corners[c][1], d);
      for (int far_plane = 0; far_plane < 2; ++far_plane) {
        const float z = far_plane ? max_depth : min_depth;
        const float p[3] = {z * d[0], z * d[1], z * d[2]};
        float* out = points_[2 * c + far_plane];
        for (int r = 0; r < 3; ++r) out[r] = M[4 * r + 0] * p[0] + M[4 * r + 1] * p[1] + M[4 * r + 2] * p[2] + M[4 * r + 3];
      }
    }
    for (int r = 0; r < 3; ++r) { bb_min_[r] = std::numeric_limits<float>::infinity(); bb_max_[r] = -bb_min_[r]; }
    for (int v = 0; v < 8; ++v)
      for (int r = 0; r < 3; ++r) { bb_min_[r] = std::min(bb_min_[r], points_[v][r]); bb_max_[r] = std::max(bb_max_[r], points_[v][r]); }
    ComputeAxesAndPlanes();
  }

  bool Intersects(const CameraFrustum& other) const {
    for (int r = 0; r < 3; ++r)
      if (std::max(bb_min_[r], other.bb_min_[r]) > std::min(bb_max_[r], other.bb_max_[r])) return false;
    if (SeparatedByPlanes(*this, other) || SeparatedByPlanes(other, *this)) return false;
    for (int a = 0; a < 6; ++a) {
      for (int b = 0; b < 6; ++b) {
        float dir[3];
        Cross(axes_[a], other.axes_[b], dir);
        if (dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2] < 1e-5f) continue;
        float tmin = std::numeric_limits<float>::infinity(), tmax = -tmin, omin = tmin, omax = -tmin;
        for (int v = 0; v < 8; ++v) {
          const float tv = Dot(dir, points_[v]), ov = Dot(dir, other.points_[v]);
          tmin = std::min(tmin, tv); tmax = std::max(tmax, tv);
          omin = std::min(omin, ov); omax = std::max(omax, ov);
        }
        if (tmax <= omin || tmin >= omax) return false;
      }
    }
    return true;
  }

 private:
  static float Dot(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
  static void Cross(const float* a, const float* b, float* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  }
  static void Sub(const float* a, const float* b, float* o) { for (int r = 0; r < 3; ++r) o[r] = a[r] - b[r]; }
  // true if all 8 vertices of `b` lie on the outer side of one plane of `a`
  static bool SeparatedByPlanes(const CameraFrustum& a, const CameraFrustum& b) {
    for (int p = 0; p < 6; ++p) {
      int v = 0;
      for (; v < 8; ++v)
        if (Dot(a.plane_n_[p], b.points_[v]) + a.plane_d_[p] < 0) break;
      if (v == 8) return true;
    }
    return false;
  }
  void SetPlane(int i, const float* n, float d) { for (int r = 0; r < 3; ++r) plane_n_[i][r] = n[r]; plane_d_[i] = d; }
  void ComputeAxesAndPlanes() {
    // vertex order: 0/1 top-left near/far, 2/3 top-right, 4/5 bottom-left, 6/7 bottom-right
    Sub(points_[7], points_[6], axes_[0]); Sub(points_[3], points_[2], axes_[1]);
    Sub(points_[5], points_[4], axes_[2]); Sub(points_[1], points_[0], axes_[3]);
    Sub(points_[2], points_[6], axes_[4]); Sub(points_[0], points_[2], axes_[5]);
    float n[3], neg[3];
    Cross(axes_[5], axes_[4], n);                                 // forward; normals point away from the frustum
    SetPlane(0, n, -Dot(n, points_[1]));                          // far
    for (int r = 0; r < 3; ++r) neg[r] = -n[r];
    SetPlane(1, neg, Dot(n, points_[0]));                         // near
    Cross(axes_[0], axes_[4], n); SetPlane(2, n, -Dot(n, points_[6]));   // right
    Cross(axes_[1], axes_[5], n); SetPlane(3, n, -Dot(n, points_[2]));   // top
    Cross(axes_[4], axes_[2], n); SetPlane(4, n, -Dot(n, points_[4]));   // left
    Cross(axes_[5], axes_[0], n); SetPlane(5, n, -Dot(n, points_[6]));   // bottom
  }

  float points_[8][3];
  float axes_[6][3];
  float plane_n_[6][3];
  float plane_d_[6];
  float bb_min_[3], bb_max_[3];
};

}  // namespace vis
