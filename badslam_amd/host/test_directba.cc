// test_directba.cc -- the reference's closed-loop BA tests restated against this backend's C++
// surface (vis::DirectBA / vis::Keyframe / vis::CUDABuffer), used exactly the way
// applications/badslam/src/badslam/test/*.cc use the CUDA classes.
// Same scenes in structure, same start offsets, same pass tolerances; scene content comes from a
// portable PRNG (splitmix64) instead of glibc rand() / Eigen::Random.
// Usage: test_directba [test-name ...]; exit code = number of failed tests.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "direct_ba.h"

using namespace vis;

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
  int below(int n) { return (int)(next() % (uint64_t)n); }
  float uniform(float lo, float hi) { return lo + (hi - lo) * (float)((next() >> 11) * (1.0 / 9007199254740992.0)); }
};

constexpr int W = 640, H = 480;
const float kCam[4] = {0.5f * H, 0.5f * H, 0.5f * W - 0.5f, 0.5f * H - 0.5f};

int g_failures = 0;
#define EXPECT_TRUE(cond, ...) do { if (!(cond)) { ++failures; printf("    FAILED: %s  ", #cond); printf(__VA_ARGS__); printf("\n"); } } while (0)

SE3f Exp(float a, float b, float c, float d, float e, float f) { const float t[6] = {a, b, c, d, e, f}; return SE3f::exp(t); }

void MakeOffsets(float kt, float kr, SE3f out[13]) {
  out[0] = SE3f();
  int n = 1;
  for (int sign = 1; sign >= -1; sign -= 2)
    for (int i = 0; i < 6; ++i) {
      float t[6] = {0, 0, 0, 0, 0, 0};
      t[i] = sign * (i < 3 ? kt : kr);
      out[n++] = SE3f::exp(t);
    }
}

DirectBA* MakeBA(const PinholeCamera4f& camera, float raw_to_float_depth, int cell, bool depth, bool desc, int min_obs = 2,
                 int max_surfels = 1000 * 1000) {
  return new DirectBA(max_surfels, raw_to_float_depth, 40, cell, 0.8f, min_obs, min_obs, min_obs, camera, camera, 0, depth, desc, nullptr, SE3f());
}

// test_pose_optimization_geometric_residual.cc:50-178
int PoseOptimizationWithGeometricResidual() {
  int failures = 0;
  Rng rng(1);
  PinholeCamera4f camera(W, H, kCam);
  SE3f global_tr_frame;
  hipStream_t stream = nullptr;
  BAHIP_CHECKED_CALL(bahip_stream_create(&stream));
  constexpr float s = 1.f / 1000;
  std::unique_ptr<DirectBA> ba(MakeBA(camera, s, 1, true, false));
  Image<u16> depth(W, H);
  depth.SetTo(65535);
  for (int p = 0; p < 3; ++p) {
    float n[3] = {rng.uniform(-1, 1), rng.uniform(-1, 1), -1.f};
    const float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (float& v : n) v /= len;
    const int max_x = W - 10 - 1, min_x = 10;
    const int left = min_x + (max_x - min_x) * ((2 * p) / (2.0f * 3 - 1));
    const int right = min_x + (max_x - min_x) * ((2 * p + 1) / (2.0f * 3 - 1));
    for (int y = 10; y < H - 10; ++y)
      for (int x = left; x < right; ++x) {
        float d[3];
        camera.UnprojectFromPixelCenterConv((float)x, (float)y, d);
        const float z = -2.5f / (n[0] * d[0] + n[1] * d[1] + n[2] * d[2]);   // ray from the origin vs plane n.x + 2.5 = 0
        depth(x, y) = (u16)(z / s + 0.5f);
      }
  }
  Image<Vec3u8> color(W, H);
  shared_ptr<Keyframe> kf(new Keyframe(stream, 0, ba->depth_params(), ba->depth_camera(), depth, color, global_tr_frame));
  ba->AddKeyframe(kf);
  ba->CreateSurfelsForKeyframe(stream, false, kf);
  EXPECT_TRUE(ba->surfel_count() > 100000, "surfels %u", ba->surfel_count());
  SE3f offsets[13];
  MakeOffsets(0.005f, 0.001f, offsets);
  float worst = 0;
  for (int i = 0; i < 13; ++i) {
    SE3f estimate;
    ba->EstimateFramePose(stream, offsets[i] * global_tr_frame.inverse(), kf->depth_buffer(), kf->normals_buffer(), kf->color_texture(), &estimate, false);
    float err[6];
    (estimate.inverse() * global_tr_frame).log(err);
    for (float e : err) worst = std::max(worst, std::fabs(e));
  }
  EXPECT_TRUE(worst <= 1.1e-6f, "worst |log(T_est^-1 T_gt)| component %.3e", worst);
  printf("    worst component %.3e (tolerance 1.1e-6)\n", worst);
  ba.reset(); kf.reset();
  bahip_stream_destroy(stream);
  return failures;
}

// test_pose_optimization_photometric_residual.cc:50-185
int PoseOptimizationColorOnlyCues() {
  int failures = 0;
  Rng rng(2);
  PinholeCamera4f camera(W, H, kCam);
  const SE3f global_tr_frame = Exp(0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f);
  hipStream_t stream = nullptr;
  BAHIP_CHECKED_CALL(bahip_stream_create(&stream));
  constexpr float s = 1.f / 1000;
  std::unique_ptr<DirectBA> ba(MakeBA(camera, s, 1, false, true));
  Image<u16> depth(W, H);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) depth(x, y) = (x == 0 || y == 0 || x == W - 1 || y == H - 1) ? 65535 : (u16)(2 / s);
  Image<Vec3u8> color(W, H);
  for (int y = 0; y < H; ++y) {
    color(0, y) = Vec3u8(0, 0, 0);
    for (int x = 1; x < W; ++x) {
      const u8 i = (u8)rng.below(16);
      const Vec3u8 left = color(x - 1, y);
      const Vec3u8 top = (y > 0) ? color(x, y - 1) : Vec3u8(0, 0, 0);
      color(x, y) = Vec3u8((u8)((left.x() + top.x()) / 2 + i), (u8)((left.y() + top.y()) / 2 + i), (u8)((left.z() + top.z()) / 2 + i));
    }
  }
  shared_ptr<Keyframe> kf(new Keyframe(stream, 0, ba->depth_params(), ba->depth_camera(), depth, color, global_tr_frame));
  ba->AddKeyframe(kf);
  ba->CreateSurfelsForKeyframe(stream, false, kf);
  SE3f offsets[13];
  MakeOffsets(0.0005f, 0.001f, offsets);
  float worst = 0;
  for (int i = 0; i < 13; ++i) {
    SE3f estimate;
    ba->EstimateFramePose(stream, global_tr_frame * offsets[i], kf->depth_buffer(), kf->normals_buffer(), kf->color_texture(), &estimate, false);
    float err[6];
    (estimate.inverse() * global_tr_frame).log(err);
    for (float e : err) worst = std::max(worst, std::fabs(e));
  }
  EXPECT_TRUE(worst <= 8e-5f, "worst component %.3e", worst);
  printf("    worst component %.3e (tolerance 8e-5)\n", worst);
  ba.reset(); kf.reset();
  bahip_stream_destroy(stream);
  return failures;
}

// test_geometry_optimization_geometric_residual.cc:50-222
int GeometryOptimizationWithGeometricResidual(bool use_pcg) {
  int failures = 0;
  Rng rng(3);
  PinholeCamera4f camera(W, H, kCam);
  const SE3f global_tr_frame = Exp(0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f);
  hipStream_t stream = nullptr;
  BAHIP_CHECKED_CALL(bahip_stream_create(&stream));
  constexpr float s = 1.f / 5000;
  std::unique_ptr<DirectBA> ba(MakeBA(camera, s, 1, true, true, /*min_obs*/ 1));
  Image<u16> depth(W, H);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      depth(x, y) = (x == 0 || y == 0 || x == W - 1 || y == H - 1) ? 65535 : (u16)((1 + 0.01f * rng.below(100)) / s + 0.5f);
  Image<Vec3u8> color(W, H);
  shared_ptr<Keyframe> kf(new Keyframe(stream, 0, ba->depth_params(), ba->depth_camera(), depth, color, global_tr_frame));
  // normals forced onto the viewing ray, written in place like the reference does (:112-122)
  Image<u16> normals(W, H);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float d[3];
      camera.UnprojectFromPixelCenterConv((float)x, (float)y, d);
      const float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + 1.f);
      const float nx = -d[0] / len, ny = -d[1] / len;
      const i8 sx = (i8)(nx * 127 + ((nx > 0) ? 0.5f : -0.5f)), sy = (i8)(ny * 127 + ((ny > 0) ? 0.5f : -0.5f));   // B/util.cuh:121-135
      normals(x, y) = (u16)((u16)(u8)sx | ((u16)(u8)sy << 8));
    }
  const_cast<CUDABuffer<u16>*>(&kf->normals_buffer())->UploadAsync(nullptr, normals);
  kf->RefreshPlanes(nullptr);   // the images were written behind the keyframe's back
  ba->AddKeyframe(kf);
  ba->CreateSurfelsForKeyframe(stream, false, kf);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) depth(x, y) = (u16)(depth(x, y) + (u16)((0.0001f * rng.below(50)) / s));
  const_cast<CUDABuffer<u16>*>(&kf->depth_buffer())->UploadAsync(stream, depth);
  kf->RefreshPlanes(stream);
  for (int i = 0; i < 10; ++i)
    ba->BundleAdjustment(stream, false, false, false, false, true, 10, 10, use_pcg, 0, (int)ba->keyframes().size() - 1, true);
  const u32 n = ba->surfel_count();
  vector<float> sx(n), sy(n), sz(n);
  ba->surfels()->DownloadPartAsync(kSurfelX * ba->surfels()->ToCUDA().pitch(), n * sizeof(float), stream, sx.data());
  ba->surfels()->DownloadPartAsync(kSurfelY * ba->surfels()->ToCUDA().pitch(), n * sizeof(float), stream, sy.data());
  ba->surfels()->DownloadPartAsync(kSurfelZ * ba->surfels()->ToCUDA().pitch(), n * sizeof(float), stream, sz.data());
  float F[12];
  global_tr_frame.inverse().matrix3x4(F);
  int num_fails = 0, visible = 0;
  for (u32 i = 0; i < n; ++i) {
    const float cx = F[0] * sx[i] + F[1] * sy[i] + F[2] * sz[i] + F[3];
    const float cy = F[4] * sx[i] + F[5] * sy[i] + F[6] * sz[i] + F[7];
    const float cz = F[8] * sx[i] + F[9] * sy[i] + F[10] * sz[i] + F[11];
    if (!(cz > 0)) continue;
    const float px = kCam[0] * cx / cz + kCam[2], py = kCam[1] * cy / cz + kCam[3];
    if (px < 0 || py < 0 || px >= W || py >= H) continue;
    ++visible;
    const float expected_z = s * depth((int)px, (int)py);
    if (std::fabs(cz - expected_z) > 1e-4f) ++num_fails;
  }
  EXPECT_TRUE(visible > 250000, "visible %d of %u", visible, n);
  EXPECT_TRUE(num_fails == 0, "%d surfels further than 1e-4 from the measured depth", num_fails);
  printf("    %d surfels checked, %d failures (must be 0)\n", visible, num_fails);
  ba.reset(); kf.reset();
  bahip_stream_destroy(stream);
  return failures;
}

// test_geometry_optimization_photometric_residual.cc:126-285 (checks :45-117)
int GeometryOptimizationWithPhotometricResidual(bool use_pcg) {
  int failures = 0;
  Rng rng(4);
  PinholeCamera4f camera(W, H, kCam);
  const SE3f global_tr_frame_0 = Exp(0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f);
  hipStream_t stream = nullptr;
  BAHIP_CHECKED_CALL(bahip_stream_create(&stream));
  constexpr float s = 1.f / 5000;
  std::unique_ptr<DirectBA> ba(MakeBA(camera, s, 1, /*depth*/ false, /*desc*/ true, /*min_obs*/ 1));
  constexpr int kFrameOffsetPx = 100;
  constexpr float kDepth = 2.f;
  Image<u16> depth(W, H);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      depth(x, y) = (x < kFrameOffsetPx || y == 0 || x == W - 1 || y == H - 1) ? 65535 : (u16)((kDepth / s) + 0.5f);
  Image<Vec3u8> color(W, H);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const u8 i = (u8)((255 / 2.f) * (1.f + std::sin(0.15f * x + 0.5f * std::sin(0.25f * y))));
      color(x, y) = Vec3u8(i, i, i);
    }
  shared_ptr<Keyframe> kf0(new Keyframe(stream, 0, ba->depth_params(), ba->depth_camera(), depth, color, global_tr_frame_0));
  Image<u16> normals(W, H);   // fronto-parallel: ImageSpaceNormalToU16(0, 0) = 0
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) normals(x, y) = 0;
  const_cast<CUDABuffer<u16>*>(&kf0->normals_buffer())->UploadAsync(nullptr, normals);
  kf0->RefreshPlanes(nullptr);
  ba->AddKeyframe(kf0);
  // the second keyframe sees the same plane shifted by kFrameOffsetPx pixels, with depth noise (:223-234)
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      if (x == 0 || y == 0 || x >= W - kFrameOffsetPx || y == H - 1) {
        depth(x, y) = 65535;
      } else {
        depth(x, y) = (u16)(depth(x + kFrameOffsetPx, y) + 0.0001f * rng.below(100) / s);
        color(x, y) = color(x + kFrameOffsetPx, y);
      }
    }
  const float offset_x = kDepth * kFrameOffsetPx / kCam[0];
  const SE3f global_tr_frame_1 = global_tr_frame_0 * Exp(offset_x, 0, 0, 0, 0, 0);
  shared_ptr<Keyframe> kf1(new Keyframe(stream, 1, ba->depth_params(), ba->depth_camera(), depth, color, global_tr_frame_1));
  const_cast<CUDABuffer<u16>*>(&kf1->normals_buffer())->UploadAsync(nullptr, normals);
  kf1->RefreshPlanes(nullptr);
  ba->AddKeyframe(kf1);
  ba->CreateSurfelsForKeyframe(stream, false, kf1);
  ba->BundleAdjustment(stream, false, false, false, false, true, 60, 60, use_pcg, 0, (int)ba->keyframes().size() - 1, true);
  const u32 n = ba->surfel_count();
  vector<float> sx(n), sy(n), sz(n);
  ba->surfels()->DownloadPartAsync(kSurfelX * ba->surfels()->ToCUDA().pitch(), n * sizeof(float), stream, sx.data());
  ba->surfels()->DownloadPartAsync(kSurfelY * ba->surfels()->ToCUDA().pitch(), n * sizeof(float), stream, sy.data());
  ba->surfels()->DownloadPartAsync(kSurfelZ * ba->surfels()->ToCUDA().pitch(), n * sizeof(float), stream, sz.data());
  float F[12];
  global_tr_frame_1.inverse().matrix3x4(F);
  int num_fails = 0, num_correct = 0;
  for (u32 i = 0; i < n; ++i) {
    const float cx = F[0] * sx[i] + F[1] * sy[i] + F[2] * sz[i] + F[3];
    const float cy = F[4] * sx[i] + F[5] * sy[i] + F[6] * sz[i] + F[7];
    const float cz = F[8] * sx[i] + F[9] * sy[i] + F[10] * sz[i] + F[11];
    if (!(cz > 0)) continue;
    const float px = kCam[0] * cx / cz + kCam[2], py = kCam[1] * cy / cz + kCam[3];
    if (px < 0 || py < 0 || px >= W || py >= H) continue;
    if (std::fabs(kDepth - cz) > 1e-3f) ++num_fails; else ++num_correct;
  }
  EXPECT_TRUE(num_correct >= 100000, "only %d surfels within 1e-3 of the plane", num_correct);
  EXPECT_TRUE(num_fails <= 75000, "%d surfels further than 1e-3 from the plane", num_fails);
  printf("    %u surfels: %d within 1e-3 of the plane (>= 100000 required), %d not (<= 75000 allowed)\n", n, num_correct, num_fails);
  ba.reset(); kf0.reset(); kf1.reset();
  bahip_stream_destroy(stream);
  return failures;
}

struct TestCase { const char* name; std::function<int()> fn; };

}  // namespace

int main(int argc, char** argv) {
  const TestCase tests[] = {
      {"PoseOptimizationWithGeometricResidual", PoseOptimizationWithGeometricResidual},
      {"PoseOptimizationColorOnlyCues", PoseOptimizationColorOnlyCues},
      {"AlternatingGeometryOptimizationWithGeometricResidual", [] { return GeometryOptimizationWithGeometricResidual(false); }},
      {"PCGGeometryOptimizationWithGeometricResidual", [] { return GeometryOptimizationWithGeometricResidual(true); }},
      {"AlternatingGeometryOptimizationWithPhotometricResidual", [] { return GeometryOptimizationWithPhotometricResidual(false); }},
      {"PCGGeometryOptimizationWithPhotometricResidual", [] { return GeometryOptimizationWithPhotometricResidual(true); }},
  };
  if (bahip_device_count() <= 0) { printf("no HIP device: these tests need an MI355X\n"); return 99; }
  for (const TestCase& t : tests) {
    bool selected = argc <= 1;
    for (int i = 1; i < argc; ++i) selected = selected || (std::string(argv[i]) == t.name);
    if (!selected) continue;
    printf("[ RUN  ] %s\n", t.name);
    const int f = t.fn();
    printf("[ %s ] %s\n", f ? "FAIL" : " OK ", t.name);
    g_failures += f ? 1 : 0;
  }
  printf("%d test(s) failed\n", g_failures);
  return g_failures;
}
