// test_route_b.cc -- the Route-B shim (kernels_hip.cc: the reference's *CUDA free functions on the bahip_* C ABI) driven the way
// the reference's own DirectBA drives them, against Route A (vis::DirectBA of this repository) on the same input: surfel
// creation, activation + one geometry iteration must give the same bits, and the per-keyframe pose normal equations of the shim
// must be the ones whose Gauss-Newton step moves a perturbed keyframe back.
#include <cmath>
#include <cstdio>

#include "../direct_ba.h"
#include "badslam/kernels.h"

using namespace vis;

namespace {
constexpr int W = 320, H = 240, K = 3, CELL = 2;
constexpr float kRawToFloat = 1.f / 5000.f, kBaselineFx = 40.f;

// A textured wall z = 2.5 m seen by cameras that differ by a lateral shift only: rendered in closed form.
void Render(const float cam[4], float tx, float ty, Image<u16>* depth, Image<Vec3u8>* rgb) {
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const float d = 2.5f + 0.2f * std::sin(0.01f * x) * std::cos(0.013f * y);
      const float gx = (x - (cam[2] - 0.5f)) / cam[0] * d + tx, gy = (y - (cam[3] - 0.5f)) / cam[1] * d + ty;
      const bool border = x == 0 || y == 0 || x == W - 1 || y == H - 1;
      (*depth)(x, y) = border ? 65535 : (u16)(d / kRawToFloat + 0.5f);
      auto ch = [](float a, float b) { return (u8)(127.5f * (1.f + std::sin(30.f * a + 0.5f * std::sin(50.f * b)))); };
      (*rgb)(x, y) = Vec3u8(ch(gx, gy), ch(gy, d), ch(d, gx));
    }
}
int g_failures = 0;
#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failures; } } while (0)
}  // namespace

int main() {
  const float camp[4] = {0.5f * H, 0.5f * H, 0.5f * W - 0.5f, 0.5f * H - 0.5f};
  const PinholeCamera4f camera(W, H, camp);
  hipStream_t stream = nullptr;

  // ---- Route A: vis::DirectBA ----
  DirectBA ba(400000, kRawToFloat, kBaselineFx, CELL, 0.8f, 1, 1, 1, camera, camera, 0, true, true, nullptr, SE3f());
  ba.SetSpatialSortCellSize(0);   // Route B keeps its caller's buffers in the reference's order: Route A must not reorder its own (index-wise memcmp below)
  vector<shared_ptr<Keyframe>> keyframes;
  for (int k = 0; k < K; ++k) {
    Image<u16> depth(W, H); Image<Vec3u8> rgb(W, H);
    const float shift[7] = {0, 0, 0, 1, 0.15f * k, -0.1f * k, 0};
    Render(camp, shift[4], shift[5], &depth, &rgb);
    shared_ptr<Keyframe> kf(new Keyframe(stream, k, ba.depth_params(), ba.depth_camera(), depth, rgb, SE3f(shift)));
    ba.AddKeyframe(kf);
    keyframes.push_back(kf);
  }
  for (int k = 0; k < K; ++k) ba.CreateSurfelsForKeyframe(stream, /*filter_new_surfels*/ false, keyframes[k]);
  const u32 count_a = ba.surfels_size();
  vector<float> created_a(8 * (size_t)count_a);
  for (int r = 0; r < 8; ++r)
    ba.surfels()->DownloadPartAsync((size_t)r * ba.surfels()->ToCUDA().pitch(), count_a * sizeof(float), stream, &created_a[(size_t)r * count_a]);

  // ---- Route B: the reference's call sequence (B/direct_ba.cc:340-405) through the shim ----
  CUDABuffer<float> surfels(kSurfelAttributeCount, 400000);
  CUDABuffer<u8> active(1, 400000);
  CUDABuffer<u32> sup0(H, W), sup1(H, W), sup2(H, W);
  CUDABuffer<u32>* supporting[kMergeBufferCount] = {&sup0, &sup1, &sup2};
  CUDABuffer<u8> flags(H, W);
  CUDABuffer<u32> indices(H, W);
  void* temp = nullptr; usize temp_bytes = 0;
  const DepthParameters dp = ba.depth_params();
  u32 surfels_size = 0, surfel_count = 0;
  for (int k = 0; k < K; ++k) {
    float G[12], F[12];
    keyframes[k]->global_T_frame().matrix3x4(G);
    keyframes[k]->frame_T_global().matrix3x4(F);
    DetermineSupportingSurfelsCUDA(stream, camera, CUDAMatrix3x4(F), dp, keyframes[k]->depth_buffer(), keyframes[k]->normals_buffer(),
                                   surfels_size, &surfels, supporting);
    u32 created = 0;
    CreateSurfelsForKeyframeCUDA(stream, CELL, false, 1, k, keyframes, camera, camera, CUDAMatrix3x4(G), CUDAMatrix3x4(F), {}, dp,
                                 keyframes[k]->depth_buffer(), keyframes[k]->normals_buffer(), keyframes[k]->radius_buffer(),
                                 keyframes[k]->color_buffer(), keyframes[k]->color_texture(), supporting, &temp, &temp_bytes, &flags,
                                 &indices, surfels_size, surfel_count, &created, &surfels);
    surfels_size += created; surfel_count += created;
  }
  EXPECT(surfels_size == count_a && count_a > 20000);
  vector<float> created_b(8 * (size_t)surfels_size);
  for (int r = 0; r < 8; ++r)
    surfels.DownloadPartAsync((size_t)r * surfels.ToCUDA().pitch(), surfels_size * sizeof(float), stream, &created_b[(size_t)r * surfels_size]);
  EXPECT(surfels_size == count_a && memcmp(created_a.data(), created_b.data(), created_a.size() * sizeof(float)) == 0);

  // ---- one geometry-only BA iteration: Route A's BundleAdjustment vs the reference's loop body through the shim ----
  int done = 0;
  ba.BundleAdjustment(stream, false, false, false, /*optimize_poses*/ false, /*optimize_geometry*/ true, 1, 1, false, 0, K - 1,
                      /*increase_ba_iteration_count*/ false, &done);
  // (the end-of-scheme tasks ran first in that call, since the iteration counters differed: B/direct_ba_alternating.cc:330-343)
  active.Clear(0, stream);
  u32 count_b = surfel_count;
  DeleteSurfelsAndUpdateRadiiCUDA(stream, 1, camera, dp, keyframes, &count_b, surfels_size, &surfels, nullptr);
  CompactSurfelsCUDA(stream, &temp, &temp_bytes, count_b, &surfels_size, &surfels.ToCUDA(), nullptr);
  UpdateSurfelActivationCUDA(stream, camera, dp, keyframes, surfels_size, &surfels, &active);
  OptimizeGeometryIterationCUDA(stream, true, true, camera, camera, dp, keyframes, surfels_size, surfels, active);
  EXPECT(surfels_size == ba.surfels_size());
  vector<float> after_a(8 * (size_t)surfels_size), after_b(8 * (size_t)surfels_size);
  for (int r = 0; r < 8; ++r) {
    ba.surfels()->DownloadPartAsync((size_t)r * ba.surfels()->ToCUDA().pitch(), surfels_size * sizeof(float), stream, &after_a[(size_t)r * surfels_size]);
    surfels.DownloadPartAsync((size_t)r * surfels.ToCUDA().pitch(), surfels_size * sizeof(float), stream, &after_b[(size_t)r * surfels_size]);
  }
  EXPECT(memcmp(after_a.data(), after_b.data(), after_a.size() * sizeof(float)) == 0);

  // ---- pose normal equations of a shifted keyframe through the shim: the Gauss-Newton step points back ----
  float shifted[7];
  memcpy(shifted, keyframes[1]->global_T_frame().data(), sizeof(shifted));
  shifted[4] += 0.004f;                                         // 4 mm along x
  float F[12];
  SE3f(shifted).inverse().matrix3x4(F);
  float Hm[21], b[6];
  AccumulatePoseEstimationCoeffsCUDA(stream, true, true, camera, camera, dp, keyframes[1]->depth_buffer(), keyframes[1]->normals_buffer(),
                                     keyframes[1]->color_texture(), CUDAMatrix3x4(F), surfels_size, surfels, false, nullptr, nullptr, Hm, b,
                                     nullptr);
  EXPECT(Hm[0] > 0 && std::isfinite(b[0]));
  // x0 ~ b0 / H00 is the leading part of the step along the translation-x tangent: T <- T * exp(-x) must reduce the 4 mm offset
  const float x0 = b[0] / Hm[0];
  EXPECT(x0 > 0.001f && x0 < 0.008f);
  printf("route B: %u surfels created (same bits as Route A), geometry step identical, pose step %.4f m against a 0.004 m offset\n",
         count_a, x0);

  // ---- one outer iteration of the PCG scheme: Route A (fused sweeps, stopping rule on the device) vs the reference's driver
  // (B/direct_ba_pcg.cc:229-646: per-keyframe PCGInitCUDA / PCGStep1CUDA, host-side stopping rule) through the shim ----
  {
    const int gauge = 0;
    const u32 N = surfels_size;
    const u32 pose_unknowns = 6 * (K - 1), U = pose_unknowns + 3 * N, kInvalid = 0xffffffffu;
    active.Clear(1, stream);
    UpdateSurfelNormalsCUDA(stream, camera, dp, keyframes, N, surfels, active);
    // the vectors are as wide as the reference allocates them (max_unknown_count, B/direct_ba_pcg.cc:249-268: room for every
    // surfel the buffer can hold, 64 keyframes, the depth and colour intrinsics), not U wide
    const u32 Umax = 6 * 63 + 3 * (u32)surfels.width() + 5 + (u32)(dp.cfactor_buffer.width() * dp.cfactor_buffer.height()) + 4;
    CUDABuffer<PCGScalar> r(1, Umax), M(1, Umax), delta(1, Umax), g(1, Umax), p(1, Umax), alpha_n_buf(1, 1), alpha_d(1, 1), beta_n_buf(1, 1);
    CUDABuffer<PCGScalar>*alpha_n = &alpha_n_buf, *beta_n = &beta_n_buf;
    r.Clear(0, stream); M.Clear(0, stream);
    const float* c = camera.parameters();
    const PixelCornerProjector projector{c[0], c[1], c[2], c[3]};
    const PixelCenterUnprojector unprojector{1.f / c[0], 1.f / c[1], -(c[2] - 0.5f) / c[0], -(c[3] - 0.5f) / c[1]};
    const DepthToColorPixelCorner d2c{1.f, 1.f, 0.f, 0.f, W, H};
    auto projection = [&](int k) {
      float Fk[12];
      keyframes[k]->frame_T_global().matrix3x4(Fk);
      return SurfelProjectionParameters{surfels.ToCUDA(), keyframes[k]->depth_buffer().ToCUDA(), keyframes[k]->normals_buffer().ToCUDA(), dp,
                                        projector, unprojector, CUDAMatrix3x4(Fk), N};
    };
    auto pose_index = [&](int k) { return k == gauge ? kInvalid : (u32)(6 * (k < gauge ? k : k - 1)); };
    for (int k = 0; k < K; ++k)
      PCGInitCUDA(stream, projection(k), d2c, unprojector, projector, keyframes[k]->color_texture(), pose_index(k), pose_unknowns, k != gauge, true, true,
                  true, false, false, kInvalid, kInvalid, &r.ToCUDA(), &M.ToCUDA(), N);
    PCGInit2CUDA(stream, U, kInvalid, dp.a, r.ToCUDA(), M.ToCUDA(), &delta.ToCUDA(), &g.ToCUDA(), &p.ToCUDA(), &alpha_n->ToCUDA());
    double prev_r_norm = 1e300;
    int without_improvement = 0, steps = 0;
    for (int step = 0; step < 30; ++step) {
      ++steps;
      if (step > 0) { std::swap(alpha_n, beta_n); g.Clear(0, stream); }
      alpha_d.Clear(0, stream);
      for (int k = 0; k < K; ++k)
        PCGStep1CUDA(stream, U, projection(k), d2c, unprojector, projector, keyframes[k]->color_texture(), pose_index(k), pose_unknowns, k != gauge, true,
                     true, true, false, false, kInvalid, kInvalid, kInvalid, &p.ToCUDA(), &g.ToCUDA(), &alpha_d.ToCUDA(), N);
      PCGStep2CUDA(stream, U, kInvalid, r.ToCUDA(), M.ToCUDA(), &delta.ToCUDA(), &g.ToCUDA(), &p.ToCUDA(), &alpha_n->ToCUDA(), &alpha_d.ToCUDA(),
                   &beta_n->ToCUDA());
      PCGScalar r_norm = 0;
      beta_n->DownloadPartAsync(0, sizeof(PCGScalar), stream, &r_norm);
      BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
      r_norm = std::sqrt(r_norm);
      if (r_norm < prev_r_norm - 1e-3) without_improvement = 0;
      else if (++without_improvement >= 3) break;
      prev_r_norm = r_norm;
      if (step < 29) PCGStep3CUDA(stream, U, &g.ToCUDA(), &p.ToCUDA(), &alpha_n->ToCUDA(), &beta_n->ToCUDA());
    }
    UpdateSurfelsFromPCGDeltaCUDA(stream, N, &surfels.ToCUDA(), true, pose_unknowns, delta.ToCUDA());
    // Route A afterwards: its call also moves the keyframes (shared with the loop above, which left the poses alone)
    ba.SetPCGGaugeKeyframe(gauge);
    ba.BundleAdjustment(stream, false, false, false, /*optimize_poses*/ true, /*optimize_geometry*/ true, 1, 1, /*use_pcg*/ true, 0, K - 1,
                        /*increase_ba_iteration_count*/ false, &done);
    EXPECT(steps == ba.last_pcg_inner_steps());
    vector<float> pcg_a(8 * (size_t)N), pcg_b(8 * (size_t)N);
    for (int row = 0; row < 8; ++row) {
      ba.surfels()->DownloadPartAsync((size_t)row * ba.surfels()->ToCUDA().pitch(), N * sizeof(float), stream, &pcg_a[(size_t)row * N]);
      surfels.DownloadPartAsync((size_t)row * surfels.ToCUDA().pitch(), N * sizeof(float), stream, &pcg_b[(size_t)row * N]);
    }
    BAHIP_CHECKED_CALL(bahip_stream_synchronize(stream));
    EXPECT(memcmp(pcg_a.data(), pcg_b.data(), pcg_a.size() * sizeof(float)) == 0);
    EXPECT(memcmp(pcg_a.data(), after_a.data(), pcg_a.size() * sizeof(float)) != 0);   // the iteration moved the surfels
    printf("route B: PCG outer iteration through PCGInit / Init2 / Step1 / Step2 / Step3 / UpdateSurfelsFromPCGDelta: %d inner steps, surfels "
           "identical to Route A's fused iteration\n", steps);
  }
  if (g_failures == 0) printf("ROUTE_B_OK\n");
  return g_failures == 0 ? 0 : 1;
}
