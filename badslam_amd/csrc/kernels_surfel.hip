// kernels_surfel.hip -- per-surfel stages of the alternating scheme: activation, normals, geometry.
//
// Reference structure (B/ = applications/badslam/src/badslam/): one kernel launch per keyframe
// per stage, every launch streaming all N surfels and doing read-modify-write on the accumulator
// rows of the surfel buffer (B/kernel_surfel_activation.cc:53-66, B/kernel_opt_geometry.cc:108-200).
// Here each stage is ONE launch: a thread owns a surfel, keeps position / normal / accumulators in
// registers and visits only the keyframes whose frustum can contain its wavefront's 64 surfels
// (wave_cull.h), so the surfel array is read once and written once per stage and both the K-fold
// RMW traffic and most of the K x N association tests disappear.  Keyframes are visited in
// ascending order, i.e. accumulation order equals the reference's sequence of launches.
#include "ba_device.h"
#include "ba_launch.h"
#include "wave_cull.h"

namespace bahip {

#ifndef BAHIP_WAVES_ATTR
#define BAHIP_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4)))   // cap the allocation at 128 VGPRs: 4 waves per SIMD (5 spills: measured slower)
#endif
constexpr int kSurfelBlock = 64;   // one wavefront per workgroup (per-wave work varies with the keyframe candidates)

__device__ __forceinline__ bool position_valid(Vec3 p) { return p.x == p.x; }   // deleted surfels carry NaN x

// B/kernel_surfel_activation.cu:38-94
__global__ void __launch_bounds__(kSurfelBlock) BAHIP_WAVES_ATTR
activation_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s, uint32_t surfels_size) {
  const uint32_t i = blockIdx.x * kSurfelBlock + threadIdx.x;
  const bool in_range = i < surfels_size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, in_range && position_valid(gp));
  bool active = false;
  for_each_candidate(
      num_kfs,
      [&](int k) { return kfs[k].activation == BAHIP_KF_ACTIVE && sphere_may_project(in, kfs[k].pose.F, wb); },
      [&](int k) {
        if (in_range && !active) {
          Assoc r;
          if (project_associate<false>(in, kfs[k].pose.F, kfs[k].geom, gp, gn, &r, nullptr)) active = true;
        }
      });
  if (in_range) s.active[i] = (s.active[i] & (uint8_t)~kSurfelActiveFlag) | (active ? kSurfelActiveFlag : 0);
}

// Normals pass: B/kernel_opt_geometry.cu:82-101 (reset), :527-553 (accumulate), :577-597 (update).
// `live` = this lane holds an active surfel; every lane of the wave must call this function.
__device__ __forceinline__ void normals_pass(const Intrinsics& in, const KfEntry* __restrict__ kfs, int num_kfs,
                                             const WaveBounds& wb, SurfelsView& s, uint32_t i, bool live, Vec3 gp,
                                             Vec3* gn_inout) {
  float sx = 0, sy = 0, sz = 0, count = 0;
  const Vec3 gn = *gn_inout;
  for_each_candidate(
      num_kfs,
      [&](int k) { return kfs[k].activation != BAHIP_KF_INACTIVE && sphere_may_project(in, kfs[k].pose.F, wb); },
      [&](int k) {
        if (!live) return;
        Assoc r;
        if (project_associate<false>(in, kfs[k].pose.F, kfs[k].geom, gp, gn, &r, nullptr)) {
          const Vec3 m = unpack_normal8(pitched_load(kfs[k].normals, kfs[k].normals_pitch, r.py, r.px));
          const Vec3 g = mul33(kfs[k].pose.GR, m);
          sx += g.x; sy += g.y; sz += g.z; count += 1.f;
        }
      });
  if (!live) return;
  // The reference leaves the sums in accum rows 0..3; keep that observable state.
  s.row(kSurfelAccum0 + 0)[i] = sx; s.row(kSurfelAccum0 + 1)[i] = sy;
  s.row(kSurfelAccum0 + 2)[i] = sz; s.row(kSurfelAccum0 + 3)[i] = count;
  if (count >= 1) {
    const uint32_t packed = pack_normal10((1.f / count) * mk3(sx, sy, sz));
    reinterpret_cast<uint32_t*>(s.row(kSurfelNormal))[i] = packed;
    *gn_inout = unpack_normal10(packed);
  }
}

__global__ void __launch_bounds__(kSurfelBlock) BAHIP_WAVES_ATTR
normals_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s) {
  const uint32_t i = blockIdx.x * kSurfelBlock + threadIdx.x;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const bool live = in_range && (s.active[ii] & kSurfelActiveFlag);
  const Vec3 gp = surfel_position(s, ii);
  Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, live && position_valid(gp));
  normals_pass(in, kfs, num_kfs, wb, s, ii, live, gp, &gn);
}

// Geometry step of one BA iteration for one surfel: normals, then either the depth-only 1x1 solve
// (B/kernel_opt_geometry.cu:417-508) or the joint position + descriptor 3x3 solve (:119-353).
template <bool kUseDepth, bool kUseDesc>
__global__ void __launch_bounds__(kSurfelBlock) BAHIP_WAVES_ATTR
geometry_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s) {
  const uint32_t i = blockIdx.x * kSurfelBlock + threadIdx.x;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const bool live = in_range && (s.active[ii] & kSurfelActiveFlag);
  const Vec3 gp = surfel_position(s, ii);
  Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, live && position_valid(gp));
  if (wb.r < 0.f) return;   // wave-uniform: no active surfel in this wavefront
  normals_pass(in, kfs, num_kfs, wb, s, ii, live, gp, &gn);

  auto cand = [&](int k) { return kfs[k].activation != BAHIP_KF_INACTIVE && sphere_may_project(in, kfs[k].pose.F, wb); };

  if (!kUseDesc) {
    float H = 0, b = 0;
    for_each_candidate(num_kfs, cand, [&](int k) {
      if (!live) return;
      Assoc r;
      if (!project_associate<false>(in, kfs[k].pose.F, kfs[k].geom, gp, gn, &r, nullptr)) return;
      const float inv_std = depth_inv_stddev(unp_nx(in, (float)r.px), unp_ny(in, (float)r.py), r.depth, r.nl, in.baseline_fx);
      const float jac = -inv_std;
      const Vec3 u = unproject(in, r.px, r.py, r.depth);
      const float raw = inv_std * dot3(r.nl, u - r.local);
      const float w = depth_residual_weight(raw);
      const float wj = w * jac;
      H += wj * jac;
      b += wj * raw;
    });
    if (!live) return;
    s.row(kSurfelAccum0 + 0)[i] = H;
    s.row(kSurfelAccum0 + 1)[i] = b;
    if (H > 1e-6f) {
      const float t = -1.f * b / H;
      const Vec3 np = gp + t * gn;
      s.row(kSurfelX)[i] = np.x; s.row(kSurfelY)[i] = np.y; s.row(kSurfelZ)[i] = np.z;
    }
    return;
  }

  const float radius_sq = s.row(kSurfelRadiusSquared)[ii];
  const float d1 = s.row(kSurfelDescriptor1)[ii];
  const float d2 = s.row(kSurfelDescriptor2)[ii];
  const TangentPoints tp = surfel_tangent_points(gp, gn, radius_sq);
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a5 = 0, a6 = 0, a7 = 0, a8 = 0;
  for_each_candidate(num_kfs, cand, [&](int k) {
    if (!live) return;
    const float* F = kfs[k].pose.F;
    Assoc r;
    if (!project_associate<false>(in, F, kfs[k].geom, gp, gn, &r, nullptr)) return;
    if (kUseDepth) {
      const float inv_std = depth_inv_stddev(unp_nx(in, (float)r.px), unp_ny(in, (float)r.py), r.depth, r.nl, in.baseline_fx);
      const float jac = -inv_std;
      const Vec3 u = unproject(in, r.px, r.py, r.depth);
      const float raw = inv_std * dot3(r.nl, u - r.local);
      const float w = depth_residual_weight(raw);
      a0 += w * jac * jac;
      a6 += w * raw * jac;
    }
    float cx, cy;
    if (depth_to_color_pixel(in, r.pxx, r.pxy, &cx, &cy)) {
      DescEval e;
      eval_descriptor<true>(in, kfs[k].lumafp, F, tp, cx, cy, d1, d2, &e);
      const float term1 = -in.cfx * (r.nl.x * r.local.z - r.nl.z * r.local.x);
      const float term2 = -in.cfy * (r.nl.y * r.local.z - r.nl.z * r.local.y);
      const float term3 = 1.f / (r.local.z * r.local.z);
      const float jp1 = -(e.gx1 * term1 + e.gy1 * term2) * term3;
      const float jp2 = -(e.gx2 * term1 + e.gy2 * term2) * term3;
      const float jd = -1.f;
      const float w1 = descriptor_residual_weight(e.r1);
      const float wr1 = w1 * e.r1;
      const float w2 = descriptor_residual_weight(e.r2);
      const float wr2 = w2 * e.r2;
      a0 += w1 * jp1 * jp1 + w2 * jp2 * jp2;
      a1 += w1 * jp1 * jd;
      a3 += w1 * jd * jd;
      a6 += wr1 * jp1 + wr2 * jp2;
      a7 += wr1 * jd;
      a2 += w2 * jp2 * jd;
      a5 += w2 * jd * jd;
      a8 += wr2 * jd;
    }
  });
  if (!live) return;
  s.row(kSurfelAccum0 + 0)[i] = a0; s.row(kSurfelAccum0 + 1)[i] = a1; s.row(kSurfelAccum0 + 2)[i] = a2;
  s.row(kSurfelAccum0 + 3)[i] = a3; s.row(kSurfelAccum0 + 4)[i] = 0;  s.row(kSurfelAccum0 + 5)[i] = a5;
  s.row(kSurfelAccum0 + 6)[i] = a6; s.row(kSurfelAccum0 + 7)[i] = a7; s.row(kSurfelAccum0 + 8)[i] = a8;

  // B/kernel_opt_geometry.cu:273-353: in-place Cholesky of the 3x3 system (H12 is exactly 0).
  float H00 = a0 + 1e-6f, H01 = a1, H02 = a2, H11 = a3 + 1e-6f, H12 = 0.f, H22 = a5 + 1e-6f;
  H00 = sqrtf(H00);
  H01 = H01 / H00;
  H11 = sqrtf(H11 - H01 * H01);
  H02 = H02 / H00;
  H12 = (H12 - H02 * H01) / H11;
  H22 = sqrtf(H22 - H02 * H02 - H12 * H12);
  const float y0 = a6 / H00;
  const float y1 = (a7 - H01 * y0) / H11;
  const float y2 = (a8 - H02 * y0 - H12 * y1) / H22;
  const float x2 = y2 / H22;
  const float x1 = (y1 - H12 * x2) / H11;
  const float x0 = (y0 - H02 * x2 - H01 * x1) / H00;
  if (x0 != 0) {
    const Vec3 np = gp - x0 * gn;
    s.row(kSurfelX)[i] = np.x; s.row(kSurfelY)[i] = np.y; s.row(kSurfelZ)[i] = np.z;
  }
  if (x1 != 0) s.row(kSurfelDescriptor1)[i] = fmaxf(-180.f, fminf(180.f, d1 - x1));
  if (x2 != 0) s.row(kSurfelDescriptor2)[i] = fmaxf(-180.f, fminf(180.f, d2 - x2));
}

// ---- launchers ---------------------------------------------------------------------------------
static inline unsigned grid_for(uint32_t n) { return (n + kSurfelBlock - 1) / kSurfelBlock; }

void launch_activation(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                       uint32_t surfels_size) {
  if (surfels_size == 0) return;
  hipLaunchKernelGGL(activation_kernel, dim3(grid_for(surfels_size)), dim3(kSurfelBlock), 0, stream, in, kfs, num_kfs, s,
                     surfels_size);
}

void launch_normals(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s) {
  if (s.size == 0) return;
  hipLaunchKernelGGL(normals_kernel, dim3(grid_for(s.size)), dim3(kSurfelBlock), 0, stream, in, kfs, num_kfs, s);
}

void launch_geometry(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs,
                     int num_kfs, const SurfelsView& s) {
  if (s.size == 0) return;
  const dim3 grid(grid_for(s.size)), block(kSurfelBlock);
  if (!use_desc) hipLaunchKernelGGL((geometry_kernel<true, false>), grid, block, 0, stream, in, kfs, num_kfs, s);
  else if (use_depth) hipLaunchKernelGGL((geometry_kernel<true, true>), grid, block, 0, stream, in, kfs, num_kfs, s);
  else hipLaunchKernelGGL((geometry_kernel<false, true>), grid, block, 0, stream, in, kfs, num_kfs, s);
}

}  // namespace bahip

// ---- diagnostics: how much work does a sweep over all keyframes contain? ----------------------------
// counts[0] = (wave, keyframe) candidates after frustum culling, [1] = of those with >= 1 associated
// lane, [2] = associated (surfel, keyframe) pairs, [3] = pairs that passed the in-image test.
namespace bahip {
__global__ void __launch_bounds__(kSurfelBlock) BAHIP_WAVES_ATTR
count_pairs_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s, unsigned long long* counts) {
  const uint32_t i = blockIdx.x * kSurfelBlock + threadIdx.x;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, in_range && position_valid(gp));
  unsigned long long cand = 0, wave_hits = 0, lane_hits = 0, in_image = 0;
  for_each_candidate(
      num_kfs, [&](int k) { return sphere_may_project(in, kfs[k].pose.F, wb); },
      [&](int k) {
        Assoc r;
        const bool hit = in_range && project_associate<false>(in, kfs[k].pose.F, kfs[k].geom, gp, gn, &r, nullptr);
        const Vec3 l = transform34(kfs[k].pose.F, gp);
        const float px = in.fx * (l.x / l.z) + in.cx, py = in.fy * (l.y / l.z) + in.cy;
        const bool inside = in_range && l.z > 0 && px >= 0 && py >= 0 && px < in.width && py < in.height;
        const unsigned long long m = __ballot(hit);
        cand += 1; wave_hits += (m != 0); lane_hits += __popcll(m); in_image += __popcll(__ballot(inside));
      });
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&counts[0], cand); atomicAdd(&counts[1], wave_hits); atomicAdd(&counts[2], lane_hits); atomicAdd(&counts[3], in_image);
  }
}
void launch_count_pairs(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                        unsigned long long* counts) {
  if (s.size) hipLaunchKernelGGL(count_pairs_kernel, dim3(grid_for(s.size)), dim3(kSurfelBlock), 0, stream, in, kfs, num_kfs, s, counts);
}
}  // namespace bahip
