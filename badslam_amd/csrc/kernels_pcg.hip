// kernels_pcg.hip -- matrix-free preconditioned conjugate gradients on the full Gauss-Newton system.
//
// Reference (B/ = applications/badslam/src/badslam/): B/kernel_pcg.cu:179-1389 driven by
// B/direct_ba_pcg.cc:229-646.  Unknowns: [6 per non-gauge keyframe | 1 or 3 per surfel | 5 + S depth
// intrinsics | 4 colour intrinsics].  PCGInit and PCGStep1 are launched once per keyframe over all
// surfels there, with 12..46 serial block reductions per block and read-modify-write of the
// surfel entries per keyframe.
//
// Here both are single launches: a thread owns a surfel and sweeps the keyframes that survive the
// wave64 frustum test (wave_cull.h).  Surfel entries of r / M / g are accumulated in registers in
// keyframe order (identical to the reference's launch order) and written once.  Entries of the
// global intrinsics blocks are keyframe-independent, so each lane sums them over the whole sweep
// and the wave reduces once.  Only the 6 pose entries need a wave reduction + atomics per visited
// keyframe, and the per-cell cfactor entries an atomic per associated pair.
#include "ba_device.h"
#include "ba_launch.h"
#include "wave_cull.h"

namespace bahip {

constexpr int kPcgBlock = 256;     // per-unknown vector kernels
constexpr int kPcgSweepBlock = 64; // surfel sweeps (init, step 1): one wavefront per workgroup, like kernels_surfel.hip
#define BAHIP_PCG_SWEEP_ATTR __attribute__((amdgpu_waves_per_eu(4)))   // 128-VGPR cap: 4 waves per SIMD
constexpr float kDiagEpsilon = 1e-8f;   // B/kernel_pcg.cu:44
constexpr float kAPriorWeight = 10.f;   // B/kernel_pcg.cu:48

__device__ __forceinline__ uint32_t kf_pose_index(const PcgLayout& L, int k) {
  // B/direct_ba_pcg.cc:329-337
  if (k == L.gauge) return 0xffffffffu;
  return (k < L.gauge) ? 6u * (uint32_t)k : 6u * (uint32_t)(k - 1);
}
// Weight of unknown u in a dot product that is summed over the ranks (PcgLayout::head_scale).
__device__ __forceinline__ float dot_weight(const PcgLayout& L, uint32_t u) {
  const bool local = L.optimize_geometry && u >= L.surfel_start && u < L.surfel_end;
  return local ? 1.f : L.head_scale;
}
__device__ __forceinline__ float prior_at(const PcgLayout& L, uint32_t u) {
  return (u == L.a_index) ? (kAPriorWeight * kAPriorWeight) : 0.f;
}

// Terms of one associated pair (B/kernel_pcg.cu:213-303,334-395 and :663-748,786-905).
struct PairTerms {
  float raw, w, Jgeom;
  float Jpose[6];
  bool di_valid;
  float Jdi[5], Jcf;
  uint32_t cf_index;
  bool color_ok;
  float raw1, raw2, w1, w2, Jg1, Jg2;
  float Jp1[6], Jp2[6];
  float Jci1[4], Jci2[4];
};

template <bool kDepthIntr, bool kColorIntr>
__device__ __forceinline__ void eval_pair_terms(const PcgLayout& L, const Intrinsics& in, const KfEntry& kf, const Assoc& r,
                                                Vec3 gn, const TangentPoints& tp, float d1, float d2, PairTerms* t) {
  const float* F = kf.pose.F;
  const Vec3 rn = r.nl;
  const float nx = unp_nx(in, (float)r.px), ny = unp_ny(in, (float)r.py);
  t->di_valid = false;
  t->color_ok = false;
  // All Jacobians come from the jac_* functions of ba_device.h -- the ones the alternating sweeps use and the ones checked
  // against the golden vectors derived from the reference's own script (tests/golden/jacobians.json).
  if (L.use_depth) {
    const float inv_std = depth_inv_stddev(nx, ny, r.depth, rn, in.baseline_fx);
    const Vec3 u = unproject(in, r.px, r.py, r.depth);
    t->raw = inv_std * dot3(rn, u - r.local);
    t->w = depth_residual_weight(t->raw);
    t->Jgeom = -inv_std;
    jac_depth_pose(rn, u, inv_std, t->Jpose);
    if (kDepthIntr) {
      const int sparse_px = r.px / in.cell, sparse_py = r.py / in.cell;
      const float cfactor = pitched_load(in.cfactor, in.cfactor_pitch, sparse_py, sparse_px);
      const float raw_inv_depth = 1.0f / (in.raw_to_float_depth * pitched_load(kf.depth, kf.depth_pitch, r.py, r.px));
      const float exp_inv_depth = expf(-in.a * raw_inv_depth);
      const float corrected = cfactor * exp_inv_depth + raw_inv_depth;
      t->di_valid = !(fabsf(corrected) < 1e-4f);
      const float dot = dot3(mk3(nx, ny, 1), rn);
      float Jdi[6];   // fx_inv, fy_inv, cx_inv, cy_inv, a, cfactor (B/kernel_opt_intrinsics.cu:107-140 = B/kernel_pcg.cu:258-303)
      jac_depth_intrinsics(r.px, r.py, r.depth, inv_std, dot3(gn, mk3(F[0], F[1], F[2])), dot3(gn, mk3(F[4], F[5], F[6])), dot, cfactor,
                           raw_inv_depth, exp_inv_depth, corrected, Jdi);
#pragma unroll
      for (int c = 0; c < 5; ++c) t->Jdi[c] = Jdi[c];
      t->Jcf = Jdi[5];
      t->cf_index = L.depth_intr_start + 5 + sparse_px + sparse_py * in.cf_width;
    }
  }
  if (L.use_desc) {
    float cx, cy;
    t->color_ok = depth_to_color_pixel(in, r.pxx, r.pxy, &cx, &cy);
    if (t->color_ok) {
      DescEval e;
      eval_descriptor<true>(in, kf.lumafp, F, tp, cx, cy, d1, d2, &e);
      t->raw1 = e.r1; t->raw2 = e.r2;
      t->w1 = descriptor_residual_weight(e.r1);
      t->w2 = descriptor_residual_weight(e.r2);
      t->Jg1 = jac_descriptor_surfel(rn, r.local, r.inv_z, e.gx1, e.gy1, in.cfx, in.cfy);
      t->Jg2 = jac_descriptor_surfel(rn, r.local, r.inv_z, e.gx2, e.gy2, in.cfx, in.cfy);
      jac_descriptor_pose(r.local, r.inv_z, e.gx1 * in.cfx, e.gy1 * in.cfy, t->Jp1);
      jac_descriptor_pose(r.local, r.inv_z, e.gx2 * in.cfx, e.gy2 * in.cfy, t->Jp2);
      if (kColorIntr) {
        jac_descriptor_color_intrinsics(e.gx1, e.gy1, nx, ny, t->Jci1);
        jac_descriptor_color_intrinsics(e.gx2, e.gy2, nx, ny, t->Jci2);
      }
    }
  }
}

// ---- PCGInit: r -= J^T W F, M += diag(J^T W J)  (B/kernel_pcg.cu:179-541) -----------------------------
template <bool kDepthIntr, bool kColorIntr>
__global__ void __launch_bounds__(kPcgSweepBlock) BAHIP_PCG_SWEEP_ATTR
pcg_init_kernel(PcgLayout L, Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s,
                float* __restrict__ r_, float* __restrict__ M_) {
  const uint32_t i = xcd_chunked_tile(blockIdx.x) * kPcgSweepBlock + threadIdx.x;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const float radius_sq = s.row(kSurfelRadiusSquared)[ii];
  const float d1 = s.row(kSurfelDescriptor1)[ii], d2 = s.row(kSurfelDescriptor2)[ii];
  const TangentPoints tp = surfel_tangent_points(gp, gn, radius_sq);   // per surfel, not per pair
  const WaveBounds wb = wave_bounds(gp, in_range && (gp.x == gp.x));
  const int lane = threadIdx.x & 63;
  float gr[3] = {0, 0, 0}, gM[3] = {0, 0, 0};     // surfel entries
  float ir[9], iM[9];                             // 5 depth + 4 colour global intrinsics entries
#pragma unroll
  for (int q = 0; q < 9; ++q) { ir[q] = 0.f; iM[q] = 0.f; }

  for_each_candidate(
      num_kfs, [&](int k) { return sphere_may_project_item(in, kfs[k].pose.F, wb); },
      [&](int k) {
        const KfEntry& kf = kfs[k];
        Assoc a;
        bool visible = in_range && project_associate<false>(in, kf.pose.F, kf.geom, gp, gn, &a, nullptr);
        if (!__any(visible)) return;
        const bool pose_kf = L.optimize_poses && (k != L.gauge);
        float pr[6] = {0, 0, 0, 0, 0, 0}, pM[6] = {0, 0, 0, 0, 0, 0};
        if (visible) {
          PairTerms t;
          eval_pair_terms<kDepthIntr, kColorIntr>(L, in, kf, a, gn, tp, d1, d2, &t);
          if (L.use_depth) {
            if (L.optimize_geometry) {
              gr[0] -= t.Jgeom * t.w * t.raw;
              gM[0] += t.Jgeom * t.w * t.Jgeom;
            }
            if (pose_kf) {
#pragma unroll
              for (int c = 0; c < 6; ++c) { const float wj = t.w * t.Jpose[c]; pr[c] += -1 * wj * t.raw; pM[c] += t.Jpose[c] * wj; }
            }
            if (kDepthIntr) {
              if (!t.di_valid) visible = false;   // B/kernel_pcg.cu:272-274: also hides the descriptor part
              if (visible) {
#pragma unroll
                for (int c = 0; c < 5; ++c) { const float wj = t.w * t.Jdi[c]; ir[c] += -1 * wj * t.raw; iM[c] += t.Jdi[c] * wj; }
                const float wj = t.w * t.Jcf;
                unsafeAtomicAdd(&r_[t.cf_index], -1 * wj * t.raw);
                unsafeAtomicAdd(&M_[t.cf_index], t.Jcf * wj);
              }
            }
          }
          if (L.use_desc && visible && t.color_ok) {
            if (L.optimize_geometry) {
              gr[0] -= t.Jg1 * t.w1 * t.raw1 + t.Jg2 * t.w2 * t.raw2;
              gM[0] += t.Jg1 * t.w1 * t.Jg1 + t.Jg2 * t.w2 * t.Jg2;
              gr[1] -= -1.f * t.w1 * t.raw1 + 0.f * t.w2 * t.raw2;
              gM[1] += -1.f * t.w1 * -1.f + 0.f * t.w2 * 0.f;
              gr[2] -= 0.f * t.w1 * t.raw1 + -1.f * t.w2 * t.raw2;
              gM[2] += 0.f * t.w1 * 0.f + -1.f * t.w2 * -1.f;
            }
            if (pose_kf) {
#pragma unroll
              for (int c = 0; c < 6; ++c) {
                const float wj1 = t.w1 * t.Jp1[c], wj2 = t.w2 * t.Jp2[c];
                pr[c] += -1 * wj1 * t.raw1 + -1 * wj2 * t.raw2;
                pM[c] += t.Jp1[c] * wj1 + t.Jp2[c] * wj2;
              }
            }
            if (kColorIntr) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const float wj1 = t.w1 * t.Jci1[c], wj2 = t.w2 * t.Jci2[c];
                ir[5 + c] += -1 * wj1 * t.raw1 + -1 * wj2 * t.raw2;
                iM[5 + c] += t.Jci1[c] * wj1 + t.Jci2[c] * wj2;
              }
            }
          }
        }
        if (pose_kf) {
          const uint32_t base = kf_pose_index(L, k);
          // 12 totals with one halving butterfly (wave_reduce.h): lane 4 j holds total j
          const float v[16] = {pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pM[0], pM[1], pM[2], pM[3], pM[4], pM[5], 0.f, 0.f, 0.f, 0.f};
          const float mine = wave_reduce_small<16>(v, lane);
          const int slot = lane >> 2;
          if ((lane & 3) == 0 && slot < 12) unsafeAtomicAdd(slot < 6 ? &r_[base + slot] : &M_[base + slot - 6], mine);
        }
      });

  if (in_range && L.optimize_geometry) {
    const uint32_t gi = L.surfel_start + (uint32_t)L.geom_stride * i;
    r_[gi] = gr[0]; M_[gi] = gM[0];
    if (L.geom_stride == 3) { r_[gi + 1] = gr[1]; M_[gi + 1] = gM[1]; r_[gi + 2] = gr[2]; M_[gi + 2] = gM[2]; }
  }
  if (kDepthIntr || kColorIntr) {
    float mine = 0.f;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const float vr = wave_sum(ir[q]), vm = wave_sum(iM[q]);
      if (lane == q) mine = vr;
      if (lane == 9 + q) mine = vm;
    }
    if (lane < 18 && mine != 0.f) {
      const int q = lane % 9;
      const bool is_depth = q < 5;
      if ((is_depth && kDepthIntr) || (!is_depth && kColorIntr)) {
        const uint32_t u = is_depth ? (L.depth_intr_start + q) : (L.color_intr_start + (q - 5));
        unsafeAtomicAdd(lane < 9 ? &r_[u] : &M_[u], mine);
      }
    }
  }
}

// ---- inner-loop control on the device -------------------------------------------------------------------------------------
// The reference reads beta_n back after every inner step and decides on the host whether the residual norm still improves
// (B/direct_ba_pcg.cc:427-456: stop after three steps without an improvement of 1e-3).  Here a one-thread kernel takes that
// decision after step 2; once `stop` is set the sweeps and vector kernels queued behind it return at once, so the host can
// queue several inner steps without waiting for any of them.
struct PcgControl {
  float prev_r_norm;
  int no_improvement;
  int stop;
  int steps;
};
__global__ void pcg_control_kernel(PcgControl* ctl, const float* beta_n) {
  if (ctl->stop) return;
  ctl->steps += 1;
  const float r_norm = sqrtf(*beta_n);
  if (r_norm < ctl->prev_r_norm - 1e-3f) ctl->no_improvement = 0;
  else if (++ctl->no_improvement >= 3) ctl->stop = 1;
  ctl->prev_r_norm = r_norm;
}
__global__ void pcg_control_init_kernel(PcgControl* ctl) {
  ctl->prev_r_norm = __builtin_huge_valf(); ctl->no_improvement = 0; ctl->stop = 0; ctl->steps = 0;
}

// ---- block-level scalar reduction helper ------------------------------------------------------------------
// One atomic per workgroup of kPcgBlock threads.  The vector kernels below run a grid-stride loop over at most
// kPcgReduceBlocks workgroups: a dot product over 9 M unknowns then ends in 1024 atomics on its scalar instead of one per
// wavefront (141 k atomics on ONE address serialise at 12.6 ns each -- 1.79 ms for a kernel that moves 0.2 GB).
constexpr unsigned kPcgReduceBlocks = 1024;
__device__ __forceinline__ void block_atomic_sum(float* dest, float value) {
  __shared__ float partial[kPcgBlock / 64];
  const float v = wave_sum(value);
  if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float total = ((partial[0] + partial[1]) + partial[2]) + partial[3];
    if (total != 0.f) unsafeAtomicAdd(dest, total);
  }
}

// PCGInit2 (B/kernel_pcg.cu:565-600)
__global__ void __launch_bounds__(kPcgBlock)
pcg_init2_kernel(PcgLayout L, float a, const float* __restrict__ r_, const float* __restrict__ M_, float* __restrict__ delta,
                 float* __restrict__ g_, float* __restrict__ p_, float* alpha_n) {
  float term = 0.f;
  for (uint32_t u = blockIdx.x * kPcgBlock + threadIdx.x; u < L.unknown_count; u += gridDim.x * kPcgBlock) {
    g_[u] = 0;
    const float r_value = r_[u] + ((u == L.a_index) ? (-kAPriorWeight * kAPriorWeight * a) : 0);
    const float p_value = r_value / (M_[u] + kDiagEpsilon + prior_at(L, u));
    p_[u] = p_value;
    delta[u] = 0;
    term += dot_weight(L, u) * (r_value * p_value);
  }
  block_atomic_sum(alpha_n, term);
}

// ---- PCGStep1: g += J^T W J p, alpha_d += p^T J^T W J p  (B/kernel_pcg.cu:646-1026) -------------------
template <bool kDepthIntr, bool kColorIntr>
__global__ void __launch_bounds__(kPcgSweepBlock) BAHIP_PCG_SWEEP_ATTR
pcg_step1_kernel(PcgLayout L, Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s,
                 const float* __restrict__ p_, float* __restrict__ g_, float* alpha_d, const PcgControl* ctl) {
  if (ctl->stop) return;
  const uint32_t i = xcd_chunked_tile(blockIdx.x) * kPcgSweepBlock + threadIdx.x;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const float radius_sq = s.row(kSurfelRadiusSquared)[ii];
  const float d1 = s.row(kSurfelDescriptor1)[ii], d2 = s.row(kSurfelDescriptor2)[ii];
  const TangentPoints tp = surfel_tangent_points(gp, gn, radius_sq);   // per surfel, not per pair
  const WaveBounds wb = wave_bounds(gp, in_range && (gp.x == gp.x));
  const int lane = threadIdx.x & 63;
  const uint32_t gi = L.optimize_geometry ? (L.surfel_start + (uint32_t)L.geom_stride * ii) : 0u;
  float ps[3] = {0, 0, 0};
  if (L.optimize_geometry) {
    ps[0] = p_[gi];
    if (L.geom_stride == 3) { ps[1] = p_[gi + 1]; ps[2] = p_[gi + 2]; }
  }
  float pdi[5] = {0, 0, 0, 0, 0}, pci[4] = {0, 0, 0, 0};
  if (kDepthIntr) for (int c = 0; c < 5; ++c) pdi[c] = p_[L.depth_intr_start + c];
  if (kColorIntr) for (int c = 0; c < 4; ++c) pci[c] = p_[L.color_intr_start + c];
  float gs[3] = {0, 0, 0};
  float gi_acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q) gi_acc[q] = 0.f;
  float ad = 0.f;

  for_each_candidate(
      num_kfs, [&](int k) { return sphere_may_project_item(in, kfs[k].pose.F, wb); },
      [&](int k) {
        const KfEntry& kf = kfs[k];
        Assoc a;
        const bool visible = in_range && project_associate<false>(in, kf.pose.F, kf.geom, gp, gn, &a, nullptr);
        if (!__any(visible)) return;
        const bool pose_kf = L.optimize_poses && (k != L.gauge);
        const uint32_t base = kf_pose_index(L, k);
        float pp[6] = {0, 0, 0, 0, 0, 0};
        if (pose_kf) for (int c = 0; c < 6; ++c) pp[c] = p_[base + c];
        float gpose[6] = {0, 0, 0, 0, 0, 0};
        if (visible) {
          PairTerms t;
          eval_pair_terms<kDepthIntr, kColorIntr>(L, in, kf, a, gn, tp, d1, d2, &t);
          if (L.use_depth) {
            float sum = 0;
            if (L.optimize_geometry) sum += t.Jgeom * ps[0];
            if (pose_kf) {
#pragma unroll
              for (int c = 0; c < 6; ++c) sum += t.Jpose[c] * pp[c];
            }
            const bool di = kDepthIntr && t.di_valid;
            float pcf = 0.f;
            if (di) {
              sum += t.Jdi[2] * pdi[2];
              sum += t.Jdi[3] * pdi[3];
              sum += t.Jdi[0] * pdi[0];
              sum += t.Jdi[1] * pdi[1];
              sum += t.Jdi[4] * pdi[4];
              pcf = p_[t.cf_index];
              sum += t.Jcf * pcf;
            }
            ad += sum * t.w * sum;
            sum *= t.w;
            if (L.optimize_geometry) gs[0] += t.Jgeom * sum;
            if (pose_kf) {
#pragma unroll
              for (int c = 0; c < 6; ++c) gpose[c] += t.Jpose[c] * sum;
            }
            if (di) {
#pragma unroll
              for (int c = 0; c < 5; ++c) gi_acc[c] += t.Jdi[c] * sum;
              unsafeAtomicAdd(&g_[t.cf_index], t.Jcf * sum);
            }
          }
          if (L.use_desc && t.color_ok) {
            float sum1 = 0, sum2 = 0;
            if (L.optimize_geometry) {
              sum1 += t.Jg1 * ps[0]; sum2 += t.Jg2 * ps[0];
              sum1 += -1.f * ps[1];
              sum2 += -1.f * ps[2];
            }
            if (pose_kf) {
#pragma unroll
              for (int c = 0; c < 6; ++c) { sum1 += t.Jp1[c] * pp[c]; sum2 += t.Jp2[c] * pp[c]; }
            }
            if (kColorIntr) {
#pragma unroll
              for (int c = 0; c < 4; ++c) { sum1 += t.Jci1[c] * pci[c]; sum2 += t.Jci2[c] * pci[c]; }
            }
            ad += sum1 * t.w1 * sum1 + sum2 * t.w2 * sum2;
            sum1 *= t.w1; sum2 *= t.w2;
            if (L.optimize_geometry) {
              gs[0] += t.Jg1 * sum1 + t.Jg2 * sum2;
              gs[1] += -1.f * sum1 + 0.f * sum2;
              gs[2] += 0.f * sum1 + -1.f * sum2;
            }
            if (pose_kf) {
#pragma unroll
              for (int c = 0; c < 6; ++c) gpose[c] += t.Jp1[c] * sum1 + t.Jp2[c] * sum2;
            }
            if (kColorIntr) {
#pragma unroll
              for (int c = 0; c < 4; ++c) gi_acc[5 + c] += t.Jci1[c] * sum1 + t.Jci2[c] * sum2;
            }
          }
        }
        if (pose_kf) {
          const float v[8] = {gpose[0], gpose[1], gpose[2], gpose[3], gpose[4], gpose[5], 0.f, 0.f};
          const float mine = wave_reduce_small<8>(v, lane);   // lane 8 j holds total j
          if ((lane & 7) == 0 && lane < 48) unsafeAtomicAdd(&g_[base + (lane >> 3)], mine);
        }
      });

  if (in_range && L.optimize_geometry) {
    g_[gi] = gs[0];
    if (L.geom_stride == 3) { g_[gi + 1] = gs[1]; g_[gi + 2] = gs[2]; }
  }
  float mine = 0.f;
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const float v = wave_sum(gi_acc[q]);
    if (lane == q) mine = v;
  }
  const float adv = wave_sum(ad);
  if (lane == 9) mine = adv;
  if (lane < 9 && mine != 0.f) {
    if (lane < 5) { if (kDepthIntr) unsafeAtomicAdd(&g_[L.depth_intr_start + lane], mine); }
    else if (kColorIntr) unsafeAtomicAdd(&g_[L.color_intr_start + lane - 5], mine);
  }
  if (lane == 9 && mine != 0.f) unsafeAtomicAdd(alpha_d, mine);
}

// AddAlphaDEpsilonTerms (B/kernel_pcg.cu:1028-1050); the reference launches it once per keyframe
// (B/kernel_pcg.cu:1102-1112), i.e. the term enters alpha_d `repeat` times -- reproduced.
__global__ void __launch_bounds__(kPcgBlock)
pcg_eps_terms_kernel(PcgLayout L, const float* __restrict__ p_, float repeat, float* alpha_d, const PcgControl* ctl) {
  if (ctl->stop) return;
  float term = 0.f;
  for (uint32_t u = blockIdx.x * kPcgBlock + threadIdx.x; u < L.unknown_count; u += gridDim.x * kPcgBlock) {
    const float pv = p_[u];
    term += dot_weight(L, u) * ((kDiagEpsilon + prior_at(L, u)) * pv * pv);
  }
  block_atomic_sum(alpha_d, repeat * term);
}

// PCGStep2 (B/kernel_pcg.cu:1117-1158)
__global__ void __launch_bounds__(kPcgBlock)
pcg_step2_kernel(PcgLayout L, float* __restrict__ r_, const float* __restrict__ M_, float* __restrict__ delta, float* __restrict__ g_,
                 const float* __restrict__ p_, const float* alpha_n, const float* alpha_d, float* beta_n, const PcgControl* ctl) {
  if (ctl->stop) return;
  float term = 0.f;
  const float ad = *alpha_d;
  const float alpha = (ad >= 1e-35f) ? (*alpha_n / ad) : 0;
  for (uint32_t u = blockIdx.x * kPcgBlock + threadIdx.x; u < L.unknown_count; u += gridDim.x * kPcgBlock) {
    const float p_value = p_[u];
    delta[u] += alpha * p_value;
    float r_value = r_[u];
    r_value -= alpha * (g_[u] + (kDiagEpsilon + prior_at(L, u)) * p_value);
    r_[u] = r_value;
    const float z_value = r_value / (M_[u] + kDiagEpsilon + prior_at(L, u));
    g_[u] = z_value;
    term += dot_weight(L, u) * (z_value * r_value);
  }
  block_atomic_sum(beta_n, term);
}

// PCGStep3 (B/kernel_pcg.cu:1212-1226)
__global__ void __launch_bounds__(kPcgBlock)
pcg_step3_kernel(PcgLayout L, const float* __restrict__ g_, float* __restrict__ p_, const float* alpha_n, const float* beta_n,
                 const PcgControl* ctl) {
  const uint32_t u = blockIdx.x * kPcgBlock + threadIdx.x;
  if (ctl->stop) return;
  if (u < L.unknown_count) {
    const float an = *alpha_n;
    const float beta = (an >= 1e-35f) ? (*beta_n / an) : 0;
    p_[u] = g_[u] + beta * p_[u];
  }
}

// UpdateSurfelsFromPCGDelta (B/kernel_pcg.cu:1306-1331)
__global__ void __launch_bounds__(kPcgBlock)
pcg_update_surfels_kernel(PcgLayout L, SurfelsView s, const float* __restrict__ delta) {
  const uint32_t i = blockIdx.x * kPcgBlock + threadIdx.x;
  if (i >= s.size) return;
  const uint32_t gi = L.surfel_start + (uint32_t)L.geom_stride * i;
  const float t = delta[gi];
  if (t != 0) {
    const Vec3 np = surfel_position(s, i) + t * surfel_normal(s, i);
    s.row(kSurfelX)[i] = np.x; s.row(kSurfelY)[i] = np.y; s.row(kSurfelZ)[i] = np.z;
  }
  if (L.geom_stride == 3) {
    float a = s.row(kSurfelDescriptor1)[i]; a += delta[gi + 1];
    s.row(kSurfelDescriptor1)[i] = fmaxf(-180.f, fminf(180.f, a));
    float b = s.row(kSurfelDescriptor2)[i]; b += delta[gi + 2];
    s.row(kSurfelDescriptor2)[i] = fmaxf(-180.f, fminf(180.f, b));
  }
}

// UpdateCFactorsFromPCGDelta (B/kernel_pcg.cu:1361-1373)
__global__ void __launch_bounds__(kPcgBlock)
pcg_update_cfactors_kernel(Intrinsics in, uint32_t start, const float* __restrict__ delta, float* cfactor, uint32_t pitch) {
  const int idx = blockIdx.x * kPcgBlock + threadIdx.x;
  if (idx >= in.cf_width * in.cf_height) return;
  const int y = idx / in.cf_width, x = idx - y * in.cf_width;
  *pitched_ptr(cfactor, pitch, y, x) += delta[start + idx];
}

// ---- launchers -----------------------------------------------------------------------------------------------
static inline unsigned gU(uint32_t n) { return (n + kPcgBlock - 1) / kPcgBlock; }
static inline unsigned gR(uint32_t n) { return gU(n) < kPcgReduceBlocks ? gU(n) : kPcgReduceBlocks; }   // grid-stride reductions

// whole XCD chunks, as in kernels_surfel.hip (xcd_chunked_tile)
static inline unsigned gS(uint32_t n) { return xcd_padded_tiles((n + kPcgSweepBlock - 1) / kPcgSweepBlock); }

void launch_pcg_init(hipStream_t st, const PcgLayout& L, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                     float* r, float* M) {
  if (!s.size) return;
  const dim3 grid(gS(s.size)), block(kPcgSweepBlock);
  const bool di = L.optimize_depth_intrinsics, ci = L.optimize_color_intrinsics;
  if (di && ci) hipLaunchKernelGGL((pcg_init_kernel<true, true>), grid, block, 0, st, L, in, kfs, num_kfs, s, r, M);
  else if (di) hipLaunchKernelGGL((pcg_init_kernel<true, false>), grid, block, 0, st, L, in, kfs, num_kfs, s, r, M);
  else if (ci) hipLaunchKernelGGL((pcg_init_kernel<false, true>), grid, block, 0, st, L, in, kfs, num_kfs, s, r, M);
  else hipLaunchKernelGGL((pcg_init_kernel<false, false>), grid, block, 0, st, L, in, kfs, num_kfs, s, r, M);
}
void launch_pcg_init2(hipStream_t st, const PcgLayout& L, float a, const float* r, const float* M, float* delta, float* g, float* p,
                      float* alpha_n) {
  if (L.unknown_count) hipLaunchKernelGGL(pcg_init2_kernel, dim3(gR(L.unknown_count)), dim3(kPcgBlock), 0, st, L, a, r, M, delta, g, p, alpha_n);
}
void launch_pcg_control_init(hipStream_t st, void* ctl) { hipLaunchKernelGGL(pcg_control_init_kernel, dim3(1), dim3(1), 0, st, static_cast<PcgControl*>(ctl)); }
void launch_pcg_control(hipStream_t st, void* ctl, const float* beta_n) {
  hipLaunchKernelGGL(pcg_control_kernel, dim3(1), dim3(1), 0, st, static_cast<PcgControl*>(ctl), beta_n);
}
size_t pcg_control_bytes() { return sizeof(PcgControl); }

void launch_pcg_step1(hipStream_t st, const PcgLayout& L, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                      const float* p, float* g, float* alpha_d, const void* ctl_) {
  const PcgControl* ctl = static_cast<const PcgControl*>(ctl_);
  if (!s.size) return;
  const dim3 grid(gS(s.size)), block(kPcgSweepBlock);
  const bool di = L.optimize_depth_intrinsics, ci = L.optimize_color_intrinsics;
  if (di && ci) hipLaunchKernelGGL((pcg_step1_kernel<true, true>), grid, block, 0, st, L, in, kfs, num_kfs, s, p, g, alpha_d, ctl);
  else if (di) hipLaunchKernelGGL((pcg_step1_kernel<true, false>), grid, block, 0, st, L, in, kfs, num_kfs, s, p, g, alpha_d, ctl);
  else if (ci) hipLaunchKernelGGL((pcg_step1_kernel<false, true>), grid, block, 0, st, L, in, kfs, num_kfs, s, p, g, alpha_d, ctl);
  else hipLaunchKernelGGL((pcg_step1_kernel<false, false>), grid, block, 0, st, L, in, kfs, num_kfs, s, p, g, alpha_d, ctl);
  hipLaunchKernelGGL(pcg_eps_terms_kernel, dim3(gR(L.unknown_count)), dim3(kPcgBlock), 0, st, L, p, (float)num_kfs, alpha_d, ctl);
}
void launch_pcg_step2(hipStream_t st, const PcgLayout& L, float* r, const float* M, float* delta, float* g, const float* p,
                      const float* alpha_n, const float* alpha_d, float* beta_n, const void* ctl) {
  if (L.unknown_count) hipLaunchKernelGGL(pcg_step2_kernel, dim3(gR(L.unknown_count)), dim3(kPcgBlock), 0, st, L, r, M, delta, g, p, alpha_n, alpha_d, beta_n,
                                          static_cast<const PcgControl*>(ctl));
}
void launch_pcg_step3(hipStream_t st, const PcgLayout& L, const float* g, float* p, const float* alpha_n, const float* beta_n, const void* ctl) {
  if (L.unknown_count) hipLaunchKernelGGL(pcg_step3_kernel, dim3(gU(L.unknown_count)), dim3(kPcgBlock), 0, st, L, g, p, alpha_n, beta_n,
                                          static_cast<const PcgControl*>(ctl));
}
void launch_pcg_update_surfels(hipStream_t st, const PcgLayout& L, const SurfelsView& s, const float* delta) {
  if (s.size) hipLaunchKernelGGL(pcg_update_surfels_kernel, dim3(gU(s.size)), dim3(kPcgBlock), 0, st, L, s, delta);
}
void launch_pcg_update_cfactors(hipStream_t st, const Intrinsics& in, uint32_t start, const float* delta, float* cfactor, uint32_t pitch) {
  hipLaunchKernelGGL(pcg_update_cfactors_kernel, dim3(gU(in.cf_width * in.cf_height)), dim3(kPcgBlock), 0, st, in, start, delta, cfactor, pitch);
}

}  // namespace bahip
