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
// keyframe order (identical to the reference's launch order) and written once.
//
// DEFINITION of the dense sums (round 3; restated by oracle_pcg.c).  Everything the reference merges with binary32
// atomics in arbitrary order (B/kernel_pcg.cu:98-154) is an EXACT sum here (exact_sum.h: 9 x int64 limbs, integer atomics),
// rounded once to binary64 and then to the PCGScalar:
//   - the 6 pose entries of a keyframe, the 5 + 4 global intrinsics entries and the pair part of alpha_d: per (64-surfel tile,
//     keyframe) the halving tree of wave_reduce.h over that keyframe's per-surfel terms, the (tile, keyframe) totals exact --
//     the intrinsics entries and alpha_d into one of 64 replicas, folded when resolved (exactness makes the replication free,
//     and 47 k tiles would otherwise serialise on one address).  Nothing is carried from one keyframe to the next, so one
//     sweep over all keyframes and one call per keyframe (the stage entry points of the C ABI) give the same sums;
//   - per-cell cfactor entries: per-pair terms, exact;
//   - dot products over the unknowns (alpha_n, beta_n, the epsilon terms of alpha_d): the binary32 products, exact.
// So the conjugate gradient is deterministic, bit-identical to the oracle (inner step count included), independent of the
// launch shape, and a surfel-sharded run -- limbs exchanged with an int64 all-reduce -- equals the unsharded one bit for bit.
#include "ba_device.h"
#include "ba_launch.h"
#include "exact_sum.h"
#include "wave_cull.h"

namespace bahip {

constexpr int kPcgBlock = 256;     // per-unknown vector kernels
constexpr int kPcgSweepBlock = 64; // surfel sweeps (init, step 1): one wavefront per workgroup, like kernels_surfel.hip
#define BAHIP_PCG_SWEEP_ATTR __attribute__((amdgpu_waves_per_eu(4)))   // 128-VGPR cap: 4 waves per SIMD
constexpr float kDiagEpsilon = 1e-8f;   // B/kernel_pcg.cu:44
constexpr float kAPriorWeight = 10.f;   // B/kernel_pcg.cu:48

__device__ __forceinline__ uint32_t kf_pose_index(const PcgLayout& L, int k) {
  // B/direct_ba_pcg.cc:329-337
  if (L.single_keyframe >= 0) return L.single_pose_index;   // per-keyframe entry points (Route B): the caller names the index
  if (k == L.gauge) return 0xffffffffu;
  return (k < L.gauge) ? 6u * (uint32_t)k : 6u * (uint32_t)(k - 1);
}
__device__ __forceinline__ bool kf_pose_is_unknown(const PcgLayout& L, int k) {
  return L.optimize_poses && (L.single_keyframe >= 0 || k != L.gauge);
}
__device__ __forceinline__ float prior_at(const PcgLayout& L, uint32_t u) {
  return (u == L.a_index) ? (kAPriorWeight * kAPriorWeight) : 0.f;
}
__device__ __forceinline__ bool is_local(const PcgLayout& L, uint32_t u) { return u >= L.head_lo && u < L.head_hi; }
__device__ __forceinline__ uint32_t head_index(const PcgLayout& L, uint32_t u) { return u < L.head_lo ? u : L.head_lo + (u - L.head_hi); }
// Slot (0..8) of a global intrinsics unknown among the replicated accumulators, or -1.
__device__ __forceinline__ int intrinsics_slot(const PcgLayout& L, uint32_t u) {
  if (L.optimize_depth_intrinsics && u >= L.depth_intr_start && u < L.depth_intr_start + 5u) return (int)(u - L.depth_intr_start);
  if (L.optimize_color_intrinsics && u >= L.color_intr_start && u < L.color_intr_start + 4u) return 5 + (int)(u - L.color_intr_start);
  return -1;
}

// ---- the exact accumulators of one PCG solve ------------------------------------------------------------------------------
// One device allocation, laid out so that what a sharded run exchanges is contiguous:
//   [ replicated slots 0..19 | head A | head B | invalid flag | replicated slot 20 | replicated slots 21..22 ]
// exchange 1 (after a sweep) = slots 0..19 + head A (+ head B after PCGInit); exchange 2 (after a dot product) = the flag's cell +
// slot 20.  The sticky flag travels with exchange 2, which precedes every evaluation of the stopping rule: once any rank has
// added a non-finite term, every rank's flag is non-zero (the integer sum counts the ranks that raised it) from the next control
// step on, all ranks stop after the same inner step and all fail the call alike (round 4 kept the flag behind the block: a
// rank that saw a NaN stopped three steps later, its peers went on, and the collective sequence diverged -- a hang).
// Slots 21 / 22 hold the dense head's share of the dot products: the head is replicated over the ranks, so every rank adds the
// same terms there and they are NOT exchanged.
__device__ __forceinline__ ExactCell* hot_cell(const PcgExact& ex, int slot, int replica) {
  return (slot < kHotExchanged1 ? ex.hot : ex.hot_tail - (size_t)kHotExchanged1 * kHotReplicas) + (size_t)slot * kHotReplicas + replica;
}
// One wavefront folds the 64 replicas of slot a (and of slot b if b >= 0) into the exact binary64 value; optionally clears them.
__device__ __forceinline__ double fold_hot(const PcgExact& ex, int a, int b, bool clear) {
  const int lane = threadIdx.x & 63;
  long long l[kExactLimbs];
  ExactCell* ca = hot_cell(ex, a, lane);
  ExactCell* cb = b >= 0 ? hot_cell(ex, b, lane) : nullptr;
#pragma unroll
  for (int j = 0; j < kExactLimbs; ++j) {
    l[j] = ca->limb[j] + (cb ? cb->limb[j] : 0ll);
    if (clear) { ca->limb[j] = 0; if (cb) cb->limb[j] = 0; }
  }
#pragma unroll
  for (int j = 0; j < kExactLimbs; ++j)
    for (int off = 32; off; off >>= 1) l[j] += __shfl_xor(l[j], off);
  return exact_value(l);
}

}  // namespace bahip

// ---- the sweeps (PCGInit, PCGStep1): compiled once per arithmetic flavour (ba_launch.h) ----
BAHIP_FLAVOURED_BEGIN
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
                                                const PixelWords& pix, const DescWords& dw, Vec3 gn, float d1, float d2, PairTerms* t) {
  const float* F = kf.pose.F;
  const Vec3 rn = r.nl;
  const float nx = r.nx, ny = r.ny;   // the association computed them (ba_device.h: Assoc)
  t->di_valid = false;
  t->color_ok = false;
  // All Jacobians come from the jac_* functions of ba_device.h -- the ones the alternating sweeps use and the ones checked
  // against the golden vectors derived from the reference's own script (tests/golden/jacobians.json).
  if (L.use_depth) {
    const float inv_std = assoc_inv_std(in, r);
    const Vec3 u = assoc_unproject(r);
    t->raw = inv_std * dot3(rn, u - r.local);
    t->w = depth_residual_weight(t->raw);
    t->Jgeom = -inv_std;
    jac_depth_pose(rn, u, inv_std, t->Jpose);
    if (kDepthIntr) {
      // cfactor of the pixel's cell and the raw depth: the words the association already loaded (the geometry plane's low
      // half is the keyframe's depth image)
      const int sparse_px = r.px / in.cell, sparse_py = r.py / in.cell;
      const float cfactor = pix.cfactor;
      const float raw_inv_depth = 1.0f / (in.raw_to_float_depth * (uint16_t)(pix.geom & 0xffffu));
      const float exp_inv_depth = exp_det(-in.a * raw_inv_depth);
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
    t->color_ok = dw.color_ok;
    if (t->color_ok) {
      DescEval e;
      eval_descriptor_from_words(in, kf.lumafp, dw, d1, d2, &e);
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

// The gathers of one (surfel, keyframe) pair, all in flight before the first is waited for (ba_device.h: project_surfel), and
// the association on the loaded words.
struct PairGather {
  PixelWords pix;
  DescWords dw;
  Assoc a;
};
__device__ __forceinline__ bool gather_and_associate(const PcgLayout& L, const Intrinsics& in, const KfEntry& kf, Vec3 gp, Vec3 gn,
                                                     const TangentPoints& tp, bool in_range, PairGather* g) {
  const float* F = kf.pose.F;
  const Projected p = project_surfel(in, F, gp);
  g->pix = load_pixel_words(in, kf.geom, p);
  if (L.use_desc) g->dw = load_descriptor_words(in, kf.lumafp, F, tp, p);
  const bool visible = in_range && associate_from_words<false>(in, F, gn, p, g->pix, &g->a, nullptr);
  if (L.use_desc) gathers_arrived(g->pix, g->dw);
  else gathers_arrived(g->pix);
  return visible;
}

// ---- PCGInit: r -= J^T W F, M += diag(J^T W J)  (B/kernel_pcg.cu:179-541) -----------------------------
template <bool kDepthIntr, bool kColorIntr>
__global__ void __launch_bounds__(kPcgSweepBlock) BAHIP_PCG_SWEEP_ATTR
pcg_init_kernel(PcgLayout L, PcgExact ex, Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s,
                float* __restrict__ r_, float* __restrict__ M_, uint32_t* __restrict__ tile_cost, const uint32_t* __restrict__ sched) {
  uint32_t tile;   // heavy work first (wave_cull.h: scheduled_tile); tile_cost: the census the schedule is built from
  if (!scheduled_tile(blockIdx.x, gridDim.x - (sched ? kHeavySlots : 0u), sched, &tile)) return;
  uint32_t visited = 0;
  const uint32_t i = tile * kPcgSweepBlock + threadIdx.x;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const float radius_sq = s.row(kSurfelRadiusSquared)[ii];
  const float d1 = s.row(kSurfelDescriptor1)[ii], d2 = s.row(kSurfelDescriptor2)[ii];
  const TangentPoints tp = surfel_tangent_points(gp, gn, radius_sq);   // per surfel, not per pair
  const WaveBounds wb = wave_bounds(gp, in_range && (gp.x == gp.x));
  const int lane = threadIdx.x & 63;
  const int replica = (int)(tile & (kHotReplicas - 1));
  const uint32_t gi = L.optimize_geometry ? (L.surfel_start + (uint32_t)L.geom_stride * ii) : 0u;
  float gr[3] = {0, 0, 0}, gM[3] = {0, 0, 0};     // surfel entries
  if (L.accumulate && in_range && L.optimize_geometry) {   // per-keyframe calls continue the chain the earlier calls left
    gr[0] = r_[gi]; gM[0] = M_[gi];
    if (L.geom_stride == 3) { gr[1] = r_[gi + 1]; gM[1] = M_[gi + 1]; gr[2] = r_[gi + 2]; gM[2] = M_[gi + 2]; }
  }
  constexpr bool kIntr = kDepthIntr || kColorIntr;
  // The exact atomics of a candidate keyframe are issued one candidate late, behind the next candidate's gathers
  // (exact_sum.h: exact_atomic_add_part_untracked): the (tile, keyframe) totals wait in pending_a / pending_b -- lanes
  // 4 j .. 4 j + 3 hold total j of a 16-value halving butterfly: pending_a = 6 pose entries of r, 6 of M, the first 4 global
  // intrinsics entries of r; pending_b (intrinsics only) = the other 5 of r and the 9 of M -- and the per-cell cfactor terms
  // of a lane in pending_cf*.
  float pending_a = 0.f, pending_b = 0.f;
  bool pending_any = false, pending_pose = false;   // wave-uniform
  uint32_t pending_base = 0;                          // wave-uniform
  uint32_t pending_cf = 0xffffffffu;                  // per lane: head index of the cell
  float pending_cf_r = 0.f, pending_cf_M = 0.f;
  auto intr_enabled = [&](int q) { return q < 5 ? kDepthIntr : kColorIntr; };
  auto flush_pending = [&]() {
    if (pending_any) {
      const int j = lane >> 2, part = lane & 3;   // two of the four lanes that hold total j add its two parts
      if (part < 2) {
        if (j < 12) {
          if (pending_pose) exact_atomic_add_part_untracked(j < 6 ? &ex.head_a[pending_base + j] : &ex.head_b[pending_base + j - 6], pending_a, part, ex.invalid);
        } else if (kIntr && intr_enabled(j - 12)) {
          exact_atomic_add_part_untracked(hot_cell(ex, kHotA + (j - 12), replica), pending_a, part, ex.invalid);
        }
        if (kIntr) {
          if (j < 5) { if (intr_enabled(4 + j)) exact_atomic_add_part_untracked(hot_cell(ex, kHotA + 4 + j, replica), pending_b, part, ex.invalid); }
          else if (j < 14 && intr_enabled(j - 5)) exact_atomic_add_part_untracked(hot_cell(ex, kHotB + (j - 5), replica), pending_b, part, ex.invalid);
        }
      }
      pending_any = false;
    }
    if (kDepthIntr) {
      if (pending_cf != 0xffffffffu) {
#pragma unroll
        for (int part = 0; part < 2; ++part) {
          exact_atomic_add_part_untracked(&ex.head_a[pending_cf], pending_cf_r, part, ex.invalid);
          exact_atomic_add_part_untracked(&ex.head_b[pending_cf], pending_cf_M, part, ex.invalid);
        }
      }
      pending_cf = 0xffffffffu;
    }
  };
  for_each_candidate(
      num_kfs, [&](int k) { return sphere_may_project_item(in, kfs[k].pose.F, wb); },
      [&](int k) {
        const KfEntry& kf = kfs[k];
        ++visited;
        PairGather pg;
        bool visible = gather_and_associate(L, in, kf, gp, gn, tp, in_range, &pg);
        flush_pending();
        if (!__any(visible)) return;
        const bool pose_kf = kf_pose_is_unknown(L, k);
        float pr[6] = {0, 0, 0, 0, 0, 0}, pM[6] = {0, 0, 0, 0, 0, 0};
        float ir[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, iM[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // this keyframe's global intrinsics terms
        if (visible) {
          PairTerms t;
          eval_pair_terms<kDepthIntr, kColorIntr>(L, in, kf, pg.a, pg.pix, pg.dw, gn, d1, d2, &t);
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
                pending_cf = head_index(L, t.cf_index);
                pending_cf_r = -1 * wj * t.raw;
                pending_cf_M = t.Jcf * wj;
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
        if (pose_kf || kIntr) {
          // (tile, keyframe) totals with the halving butterfly of wave_reduce.h: lanes 4 j .. 4 j + 3 hold total j
          const float va[16] = {pr[0], pr[1], pr[2], pr[3], pr[4], pr[5], pM[0], pM[1], pM[2], pM[3], pM[4], pM[5], ir[0], ir[1], ir[2], ir[3]};
          pending_a = wave_reduce_small<16>(va, lane);
          if (kIntr) {
            const float vb[16] = {ir[4], ir[5], ir[6], ir[7], ir[8], iM[0], iM[1], iM[2], iM[3], iM[4], iM[5], iM[6], iM[7], iM[8], 0.f, 0.f};
            pending_b = wave_reduce_small<16>(vb, lane);
          }
          pending_any = true;
          pending_pose = pose_kf;
          pending_base = pose_kf ? kf_pose_index(L, k) : 0u;   // pose unknowns come first: head index == unknown index
        }
      });
  flush_pending();
  if (tile_cost && lane == 0 && visited) atomicAdd(&tile_cost[tile], visited);

  if (in_range && L.optimize_geometry) {
    r_[gi] = gr[0]; M_[gi] = gM[0];
    if (L.geom_stride == 3) { r_[gi + 1] = gr[1]; M_[gi + 1] = gM[1]; r_[gi + 2] = gr[2]; M_[gi + 2] = gM[2]; }
  }
}

BAHIP_FLAVOURED_END
namespace bahip {
// ---- inner-loop control on the device -------------------------------------------------------------------------------------
// The reference reads beta_n back after every inner step and decides on the host whether the residual norm still improves
// (B/direct_ba_pcg.cc:427-456: stop after three steps without an improvement of 1e-3).  Here a one-wavefront kernel resolves
// beta_n and takes that decision after step 2; once `stop` is set the sweeps and vector kernels queued behind it return at
// once, so the host can queue several inner steps without waiting for any of them.
struct PcgControl {
  double prev_r_norm;
  int no_improvement;
  int stop;
  int steps;
  int pad;
};
static_assert(sizeof(PcgControl) == 24, "PcgControl lives behind the scalars of the PCG buffer");
}  // namespace bahip
#ifndef BAHIP_FAST_MATH   // exists once (the exact unit)
namespace bahip {
// The sticky flag travels in exchange 2 as a 64-bit integer sum: after an exchange it holds (ranks that raised it) x (what it held
// before), and a host that keeps queuing exchanges for the rest of a step group after the device has stopped would multiply it by the
// world size every time -- at 64 ranks the low 32 bits, which is all the readers look at, wrap to 0 after six exchanges.
// Every control step therefore puts it back to 0 / 1, stopped or not.
__device__ __forceinline__ void pcg_renormalise_invalid(const PcgExact& ex) {
  if (threadIdx.x == 0) {
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(ex.invalid);
    if (*flag != 0ull) *flag = 1ull;
  }
}
// after PCGInit2: alpha_n = the exact dot product, rounded; the control block starts over
__global__ void __launch_bounds__(64) pcg_control_init_kernel(PcgExact ex, PcgControl* ctl, float* alpha_n) {
  pcg_renormalise_invalid(ex);
  const double v = fold_hot(ex, kHotDotLocal, kHotDotHead, true);
  if (threadIdx.x == 0) {
    *alpha_n = (*ex.invalid) ? __builtin_nanf("") : (float)v;
    ctl->prev_r_norm = __builtin_huge_val(); ctl->no_improvement = 0; ctl->stop = 0; ctl->steps = 0;
  }
}
// after PCGStep2: beta_n, then the stopping rule.  r_norm is a PCGScalar (binary32 square root), the comparison is evaluated
// in double like the reference's `r_norm < prev_r_norm - 1e-3` (B/direct_ba_pcg.cc:441-446).
__global__ void __launch_bounds__(64) pcg_control_kernel(PcgExact ex, PcgControl* ctl, float* beta_n) {
  pcg_renormalise_invalid(ex);
  if (ctl->stop) return;
  const double v = fold_hot(ex, kHotDotLocal, kHotDotHead, true);
  if (threadIdx.x == 0) {
    const float bn = (*ex.invalid) ? __builtin_nanf("") : (float)v;
    *beta_n = bn;
    ctl->steps += 1;
    const float r_norm = __builtin_sqrtf(bn);   // IEEE (no fast-math): the correctly rounded root, like the oracle's sqrtf
    if ((double)r_norm < ctl->prev_r_norm - 1e-3) ctl->no_improvement = 0;
    else if (++ctl->no_improvement >= 3) ctl->stop = 1;
    ctl->prev_r_norm = (double)r_norm;
  }
}

// ---- resolving the exact accumulators into the PCGScalar vectors -------------------------------------------------------------
// Workgroups [0, head blocks): one thread per dense-head unknown: exact value -> va[u] (and vb[u]); the cell is cleared.  The
// last workgroup folds the replicated slots: the global intrinsics entries and, after PCGStep1, alpha_d.
template <bool kInit>
__global__ void __launch_bounds__(kPcgBlock)
pcg_resolve_kernel(PcgLayout L, PcgExact ex, float* __restrict__ va, float* __restrict__ vb, float* alpha_d, double eps_repeat,
                   const PcgControl* ctl) {
  if (!kInit && ctl->stop) return;
  const bool invalid = *ex.invalid != 0u;
  const uint32_t head_count = L.head_lo + (L.unknown_count - L.head_hi);
  if (blockIdx.x + 1 < gridDim.x) {
    const uint32_t h = blockIdx.x * kPcgBlock + threadIdx.x;
    if (h >= head_count) return;
    const uint32_t u = h < L.head_lo ? h : L.head_hi + (h - L.head_lo);
    if (intrinsics_slot(L, u) >= 0) return;   // written by the last workgroup
    ExactCell* ca = &ex.head_a[h];
    va[u] = invalid ? __builtin_nanf("") : (float)exact_value(ca->limb);
#pragma unroll
    for (int j = 0; j < kExactLimbs; ++j) ca->limb[j] = 0;
    if (kInit) {
      ExactCell* cb = &ex.head_b[h];
      vb[u] = invalid ? __builtin_nanf("") : (float)exact_value(cb->limb);
#pragma unroll
      for (int j = 0; j < kExactLimbs; ++j) cb->limb[j] = 0;
    }
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int task = wave; task < (kInit ? 18 : 10); task += kPcgBlock / 64) {
    if (!kInit && task == 9) {
      // alpha_d = pairs + repeat * epsilon terms (AddAlphaDEpsilonTerms runs once per keyframe in the reference,
      // B/kernel_pcg.cu:1102-1112: the term enters `repeat` times), both exact sums, combined in binary64
      const double pairs = fold_hot(ex, kHotAlphaD, -1, true);
      const double eps = fold_hot(ex, kHotEpsLocal, kHotEpsHead, true);
      if (lane == 0) *alpha_d = invalid ? __builtin_nanf("") : (float)(pairs + eps_repeat * eps);
      continue;
    }
    const int q = task % 9;
    const bool enabled = (q < 5) ? L.optimize_depth_intrinsics : L.optimize_color_intrinsics;
    if (!enabled) continue;
    const double v = fold_hot(ex, (task < 9 ? kHotA : kHotB) + q, -1, true);
    const uint32_t u = (q < 5) ? (L.depth_intr_start + q) : (L.color_intr_start + (q - 5));
    if (lane == 0) (task < 9 ? va : vb)[u] = invalid ? __builtin_nanf("") : (float)v;
  }
}

// ---- exact dot products in the per-unknown kernels -----------------------------------------------------------------------------
// A thread's terms over the local (surfel) unknowns go into a private column of limbs in workgroup memory (the limb index is
// data dependent); terms of dense-head unknowns -- few -- go straight to the head's replicated slot with atomics.  At the end
// the workgroup folds its columns and adds 9 limbs per sum to one of the 64 replicas.
template <int kSets>
struct BlockExact {
  long long limbs[kSets][kExactLimbs][kPcgBlock];
};
template <int kSets>
__device__ __forceinline__ void block_exact_clear(BlockExact<kSets>& b) {
#pragma unroll
  for (int set = 0; set < kSets; ++set)
#pragma unroll
    for (int j = 0; j < kExactLimbs; ++j) b.limbs[set][j][threadIdx.x] = 0;
}
template <int kSets>
__device__ __forceinline__ void block_exact_add(BlockExact<kSets>& b, int set, float v, unsigned* invalid) {
  exact_lds_add(&b.limbs[set][0][0], kPcgBlock, (int)threadIdx.x, v, invalid);
}
template <int kSets>
__device__ __forceinline__ void block_exact_flush(BlockExact<kSets>& b, const PcgExact& ex, const int (&slots)[kSets]) {
  __syncthreads();
  const int replica = (int)(blockIdx.x & (kHotReplicas - 1));
  const int part = threadIdx.x & 15;
  for (int row = threadIdx.x >> 4; row < kSets * kExactLimbs; row += kPcgBlock >> 4) {   // 16 threads fold one row of 256 columns
    const long long* line = &b.limbs[0][0][0] + (size_t)row * kPcgBlock;
    long long sum = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) sum += line[c * 16 + part];
    for (int off = 8; off; off >>= 1) sum += __shfl_xor(sum, off);
    if (part == 0 && sum != 0) limb_atomic_add(&hot_cell(ex, slots[row / kExactLimbs], replica)->limb[row % kExactLimbs], sum);
  }
}

// PCGInit2 (B/kernel_pcg.cu:565-600); also the epsilon terms of the first step's alpha_d (they depend on p alone)
__global__ void __launch_bounds__(kPcgBlock)
pcg_init2_kernel(PcgLayout L, PcgExact ex, float a, const float* __restrict__ r_, const float* __restrict__ M_, float* __restrict__ delta,
                 float* __restrict__ g_, float* __restrict__ p_) {
  __shared__ BlockExact<2> acc;
  block_exact_clear(acc);
  const int replica = (int)(blockIdx.x & (kHotReplicas - 1));
  for (uint32_t u = blockIdx.x * kPcgBlock + threadIdx.x; u < L.unknown_count; u += gridDim.x * kPcgBlock) {
    g_[u] = 0;
    const float r_value = r_[u] + ((u == L.a_index) ? (-kAPriorWeight * kAPriorWeight * a) : 0);
    const float p_value = r_value / (M_[u] + kDiagEpsilon + prior_at(L, u));
    p_[u] = p_value;
    delta[u] = 0;
    const float dot_term = r_value * p_value;
    const float eps_term = (kDiagEpsilon + prior_at(L, u)) * p_value * p_value;
    if (is_local(L, u)) {
      block_exact_add(acc, 0, dot_term, ex.invalid);
      block_exact_add(acc, 1, eps_term, ex.invalid);
    } else {
      exact_atomic_add(hot_cell(ex, kHotDotHead, replica), dot_term, ex.invalid);
      exact_atomic_add(hot_cell(ex, kHotEpsHead, replica), eps_term, ex.invalid);
    }
  }
  const int slots[2] = {kHotDotLocal, kHotEpsLocal};
  block_exact_flush(acc, ex, slots);
}

}  // namespace bahip
#endif
BAHIP_FLAVOURED_BEGIN
// ---- PCGStep1: g += J^T W J p, alpha_d += p^T J^T W J p  (B/kernel_pcg.cu:646-1026) -------------------
// Where the (tile, keyframe) totals of the dense head go: straight to the exact accumulators in global memory (two 64-bit integer
// atomics per total, issued one candidate late), or into a copy of the pose block of the head that a persistent workgroup keeps
// in LDS and flushes once (pcg_step1_lds_kernel).  The cfactor-cell entries (one per pair) go to global memory either way.
constexpr int kPcgLdsHotCells = 16;   // of the LDS form's table: [0] alpha_d, [1 .. 9] the nine global intrinsics entries of g; padded
__device__ __forceinline__ uint32_t pcg_lds_hot_index(int slot) { return slot == kHotAlphaD ? 0u : 1u + (uint32_t)(slot - kHotA); }
__device__ __forceinline__ int pcg_lds_hot_slot(uint32_t index) { return index == 0u ? (int)kHotAlphaD : (int)kHotA + (int)index - 1; }
struct PcgGlobalSink {
  PcgExact ex;
  int replica;
  __device__ __forceinline__ void add_pose(uint32_t head_index, float v, int part) const { exact_atomic_add_part_untracked(&ex.head_a[head_index], v, part, ex.invalid); }
  __device__ __forceinline__ void add_hot(int slot, float v, int part) const { exact_atomic_add_part_untracked(hot_cell(ex, slot, replica), v, part, ex.invalid); }
};
struct PcgLdsSink {
  PcgExact ex;
  uint32_t table;        // LDS byte address of the table: ExactCell[pose_cells + kPcgLdsHotCells]
  uint32_t pose_cells;   // 6 per pose unknown: head indices [0, pose_cells)
  __device__ __forceinline__ void add_cell(uint32_t cell, float v, int part) const {
    const ExactSplit sp = exact_split(v);
    if (sp.limb >= 0) {
      const long long addend = part ? sp.hi : sp.lo;
      auto* limb = reinterpret_cast<__attribute__((address_space(3))) long long*>(table + cell * (uint32_t)sizeof(ExactCell) + (uint32_t)(sp.limb + part) * 8u);
      if (addend != 0) __hip_atomic_fetch_add(limb, addend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (sp.limb == -2 && part == 0) {
      atomicOr(ex.invalid, 1u);
    }
  }
  __device__ __forceinline__ void add_pose(uint32_t head_index, float v, int part) const { add_cell(head_index, v, part); }
  __device__ __forceinline__ void add_hot(int slot, float v, int part) const { add_cell(pose_cells + pcg_lds_hot_index(slot), v, part); }
};
template <bool kDepthIntr, bool kColorIntr, typename Sink>
__device__ __forceinline__ void pcg_step1_tile(const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* __restrict__ kfs, int num_kfs,
                                               const SurfelsView& s, const float* __restrict__ p_, float* __restrict__ g_, uint32_t tile, const Sink& sink) {
  const uint32_t i = tile * kPcgSweepBlock + (threadIdx.x & 63);
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
  if (L.accumulate && in_range && L.optimize_geometry) {
    gs[0] = g_[gi];
    if (L.geom_stride == 3) { gs[1] = g_[gi + 1]; gs[2] = g_[gi + 2]; }
  }
  constexpr bool kIntr = kDepthIntr || kColorIntr;
  // exact atomics one candidate late, as in pcg_init_kernel.  The (tile, keyframe) totals: 6 pose entries of g, the keyframe's
  // share of alpha_d and (intrinsics) the 9 global intrinsics entries of g -- 8 or 16 values, one halving butterfly.
  float pending = 0.f;
  bool pending_any = false, pending_pose = false;   // wave-uniform
  uint32_t pending_base = 0;                          // wave-uniform
  uint32_t pending_cf = 0xffffffffu;                  // per lane
  float pending_cf_g = 0.f;
  auto flush_pending = [&]() {
    if (pending_any) {
      const int j = kIntr ? (lane >> 2) : (lane >> 3), part = kIntr ? (lane & 3) : (lane & 7);   // the lanes that hold total j
      if (part < 2) {
        if (j < 6) { if (pending_pose) sink.add_pose(pending_base + j, pending, part); }
        else if (j == 6) sink.add_hot(kHotAlphaD, pending, part);
        else if (kIntr && j < 16 && ((j - 7) < 5 ? kDepthIntr : kColorIntr)) sink.add_hot(kHotA + (j - 7), pending, part);
      }
      pending_any = false;
    }
    if (kDepthIntr) {
      if (pending_cf != 0xffffffffu) {
        exact_atomic_add_part_untracked(&ex.head_a[pending_cf], pending_cf_g, 0, ex.invalid);
        exact_atomic_add_part_untracked(&ex.head_a[pending_cf], pending_cf_g, 1, ex.invalid);
      }
      pending_cf = 0xffffffffu;
    }
  };
  for_each_candidate(
      num_kfs, [&](int k) { return sphere_may_project_item(in, kfs[k].pose.F, wb); },
      [&](int k) {
        const KfEntry& kf = kfs[k];
        PairGather pg;
        const bool visible = gather_and_associate(L, in, kf, gp, gn, tp, in_range, &pg);
        const bool pose_kf = kf_pose_is_unknown(L, k);
        const uint32_t base = kf_pose_index(L, k);
        float pp[6] = {0, 0, 0, 0, 0, 0};
        if (pose_kf) for (int c = 0; c < 6; ++c) pp[c] = p_[base + c];   // (wave-uniform address: scalar loads)
        flush_pending();
        if (!__any(visible)) return;
        float gpose[6] = {0, 0, 0, 0, 0, 0};
        float gi_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // this keyframe's terms of the global intrinsics entries
        float ad = 0.f;                                   // ... and of alpha_d
        if (visible) {
          PairTerms t;
          eval_pair_terms<kDepthIntr, kColorIntr>(L, in, kf, pg.a, pg.pix, pg.dw, gn, d1, d2, &t);
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
              pending_cf = head_index(L, t.cf_index);
              pending_cf_g = t.Jcf * sum;
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
        if (kIntr) {
          const float v[16] = {gpose[0], gpose[1], gpose[2], gpose[3], gpose[4], gpose[5], ad, gi_acc[0], gi_acc[1], gi_acc[2], gi_acc[3], gi_acc[4],
                               gi_acc[5], gi_acc[6], gi_acc[7], gi_acc[8]};
          pending = wave_reduce_small<16>(v, lane);   // lanes 4 j .. 4 j + 3 hold total j
        } else {
          const float v[8] = {gpose[0], gpose[1], gpose[2], gpose[3], gpose[4], gpose[5], ad, 0.f};
          pending = wave_reduce_small<8>(v, lane);    // lanes 8 j .. 8 j + 7 hold total j
        }
        pending_any = true;
        pending_pose = pose_kf;
        pending_base = pose_kf ? base : 0u;
      });
  flush_pending();

  if (in_range && L.optimize_geometry) {
    g_[gi] = gs[0];
    if (L.geom_stride == 3) { g_[gi + 1] = gs[1]; g_[gi + 2] = gs[2]; }
  }
}

template <bool kDepthIntr, bool kColorIntr>
__global__ void __launch_bounds__(kPcgSweepBlock) BAHIP_PCG_SWEEP_ATTR
pcg_step1_kernel(PcgLayout L, PcgExact ex, Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s,
                 const float* __restrict__ p_, float* __restrict__ g_, const PcgControl* ctl, const uint32_t* __restrict__ sched) {
  if (ctl->stop) return;
  uint32_t tile;   // heavy work first (wave_cull.h: scheduled_tile)
  if (!scheduled_tile(blockIdx.x, gridDim.x - (sched ? kHeavySlots : 0u), sched, &tile)) return;
  const PcgGlobalSink sink{ex, (int)(tile & (kHotReplicas - 1))};
  pcg_step1_tile<kDepthIntr, kColorIntr>(L, ex, in, kfs, num_kfs, s, p_, g_, tile, sink);
}

// Persistent form (round 4), like pose_accumulate_lds_kernel: one workgroup of 16 wavefronts per compute unit keeps the pose
// block of g's dense head -- 6 exact cells per pose unknown, 72 bytes each: 86 KB at 200 keyframes -- and the hot scalars in
// LDS; its wavefronts draw tiles (batches from a counter per XCD, single tiles from a word in LDS) and add their (tile,
// keyframe) totals there; the workgroup adds its non-zero limbs to the global accumulators once.  The one-tile-per-wavefront
// form issued 14 global 64-bit atomics per (tile, keyframe) with an association -- 12 M per inner step at the bench size, half a
// millisecond of the memory side's 23.6 G atomic requests per second behind a 0.8 ms sweep.  Integer sums: the same bits.
constexpr int kPcgLdsWaves = 16;
constexpr uint32_t kPcgLdsBatch = 32;
template <bool kDepthIntr, bool kColorIntr>
__global__ void __launch_bounds__(64 * kPcgLdsWaves) BAHIP_PCG_SWEEP_ATTR
pcg_step1_lds_kernel(PcgLayout L, PcgExact ex, Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s,
                     const float* __restrict__ p_, float* __restrict__ g_, const PcgControl* ctl, const uint32_t* __restrict__ sched,
                     uint32_t padded_tiles, uint32_t* __restrict__ tile_counters, int parity, uint32_t pose_cells) {
  extern __shared__ long long pcg_table[];   // ExactCell[pose_cells + kPcgLdsHotCells], then the batch word
  if (blockIdx.x == 0 && threadIdx.x < 8) tile_counters[(parity ^ 1) * 8 + threadIdx.x] = 0;
  if (ctl->stop) return;
  const int lane = threadIdx.x & 63;
  const uint32_t cells = pose_cells + kPcgLdsHotCells, words = cells * kExactLimbs;
  unsigned long long& batch_state = *reinterpret_cast<unsigned long long*>(pcg_table + words);
  const uint32_t xcd = blockIdx.x & 7u, per_xcd = sched_positions(padded_tiles, sched) >> 3;
  uint32_t* counter = tile_counters + parity * 8 + xcd;
  for (uint32_t e = threadIdx.x; e < words; e += blockDim.x) pcg_table[e] = 0;
  if (threadIdx.x == 0) batch_state = ((unsigned long long)atomicAdd(counter, kPcgLdsBatch) << 32) | ((unsigned long long)kPcgLdsBatch << 24);
  __syncthreads();
  const PcgLdsSink sink{ex, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) long long*)pcg_table, pose_cells};
  for (;;) {
    unsigned long long taken = 0;
    if (lane == 0) taken = atomicAdd(&batch_state, 1ull);
    const uint32_t first = __builtin_amdgcn_readfirstlane((uint32_t)(taken >> 32));
    const uint32_t size = __builtin_amdgcn_readfirstlane((uint32_t)taken >> 24);
    const uint32_t index = __builtin_amdgcn_readfirstlane((uint32_t)taken & 0xffffffu);
    if (first >= per_xcd) break;
    if (index < size) {
      uint32_t tile;
      if (first + index < per_xcd && scheduled_tile((first + index) * 8u + xcd, padded_tiles, sched, &tile))
        pcg_step1_tile<kDepthIntr, kColorIntr>(L, ex, in, kfs, num_kfs, s, p_, g_, tile, sink);
    } else if (index == size) {
      if (lane == 0) {
        const uint32_t left = per_xcd > first + size ? per_xcd - (first + size) : 0u;
        const uint32_t want = min(kPcgLdsBatch, max(2u, left / (4u * (gridDim.x >> 3))));
        const uint32_t next = atomicAdd(counter, want);
        __hip_atomic_store(&batch_state, ((unsigned long long)next << 32) | ((unsigned long long)want << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else {
      while ((uint32_t)(__hip_atomic_load(&batch_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32) == first)
        __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  const int replica = (int)(blockIdx.x & (kHotReplicas - 1));
  for (uint32_t e = threadIdx.x; e < words; e += blockDim.x) {
    const long long v = pcg_table[e];
    if (v != 0) {
      const uint32_t cell = e / kExactLimbs, limb = e - cell * kExactLimbs;
      if (cell >= pose_cells + 10u) continue;   // (padding)
      ExactCell* target = cell < pose_cells ? &ex.head_a[cell] : hot_cell(ex, pcg_lds_hot_slot(cell - pose_cells), replica);
      limb_atomic_add(&target->limb[limb], v);
    }
  }
}

BAHIP_FLAVOURED_END
#ifndef BAHIP_FAST_MATH   // exists once (the exact unit)
namespace bahip {
// PCGStep2 (B/kernel_pcg.cu:1117-1158)
__global__ void __launch_bounds__(kPcgBlock)
pcg_step2_kernel(PcgLayout L, PcgExact ex, float* __restrict__ r_, const float* __restrict__ M_, float* __restrict__ delta,
                 float* __restrict__ g_, const float* __restrict__ p_, const float* alpha_n, const float* alpha_d, const PcgControl* ctl) {
  if (ctl->stop) return;
  __shared__ BlockExact<1> acc;
  block_exact_clear(acc);
  const int replica = (int)(blockIdx.x & (kHotReplicas - 1));
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
    const float term = z_value * r_value;
    if (is_local(L, u)) block_exact_add(acc, 0, term, ex.invalid);
    else exact_atomic_add(hot_cell(ex, kHotDotHead, replica), term, ex.invalid);
  }
  const int slots[1] = {kHotDotLocal};
  block_exact_flush(acc, ex, slots);
}

// PCGStep3 (B/kernel_pcg.cu:1212-1226), and the epsilon terms of the next step's alpha_d from the new p
// (AddAlphaDEpsilonTerms, B/kernel_pcg.cu:1028-1050)
__global__ void __launch_bounds__(kPcgBlock)
pcg_step3_kernel(PcgLayout L, PcgExact ex, const float* __restrict__ g_, float* __restrict__ p_, const float* alpha_n, const float* beta_n,
                 const PcgControl* ctl) {
  if (ctl->stop) return;
  __shared__ BlockExact<1> acc;
  block_exact_clear(acc);
  const int replica = (int)(blockIdx.x & (kHotReplicas - 1));
  const float an = *alpha_n;
  const float beta = (an >= 1e-35f) ? (*beta_n / an) : 0;
  for (uint32_t u = blockIdx.x * kPcgBlock + threadIdx.x; u < L.unknown_count; u += gridDim.x * kPcgBlock) {
    const float pv = g_[u] + beta * p_[u];
    p_[u] = pv;
    const float term = (kDiagEpsilon + prior_at(L, u)) * pv * pv;
    if (is_local(L, u)) block_exact_add(acc, 0, term, ex.invalid);
    else exact_atomic_add(hot_cell(ex, kHotEpsHead, replica), term, ex.invalid);
  }
  const int slots[1] = {kHotEpsLocal};
  block_exact_flush(acc, ex, slots);
}
// the epsilon terms alone (per-stage entry points: a caller's PCGStep1 sees a p that no step 3 of ours produced)
__global__ void __launch_bounds__(kPcgBlock)
pcg_eps_terms_kernel(PcgLayout L, PcgExact ex, const float* __restrict__ p_) {
  __shared__ BlockExact<1> acc;
  block_exact_clear(acc);
  const int replica = (int)(blockIdx.x & (kHotReplicas - 1));
  for (uint32_t u = blockIdx.x * kPcgBlock + threadIdx.x; u < L.unknown_count; u += gridDim.x * kPcgBlock) {
    const float pv = p_[u];
    const float term = (kDiagEpsilon + prior_at(L, u)) * pv * pv;
    if (is_local(L, u)) block_exact_add(acc, 0, term, ex.invalid);
    else exact_atomic_add(hot_cell(ex, kHotEpsHead, replica), term, ex.invalid);
  }
  const int slots[1] = {kHotEpsLocal};
  block_exact_flush(acc, ex, slots);
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

// ---- test hook: the exact sum of n binary32 values through the same device code paths ------------------------------------------
// mode 0: every term with exact_atomic_add into one of 64 replicas; mode 1: through the per-thread columns in workgroup memory
// and block_exact_flush, as the per-unknown kernels do.  The result is the binary64 value fold_hot + exact_value produce.
__global__ void __launch_bounds__(kPcgBlock) exact_sum_debug_kernel(PcgExact ex, const float* __restrict__ v, size_t n, int mode) {
  __shared__ BlockExact<1> acc;
  block_exact_clear(acc);
  const int replica = (int)(blockIdx.x & (kHotReplicas - 1));
  for (size_t i = (size_t)blockIdx.x * kPcgBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kPcgBlock) {
    if (mode == 0) exact_atomic_add(hot_cell(ex, kHotDotHead, replica), v[i], ex.invalid);
    else block_exact_add(acc, 0, v[i], ex.invalid);
  }
  const int slots[1] = {kHotDotLocal};
  block_exact_flush(acc, ex, slots);
}
__global__ void __launch_bounds__(64) exact_sum_debug_resolve_kernel(PcgExact ex, double* out) {
  const double v = fold_hot(ex, kHotDotLocal, kHotDotHead, true);
  if (threadIdx.x == 0) *out = (*ex.invalid) ? __builtin_nan("") : v;
}
void launch_exact_sum_debug(hipStream_t st, const PcgExact& ex, const float* values, size_t n, int mode, double* out) {
  const unsigned blocks = n ? (unsigned)((n + kPcgBlock - 1) / kPcgBlock < 512 ? (n + kPcgBlock - 1) / kPcgBlock : 512) : 1u;
  hipLaunchKernelGGL(exact_sum_debug_kernel, dim3(blocks), dim3(kPcgBlock), 0, st, ex, values, n, mode);
  hipLaunchKernelGGL(exact_sum_debug_resolve_kernel, dim3(1), dim3(64), 0, st, ex, out);
}

}  // namespace bahip
#endif
namespace bahip {
// ---- launchers -----------------------------------------------------------------------------------------------
static inline unsigned gU(uint32_t n) { return (n + kPcgBlock - 1) / kPcgBlock; }
// The per-unknown kernels run a grid-stride loop over at most kPcgReduceBlocks workgroups: each ends in 9 atomics per sum.
constexpr unsigned kPcgReduceBlocks = 1024;
static inline unsigned gR(uint32_t n) { return gU(n) < kPcgReduceBlocks ? gU(n) : kPcgReduceBlocks; }

// whole XCD chunks, as in kernels_surfel.hip (xcd_chunked_tile)
static inline unsigned gS(uint32_t n) { return xcd_padded_tiles((n + kPcgSweepBlock - 1) / kPcgSweepBlock); }

}  // namespace bahip
#ifndef BAHIP_FAST_MATH   // exists once (the exact unit)
namespace bahip {
size_t pcg_exact_cells(uint32_t head_count) { return (size_t)kHotSlots * kHotReplicas + 2 * (size_t)head_count + 1; }
PcgExact pcg_exact_view(void* buffer, uint32_t head_count) {
  PcgExact ex;
  ExactCell* base = static_cast<ExactCell*>(buffer);
  ex.hot = base;
  ex.head_a = base + (size_t)kHotExchanged1 * kHotReplicas;
  ex.head_b = ex.head_a + head_count;
  ex.invalid = reinterpret_cast<unsigned*>(ex.head_b + head_count);   // a whole cell: the first word of what exchange 2 carries
  ex.hot_tail = ex.head_b + head_count + 1;
  return ex;
}

}  // namespace bahip
#endif
BAHIP_FLAVOURED_BEGIN
void launch_pcg_init(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                     const SurfelsView& s, float* r, float* M, uint32_t* tile_cost, const uint32_t* sched) {
  if (!s.size) return;
  const dim3 grid(sched_positions(gS(s.size), sched)), block(kPcgSweepBlock);
  const bool di = L.optimize_depth_intrinsics, ci = L.optimize_color_intrinsics;
  if (di && ci) hipLaunchKernelGGL((pcg_init_kernel<true, true>), grid, block, 0, st, L, ex, in, kfs, num_kfs, s, r, M, tile_cost, sched);
  else if (di) hipLaunchKernelGGL((pcg_init_kernel<true, false>), grid, block, 0, st, L, ex, in, kfs, num_kfs, s, r, M, tile_cost, sched);
  else if (ci) hipLaunchKernelGGL((pcg_init_kernel<false, true>), grid, block, 0, st, L, ex, in, kfs, num_kfs, s, r, M, tile_cost, sched);
  else hipLaunchKernelGGL((pcg_init_kernel<false, false>), grid, block, 0, st, L, ex, in, kfs, num_kfs, s, r, M, tile_cost, sched);
}
BAHIP_FLAVOURED_END
#ifndef BAHIP_FAST_MATH   // exists once (the exact unit)
namespace bahip {
static inline unsigned resolve_grid(const PcgLayout& L) { return gU(L.head_lo + (L.unknown_count - L.head_hi)) + 1; }
void launch_pcg_resolve_init(hipStream_t st, const PcgLayout& L, const PcgExact& ex, float* r, float* M) {
  hipLaunchKernelGGL((pcg_resolve_kernel<true>), dim3(resolve_grid(L)), dim3(kPcgBlock), 0, st, L, ex, r, M, nullptr, 0.0, nullptr);
}
void launch_pcg_resolve_step1(hipStream_t st, const PcgLayout& L, const PcgExact& ex, float* g, float* alpha_d, double eps_repeat, const void* ctl) {
  hipLaunchKernelGGL((pcg_resolve_kernel<false>), dim3(resolve_grid(L)), dim3(kPcgBlock), 0, st, L, ex, g, nullptr, alpha_d, eps_repeat,
                     static_cast<const PcgControl*>(ctl));
}
void launch_pcg_init2(hipStream_t st, const PcgLayout& L, const PcgExact& ex, float a, const float* r, const float* M, float* delta, float* g,
                      float* p) {
  if (L.unknown_count) hipLaunchKernelGGL(pcg_init2_kernel, dim3(gR(L.unknown_count)), dim3(kPcgBlock), 0, st, L, ex, a, r, M, delta, g, p);
}
void launch_pcg_control_init(hipStream_t st, const PcgExact& ex, void* ctl, float* alpha_n) {
  hipLaunchKernelGGL(pcg_control_init_kernel, dim3(1), dim3(64), 0, st, ex, static_cast<PcgControl*>(ctl), alpha_n);
}
void launch_pcg_control(hipStream_t st, const PcgExact& ex, void* ctl, float* beta_n) {
  hipLaunchKernelGGL(pcg_control_kernel, dim3(1), dim3(64), 0, st, ex, static_cast<PcgControl*>(ctl), beta_n);
}
size_t pcg_control_bytes() { return sizeof(PcgControl); }

}  // namespace bahip
#endif
BAHIP_FLAVOURED_BEGIN
static int g_pcg_lds_form = bahip_env_int("BAHIP_PCG_LDS", 1);   // 0: always the one-tile-per-wavefront form
void set_pcg_lds_form(int mode) { g_pcg_lds_form = (mode >= 0 && mode <= 2) ? mode : 1; }   // 2: also on grids that do not fill the chip (tests)
constexpr size_t kPcgLdsTableLimit = 128 * 1024;
template <bool kDepthIntr, bool kColorIntr>
static bool launch_pcg_step1_lds(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                                 const SurfelsView& s, const float* p, float* g, const PcgControl* ctl, const uint32_t* sched,
                                 uint32_t* tile_counters, int* parity_inout) {
  // the pose block of the unknown vector (and of the dense head: head index = unknown index there) ends where the next block begins
  uint32_t pose_end = L.unknown_count;
  if (L.optimize_geometry) pose_end = std::min(pose_end, L.surfel_start);
  if (L.optimize_depth_intrinsics) pose_end = std::min(pose_end, L.depth_intr_start);
  if (L.optimize_color_intrinsics) pose_end = std::min(pose_end, L.color_intr_start);
  const uint32_t pose_cells = L.optimize_poses ? pose_end : 0u;
  const size_t bytes = ((size_t)pose_cells + kPcgLdsHotCells) * sizeof(ExactCell) + sizeof(long long);
  if (bytes > kPcgLdsTableLimit) return false;
  static bool raised[64] = {}, failed[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!raised[dev] && !failed[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pcg_step1_lds_kernel<kDepthIntr, kColorIntr>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kPcgLdsTableLimit) == hipSuccess) raised[dev] = true;
    else { failed[dev] = true; (void)hipGetLastError(); }
  }
  if (failed[dev] && bytes > 64 * 1024) return false;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const unsigned tiles = gS(s.size);
  const unsigned grid = std::max(8u, std::min((unsigned)cus, ((sched_positions(tiles, sched) + kPcgLdsWaves - 1) / kPcgLdsWaves + 7u) & ~7u));
  const int parity = *parity_inout;
  *parity_inout = parity ^ 1;
  hipLaunchKernelGGL((pcg_step1_lds_kernel<kDepthIntr, kColorIntr>), dim3(grid), dim3(64 * kPcgLdsWaves), bytes, st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched,
                     tiles, tile_counters, parity, pose_cells);
  return true;
}

// launches of the step-1 sweep by form since the process started: [0] one tile per wavefront, [1] persistent with the pose block in LDS
static long long g_pcg_step1_form_launches[2] = {0, 0};
void pcg_step1_form_launches(long long out[2]) { out[0] = g_pcg_step1_form_launches[0]; out[1] = g_pcg_step1_form_launches[1]; }
void launch_pcg_step1(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                      const SurfelsView& s, const float* p, float* g, const void* ctl_, const uint32_t* sched, uint32_t* tile_counters,
                      int* parity_inout) {
  const PcgControl* ctl = static_cast<const PcgControl*>(ctl_);
  if (!s.size) return;
  const bool di = L.optimize_depth_intrinsics, ci = L.optimize_color_intrinsics;
  // the persistent LDS form when the grid fills the chip, the per-keyframe entry points (Route B) are not in play and the pose
  // block of the head fits the table
  if (tile_counters && parity_inout && L.single_keyframe < 0 && (g_pcg_lds_form == 2 || (g_pcg_lds_form == 1 && gS(s.size) >= 8192))) {
    bool launched;
    if (di && ci) launched = launch_pcg_step1_lds<true, true>(st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched, tile_counters, parity_inout);
    else if (di) launched = launch_pcg_step1_lds<true, false>(st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched, tile_counters, parity_inout);
    else if (ci) launched = launch_pcg_step1_lds<false, true>(st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched, tile_counters, parity_inout);
    else launched = launch_pcg_step1_lds<false, false>(st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched, tile_counters, parity_inout);
    if (launched) { ++g_pcg_step1_form_launches[1]; return; }
  }
  ++g_pcg_step1_form_launches[0];
  const dim3 grid(sched_positions(gS(s.size), sched)), block(kPcgSweepBlock);
  if (di && ci) hipLaunchKernelGGL((pcg_step1_kernel<true, true>), grid, block, 0, st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched);
  else if (di) hipLaunchKernelGGL((pcg_step1_kernel<true, false>), grid, block, 0, st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched);
  else if (ci) hipLaunchKernelGGL((pcg_step1_kernel<false, true>), grid, block, 0, st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched);
  else hipLaunchKernelGGL((pcg_step1_kernel<false, false>), grid, block, 0, st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched);
}
BAHIP_FLAVOURED_END
#ifndef BAHIP_FAST_MATH   // exists once (the exact unit)
namespace bahip {
void launch_pcg_eps_terms(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const float* p) {
  if (L.unknown_count) hipLaunchKernelGGL(pcg_eps_terms_kernel, dim3(gR(L.unknown_count)), dim3(kPcgBlock), 0, st, L, ex, p);
}
void launch_pcg_step2(hipStream_t st, const PcgLayout& L, const PcgExact& ex, float* r, const float* M, float* delta, float* g, const float* p,
                      const float* alpha_n, const float* alpha_d, const void* ctl) {
  if (L.unknown_count) hipLaunchKernelGGL(pcg_step2_kernel, dim3(gR(L.unknown_count)), dim3(kPcgBlock), 0, st, L, ex, r, M, delta, g, p, alpha_n, alpha_d,
                                          static_cast<const PcgControl*>(ctl));
}
void launch_pcg_step3(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const float* g, float* p, const float* alpha_n, const float* beta_n,
                      const void* ctl) {
  if (L.unknown_count) hipLaunchKernelGGL(pcg_step3_kernel, dim3(gR(L.unknown_count)), dim3(kPcgBlock), 0, st, L, ex, g, p, alpha_n, beta_n,
                                          static_cast<const PcgControl*>(ctl));
}
void launch_pcg_update_surfels(hipStream_t st, const PcgLayout& L, const SurfelsView& s, const float* delta) {
  if (s.size) hipLaunchKernelGGL(pcg_update_surfels_kernel, dim3(gU(s.size)), dim3(kPcgBlock), 0, st, L, s, delta);
}
void launch_pcg_update_cfactors(hipStream_t st, const Intrinsics& in, uint32_t start, const float* delta, float* cfactor, uint32_t pitch) {
  hipLaunchKernelGGL(pcg_update_cfactors_kernel, dim3(gU(in.cf_width * in.cf_height)), dim3(kPcgBlock), 0, st, in, start, delta, cfactor, pitch);
}


// dispatchers (ba_launch.h: "Two arithmetic flavours")
void launch_pcg_init(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                     const SurfelsView& s, float* r, float* M, uint32_t* tile_cost, const uint32_t* sched) {
  BAHIP_PICK(in, launch_pcg_init(st, L, ex, in, kfs, num_kfs, s, r, M, tile_cost, sched));
}
void launch_pcg_step1(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                      const SurfelsView& s, const float* p, float* g, const void* ctl, const uint32_t* sched, uint32_t* tile_counters,
                      int* parity_inout) {
  BAHIP_PICK(in, launch_pcg_step1(st, L, ex, in, kfs, num_kfs, s, p, g, ctl, sched, tile_counters, parity_inout));
}
void set_pcg_lds_form(int mode) { exact::set_pcg_lds_form(mode); fast::set_pcg_lds_form(mode); }
void pcg_step1_form_launches(long long out[2]) {
  long long a[2], b[2];
  exact::pcg_step1_form_launches(a); fast::pcg_step1_form_launches(b);
  out[0] = a[0] + b[0]; out[1] = a[1] + b[1];
}
}  // namespace bahip
#endif
