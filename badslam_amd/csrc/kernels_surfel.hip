// kernels_surfel.hip -- per-surfel stages of the alternating scheme: activation, normals, geometry.
//
// Reference structure (B/ = applications/badslam/src/badslam/): one kernel launch per keyframe
// per stage, every launch streaming all N surfels and doing read-modify-write on the accumulator
// rows of the surfel buffer (B/kernel_surfel_activation.cc:53-66, B/kernel_opt_geometry.cc:108-200).
// Here each stage is ONE launch: a thread owns a surfel, keeps position / normal / accumulators in
// registers and visits only the keyframes whose frustum can contain its wavefront's 64 surfels
// (wave_cull.h), so the surfel array is read once and written once per stage and both the K-fold
// RMW traffic and most of the K x N association tests disappear.
//
// Per-surfel sums over keyframes are DEFINED here as four interleaved partial sums: partial j adds the keyframes k
// with k % 4 == j in ascending order, and the total is ((p0 + p1) + p2) + p3 (the oracle computes exactly that; the
// reference adds the keyframes' contributions one launch after the other).  The definition lets the same bits come
// out of two launch shapes (tile_sums):
//   kWaves = 1  one wavefront per 64-surfel tile walks the four classes one after the other - least overhead, used
//               when there are enough tiles to fill the chip (a single-GPU run);
//   kWaves = 2, 4  a workgroup of two / four wavefronts per tile sharing the classes, partials combined through LDS -
//               divides the longest wavefront, which is what bounds the launch once the surfel set is a shard of a
//               multi-GPU run and no longer fills the chip.
//
// The same definition is what makes KEYFRAME sharding (capi_rccl.hip: bahip_context_set_keyframe_sharding) reproduce the unsharded
// bits: a rank that holds the images of whole classes only (keyframe k lives on rank (k % 4) % world, world = 2 | 4) computes
// exactly those classes' partials (kSumsProduce: stored to a buffer [class][sum][surfel] that is zero elsewhere), the ranks
// exchange the buffers as integer sums of bit patterns (x + 0 keeps every bit), and every rank then combines the four
// partials as defined (kSumsConsume).  The geometry step becomes three launches with two exchanges between them.
#include <stdlib.h>
#include <cstdio>

#include "ba_device.h"
#include "ba_launch.h"
#include "wave_cull.h"

namespace bahip {

#ifndef BAHIP_WAVES_ATTR
#define BAHIP_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4)))   // cap the allocation at 128 VGPRs: 4 waves per SIMD (5 spills: measured slower)
#endif
constexpr int kSurfelBlock = 64;   // one wavefront per workgroup (per-wave work varies with the keyframe candidates)
constexpr int kMaxSumClasses = 8;  // interleaved partial sums per surfel: Intrinsics::sum_classes = 4 or 8 (part of the numerical
                                   // definition, not a tuning knob; 8 exists for keyframe sharding over 8 ranks)

// tot[q] = ((p0[q] + p1[q]) + p2[q]) + p3[q] (... + p7[q]), p_c = what visit(acc, c) accumulates over the keyframes of class c
// (the keyframes whose index in the bound table is congruent to c modulo `classes`), c ascending.
// kWaves > 1: the calling workgroup has kWaves wavefronts holding the same 64 surfels (wavefront w takes the classes
// w, w + kWaves, ...); every thread must call.
enum SumsMode { kSumsFused = 0, kSumsProduce = 1, kSumsConsume = 2 };
__device__ __forceinline__ bool position_valid(Vec3 p) { return p.x == p.x; }   // deleted surfels carry NaN x
static inline unsigned grid_for(uint32_t n) { return xcd_padded_tiles((n + kSurfelBlock - 1) / kSurfelBlock); }   // whole XCD runs
#if defined(BAHIP_TILE_TIMELINE) && !defined(BAHIP_FAST_MATH)
// experiment build only (scripts/tile_timeline.py): when each tile of the LAST geometry launch started and ended (100 MHz clock)
__device__ unsigned long long g_geometry_timeline[65536][2];
void geometry_timeline_dump(const char* path) {
  static unsigned long long host[65536][2];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_geometry_timeline), sizeof(host)) != hipSuccess) return;
  if (FILE* f = fopen(path, "wb")) { fwrite(host, sizeof(host), 1, f); fclose(f); }
}
#define BAHIP_RECORD_GEOMETRY_TIMELINE 1
#endif
}  // namespace bahip

// ---- the sweeps: compiled once per arithmetic flavour (ba_launch.h) ---------------------------------------------------------------
BAHIP_FLAVOURED_BEGIN

template <int kWaves, int kCount, int kMode = kSumsFused, typename Visit>
__device__ __forceinline__ void tile_sums(float (&tot)[kCount], float* lds, Visit visit, int classes, const ClassPartials& cp = ClassPartials{},
                                          uint32_t i = 0, bool in_range = false) {
  static_assert(kWaves == 1 || kWaves == 2 || kWaves == 4, "1, 2 or 4 wavefronts per tile");
  static_assert(kMode == kSumsFused || kWaves == 1, "the class partials are exchanged by the one-wavefront shape only");
  if (kMode == kSumsProduce) {
#pragma nounroll
    for (int c = 0; c < classes; ++c) {
      if (!((cp.owned >> c) & 1u)) continue;   // another rank holds this class's images
      float acc[kCount];
#pragma unroll
      for (int q = 0; q < kCount; ++q) acc[q] = 0.f;
      visit(acc, c);
      if (in_range) {
#pragma unroll
        for (int q = 0; q < kCount; ++q) cp.data[(size_t)(c * kCount + q) * cp.stride + i] = acc[q];
      }
    }
#pragma unroll
    for (int q = 0; q < kCount; ++q) tot[q] = 0.f;
  } else if (kMode == kSumsConsume) {
#pragma unroll
    for (int q = 0; q < kCount; ++q) {
      float t = cp.data[(size_t)(0 * kCount + q) * cp.stride + i] + cp.data[(size_t)(1 * kCount + q) * cp.stride + i];
      for (int c = 2; c < classes; ++c) t += cp.data[(size_t)(c * kCount + q) * cp.stride + i];
      tot[q] = t;
    }
  } else if (kWaves == 1) {
#pragma unroll
    for (int q = 0; q < kCount; ++q) tot[q] = 0.f;   // +0 + p0 == p0 bit for bit (the partials are never -0)
#pragma nounroll
    for (int c = 0; c < classes; ++c) {
      float acc[kCount];
#pragma unroll
      for (int q = 0; q < kCount; ++q) acc[q] = 0.f;
      visit(acc, c);
#pragma unroll
      for (int q = 0; q < kCount; ++q) tot[q] += acc[q];
    }
  } else {
    const int part = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma nounroll
    for (int c = part; c < classes; c += kWaves) {
      float acc[kCount];
#pragma unroll
      for (int q = 0; q < kCount; ++q) acc[q] = 0.f;
      visit(acc, c);
#pragma unroll
      for (int q = 0; q < kCount; ++q) lds[(c * kCount + q) * 64 + lane] = acc[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kCount; ++q) {
      float t = lds[(0 * kCount + q) * 64 + lane] + lds[(1 * kCount + q) * 64 + lane];
      for (int c = 2; c < classes; ++c) t += lds[(c * kCount + q) * 64 + lane];
      tot[q] = t;
    }
    __syncthreads();   // the buffer is reused by the next call
  }
}

// B/kernel_surfel_activation.cu:38-94
__global__ void __launch_bounds__(kSurfelBlock) BAHIP_WAVES_ATTR
activation_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s, uint32_t surfels_size) {
  const uint32_t i = xcd_chunked_tile(blockIdx.x) * kSurfelBlock + threadIdx.x;
  const bool in_range = i < surfels_size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, in_range && position_valid(gp));
  bool active = false;
  // the reference stops testing a surfel once it is active (B/kernel_surfel_activation.cu:69-72); here the wavefront stops
  // visiting keyframes once every one of its surfels is
  for_each_candidate_until(
      num_kfs,
      [&](int k) { return kfs[k].activation == BAHIP_KF_ACTIVE && sphere_may_project(in, kfs[k].pose.F, wb); },
      [&](int k) {
        if (in_range && !active) {
          Assoc r;
          if (project_associate<false>(in, kfs[k].pose.F, kfs[k].geom, gp, gn, &r, nullptr)) active = true;
        }
        return __all(active || !in_range) != 0;
      });
  if (in_range) s.active[i] = (s.active[i] & (uint8_t)~kSurfelActiveFlag) | (active ? kSurfelActiveFlag : 0);
}

// Normals pass: B/kernel_opt_geometry.cu:82-101 (reset), :527-553 (accumulate), :577-597 (update).
// `live` = this lane holds an active surfel; every thread of the workgroup must call this function.
// kActivate: the surfel activation of B/kernel_surfel_activation.cu:38-94 is decided here instead of in a sweep of its
// own -- a surfel is active iff it is associated with >= 1 keyframe whose activation is kActive, and this pass tests
// exactly those associations (same position, same old normal) on its way over the non-inactive keyframes.  Lanes with
// `decide` set enter as live candidates, count their associations with kActive keyframes (a fifth sum), get their flag
// written and stay live only if that count is >= 1; what an inactive surfel accumulated is dropped.
// kMode (keyframe sharding, tile_sums): kSumsProduce stores this rank's class partials and returns; kSumsConsume takes the sums
// from the exchanged partials instead of visiting keyframes.
// Candidate masks of the normals pass, replayed by the position pass of the same tile (wave_cull.h: for_each_candidate_cached): per
// class kMaskChunks chunks of 64 keyframes (4 classes: 1024 keyframes; beyond that the position pass tests again).
constexpr int kMaskChunks = 4;
template <int kWaves, bool kActivate, int kMode = kSumsFused>
__device__ __forceinline__ void normals_pass(const Intrinsics& in, const KfEntry* __restrict__ kfs, int num_kfs,
                                             const WaveBounds& wb, SurfelsView& s, uint32_t i, bool* live_inout, bool decide,
                                             Vec3 gp, Vec3* gn_inout, float* lds, const ClassPartials& cp = ClassPartials{},
                                             bool in_range = false, unsigned long long* masks = nullptr) {
  const bool writer = kWaves == 1 || (threadIdx.x >> 6) == 0;
  const Vec3 gn = *gn_inout;
  bool live = *live_inout;
  constexpr int kCount = kActivate ? 5 : 4;
  float sum[kCount];   // x, y, z, count [, count over kActive keyframes]
  tile_sums<kWaves, kCount, kMode>(sum, lds, [&](float (&acc)[kCount], int cls) {
    struct Started { Projected p; PixelWords pix; };
    for_each_candidate_pipelined<Started>(
        num_kfs,
        [&](int k) {
          float f[12];
          int32_t activation;
          load_candidate(kfs[k].pose.F, &kfs[k].activation, f, &activation);
          return activation != BAHIP_KF_INACTIVE && sphere_may_project(in, f, wb);
        },
        [&](int k) {   // candidate k + 1's gathers go out before candidate k's words are waited for (wave_cull.h)
          Started st;
          st.p = project_surfel(in, kfs[k].pose.F, gp);
          st.pix = load_pixel_words(in, kfs[k].geom, st.p);   // both gathers in flight at once (ba_device.h)
          return st;
        },
        [&](int k, const Started& st) {
          const Projected& p = st.p;
          const PixelWords& pix = st.pix;
          Assoc r;
          const bool associated = live && associate_from_words<false>(in, kfs[k].pose.F, gn, p, pix, &r, nullptr);
          gathers_arrived(pix);
          if (associated) {
            const Vec3 g = mul33(kfs[k].pose.GR, unpack_normal8(r.normal_bits));
            acc[0] += g.x; acc[1] += g.y; acc[2] += g.z; acc[3] += 1.f;
            if (kActivate) acc[kCount - 1] += (kfs[k].activation == BAHIP_KF_ACTIVE) ? 1.f : 0.f;
          }
        },
        in.sum_classes, cls, masks ? masks + cls * kMaskChunks : nullptr, masks ? kMaskChunks : 0, false);
  }, in.sum_classes, cp, i, in_range);
  if (kMode == kSumsProduce) return;
  if (kActivate && decide) {
    const bool active = live && sum[kCount - 1] >= 1.f;
    if (writer) s.active[i] = (s.active[i] & (uint8_t)~kSurfelActiveFlag) | (active ? kSurfelActiveFlag : 0);
    live = active;
    *live_inout = active;
  }
  if (!live) return;
  const float sx = sum[0], sy = sum[1], sz = sum[2], count = sum[3];
  // (The reference parks these sums in accum rows 0..3 between its kernels, B/kernel_opt_geometry.cu:527-597; here they live
  // in registers and are not stored: rows 8..16 are scratch, nothing reads them after the step, and until round 3 writing
  // them -- with the nine sums of the position pass -- tripled the sweep's write traffic: 77 instead of 25 bytes per surfel.)
  if (count >= 1) {
    const uint32_t packed = pack_normal10((1.f / count) * mk3(sx, sy, sz));
    if (writer) reinterpret_cast<uint32_t*>(s.row(kSurfelNormal))[i] = packed;
    *gn_inout = unpack_normal10(packed);
  }
}

template <int kWaves>
__global__ void __launch_bounds__(64 * kWaves) BAHIP_WAVES_ATTR
normals_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s) {
  __shared__ float lds[kWaves == 1 ? 1 : kMaxSumClasses * 4 * 64];
  const int lane = threadIdx.x & 63;
  const uint32_t i = xcd_chunked_tile(blockIdx.x) * kSurfelBlock + lane;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  bool live = in_range && (s.active[ii] & kSurfelActiveFlag);
  const Vec3 gp = surfel_position(s, ii);
  Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, live && position_valid(gp));
  normals_pass<kWaves, false>(in, kfs, num_kfs, wb, s, ii, &live, false, gp, &gn, lds);
}

// Geometry step of one BA iteration for one surfel: normals, then either the depth-only 1x1 solve
// (B/kernel_opt_geometry.cu:417-508) or the joint position + descriptor 3x3 solve (:119-353).
// kActivate: surfels [0, activate_count) get their activation flag decided inside the normals pass (see normals_pass) --
// UpdateSurfelActivationCUDA and OptimizeGeometryIterationCUDA of one BA iteration in one sweep; surfels beyond
// activate_count keep the flag they have (the reference flags newly created surfels active without a test,
// B/direct_ba_alternating.cc:448-452).
//
// kPhase (keyframe sharding; 0 = the whole step in one launch): the step cut where the per-surfel sums over ALL keyframes are
// needed, so that the ranks can exchange their class partials in between:
//   1  normals pass over this rank's classes -> partials `cpn`;
//   2  normals (and activation) finished from the exchanged `cpn`, then the position pass over this rank's classes -> `cpp`;
//   3  position / descriptor solve from the exchanged `cpp`.
// Every rank holds all surfels and ends each phase with the same bits (flags and normals after 2, positions after 3).
template <bool kUseDepth, bool kUseDesc, int kWaves, bool kActivate, int kPhase>
__device__ __forceinline__ void geometry_step(const Intrinsics& in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView& s,
                                              uint32_t activate_count, float* lds, const ClassPartials& cpn, const ClassPartials& cpp,
                                              uint32_t tile /* workgroup-uniform (kWaves > 1) or wave-uniform */, unsigned long long* replay_masks) {
  constexpr int kNormalsMode = kPhase == 1 ? kSumsProduce : kPhase == 2 ? kSumsConsume : kSumsFused;
  constexpr int kPositionMode = kPhase == 2 ? kSumsProduce : kPhase == 3 ? kSumsConsume : kSumsFused;
  const int lane = threadIdx.x & 63;
  const bool writer = kWaves == 1 || (threadIdx.x >> 6) == 0;
  const uint32_t i = tile * kSurfelBlock + lane;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  // (phase 3: the flags were decided by phase 2)
  const bool decide = kActivate && kPhase != 3 && in_range && i < activate_count;
  bool live = in_range && (decide || (s.active[ii] & kSurfelActiveFlag));
  const Vec3 gp = surfel_position(s, ii);
  Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, live && position_valid(gp));
  if (wb.r < 0.f) {   // workgroup-uniform (all wavefronts of the tile hold the same surfels): nothing a keyframe could see
    if (decide && writer && kPhase != 1) s.active[ii] = s.active[ii] & (uint8_t)~kSurfelActiveFlag;   // (deleted surfels: never active)
    return;
  }
  // one wavefront per tile, the whole step in one launch: the position pass replays the candidate masks of the normals pass
  constexpr bool kReplayMasks = kWaves == 1 && kPhase == 0;   // (measured: geometry sweep 0.749 -> 0.731 ms at the bench size, profiles/r5_ab_*.txt)
  unsigned long long* masks = kReplayMasks ? replay_masks : nullptr;   // kMaxSumClasses * kMaskChunks words of LDS per wavefront, owned by the kernel
  if (kPhase != 3) normals_pass<kWaves, kActivate, kNormalsMode>(in, kfs, num_kfs, wb, s, ii, &live, decide, gp, &gn, lds, cpn, in_range, masks);
  if (kPhase == 1) return;

  auto cand = [&](int k) {
    float f[12];
    int32_t activation;
    load_candidate(kfs[k].pose.F, &kfs[k].activation, f, &activation);
    return activation != BAHIP_KF_INACTIVE && sphere_may_project(in, f, wb);
  };

  if (!kUseDesc) {
    float hb[2];
    tile_sums<kWaves, 2, kPositionMode>(hb, lds, [&](float (&acc)[2], int cls) {
      for_each_candidate_cached(num_kfs, cand, [&](int k) {
        const Projected p = project_surfel(in, kfs[k].pose.F, gp);
        const PixelWords pix = load_pixel_words(in, kfs[k].geom, p);
        Assoc r;
        const bool associated = live && associate_from_words<false>(in, kfs[k].pose.F, gn, p, pix, &r, nullptr);
        gathers_arrived(pix);
        if (!associated) return;
        const float inv_std = assoc_inv_std(in, r);
        const float jac = -inv_std;
        const Vec3 u = assoc_unproject(r);
        const float raw = inv_std * dot3(r.nl, u - r.local);
        const float w = depth_residual_weight(raw);
        const float wj = w * jac;
        acc[0] = mad(wj, jac, acc[0]);
        acc[1] = mad(wj, raw, acc[1]);
      }, in.sum_classes, cls, masks ? masks + cls * kMaskChunks : nullptr, masks ? kMaskChunks : 0, masks != nullptr);
    }, in.sum_classes, cpp, ii, in_range);
    if (kPhase == 2 || !live || !writer) return;
    const float H = hb[0], b = hb[1];
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
  float tot[8];   // a0 a1 a2 a3 a5 a6 a7 a8 of B/kernel_opt_geometry.cu:119-230 (a4 = H12 is exactly 0)
  tile_sums<kWaves, 8, kPositionMode>(tot, lds, [&](float (&acc)[8], int cls) {
    float &a0 = acc[0], &a1 = acc[1], &a2 = acc[2], &a3 = acc[3], &a5 = acc[4], &a6 = acc[5], &a7 = acc[6], &a8 = acc[7];
    // Two-stage candidate loop for the pixel word (wave_cull.h: for_each_candidate_pipelined): candidate k + 1 is projected and its
    // geometry word requested before candidate k is worked on; the footprint gathers stay where they were, behind the start of the
    // candidate's own turn (the state they would add -- six more registers -- spills).  One-wavefront form only: the four-wavefront
    // form of small clouds has no registers left for the state either.
    struct StartedPair { Projected p; PixelWords pix; };
    auto start_pair = [&](int k) {
      StartedPair st;
      st.p = project_surfel(in, kfs[k].pose.F, gp);
      st.pix = load_pixel_words(in, kfs[k].geom, st.p);
      return st;
    };
    auto finish_pair = [&](int k, const StartedPair& st) {
      const float* F = kfs[k].pose.F;
      const Projected& p = st.p;
      const PixelWords& pix = st.pix;
      const DescWords dw = load_descriptor_words(in, kfs[k].lumafp, F, tp, p);
      Assoc r;
      const bool associated = live && associate_from_words<false>(in, F, gn, p, pix, &r, nullptr);
#ifdef BAHIP_PIN_POSITION
      gathers_arrived(pix, dw);   // measured: -9 % on this pass (the footprint gathers of pairs that fail the association cost more than the second round trip saves here)
#endif
      if (!associated) return;
      if (kUseDepth) {
        const float inv_std = assoc_inv_std(in, r);
        const float jac = -inv_std;
        const Vec3 u = assoc_unproject(r);
        const float raw = inv_std * dot3(r.nl, u - r.local);
        const float w = depth_residual_weight(raw);
        a0 = mad(w * jac, jac, a0);
        a6 = mad(w * raw, jac, a6);
      }
      if (dw.color_ok) {
        DescEval e;
        eval_descriptor_from_words(in, kfs[k].lumafp, dw, d1, d2, &e);
        const float jp1 = jac_descriptor_surfel(r.nl, r.local, r.inv_z, e.gx1, e.gy1, in.cfx, in.cfy);
        const float jp2 = jac_descriptor_surfel(r.nl, r.local, r.inv_z, e.gx2, e.gy2, in.cfx, in.cfy);
        const float jd = -1.f;
        const float w1 = descriptor_residual_weight(e.r1);
        const float wr1 = w1 * e.r1;
        const float w2 = descriptor_residual_weight(e.r2);
        const float wr2 = w2 * e.r2;
        a0 = mad(w2 * jp2, jp2, mad(w1 * jp1, jp1, a0));
        a1 += w1 * jp1 * jd;
        a3 += w1 * jd * jd;
        a6 = mad(wr2, jp2, mad(wr1, jp1, a6));
        a7 += wr1 * jd;
        a2 += w2 * jp2 * jd;
        a5 += w2 * jd * jd;
        a8 += wr2 * jd;
      }
    };
    if (kWaves == 1)
      for_each_candidate_pipelined<StartedPair>(num_kfs, cand, start_pair, finish_pair, in.sum_classes, cls, masks ? masks + cls * kMaskChunks : nullptr,
                                                masks ? kMaskChunks : 0, masks != nullptr);
    else
      for_each_candidate_cached(num_kfs, cand, [&](int k) { finish_pair(k, start_pair(k)); }, in.sum_classes, cls,
                                masks ? masks + cls * kMaskChunks : nullptr, masks ? kMaskChunks : 0, masks != nullptr);
  }, in.sum_classes, cpp, ii, in_range);
  if (kPhase == 2 || !live || !writer) return;
  const float a0 = tot[0], a1 = tot[1], a2 = tot[2], a3 = tot[3], a5 = tot[4], a6 = tot[5], a7 = tot[6], a8 = tot[7];

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


template <bool kUseDepth, bool kUseDesc, int kWaves, bool kActivate>
__global__ void __launch_bounds__(64 * kWaves) BAHIP_WAVES_ATTR
geometry_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s, uint32_t activate_count,
                const uint32_t* __restrict__ sched, const int* __restrict__ stop /* device-driven BA loop (capi_ba.hip:
                bahip_alternating_iterations): a launch queued behind the iteration that ended the loop does nothing; NULL: always runs */) {
  __shared__ float lds[kWaves == 1 ? 1 : kMaxSumClasses * 8 * 64];
  __shared__ unsigned long long candidate_masks[kWaves == 1 ? kMaxSumClasses * kMaskChunks : 1];
  if (stop && load_global(stop) != 0) return;
#ifdef BAHIP_RECORD_GEOMETRY_TIMELINE
  const unsigned long long t0 = wall_clock64();
#endif
  uint32_t tile;   // heavy work first (wave_cull.h: scheduled_tile); workgroup-uniform
  if (!scheduled_tile(blockIdx.x, gridDim.x - (sched ? kHeavySlots : 0u), sched, &tile)) return;
  geometry_step<kUseDepth, kUseDesc, kWaves, kActivate, 0>(in, kfs, num_kfs, s, activate_count, lds, ClassPartials{}, ClassPartials{}, tile, candidate_masks);
#ifdef BAHIP_RECORD_GEOMETRY_TIMELINE
  if (threadIdx.x == 0 && blockIdx.x < 65536) { g_geometry_timeline[blockIdx.x][0] = t0; g_geometry_timeline[blockIdx.x][1] = wall_clock64(); }
#endif
}

// The HYBRID shape (round 6), for clouds whose tiles do not fill the chip several times over (a shard of a multi-GPU run): workgroups
// of four wavefronts; the first kHeavySlots workgroups take one tile of the schedule's heavy list each, a keyframe class per wavefront
// (the four-wavefront shape), every other workgroup takes FOUR regular tiles, a whole tile per wavefront (the one-wavefront shape with
// its replayed candidate masks and its two-stage candidate loop).  With four wavefronts on every tile a light tile pays the set-up and
// the LDS combination four times (93 wavefront-us per tile against 60, profiles/r6_shard_experiments.txt); with one wavefront on every
// tile the launch lasts as long as its heaviest tile.  Same bits as either shape: the sums are defined per class.
#ifndef BAHIP_HYBRID_WAVES
#define BAHIP_HYBRID_WAVES 4   // both shapes in one kernel: 128 VGPRs + 17 spilled (60 bytes of scratch per lane); 3 = no spills but three wavefronts
                               // per SIMD: geometry stage of the emulated 8-rank share 0.145 / 0.144 / 0.148 ms against 0.155 / 0.155 / 0.156 (call 28)
#endif
template <bool kUseDepth, bool kUseDesc, bool kActivate>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BAHIP_HYBRID_WAVES)))
geometry_hybrid_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s, uint32_t activate_count,
                       const uint32_t* __restrict__ sched /* not NULL */, uint32_t padded_tiles, const int* __restrict__ stop) {
  __shared__ float lds[kMaxSumClasses * 8 * 64];
  __shared__ unsigned long long candidate_masks[4][kMaxSumClasses * kMaskChunks];
  if (stop && load_global(stop) != 0) return;
  if (blockIdx.x < kHeavySlots) {
    if (blockIdx.x >= sched[0]) return;
    geometry_step<kUseDepth, kUseDesc, 4, kActivate, 0>(in, kfs, num_kfs, s, activate_count, lds, ClassPartials{}, ClassPartials{}, sched[8 + blockIdx.x], nullptr);
  } else {
    const uint32_t wave = threadIdx.x >> 6, position = (blockIdx.x - kHeavySlots) * 4u + wave;
    if (position >= padded_tiles) return;
    const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)sched[kSchedPerm + position]);
    if (sched[kSchedPerm + padded_tiles + tile] != 0) return;   // on the heavy list
    geometry_step<kUseDepth, kUseDesc, 1, kActivate, 0>(in, kfs, num_kfs, s, activate_count, nullptr, ClassPartials{}, ClassPartials{}, tile, candidate_masks[wave]);
  }
}

// One phase of the keyframe-sharded geometry step (geometry_step: kPhase).
template <bool kUseDepth, bool kUseDesc, bool kActivate, int kPhase>
__global__ void __launch_bounds__(64) BAHIP_WAVES_ATTR
geometry_phase_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s, uint32_t activate_count,
                      ClassPartials cpn, ClassPartials cpp) {
  geometry_step<kUseDepth, kUseDesc, 1, kActivate, kPhase>(in, kfs, num_kfs, s, activate_count, nullptr, cpn, cpp, xcd_run_tile(blockIdx.x, gridDim.x), nullptr);   // (buffer order)
}

// Surfel activation under keyframe sharding: this rank's kActive keyframes only; hits[i] = 1 where one of them sees surfel i
// (the words are summed over the ranks as integers, then activation_from_hits_kernel sets the flags).
__global__ void __launch_bounds__(kSurfelBlock) BAHIP_WAVES_ATTR
activation_hits_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s, uint32_t surfels_size,
                       uint32_t owner_mask, uint32_t owner_rank, uint32_t* __restrict__ hits) {
  const uint32_t i = xcd_chunked_tile(blockIdx.x) * kSurfelBlock + threadIdx.x;
  const bool in_range = i < surfels_size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, in_range && position_valid(gp));
  bool active = false;
  for_each_candidate_until(
      num_kfs,
      [&](int k) {
        return ((uint32_t)k & owner_mask) == owner_rank && kfs[k].activation == BAHIP_KF_ACTIVE && sphere_may_project(in, kfs[k].pose.F, wb);
      },
      [&](int k) {
        if (in_range && !active) {
          Assoc r;
          if (project_associate<false>(in, kfs[k].pose.F, kfs[k].geom, gp, gn, &r, nullptr)) active = true;
        }
        return __all(active || !in_range) != 0;
      });
  if (in_range) hits[i] = active ? 1u : 0u;
}
// ---- launchers ---------------------------------------------------------------------------------
void launch_activation(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                       uint32_t surfels_size) {
  if (surfels_size == 0) return;
  hipLaunchKernelGGL(activation_kernel, dim3(grid_for(surfels_size)), dim3(kSurfelBlock), 0, stream, in, kfs, num_kfs, s,
                     surfels_size);
}

// Launch shape of the normals / geometry passes (see the header comment): one wavefront per tile when the tiles alone
// fill the chip several times over, else four.  Results do not depend on it; BAHIP_TILE_WAVES=1|4 or
// bahip_debug_set_launch_shapes force one (tests run both).
static int g_forced_tile_waves = bahip_env_int("BAHIP_TILE_WAVES", 0);
static long long g_geometry_hybrid_launches = 0;
long long geometry_hybrid_launches() { return g_geometry_hybrid_launches; }
void set_tile_waves(int waves) { g_forced_tile_waves = waves; }
static int tile_waves(uint32_t surfels) {
  const int forced = g_forced_tile_waves;
  if (forced == 1 || forced == 4) return forced;
  // measured on MI355X (geometry pass, ms): 11.7 k tiles: 0.90 with one wavefront per tile, 0.95 with four;
  // 5.9 k tiles: 0.65 vs 0.51
  return grid_for(surfels) >= 8192 ? 1 : 4;
}

void launch_normals(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s) {
  if (s.size == 0) return;
  const dim3 grid(grid_for(s.size));
  switch (tile_waves(s.size)) {
    case 1: hipLaunchKernelGGL(normals_kernel<1>, grid, dim3(64), 0, stream, in, kfs, num_kfs, s); break;
    default: hipLaunchKernelGGL(normals_kernel<4>, grid, dim3(256), 0, stream, in, kfs, num_kfs, s); break;
  }
}

template <int kWaves, bool kActivate>
static void launch_geometry_shape(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs,
                                  int num_kfs, const SurfelsView& s, uint32_t activate_count, const uint32_t* sched, const int* stop) {
  const dim3 grid(sched_positions(grid_for(s.size), sched)), block(64 * kWaves);
  if (!use_desc) hipLaunchKernelGGL((geometry_kernel<true, false, kWaves, kActivate>), grid, block, 0, stream, in, kfs, num_kfs, s, activate_count, sched, stop);
  else if (use_depth) hipLaunchKernelGGL((geometry_kernel<true, true, kWaves, kActivate>), grid, block, 0, stream, in, kfs, num_kfs, s, activate_count, sched, stop);
  else hipLaunchKernelGGL((geometry_kernel<false, true, kWaves, kActivate>), grid, block, 0, stream, in, kfs, num_kfs, s, activate_count, sched, stop);
}

template <bool kActivate>
static void launch_geometry_hybrid(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs,
                                   int num_kfs, const SurfelsView& s, uint32_t activate_count, const uint32_t* sched, const int* stop) {
  const uint32_t padded = grid_for(s.size);
  const dim3 grid(kHeavySlots + (padded + 3u) / 4u), block(256);
  if (!use_desc) hipLaunchKernelGGL((geometry_hybrid_kernel<true, false, kActivate>), grid, block, 0, stream, in, kfs, num_kfs, s, activate_count, sched, padded, stop);
  else if (use_depth) hipLaunchKernelGGL((geometry_hybrid_kernel<true, true, kActivate>), grid, block, 0, stream, in, kfs, num_kfs, s, activate_count, sched, padded, stop);
  else hipLaunchKernelGGL((geometry_hybrid_kernel<false, true, kActivate>), grid, block, 0, stream, in, kfs, num_kfs, s, activate_count, sched, padded, stop);
}

// activate_count < 0: the activation flags are taken as they are; >= 0: surfels [0, activate_count) are (re)activated first.
// `sched`: the schedule of a grid of grid_for(s.size) tiles (wave_cull.h: scheduled_tile), or NULL.
void launch_geometry(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs,
                     int num_kfs, const SurfelsView& s, long long activate_count, const uint32_t* sched, const int* stop) {
  if (s.size == 0) return;
  const uint32_t n = activate_count < 0 ? 0u : (uint32_t)activate_count;
  // few tiles and a schedule (its heavy list): the hybrid shape; BAHIP_TILE_WAVES / bahip_debug_set_launch_shapes: 5 forces it (where a
  // schedule exists), 1 / 4 the uniform shapes
  const bool hybrid = sched != nullptr && (g_forced_tile_waves == 5 || (g_forced_tile_waves == 0 && grid_for(s.size) < 8192));
  if (hybrid) {
    ++g_geometry_hybrid_launches;
    if (activate_count < 0) launch_geometry_hybrid<false>(stream, use_depth, use_desc, in, kfs, num_kfs, s, n, sched, stop);
    else launch_geometry_hybrid<true>(stream, use_depth, use_desc, in, kfs, num_kfs, s, n, sched, stop);
  } else if (tile_waves(s.size) == 1) {
    if (activate_count < 0) launch_geometry_shape<1, false>(stream, use_depth, use_desc, in, kfs, num_kfs, s, n, sched, stop);
    else launch_geometry_shape<1, true>(stream, use_depth, use_desc, in, kfs, num_kfs, s, n, sched, stop);
  } else {
    if (activate_count < 0) launch_geometry_shape<4, false>(stream, use_depth, use_desc, in, kfs, num_kfs, s, n, sched, stop);
    else launch_geometry_shape<4, true>(stream, use_depth, use_desc, in, kfs, num_kfs, s, n, sched, stop);
  }
}

// ---- keyframe sharding ------------------------------------------------------------------------------------------------
template <bool kUseDepth, bool kUseDesc, bool kActivate>
static void launch_geometry_phase_of(hipStream_t stream, int phase, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                                     uint32_t activate_count, const ClassPartials& cpn, const ClassPartials& cpp) {
  const dim3 grid(grid_for(s.size)), block(64);
  if (phase == 1) hipLaunchKernelGGL((geometry_phase_kernel<kUseDepth, kUseDesc, kActivate, 1>), grid, block, 0, stream, in, kfs, num_kfs, s, activate_count, cpn, cpp);
  else if (phase == 2) hipLaunchKernelGGL((geometry_phase_kernel<kUseDepth, kUseDesc, kActivate, 2>), grid, block, 0, stream, in, kfs, num_kfs, s, activate_count, cpn, cpp);
  else hipLaunchKernelGGL((geometry_phase_kernel<kUseDepth, kUseDesc, kActivate, 3>), grid, block, 0, stream, in, kfs, num_kfs, s, activate_count, cpn, cpp);
}

// One phase (1, 2, 3: geometry_step) of the keyframe-sharded geometry step; the caller clears the partials a phase produces
// beforehand and sums them over the ranks afterwards.
void launch_geometry_phase(hipStream_t stream, int phase, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                           const SurfelsView& s, long long activate_count, const ClassPartials& cpn, const ClassPartials& cpp) {
  if (s.size == 0) return;
  const uint32_t n = activate_count < 0 ? 0u : (uint32_t)activate_count;
  const bool act = activate_count >= 0;
  if (!use_desc) {
    if (act) launch_geometry_phase_of<true, false, true>(stream, phase, in, kfs, num_kfs, s, n, cpn, cpp);
    else launch_geometry_phase_of<true, false, false>(stream, phase, in, kfs, num_kfs, s, n, cpn, cpp);
  } else if (use_depth) {
    if (act) launch_geometry_phase_of<true, true, true>(stream, phase, in, kfs, num_kfs, s, n, cpn, cpp);
    else launch_geometry_phase_of<true, true, false>(stream, phase, in, kfs, num_kfs, s, n, cpn, cpp);
  } else {
    if (act) launch_geometry_phase_of<false, true, true>(stream, phase, in, kfs, num_kfs, s, n, cpn, cpp);
    else launch_geometry_phase_of<false, true, false>(stream, phase, in, kfs, num_kfs, s, n, cpn, cpp);
  }
}

void launch_activation_hits(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s, uint32_t surfels_size,
                            int kf_rank, int kf_world, uint32_t* hits) {
  if (surfels_size == 0) return;
  hipLaunchKernelGGL(activation_hits_kernel, dim3(grid_for(surfels_size)), dim3(kSurfelBlock), 0, stream, in, kfs, num_kfs, s, surfels_size,
                     (uint32_t)(kf_world - 1), (uint32_t)kf_rank, hits);
}
BAHIP_FLAVOURED_END

// ---- what exists once (the exact unit): colour assignment, diagnostics, and the dispatchers carrying the public names -------------
#ifndef BAHIP_FAST_MATH
namespace bahip {
// ---- DirectBA::AssignColors (B/kernel_assign_colors.cu:41-125, B/kernel_assign_colors.cc:39-80) --------------------------
// Bilinear RGBA sample at unnormalised coordinates, clamp addressing, texel centres at +0.5 (the tex2D<float4> of the
// reference on the keyframe's pitch-linear uchar4 image); per channel the arithmetic of sample_luma (oracle: orc_sample_rgba).
__device__ __forceinline__ void sample_rgba(const uint8_t* color, uint32_t pitch, int w, int h, float x, float y, float (&out)[4]) {
  float xb = x - 0.5f, yb = y - 0.5f;
  if (!(xb >= -1.f)) xb = -1.f;
  if (xb > (float)w) xb = (float)w;
  if (!(yb >= -1.f)) yb = -1.f;
  if (yb > (float)h) yb = (float)h;
  const float fx = floorf(xb), fy = floorf(yb);
  const float a = xb - fx, b = yb - fy;
  const int x0 = min(max((int)fx, 0), w - 1), x1 = min(max((int)fx + 1, 0), w - 1);
  const int y0 = min(max((int)fy, 0), h - 1), y1 = min(max((int)fy + 1, 0), h - 1);
  const uchar4* image = reinterpret_cast<const uchar4*>(color);
  const uchar4 tl = pitched_load(image, pitch, y0, x0), tr = pitched_load(image, pitch, y0, x1);
  const uchar4 bl = pitched_load(image, pitch, y1, x0), br = pitched_load(image, pitch, y1, x1);
  const uint8_t ctl[4] = {tl.x, tl.y, tl.z, tl.w}, ctr[4] = {tr.x, tr.y, tr.z, tr.w};
  const uint8_t cbl[4] = {bl.x, bl.y, bl.z, bl.w}, cbr[4] = {br.x, br.y, br.z, br.w};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float vtl = (float)ctl[c] * (1.0f / 255.0f), vtr = (float)ctr[c] * (1.0f / 255.0f);
    const float vbl = (float)cbl[c] * (1.0f / 255.0f), vbr = (float)cbr[c] * (1.0f / 255.0f);
    const float top = mad(a, vtr - vtl, vtl);
    const float bot = mad(a, vbr - vbl, vbl);
    out[c] = mad(b, bot - top, top);
  }
}

// Every keyframe a surfel is associated with (whatever its activation) contributes the RGBA sample at the surfel's colour
// pixel, in keyframe order; the mean, rounded, becomes the surfel colour.  Surfels no keyframe sees keep theirs.  The
// reference parks count and sums in accumulator rows 0-4 between its K + 2 launches; here they live in registers.
__global__ void __launch_bounds__(kSurfelBlock) BAHIP_WAVES_ATTR
assign_colors_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s) {
  const uint32_t i = xcd_chunked_tile(blockIdx.x) * kSurfelBlock + threadIdx.x;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, in_range && position_valid(gp));
  float count = 0.f, sum[4] = {0.f, 0.f, 0.f, 0.f};
  for_each_candidate(
      num_kfs, [&](int k) { return sphere_may_project_item(in, kfs[k].pose.F, wb); },
      [&](int k) {
        if (!in_range) return;
        Assoc r;
        if (!project_associate<false>(in, kfs[k].pose.F, kfs[k].geom, gp, gn, &r, nullptr)) return;
        float cx, cy;
        if (!depth_to_color_pixel(in, r.pxx, r.pxy, &cx, &cy)) return;
        float c[4];
        sample_rgba(kfs[k].color, kfs[k].color_pitch, in.cwidth, in.cheight, cx, cy, c);
        count += 1.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) sum[q] += c[q];
      });
  if (in_range && count > 0.f) {
    uchar4 out;
    out.x = (uint8_t)(255.f * sum[0] / count + 0.5f);
    out.y = (uint8_t)(255.f * sum[1] / count + 0.5f);
    out.z = (uint8_t)(255.f * sum[2] / count + 0.5f);
    out.w = (uint8_t)(255.f * sum[3] / count + 0.5f);
    reinterpret_cast<uchar4*>(s.row(kSurfelColor))[i] = out;
  }
}

__global__ void activation_from_hits_kernel(SurfelsView s, uint32_t surfels_size, const uint32_t* __restrict__ hits) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < surfels_size) s.active[i] = (s.active[i] & (uint8_t)~kSurfelActiveFlag) | (hits[i] ? kSurfelActiveFlag : 0);
}

void launch_assign_colors(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s) {
  if (s.size == 0) return;
  hipLaunchKernelGGL(assign_colors_kernel, dim3(grid_for(s.size)), dim3(kSurfelBlock), 0, stream, in, kfs, num_kfs, s);
}

int geometry_normals_sums(bool activate) { return activate ? 5 : 4; }
int geometry_position_sums(bool use_desc) { return use_desc ? 8 : 2; }
void launch_activation_from_hits(hipStream_t stream, const SurfelsView& s, uint32_t surfels_size, const uint32_t* hits) {
  if (surfels_size == 0) return;
  hipLaunchKernelGGL(activation_from_hits_kernel, dim3((surfels_size + 255) / 256), dim3(256), 0, stream, s, surfels_size, hits);
}


// dispatchers (ba_launch.h: "Two arithmetic flavours")
void launch_activation(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s, uint32_t surfels_size) {
  BAHIP_PICK(in, launch_activation(stream, in, kfs, num_kfs, s, surfels_size));
}
void launch_normals(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s) {
  BAHIP_PICK(in, launch_normals(stream, in, kfs, num_kfs, s));
}
void launch_geometry(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                     long long activate_count, const uint32_t* sched, const int* stop) {
  BAHIP_PICK(in, launch_geometry(stream, use_depth, use_desc, in, kfs, num_kfs, s, activate_count, sched, stop));
}
void launch_geometry_phase(hipStream_t stream, int phase, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                           const SurfelsView& s, long long activate_count, const ClassPartials& cpn, const ClassPartials& cpp) {
  BAHIP_PICK(in, launch_geometry_phase(stream, phase, use_depth, use_desc, in, kfs, num_kfs, s, activate_count, cpn, cpp));
}
void launch_activation_hits(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s, uint32_t surfels_size,
                            int kf_rank, int kf_world, uint32_t* hits) {
  BAHIP_PICK(in, launch_activation_hits(stream, in, kfs, num_kfs, s, surfels_size, kf_rank, kf_world, hits));
}
void set_tile_waves(int waves) { exact::set_tile_waves(waves); fast::set_tile_waves(waves); }
long long geometry_hybrid_launches() { return exact::geometry_hybrid_launches() + fast::geometry_hybrid_launches(); }
}  // namespace bahip

// ---- diagnostics: how much work does a sweep over all keyframes contain? ----------------------------
// counts[0] = (wave, keyframe) candidates after frustum culling, [1] = of those with >= 1 associated
// lane, [2] = associated (surfel, keyframe) pairs, [3] = pairs that passed the in-image test.
namespace bahip {
__global__ void __launch_bounds__(kSurfelBlock) BAHIP_WAVES_ATTR
count_pairs_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s, unsigned long long* counts) {
  const uint32_t i = xcd_chunked_tile(blockIdx.x) * kSurfelBlock + threadIdx.x;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const WaveBounds wb = wave_bounds(gp, in_range && position_valid(gp));
  unsigned long long cand = 0, wave_hits = 0, lane_hits = 0, in_image = 0;
  for_each_candidate(
      num_kfs, [&](int k) { return sphere_may_project_item(in, kfs[k].pose.F, wb); },
      [&](int k) {
        Assoc r;
        const bool hit = in_range && project_associate<false>(in, kfs[k].pose.F, kfs[k].geom, gp, gn, &r, nullptr);
        const Vec3 l = transform34(kfs[k].pose.F, gp);
        const float px = mad(in.fx, l.x * (1.f / l.z), in.cx), py = mad(in.fy, l.y * (1.f / l.z), in.cy);
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
#endif   // !BAHIP_FAST_MATH
