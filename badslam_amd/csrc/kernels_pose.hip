// kernels_pose.hip -- pose normal equations and the batched Gauss-Newton pose solve.
//
// Reference (B/ = applications/badslam/src/badslam/): for every keyframe and every GN step the
// host clears H/b with 2-4 tiny kernels, launches AccumulatePoseEstimationCoeffsCUDAKernel over
// all surfels (B/kernel_opt_pose.cu:251-383; 81 serial CUB block reductions per block,
// B/gauss_newton.cuh:46-93), copies 27 floats back, synchronises, and solves the 6x6 system with
// Eigen on the host (B/kernel_opt_pose.cc:67-96, B/direct_ba_alternating.cc:126-244).
//
// Here one launch per GN *round* handles every keyframe that is still iterating: a thread owns a
// surfel (position, normal, radius, descriptors in registers) and loops over the work items; the
// three residuals of a (surfel, keyframe) pair are folded into one per-lane 27-vector which is
// reduced across the wave64 in a fixed tree with cross-lane adds, converted to fixed point and merged
// with one 64-bit integer atomic per scalar per wave (a DEFINED, order-free sum: ba_device.h, HbFixed).
// The 6x6 LDLT (binary64, like the reference), T <- T*exp(-x), the convergence test and the activation
// update of the BA loop run in a second tiny kernel on the device, which publishes its counters to
// mapped host memory: a round costs two launches and no stream synchronisation.
#include <stdlib.h>

#include "ba_device.h"
#include <cstdio>
#include "ba_launch.h"
#include "se3_device.h"
#include "wave_cull.h"
#include "wave_reduce.h"

namespace bahip {
#ifndef BAHIP_WAVES_ATTR
#define BAHIP_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(4)))   // cap the allocation at 128 VGPRs: 4 waves per SIMD (5 spills: measured slower)
#endif
constexpr int kPoseBlock = 64;    // one wavefront per workgroup: no LDS, no barriers, and finished waves free their slot at once
// Where a sweep reports a tile total it could not represent (hb_split): the low word of the unused 28th coefficient of work item
// 0's row -- inside the buffer the ranks of a sharded run exchange, so that after the exchange EVERY rank's solve launch sees the
// flag and every host fails the call alike (a flag in the rank's own counter record let the other ranks run on).
__host__ __device__ inline int* pose_invalid_word(HbFixed* Hb) { return reinterpret_cast<int*>(Hb + 27 * kHbLimbs); }
}  // namespace bahip

// ---- the accumulate sweep: compiled once per arithmetic flavour (ba_launch.h) -----------------------------------------------------
BAHIP_FLAVOURED_BEGIN

// acc += w * [upper(J J^T) | r J], as fused multiply-add chains (the oracle's orc_accumulate_pose_coeffs spells the same chain:
// the per-lane sums, the wave tree and the fixed-point totals are part of the numerical definition, ba_device.h: HbFixed).
// (Two entries per instruction with v_pk_fma_f32 -- entries (r, c) (r, c + 1) for even c, 45 VALU instructions fewer per pair,
// bit-identical -- was measured in round 3: the pose sweep got 3 % SLOWER.  gfx950 issues a plain wave64 binary32 instruction
// in about half the cycles of a packed one, so packing buys nothing here and costs register-pair alignment.)
__device__ __forceinline__ void accumulate_jtj(float (&acc)[28], const float (&J)[6], float wgt, float raw) {
  int q = 0;
#pragma unroll
  for (int row = 0; row < 6; ++row) {
    const float wj = wgt * J[row];
#pragma unroll
    for (int col = row; col < 6; ++col, ++q) acc[q] = __builtin_fmaf(wj, J[col], acc[q]);
  }
  const float wr = wgt * raw;
#pragma unroll
  for (int c = 0; c < 6; ++c) acc[21 + c] = __builtin_fmaf(wr, J[c], acc[21 + c]);
}

// tile_bounds: one bounding sphere per 64-surfel tile.  The first Gauss-Newton round of a pose phase (stored_bounds == 0)
// computes them from the positions and stores them; the later rounds -- which iterate only the few keyframes that have not
// converged, yet launch one wavefront (or several) per tile all the same -- read the sphere back (one scalar load), test it
// against the remaining work items and return at once if none can see the tile, before any surfel is loaded.  Positions do
// not change between the rounds of a phase, so the stored sphere is the one that would be recomputed.
#ifdef BAHIP_COUNT_CANDIDATES
// experiment build only: (surfel tile, keyframe) candidates the cull lets through / with at least one associated lane / lanes
__device__ unsigned long long g_candidate_counters[4];
void pose_counters_dump() {
  unsigned long long c[4];
  if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_candidate_counters), sizeof(c)) == hipSuccess)
    fprintf(stderr, "candidates %llu  with-any-association %llu  associated lanes %llu\n", c[0], c[1], c[2]);
}
#endif
#ifdef BAHIP_TILE_TIMELINE
// experiment build only: start / end of every tile of the last FULL round of the persistent pose sweep, by draw position
__device__ unsigned long long g_pose_timeline[65536][4];   // start, end, tile, (candidates visited << 32) | candidates with an association
__device__ unsigned int g_pose_tile_stats[65536][2];
void pose_timeline_dump(const char* path) {
  static unsigned long long host[65536][4];
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pose_timeline), sizeof(host)) != hipSuccess) return;
  if (FILE* f = fopen(path, "wb")) { fwrite(host, sizeof(host), 1, f); fclose(f); }
}
#endif
// Where the 27 tile totals of a (tile, work item) pair go.
//
// GlobalSink: two 64-bit integer atomics per total on the Hb buffer.  They are issued one candidate late, behind the next
// candidate's gathers: vmcnt counts loads, stores and atomics alike and retires them in order, so a wavefront that issues its
// atomics and then the next gathers cannot use the gathered words before the atomics have been acknowledged by the memory
// side -- two round trips per candidate.  Held back (one VGPR: every lane holds the tile total of one slot), the
// acknowledgement has a whole candidate's arithmetic to arrive in.  As asm statements: the compiler's waitcnt pass otherwise
// waits for the (non-returning) atomics at the top of the next candidate.  Not tracking them is safe: vmcnt retires in order,
// so an untracked older operation can only make a later s_waitcnt vmcnt(N) wait for more than the compiler intended, never
// for less.  (Mnemonic of the gfx9 family, validated on gfx950; later families call it global_atomic_add_u64.)
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__))
#error "GlobalSink::flush spells the gfx9 family's 64-bit integer atomics (global_atomic_add_x2 / _sub_x2): this backend is built for gfx950"
#endif
struct GlobalSink {
  HbFixed* Hb;
  int* invalid;
  float pending;
  int pending_w;     // wave-uniform
  int slot_offset;   // byte offset of this lane's coefficient in a work item's row, or -1 (lane holds no total)
  __device__ __forceinline__ void flush() {
    if (pending_w >= 0 && slot_offset >= 0) {
      const HbFixed* row = Hb + (size_t)pending_w * kHbStride;   // wave-uniform: a scalar base, the lane adds its 32-bit offset
      // magnitudes added or subtracted according to the total's sign (ba_device.h: hb_split_magnitudes): the same sums
      const HbMagnitudes m = hb_split_magnitudes(pending);
      const unsigned long long lo = m.lo, hi = ((unsigned long long)m.hi_hi << 32) | m.hi_lo;
      if (m.valid) {
        if (m.negative) {
          if (lo) asm volatile("global_atomic_sub_x2 %0, %1, %2" ::"v"(slot_offset), "v"(lo), "s"(row) : "memory");
          if (hi) asm volatile("global_atomic_sub_x2 %0, %1, %2 offset:8" ::"v"(slot_offset), "v"(hi), "s"(row) : "memory");
        } else {
          if (lo) asm volatile("global_atomic_add_x2 %0, %1, %2" ::"v"(slot_offset), "v"(lo), "s"(row) : "memory");
          if (hi) asm volatile("global_atomic_add_x2 %0, %1, %2 offset:8" ::"v"(slot_offset), "v"(hi), "s"(row) : "memory");
        }
      } else {
        atomicOr(invalid, 1);
      }
    }
    pending_w = -1;
  }
  __device__ __forceinline__ void gathers_done() { flush(); }
  __device__ __forceinline__ void add(int /*item*/, int w, float total) {
    pending = total; pending_w = w;
  }
  __device__ __forceinline__ void finish() { flush(); }
};
// LdsSink: the workgroup's own table of limb pairs in LDS, one row per work item; ds_add_u64 costs no memory round trip and
// does not take part in vmcnt, so the totals are added at once.  The workgroup flushes the table to Hb when it runs out of
// tiles (pose_accumulate_lds_kernel): per launch 256 workgroups x K x 27 x 2 global atomics instead of 2 per total of every
// (tile, work item) pair -- 25 M memory-side atomics and 97 MB of write traffic per launch at the bench size (profiles/r2_e).
struct LdsSink {
  uint32_t lane_offset;   // LDS byte address of this lane's coefficient in row 0 of the table [num_items][kHbCoefficients][kHbLimbs]
  int* invalid;
  bool holds_total;       // this lane holds one of the 27 tile totals
  float pending;
  int pending_item;       // wave-uniform
  // The totals are added right behind the reduction (add() flushes at once; BAHIP_LDS_DEFERRED restores the GlobalSink shape,
  // where they wait in `pending` for the next candidate).  History: the round-3 builds gave wrong sums for rows > 0 in the immediate
  // shape and the oracle's bits in the deferred one.  The round-4 bisect (profiles/r4_lds_anomaly_bisect.txt) reproduced that on the
  // old commits, showed that the sink and the LDS atomics were innocent -- a shadow table kept by global atomics held the same wrong
  // values, add() was called once per (tile, item): the values REACHING the sink were already wrong, i.e. code generation upstream
  // of it -- and that every build from commit b4280f8 on is right in both shapes.  The instruction at fault was not pinned down;
  // tests/test_gpu_scale_parity.py runs the sweep at 1 / 2 / 16 wavefronts x 1 / 37 / 292 items x parts x rounds-ahead against the
  // oracle so that a toolchain or source change that brings it back cannot pass.
  __device__ __forceinline__ void flush() {
    if (pending_item >= 0 && holds_total) {
      // magnitudes added or subtracted according to the total's sign (ba_device.h: hb_split_magnitudes): the same sums as the
      // signed limbs of hb_split, without its four exponent ranges and without the 64-bit negations
      const HbMagnitudes m = hb_split_magnitudes(pending);
      const uint32_t address = lane_offset + (uint32_t)pending_item * (uint32_t)(kHbStride * sizeof(HbFixed));
      auto* cell = reinterpret_cast<__attribute__((address_space(3))) HbFixed*>(address);
      const HbFixed lo = (HbFixed)(unsigned long long)m.lo, hi = (HbFixed)(((unsigned long long)m.hi_hi << 32) | m.hi_lo);
      if (m.valid) {
        if (m.negative) {
          if (lo) __hip_atomic_fetch_sub(cell, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (hi) __hip_atomic_fetch_sub(cell + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          if (lo) __hip_atomic_fetch_add(cell, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (hi) __hip_atomic_fetch_add(cell + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
        atomicOr(invalid, 1);
      }
    }
    pending_item = -1;
  }
  __device__ __forceinline__ void gathers_done() { flush(); }
  __device__ __forceinline__ void add(int item, int /*w*/, float total) {
    pending = total; pending_item = item;
#ifndef BAHIP_LDS_DEFERRED
    flush();
#endif
  }
  __device__ __forceinline__ void finish() { flush(); }
};

// One 64-surfel tile against the work items still iterating (every `parts`-th of them, starting at `part`).
template <bool kUseDepth, bool kUseDesc, typename Sink>
__device__ __forceinline__ void pose_tile(const Intrinsics& in, const KfEntry* __restrict__ frames, const PoseWork* __restrict__ work,
                                          int num_work, const SurfelsView& s, WaveBounds* __restrict__ tile_bounds, int stored_bounds,
                                          int num_listed, uint32_t tile, int parts, int part, Sink& sink, uint32_t* __restrict__ tile_cost,
                                          int item_begin = 0, int item_count = -1) {
  const int lane = threadIdx.x & 63;
  // Later rounds (stored_bounds): the items are the num_listed entries of the list behind the counter records -- the work
  // items still iterating -- instead of all num_work work items.  A launch may cover a slice of the items only
  // ([item_begin, item_begin + item_count): the LDS form cuts lists longer than its table); `item` below counts from the
  // slice's start.
  const int* __restrict__ listed = reinterpret_cast<const int*>(work + num_work + kPoseTailRecords) + item_begin;
  const int num_items = item_count >= 0 ? item_count : (stored_bounds ? num_listed : num_work);
  auto work_item_of = [&](int item) { return stored_bounds ? load_global(listed + item) : item_begin + item; };
  WaveBounds wb;
  if (stored_bounds) {
    wb = tile_bounds[tile];     // wave-uniform address: scalar loads
    if (wb.r < 0.f) return;
    bool any = false;
    for (int base = part; base < num_items && !any; base += 64 * parts) {
      const int item = base + lane * parts;
      bool sees = false;
      if (item < num_items) {
        float f[12];
        load_candidate(work[work_item_of(item)].F, nullptr, f, nullptr);
        sees = sphere_may_project(in, f, wb);
      }
      any = __any(sees) != 0;
    }
    if (!any) return;
  }
  const uint32_t i = tile * kPoseBlock + lane;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  float radius_sq = 0, d1 = 0, d2 = 0;
  if (kUseDesc) {
    radius_sq = s.row(kSurfelRadiusSquared)[ii];
    d1 = s.row(kSurfelDescriptor1)[ii];
    d2 = s.row(kSurfelDescriptor2)[ii];
  }
  const TangentPoints tp = surfel_tangent_points(gp, gn, radius_sq);

  // Only the work items whose frustum can contain this wavefront's surfels are visited (wave_cull.h).
  if (!stored_bounds) {
    wb = wave_bounds(gp, in_range && (gp.x == gp.x));
    if (part == 0 && lane == 0) tile_bounds[tile] = wb;
  }
  int my_w = 0;   // the work item of this lane's candidate item in the current chunk of 64 (read back by lane in the body)
#ifdef BAHIP_TILE_TIMELINE
  uint32_t with_association = 0;
#endif
  uint32_t visited = 0;   // wave-uniform: candidates this wavefront visits = what the tile costs (feeds tile_order_kernel)
  for_each_candidate(
      num_items,
      [&](int item) {
        const int w = work_item_of(item);
        my_w = w;
        float f[12];
        int32_t done;
        load_candidate(work[w].F, &work[w].skip, f, &done);   // skip = done, or the keyframe's images live on another rank
        return !done && sphere_may_project(in, f, wb);
      },
      [&](int item) {
    // item = base + lane' * parts for the lane' that tested it: its w comes from that lane's register, not from memory
    // (a list lookup here was a dependent load and a wait at the top of every candidate of the later rounds)
    const int w = __builtin_amdgcn_readfirstlane(stored_bounds ? __builtin_amdgcn_readlane(my_w, ((item - part) / parts) & 63) : item_begin + item);
    const float* F = work[w].F;
    ++visited;
    // (the entry of the frame table that provides the images: today every writer binds work item w to entry w, but the field is what
    // pose_solve honours too, and reading frames[w] directly measured no gain over this dependent scalar load: ADVICE r5)
    const KfEntry& kf = frames[__builtin_amdgcn_readfirstlane(load_global(&work[w].kf_index))];
    // every gather of the pair goes out before the first one is waited for (ba_device.h: project_surfel)
    const Projected p = project_surfel(in, F, gp);
    const PixelWords pix = load_pixel_words(in, kf.geom, p);
    DescWords dw;
    if (kUseDesc) dw = load_descriptor_words(in, kf.lumafp, F, tp, p);   // (issued after the association instead: pose sweep +7 %)
    Assoc r;
    const bool visible = in_range && associate_from_words<false>(in, F, gn, p, pix, &r, nullptr);
    // every gather has arrived from here on, on every path (a load still pending at the loop's back edge would make the
    // compiler wait for it - and, vmcnt being in-order, for the atomics behind it - at the top of the next candidate)
    if (kUseDesc) gathers_arrived(pix, dw);
    else gathers_arrived(pix);
    sink.gathers_done();
#ifdef BAHIP_COUNT_CANDIDATES
    {
      const unsigned long long associated_lanes = __ballot(visible);
      if (!stored_bounds && lane == 0) {
        atomicAdd(&g_candidate_counters[0], 1ull);
        atomicAdd(&g_candidate_counters[1], associated_lanes ? 1ull : 0ull);
        atomicAdd(&g_candidate_counters[2], (unsigned long long)__popcll(associated_lanes));
      }
    }
#endif
    if (__builtin_amdgcn_ballot_w64(visible) == 0ull) return;   // (a scalar compare on the mask; __any costs a select and a compare per lane)
#ifdef BAHIP_TILE_TIMELINE
    ++with_association;
#endif

    float acc[28];   // 21 H + 6 b + 1 pad (kHbCoefficients)
    // The depth residual opens the 27 sums as plain products, in every lane: a lane without an association computes on whatever its
    // Assoc holds (no memory is touched), and eight selects then give it J = 0, weight 0, residual 0, so that its products are
    // zeros -- instead of 27 cleared accumulators + 27 fused multiply-adds onto them under the lanes' mask.  Same bits where it
    // matters: fma(a, b, +0) and a * b differ in the sign of a zero product only, a zero addend never changes a non-zero sum, and a
    // total that is a zero of either sign is the fixed-point 0 (hb_split).  (Not applicable to the lanes' garbage directly: their
    // Jacobians may be infinite or NaN, and 0 * inf is not 0.)
    if (kUseDepth) {
      float J[6];
      const float inv_std = assoc_inv_std(in, r);
      const Vec3 u = assoc_unproject(r);
      float raw = inv_std * dot3(r.nl, u - r.local);
      jac_depth_pose(r.nl, u, inv_std, J);
      float wgt = depth_residual_weight(raw);
#pragma unroll
      for (int c = 0; c < 6; ++c) J[c] = visible ? J[c] : 0.f;
      wgt = visible ? wgt : 0.f;
      raw = visible ? raw : 0.f;
      int q = 0;
#pragma unroll
      for (int row = 0; row < 6; ++row) {
        const float wj = wgt * J[row];
#pragma unroll
        for (int col = row; col < 6; ++col, ++q) acc[q] = wj * J[col];
      }
      const float wr = wgt * raw;
#pragma unroll
      for (int c = 0; c < 6; ++c) acc[21 + c] = wr * J[c];
      acc[27] = 0.f;
    } else {
#pragma unroll
      for (int q = 0; q < 28; ++q) acc[q] = 0.f;
    }

    if (visible) {
      float J[6];
      if (kUseDesc) {
        // B/kernel_opt_pose.cu:303-353: nothing is added when the colour-pixel transform fails.
        if (dw.color_ok) {
          DescEval e;
          eval_descriptor_from_words(in, kf.lumafp, dw, d1, d2, &e);
          // B/kernel_opt_pose.cu:96-142
          const Vec3 ls = r.local;
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float gx = (t ? e.gx2 : e.gx1) * in.cfx;
            const float gy = (t ? e.gy2 : e.gy1) * in.cfy;
            const float raw = t ? e.r2 : e.r1;
            jac_descriptor_pose(ls, r.inv_z, gx, gy, J);
            const float wgt = descriptor_residual_weight(raw);
            accumulate_jtj(acc, J, wgt, raw);
          }
        }
      }
    }

    // wave64 halving reduction (wave_reduce.h: a fixed tree over the 64 lanes), then two 64-bit integer adds per scalar
    // per wave on the fixed-point limbs: order-free, hence deterministic
    float total = wave_reduce28(acc, lane);
    // The cross-lane steps of the reduction must run with every lane enabled: a DPP / permlane source lane that EXEC has
    // switched off reads as zero.  The totals are consumed under `this lane holds one` (27 lanes), and the compiler sank the
    // last DPP add into that branch -- rows of the LDS table then received totals that lacked their odd neighbour's half.
    // The empty asm pins the finished value in uniform control flow.
    asm volatile("" : "+v"(total));
    sink.add(item, w, total);
  }, parts, part);
  sink.finish();
  // first round of a phase over the keyframe table: the tile's cost for the run order of the next launches (wave_cull.h)
  if (tile_cost && !stored_bounds && lane == 0 && visited) atomicAdd(&tile_cost[tile], visited);
#ifdef BAHIP_TILE_TIMELINE
  if (!stored_bounds && lane == 0 && tile < 65536) { g_pose_tile_stats[tile][0] = visited; g_pose_tile_stats[tile][1] = with_association; }
#endif
}

// One wavefront per workgroup and tile; gridDim.y wavefronts share a tile's work items (shards of a multi-GPU run, and work
// item tables too large for the LDS form below).
template <bool kUseDepth, bool kUseDesc>
__global__ void __launch_bounds__(kPoseBlock) BAHIP_WAVES_ATTR
pose_accumulate_kernel(Intrinsics in, const KfEntry* __restrict__ frames, const PoseWork* __restrict__ work,
                       int num_work, SurfelsView s, HbFixed* __restrict__ Hb, WaveBounds* __restrict__ tile_bounds, int stored_bounds,
                       int num_listed, int* __restrict__ invalid /* counter word kPoseCounterInvalid; its own argument so that `work`
                       stays read-only to the compiler: the per-candidate pose rows are then scalar loads */,
                       uint32_t* __restrict__ tile_cost, const uint32_t* __restrict__ sched,
                       const int* __restrict__ listed_count /* a later round queued before the host knew how many work items are
                       left (run_pose_rounds): the count the previous round's solve kernel left on the device; 0 = nothing to do */,
                       const int* __restrict__ stop /* device-driven BA loop: non-zero = nothing to do */) {
  const int lane = threadIdx.x & 63;
  if (stop && __builtin_amdgcn_readfirstlane(load_global(stop)) != 0) return;
  if (listed_count) {
    num_listed = __builtin_amdgcn_readfirstlane(load_global(listed_count));
    if (num_listed == 0) return;
  }
  const int slot = wave_reduce28_slot(lane);
  GlobalSink sink{Hb, invalid, 0.f, -1, (slot >= 0 && slot < 27) ? slot * kHbLimbs * (int)sizeof(HbFixed) : -1};
  uint32_t tile;   // heavy work first (wave_cull.h: scheduled_tile)
  if (!scheduled_tile(blockIdx.x, gridDim.x - (sched ? kHeavySlots : 0u), sched, &tile)) return;
  pose_tile<kUseDepth, kUseDesc>(in, frames, work, num_work, s, tile_bounds, stored_bounds, num_listed, tile, (int)gridDim.y, (int)blockIdx.y, sink,
                                 tile_cost);
}

// Persistent form: one workgroup of 16 wavefronts per compute unit keeps the normal equations of every work item in LDS
// (K x 28 x 2 limbs: 90 KB at 200 keyframes; gfx950 has 160 KB per CU).  Its wavefronts draw tiles instead of owning one
// each, so a wavefront with many candidate keyframes does not hold the others back (one tile per wavefront of a multi-wave
// workgroup did: -8 %, round 1).  Two levels: the workgroup takes batches of kPoseBatch tiles from a counter in global memory
// -- one per XCD, in the order of xcd_chunked_tile, so the spatial runs still meet in one L2 -- and its wavefronts take single
// tiles of the batch from a word in LDS (wave-level draws from the global counters were tried first: 5 860 returning atomics
// per address per launch serialise, 0.5 ms for a round with three work items left).  `tile_counters`: two sets of 8; a
// launch draws from set `parity` and clears the other for the next launch.
#ifndef BAHIP_POSE_LDS_WAVES
#define BAHIP_POSE_LDS_WAVES 16
#endif
constexpr int kPoseLdsWaves = BAHIP_POSE_LDS_WAVES;
#ifndef BAHIP_POSE_BATCH
#define BAHIP_POSE_BATCH 32
#endif
constexpr uint32_t kPoseBatch = BAHIP_POSE_BATCH;
constexpr int kPoseShortList = 16;   // later rounds with at most this many work items: static deal of the tiles (below)
// kSlice: the launch covers the items [slice_begin, slice_begin + slice_count) only -- a list longer than the table (292 work
// items) is cut into slices, one launch each (1000 keyframes: four); without it the two arguments are not looked at, which
// keeps them out of the scalar registers of the common case.
template <bool kUseDepth, bool kUseDesc, bool kSlice>
__global__ void __launch_bounds__(64 * kPoseLdsWaves) BAHIP_WAVES_ATTR
pose_accumulate_lds_kernel(Intrinsics in, const KfEntry* __restrict__ frames, const PoseWork* __restrict__ work,
                           int num_work, SurfelsView s, HbFixed* __restrict__ Hb, WaveBounds* __restrict__ tile_bounds, int stored_bounds,
                           int num_listed, int* __restrict__ invalid, uint32_t padded_tiles, uint32_t* __restrict__ tile_counters, int parity,
                           int slice_begin, int slice_count, uint32_t* __restrict__ tile_cost, const uint32_t* __restrict__ sched,
                           const int* __restrict__ listed_count /* as in pose_accumulate_kernel */,
                           uint32_t parts_shift /* 2^parts_shift wavefronts share a tile's work items: the unit a wavefront draws is
                           (tile, part) -- small grids (a shard of a multi-GPU run) otherwise last as long as their longest tile */,
                           const int* __restrict__ stop /* as in pose_accumulate_kernel */) {
  extern __shared__ HbFixed table[];
  const int lane = threadIdx.x & 63;
  // (both sets of tile counters are in a defined state after every launch, also one that finds nothing to do)
  if (blockIdx.x == 0 && threadIdx.x < 8) tile_counters[(parity ^ 1) * 8 + threadIdx.x] = 0;
  if (stop && __builtin_amdgcn_readfirstlane(load_global(stop)) != 0) return;
  if (listed_count) {
    num_listed = __builtin_amdgcn_readfirstlane(load_global(listed_count));
    if (num_listed == 0) return;
  }
  const int item_begin = kSlice ? slice_begin : 0;
  const int num_items = kSlice ? slice_count : (stored_bounds ? num_listed : num_work);
  // (first position of the current batch << 32) | (its size << 24) | positions of it already taken; the word behind the table (a
  // separate __shared__ variable next to the dynamic array was placed ON the array by this toolchain).  Batches shrink towards
  // the end of the XCD's queue (guided self-scheduling: a quarter of a workgroup's fair share of what is left, 2 .. kPoseBatch
  // positions): a workgroup's LAST batch is what the launch waits for, and 32 positions are two rounds of its 16 wavefronts.
  unsigned long long& batch_state = *reinterpret_cast<unsigned long long*>(table + (size_t)num_items * kHbStride);
  // units of an XCD's queue: positions (heavy tiles first, wave_cull.h) x parts
  const uint32_t xcd = blockIdx.x & 7u, per_xcd = (sched_positions(padded_tiles, sched) >> 3) << parts_shift;
  uint32_t* counter = tile_counters + parity * 8 + xcd;
  for (int e = threadIdx.x; e < num_items * kHbStride; e += blockDim.x) table[e] = 0;
  // A later round over a short list (the two or three keyframes that have not converged -- the steady state of a BA loop): nearly
  // every tile ends at its stored bound, and drawing batches of such tiles from the XCD's queue (a global atomic per batch, fifteen
  // wavefronts waiting for the sixteenth's fetch) took several times as long as the tiles: 43 us on an eighth of the bench cloud,
  // 70 - 125 us on all of it (round 4 traces).  Such a launch deals the positions to the wavefronts statically instead, heavy
  // tiles first and interleaved; the items are not split into parts.  (Integer sums: the same bits either way.)
  const bool short_list = !kSlice && stored_bounds && num_listed <= kPoseShortList;   // workgroup-uniform, launch-uniform
  if (threadIdx.x == 0 && !short_list) batch_state = ((unsigned long long)atomicAdd(counter, kPoseBatch) << 32) | ((unsigned long long)kPoseBatch << 24);
  __syncthreads();
  const int slot = wave_reduce28_slot(lane);
  const uint32_t table_address = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) HbFixed*)table;   // LDS byte address
  LdsSink sink{table_address + (uint32_t)(slot > 0 ? slot : 0) * (uint32_t)(kHbLimbs * sizeof(HbFixed)), invalid, slot >= 0 && slot < 27, 0.f, -1};
  // (one call site of pose_tile for both ways of handing out tiles: the body is ~2700 instructions)
  const uint32_t positions = sched_positions(padded_tiles, sched), static_stride = gridDim.x * (blockDim.x >> 6);
  uint32_t static_position = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  for (;;) {
    uint32_t tile = 0, position = 0;
    int parts = 1, part = 0;
    bool have_tile = false;
    if (short_list) {
      if (static_position >= positions) break;
      position = static_position;
      static_position += static_stride;
      have_tile = scheduled_tile(position, padded_tiles, sched, &tile);
    } else {
      unsigned long long taken = 0;
      if (lane == 0) taken = atomicAdd(&batch_state, 1ull);
      const uint32_t first = __builtin_amdgcn_readfirstlane((uint32_t)(taken >> 32));
      const uint32_t size = __builtin_amdgcn_readfirstlane((uint32_t)taken >> 24);
      const uint32_t index = __builtin_amdgcn_readfirstlane((uint32_t)taken & 0xffffffu);
      if (first >= per_xcd) break;                         // the XCD's tiles are used up
      if (index < size) {
        const uint32_t unit = first + index;
        position = (unit >> parts_shift) * 8u + xcd;
        parts = 1 << parts_shift;
        part = (int)(unit & ((1u << parts_shift) - 1u));
        have_tile = unit < per_xcd && scheduled_tile(position, padded_tiles, sched, &tile);
      } else if (index == size) {                          // this wavefront took the batch's last-plus-one: it fetches the next batch
        if (lane == 0) {
          const uint32_t left = per_xcd > first + size ? per_xcd - (first + size) : 0u;   // (as of this workgroup's last fetch)
          const uint32_t want = min(kPoseBatch, max(2u, left / (4u * (gridDim.x >> 3))));
          const uint32_t next = atomicAdd(counter, want);
          __hip_atomic_store(&batch_state, ((unsigned long long)next << 32) | ((unsigned long long)want << 24), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {                                             // the others wait for it (a few microseconds per batch)
        while ((uint32_t)(__hip_atomic_load(&batch_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> 32) == first)
          __builtin_amdgcn_s_sleep(8);
      }
    }
    if (have_tile) {
#ifdef BAHIP_TILE_TIMELINE
      const unsigned long long t0 = wall_clock64();
#endif
      pose_tile<kUseDepth, kUseDesc>(in, frames, work, num_work, s, tile_bounds, stored_bounds, num_listed, tile, parts, part, sink, tile_cost,
                                     item_begin, kSlice ? num_items : -1);
#ifdef BAHIP_TILE_TIMELINE
      if (!stored_bounds && lane == 0 && position < 65536 && tile < 65536) {
        g_pose_timeline[position][0] = t0; g_pose_timeline[position][1] = wall_clock64(); g_pose_timeline[position][2] = tile;
        g_pose_timeline[position][3] = ((unsigned long long)g_pose_tile_stats[tile][0] << 32) | g_pose_tile_stats[tile][1];
      }
#endif
    }
  }
  __syncthreads();
  const int* __restrict__ listed = reinterpret_cast<const int*>(work + num_work + kPoseTailRecords);
  for (int e = threadIdx.x; e < num_items * kHbStride; e += blockDim.x) {
    const HbFixed v = table[e];
    if (v != 0) {
      const int item = e / kHbStride;
      const int w = stored_bounds ? listed[item_begin + item] : item_begin + item;
      __hip_atomic_fetch_add(&Hb[(size_t)w * kHbStride + (e - item * kHbStride)], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- launchers of the sweep ----
static int g_forced_pose_parts = bahip_env_int("BAHIP_POSE_PARTS", 0);
void set_pose_parts(int parts) { g_forced_pose_parts = parts; }


// 0 = chosen from the sizes (default), 1 = always the one-tile-per-wavefront form with global atomics, 2 = the persistent LDS
// form whenever the table fits (tests run both: same bits)
static int g_forced_pose_form = bahip_env_int("BAHIP_POSE_FORM", 0);
void set_pose_form(int form) { g_forced_pose_form = form; }
static int g_pose_lds_items = 0;   // test hook: work items per launch of the LDS form (0: what the table holds)
void set_pose_lds_items(int items) { g_pose_lds_items = items; }
static long long g_pose_form_launches[2] = {0, 0};   // [0] one tile per wavefront + global atomics, [1] persistent + LDS
void pose_form_launches(long long out[2], bool reset) {
  out[0] = g_pose_form_launches[0]; out[1] = g_pose_form_launches[1];
  if (reset) g_pose_form_launches[0] = g_pose_form_launches[1] = 0;
}
// Kernel dispatches of the accumulate sweep since the process started (every slice of a sliced launch counts; never reset): lets a
// profile of a bench run pick the dispatches of the timed region out of rocprofv3's per-dispatch rows (scripts/summarize_profile.py).
static long long g_pose_kernel_dispatches = 0;
long long pose_kernel_dispatches() { return g_pose_kernel_dispatches; }
constexpr size_t kPoseLdsTableLimit = 128 * 1024;   // of the 160 KB of a compute unit
#ifndef BAHIP_POSE_LDS_MIN_TILES
#define BAHIP_POSE_LDS_MIN_TILES 2048
#endif
constexpr unsigned kPoseLdsMinTiles = BAHIP_POSE_LDS_MIN_TILES;   // smaller grids: one tile per wavefront, global atomics
static int g_pose_lds_parts_shift = bahip_env_int("BAHIP_POSE_LDS_PARTS_SHIFT", -1);   // -1: from the grid size
void set_pose_lds_parts_shift(int shift) { g_pose_lds_parts_shift = (shift >= 0 && shift <= 3) ? shift : -1; }

static int g_pose_lds_waves = 0;   // test hook: wavefronts per workgroup of the LDS form (0: kPoseLdsWaves)
void set_pose_lds_waves(int waves) { g_pose_lds_waves = (waves >= 1 && waves <= kPoseLdsWaves) ? waves : 0; }

// What a launch of the persistent form needs to know about the device the stream runs on, per device (a process
// that drives contexts on several devices must not reuse the first device's numbers or its opt-in).
struct PoseLdsDevice { int compute_units = 0; bool raised[8] = {}; bool failed[8] = {}; };
static PoseLdsDevice& pose_lds_device() {
  static PoseLdsDevice devices[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  PoseLdsDevice& d = devices[dev];
  if (d.compute_units == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    d.compute_units = cus;
  }
  return d;
}

// false: the launch could not be made (the opt-in for more than 64 KB of dynamic LDS was refused): the caller falls back to the
// one-tile-per-wavefront form.
template <bool kUseDepth, bool kUseDesc, bool kSlice>
static bool launch_pose_lds(hipStream_t stream, const Intrinsics& in, const KfEntry* frames, const PoseWork* pw, int num_work,
                            const SurfelsView& s, HbFixed* Hb, WaveBounds* tb, int sb, int num_listed, unsigned tiles, size_t table_bytes,
                            uint32_t* tile_counters, int parity, int slice_begin, int slice_count, uint32_t* tile_cost, const uint32_t* sched,
                            const int* listed_count, uint32_t parts_shift, const int* stop) {
  int* invalid = pose_invalid_word(Hb);
  PoseLdsDevice& device = pose_lds_device();
  constexpr int variant = (kUseDepth ? 1 : 0) + (kUseDesc ? 2 : 0) + (kSlice ? 4 : 0);
  if (!device.raised[variant] && !device.failed[variant]) {   // dynamic LDS beyond 64 KB needs the opt-in
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&pose_accumulate_lds_kernel<kUseDepth, kUseDesc, kSlice>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPoseLdsTableLimit) == hipSuccess) device.raised[variant] = true;
    else { device.failed[variant] = true; (void)hipGetLastError(); }
  }
  if (device.failed[variant] && table_bytes + sizeof(HbFixed) > 64 * 1024) return false;
  const int waves = g_pose_lds_waves > 0 ? g_pose_lds_waves : kPoseLdsWaves;
  // one workgroup per compute unit, fewer when there is less to do than that (at least one per XCD queue)
  const unsigned units = sched_positions(tiles, sched) << parts_shift;
  const unsigned grid = std::max(8u, std::min((unsigned)device.compute_units, ((units + waves - 1) / waves + 7u) & ~7u));
  ++g_pose_kernel_dispatches;
  hipLaunchKernelGGL((pose_accumulate_lds_kernel<kUseDepth, kUseDesc, kSlice>), dim3(grid), dim3(64 * waves), table_bytes + sizeof(HbFixed) /* the batch word */, stream, in, frames,
                     pw, num_work, s, Hb, tb, sb, num_listed, invalid, tiles, tile_counters, parity, slice_begin, slice_count, tile_cost, sched, listed_count, parts_shift, stop);
  return true;
}
template <bool kSlice>
static bool launch_pose_lds_any(bool use_depth, bool use_desc, hipStream_t stream, const Intrinsics& in, const KfEntry* frames, const PoseWork* pw,
                                int num_work, const SurfelsView& s, HbFixed* Hb, WaveBounds* tb, int sb, int num_listed, unsigned tiles,
                                size_t table_bytes, uint32_t* tile_counters, int parity, int slice_begin, int slice_count, uint32_t* tile_cost,
                                const uint32_t* sched, const int* listed_count, uint32_t parts_shift, const int* stop) {
  if (use_depth && use_desc) return launch_pose_lds<true, true, kSlice>(stream, in, frames, pw, num_work, s, Hb, tb, sb, num_listed, tiles, table_bytes, tile_counters, parity, slice_begin, slice_count, tile_cost, sched, listed_count, parts_shift, stop);
  if (use_depth) return launch_pose_lds<true, false, kSlice>(stream, in, frames, pw, num_work, s, Hb, tb, sb, num_listed, tiles, table_bytes, tile_counters, parity, slice_begin, slice_count, tile_cost, sched, listed_count, parts_shift, stop);
  return launch_pose_lds<false, true, kSlice>(stream, in, frames, pw, num_work, s, Hb, tb, sb, num_listed, tiles, table_bytes, tile_counters, parity, slice_begin, slice_count, tile_cost, sched, listed_count, parts_shift, stop);
}

// Can a later round over (at most) `num_items` work items be queued before the host knows how many are left?  Yes unless the
// launch would have to be cut into slices by the host (more items than the LDS table holds).
bool pose_round_can_be_queued_ahead(uint32_t /*surfels*/, int num_items, bool /*have_tile_counters*/) {
  // (decided from the item count alone, whatever form this rank's launch takes: the ranks of a sharded run hold different
  // numbers of surfels and must all make the same decision, or they disagree on the number of exchanges per host wait)
  const size_t item_bytes = sizeof(HbFixed) * kHbStride;
  const int per_launch = g_pose_lds_items > 0 ? std::min(g_pose_lds_items, (int)(kPoseLdsTableLimit / item_bytes)) : (int)(kPoseLdsTableLimit / item_bytes);
  return num_items <= per_launch;
}

void launch_pose_accumulate(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* frames,
                            const void* work, int num_work, const SurfelsView& s, HbFixed* Hb, void* tile_bounds, bool stored_bounds,
                            int num_listed, uint32_t* tile_counters, int* parity_inout, uint32_t* tile_cost, const uint32_t* sched,
                            const int* listed_count, const int* stop) {
  if (s.size == 0 || num_work == 0) return;
  // Small surfel sets (a shard of a multi-GPU run) leave the chip under-filled and the launch then lasts as long as the
  // wavefront with the most candidate keyframes: split every wavefront's candidates over several wavefronts
  // (the sums are merged by integer adds, so who visits a keyframe does not matter: same bits for every split).
  const unsigned tiles = xcd_padded_tiles((s.size + kPoseBlock - 1) / kPoseBlock);   // whole XCD runs
  const int forced = g_forced_pose_parts;
  const unsigned parts = (forced == 1 || forced == 2 || forced == 4 || forced == 8) ? (unsigned)forced
                         : tiles >= 32768 ? 1 : tiles >= 8192 ? 2 : 4;
  // measured (r2, pose stage per iteration, ms): 46.9 k tiles 1.03 / 1.08 / 1.30 with 1 / 2 / 4 parts; 23.4 k tiles 0.578 / 0.557 / 0.618
  // with 1 / 2 / 4; 11.7 k tiles 0.29 / 0.30 with 2 / 4; 5.9 k tiles 0.155 / 0.172 with 4 / 8
  const PoseWork* pw = static_cast<const PoseWork*>(work);
  WaveBounds* tb = static_cast<WaveBounds*>(tile_bounds);
  const int sb = stored_bounds ? 1 : 0;
  // The persistent LDS form when the table of the work items in this launch fits.  (round 3, with the schedule: 23.4 k tiles --
  // half of the bench scene -- 0.43 ms in the LDS form against 0.77-0.89 ms with global atomics, 11.7 k tiles 0.32 against
  // 0.39-0.45; 5.9 k tiles 0.30 against 0.25 while every wavefront of the LDS form took whole tiles: round 4 lets the
  // wavefronts of the LDS form share a tile's work items as well -- g_pose_lds_parts_shift below -- and uses it from
  // kPoseLdsMinTiles tiles on.)
  const int num_items = stored_bounds ? num_listed : num_work;
  const size_t item_bytes = sizeof(HbFixed) * kHbStride;
  const bool lds_form = tile_counters != nullptr && (g_forced_pose_form == 2 || (g_forced_pose_form == 0 && tiles >= kPoseLdsMinTiles && forced == 0));
  if (lds_form) {
    // units per wavefront slot of the chip (256 x 16): below ~4 a launch lasts as long as its longest unit
    const uint32_t parts_shift = g_pose_lds_parts_shift >= 0 ? (uint32_t)g_pose_lds_parts_shift : tiles >= 16384 ? 0u : tiles >= 8192 ? 1u : 2u;
    const int per_launch = g_pose_lds_items > 0 ? std::min(g_pose_lds_items, (int)(kPoseLdsTableLimit / item_bytes))
                                                : (int)(kPoseLdsTableLimit / item_bytes);   // 292 work items
    bool launched = true;
    if (num_items <= per_launch) {
      const int parity = *parity_inout;
      launched = launch_pose_lds_any<false>(use_depth, use_desc, stream, in, frames, pw, num_work, s, Hb, tb, sb, num_listed, tiles, item_bytes * (size_t)num_items, tile_counters, parity, 0, 0, tile_cost, sched, listed_count, parts_shift, stop);
      if (launched) *parity_inout = parity ^ 1;
    } else {
      // more work items than the table holds: slices of equal size, one launch each (every launch sweeps all tiles and culls
      // against its slice; the first round's launches all store the same tile bounds)
      const int slices = (num_items + per_launch - 1) / per_launch, per_slice = (num_items + slices - 1) / slices;
      for (int begin = 0; begin < num_items && launched; begin += per_slice) {
        const int count = std::min(per_slice, num_items - begin);
        const int parity = *parity_inout;
        launched = launch_pose_lds_any<true>(use_depth, use_desc, stream, in, frames, pw, num_work, s, Hb, tb, sb, num_listed, tiles, item_bytes * (size_t)count, tile_counters, parity, begin, count, tile_cost, sched, nullptr, parts_shift, stop);
        if (launched) *parity_inout = parity ^ 1;
      }
    }
    if (launched) { ++g_pose_form_launches[1]; return; }
    // (only a refused LDS opt-in gets here, before the first slice: nothing has been added yet)
  }
  ++g_pose_form_launches[0];
  ++g_pose_kernel_dispatches;
  const dim3 grid(sched_positions(tiles, sched), parts), block(kPoseBlock);
  int* invalid = pose_invalid_word(Hb);
  if (use_depth && use_desc) hipLaunchKernelGGL((pose_accumulate_kernel<true, true>), grid, block, 0, stream, in, frames, pw, num_work, s, Hb, tb, sb, num_listed, invalid, tile_cost, sched, listed_count, stop);
  else if (use_depth) hipLaunchKernelGGL((pose_accumulate_kernel<true, false>), grid, block, 0, stream, in, frames, pw, num_work, s, Hb, tb, sb, num_listed, invalid, tile_cost, sched, listed_count, stop);
  else hipLaunchKernelGGL((pose_accumulate_kernel<false, true>), grid, block, 0, stream, in, frames, pw, num_work, s, Hb, tb, sb, num_listed, invalid, tile_cost, sched, listed_count, stop);
}

BAHIP_FLAVOURED_END

// ---- what exists once (the exact unit): the Gauss-Newton solve, the loop control, the tile schedule, test hooks, dispatchers -----
#ifndef BAHIP_FAST_MATH
namespace bahip {
size_t pose_tile_bounds_bytes(uint32_t surfels) {
  const size_t tiles = xcd_padded_tiles((surfels + kPoseBlock - 1) / kPoseBlock);
  return tiles * sizeof(WaveBounds);
}

// B/convergence_analysis.h:43-51
__device__ __forceinline__ bool is_scale1_pose_converged(const float* x) {
  float sq = 0.f;
  for (int c = 0; c < 3; ++c) sq += x[c] * x[c];
  for (int c = 3; c < 6; ++c) { const float v = x[c] * 10.f; sq += v * v; }
  return sq < 1e-06f;
}

// LDLT with symmetric diagonal pivoting + pseudo-inverse rule on D, binary64 (what Eigen's
// H.cast<double>().selfadjointView<Upper>().ldlt().solve(b) does, B/direct_ba_alternating.cc:206).
// Every index below is a compile-time constant once the loops are unrolled -- the pivot search and the row / column / perm
// exchanges are selects over the candidates, not dynamically indexed accesses -- so the 36 + 6 + 6 binary64 values live in
// registers: with dynamic indices they lived in scratch memory and one step of 200 keyframes took 24 us, most of it the
// latency of dependent scratch accesses (round 4 trace, profiles/r4_emu8_timeline.txt).  Same operations on the same values in
// the same order as the loop form (which the oracle restates, oracle_pose.c), only where the values are kept differs.
template <int N>
__device__ __forceinline__ void ldlt_solve(double (&A)[N * N], const double (&b)[N], double (&x)[N]) {
  constexpr double kTiny = 2.2250738585072014e-308;
  int perm[N];
#pragma unroll
  for (int c = 0; c < N; ++c) perm[c] = c;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    int piv = k;
    double best = fabs(A[k * N + k]);
#pragma unroll
    for (int c = k + 1; c < N; ++c) {
      const double v = fabs(A[c * N + c]);
      const bool better = v > best;
      best = better ? v : best;
      piv = better ? c : piv;
    }
#pragma unroll
    for (int c = k + 1; c < N; ++c) {
      const bool exchange = piv == c;   // true for at most one c: rows, then columns, then perm, as the loop form
#pragma unroll
      for (int j = 0; j < N; ++j) { const double t = A[k * N + j], u = A[c * N + j]; A[k * N + j] = exchange ? u : t; A[c * N + j] = exchange ? t : u; }
#pragma unroll
      for (int j = 0; j < N; ++j) { const double t = A[j * N + k], u = A[j * N + c]; A[j * N + k] = exchange ? u : t; A[j * N + c] = exchange ? t : u; }
      const int tp = perm[k], up = perm[c];
      perm[k] = exchange ? up : tp;
      perm[c] = exchange ? tp : up;
    }
    const double d = A[k * N + k];
    if (fabs(d) > kTiny) {
#pragma unroll
      for (int c = k + 1; c < N; ++c) A[c * N + k] /= d;
#pragma unroll
      for (int c = k + 1; c < N; ++c)
#pragma unroll
        for (int j = k + 1; j <= c; ++j) {
          A[c * N + j] -= A[c * N + k] * d * A[j * N + k];
          A[j * N + c] = A[c * N + j];
        }
    } else {
#pragma unroll
      for (int c = k + 1; c < N; ++c) A[c * N + k] = 0;
    }
  }
  double y[N];
#pragma unroll
  for (int c = 0; c < N; ++c) {
    y[c] = b[0];
#pragma unroll
    for (int j = 1; j < N; ++j) y[c] = perm[c] == j ? b[j] : y[c];
  }
#pragma unroll
  for (int c = 0; c < N; ++c)
#pragma unroll
    for (int j = 0; j < c; ++j) y[c] -= A[c * N + j] * y[j];
#pragma unroll
  for (int c = 0; c < N; ++c) { const double d = A[c * N + c]; y[c] = (fabs(d) > kTiny) ? y[c] / d : 0.0; }
#pragma unroll
  for (int c = N - 1; c >= 0; --c)
#pragma unroll
    for (int j = c + 1; j < N; ++j) y[c] -= A[j * N + c] * y[j];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    x[j] = 0.0;
#pragma unroll
    for (int c = 0; c < N; ++c) x[j] = perm[c] == j ? y[c] : x[j];
  }
}

// One Gauss-Newton update of a pose: B/direct_ba_alternating.cc:173-244.  hb = H (21, row-major upper triangle) and b (6) in
// binary32 (what the reference hands to Eigen: H.cast<double>()...ldlt().solve(b.cast<double>())); x = the binary32 step,
// T_next = T * exp(-x).
__device__ void pose_gn_step(const float* hb, const float* T, float* xf, float* T_next) {
  double A[36], b[6], x[6];
  int q = 0;
#pragma unroll
  for (int row = 0; row < 6; ++row)
#pragma unroll
    for (int col = row; col < 6; ++col) { const double v = (double)hb[q]; A[row * 6 + col] = v; A[col * 6 + row] = v; ++q; }
#pragma unroll
  for (int c = 0; c < 6; ++c) b[c] = (double)hb[21 + c];
  ldlt_solve<6>(A, b, x);
  float mx[6];
  for (int c = 0; c < 6; ++c) { xf[c] = (float)x[c]; mx[c] = -1.f * xf[c]; }
  float upd[7];
  se3_exp(mx, upd);
  se3_mul(T, upd, T_next);
}

// The top of an iteration of the device-driven loop in ONE launch (num_kfs <= 1024): the activation window with its co-visible
// propagation (mode 1), or the propagation that closes the previous iteration (mode 2), or neither (mode 0) -- then the work
// items of the pose phase (pose_init_from_keyframes_kernel).  Geometry, which runs between this and the pose rounds, touches
// neither the work items nor the activations, so the phase's items can be set up before it: one launch and one launch gap less
// per iteration than window kernel + init kernel.
// (A workgroup of 1024 threads runs the body: the kernel below, or the last solve launch of the previous iteration's pose phase,
// pose_solve_begin_kernel.)
__device__ __forceinline__ void iteration_begin_body(KfEntry* __restrict__ frames, int num_kfs, int mode, const uint8_t* __restrict__ in_window,
                                                     const int* __restrict__ offsets, const int* __restrict__ indices, PoseWork* __restrict__ work,
                                                     HbFixed* __restrict__ Hb, PoseWork* __restrict__ host_out) {
  const int k = threadIdx.x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (mode == 1) {
    const bool inside = k < num_kfs && in_window[k];
    if (k < num_kfs) frames[k].activation = inside ? BAHIP_KF_ACTIVE : BAHIP_KF_INACTIVE;
    if (__syncthreads_count(inside ? 1 : 0) != num_kfs) {        // (workgroup-uniform) a window that holds every keyframe leaves nothing to wake up
      for (int row = wave; row < num_kfs; row += 16) {
        if (!in_window[row]) continue;
        for (int j = offsets[row] + lane; j < offsets[row + 1]; j += 64) {
          const int other = indices[j];
          if (!in_window[other]) frames[other].activation = BAHIP_KF_COVISIBLE_ACTIVE;
        }
      }
    }
  } else if (mode == 2) {
    // sources are the keyframes that are kActive NOW; a write only turns kInactive into kCovisibleActive (never a source)
    const bool source = k < num_kfs && frames[k].activation == BAHIP_KF_ACTIVE;
    __shared__ uint8_t is_source[1024];
    is_source[k] = source ? 1 : 0;
    __syncthreads();
    for (int row = wave; row < num_kfs; row += 16) {
      if (!is_source[row]) continue;
      for (int j = offsets[row] + lane; j < offsets[row + 1]; j += 64) {
        const int other = indices[j];
        if (!is_source[other] && frames[other].activation == BAHIP_KF_INACTIVE) frames[other].activation = BAHIP_KF_COVISIBLE_ACTIVE;
      }
    }
  }
  __threadfence();
  __syncthreads();
  const bool in_range = k < num_kfs;
  const bool inactive = in_range && __hip_atomic_load(&frames[k].activation, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == BAHIP_KF_INACTIVE;
  if (in_range) {
    PoseWork& pw = work[k];
    pw.kf_index = k;
    pw.iterations = 0;
    pw.converged = 0;
    pw.done = inactive ? 1 : 0;
    pw.skip = inactive ? 1 : 0;
    pw.moved = 0;
    for (int c = 0; c < 7; ++c) { pw.T[c] = frames[k].global_T_frame[c]; pw.T0[c] = frames[k].global_T_frame[c]; }
    for (int c = 0; c < 12; ++c) pw.F[c] = frames[k].pose.F[c];
    if (pw.done) host_out[k] = pw;
  }
  // the normal equations of the phase start at zero: consecutive threads clear consecutive words (a thread per work item clearing its own
  // 448 bytes issued 56 stores of 64 lines each)
  for (int e = threadIdx.x; e < num_kfs * kHbStride; e += blockDim.x) Hb[e] = 0;
  int* counters = reinterpret_cast<int*>(work + num_kfs);
  const int num_inactive = __syncthreads_count(inactive ? 1 : 0);
  if (threadIdx.x < kPoseTailRecords * 32) counters[threadIdx.x] = (threadIdx.x == kPoseCounterConverged) ? num_inactive : 0;
}
__global__ void __launch_bounds__(1024)
iteration_begin_kernel(KfEntry* __restrict__ frames, int num_kfs, int mode, const uint8_t* __restrict__ in_window, const int* __restrict__ offsets,
                       const int* __restrict__ indices, PoseWork* __restrict__ work, HbFixed* __restrict__ Hb, PoseWork* __restrict__ host_out,
                       const int* __restrict__ stop) {
  if (stop && load_global(stop) != 0) return;
  iteration_begin_body(frames, num_kfs, mode, in_window, offsets, indices, work, Hb, host_out);
}

// `round`: the Gauss-Newton round this launch closes (its not-done count goes to counter [round]).  update_activation: when
// a work item is done, the activation update of B/direct_ba_alternating.cc:556-577 is applied to its keyframe right here:
// moved (the logarithm of old^-1 * new fails the convergence test) -> kActive, else kInactive and one more converged keyframe.
// The reference does that on the host between the iterations; with the device table authoritative during the BA loop the
// next iteration's sweeps can be queued while the host is still copying the results into its Keyframe objects.
// host_out: page-locked host memory mapped into the device (zero-copy): a work item that is done writes its final record there,
// so the host needs no copy of the work array, only the 256 bytes of counters per round.
//
// kBeginsNext (one workgroup of 1024 threads, num_work <= 1024): the launch that ends a pose phase of the device-driven loop also
// does what the top of the next iteration would do in a launch of its own (iteration_begin_body) -- when the phase is complete and
// the loop goes on.  One launch and one launch dependency less per BA iteration (~15 us; a tenth of an iteration at an eighth of
// the cloud).
template <bool kBeginsNext>
__device__ __forceinline__ void pose_solve_body(PoseWork* __restrict__ work, int num_work, HbFixed* __restrict__ Hb,
                                                KfEntry* __restrict__ frames, int write_back, int update_activation, int round,
                                                PoseWork* __restrict__ host_out, int sequence, const PoseLoopControl& loop) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  int* counters = reinterpret_cast<int*>(work + num_work);
  // device-driven BA loop: once the loop has stopped, the launches queued behind do no work -- but still publish their
  // sequence number, the host may be waiting for exactly this launch
  const bool stopped = loop.ctl != nullptr && __hip_atomic_load(&loop.ctl[kLoopStop], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  bool stepped = false;
  if (!stopped && w == 0 && *pose_invalid_word(Hb) != 0) {   // raised by a sweep of this round, on this rank or another (see pose_invalid_word)
    atomicOr(&counters[kPoseCounterInvalid], 1);
    Hb[27 * kHbLimbs] = 0;
  }
  if (!stopped && w < num_work && !work[w].done) {
    stepped = true;
    PoseWork& pw = work[w];
    HbFixed* fixed = Hb + (size_t)w * kHbStride;
    float hb[27];
    // the fixed-point totals are rounded to binary32 first, so H and b are what bahip_accumulate_pose_estimation_coeffs returns
    bool out_of_range = false;
#pragma unroll
    for (int c = 0; c < 27; ++c) {
      const HbFixed hi = fixed[c * kHbLimbs + 1];
      out_of_range = out_of_range || hi >= kHbSumLimit || hi <= -kHbSumLimit;
      hb[c] = (float)hb_value(fixed[c * kHbLimbs], hi);
    }
    if (out_of_range) atomicOr(&counters[kPoseCounterInvalid], 1);
    for (int c = 0; c < kHbStride; ++c) fixed[c] = 0;
    float xf[6], next[7];
    pose_gn_step(hb, pw.T, xf, next);
    for (int c = 0; c < 7; ++c) pw.T[c] = next[c];
    float inv[7];
    se3_inverse(next, inv);
    se3_matrix3x4(inv, pw.F);
    pw.iterations += 1;
    const bool conv = is_scale1_pose_converged(xf);
    if (conv) pw.converged = 1;
    if (conv || pw.iterations >= BAHIP_MAX_POSE_ITERATIONS) {
      pw.done = 1;
      pw.skip = 1;
      if (!conv && loop.ctl) atomicAdd(&loop.ctl[kLoopNotConverged], 1);
      if (write_back) {
        KfEntry& kf = frames[pw.kf_index];
        for (int c = 0; c < 7; ++c) kf.global_T_frame[c] = next[c];
        for (int c = 0; c < 12; ++c) kf.pose.F[c] = pw.F[c];
        se3_rotation(next, kf.pose.GR);
        if (update_activation) {
          float inv0[7], diff[7], lg[6];
          se3_inverse(pw.T0, inv0);        // Keyframe::frame_T_global() of the old pose
          se3_mul(inv0, next, diff);
          se3_log(diff, lg);
          const bool moved = !is_scale1_pose_converged(lg);
          kf.activation = moved ? BAHIP_KF_ACTIVE : BAHIP_KF_INACTIVE;
          pw.moved = moved ? 1 : 0;
          if (!moved) atomicAdd(&counters[kPoseCounterConverged], 1);
        }
      }
      host_out[w] = pw;
    } else {
      const int slot = atomicAdd(&counters[round], 1);
      counters[kPoseTailRecords * 32 + slot] = w;   // the list of work items still iterating (ba_device.h: pose_work_records)
    }
  }
  // The last workgroup to finish publishes the counters to the host copy and then the launch's sequence number, which the
  // host polls: the records written to host_out above (mapped, coherent host memory) are fenced before it.
  const int steps_here = __syncthreads_count(stepped ? 1 : 0);
  if (loop.ctl && threadIdx.x == 0 && steps_here) {
    atomicAdd(&loop.ctl[kLoopSteps], steps_here);
    atomicAdd(&counters[kPoseCounterWorked], steps_here);
  }
  const bool publish = loop.publish != 0;
  if (publish) __threadfence_system(); else __threadfence();
  __syncthreads();
  __shared__ int is_last, begins_next;
  if (threadIdx.x == 0) { is_last = atomicAdd(&counters[kPoseCounterTicket], 1) == (int)gridDim.x - 1; begins_next = 0; }
  __syncthreads();
  if (is_last) {   // workgroup-uniform; 64 threads = the 64 counter words
    if (loop.ctl && threadIdx.x == 0) {
      // (sticky over the loop: the next phase's work-item set-up clears the counter words)
      if (__hip_atomic_load(&counters[kPoseCounterInvalid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&loop.ctl[kLoopInvalid], 1);
      // what this round did, and -- at the end of a phase -- whether the loop goes on (B/direct_ba_alternating.cc:693-701)
      const int worked = __hip_atomic_load(&counters[kPoseCounterWorked], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      counters[kPoseCounterWorked] = 0;
      if (loop.round_log) loop.round_log[loop.log_slot] = worked;
      if (worked) atomicAdd(&loop.ctl[kLoopRounds], 1);
      if (!stopped && loop.phase_end) {
        const int iterating = __hip_atomic_load(&counters[round], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (iterating > 0) {
          __hip_atomic_store(&loop.ctl[kLoopStop], 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // the host continues this phase
        } else {
          atomicAdd(&loop.ctl[kLoopIterationsDone], 1);
          const int converged = __hip_atomic_load(&counters[kPoseCounterConverged], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (loop.iteration >= loop.min_iterations - 1 && converged == num_work)
            __hip_atomic_store(&loop.ctl[kLoopStop], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else if (kBeginsNext && loop.next_mode >= 0)
            begins_next = 1;
        }
      }
      if (publish) for (int c = 0; c < kLoopWords; ++c) loop.host_ctl[c] = __hip_atomic_load(&loop.ctl[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    int* host_counters = reinterpret_cast<int*>(host_out + num_work);
    const int c = threadIdx.x;
    if (publish && c < kPoseTailRecords * 32 && c != kPoseCounterTicket && c != kPoseCounterSequence)
      host_counters[c] = __hip_atomic_load(&counters[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (publish) __threadfence_system(); else __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) counters[kPoseCounterTicket] = 0;
    if (kBeginsNext) {
      // the host copy of this phase's counters is out; the work items of the next phase (and its counter words) are set up
      // before the sequence number says that this launch is complete
      if (begins_next) {   // workgroup-uniform (read after the barrier above)
        iteration_begin_body(frames, num_work, loop.next_mode, loop.in_window, loop.covis_offsets, loop.covis_indices, work, Hb, host_out);
        if (publish) __threadfence_system(); else __threadfence();
        __syncthreads();
      }
    }
    if (publish && threadIdx.x == 0) __hip_atomic_store(&host_counters[kPoseCounterSequence], sequence, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ void pose_solve_kernel(PoseWork* __restrict__ work, int num_work, HbFixed* __restrict__ Hb,
                                  KfEntry* __restrict__ frames, int write_back, int update_activation, int round,
                                  PoseWork* __restrict__ host_out, int sequence, PoseLoopControl loop) {
  pose_solve_body<false>(work, num_work, Hb, frames, write_back, update_activation, round, host_out, sequence, loop);
}
__global__ void __launch_bounds__(1024)
pose_solve_begin_kernel(PoseWork* __restrict__ work, int num_work, HbFixed* __restrict__ Hb,
                        KfEntry* __restrict__ frames, int write_back, int update_activation, int round,
                        PoseWork* __restrict__ host_out, int sequence, PoseLoopControl loop) {
  pose_solve_body<true>(work, num_work, Hb, frames, write_back, update_activation, round, host_out, sequence, loop);
}

// Test hook: pose_gn_step on explicit inputs.  in = hb[27] | T[7]; out = x[6] | T_next[7] | frame_T_global of T_next [12].
__global__ void pose_step_debug_kernel(const float* __restrict__ in, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  float hb[27], T[7], xf[6], next[7], inv[7], F[12];
  for (int c = 0; c < 27; ++c) hb[c] = in[c];
  for (int c = 0; c < 7; ++c) T[c] = in[27 + c];
  pose_gn_step(hb, T, xf, next);
  se3_inverse(next, inv);
  se3_matrix3x4(inv, F);
  for (int c = 0; c < 6; ++c) out[c] = xf[c];
  for (int c = 0; c < 7; ++c) out[6 + c] = next[c];
  for (int c = 0; c < 12; ++c) out[13 + c] = F[c];
}
void launch_pose_step_debug(hipStream_t stream, const float* in, float* out) {
  hipLaunchKernelGGL(pose_step_debug_kernel, dim3(1), dim3(64), 0, stream, in, out);
}

// Builds one work item per bound keyframe (skipping kInactive ones), B/direct_ba_alternating.cc:547-553.  One workgroup
// (kSingleBlock): the count of inactive keyframes is a __syncthreads_count instead of one thread walking the table (200
// dependent loads took 10 us); larger tables take the multi-block form, whose thread 0 counts.
template <bool kSingleBlock>
__global__ void __launch_bounds__(kSingleBlock ? 1024 : 64)
pose_init_from_keyframes_kernel(const KfEntry* __restrict__ frames, int num_kfs, PoseWork* __restrict__ work,
                                HbFixed* __restrict__ Hb, PoseWork* __restrict__ host_out, uint32_t owner_mask, uint32_t owner_rank,
                                const int* __restrict__ stop) {
  if (stop && load_global(stop) != 0) return;   // device-driven BA loop: the loop has ended, the work items stay as they are
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = k < num_kfs;
  const bool inactive = in_range && frames[k].activation == BAHIP_KF_INACTIVE;
  if (in_range) {
    PoseWork& pw = work[k];
    pw.kf_index = k;
    pw.iterations = 0;
    pw.converged = 0;
    pw.done = inactive ? 1 : 0;
    // keyframe sharding: every rank solves every work item from the summed normal equations, but sweeps only the keyframes it
    // holds (keyframe k lives on rank k % world, world a power of two; owner_mask = world - 1, 0 without sharding)
    pw.skip = (inactive || ((uint32_t)k & owner_mask) != owner_rank) ? 1 : 0;
    pw.moved = 0;
    for (int c = 0; c < 7; ++c) { pw.T[c] = frames[k].global_T_frame[c]; pw.T0[c] = frames[k].global_T_frame[c]; }
    for (int c = 0; c < 12; ++c) pw.F[c] = frames[k].pose.F[c];
    for (int c = 0; c < kHbStride; ++c) Hb[(size_t)k * kHbStride + c] = 0;
    if (pw.done) host_out[k] = pw;   // skipped keyframes: their (unchanged) record
  }
  // the counters behind the work items: per-round not-done counts = 0, converged = the inactive keyframes
  int* counters = reinterpret_cast<int*>(work + num_kfs);
  if (kSingleBlock) {
    const int num_inactive = __syncthreads_count(inactive ? 1 : 0);
    if (threadIdx.x < kPoseTailRecords * 32) counters[threadIdx.x] = (threadIdx.x == kPoseCounterConverged) ? num_inactive : 0;
  } else if (k == 0) {
    int num_inactive = 0;
    for (int j = 0; j < num_kfs; ++j) num_inactive += (frames[j].activation == BAHIP_KF_INACTIVE) ? 1 : 0;
    for (int c = 0; c < kPoseTailRecords * 32; ++c) counters[c] = (c == kPoseCounterConverged) ? num_inactive : 0;
  }
}

// DirectBA::DetermineCovisibleActiveKeyframes (B/direct_ba.cc:549-564) on the device table: every kInactive keyframe that is
// co-visible with a kActive one becomes kCovisibleActive.  Lists in CSR form over bound keyframe indices.  Order-free: a
// write only turns kInactive into kCovisibleActive and only kActive entries are sources.
__global__ void propagate_covisible_kernel(KfEntry* __restrict__ frames, int num_kfs, const int* __restrict__ offsets,
                                           const int* __restrict__ indices, const int* __restrict__ stop) {
  if (stop && load_global(stop) != 0) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= num_kfs || frames[k].activation != BAHIP_KF_ACTIVE) return;
  for (int j = offsets[k]; j < offsets[k + 1]; ++j) {
    const int other = indices[j];
    if (frames[other].activation == BAHIP_KF_INACTIVE) frames[other].activation = BAHIP_KF_COVISIBLE_ACTIVE;
  }
}

// Top of an alternating iteration with a fixed active window (B/direct_ba_alternating.cc:353-371): keyframes inside the window
// become kActive, all others kInactive (the co-visible ones are then raised by propagate_covisible_kernel).
__global__ void window_activation_kernel(KfEntry* __restrict__ frames, int num_kfs, const uint8_t* __restrict__ in_window,
                                         const int* __restrict__ stop) {
  if (stop && load_global(stop) != 0) return;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < num_kfs) frames[k].activation = in_window[k] ? BAHIP_KF_ACTIVE : BAHIP_KF_INACTIVE;
}

// Both steps in one workgroup (num_kfs <= 1024): one launch at the top of an iteration instead of two.  The rows of the
// co-visibility CSR are dealt to the 16 wavefronts, the entries of a row to the lanes (a thread per row walking its row
// alone took 43 us at 200 keyframes: 200 dependent loads).
__global__ void __launch_bounds__(1024) window_and_propagate_kernel(KfEntry* __restrict__ frames, int num_kfs,
                                                                    const uint8_t* __restrict__ in_window,
                                                                    const int* __restrict__ offsets, const int* __restrict__ indices,
                                                                    const int* __restrict__ stop) {
  if (stop && load_global(stop) != 0) return;
  const int k = threadIdx.x;
  const bool inside = k < num_kfs && in_window[k];
  if (k < num_kfs) frames[k].activation = inside ? BAHIP_KF_ACTIVE : BAHIP_KF_INACTIVE;
  // (also the barrier between the two steps) a window that holds every keyframe leaves nothing to wake up
  if (__syncthreads_count(inside ? 1 : 0) == num_kfs) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int row = wave; row < num_kfs; row += 16) {
    if (!in_window[row]) continue;   // wave-uniform
    for (int j = offsets[row] + lane; j < offsets[row + 1]; j += 64) {
      const int other = indices[j];
      if (!in_window[other]) frames[other].activation = BAHIP_KF_COVISIBLE_ACTIVE;
    }
  }
}

bool launch_iteration_begin(hipStream_t stream, KfEntry* frames, int num_kfs, int mode, const uint8_t* in_window, const int* offsets, const int* indices,
                            void* work, HbFixed* Hb, void* host_out, const int* stop) {
  if (num_kfs == 0 || num_kfs > 1024) return false;
  hipLaunchKernelGGL(iteration_begin_kernel, dim3(1), dim3(1024), 0, stream, frames, num_kfs, mode, in_window, offsets, indices,
                     static_cast<PoseWork*>(work), Hb, static_cast<PoseWork*>(host_out), stop);
  return true;
}

// ---- launchers -----------------------------------------------------------------------------------
void launch_window_activation(hipStream_t stream, KfEntry* frames, int num_kfs, const uint8_t* in_window, const int* offsets,
                              const int* indices, const int* stop) {
  if (num_kfs == 0) return;
  if (num_kfs <= 1024) {
    hipLaunchKernelGGL(window_and_propagate_kernel, dim3(1), dim3(1024), 0, stream, frames, num_kfs, in_window, offsets, indices, stop);
  } else {
    hipLaunchKernelGGL(window_activation_kernel, dim3((num_kfs + 63) / 64), dim3(64), 0, stream, frames, num_kfs, in_window, stop);
    hipLaunchKernelGGL(propagate_covisible_kernel, dim3((num_kfs + 63) / 64), dim3(64), 0, stream, frames, num_kfs, offsets, indices, stop);
  }
}
void launch_propagate_covisible(hipStream_t stream, KfEntry* frames, int num_kfs, const int* offsets, const int* indices, const int* stop) {
  if (num_kfs) hipLaunchKernelGGL(propagate_covisible_kernel, dim3((num_kfs + 63) / 64), dim3(64), 0, stream, frames, num_kfs, offsets, indices, stop);
}

// The schedule of the sweeps that follow (wave_cull.h: scheduled_tile) from the candidates every tile visited in the pose sweep's
// first round (tile_cost; cleared here for the next census).  One workgroup; every pass over the tiles reads them with
// consecutive threads on consecutive tiles (a thread per run walking its 128 tiles took 0.3 ms).
//   1. the tiles of at least 2.5 x the mean cost go to the heavy list (at most kHeavySlots; whatever exceeds that stays regular);
//   2. the runs are ranked by the descending cost of their most expensive quarter (a run that is light on average but has a
//      heavy stretch must not come late: the end of a launch waits for single tiles, and the pose sweep's workgroups take a
//      stretch of consecutive positions at once) -- by counting, ties by run index, so it is a permutation whatever the costs are;
//   3. position row * 8 + x runs on XCD x: within every row of eight runs the heavier run goes to the XCD that has received less
//      so far (the XCDs' queues are separate -- the hardware deals workgroups round-robin, the pose sweep keeps a counter per XCD --
//      and with the rows dealt in sorted order their totals differed by 5 %: 35 us between the first and the last queue running dry);
//   4. the tiles of the last rows (a quarter of them, at most 12 288 tiles) are placed one by one, by descending cost (a counting sort; heavy tiles, which
//      are skipped there anyway, last), so that the launch runs out on tiles of one or two candidates.
constexpr int kTileOrderMaxRuns = 2560;   // 327 680 tiles = 21 M surfels
#ifndef BAHIP_SCHED_TAIL_DIV
#define BAHIP_SCHED_TAIL_DIV 4   // measured at the bench size: 8 -> 4: both sweeps 1 % shorter; 3: no further gain
#endif
#ifndef BAHIP_SCHED_TAIL_TILES
#define BAHIP_SCHED_TAIL_TILES 12288   // three rounds of the 4096 wavefront slots: the tile-by-tile tail gives up L2 locality, no need for more
#endif
constexpr uint32_t kSchedTailTiles = BAHIP_SCHED_TAIL_TILES;
__global__ void __launch_bounds__(1024) tile_order_kernel(uint32_t* __restrict__ tile_cost, uint32_t padded_tiles, uint32_t* __restrict__ sched) {
  __shared__ uint32_t quarter_cost[4 * kTileOrderMaxRuns];   // cost of every quarter of a run, heavy tiles left out
  __shared__ uint32_t key[kTileOrderMaxRuns];
  __shared__ uint32_t sorted[kTileOrderMaxRuns];
  __shared__ uint32_t order[kTileOrderMaxRuns];              // order[row * 8 + x] = the run at that place of XCD x's queue
  __shared__ uint32_t bucket[257];
  __shared__ unsigned long long total;
  __shared__ uint32_t heavy_count;
  const uint32_t runs = xcd_run_count(padded_tiles), run_tiles = padded_tiles / runs, quarter_tiles = run_tiles / 4;
  uint32_t* perm = sched + kSchedPerm;
  uint32_t* flags = perm + padded_tiles;
  if (threadIdx.x == 0) { total = 0; heavy_count = 0; }
  for (uint32_t q = threadIdx.x; q < 4 * runs; q += blockDim.x) quarter_cost[q] = 0;
  for (uint32_t q = threadIdx.x; q < 257; q += blockDim.x) bucket[q] = 0;
  __syncthreads();
  unsigned long long mine_total = 0;
  for (uint32_t t = threadIdx.x; t < padded_tiles; t += blockDim.x) mine_total += tile_cost[t];
  atomicAdd(&total, mine_total);
  __syncthreads();
  const uint32_t threshold = (uint32_t)((total * 5ull) / (2ull * padded_tiles)) + 1u;   // 2.5 x the mean, at least 1
  for (uint32_t t = threadIdx.x; t < padded_tiles; t += blockDim.x) {
    const uint32_t c = tile_cost[t];
    uint32_t heavy = 0;
    if (c >= threshold) {
      const uint32_t slot = atomicAdd(&heavy_count, 1u);
      if (slot < kHeavySlots) { sched[8 + slot] = t; heavy = 1; }
    }
    flags[t] = heavy;
    if (!heavy && c) atomicAdd(&quarter_cost[t / quarter_tiles], c);
  }
  __syncthreads();
  if (threadIdx.x == 0) sched[0] = min(heavy_count, kHeavySlots);
  for (uint32_t r = threadIdx.x; r < runs; r += blockDim.x)
    key[r] = max(max(quarter_cost[4 * r], quarter_cost[4 * r + 1]), max(quarter_cost[4 * r + 2], quarter_cost[4 * r + 3]));
  __syncthreads();
  for (uint32_t r = threadIdx.x; r < runs; r += blockDim.x) {
    const uint32_t mine = key[r];
    uint32_t rank = 0;
    for (uint32_t o = 0; o < runs; ++o) rank += (key[o] > mine || (key[o] == mine && o < r)) ? 1u : 0u;
    sorted[rank] = r;
  }
  __syncthreads();
  for (uint32_t r = threadIdx.x; r < runs; r += blockDim.x)   // (key is free now: a run's total, for the balance below)
    key[r] = quarter_cost[4 * r] + quarter_cost[4 * r + 1] + quarter_cost[4 * r + 2] + quarter_cost[4 * r + 3];
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t row = 0; row * 8 < runs; ++row) {
      uint32_t run_of[8], cost_of[8], xcd_of[8];
#pragma unroll
      for (uint32_t c = 0; c < 8; ++c) run_of[c] = sorted[row * 8 + c];
#pragma unroll
      for (uint32_t c = 0; c < 8; ++c) { cost_of[c] = key[run_of[c]]; xcd_of[c] = c; }
      for (uint32_t a = 1; a < 8; ++a)        // runs by descending total, XCDs by ascending load (insertion sorts of eight)
        for (uint32_t b = a; b > 0 && cost_of[b] > cost_of[b - 1]; --b) {
          const uint32_t t = cost_of[b]; cost_of[b] = cost_of[b - 1]; cost_of[b - 1] = t;
          const uint32_t u = run_of[b]; run_of[b] = run_of[b - 1]; run_of[b - 1] = u;
        }
      for (uint32_t a = 1; a < 8; ++a)
        for (uint32_t b = a; b > 0 && load[xcd_of[b]] < load[xcd_of[b - 1]]; --b) { const uint32_t t = xcd_of[b]; xcd_of[b] = xcd_of[b - 1]; xcd_of[b - 1] = t; }
      for (uint32_t c = 0; c < 8; ++c) {
        order[row * 8 + xcd_of[c]] = run_of[c];
        load[xcd_of[c]] += cost_of[c];
      }
    }
  }
  __syncthreads();
  // the permutation: whole runs for the leading rows ...
  const uint32_t rows = runs / 8, tail_rows = max(1u, min(rows / BAHIP_SCHED_TAIL_DIV, kSchedTailTiles / (8u * run_tiles))), head_positions = (rows - tail_rows) * 8 * run_tiles;
  const uint32_t shift = padded_tiles >= kXcdLargeGrid ? 7u : 5u;
  for (uint32_t p = threadIdx.x; p < head_positions; p += blockDim.x) {
    const uint32_t xcd = p & 7u, j = p >> 3;
    perm[p] = (order[((j >> shift) << 3) + xcd] << shift) + (j & ((1u << shift) - 1u));
  }
  // ... and the tiles of the last rows one by one: bucket 0 = cost >= 255 ... bucket 255 = cost 0, bucket 256 = heavy (skipped)
  const uint32_t tail_tiles = padded_tiles - head_positions, first_tail_run = (rows - tail_rows) * 8;
  auto tail_tile = [&](uint32_t e) { return order[first_tail_run + e / run_tiles] * run_tiles + e % run_tiles; };
  auto bucket_of = [&](uint32_t t) { return flags[t] ? 256u : 255u - min(tile_cost[t], 255u); };
  for (uint32_t e = threadIdx.x; e < tail_tiles; e += blockDim.x) atomicAdd(&bucket[bucket_of(tail_tile(e))], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t running = 0;
    for (uint32_t q = 0; q < 257; ++q) { const uint32_t n = bucket[q]; bucket[q] = running; running += n; }
  }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < tail_tiles; e += blockDim.x) {
    const uint32_t t = tail_tile(e);
    perm[head_positions + atomicAdd(&bucket[bucket_of(t)], 1u)] = t;
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < padded_tiles; t += blockDim.x) tile_cost[t] = 0;
}
bool launch_tile_order(hipStream_t stream, uint32_t* tile_cost, uint32_t padded_tiles, uint32_t* sched) {
  if (padded_tiles == 0 || xcd_run_count(padded_tiles) > (uint32_t)kTileOrderMaxRuns) return false;
  hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, tile_cost, padded_tiles, sched);
  return true;
}
size_t tile_schedule_words(uint32_t padded_tiles) { return sched_words(padded_tiles); }
uint32_t pose_padded_tiles(uint32_t surfels) { return xcd_padded_tiles((surfels + kPoseBlock - 1) / kPoseBlock); }

void launch_pose_solve(hipStream_t stream, void* work, int num_work, HbFixed* Hb, KfEntry* frames, int write_back,
                       int update_activation, int round, void* host_out, int sequence, const PoseLoopControl* loop) {
  if (num_work == 0) return;
  if (loop && loop->next_mode >= 0 && num_work <= 1024) {
    hipLaunchKernelGGL(pose_solve_begin_kernel, dim3(1), dim3(1024), 0, stream, static_cast<PoseWork*>(work),
                       num_work, Hb, frames, write_back, update_activation, round, static_cast<PoseWork*>(host_out), sequence, *loop);
    return;
  }
  PoseLoopControl plain = loop ? *loop : PoseLoopControl{};
  plain.next_mode = -1;
  hipLaunchKernelGGL(pose_solve_kernel, dim3((num_work + 63) / 64), dim3(64), 0, stream, static_cast<PoseWork*>(work),
                     num_work, Hb, frames, write_back, update_activation, round, static_cast<PoseWork*>(host_out), sequence, plain);
}

void launch_pose_init_from_keyframes(hipStream_t stream, const KfEntry* frames, int num_kfs, void* work, HbFixed* Hb, void* host_out,
                                     int kf_rank, int kf_world, const int* stop) {
  if (num_kfs == 0) return;
  const uint32_t mask = (uint32_t)(kf_world - 1), rank = (uint32_t)kf_rank;
  if (num_kfs <= 1024)
    hipLaunchKernelGGL(pose_init_from_keyframes_kernel<true>, dim3(1), dim3(1024), 0, stream, frames, num_kfs,
                       static_cast<PoseWork*>(work), Hb, static_cast<PoseWork*>(host_out), mask, rank, stop);
  else
    hipLaunchKernelGGL(pose_init_from_keyframes_kernel<false>, dim3((num_kfs + 63) / 64), dim3(64), 0, stream, frames, num_kfs,
                       static_cast<PoseWork*>(work), Hb, static_cast<PoseWork*>(host_out), mask, rank, stop);
}

// dispatchers (ba_launch.h: "Two arithmetic flavours"); the hooks reach both flavours, the counters add up
void launch_pose_accumulate(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* frames,
                            const void* work, int num_work, const SurfelsView& s, HbFixed* Hb, void* tile_bounds, bool stored_bounds,
                            int num_listed, uint32_t* tile_counters, int* parity_inout, uint32_t* tile_cost, const uint32_t* sched,
                            const int* listed_count, const int* stop) {
  BAHIP_PICK(in, launch_pose_accumulate(stream, use_depth, use_desc, in, frames, work, num_work, s, Hb, tile_bounds, stored_bounds, num_listed,
                                        tile_counters, parity_inout, tile_cost, sched, listed_count, stop));
}
bool pose_round_can_be_queued_ahead(uint32_t surfels, int num_items, bool have_tile_counters) {
  return exact::pose_round_can_be_queued_ahead(surfels, num_items, have_tile_counters);   // (the hooks it looks at are set alike in both)
}
void set_pose_parts(int parts) { exact::set_pose_parts(parts); fast::set_pose_parts(parts); }
void set_pose_form(int form) { exact::set_pose_form(form); fast::set_pose_form(form); }
void set_pose_lds_items(int items) { exact::set_pose_lds_items(items); fast::set_pose_lds_items(items); }
void set_pose_lds_parts_shift(int shift) { exact::set_pose_lds_parts_shift(shift); fast::set_pose_lds_parts_shift(shift); }
void set_pose_lds_waves(int waves) { exact::set_pose_lds_waves(waves); fast::set_pose_lds_waves(waves); }
long long pose_kernel_dispatches() { return exact::pose_kernel_dispatches() + fast::pose_kernel_dispatches(); }
void pose_form_launches(long long out[2], bool reset) {
  long long a[2], b[2];
  exact::pose_form_launches(a, reset); fast::pose_form_launches(b, reset);
  out[0] = a[0] + b[0]; out[1] = a[1] + b[1];
}
#ifdef BAHIP_COUNT_CANDIDATES
void pose_counters_dump() { exact::pose_counters_dump(); }
#endif
#ifdef BAHIP_TILE_TIMELINE
void pose_timeline_dump(const char* path) { exact::pose_timeline_dump(path); }
#endif

}  // namespace bahip

// ---- test hook: per-pair evaluation ------------------------------------------------------------------
// Evaluates association, raw residuals, weights and pose Jacobians of individual (surfel, frame)
// pairs with exactly the device functions the production kernels use; lets the parity tests
// compare per-pair quantities against the oracle.  out: 40 floats per surfel index:
// [0] associated, [1] px, [2] py, [3] colour-valid, [4] calibrated depth, [5] depth residual,
// [6] depth weight, [7] inv stddev, [8..13] depth J, [14..15] desc residuals, [16..17] desc weights,
// [18..23] desc J1, [24..29] desc J2, [30..33] gradients, [34..35] pxx, pxy.
#endif   // !BAHIP_FAST_MATH (the per-pair hook exists in both flavours: the flavours' association decisions are compared pair by pair)
BAHIP_FLAVOURED_BEGIN
__global__ void evaluate_pairs_kernel(Intrinsics in, KfEntry frame, SurfelsView s, const uint32_t* __restrict__ indices,
                                      int count, float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  float* o = out + 40 * (size_t)t;
  for (int c = 0; c < 40; ++c) o[c] = 0.f;
  const uint32_t i = indices[t];
  if (i >= s.size) return;
  const float* F = frame.pose.F;
  const Vec3 gp = surfel_position(s, i);
  const Vec3 gn = surfel_normal(s, i);
  Assoc r;
  if (!project_associate<false>(in, F, frame.geom, gp, gn, &r, nullptr)) return;
  o[0] = 1.f; o[1] = (float)r.px; o[2] = (float)r.py; o[4] = r.depth; o[34] = r.pxx; o[35] = r.pxy;
  const float inv_std = assoc_inv_std(in, r);
  const Vec3 u = assoc_unproject(r);
  const float raw = inv_std * dot3(r.nl, u - r.local);
  o[5] = raw; o[6] = depth_residual_weight(raw); o[7] = inv_std;
  { float Jd[6]; jac_depth_pose(r.nl, u, inv_std, Jd); for (int c = 0; c < 6; ++c) o[8 + c] = Jd[c]; }
  float cx, cy;
  if (!depth_to_color_pixel(in, r.pxx, r.pxy, &cx, &cy)) return;
  o[3] = 1.f;
  DescEval e;
  eval_descriptor<true>(in, frame.lumafp, F, gp, gn, s.row(kSurfelRadiusSquared)[i], cx, cy,
                        s.row(kSurfelDescriptor1)[i], s.row(kSurfelDescriptor2)[i], &e);
  o[14] = e.r1; o[15] = e.r2; o[16] = descriptor_residual_weight(e.r1); o[17] = descriptor_residual_weight(e.r2);
  o[30] = e.gx1; o[31] = e.gy1; o[32] = e.gx2; o[33] = e.gy2;
  for (int q = 0; q < 2; ++q) {
    const float gx = (q ? e.gx2 : e.gx1) * in.cfx, gy = (q ? e.gy2 : e.gy1) * in.cfy;
    float Jq[6];
    jac_descriptor_pose(r.local, gx, gy, Jq);
    for (int c = 0; c < 6; ++c) o[18 + 6 * q + c] = Jq[c];
  }
}
void launch_evaluate_pairs(hipStream_t stream, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s,
                           const uint32_t* indices, int count, float* out) {
  if (count) hipLaunchKernelGGL(evaluate_pairs_kernel, dim3((count + 63) / 64), dim3(64), 0, stream, in, frame, s, indices, count, out);
}
BAHIP_FLAVOURED_END
#ifndef BAHIP_FAST_MATH
namespace bahip {
void launch_evaluate_pairs(hipStream_t stream, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s,
                           const uint32_t* indices, int count, float* out) {
  BAHIP_PICK(in, launch_evaluate_pairs(stream, in, frame, s, indices, count, out));
}
// Debug: one wave64; in = 64 x 28 floats (lane-major).  out[0..27] = wave_reduce28 totals scattered by slot,
// out[28..55] = wave_sum of every column (what lane 17 receives; every lane holds the same value).
__global__ void wave_reduce_debug_kernel(const float* __restrict__ in, float* __restrict__ out) {
  const int lane = threadIdx.x;
  float acc[28];
#pragma unroll
  for (int q = 0; q < 28; ++q) acc[q] = in[lane * 28 + q];
  const float mine = wave_reduce28(acc, lane);
  const int slot = wave_reduce28_slot(lane);
  if (slot >= 0) out[slot] = mine;
#pragma unroll
  for (int q = 0; q < 28; ++q) {
    const float v = wave_sum(acc[q]);
    if (lane == 17) out[28 + q] = v;
  }
  float v8[8], v16[16];
#pragma unroll
  for (int q = 0; q < 8; ++q) v8[q] = acc[q];
#pragma unroll
  for (int q = 0; q < 16; ++q) v16[q] = acc[q];
  const float t8 = wave_reduce_small<8>(v8, lane), t16 = wave_reduce_small<16>(v16, lane);
  if ((lane & 7) == 0) out[56 + (lane >> 3)] = t8;
  if ((lane & 3) == 0) out[64 + (lane >> 2)] = t16;
}
void launch_wave_reduce_debug(hipStream_t stream, const float* in, float* out) {
  hipLaunchKernelGGL(wave_reduce_debug_kernel, dim3(1), dim3(64), 0, stream, in, out);
}
// Debug: the Jacobian functions of ba_device.h on explicit inputs (one thread).  Layout of `in` / `out` per kind:
//   0 depth/pose          in nl[3] u[3] inv_std                                        out J[6]
//   1 descriptor/pose     in ls[3] gx gy                                               out J[6]
//   2 descriptor/surfel   in rn[3] lp[3] gx gy cfx cfy                                 out J[1]
//   3 depth/intrinsics    in px py depth inv_std ndF0 ndF1 dot cfactor raw_inv exp_inv corrected   out J[6]
//   4 descriptor/colour   in gx gy nx ny                                               out J[4]
__global__ void jacobian_debug_kernel(int kind, const float* __restrict__ in, float* __restrict__ out) {
  if (threadIdx.x != 0) return;
  if (kind == 0) { float J[6]; jac_depth_pose(mk3(in[0], in[1], in[2]), mk3(in[3], in[4], in[5]), in[6], J); for (int c = 0; c < 6; ++c) out[c] = J[c]; }
  else if (kind == 1) { float J[6]; jac_descriptor_pose(mk3(in[0], in[1], in[2]), in[3], in[4], J); for (int c = 0; c < 6; ++c) out[c] = J[c]; }
  else if (kind == 2) { out[0] = jac_descriptor_surfel(mk3(in[0], in[1], in[2]), mk3(in[3], in[4], in[5]), in[6], in[7], in[8], in[9]); }
  else if (kind == 3) { float J[6]; jac_depth_intrinsics((int)in[0], (int)in[1], in[2], in[3], in[4], in[5], in[6], in[7], in[8], in[9], in[10], J); for (int c = 0; c < 6; ++c) out[c] = J[c]; }
  else if (kind == 4) { float J[4]; jac_descriptor_color_intrinsics(in[0], in[1], in[2], in[3], J); for (int c = 0; c < 4; ++c) out[c] = J[c]; }
}
// Debug: rcp_exact / sqrt_exact (ba_device.h) on n explicit inputs; kind 0 = reciprocal, 1 = square root.
__global__ void exact_math_debug_kernel(int kind, const float* __restrict__ in, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (kind == 0) out[i] = rcp_exact(in[i]);
  else if (kind == 1) out[i] = sqrt_exact(in[i]);
  else if (kind == 4) out[i] = atan_det(in[i]);
  else if (kind == 5) out[i] = exp_det(in[i]);
  else {   // 2: sin, 3: cos (se3_device.h: sincos_det)
    float sn, cs;
    sincos_det(in[i], &sn, &cs);
    out[i] = kind == 2 ? sn : cs;
  }
}
// Test hook: hb_split on explicit values.  out: [lo, hi, valid] per value.
__global__ void pose_limbs_debug_kernel(const float* __restrict__ in, long long* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // what the sinks add: the magnitudes of hb_split_magnitudes with the total's sign (the oracle's hb_split returns these limbs)
  const HbMagnitudes m = hb_split_magnitudes(in[i]);
  const long long lo = (long long)m.lo, hi = (long long)(((unsigned long long)m.hi_hi << 32) | m.hi_lo);
  out[3 * i] = m.valid ? (m.negative ? -lo : lo) : 0; out[3 * i + 1] = m.valid ? (m.negative ? -hi : hi) : 0; out[3 * i + 2] = m.valid ? 1 : 0;
}
void launch_pose_limbs_debug(hipStream_t stream, const float* in, long long* out, size_t n) {
  if (n) hipLaunchKernelGGL(pose_limbs_debug_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, out, n);
}
void launch_exact_math_debug(hipStream_t stream, int kind, const float* in, float* out, size_t n) {
  if (n) hipLaunchKernelGGL(exact_math_debug_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, kind, in, out, n);
}
void launch_jacobian_debug(hipStream_t stream, int kind, const float* in, float* out) {
  hipLaunchKernelGGL(jacobian_debug_kernel, dim3(1), dim3(64), 0, stream, kind, in, out);
}
}  // namespace bahip
#endif   // !BAHIP_FAST_MATH
