// wave_cull.h -- wave64-level keyframe culling.
//
// The reference tests every surfel against every keyframe in every pass (K x N association
// tests; B/kernel_opt_geometry.cc:114-117 etc.), although a keyframe sees only a few percent of
// the surfels.  Here a wavefront owns 64 consecutive surfels (spatially compact because surfels
// are created in 8x8-cell tile order), computes their bounding sphere with cross-lane min/max,
// and tests that sphere against the frustum of 64 keyframes at a time (one keyframe per lane).
// The surviving keyframes form a 64-bit ballot mask held in SGPRs; the per-surfel work then
// loops over the set bits only, in ascending keyframe order (so floating-point accumulation
// order -- and therefore every result bit -- is identical to the un-culled sweep).
//
// The test is conservative: a keyframe is skipped only if NO point of the (inflated) sphere can
// project into its image with z > 0, which is a precondition of association
// (B/surfel_projection_nvcc_only.cuh:332-345).  It never changes results.
#pragma once

#include "ba_device.h"

namespace bahip {

// Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD b % 8), each with its own L2.  With the surfel
// buffer in spatial order, neighbouring tiles read the same image lines; giving every XCD runs of consecutive tiles
// (instead of every 8th tile) lets those lines be shared in one L2, while the runs still interleave over the XCDs for
// balance.  Run length: 128 tiles for a grid that fills the chip many times over (measured at 46.9 k tiles, it/s: 32 -> 474,
// 64 -> 469, 96 -> 470, 128 -> 487, 192 -> 473, 256 -> 479, 512 -> 468), 32 for smaller grids (a shard of a multi-GPU run:
// 5.9 k tiles are 5.7 runs of 128 per XCD, i.e. 20 % imbalance).  The grid is padded to whole runs on all 8 XCDs
// (xcd_padded_tiles); the kernel derives the run length from the padded grid size.
constexpr uint32_t kXcdLargeGrid = 32768;
__host__ __device__ __forceinline__ uint32_t xcd_padded_tiles(uint32_t tiles) {
  uint32_t padded = (tiles + 255u) / 256u * 256u;                          // 8 XCDs x runs of 32
  if (padded >= kXcdLargeGrid) padded = (tiles + 1023u) / 1024u * 1024u;   // 8 XCDs x runs of 128
  return padded;
}
// The tile that workgroup `block` of a grid of xcd_padded_tiles(...) workgroups owns.
__device__ __forceinline__ uint32_t xcd_chunked_tile(uint32_t block) {
  const uint32_t shift = gridDim.x >= kXcdLargeGrid ? 7u : 5u;   // wave-uniform
  const uint32_t xcd = block & 7u, j = block >> 3;
  return ((((j >> shift) << 3) + xcd) << shift) + (j & ((1u << shift) - 1u));
}

// The same for a persistent grid: the tile that position `block` of a grid of `padded_tiles` one-tile workgroups would own.
__device__ __forceinline__ uint32_t xcd_chunked_tile_of(uint32_t block, uint32_t padded_tiles) {
  const uint32_t shift = padded_tiles >= kXcdLargeGrid ? 7u : 5u;
  const uint32_t xcd = block & 7u, j = block >> 3;
  return ((((j >> shift) << 3) + xcd) << shift) + (j & ((1u << shift) - 1u));
}

// Heavy work first.  A tile's work is proportional to the keyframes that see it -- 20 on average at the bench size, 80 in the
// middle of the scene, and far more for the few tiles that straddle a jump of the Morton curve (a bounding sphere that culls
// nothing) -- so with the tiles taken in buffer order a launch ends with a tail: measured with scripts/tile_timeline.py, both
// sweeps spent 18 % of their duration with fewer than half of the 4096 wavefront slots busy (914 us where 745 us of perfectly
// packed work was done; the last 120 us belonged to EIGHT tiles of 180-270 us each).  The schedule (one buffer of words, built
// by tile_order_kernel in kernels_pose.hip from the per-tile candidate counts the pose sweep's first round records):
//   [0]                           number of heavy tiles (cost >= 2.5 x the mean; at most kHeavySlots)
//   [8 .. 8 + kHeavySlots)        the heavy tiles: the first kHeavySlots positions of a launch run these (or nothing)
//   [kSchedPerm .. + tiles)       the tile of every further position: whole runs of consecutive tiles (the XCD runs: consecutive
//                                 positions go to the 8 XCDs) ranked by the descending cost of their most expensive quarter
//                                 (longest-processing-time-first), each row of eight dealt to the XCDs so that their totals
//                                 stay level; the tiles of the last rows individually by descending cost, so that a launch
//                                 ends on its cheapest tiles
//   [kSchedPerm + tiles ..)       one word per tile: non-zero = heavy, i.e. already done when its regular position comes up
// A launch with a schedule has kHeavySlots more positions than tiles.  NULL = buffer order.  A scheduling hint only: every tile
// is processed exactly once either way, and no result depends on the order.
constexpr uint32_t kHeavySlots = 1024;
constexpr uint32_t kSchedPerm = 8 + kHeavySlots;
__host__ __device__ __forceinline__ uint32_t xcd_run_count(uint32_t padded_tiles) { return padded_tiles >> (padded_tiles >= kXcdLargeGrid ? 7u : 5u); }
__host__ __device__ __forceinline__ size_t sched_words(uint32_t padded_tiles) { return (size_t)kSchedPerm + 2 * (size_t)padded_tiles; }
// Positions of a launch over `padded_tiles` tiles.
__host__ __device__ __forceinline__ uint32_t sched_positions(uint32_t padded_tiles, const uint32_t* sched) { return padded_tiles + (sched ? kHeavySlots : 0u); }
// The tile of buffer-order position `block` (the XCD runs of xcd_chunked_tile).
__host__ __device__ __forceinline__ uint32_t xcd_run_tile(uint32_t block, uint32_t padded_tiles) {
  const uint32_t shift = padded_tiles >= kXcdLargeGrid ? 7u : 5u;
  const uint32_t xcd = block & 7u, j = block >> 3;
  return ((((j >> shift) << 3) + xcd) << shift) + (j & ((1u << shift) - 1u));
}
// The tile position `position` of a launch works on; false = nothing to do there.  Everything is wave-uniform (scalar loads).
__device__ __forceinline__ bool scheduled_tile(uint32_t position, uint32_t padded_tiles, const uint32_t* __restrict__ sched, uint32_t* tile_out) {
  if (!sched) {
    *tile_out = xcd_run_tile(position, padded_tiles);
    return true;
  }
  if (position < kHeavySlots) {
    if (position >= sched[0]) return false;
    *tile_out = sched[8 + position];
    return true;
  }
  const uint32_t tile = sched[kSchedPerm + (position - kHeavySlots)];
  *tile_out = tile;
  return sched[kSchedPerm + padded_tiles + tile] == 0;
}

struct WaveBounds {
  float cx, cy, cz;   // sphere centre (global frame)
  float r;            // inflated radius; negative = no valid surfel in this wave
};

// All-lanes min / max with the cross-lane primitives of wave_reduce.h (no LDS traffic).
template <typename Op>
__device__ __forceinline__ float wave_all(float v, Op op) {
  {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  v = op(v, dpp::mov<dpp::kRowRor8>(v));
  v = op(v, __uint_as_float(__builtin_amdgcn_ds_swizzle(__float_as_uint(v), 0x101F)));
  v = op(v, dpp::mov<dpp::kQuadXor2>(v));
  v = op(v, dpp::mov<dpp::kQuadXor1>(v));
  return v;
}
__device__ __forceinline__ float wave_min(float v) { return wave_all(v, [](float a, float b) { return fminf(a, b); }); }
__device__ __forceinline__ float wave_max(float v) { return wave_all(v, [](float a, float b) { return fmaxf(a, b); }); }

// Bounding sphere of the positions held by the lanes with valid == true (NaN positions -- deleted
// surfels -- must be passed as valid == false).
__device__ __forceinline__ WaveBounds wave_bounds(Vec3 p, bool valid) {
  const float inf = __builtin_huge_valf();
  const float minx = wave_min(valid ? p.x : inf), maxx = wave_max(valid ? p.x : -inf);
  const float miny = wave_min(valid ? p.y : inf), maxy = wave_max(valid ? p.y : -inf);
  const float minz = wave_min(valid ? p.z : inf), maxz = wave_max(valid ? p.z : -inf);
  WaveBounds b;
  if (!(maxx >= minx)) { b.cx = b.cy = b.cz = 0.f; b.r = -1.f; return b; }
  b.cx = 0.5f * (minx + maxx); b.cy = 0.5f * (miny + maxy); b.cz = 0.5f * (minz + maxz);
  const float dx = maxx - b.cx, dy = maxy - b.cy, dz = maxz - b.cz;
  // half-diagonal of the box, inflated by 0.1 % + 1 cm to dominate every rounding error below
  b.r = sqrtf(dx * dx + dy * dy + dz * dz) * 1.001f + 0.01f;
  return b;
}

// Can any point within distance b.r of the centre project into the image of the frame with
// frame_T_global = F?  Planes through the camera centre: px >= 0, px < W, py >= 0, py < H, z > 0.
__device__ __forceinline__ bool sphere_may_project(const Intrinsics& in, const float* F, const WaveBounds& b) {
  if (b.r < 0.f) return false;
  const float lx = F[0] * b.cx + F[1] * b.cy + F[2] * b.cz + F[3];
  const float ly = F[4] * b.cx + F[5] * b.cy + F[6] * b.cz + F[7];
  const float lz = F[8] * b.cx + F[9] * b.cy + F[10] * b.cz + F[11];
  if (lz + b.r <= 0.f) return false;
  const float w = (float)in.width, h = (float)in.height;
  // left:  fx*x + cx*z >= 0 ; right: fx*x + (cx - W)*z < 0   (z > 0)
  if (in.fx * lx + in.cx * lz + b.r * sqrtf(in.fx * in.fx + in.cx * in.cx) < 0.f) return false;
  if (in.fx * lx + (in.cx - w) * lz - b.r * sqrtf(in.fx * in.fx + (in.cx - w) * (in.cx - w)) > 0.f) return false;
  if (in.fy * ly + in.cy * lz + b.r * sqrtf(in.fy * in.fy + in.cy * in.cy) < 0.f) return false;
  if (in.fy * ly + (in.cy - h) * lz - b.r * sqrtf(in.fy * in.fy + (in.cy - h) * (in.cy - h)) > 0.f) return false;
  return true;
}

// The lane's own candidate item for the test above: the 12 coefficients of F and (optionally) a flag word of the same record,
// all loads in flight at once and ALL ARRIVED on return.  (With the loads left to the short-circuit tests, a lane that
// fails an early test leaves later loads pending; the compiler then has to put an s_waitcnt vmcnt(0) at the top of the
// candidate loop that follows, where it is executed per candidate and also waits for the previous candidate's atomics.)
__device__ __forceinline__ void load_candidate(const float* __restrict__ F, const int32_t* __restrict__ flag_ptr, float (&f)[12], int32_t* flag) {
#pragma unroll
  for (int c = 0; c < 12; ++c) f[c] = load_global(F + c);
  int32_t v = 0;
  if (flag_ptr) v = load_global(flag_ptr);
  asm volatile("" ::"v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(f[4]), "v"(f[5]), "v"(f[6]), "v"(f[7]), "v"(f[8]), "v"(f[9]),
               "v"(f[10]), "v"(f[11]), "v"(v));
  if (flag) *flag = v;
}
__device__ __forceinline__ bool sphere_may_project_item(const Intrinsics& in, const float* __restrict__ F, const WaveBounds& b) {
  float f[12];
  load_candidate(F, nullptr, f, nullptr);
  return sphere_may_project(in, f, b);
}

// Calls body(k) (k wave-uniform, ascending) for every item k in [0, num_items) with k % parts == part whose
// lane-level predicate pred(k) holds.  The candidate set lives in a 64-bit scalar mask (one item per lane).
// parts > 1 splits the items of one surfel tile over several wavefronts: the per-keyframe pose sums are merged
// by atomics (any split is fine), the per-surfel sums are defined as four interleaved partial sums (kernels_surfel.hip).
template <typename Pred, typename Body>
__device__ __forceinline__ void for_each_candidate(int num_items, Pred pred, Body body, int parts = 1, int part = 0) {
  const int lane = threadIdx.x & 63;
  for (int base = part; base < num_items; base += 64 * parts) {
    const int item = base + lane * parts;
    const bool cand = (item < num_items) && pred(item);
    unsigned long long m = __ballot(cand);
    while (m) {
      const int k = base + __builtin_ctzll(m) * parts;
      m &= m - 1;
      body(k);
    }
  }
}

// The same loop with the candidate masks kept: a first sweep (replay == false) stores the ballot of every 64-item chunk it tests in
// `masks` (LDS or global, one 64-bit word per chunk of this part, at most max_masks), a later sweep over the SAME items with the same
// predicate (replay == true) reads them back instead of loading the items and testing again -- the two passes of the geometry step
// visit the same keyframes (same bounding sphere, same poses, same activations).  Chunks beyond max_masks are tested both times.
template <typename Pred, typename Body>
__device__ __forceinline__ void for_each_candidate_cached(int num_items, Pred pred, Body body, int parts, int part, unsigned long long* masks,
                                                          int max_masks, bool replay) {
  const int lane = threadIdx.x & 63;
  int chunk = 0;
  for (int base = part; base < num_items; base += 64 * parts, ++chunk) {
    unsigned long long m;
    if (replay && chunk < max_masks) {
      m = masks[chunk];                     // wave-uniform address
    } else {
      const int item = base + lane * parts;
      const bool cand = (item < num_items) && pred(item);
      m = __ballot(cand);
      if (!replay && chunk < max_masks && lane == 0) masks[chunk] = m;
    }
    while (m) {
      const int k = base + __builtin_ctzll(m) * parts;
      m &= m - 1;
      body(k);
    }
  }
}

// Two-stage form of for_each_candidate_cached: issue(k) starts candidate k -- the projection and its gathers -- and returns its state,
// consume(k, state) finishes it.  Within a chunk candidate k + 1 is issued BEFORE candidate k is consumed, so its gathers are in flight
// while k's arithmetic runs (vmcnt retires in order: the wait in front of consume(k) leaves the younger loads outstanding).  For passes
// whose per-candidate arithmetic is short against a gather's round trip (the normals pass: ~350 issue cycles per candidate, four
// wavefronts per SIMD cover 0.6 us of a ~1 us round trip).  Candidates are consumed in the same ascending order: the same bits.
template <typename State, typename Pred, typename Issue, typename Consume>
__device__ __forceinline__ void for_each_candidate_pipelined(int num_items, Pred pred, Issue issue, Consume consume, int parts, int part,
                                                             unsigned long long* masks, int max_masks, bool replay) {
  const int lane = threadIdx.x & 63;
  int chunk = 0;
  for (int base = part; base < num_items; base += 64 * parts, ++chunk) {
    unsigned long long m;
    if (replay && chunk < max_masks) {
      m = masks[chunk];                     // wave-uniform address
    } else {
      const int item = base + lane * parts;
      const bool cand = (item < num_items) && pred(item);
      m = __ballot(cand);
      if (!replay && chunk < max_masks && lane == 0) masks[chunk] = m;
    }
    if (!m) continue;
    int k = base + __builtin_ctzll(m) * parts;
    m &= m - 1;
    State state = issue(k);
    while (m) {
      const int next = base + __builtin_ctzll(m) * parts;
      m &= m - 1;
      State ahead = issue(next);
      consume(k, state);
      k = next;
      state = ahead;
    }
    consume(k, state);
  }
}
// Like for_each_candidate (one part), but stops as soon as body(k) returns true for the whole wavefront
// (wave-uniform return value), e.g. "every lane has found what it was looking for".
template <typename Pred, typename Body>
__device__ __forceinline__ void for_each_candidate_until(int num_items, Pred pred, Body body) {
  const int lane = threadIdx.x & 63;
  for (int base = 0; base < num_items; base += 64) {
    const int item = base + lane;
    const bool cand = (item < num_items) && pred(item);
    unsigned long long m = __ballot(cand);
    while (m) {
      const int k = base + __builtin_ctzll(m);
      m &= m - 1;
      if (body(k)) return;
    }
  }
}

}  // namespace bahip
