// kernels_lifecycle.hip -- surfel creation, supporting-surfel maps + merging, deletion + radius
// update, compaction (B/ = applications/badslam/src/badslam/):
//   B/kernel_supporting_surfels.{cc,cu}, B/kernel_create_surfels.{cc,cu},
//   B/kernel_delete_surfels.{cc,cu}, B/kernel_compact_surfels.cu.
//
// The reference resolves ownership of a sparse cell with atomicCAS, so which pixel / surfel wins
// is arbitrary and run-to-run different.  Here every choice is made deterministic and identical
// to a sequential sweep in ascending index order: supporting slots hold the (up to) three
// smallest associated surfel indices of a cell (atomicMin insertion chain), the merge decision
// of a surfel is evaluated against exactly those, and a new surfel is created for the first
// qualifying pixel of a cell in row-major order.
#include <hip/hip_fp16.h>
#include <hipcub/hipcub.hpp>

#include "ba_device.h"
#include "ba_launch.h"
#include "wave_cull.h"

namespace bahip {

constexpr int kLcBlock = 256;

__device__ __forceinline__ bool is_deleted_bits(float x) { return __float_as_uint(x) == kDeletedSurfelBits; }

// ---- supporting surfels ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kLcBlock)
supporting_fill_kernel(SupportingView sup, int width, int height) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= width || y >= height) return;
#pragma unroll
  for (int b = 0; b < BAHIP_MERGE_BUFFER_COUNT; ++b) *pitched_ptr(sup.b[b], sup.pitch, y, x) = kInvalidIndex;
}

// A batch of keyframes sweeps the same cloud once per keyframe (creation) or twice (merging), and one keyframe sees a few percent
// of it.  LifecycleBounds: bounding spheres of the 64-surfel tiles [0, tiles), taken once per batch (lifecycle_bounds_kernel);
// a wavefront whose tile cannot project into the keyframe (wave_cull.h: sphere_may_project, the test of the BA sweeps) returns
// before it loads a surfel.  The spheres stay valid through a batch: creation appends behind the bounded tiles, merging only
// marks surfels deleted (a sphere then bounds a superset).  tiles == 0: no bounds, every wavefront sweeps.
struct LifecycleBounds {
  const WaveBounds* spheres;
  uint32_t tiles;
  // (when the batch knows its keyframes: bahip_lifecycle_batch_set_frames) the bounded tiles THIS keyframe can see, in no particular
  // order; the sweep then runs over the list and, behind it, over the tiles appended since the bounds were taken -- a launch of a few
  // hundred workgroups instead of one over the whole cloud whose wavefronts nearly all return at their bound
  const uint32_t* list;
  uint32_t list_count;
};
// The surfel of this thread, or false if there is none.  Wave-uniform tile; without a list: the plain index and the bound test.
__device__ __forceinline__ bool lifecycle_surfel(const Intrinsics& in, const KfEntry& frame, const LifecycleBounds& lb, uint32_t size, uint32_t* index,
                                                 uint32_t block /* of this sweep: blockIdx.x, or its offset inside a launch that runs two sweeps */) {
  const uint32_t t = block * kLcBlock + threadIdx.x;
  if (lb.list) {
    const uint32_t w = t >> 6;
    const uint32_t tile = w < lb.list_count ? lb.list[w] : lb.tiles + (w - lb.list_count);
    *index = tile * 64u + (t & 63u);
    return *index < size;
  }
  *index = t;
  if (t >= size) return false;
  const uint32_t tile = t >> 6;   // wave-uniform: kLcBlock is a multiple of 64
  if (tile >= lb.tiles) return true;
  const WaveBounds wb = lb.spheres[tile];
  return sphere_may_project(in, frame.pose.F, wb);
}
// Which bounded tiles each frame of a batch can see.  One pass: frame f's list lives at lists[f * tiles ...) (room for every tile: 4 bytes
// x tiles x frames, 27 MB at the bench scene), its length in cursors[f]; a workgroup takes 256 tiles and kVisibleFramesPerGroup frames.
// (Round 5 counted first, let the host turn the counts into offsets and ran the sweep again -- two launches of one thread per tile
// looping over all frames, 190 us each, and a host wait in between, three times per BundleAdjustment call.)
constexpr int kVisibleFramesPerGroup = 8;
__global__ void __launch_bounds__(kLcBlock)
lifecycle_visible_tiles_kernel(Intrinsics in, const float* __restrict__ frames_F /* [num_frames][12] */, int num_frames,
                               const WaveBounds* __restrict__ spheres, uint32_t tiles, uint32_t* __restrict__ cursors /* [num_frames], zeroed */,
                               uint32_t* __restrict__ lists /* [num_frames][tiles] */) {
  const uint32_t tile = blockIdx.x * kLcBlock + threadIdx.x;
  WaveBounds wb;
  wb.cx = wb.cy = wb.cz = 0.f; wb.r = -1.f;
  if (tile < tiles) wb = spheres[tile];
  const int f0 = blockIdx.y * kVisibleFramesPerGroup, f1 = min(num_frames, f0 + kVisibleFramesPerGroup);
  // one reservation per workgroup and frame: returning atomics on ONE cursor are served one after the other (a reservation per wavefront
  // was 536 of them per frame)
  __shared__ uint32_t wave_count[kLcBlock / 64];
  __shared__ uint32_t group_base;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int f = f0; f < f1; ++f) {
    const bool visible = sphere_may_project(in, frames_F + 12 * f, wb);   // (r < 0: never)
    const unsigned long long m = __ballot(visible);
    if (lane == 0) wave_count[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t total = 0;
      for (int w = 0; w < kLcBlock / 64; ++w) total += wave_count[w];
      group_base = total ? atomicAdd(&cursors[f], total) : 0u;
    }
    __syncthreads();
    uint32_t before = group_base;
    for (int w = 0; w < wave; ++w) before += wave_count[w];
    if (visible) lists[(size_t)f * tiles + before + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = tile;
    __syncthreads();   // (wave_count and group_base are reused by the next frame)
  }
}
__global__ void __launch_bounds__(kLcBlock)
lifecycle_bounds_kernel(SurfelsView s, uint32_t tiles, WaveBounds* __restrict__ spheres) {
  const uint32_t i = blockIdx.x * kLcBlock + threadIdx.x;
  if ((i >> 6) >= tiles) return;
  const Vec3 gp = surfel_position(s, i);   // tiles are whole: i < s.size
  const WaveBounds wb = wave_bounds(gp, gp.x == gp.x);
  if ((threadIdx.x & 63) == 0) spheres[i >> 6] = wb;
}

// Phase A: every associated surfel offers its index to the cell's slot chain.
// pending_flags (a pipelined merge batch, merge_apply_insert_kernel): a surfel whose word has bit 0 set has been merged away by the
// PREVIOUS keyframe of the batch, whose apply sweep -- the one that writes the NaN -- runs beside this sweep: it is skipped here as
// its NaN position would skip it.
__device__ __forceinline__ void supporting_insert_body(const Intrinsics& in, const KfEntry& frame, const SurfelsView& s, const SupportingView& sup,
                                                       const LifecycleBounds& lb, uint32_t size, uint32_t block, const uint32_t* __restrict__ pending_flags) {
  uint32_t i;
  if (!lifecycle_surfel(in, frame, lb, size, &i, block)) return;
  if (pending_flags && (pending_flags[i] & 1u)) return;
  Assoc r;
  if (!project_associate<false>(in, frame.pose.F, frame.geom, surfel_position(s, i), surfel_normal(s, i), &r, nullptr)) return;
  const int cx = r.px / in.cell, cy = r.py / in.cell;
  uint32_t cur = i;
#pragma unroll
  for (int b = 0; b < BAHIP_MERGE_BUFFER_COUNT; ++b) {
    const uint32_t old = atomicMin(pitched_ptr(sup.b[b], sup.pitch, cy, cx), cur);
    if (old == kInvalidIndex) break;     // an empty slot absorbed the value
    if (old > cur) cur = old;            // displaced a larger index: push it down the chain
  }
}
__global__ void __launch_bounds__(kLcBlock)
supporting_insert_kernel(Intrinsics in, KfEntry frame, SurfelsView s, SupportingView sup, LifecycleBounds lb,
                         const uint32_t* __restrict__ size_on_device /* a creation batch: the cloud's current size lives on the device, s.size bounds it */) {
  supporting_insert_body(in, frame, s, sup, lb, size_on_device ? min(*size_on_device, s.size) : s.size, blockIdx.x, nullptr);
}

__device__ __forceinline__ bool merge_test(const SurfelsView& s, uint32_t a, uint32_t b, float cos_thr, float cell_merge_dist_sq) {
  // B/kernel_supporting_surfels.cu:66-83
  if (!(dot3(surfel_normal(s, b), surfel_normal(s, a)) > cos_thr)) return false;
  const Vec3 pa = surfel_position(s, a), pb = surfel_position(s, b);
  const float min_r = fminf(s.row(kSurfelRadiusSquared)[b], s.row(kSurfelRadiusSquared)[a]);
  const Vec3 d = pb - pa;
  return (d.x * d.x + d.y * d.y + d.z * d.z) < min_r * cell_merge_dist_sq;
}

// Phase B: decide (without modifying positions) which surfels a sequential ascending sweep would
// merge away.  flags: one u32 per surfel.
__global__ void __launch_bounds__(kLcBlock)
merge_decide_kernel(Intrinsics in, KfEntry frame, SurfelsView s, SupportingView sup, float cell_merge_dist_sq,
                    float cos_thr, uint32_t* __restrict__ flags, uint32_t* __restrict__ cell_of /* per surfel: 1 + the sparse cell it is
                    associated with in this keyframe, 0 = none -- merge_apply_kernel empties exactly those cells again */, LifecycleBounds lb) {
  uint32_t i;
  if (!lifecycle_surfel(in, frame, lb, s.size, &i, blockIdx.x)) return;   // merge_apply_kernel visits the same surfels: the other flags are never read
  flags[i] = 0;
  cell_of[i] = 0;
  Assoc r;
  if (!project_associate<false>(in, frame.pose.F, frame.geom, surfel_position(s, i), surfel_normal(s, i), &r, nullptr)) return;
  const int cx = r.px / in.cell, cy = r.py / in.cell;
  cell_of[i] = 1u + (uint32_t)cy * (uint32_t)in.cf_width + (uint32_t)cx;
  const uint32_t s0 = *pitched_ptr(sup.b[0], sup.pitch, cy, cx);
  const uint32_t s1 = *pitched_ptr(sup.b[1], sup.pitch, cy, cx);
  const uint32_t s2 = *pitched_ptr(sup.b[2], sup.pitch, cy, cx);
  if (i == s0) return;
  // deletion state of the occupants of slots 1 and 2 at the time later surfels are tested
  const bool d1 = (s1 != kInvalidIndex) && merge_test(s, s1, s0, cos_thr, cell_merge_dist_sq);
  bool deleted;
  if (i == s1) {
    deleted = d1;
  } else {
    const bool d2 = (s2 != kInvalidIndex) &&
                    (merge_test(s, s2, s0, cos_thr, cell_merge_dist_sq) || (!d1 && merge_test(s, s2, s1, cos_thr, cell_merge_dist_sq)));
    if (i == s2) deleted = d2;
    else deleted = merge_test(s, i, s0, cos_thr, cell_merge_dist_sq) ||
                   (!d1 && merge_test(s, i, s1, cos_thr, cell_merge_dist_sq)) ||
                   (!d2 && merge_test(s, i, s2, cos_thr, cell_merge_dist_sq));
  }
  flags[i] = deleted ? 1u : 0u;
}

// ... and, inside a merge batch that knows its frames (empty_the_planes), leaves the supporting planes as they were before the
// insertion, all slots empty: the surfels that inserted themselves are the associated ones, whose cells merge_decide_kernel has
// recorded -- so the NEXT keyframe of the batch needs no fill launch (capi_lifecycle.hip: determine_supporting_impl; a merge batch of 200
// keyframes is launch-bound).  Otherwise the planes keep the lists, the reference function's second output.
__device__ __forceinline__ void merge_apply_body(const Intrinsics& in, const KfEntry& frame, const SurfelsView& s, const uint32_t* __restrict__ flags,
                                                 const uint32_t* __restrict__ cell_of, const SupportingView& sup, int empty_the_planes,
                                                 uint32_t* __restrict__ deleted_count, const LifecycleBounds& lb, uint32_t block) {
  uint32_t i;
  if (!lifecycle_surfel(in, frame, lb, s.size, &i, block)) return;   // (wave-uniform but for the last tile's tail lanes; the ballot below counts active lanes)
  const bool del = flags[i];
  const uint32_t cell = empty_the_planes ? cell_of[i] : 0u;
  if (cell) {
    const uint32_t cy = (cell - 1u) / (uint32_t)in.cf_width, cx = (cell - 1u) - cy * (uint32_t)in.cf_width;
#pragma unroll
    for (int b = 0; b < BAHIP_MERGE_BUFFER_COUNT; ++b) *pitched_ptr(sup.b[b], sup.pitch, (int)cy, (int)cx) = kInvalidIndex;
  }
  if (del) s.row(kSurfelX)[i] = __uint_as_float(kDeletedSurfelBits);
  const unsigned long long m = __ballot(del);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(deleted_count, (uint32_t)__popcll(m));
}
__global__ void __launch_bounds__(kLcBlock)
merge_apply_kernel(Intrinsics in, KfEntry frame, SurfelsView s, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ cell_of,
                   SupportingView sup, int empty_the_planes, uint32_t* __restrict__ deleted_count, LifecycleBounds lb) {
  merge_apply_body(in, frame, s, flags, cell_of, sup, empty_the_planes, deleted_count, lb, blockIdx.x);
}
// A merge batch that knows its frames, pipelined (bahip_merge_surfels_for_keyframes): the apply sweep of keyframe j and the insert sweep
// of keyframe j + 1 in ONE launch -- workgroups [0, apply_groups) apply, the rest insert -- so that a keyframe costs two dependent
// launches instead of three (the batch is bound by launch dependencies and the latency chains inside these small kernels: 54 us per
// keyframe for 39 us of kernels, profiles/r5_drop_in_trace_by_kernel.csv).  The two sweeps touch disjoint memory but for the surfels
// keyframe j merges away, which keyframe j + 1 must not insert: the insert sweep reads their decision word (supporting_insert_body)
// instead of the NaN the apply sweep is writing; the keyframes alternate between two sets of supporting planes, so the set keyframe
// j + 1 fills is the one keyframe j - 1 left empty.  Same planes, same deletions as the three launches per keyframe.
__global__ void __launch_bounds__(kLcBlock)
merge_apply_insert_kernel(Intrinsics in, KfEntry apply_frame, KfEntry insert_frame, SurfelsView s, const uint32_t* __restrict__ flags,
                          const uint32_t* __restrict__ cell_of, SupportingView apply_sup, SupportingView insert_sup,
                          uint32_t* __restrict__ deleted_count, LifecycleBounds apply_lb, LifecycleBounds insert_lb, uint32_t apply_groups) {
  if (blockIdx.x < apply_groups) merge_apply_body(in, apply_frame, s, flags, cell_of, apply_sup, 1, deleted_count, apply_lb, blockIdx.x);
  else supporting_insert_body(in, insert_frame, s, insert_sup, insert_lb, s.size, blockIdx.x - apply_groups, flags);
}

// ---- a merge batch by cell lists (round 6) ------------------------------------------------------------------------------------------
// Which sparse cell of keyframe j a surfel is associated with does not change during a merge batch (merging only marks surfels deleted),
// and the decision about a surfel needs nothing but the members of its own cell: the three lowest indices still alive and the merge
// tests against them (merge_decide_kernel).  So the batch is cut in two:
//   up front, for ALL keyframes of the batch at once: the associated (surfel, keyframe) PAIRS, grouped by (keyframe, cell) --
//     merge_batch_associate_kernel notes each pair's cell and its rank in the cell (the returning atomic that counts the cell), a
//     library scan turns the counts into offsets, merge_batch_fill_kernel writes each pair to offset + rank (no atomics);
//   then ONE launch per keyframe, one thread per PAIR, no atomics, no planes (merge_pairs_kernel): which of the cell's members earlier
//     keyframes of the batch have deleted, the three lowest indices of the rest, merge_decide_kernel's decision about this pair's surfel.
// A deletion is recorded as the STEP at which it happened (deleted_at[i] = j; the word starts as ~0): for step j a member counts as
// alive iff deleted_at >= j, so the threads of one launch read the same answer whether or not a neighbour has already written its own
// deletion, and positions stay readable; merge_batch_apply_kernel writes the deleted markers once the batch is through.
// Per keyframe that is one dependency chain of four loads (pair -> deleted_at of its mates -> rows -> word) instead of insert (an
// atomicMin chain per surfel), decide and apply.  Same deletions as the keyframe-by-keyframe merges
// (tests/test_gpu_lifecycle_stages.py::test_merge_bit_exact).
__global__ void __launch_bounds__(kLcBlock)
merge_batch_associate_kernel(Intrinsics in, const MergeBatchFrame* __restrict__ frames, SurfelsView s, const uint32_t* __restrict__ lists, uint32_t bounded_tiles,
                             uint32_t* __restrict__ counts /* [n * cells (+ 1)], zeroed */, uint32_t cells, uint32_t* __restrict__ pair_cells,
                             uint32_t* __restrict__ pair_ranks) {
  const MergeBatchFrame& f = frames[blockIdx.y];
  const uint32_t all_tiles = (s.size + 63u) / 64u, tail = all_tiles > bounded_tiles ? all_tiles - bounded_tiles : 0u;
  const uint32_t positions = f.list_count + tail;
  for (uint32_t w = (blockIdx.x * kLcBlock + threadIdx.x) >> 6; w < positions; w += gridDim.x * (kLcBlock / 64)) {
    const uint32_t tile = w < f.list_count ? lists[f.list_offset + w] : bounded_tiles + (w - f.list_count);
    const uint32_t i = tile * 64u + (threadIdx.x & 63u);
    const size_t at = ((size_t)f.pair_offset + w) * 64u + (threadIdx.x & 63u);
    uint32_t cell = 0;
    Assoc r;
    if (i < s.size && project_associate<false>(in, f.entry.pose.F, f.entry.geom, surfel_position(s, i), surfel_normal(s, i), &r, nullptr)) {
      cell = 1u + (uint32_t)(r.py / in.cell) * (uint32_t)in.cf_width + (uint32_t)(r.px / in.cell);
      pair_ranks[at] = atomicAdd(&counts[(size_t)blockIdx.y * cells + (cell - 1u)], 1u);
    }
    pair_cells[at] = cell;
  }
}
__global__ void __launch_bounds__(kLcBlock)
merge_batch_fill_kernel(const MergeBatchFrame* __restrict__ frames, uint32_t surfels_size, const uint32_t* __restrict__ lists, uint32_t bounded_tiles,
                        const uint32_t* __restrict__ offsets, uint32_t cells, const uint32_t* __restrict__ pair_cells, const uint32_t* __restrict__ pair_ranks,
                        uint32_t* __restrict__ members, uint2* __restrict__ member_cell /* begin, count of the member's cell */) {
  const MergeBatchFrame& f = frames[blockIdx.y];
  const uint32_t all_tiles = (surfels_size + 63u) / 64u, tail = all_tiles > bounded_tiles ? all_tiles - bounded_tiles : 0u;
  const uint32_t positions = f.list_count + tail;
  for (uint32_t w = (blockIdx.x * kLcBlock + threadIdx.x) >> 6; w < positions; w += gridDim.x * (kLcBlock / 64)) {
    const size_t at = ((size_t)f.pair_offset + w) * 64u + (threadIdx.x & 63u);
    const uint32_t cell = pair_cells[at];
    if (!cell) continue;
    const uint32_t tile = w < f.list_count ? lists[f.list_offset + w] : bounded_tiles + (w - f.list_count);
    const size_t slot = (size_t)blockIdx.y * cells + (cell - 1u);
    const uint32_t begin = offsets[slot], count = offsets[slot + 1] - begin, pos = begin + pair_ranks[at];
    members[pos] = tile * 64u + (threadIdx.x & 63u);
    member_cell[pos] = make_uint2(begin, count);
  }
}
// One keyframe (step) of the batch: thread p decides the surfel of pair first + p.  The rows of the (up to) four surfels a decision
// looks at are requested together, before any test: merge_test's early return would put a round trip to memory between its loads.
struct MergeRows { Vec3 p, n; float r2; };
__device__ __forceinline__ MergeRows merge_rows(const SurfelsView& s, uint32_t i) {
  MergeRows r;
  r.p = surfel_position(s, i); r.n = surfel_normal(s, i); r.r2 = s.row(kSurfelRadiusSquared)[i];
  return r;
}
__device__ __forceinline__ bool merge_test_rows(const MergeRows& a, const MergeRows& b, float cos_thr, float cell_merge_dist_sq) {   // merge_test(s, a, b, ...)
  if (!(dot3(b.n, a.n) > cos_thr)) return false;
  const float min_r = fminf(b.r2, a.r2);
  const Vec3 d = b.p - a.p;
  return (d.x * d.x + d.y * d.y + d.z * d.z) < min_r * cell_merge_dist_sq;
}
__global__ void __launch_bounds__(kLcBlock)
merge_pairs_kernel(SurfelsView s, const uint32_t* __restrict__ members, const uint2* __restrict__ member_cell,
                   uint32_t first, uint32_t end /* the pairs of this frame */, uint32_t step,
                   uint32_t* __restrict__ deleted_at, float cell_merge_dist_sq, float cos_thr) {
  for (uint32_t pos = first + blockIdx.x * kLcBlock + threadIdx.x; pos < end; pos += gridDim.x * kLcBlock) {
    const uint2 c = member_cell[pos];
    const uint32_t i = members[pos];
    if (c.y < 2u) continue;
    // the three lowest indices among the cell's members that no EARLIER step has deleted: the cell's slots.  The members of a cell are
    // neighbours in `members` (this pair is one of them): up to eight are requested at once, their words at once, before any is looked at
    uint32_t s0 = kInvalidIndex, s1 = kInvalidIndex, s2 = kInvalidIndex;
    auto offer = [&](uint32_t k, uint32_t at) {
      if (at < step) return;
      if (k < s0) { const uint32_t t = s0; s0 = k; k = t; }
      if (k < s1) { const uint32_t t = s1; s1 = k; k = t; }
      if (k < s2) s2 = k;
    };
    {
      uint32_t mate[8], at[8];
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) mate[k] = members[c.x + (k < c.y ? k : 0u)];
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) at[k] = deleted_at[mate[k]];
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) if (k < c.y) offer(mate[k], at[k]);
    }
    for (uint32_t k = 8u; k < c.y; ++k) { const uint32_t o = members[c.x + k]; offer(o, deleted_at[o]); }
    if (i == s0 || s1 == kInvalidIndex || deleted_at[i] < step) continue;
    const MergeRows r0 = merge_rows(s, s0), r1 = merge_rows(s, s1), r2 = merge_rows(s, s2 != kInvalidIndex ? s2 : s1), ri = merge_rows(s, i);
    // merge_decide_kernel's decisions (B/kernel_supporting_surfels.cu:60-86 as a sequential ascending sweep would take them)
    const bool d1 = merge_test_rows(r1, r0, cos_thr, cell_merge_dist_sq);
    bool del;
    if (i == s1) {
      del = d1;
    } else {
      const bool d2 = (s2 != kInvalidIndex) && (merge_test_rows(r2, r0, cos_thr, cell_merge_dist_sq) || (!d1 && merge_test_rows(r2, r1, cos_thr, cell_merge_dist_sq)));
      if (i == s2) del = d2;
      else del = merge_test_rows(ri, r0, cos_thr, cell_merge_dist_sq) || (!d1 && merge_test_rows(ri, r1, cos_thr, cell_merge_dist_sq)) ||
                 (!d2 && merge_test_rows(ri, r2, cos_thr, cell_merge_dist_sq));
    }
    // (counted by merge_batch_apply_kernel: an atomic per wavefront on ONE counter here -- ~2000 per launch, served one after the other --
    // cost 3.4 of the launch's 16 us)
    if (del) deleted_at[i] = step;
  }
}
// frame_first[j] = the first pair of frame j (offsets[j * cells]); [n] = all pairs
__global__ void __launch_bounds__(kLcBlock)
merge_batch_frame_first_kernel(const uint32_t* __restrict__ offsets, uint32_t cells, uint32_t num_frames, uint32_t* __restrict__ frame_first) {
  const uint32_t j = blockIdx.x * kLcBlock + threadIdx.x;
  if (j <= num_frames) frame_first[j] = offsets[(size_t)j * cells];
}
// ... and counts them: one atomic per workgroup of a fixed grid
__global__ void __launch_bounds__(kLcBlock)
merge_batch_apply_kernel(SurfelsView s, const uint32_t* __restrict__ deleted_at, uint32_t* __restrict__ deleted_count) {
  __shared__ uint32_t wave_counts[kLcBlock / 64];
  uint32_t mine = 0;
  for (uint32_t i = blockIdx.x * kLcBlock + threadIdx.x; i < s.size; i += gridDim.x * kLcBlock) {
    if (deleted_at[i] != kInvalidIndex) { s.row(kSurfelX)[i] = __uint_as_float(kDeletedSurfelBits); ++mine; }
  }
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
  if ((threadIdx.x & 63) == 0) wave_counts[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int w = 0; w < kLcBlock / 64; ++w) total += wave_counts[w];
    if (total) atomicAdd(deleted_count, total);
  }
}

// ---- creation ---------------------------------------------------------------------------------------
// New surfels are numbered tile-major: tiles of 8x8 sparse cells (TP = 8*cell pixels per side),
// row-major inside a tile, so that the 64 surfels of a wavefront form a compact patch that
// wave_cull.h can bound tightly.  The reference numbers them by a prefix sum over the row-major
// pixel index (B/kernel_create_surfels.cu:357-390).  The order IS observable once surfels merge -- of
// two mergeable surfels the lower index survives (B/kernel_supporting_surfels.cu:60-86): ~1 % of the
// survivors of tests/golden/e2e_vga.npz -- so the reference's order is offered as a mode: with one
// "tile" that covers the whole image (Intrinsics::create_tile >= width, height) the same sequence
// is y * create_tile + x, i.e. row-major (bahip_context_set_creation_order; tests/test_gpu_e2e_vga.py
// holds that mode against the unmodified reference run).
// The flag / index vectors are laid out in that sequence order, padded to whole tiles.
__device__ __forceinline__ int tile_side(const Intrinsics& in) { return in.create_tile; }
__device__ __forceinline__ int tiles_per_row(const Intrinsics& in) { return (in.width + tile_side(in) - 1) / tile_side(in); }
__device__ __forceinline__ size_t tile_seq(const Intrinsics& in, int x, int y) {
  const int tp = tile_side(in);
  const int tx = x / tp, ty = y / tp, lx = x - tx * tp, ly = y - ty * tp;
  return (((size_t)ty * tiles_per_row(in) + tx) * tp + ly) * tp + lx;
}
__device__ __forceinline__ bool tile_xy(const Intrinsics& in, size_t seq, int* x, int* y) {
  const int tp = tile_side(in);
  const int lx = (int)(seq % tp); seq /= tp;
  const int ly = (int)(seq % tp); seq /= tp;
  const int tpr = tiles_per_row(in);
  const int tx = (int)(seq % tpr), ty = (int)(seq / tpr);
  *x = tx * tp + lx; *y = ty * tp + ly;
  return *x < in.width && *y < in.height;
}

// One thread per sparse cell: B/kernel_create_surfels.cu:41-75 with a deterministic winner.
__global__ void __launch_bounds__(kLcBlock)
create_flag_kernel(Intrinsics in, KfEntry frame, SupportingView sup, uint8_t* __restrict__ flags /* tile-major, padded */,
                   int leave_planes_empty /* a creation batch, not its last keyframe: the next keyframe needs no fill launch */) {
  const int cxy = blockIdx.x * kLcBlock + threadIdx.x;
  if (cxy >= in.cf_width * in.cf_height) return;
  const int cy = cxy / in.cf_width, cx = cxy - cy * in.cf_width;
  uint32_t* slot = pitched_ptr(sup.b[0], sup.pitch, cy, cx);
  const bool free_cell = (*slot == kInvalidIndex);
  bool claimed = false;
  for (int dy = 0; dy < in.cell; ++dy) {
    for (int dx = 0; dx < in.cell; ++dx) {
      const int x = cx * in.cell + dx, y = cy * in.cell + dy;
      if (x >= in.width || y >= in.height) continue;
      bool flag = false;
      if (free_cell && !claimed && x >= 1 && y >= 1 && x < in.width - 1 && y < in.height - 1 &&
          !(pitched_load(frame.depth, frame.depth_pitch, y, x) & kInvalidDepthBit)) {
        flag = true;
        claimed = true;
      }
      flags[tile_seq(in, x, y)] = flag ? 1 : 0;
    }
  }
  if (leave_planes_empty) {   // this thread is the only reader of its cell: it may as well empty it for the next keyframe of the batch
#pragma unroll
    for (int b = 0; b < BAHIP_MERGE_BUFFER_COUNT; ++b) *pitched_ptr(sup.b[b], sup.pitch, cy, cx) = kInvalidIndex;
  } else if (claimed) {
    *slot = 0;
  }
}

// B/kernel_create_surfels.cu:213-276 + :314-337: outlier filter for new surfels.
//
// One candidate pixel against one co-visible keyframe: bit 0 = an observation, bit 1 = a free-space violation.
__device__ __forceinline__ uint32_t create_filter_pair(const Intrinsics& in, const KfEntry* __restrict__ kfs, const int* __restrict__ covis,
                                                       const float* __restrict__ covis_T_frame, int c, const Vec3& input_pos, const Vec3& m) {
  const KfEntry& ck = kfs[covis[c]];
  const float* M = covis_T_frame + 12 * c;
  Vec3 lp;
  lp.z = M[8] * input_pos.x + M[9] * input_pos.y + M[10] * input_pos.z + M[11];
  if (!(lp.z > 0.f)) return 0;
  lp.x = M[0] * input_pos.x + M[1] * input_pos.y + M[2] * input_pos.z + M[3];
  lp.y = M[4] * input_pos.x + M[5] * input_pos.y + M[6] * input_pos.z + M[7];
  const float pxx = in.fx * (lp.x / lp.z) + in.cx, pxy = in.fy * (lp.y / lp.z) + in.cy;
  if (!(pxx >= 0.f) || !(pxy >= 0.f) || !(pxx < (float)in.width) || !(pxy < (float)in.height)) return 0;
  const int px = (int)pxx, py = (int)pxy;
  // B/surfel_projection_nvcc_only.cuh:131-231
  const uint16_t raw = pitched_load(ck.depth, ck.depth_pitch, py, px);
  if (raw & kInvalidDepthBit) return 0;
  const Vec3 nl = rotate34(M, m);
  const float d = raw_to_calibrated_depth(in.a, cfactor_at(in, px, py), in.raw_to_float_depth, raw);
  const float thr = 10.f * depth_stddev(unp_nx(in, (float)px), unp_ny(in, (float)py), d, nl, in.baseline_fx);
  const float diff = d - lp.z;
  if (diff > thr) return 2;
  else if (diff < -thr) return 0;
  if (dot3(lp, nl) > 0) return 0;   // sign of (1 / |p|) * dot(p, n), see project_associate
  if (dot3(nl, unpack_normal8(pitched_load(ck.normals, ck.normals_pitch, py, px))) < kCosNormalCompat) return 0;
  return 1;
}

// One lane per pixel cell.  Past the first keyframes of a scene most cells are supported by existing surfels and a wavefront
// holds a handful of candidates: a lane-per-candidate loop over the co-visible keyframes then runs at 1-2 lanes in 64 and is a
// chain of dependent gathers (79 us per keyframe at the bench scene, round 4 trace).  With few candidates the wavefront takes
// them one at a time and spreads the co-visible keyframes over its lanes; the two counts are integers, so both shapes decide
// alike.
__device__ __forceinline__ void create_filter_body(const Intrinsics& in, const KfEntry& frame, const KfEntry* __restrict__ kfs, const int* __restrict__ covis,
                                                   const float* __restrict__ covis_T_frame /* 12 floats each */, int n_covis,
                                                   int min_observation_count, int padded_count, uint8_t* __restrict__ flags, int idx) {
  const int lane = threadIdx.x & 63;
  int x = 0, y = 0;
  const bool candidate = idx < padded_count && flags[idx] && tile_xy(in, (size_t)idx, &x, &y);
  unsigned long long todo = __ballot(candidate);
  if (!todo) return;
  Vec3 input_pos = mk3(0, 0, 0), m = mk3(0, 0, 0);
  if (candidate) {
    const float cd = raw_to_calibrated_depth(in.a, cfactor_at(in, x, y), in.raw_to_float_depth, pitched_load(frame.depth, frame.depth_pitch, y, x));
    input_pos = unproject(in, x, y, cd);
    m = unpack_normal8(pitched_load(frame.normals, frame.normals_pitch, y, x));
  }
  uint32_t observations = 1, violations = 0;
  const int passes = (n_covis + 63) >> 6;
  if (__popcll(todo) * passes < n_covis) {
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      Vec3 cp, cm;
      cp.x = __shfl(input_pos.x, src); cp.y = __shfl(input_pos.y, src); cp.z = __shfl(input_pos.z, src);
      cm.x = __shfl(m.x, src); cm.y = __shfl(m.y, src); cm.z = __shfl(m.z, src);
      uint32_t obs = 0, vio = 0;
      for (int c = lane; c < passes * 64; c += 64) {
        const uint32_t r = c < n_covis ? create_filter_pair(in, kfs, covis, covis_T_frame, c, cp, cm) : 0u;
        obs += (uint32_t)__popcll(__ballot(r & 1u));
        vio += (uint32_t)__popcll(__ballot(r & 2u));
      }
      if (lane == src) { observations += obs; violations += vio; }
    }
  } else if (candidate) {
    for (int c = 0; c < n_covis; ++c) {
      const uint32_t r = create_filter_pair(in, kfs, covis, covis_T_frame, c, input_pos, m);
      observations += r & 1u;
      violations += r >> 1;
    }
  }
  if (candidate && (observations < (uint32_t)min_observation_count || violations > observations)) flags[idx] = 0;
}
__global__ void __launch_bounds__(kLcBlock)
create_filter_kernel(Intrinsics in, KfEntry frame, const KfEntry* __restrict__ kfs, const int* __restrict__ covis,
                     const float* __restrict__ covis_T_frame, int n_covis, int min_observation_count, int padded_count, uint8_t* __restrict__ flags) {
  create_filter_body(in, frame, kfs, covis, covis_T_frame, n_covis, min_observation_count, padded_count, flags, blockIdx.x * kLcBlock + threadIdx.x);
}

// B/kernel_create_surfels.cu:91-160: the attributes of the surfel that pixel (x, y) of `frame` creates, written to index si.
__device__ __forceinline__ void append_surfel(const Intrinsics& in, const KfEntry& frame, int x, int y, uint32_t si, const SurfelsView& s,
                                              Vec3* stored_position = nullptr, uint32_t* stored_normal = nullptr) {
  float G[12];
  {
    // global_T_frame as 3x4: rotation = transpose of frame_T_global's, translation from the pose
    const float* q = frame.global_T_frame;
    const float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const float tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const float twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const float txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const float tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    G[0] = 1 - (tyy + tzz); G[1] = txy - twz;       G[2] = txz + twy;        G[3] = q[4];
    G[4] = txy + twz;       G[5] = 1 - (txx + tzz); G[6] = tyz - twx;        G[7] = q[5];
    G[8] = txz - twy;       G[9] = tyz + twx;       G[10] = 1 - (txx + tyy); G[11] = q[6];
  }
  const float cd = raw_to_calibrated_depth(in.a, cfactor_at(in, x, y), in.raw_to_float_depth, pitched_load(frame.depth, frame.depth_pitch, y, x));
  const Vec3 gp = transform34(G, unproject(in, x, y, cd));
  s.row(kSurfelX)[si] = gp.x; s.row(kSurfelY)[si] = gp.y; s.row(kSurfelZ)[si] = gp.z;
  const Vec3 gn = rotate34(G, unpack_normal8(pitched_load(frame.normals, frame.normals_pitch, y, x)));
  const uint32_t packed_normal = pack_normal10(gn);
  reinterpret_cast<uint32_t*>(s.row(kSurfelNormal))[si] = packed_normal;
  if (stored_position) *stored_position = gp;
  if (stored_normal) *stored_normal = packed_normal;
  const float radius_sq = __half2float(__ushort_as_half(pitched_load(frame.radius, frame.radius_pitch, y, x)));
  s.row(kSurfelRadiusSquared)[si] = radius_sq;
  float cx, cy;
  depth_to_color_pixel(in, x + 0.5f, y + 0.5f, &cx, &cy);
  // colour: bilinear RGB sample, truncated to u8
  {
    float xb = cx - 0.5f, yb = cy - 0.5f;
    const int w = in.cwidth, h = in.cheight;
    if (!(xb >= -1.f)) xb = -1.f; if (xb > (float)w) xb = (float)w;
    if (!(yb >= -1.f)) yb = -1.f; if (yb > (float)h) yb = (float)h;
    const float fx = floorf(xb), fy = floorf(yb), a = xb - fx, b = yb - fy;
    const int x0 = max(0, min((int)fx, w - 1)), x1 = max(0, min((int)fx + 1, w - 1));
    const int y0 = max(0, min((int)fy, h - 1)), y1 = max(0, min((int)fy + 1, h - 1));
    uint8_t out[4] = {0, 0, 0, 0};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float tl = frame.color[(size_t)y0 * frame.color_pitch + 4 * x0 + ch] * (1.0f / 255.0f);
      const float tr = frame.color[(size_t)y0 * frame.color_pitch + 4 * x1 + ch] * (1.0f / 255.0f);
      const float bl = frame.color[(size_t)y1 * frame.color_pitch + 4 * x0 + ch] * (1.0f / 255.0f);
      const float br = frame.color[(size_t)y1 * frame.color_pitch + 4 * x1 + ch] * (1.0f / 255.0f);
      const float top = tl + a * (tr - tl), bot = bl + a * (br - bl);
      out[ch] = (uint8_t)(255.f * (top + b * (bot - top)));
    }
    reinterpret_cast<uchar4*>(s.row(kSurfelColor))[si] = make_uchar4(out[0], out[1], out[2], 0);
  }
  // descriptors are initialised so that both residuals are zero in the creating keyframe
  const Vec3 gn_stored = gn;  // the reference uses the unquantised normal here (B/kernel_create_surfels.cu:133-140)
  DescEval e;
  eval_descriptor<false>(in, frame.lumafp, frame.pose.F, gp, gn_stored, radius_sq, cx, cy, 0.f, 0.f, &e);
  s.row(kSurfelDescriptor1)[si] = e.r1;
  s.row(kSurfelDescriptor2)[si] = e.r2;
}

// B/kernel_create_surfels.cu:91-160,357-390
__global__ void __launch_bounds__(kLcBlock)
create_append_kernel(Intrinsics in, KfEntry frame, const uint8_t* __restrict__ flags, const uint32_t* __restrict__ indices,
                     int padded_count, uint32_t surfels_size, SurfelsView s) {
  const int idx = blockIdx.x * kLcBlock + threadIdx.x;
  if (idx >= padded_count) return;
  if (flags[idx] != 1) return;
  int x, y;
  if (!tile_xy(in, (size_t)idx, &x, &y)) return;
  append_surfel(in, frame, x, y, surfels_size + indices[idx] - 1 /* inclusive scan */, s);
}

// The append of a creation BATCH (bahip_create_surfels_for_keyframes): scan, append and the advance of the cloud's size in ONE
// launch -- round 4 spent five launches per keyframe on them (two of the library scan, the append, a one-thread advance, a memset),
// and a batch of 200 keyframes is launch-bound.  gridDim.x <= kAppendGroups workgroups, all resident at once,
// own consecutive slices of the flag sequence:
//   count the flags of the own slice -> publish the count as ONE tagged word (tag << 24 | count; the tag is the keyframe's number in
//   the batch + 1, so a word of the previous keyframe's launch is never mistaken; the words are cleared once per batch) -> every
//   workgroup waits until all slices' words carry the tag (agent-scope atomic loads of the words themselves: the data IS the flag,
//   no fence, no cache write-back / invalidate -- a first version with a separate arrival counter and release / acquire fences took
//   37 us per launch) -> prefix over the counts = the slice's first index, their sum = what the keyframe creates -> the soft failure
//   of B/kernel_create_surfels.cc:162-165 or the append, in sequence order (a block-level scan per 256 flags) -> workgroup 0 writes
//   the new size to the OTHER of two cells (the next keyframe's launches read that one: nobody reads a size while it is written).
// Same surfels at the same indices as flag scan + create_append_kernel.
constexpr int kAppendGroups = 256;
__global__ void __launch_bounds__(kLcBlock)
create_append_fused_kernel(Intrinsics in, KfEntry frame, const uint8_t* __restrict__ flags, int padded_count, SurfelsView s,
                           const uint32_t* __restrict__ size_in, uint32_t* __restrict__ size_out, uint32_t capacity,
                           uint32_t* __restrict__ capacity_exceeded, uint32_t* __restrict__ group_words, uint32_t tag) {
  __shared__ uint32_t wave_sums[kLcBlock / 64];
  __shared__ uint32_t wave_sums_b[kLcBlock / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int groups = (int)gridDim.x;
  const int per_group = (((padded_count + groups - 1) / groups + kLcBlock - 1) / kLcBlock) * kLcBlock;
  const int begin = min(padded_count, (int)blockIdx.x * per_group), end = min(padded_count, begin + per_group);
  // the flags of the own slice (a slice is far below 2^24 entries)
  uint32_t mine = 0;
  for (int idx = begin + tid; idx < end; idx += kLcBlock) mine += flags[idx] == 1 ? 1u : 0u;
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
  if (lane == 0) wave_sums[wave] = mine;
  __syncthreads();
  if (tid == 0) {
    uint32_t total = 0;
    for (int w = 0; w < kLcBlock / 64; ++w) total += wave_sums[w];
    __hip_atomic_store(&group_words[blockIdx.x], (tag << 24) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // every slice's word, once it carries this launch's tag: prefix over the slices in front of this one, and the keyframe's total
  uint32_t before = 0, all = 0;
  for (int g = tid; g < groups; g += kLcBlock) {
    uint32_t word;
    while (((word = __hip_atomic_load(&group_words[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 24) != tag) __builtin_amdgcn_s_sleep(1);
    const uint32_t c = word & 0xffffffu;
    all += c;
    if (g < (int)blockIdx.x) before += c;
  }
  for (int d = 32; d > 0; d >>= 1) { before += __shfl_xor(before, d); all += __shfl_xor(all, d); }
  __syncthreads();   // (wave_sums has been read)
  if (lane == 0) { wave_sums[wave] = before; wave_sums_b[wave] = all; }
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int w = 0; w < kLcBlock / 64; ++w) { base += wave_sums[w]; total += wave_sums_b[w]; }
  const uint32_t size = *size_in;
  const bool fits = (uint64_t)size + total <= capacity;
  if (blockIdx.x == 0 && tid == 0) {
    *size_out = fits ? size + total : size;
    if (!fits) *capacity_exceeded = 1u;       // the soft failure: this keyframe creates nothing
  }
  if (!fits) return;
  uint32_t running = size + base;
  for (int slab = begin; slab < end; slab += kLcBlock) {
    const int idx = slab + tid;
    const bool flagged = idx < end && flags[idx] == 1;
    const unsigned long long m = __ballot(flagged);
    const uint32_t rank_in_wave = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    __syncthreads();   // (the sums of the previous round have been read)
    if (lane == 0) wave_sums[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t waves_before = 0, slab_total = 0;
    for (int w = 0; w < kLcBlock / 64; ++w) { const uint32_t c = wave_sums[w]; slab_total += c; if (w < wave) waves_before += c; }
    int x, y;
    if (flagged && tile_xy(in, (size_t)idx, &x, &y)) append_surfel(in, frame, x, y, running + waves_before + rank_in_wave, s);
    running += slab_total;
  }
}

// ---- a creation batch whose keyframes do not wait for each other's sweeps (round 6) -------------------------------------------------
// The keyframes of a creation batch depend on each other only through the surfels the earlier ones APPEND: whether a sparse cell of
// keyframe j is free is decided by the cloud as it was when the batch began plus what keyframes 0 .. j - 1 of the batch appended, while
// which pixel of a free cell would create a surfel and whether that candidate passes the outlier filter are functions of the
// keyframes' images and poses alone (create_flag_kernel, create_filter_pair read no surfel).  So the batch is cut in two:
//   up front, for ALL keyframes of the batch at once (a handful of launches whatever the batch's length):
//     occupancy[j][cell] <- 1 where a surfel of the cloud at the batch's begin is associated with keyframe j   (create_batch_occupancy_kernel)
//     candidates[j][seq] <- 1 for the first valid pixel of every cell that is free so far                      (create_batch_flag_kernel)
//     candidates[j][seq] <- 0 where the candidate fails the filter                                             (create_batch_filter_kernel)
//     the candidates of all keyframes as ONE compact list in (keyframe, sequence) order (a library scan over the flags), and for every
//     entry its cell and the eight data rows of the surfel it would create -- append_surfel writes them into a record buffer
//     (create_batch_records_kernel): nothing of that depends on the cloud either
//   then the chain, ONE launch per keyframe (create_chain_kernel) instead of four:
//     the candidates of keyframe j whose cell is still free are counted and scanned like create_append_fused_kernel does (tagged-word
//     grid handshake, soft failure, two size cells) and their records COPIED to the end of the cloud; every appended surfel is at once
//     projected into keyframe j + 1 (push), and further workgroups of the same launch project what keyframes 0 .. j - 1 appended into
//     keyframe j + 1 (pull): when the launch ends, occupancy[j + 1] is complete.  The chain's launch is a short dependency chain:
//     candidate cell -> occupancy byte -> handshake -> record rows -> the next keyframe's pixel word.
// Same surfels at the same indices as n one-keyframe creations (tests/test_gpu_lifecycle_stages.py).
__device__ __forceinline__ void mark_occupied(const Intrinsics& in, const float* __restrict__ F, const uint32_t* __restrict__ geom, const Vec3& gp, const Vec3& gn,
                                              uint8_t* __restrict__ occupancy) {
  Assoc r;
  if (!project_associate<false>(in, F, geom, gp, gn, &r, nullptr)) return;
  occupancy[(size_t)(r.py / in.cell) * (size_t)in.cf_width + (size_t)(r.px / in.cell)] = 1;   // (every writer stores the same value)
}
__global__ void __launch_bounds__(kLcBlock)
create_batch_occupancy_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, const CreateBatchItem* __restrict__ items, SurfelsView s /* size: the cloud at the
                              batch's begin */, const uint32_t* __restrict__ lists, uint32_t bounded_tiles, uint8_t* __restrict__ occupancy, size_t cells) {
  const CreateBatchItem item = items[blockIdx.y];
  const KfEntry& frame = kfs[item.kf_index];
  const uint32_t all_tiles = (s.size + 63u) / 64u, tail = all_tiles > bounded_tiles ? all_tiles - bounded_tiles : 0u;
  const uint32_t positions = item.list_count + tail;
  for (uint32_t w = (blockIdx.x * kLcBlock + threadIdx.x) >> 6; w < positions; w += gridDim.x * (kLcBlock / 64)) {
    const uint32_t tile = w < item.list_count ? lists[item.list_offset + w] : bounded_tiles + (w - item.list_count);
    const uint32_t i = tile * 64u + (threadIdx.x & 63u);
    if (i < s.size) mark_occupied(in, frame.pose.F, frame.geom, surfel_position(s, i), surfel_normal(s, i), occupancy + blockIdx.y * cells);
  }
}
// create_flag_kernel's choice for every cell that no surfel occupies so far (the candidate vector starts cleared)
__global__ void __launch_bounds__(kLcBlock)
create_batch_flag_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, const CreateBatchItem* __restrict__ items, const uint8_t* __restrict__ occupancy, size_t cells,
                         uint8_t* __restrict__ candidates, size_t padded_count) {
  const int cxy = blockIdx.x * kLcBlock + threadIdx.x;
  if (cxy >= in.cf_width * in.cf_height) return;
  if (occupancy[blockIdx.y * cells + (size_t)cxy]) return;
  const KfEntry& frame = kfs[items[blockIdx.y].kf_index];
  const int cy = cxy / in.cf_width, cx = cxy - cy * in.cf_width;
  for (int dy = 0; dy < in.cell; ++dy) {
    for (int dx = 0; dx < in.cell; ++dx) {
      const int x = cx * in.cell + dx, y = cy * in.cell + dy;
      if (x >= 1 && y >= 1 && x < in.width - 1 && y < in.height - 1 && !(pitched_load(frame.depth, frame.depth_pitch, y, x) & kInvalidDepthBit)) {
        candidates[blockIdx.y * padded_count + tile_seq(in, x, y)] = 1;
        return;
      }
    }
  }
}

// the outlier filter over the candidates of every keyframe of the batch (a keyframe without co-visible keyframes: one observation, its own)
__global__ void __launch_bounds__(kLcBlock)
create_batch_filter_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, const CreateBatchItem* __restrict__ items, const int* __restrict__ covis,
                           const float* __restrict__ covis_T_frame, int min_observation_count, int padded_count, uint8_t* __restrict__ candidates) {
  const CreateBatchItem item = items[blockIdx.y];
  create_filter_body(in, kfs[item.kf_index], kfs, covis + item.covis_offset, covis_T_frame + 12 * (size_t)item.covis_offset, item.n_covis, min_observation_count,
                     padded_count, candidates + (size_t)blockIdx.y * (size_t)padded_count, blockIdx.x * kLcBlock + threadIdx.x);
}

// The compact candidate list: position scan[idx] - 1 of flagged entry idx = item * padded_count + seq (inclusive scan of the flags).
// For every entry: the sparse cell of its pixel, and the surfel it would create, written by append_surfel itself into `records` -- a
// SurfelsView over a buffer with one column per list position.  first_of_item[j] = the list position of keyframe j's first candidate.
__global__ void __launch_bounds__(kLcBlock)
create_batch_records_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, const CreateBatchItem* __restrict__ items, const uint8_t* __restrict__ candidates,
                            const uint32_t* __restrict__ scan, int padded_count, uint32_t* __restrict__ cand_cell, SurfelsView records,
                            uint32_t* __restrict__ first_of_item /* [n + 1] */) {
  const int seq = blockIdx.x * kLcBlock + threadIdx.x;
  const size_t base = (size_t)blockIdx.y * (size_t)padded_count;
  if (seq == 0) {
    first_of_item[blockIdx.y] = base ? scan[base - 1] : 0u;
    if (blockIdx.y + 1 == gridDim.y) first_of_item[gridDim.y] = scan[base + (size_t)padded_count - 1];
  }
  if (seq >= padded_count || candidates[base + seq] != 1) return;
  int x, y;
  if (!tile_xy(in, (size_t)seq, &x, &y)) return;   // (never: padding is not flagged)
  const uint32_t pos = scan[base + seq] - 1u;
  cand_cell[pos] = (uint32_t)(y / in.cell) * (uint32_t)in.cf_width + (uint32_t)(x / in.cell);
  append_surfel(in, kfs[items[blockIdx.y].kf_index], x, y, pos, records);
}

// One keyframe of the chain.  Workgroups [0, append_groups): the keyframe's candidates [first, end) of the compact list whose cell is
// still free are counted, scanned (create_append_fused_kernel's handshake, soft failure and size cells) and their records copied to
// the cloud, each appended surfel pushed into next_occupancy; the other workgroups: what the batch appended before this keyframe --
// surfels [batch_begin_size, *size_in) -- pulled into next_occupancy.  next_occupancy == nullptr: nothing follows.
__global__ void __launch_bounds__(kLcBlock)
create_chain_kernel(Intrinsics in, KfEntry next_frame, const uint32_t* __restrict__ cand_cell, SurfelsView records, uint32_t first, uint32_t end_of_frame,
                    const uint8_t* __restrict__ occupancy, uint8_t* __restrict__ next_occupancy, SurfelsView s, uint32_t batch_begin_size,
                    const uint32_t* __restrict__ size_in, uint32_t* __restrict__ size_out, uint32_t capacity,
                    uint32_t* __restrict__ capacity_exceeded, uint32_t* __restrict__ group_words, uint32_t tag, uint32_t append_groups) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x >= append_groups) {   // ---- pull
    const uint32_t end = min(*size_in, s.size);
    const uint32_t stride = (gridDim.x - append_groups) * kLcBlock;
    for (uint32_t i = batch_begin_size + (blockIdx.x - append_groups) * kLcBlock + tid; i < end; i += stride)
      mark_occupied(in, next_frame.pose.F, next_frame.geom, surfel_position(s, i), surfel_normal(s, i), next_occupancy);
    return;
  }
  __shared__ uint32_t wave_sums[kLcBlock / 64];
  __shared__ uint32_t wave_sums_b[kLcBlock / 64];
  const uint32_t size = *size_in;   // (requested first: the launch's longest wait is for memory)
  const uint32_t groups = append_groups, count = end_of_frame - first;
  const uint32_t per_group = (((count + groups - 1u) / groups + kLcBlock - 1u) / kLcBlock) * kLcBlock;
  const uint32_t begin = first + min(count, blockIdx.x * per_group), end = first + min(count, blockIdx.x * per_group + per_group);
  // a candidate creates a surfel iff no surfel has come to occupy its cell: the cloud at the batch's begin was looked at before the
  // candidates were chosen, what the batch appended since is in `occupancy` (complete: the previous launch of the chain wrote it)
  uint32_t mine = 0;
  for (uint32_t c = begin + tid; c < end; c += kLcBlock) mine += occupancy[cand_cell[c]] == 0 ? 1u : 0u;
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
  if (lane == 0) wave_sums[wave] = mine;
  __syncthreads();
  if (tid == 0) {
    uint32_t total = 0;
    for (int w = 0; w < kLcBlock / 64; ++w) total += wave_sums[w];
    __hip_atomic_store(&group_words[blockIdx.x], (tag << 24) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  uint32_t before = 0, all = 0;
  for (uint32_t g = tid; g < groups; g += kLcBlock) {
    uint32_t word;
    while (((word = __hip_atomic_load(&group_words[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 24) != tag) __builtin_amdgcn_s_sleep(1);
    const uint32_t n = word & 0xffffffu;
    all += n;
    if (g < blockIdx.x) before += n;
  }
  for (int d = 32; d > 0; d >>= 1) { before += __shfl_xor(before, d); all += __shfl_xor(all, d); }
  __syncthreads();   // (wave_sums has been read)
  if (lane == 0) { wave_sums[wave] = before; wave_sums_b[wave] = all; }
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (int w = 0; w < kLcBlock / 64; ++w) { base += wave_sums[w]; total += wave_sums_b[w]; }
  const bool fits = (uint64_t)size + total <= capacity;
  if (blockIdx.x == 0 && tid == 0) {
    *size_out = fits ? size + total : size;
    if (!fits) *capacity_exceeded = 1u;       // the soft failure: this keyframe creates nothing
  }
  if (!fits) return;
  uint32_t running = size + base;
  for (uint32_t slab = begin; slab < end; slab += kLcBlock) {
    const uint32_t c = slab + tid;
    const bool flagged = c < end && occupancy[cand_cell[c]] == 0;
    const unsigned long long m = __ballot(flagged);
    const uint32_t rank_in_wave = (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    __syncthreads();   // (the sums of the previous round have been read)
    if (lane == 0) wave_sums[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t waves_before = 0, slab_total = 0;
    for (int w = 0; w < kLcBlock / 64; ++w) { const uint32_t n = wave_sums[w]; slab_total += n; if (w < wave) waves_before += n; }
    if (flagged) {
      const uint32_t si = running + waves_before + rank_in_wave;
      float row[kSurfelAccum0];
#pragma unroll
      for (int r = 0; r < kSurfelAccum0; ++r) row[r] = records.row(r)[c];
#pragma unroll
      for (int r = 0; r < kSurfelAccum0; ++r) s.row(r)[si] = row[r];
      if (next_occupancy)   // push: position and (stored, i.e. quantised) normal as a sweep over the cloud would read them back
        mark_occupied(in, next_frame.pose.F, next_frame.geom, mk3(row[kSurfelX], row[kSurfelY], row[kSurfelZ]),
                      unpack_normal10(__float_as_uint(row[kSurfelNormal])), next_occupancy);
    }
    running += slab_total;
  }
}

// ---- deletion + radius update (B/kernel_delete_surfels.cu:42-176), one launch for all keyframes ------
__global__ void __launch_bounds__(kLcBlock)
delete_update_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s,
                     int min_observation_count, uint32_t* __restrict__ deleted_count) {
  const uint32_t i = blockIdx.x * kLcBlock + threadIdx.x;
  bool newly_deleted = false;
  if (i < s.size) {
    const Vec3 gp = surfel_position(s, i);
    const Vec3 gn = surfel_normal(s, i);
    float obs = 0, viol = 0, min_r = __builtin_huge_valf();
    for (int k = 0; k < num_kfs; ++k) {
      Assoc r;
      bool fsv = false;
      if (project_associate<true>(in, kfs[k].pose.F, kfs[k].geom, gp, gn, &r, &fsv)) {
        obs += 1.f;
        min_r = fminf(min_r, __half2float(__ushort_as_half(pitched_load(kfs[k].radius, kfs[k].radius_pitch, r.py, r.px))));
      } else if (fsv) {
        viol += 1.f;
      }
    }
    s.row(kSurfelAccum0 + 0)[i] = obs; s.row(kSurfelAccum0 + 1)[i] = viol; s.row(kSurfelAccum0 + 2)[i] = min_r;
    if (obs < (float)min_observation_count || viol > obs) {
      if (!is_deleted_bits(gp.x)) { s.row(kSurfelX)[i] = __uint_as_float(kDeletedSurfelBits); newly_deleted = true; }
    } else {
      s.row(kSurfelRadiusSquared)[i] = min_r;
    }
  }
  const unsigned long long m = __ballot(newly_deleted);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(deleted_count, (uint32_t)__popcll(m));
}

// ---- compaction (B/kernel_compact_surfels.cu:101-157) --------------------------------------------------
__global__ void __launch_bounds__(kLcBlock)
compact_flag_kernel(SurfelsView s, uint32_t* __restrict__ invalid) {
  const uint32_t i = blockIdx.x * kLcBlock + threadIdx.x;
  if (i < s.size) invalid[i] = is_deleted_bits(s.row(kSurfelX)[i]) ? 1u : 0u;
}
__global__ void __launch_bounds__(kLcBlock)
compact_free_list_kernel(SurfelsView s, const uint32_t* __restrict__ invalid, const uint32_t* __restrict__ free_rank,
                         uint32_t free_spot_count, uint32_t* __restrict__ free_list) {
  const uint32_t i = blockIdx.x * kLcBlock + threadIdx.x;
  if (i < s.size && invalid[i] && free_rank[i] < free_spot_count) free_list[free_rank[i]] = i;
}
__global__ void __launch_bounds__(kLcBlock)
compact_move_kernel(SurfelsView s, const uint32_t* __restrict__ invalid, const uint32_t* __restrict__ free_rank,
                    const uint32_t* __restrict__ free_list, uint32_t free_spot_count, uint32_t surfel_count) {
  const uint32_t i = blockIdx.x * kLcBlock + threadIdx.x;
  if (i >= s.size || invalid[i]) return;
  // number of valid surfels with a larger index = reverse index of this surfel
  const uint32_t valid_before = i - free_rank[i];
  const uint32_t reverse_index = surfel_count - 1 - valid_before;
  if (reverse_index >= free_spot_count) return;
  const uint32_t dst = free_list[reverse_index];
  if (dst < i) {
#pragma unroll
    for (int row = 0; row < 8; ++row) s.row(row)[dst] = s.row(row)[i];
    if (s.active) s.active[dst] = s.active[i];
  }
}

// ---- launchers -------------------------------------------------------------------------------------------
static inline unsigned g1(uint32_t n) { return (n + kLcBlock - 1) / kLcBlock; }

void launch_supporting_fill(hipStream_t st, const SupportingView& sup, int w, int h) {
  hipLaunchKernelGGL(supporting_fill_kernel, dim3((w + 63) / 64, (h + 3) / 4), dim3(kLcBlock), 0, st, sup, w, h);
}
void launch_lifecycle_bounds(hipStream_t st, const SurfelsView& s, uint32_t tiles, void* spheres) {
  if (tiles) hipLaunchKernelGGL(lifecycle_bounds_kernel, dim3(g1(tiles * 64u)), dim3(kLcBlock), 0, st, s, tiles, static_cast<WaveBounds*>(spheres));
}
static LifecycleBounds device_cull(const LifecycleCull& cull) {
  LifecycleBounds lb;
  lb.spheres = static_cast<const WaveBounds*>(cull.spheres);
  lb.tiles = cull.spheres ? cull.tiles : 0u;
  lb.list = cull.spheres ? cull.list : nullptr;
  lb.list_count = lb.list ? cull.list_count : 0u;
  return lb;
}
// workgroups of a per-keyframe sweep: over everything, or over the keyframe's list and the tiles behind the bounded ones
static unsigned sweep_groups(const LifecycleBounds& lb, uint32_t size) {
  if (!lb.list) return g1(size);
  const uint32_t all_tiles = (size + 63u) / 64u, tail = all_tiles > lb.tiles ? all_tiles - lb.tiles : 0u;
  return g1((lb.list_count + tail) * 64u);
}
void launch_lifecycle_visible_tiles(hipStream_t st, const Intrinsics& in, const float* frames_F, int num_frames, const void* spheres, uint32_t tiles,
                                    uint32_t* cursors, uint32_t* lists) {
  if (tiles && num_frames)
    hipLaunchKernelGGL(lifecycle_visible_tiles_kernel, dim3(g1(tiles), (num_frames + kVisibleFramesPerGroup - 1) / kVisibleFramesPerGroup), dim3(kLcBlock), 0, st, in,
                       frames_F, num_frames, static_cast<const WaveBounds*>(spheres), tiles, cursors, lists);
}
void launch_supporting_insert(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s, const SupportingView& sup,
                              const LifecycleCull& cull, const uint32_t* size_on_device) {
  const LifecycleBounds lb = device_cull(cull);
  const unsigned groups = sweep_groups(lb, s.size);
  if (s.size && groups) hipLaunchKernelGGL(supporting_insert_kernel, dim3(groups), dim3(kLcBlock), 0, st, in, frame, s, sup, lb, size_on_device);
}
// flags, cell_of: one word per surfel each (scratch rows).  empty_the_planes: see merge_apply_kernel.
void launch_merge(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s, const SupportingView& sup,
                  float cell_merge_dist_sq, float cos_thr, uint32_t* flags, uint32_t* cell_of, bool empty_the_planes, uint32_t* deleted_count,
                  const LifecycleCull& cull) {
  if (!s.size) return;
  const LifecycleBounds lb = device_cull(cull);
  const unsigned groups = sweep_groups(lb, s.size);
  if (!groups) return;
  hipLaunchKernelGGL(merge_decide_kernel, dim3(groups), dim3(kLcBlock), 0, st, in, frame, s, sup, cell_merge_dist_sq, cos_thr, flags, cell_of, lb);
  hipLaunchKernelGGL(merge_apply_kernel, dim3(groups), dim3(kLcBlock), 0, st, in, frame, s, flags, cell_of, sup, empty_the_planes ? 1 : 0, deleted_count, lb);
}
// The pieces of a pipelined merge batch (merge_apply_insert_kernel): decide alone; apply (of `apply_frame`, may be absent: the batch's
// first step) beside insert (of `insert_frame`, may be absent: its last step).
void launch_merge_decide(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s, const SupportingView& sup,
                         float cell_merge_dist_sq, float cos_thr, uint32_t* flags, uint32_t* cell_of, const LifecycleCull& cull) {
  if (!s.size) return;
  const LifecycleBounds lb = device_cull(cull);
  const unsigned groups = sweep_groups(lb, s.size);
  if (groups) hipLaunchKernelGGL(merge_decide_kernel, dim3(groups), dim3(kLcBlock), 0, st, in, frame, s, sup, cell_merge_dist_sq, cos_thr, flags, cell_of, lb);
}
void launch_merge_apply_insert(hipStream_t st, const Intrinsics& in, const KfEntry* apply_frame, const KfEntry* insert_frame, const SurfelsView& s,
                               const uint32_t* flags, const uint32_t* cell_of, const SupportingView& apply_sup, const SupportingView& insert_sup,
                               uint32_t* deleted_count, const LifecycleCull& apply_cull, const LifecycleCull& insert_cull) {
  if (!s.size) return;
  const LifecycleBounds alb = device_cull(apply_cull), ilb = device_cull(insert_cull);
  const unsigned apply_groups = apply_frame ? sweep_groups(alb, s.size) : 0u, insert_groups = insert_frame ? sweep_groups(ilb, s.size) : 0u;
  if (apply_groups + insert_groups == 0) return;
  const KfEntry& a = apply_frame ? *apply_frame : *insert_frame;
  const KfEntry& i = insert_frame ? *insert_frame : *apply_frame;
  hipLaunchKernelGGL(merge_apply_insert_kernel, dim3(apply_groups + insert_groups), dim3(kLcBlock), 0, st, in, a, i, s, flags, cell_of, apply_sup, insert_sup,
                     deleted_count, alb, ilb, apply_groups);
}
// a merge batch by cell lists: the pairs of all frames up front, then launch_merge_pairs per frame, launch_merge_batch_apply at the end
hipError_t launch_merge_batch_lists(hipStream_t st, const Intrinsics& in, const MergeBatchFrame* frames, int num_frames, uint32_t max_positions, const SurfelsView& s,
                                    const uint32_t* lists, uint32_t bounded_tiles, uint32_t* counts, uint32_t* offsets, uint32_t* pair_cells, uint32_t* pair_ranks,
                                    uint32_t* members, void* member_cell, uint32_t* frame_first, void* scan_temp, size_t scan_temp_bytes) {
  const uint32_t cells = (uint32_t)in.cf_width * (uint32_t)in.cf_height;
  const size_t entries = (size_t)num_frames * cells + 1;
  hipError_t e = hipMemsetAsync(counts, 0, entries * sizeof(uint32_t), st);
  if (e != hipSuccess) return e;
  const dim3 grid(std::max(1u, std::min<unsigned>(g1(max_positions * 64u), 2048u)), num_frames);
  if (max_positions)
    hipLaunchKernelGGL(merge_batch_associate_kernel, grid, dim3(kLcBlock), 0, st, in, frames, s, lists, bounded_tiles, counts, cells, pair_cells, pair_ranks);
  e = hipcub::DeviceScan::ExclusiveSum(scan_temp, scan_temp_bytes, counts, offsets, (int)entries, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(merge_batch_frame_first_kernel, dim3(g1((uint32_t)num_frames + 1u)), dim3(kLcBlock), 0, st, offsets, cells, (uint32_t)num_frames, frame_first);
  if (!max_positions) return hipGetLastError();
  hipLaunchKernelGGL(merge_batch_fill_kernel, grid, dim3(kLcBlock), 0, st, frames, s.size, lists, bounded_tiles, offsets, cells, pair_cells, pair_ranks, members,
                     static_cast<uint2*>(member_cell));
  return hipGetLastError();
}
size_t merge_batch_scan_temp_bytes(size_t entries) {
  size_t bytes = 0;
  hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)entries);
  return bytes;
}
void launch_merge_pairs(hipStream_t st, const SurfelsView& s, const uint32_t* members, const void* member_cell, uint32_t first_pair,
                        uint32_t end_pair, uint32_t step, uint32_t* deleted_at, float cell_merge_dist_sq, float cos_thr) {
  if (end_pair <= first_pair) return;
  hipLaunchKernelGGL(merge_pairs_kernel, dim3(g1(end_pair - first_pair)), dim3(kLcBlock), 0, st, s, members, static_cast<const uint2*>(member_cell),
                     first_pair, end_pair, step, deleted_at, cell_merge_dist_sq, cos_thr);
}
void launch_merge_batch_apply(hipStream_t st, const SurfelsView& s, const uint32_t* deleted_at, uint32_t* deleted_count) {
  if (s.size) hipLaunchKernelGGL(merge_batch_apply_kernel, dim3(std::min<unsigned>(g1(s.size), 512u)), dim3(kLcBlock), 0, st, s, deleted_at, deleted_count);
}
void launch_create_flag(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const SupportingView& sup, uint8_t* flags, bool leave_planes_empty) {
  hipLaunchKernelGGL(create_flag_kernel, dim3(g1(in.cf_width * in.cf_height)), dim3(kLcBlock), 0, st, in, frame, sup, flags, leave_planes_empty ? 1 : 0);
}
size_t create_padded_count(const Intrinsics& in) {
  const size_t tp = (size_t)in.create_tile;
  return ((in.width + tp - 1) / tp) * ((in.height + tp - 1) / tp) * tp * tp;
}
void launch_create_filter(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const KfEntry* kfs, const int* covis,
                          const float* covis_T_frame, int n_covis, int min_obs, uint8_t* flags) {
  const int padded = (int)create_padded_count(in);
  hipLaunchKernelGGL(create_filter_kernel, dim3(g1(padded)), dim3(kLcBlock), 0, st, in, frame, kfs, covis,
                     covis_T_frame, n_covis, min_obs, padded, flags);
}
void launch_create_append(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const uint8_t* flags,
                          const uint32_t* indices, uint32_t surfels_size, const SurfelsView& s) {
  const int padded = (int)create_padded_count(in);
  hipLaunchKernelGGL(create_append_kernel, dim3(g1(padded)), dim3(kLcBlock), 0, st, in, frame, flags, indices, padded, surfels_size, s);
}
int create_append_groups() { return kAppendGroups; }
static int g_append_groups_limit = 0;   // test hook: as if the device held at most that many workgroups of the kernel (0: what it holds)
void set_append_groups_limit(int groups) { g_append_groups_limit = groups > 0 ? groups : 0; }
// group_words: kAppendGroups words, cleared once per batch; tag: the keyframe's number in the batch + 1 (1 .. 255; the caller clears the
// words again before a tag repeats)
void launch_create_append_fused(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const uint8_t* flags, const SurfelsView& s,
                                const uint32_t* size_in, uint32_t* size_out, uint32_t capacity, uint32_t* capacity_exceeded,
                                uint32_t* group_words, uint32_t tag) {
  const int padded = (int)create_padded_count(in);
  // The kernel's grid handshake (every workgroup waits for every other's tagged word) ends only if ALL workgroups of the launch are
  // resident at once, which a plain launch does not promise: the grid is therefore clamped to what the device the stream
  // runs on can hold of this kernel -- occupancy per compute unit x compute units, asked once per device; a partition mode with 38
  // CUs, a CU mask or a build whose register use lowers the occupancy then gets fewer, longer slices (per_group adapts to gridDim).
  static int resident_limit[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (resident_limit[dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, create_append_fused_kernel, kLcBlock, 0) != hipSuccess || per_cu < 1) { per_cu = 1; (void)hipGetLastError(); }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) { cus = 1; (void)hipGetLastError(); }
    resident_limit[dev] = std::max(1, per_cu * cus);
  }
  const int limit = g_append_groups_limit > 0 ? std::min(g_append_groups_limit, resident_limit[dev]) : resident_limit[dev];
  const int groups = std::min(std::min(kAppendGroups, limit), (int)g1(padded));
  hipLaunchKernelGGL(create_append_fused_kernel, dim3(groups), dim3(kLcBlock), 0, st, in, frame, flags, padded, s, size_in, size_out, capacity,
                     capacity_exceeded, group_words, tag);
}
// the grid of the fused append / the chain's append part: what the device can hold of `kernel` at once, at most kAppendGroups
template <typename Kernel>
static int resident_append_groups(Kernel kernel, int* cache /* per device */) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cache[dev] == 0) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kLcBlock, 0) != hipSuccess || per_cu < 1) { per_cu = 1; (void)hipGetLastError(); }
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) { cus = 1; (void)hipGetLastError(); }
    cache[dev] = std::max(1, per_cu * cus);
  }
  const int limit = g_append_groups_limit > 0 ? std::min(g_append_groups_limit, cache[dev]) : cache[dev];
  return std::min(kAppendGroups, limit);
}
struct U8ToU32 {
  __host__ __device__ uint32_t operator()(const uint8_t& v) const { return (uint32_t)v; }
};
size_t create_batch_scan_temp_bytes(size_t entries) {
  size_t bytes = 0;
  hipcub::TransformInputIterator<uint32_t, U8ToU32, const uint8_t*> it((const uint8_t*)nullptr, U8ToU32());
  hipcub::DeviceScan::InclusiveSum(nullptr, bytes, it, (uint32_t*)nullptr, (int)entries);
  return bytes;
}
// records: a view over [kSurfelAccum0 rows][record_capacity columns] floats (its pitch the row length in bytes)
hipError_t launch_create_batch_prepare(hipStream_t st, const Intrinsics& in, const KfEntry* kfs, const CreateBatchItem* items, int num_items, uint32_t max_list_count,
                                       const SurfelsView& cloud_at_begin, const uint32_t* lists, uint32_t bounded_tiles, uint8_t* occupancy, uint8_t* candidates,
                                       bool filter_new_surfels, const int* covis, const float* covis_T_frame, int min_obs, uint32_t* scan, void* scan_temp,
                                       size_t scan_temp_bytes, uint32_t* cand_cell, const SurfelsView& records, uint32_t* first_of_item) {
  if (num_items <= 0) return hipSuccess;
  const size_t cells = (size_t)in.cf_width * (size_t)in.cf_height;
  const int padded = (int)create_padded_count(in);
  const uint32_t all_tiles = (cloud_at_begin.size + 63u) / 64u, tail = all_tiles > bounded_tiles ? all_tiles - bounded_tiles : 0u;
  const uint32_t positions = max_list_count + tail;
  if (cloud_at_begin.size && positions)
    hipLaunchKernelGGL(create_batch_occupancy_kernel, dim3(std::min<unsigned>(g1(positions * 64u), 1024u), num_items), dim3(kLcBlock), 0, st, in, kfs, items,
                       cloud_at_begin, lists, bounded_tiles, occupancy, cells);
  hipLaunchKernelGGL(create_batch_flag_kernel, dim3(g1((uint32_t)cells), num_items), dim3(kLcBlock), 0, st, in, kfs, items, occupancy, cells, candidates, (size_t)padded);
  if (filter_new_surfels)
    hipLaunchKernelGGL(create_batch_filter_kernel, dim3(g1(padded), num_items), dim3(kLcBlock), 0, st, in, kfs, items, covis, covis_T_frame, min_obs, padded, candidates);
  hipcub::TransformInputIterator<uint32_t, U8ToU32, const uint8_t*> it(candidates, U8ToU32());
  const hipError_t e = hipcub::DeviceScan::InclusiveSum(scan_temp, scan_temp_bytes, it, scan, (int)((size_t)num_items * (size_t)padded), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(create_batch_records_kernel, dim3(g1(padded), num_items), dim3(kLcBlock), 0, st, in, kfs, items, candidates, scan, padded, cand_cell, records,
                     first_of_item);
  return hipGetLastError();
}
// appended_bound: an upper bound of what the batch has appended before this keyframe (sizes the pull part of the grid)
void launch_create_chain(hipStream_t st, const Intrinsics& in, const KfEntry* next_frame, const uint32_t* cand_cell, const SurfelsView& records, uint32_t first,
                         uint32_t end_of_frame, const uint8_t* occupancy, uint8_t* next_occupancy, const SurfelsView& s, uint32_t batch_begin_size,
                         const uint32_t* size_in, uint32_t* size_out, uint32_t capacity, uint32_t* capacity_exceeded, uint32_t* group_words, uint32_t tag,
                         uint32_t appended_bound) {
  static int resident_limit[64] = {};
  const unsigned append_groups = (unsigned)std::max(1, std::min(resident_append_groups(create_chain_kernel, resident_limit), (int)g1(end_of_frame - first)));
  const unsigned pull_groups = (next_frame && next_occupancy && appended_bound) ? std::min<unsigned>(g1(appended_bound), 512u) : 0u;
  KfEntry none;
  memset(&none, 0, sizeof(none));
  hipLaunchKernelGGL(create_chain_kernel, dim3(append_groups + pull_groups), dim3(kLcBlock), 0, st, in, next_frame ? *next_frame : none, cand_cell, records, first,
                     end_of_frame, occupancy, (next_frame ? next_occupancy : nullptr), s, batch_begin_size, size_in, size_out, capacity, capacity_exceeded, group_words,
                     tag, append_groups);
}
void launch_delete_update(hipStream_t st, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                          int min_obs, uint32_t* deleted_count) {
  if (s.size) hipLaunchKernelGGL(delete_update_kernel, dim3(g1(s.size)), dim3(kLcBlock), 0, st, in, kfs, num_kfs, s, min_obs, deleted_count);
}

// Inclusive scan u8 -> u32 (new surfel indices) and exclusive scan u32 -> u32 (free ranks).
size_t scan_temp_bytes(size_t n) {
  size_t a = 0, b = 0;
  hipcub::DeviceScan::InclusiveSum(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
  hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n);
  return a > b ? a : b;
}
hipError_t scan_flags_inclusive(hipStream_t st, void* temp, size_t temp_bytes, const uint8_t* flags, uint32_t* out, int n) {
  hipcub::TransformInputIterator<uint32_t, U8ToU32, const uint8_t*> it(flags, U8ToU32());
  return hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, it, out, n, st);
}
hipError_t scan_u32_exclusive(hipStream_t st, void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, int n) {
  return hipcub::DeviceScan::ExclusiveSum(temp, temp_bytes, in, out, n, st);
}

hipError_t launch_compact(hipStream_t st, const SurfelsView& s, uint32_t* invalid, uint32_t* free_rank, uint32_t* free_list,
                          uint32_t surfel_count, void* temp, size_t temp_bytes) {
  if (!s.size) return hipSuccess;
  const uint32_t free_spot_count = s.size - surfel_count;
  hipLaunchKernelGGL(compact_flag_kernel, dim3(g1(s.size)), dim3(kLcBlock), 0, st, s, invalid);
  const hipError_t scan = scan_u32_exclusive(st, temp, temp_bytes, invalid, free_rank, (int)s.size);
  if (scan != hipSuccess) return scan;   // nothing has been moved yet
  hipLaunchKernelGGL(compact_free_list_kernel, dim3(g1(s.size)), dim3(kLcBlock), 0, st, s, invalid, free_rank, free_spot_count, free_list);
  hipLaunchKernelGGL(compact_move_kernel, dim3(g1(s.size)), dim3(kLcBlock), 0, st, s, invalid, free_rank, free_list,
                     free_spot_count, surfel_count);
  return hipGetLastError();
}

// ---- spatial order of the surfel buffer ---------------------------------------------------------------------------------
// The sweeps gather image patches of every keyframe that sees a wavefront's 64 surfels.  In creation order, the surfels
// that one image region shows are scattered over the buffer (each keyframe appended the part of the surface it saw
// first), so the same image lines are pulled into L2 many times per sweep.  Sorting the buffer along a Morton curve
// over a fixed world grid puts the surfels of a region - whoever created them - next to each other: measured at the
// bench size 249 -> 304 BA iterations/s.  Surfel identity is an index, so this is a maintenance operation like
// compaction (B/kernel_compact_surfels.cu), not part of an iteration; it moves the same rows compaction moves.
__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {   // 21 bits -> every third bit
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
// Morton key of the grid cell floor(p * inv_cell) + 2^20 per axis, clamped to 21 bits; deleted surfels (NaN x) sort last.
__device__ __forceinline__ unsigned long long surfel_sort_key(Vec3 p, float inv_cell) {
  if (!(p.x == p.x)) return ~0ull;
  unsigned long long q[3];
  const float c[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float g = floorf(c[a] * inv_cell) + 1048576.f;
    g = fminf(fmaxf(g, 0.f), 2097151.f);
    q[a] = (unsigned long long)g;
  }
  return spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
}
__global__ void __launch_bounds__(kLcBlock)
sort_keys_kernel(SurfelsView s, float inv_cell, unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx) {
  const uint32_t i = blockIdx.x * kLcBlock + threadIdx.x;
  if (i >= s.size) return;
  keys[i] = surfel_sort_key(surfel_position(s, i), inv_cell);
  idx[i] = i;
}
// the 8 data rows (and the active flags) of surfel idx[i] to place i of a dense scratch [8][n] (+ [n] bytes)
__global__ void __launch_bounds__(kLcBlock)
gather_rows_kernel(SurfelsView s, const uint32_t* __restrict__ idx, uint32_t n, float* __restrict__ rows_out, uint8_t* __restrict__ active_out) {
  const uint32_t i = blockIdx.x * kLcBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t src = idx[i];
#pragma unroll
  for (int row = 0; row < kSurfelAccum0; ++row) rows_out[(size_t)row * n + i] = s.row(row)[src];
  if (s.active) active_out[i] = s.active[src];
}
__global__ void __launch_bounds__(kLcBlock)
scatter_rows_back_kernel(SurfelsView s, uint32_t n, const float* __restrict__ rows_in, const uint8_t* __restrict__ active_in) {
  const uint32_t i = blockIdx.x * kLcBlock + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int row = 0; row < kSurfelAccum0; ++row) s.row(row)[i] = rows_in[(size_t)row * n + i];
  if (s.active) s.active[i] = active_in[i];
}

// Stable sort (radix sort of (key, index) pairs), then the 8 data rows and the active flags are gathered into a dense copy and
// written back: four kernels and the library's sort passes, all on the stream, no allocation and no host wait (round 6: the end
// tasks AND the loop's compaction reorder now; the first version allocated and freed six buffers and synchronised per call).
// scratch: sort_scratch_bytes(n) bytes, owned by the caller.
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
size_t sort_scratch_bytes(uint32_t n) {
  size_t temp_bytes = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (int)n, 0, 63, (hipStream_t)0);
  return 2 * align256(sizeof(unsigned long long) * (size_t)n) + 2 * align256(sizeof(uint32_t) * (size_t)n) + align256(sizeof(float) * (size_t)n * kSurfelAccum0) +
         align256((size_t)n) + align256(temp_bytes);
}
hipError_t sort_surfels_spatially(hipStream_t st, const SurfelsView& s, float inv_cell, void* scratch, size_t scratch_bytes) {
  const uint32_t n = s.size;
  if (n < 2) return hipSuccess;
  if (!scratch || scratch_bytes < sort_scratch_bytes(n)) return hipErrorInvalidValue;
  char* p = static_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* q = p; p += align256(bytes); return q; };
  unsigned long long* keys_in = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * (size_t)n));
  unsigned long long* keys_out = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * (size_t)n));
  uint32_t* idx_in = reinterpret_cast<uint32_t*>(take(sizeof(uint32_t) * (size_t)n));
  uint32_t* idx_out = reinterpret_cast<uint32_t*>(take(sizeof(uint32_t) * (size_t)n));
  float* rows = reinterpret_cast<float*>(take(sizeof(float) * (size_t)n * kSurfelAccum0));
  uint8_t* active = reinterpret_cast<uint8_t*>(take((size_t)n));
  void* temp = p;
  size_t temp_bytes = scratch_bytes - (size_t)(p - static_cast<char*>(scratch));
  hipLaunchKernelGGL(sort_keys_kernel, dim3(g1(n)), dim3(kLcBlock), 0, st, s, inv_cell, keys_in, idx_in);
  const hipError_t sorted = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, idx_in, idx_out, (int)n, 0, 63, st);
  if (sorted != hipSuccess) return sorted;   // nothing has been moved yet
  hipLaunchKernelGGL(gather_rows_kernel, dim3(g1(n)), dim3(kLcBlock), 0, st, s, idx_out, n, rows, active);
  hipLaunchKernelGGL(scatter_rows_back_kernel, dim3(g1(n)), dim3(kLcBlock), 0, st, s, n, rows, active);
  return hipGetLastError();
}

// ---- surfel shards <-> the whole cloud (multi-GPU surfel sharding) --------------------------------------------------------------
// Rank r of `world` owns every world-th chunk of `chunk` consecutive surfels of the cloud: local surfel l is global surfel
// ((l / chunk) * world + rank) * chunk + l % chunk.  shard_to_cloud_kernel copies a shard's data rows (0 .. 7) and active
// flags to their global places in a zeroed full-size buffer -- the ranks then sum their buffers as 64-bit integers, which
// leaves every bit pattern (a deleted surfel's NaN marker included) as it is --, cloud_to_shard_kernel takes the shard back out.
__device__ __forceinline__ uint32_t shard_global_index(uint32_t local, uint32_t rank, uint32_t world, uint32_t chunk) {
  return ((local / chunk) * world + rank) * chunk + local % chunk;
}
__global__ void __launch_bounds__(kLcBlock)
shard_to_cloud_kernel(SurfelsView shard, SurfelsView cloud, uint32_t rank, uint32_t world, uint32_t chunk) {
  const uint32_t l = blockIdx.x * kLcBlock + threadIdx.x;
  if (l >= shard.size) return;
  const uint32_t g = shard_global_index(l, rank, world, chunk);
#pragma unroll
  for (int row = 0; row < kSurfelAccum0; ++row) cloud.row(row)[g] = shard.row(row)[l];
  if (shard.active && cloud.active) cloud.active[g] = shard.active[l];
}
__global__ void __launch_bounds__(kLcBlock)
cloud_to_shard_kernel(SurfelsView cloud, SurfelsView shard, uint32_t rank, uint32_t world, uint32_t chunk) {
  const uint32_t l = blockIdx.x * kLcBlock + threadIdx.x;
  if (l >= shard.size) return;
  const uint32_t g = shard_global_index(l, rank, world, chunk);
#pragma unroll
  for (int row = 0; row < kSurfelAccum0; ++row) shard.row(row)[l] = cloud.row(row)[g];
  if (shard.active && cloud.active) shard.active[l] = cloud.active[g];
}
void launch_shard_to_cloud(hipStream_t st, const SurfelsView& shard, const SurfelsView& cloud, uint32_t rank, uint32_t world, uint32_t chunk) {
  if (shard.size) hipLaunchKernelGGL(shard_to_cloud_kernel, dim3((shard.size + kLcBlock - 1) / kLcBlock), dim3(kLcBlock), 0, st, shard, cloud, rank, world, chunk);
}
void launch_cloud_to_shard(hipStream_t st, const SurfelsView& cloud, const SurfelsView& shard, uint32_t rank, uint32_t world, uint32_t chunk) {
  if (shard.size) hipLaunchKernelGGL(cloud_to_shard_kernel, dim3((shard.size + kLcBlock - 1) / kLcBlock), dim3(kLcBlock), 0, st, cloud, shard, rank, world, chunk);
}

}  // namespace bahip
