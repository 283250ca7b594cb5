// kernels_intrinsics.hip -- intrinsics + depth-deformation step of the alternating scheme.
//
// Reference (B/ = applications/badslam/src/badslam/): one launch of
// AccumulateIntrinsicsCoefficientsCUDAKernel per keyframe over all surfels, each doing 20 + 28
// serial CUB block reductions and per-pair atomics (B/kernel_opt_intrinsics.cu:47-263), then the
// Schur complement kernel (:266-350), a 5x5 LDLT on the host (B/kernel_opt_intrinsics.cc:171) and
// the per-cell back-substitution (:375-423).
//
// Here: ONE surfel-centric launch with wave64 frustum culling sweeps all keyframes.  The global
// blocks A(15), b1(5), colour H(10), b(4) do not depend on the keyframe, so every lane keeps its
// 34 partial sums in registers across the whole sweep and the wave reduces them once at the end;
// only the per-cell terms B, D, b2, observation count go out as atomics per associated pair.
//
// Per-cell accumulators: ONE record of 8 values per sparse cell, {B0..B4, D, b2, observation count}, instead of eight arrays.
// A memory-side atomic costs a request per cache line it touches, whatever the number of lanes in it: eight scattered
// atomics per associated pair (52 M pairs x 8 at the bench size) took 12.9 ms.  The wavefront therefore transposes its
// 64 x 8 contributions through LDS and issues 8 instructions in which 8 consecutive lanes carry the 8 values of one
// surfel - one request per surfel instead of eight requests to eight lines.
//
// That still is one memory-side atomic request per associated pair, and gfx950 completes ~23.6 G of them per second whatever
// their width, scope or target (scripts/experiments/atomic_rates.hip: 52 M records of 8 values take 2.2 ms, shared or per-XCD
// tables, f64 / f32 / u64 alike) -- and while the queue to the atomic units is full the CU's other vector memory
// instructions, the next candidate's gathers included, wait behind it: sweep (1.45 ms at the bench size) and atomics (2.6 ms)
// add up instead of overlapping.  So the records are BINNED instead: the sparse cell grid is cut into blocks of 32 x 32 cells,
// each block has an append buffer (eight planes: the cell within the block and seven values; the observation count is 1),
// a wavefront reserves room for its records with ONE returning atomic per block it touches (one or two, seldom more) and
// stores them with plain coalesced stores; a second kernel adds the records of a block slice by slice into a table in LDS
// (1024 cells x 8 doubles = 64 KB, ds_add_f64) and adds the table to the global accumulators -- one request per touched cell
// and slice instead of one per pair.  (That second kernel is bound by the LDS atomics: ds_add_f64 retires ~0.75 lanes per clock
// and CU here, 0.9 ms for the 52 M x 8 additions of the bench size; ds_add_f32 is slower still.)  A record that finds its block's buffer full (the host sizes the buffers from the
// previous call's counts, so: the first call on a larger scene), or that belongs to a fourth block of one wavefront, takes
// the direct path above; the sums below do not depend on the path.
//
// DEFINITION of the sums (the oracle restates it, oracle_intrinsics.c).  The terms are binary32, exactly the reference's
// expressions; they are ADDED IN BINARY64: the 34 global sums as per-surfel binary32 chains over the keyframes in ascending
// order, then the xor butterfly over the 64 surfels of a tile (wave_sum), then binary64 over the tiles; the per-cell sums in
// binary64 pair by pair.  A binary64 sum of binary32 terms depends on the order of the additions only in bits that the
// final rounding to binary32 discards (a tie aside), so the atomics' order, the launch shape and a multi-GPU exchange
// (BAHIP_SUM_F64) do not show in the result, and kernels and oracle agree to the last bit (the reference adds binary32
// atomics in arbitrary order, B/kernel_opt_intrinsics.cu:217-262).
#include "ba_device.h"
#include "ba_launch.h"
#include "wave_cull.h"

namespace bahip {

constexpr int kIntrBlock = 256;    // per-cell kernels (Schur complement, back-substitution)
constexpr int kIntrSweepBlock = 64; // the surfel sweep: one wavefront per workgroup, like the other sweeps
constexpr int kARows = 5;
constexpr int kCellFloats = 8;     // B0..B4, D, b2, observation count
constexpr int kBinShift = 5;       // blocks of 32 x 32 sparse cells
constexpr int kBinCells = 1 << (2 * kBinShift);
constexpr int kBinGroups = 3;      // blocks one wavefront can append to per candidate keyframe (the rest: direct atomics)
constexpr uint32_t kBinSlice = 32768;   // records one workgroup of the reduction adds into its LDS table
constexpr int kBinReduceBlock = 512;
#ifndef BAHIP_INTR_REDUCE_FORM_DEFAULT
#define BAHIP_INTR_REDUCE_FORM_DEFAULT 1   // measured: 0.57 ms against 0.85 ms for the 52 M records of the bench scene
#endif
#ifndef BAHIP_INTR_BIN_SUBS
#define BAHIP_INTR_BIN_SUBS 16
#endif
// Every block has kBinSubs append buffers, a wavefront uses the one of its tile number: atomics on ONE address are served one
// after the other, and with one cursor per block (80 at 640 x 480) the reservations alone took longer than the whole sweep.
constexpr int kBinSubs = BAHIP_INTR_BIN_SUBS;
__device__ __forceinline__ float& cell_B(float* cells, int cell, int c) { return cells[(size_t)cell * kCellFloats + c]; }
__device__ __forceinline__ float& cell_D(float* cells, int cell) { return cells[(size_t)cell * kCellFloats + 5]; }
__device__ __forceinline__ float& cell_b2(float* cells, int cell) { return cells[(size_t)cell * kCellFloats + 6]; }
__device__ __forceinline__ float& cell_obs(float* cells, int cell) { return cells[(size_t)cell * kCellFloats + 7]; }

// Layout of the accumulation scratch: [0..14] A, [15..19] b1, [20..29] colour H, [30..33] colour b.
//
// Shape of the sweep (as the PCG sweeps, kernels_pcg.hip): one 64-surfel tile per wavefront, tiles dealt to the XCDs in runs
// (xcd_chunked_tile); per candidate keyframe all gathers of the pair -- the geometry word and the cfactor of the pixel, three
// luminance words -- are issued before the first is waited for, and the per-cell atomics of a candidate go out one candidate
// LATE, behind the next candidate's gathers: vmcnt retires in order, so an atomic issued before a gather would make the wait
// for that gather a wait for the atomic's round trip.
// Round 5: kIntrLdsSums of the 34 per-lane sums live in LDS instead of registers -- slot q of lane l at lds_sums[q * 64 + l], touched by
// that lane only (a plain read / add / write: the same binary32 additions in the same order, no atomics, no bank conflicts) -- which takes
// the sweep from 151 VGPRs and 3 wavefronts per SIMD to 4.  24 of them is what 16 one-wavefront workgroups per CU leave room for next to
// the transpose buffer (16 x (24 x 256 + 2304) bytes = 132 KB of the 160 KB).  0 restores the round-4 form.
// Measured (call 26): stage 1.98-2.00 -> 1.92-1.94 ms per iteration, bit-identical.  What the sweep's time is (calls 27 / 28: counters and
// timing builds without the record stores / the reservation atomics): 6.1e8 VALU wave-instructions per launch against the pose sweep's
// 4.2e8 in the same state -- depth and colour Jacobians, exp_det once a != 0, two IEEE divisions -- i.e. 1.3 x the pose sweep's work;
// the eight record stores per candidate cost 9 % of the launch, the returning reservation atomic 4 %, both together 13 %.
#ifndef BAHIP_INTR_LDS_SUMS
#define BAHIP_INTR_LDS_SUMS 24
#endif
constexpr int kIntrLdsSums = BAHIP_INTR_LDS_SUMS;
constexpr int kIntrRegSums = 34 - kIntrLdsSums;
#ifndef BAHIP_INTR_WAVES_PER_EU
#define BAHIP_INTR_WAVES_PER_EU (BAHIP_INTR_LDS_SUMS >= 24 ? 4 : 3)   // all 34 sums in registers: 151 VGPRs (at 4 waves: 57 spills)
#endif
}  // namespace bahip

// ---- the sweep: compiled once per arithmetic flavour (ba_launch.h) ----------------------------------------------------------------
BAHIP_FLAVOURED_BEGIN
template <bool kDepth, bool kColor>
__global__ void __launch_bounds__(kIntrSweepBlock) __attribute__((amdgpu_waves_per_eu(BAHIP_INTR_WAVES_PER_EU)))
intrinsics_accumulate_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s,
                             double* __restrict__ glob /* 34 */, double* __restrict__ cells /* S records of kCellFloats */, IntrBins bins,
                             const uint32_t* __restrict__ sched, uint32_t padded_tiles, uint32_t position_begin /* the launch covers the
                             positions [position_begin, position_begin + gridDim.x) of the schedule: a slice of the sweep */) {
  __shared__ float xpose[64 * (kCellFloats + 1)];   // lane-major: 8 values + the cell index, stride 9 (conflict-free both ways)
  uint32_t tile;   // heavy work first (wave_cull.h: scheduled_tile)
  if (!scheduled_tile(blockIdx.x + position_begin, padded_tiles, sched, &tile)) return;
  const uint32_t i = tile * kIntrSweepBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const float radius_sq = s.row(kSurfelRadiusSquared)[ii];
  const float d1 = s.row(kSurfelDescriptor1)[ii], d2 = s.row(kSurfelDescriptor2)[ii];
  const TangentPoints tp = surfel_tangent_points(gp, gn, radius_sq);   // per surfel, not per pair
  const WaveBounds wb = wave_bounds(gp, in_range && (gp.x == gp.x));
  if (wb.r < 0.f) return;   // a tile of the grid's padding, or one without a valid surfel (wave-uniform)
  __shared__ float lds_sums[(kIntrLdsSums > 0 ? kIntrLdsSums : 1) * 64];
  float reg_sums[kIntrRegSums > 0 ? kIntrRegSums : 1];
#pragma unroll
  for (int q = 0; q < kIntrRegSums; ++q) reg_sums[q] = 0.f;
#pragma unroll
  for (int q = 0; q < kIntrLdsSums; ++q) lds_sums[q * 64 + lane] = 0.f;
  // sum q (a compile-time index after unrolling): the first kIntrRegSums in registers, the others in this lane's LDS slots
  auto sum_add = [&](int q, float term) {
    if (q < kIntrRegSums) reg_sums[q] += term;
    else lds_sums[(q - kIntrRegSums) * 64 + lane] += term;
  };
  auto sum_value = [&](int q) { return q < kIntrRegSums ? reg_sums[q] : lds_sums[(q - kIntrRegSums) * 64 + lane]; };

  // the per-cell terms of the previous candidate: {B0..B4, D, b2} (the observation count is 1) and the cell as
  // (block << 10 | cell within the block), -1: none; and where they go: pending_slot = (group << 8 | rank within the group)
  // for the lanes whose block got a reservation -- group g's returning atomic was issued by lane reserve_lane[g] into
  // reserve_base[g], one candidate ago, so it has arrived with this candidate's gathers -- and -1 for the direct path.
  float pending[kCellFloats - 1];
  int pending_cell = -1, pending_slot = -1;
  uint32_t reserve_base[kBinGroups] = {0, 0, 0};
  int reserve_lane[kBinGroups] = {0, 0, 0};   // wave-uniform
  const bool binned = bins.capacity != 0;     // wave-uniform
  const uint32_t sub = tile & (kBinSubs - 1);
  auto flush_pending = [&]() {
    if (!kDepth) return;
    unsigned long long contributing = __ballot(pending_cell >= 0);
    if (!contributing) return;
    if (binned) {
      uint32_t slot = 0xffffffffu;
#pragma unroll
      for (int g = 0; g < kBinGroups; ++g) {
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)reserve_base[g], reserve_lane[g]);
        if (pending_slot >= 0 && (pending_slot >> 8) == g) slot = base + (uint32_t)(pending_slot & 0xff);
      }
      const bool stored = pending_cell >= 0 && slot < bins.capacity;
      if (stored) {
        uint32_t* rec = bins.records + ((size_t)(pending_cell >> (2 * kBinShift)) * kBinSubs + sub) * kCellFloats * bins.capacity + slot;
        rec[0] = (uint32_t)(pending_cell & (kBinCells - 1));
#pragma unroll
        for (int c = 0; c < kCellFloats - 1; ++c) rec[(size_t)(c + 1) * bins.capacity] = __float_as_uint(pending[c]);
        pending_cell = -1;
      }
      contributing = __ballot(pending_cell >= 0);
    }
    if (contributing) {
      // direct path: transpose through LDS -- afterwards lanes 8 s .. 8 s + 7 of pass j carry the 8 values of the surfel in
      // lane 8 j + s -- and one atomic instruction per 8 surfels
      int cell = -1;
      if (pending_cell >= 0) {
        const int block = pending_cell >> (2 * kBinShift), within = pending_cell & (kBinCells - 1);
        const int bx = block % bins.bins_x, by = block / bins.bins_x;
        cell = ((by << kBinShift) + (within >> kBinShift)) * in.cf_width + (bx << kBinShift) + (within & ((1 << kBinShift) - 1));
      }
      float* mine = xpose + lane * (kCellFloats + 1);
#pragma unroll
      for (int c = 0; c < kCellFloats - 1; ++c) mine[c] = pending[c];
      mine[kCellFloats - 1] = 1.0f;
      mine[kCellFloats] = __int_as_float(cell);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (!((contributing >> (8 * j)) & 0xffull)) continue;   // wave-uniform
        const float* src = xpose + (8 * j + (lane >> 3)) * (kCellFloats + 1);
        const int c = __float_as_int(src[kCellFloats]);
        if (c >= 0) unsafeAtomicAdd(&cells[(size_t)c * kCellFloats + (lane & 7)], (double)src[lane & 7]);
      }
      __builtin_amdgcn_wave_barrier();   // the next flush overwrites the buffer
    }
    pending_cell = -1;
  };
  // Room for the records of this candidate: one returning atomic per block the wavefront touches, issued now and read at the
  // next flush.
  auto reserve_pending = [&]() {
    if (!kDepth || !binned) return;
    pending_slot = -1;
    unsigned long long todo = __ballot(pending_cell >= 0);
#pragma unroll
    for (int g = 0; g < kBinGroups; ++g) {
      if (!todo) break;   // wave-uniform
      const int first = __ffsll((long long)todo) - 1;
      const int block = __builtin_amdgcn_readlane(pending_cell >> (2 * kBinShift), first);
      const bool member = pending_cell >= 0 && (pending_cell >> (2 * kBinShift)) == block;
      const unsigned long long m = __ballot(member);
      if (member) pending_slot = (g << 8) | (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      reserve_lane[g] = first;
      if (lane == first) reserve_base[g] = atomicAdd(&bins.cursors[block * kBinSubs + sub], (uint32_t)__popcll(m));
      todo &= ~m;
    }
  };

  for_each_candidate(
      num_kfs, [&](int k) { return sphere_may_project_item(in, kfs[k].pose.F, wb); },
      [&](int k) {
        const KfEntry& kf = kfs[k];
        const float* F = kf.pose.F;
        const Projected p = project_surfel(in, F, gp);
        const PixelWords pix = load_pixel_words(in, kf.geom, p);
        DescWords dw;
        if (kColor) dw = load_descriptor_words(in, kf.lumafp, F, tp, p);
        Assoc r;
        const bool associated = in_range && associate_from_words<false>(in, F, gn, p, pix, &r, nullptr);
        if (kColor) gathers_arrived(pix, dw);
        else gathers_arrived(pix);
        flush_pending();
        if (!__any(associated)) return;
        if (associated) {
          const float nx = r.nx, ny = r.ny;   // the association computed them (ba_device.h: Assoc)
          if (kDepth) {
            // B/kernel_opt_intrinsics.cu:81-120.  cfactor of the pixel's cell and the raw depth: the words the association
            // already loaded (the geometry plane's low half is the keyframe's depth image)
            const int sparse_px = r.px / in.cell, sparse_py = r.py / in.cell;
            const float cfactor = pix.cfactor;
            const float raw_inv_depth = 1.0f / (in.raw_to_float_depth * (uint16_t)(pix.geom & 0xffffu));
            const float exp_inv_depth = exp_det(-in.a * raw_inv_depth);
            const float corrected = cfactor * exp_inv_depth + raw_inv_depth;
            if (fabsf(corrected) > 1e-4f) {
              const float dot = dot3(mk3(nx, ny, 1), r.nl);
              const float inv_std = assoc_inv_std(in, r);
              float J[kARows + 1];
              jac_depth_intrinsics(r.px, r.py, r.depth, inv_std, dot3(gn, mk3(F[0], F[1], F[2])), dot3(gn, mk3(F[4], F[5], F[6])), dot, cfactor,
                                   raw_inv_depth, exp_inv_depth, corrected, J);
              const Vec3 u = mk3(r.depth * nx, r.depth * ny, r.depth);
              const float raw = inv_std * dot3(r.nl, u - r.local);
              const float w = depth_residual_weight(raw);
              int q = 0;
#pragma unroll
              for (int row = 0; row < kARows; ++row)
#pragma unroll
                for (int col = row; col < kARows; ++col) sum_add(q++, w * J[row] * J[col]);
              const float wr = w * raw;
#pragma unroll
              for (int c = 0; c < kARows; ++c) sum_add(15 + c, wr * J[c]);
              pending_cell = (((sparse_py >> kBinShift) * bins.bins_x + (sparse_px >> kBinShift)) << (2 * kBinShift)) |
                             ((sparse_py & ((1 << kBinShift) - 1)) << kBinShift) | (sparse_px & ((1 << kBinShift) - 1));
#pragma unroll
              for (int c = 0; c < kARows; ++c) pending[c] = w * J[c] * J[kARows];
              pending[5] = w * J[kARows] * J[kARows];
              pending[6] = w * raw * J[kARows];
            }
          }
          if (kColor && dw.color_ok) {
            DescEval e;
            eval_descriptor_from_words(in, kf.lumafp, dw, d1, d2, &e);
            // B/kernel_opt_intrinsics.cu:142-150,200-215: validity flag is "residual != 0"
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const float gx = t ? e.gx2 : e.gx1, gy = t ? e.gy2 : e.gy1, raw = t ? e.r2 : e.r1;
              if (raw != 0) {
                float J[4];
                jac_descriptor_color_intrinsics(gx, gy, nx, ny, J);
                const float w = descriptor_residual_weight(raw);
                int q = 20;
#pragma unroll
                for (int row = 0; row < 4; ++row)
#pragma unroll
                  for (int col = row; col < 4; ++col) sum_add(q++, w * J[row] * J[col]);
                const float wr = w * raw;
#pragma unroll
                for (int c = 0; c < 4; ++c) sum_add(30 + c, wr * J[c]);
              }
            }
          }
        }
        reserve_pending();
      });
  flush_pending();

  float mine = 0.f;
#pragma unroll
  for (int q = 0; q < 34; ++q) {
    const float v = wave_sum(sum_value(q));
    if (lane == q) mine = v;
  }
  if (lane < 34 && mine != 0.f) unsafeAtomicAdd(&glob[lane], (double)mine);
}

// positions [position_begin, position_begin + position_count) of the sweep's schedule (position_count == 0: all of them)
void launch_intrinsics_accumulate(hipStream_t st, bool depth, bool color, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                                  const SurfelsView& s, double* glob, double* cells, const IntrBins& bins, const uint32_t* sched,
                                  uint32_t position_begin, uint32_t position_count) {
  if (!s.size) return;
  const uint32_t padded = xcd_padded_tiles((s.size + kIntrSweepBlock - 1) / kIntrSweepBlock), positions = sched_positions(padded, sched);
  if (position_begin >= positions) return;
  const uint32_t count = position_count ? std::min(position_count, positions - position_begin) : positions - position_begin;
  const dim3 grid(count), block(kIntrSweepBlock);
  if (depth && color) hipLaunchKernelGGL((intrinsics_accumulate_kernel<true, true>), grid, block, 0, st, in, kfs, num_kfs, s, glob, cells, bins, sched, padded, position_begin);
  else if (depth) hipLaunchKernelGGL((intrinsics_accumulate_kernel<true, false>), grid, block, 0, st, in, kfs, num_kfs, s, glob, cells, bins, sched, padded, position_begin);
  else hipLaunchKernelGGL((intrinsics_accumulate_kernel<false, true>), grid, block, 0, st, in, kfs, num_kfs, s, glob, cells, bins, sched, padded, position_begin);
}
BAHIP_FLAVOURED_END

// ---- what exists once (the exact unit): record reduction, Schur complement, back-substitution, the dispatcher ---------------------
#ifndef BAHIP_FAST_MATH
namespace bahip {
void launch_intrinsics_accumulate(hipStream_t st, bool depth, bool color, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                                  const SurfelsView& s, double* glob, double* cells, const IntrBins& bins, const uint32_t* sched,
                                  uint32_t position_begin, uint32_t position_count) {
  BAHIP_PICK(in, launch_intrinsics_accumulate(st, depth, color, in, kfs, num_kfs, s, glob, cells, bins, sched, position_begin, position_count));
}

// The records of one slice of one block's append buffer, added into a table in LDS and from there into the global per-cell
// accumulators (see the head of this file).
__global__ void __launch_bounds__(kBinReduceBlock)
intrinsics_bin_reduce_kernel(IntrBins bins, int slices_per_bin, int cf_width, int cf_height, double* __restrict__ cells) {
  // kCellFloats planes of kBinCells doubles: the 64 lanes of one ds_add_f64 carry the same value index c of 64 records, and with
  // the records side by side (cell * 8 + c) they would meet in 4 of the 64 banks
  extern __shared__ double table[];
  const int buffer = blockIdx.x / slices_per_bin, slice = blockIdx.x % slices_per_bin, block = buffer / kBinSubs;
  const uint32_t count = min(bins.cursors[buffer], bins.capacity);   // the cursor keeps counting past the capacity (the host reads it)
  const uint32_t begin = (uint32_t)slice * kBinSlice;
  if (begin >= count) return;
  const uint32_t end = min(count, begin + kBinSlice);
  // (the observation counts, plane 7, are integers: 32-bit LDS atomics are several times faster than ds_add_f64)
  uint32_t* counts = reinterpret_cast<uint32_t*>(table + (kCellFloats - 1) * kBinCells);
  for (int e = threadIdx.x; e < kBinCells * kCellFloats; e += kBinReduceBlock) table[e] = 0.0;
  __syncthreads();
  const uint32_t* rec = bins.records + (size_t)buffer * kCellFloats * bins.capacity;
  // kBinUnroll records per thread and round, all their loads in flight before the first addition
  constexpr int kBinUnroll = 4;
  for (uint32_t r0 = begin + threadIdx.x; r0 < end; r0 += kBinUnroll * kBinReduceBlock) {
    uint32_t within[kBinUnroll];
    float v[kBinUnroll][kCellFloats - 1];
#pragma unroll
    for (int u = 0; u < kBinUnroll; ++u) {
      const uint32_t r = min(r0 + u * kBinReduceBlock, end - 1);   // clamped: loaded, not added
      within[u] = rec[r];
#pragma unroll
      for (int c = 0; c < kCellFloats - 1; ++c) v[u][c] = __uint_as_float(rec[(size_t)(c + 1) * bins.capacity + r]);
    }
#pragma unroll
    for (int u = 0; u < kBinUnroll; ++u) {
      if (r0 + u * kBinReduceBlock >= end) break;
#pragma unroll
      for (int c = 0; c < kCellFloats - 1; ++c) atomicAdd(table + c * kBinCells + within[u], (double)v[u][c]);
      atomicAdd(counts + within[u], 1u);
    }
  }
  __syncthreads();
  const int bx = block % bins.bins_x, by = block / bins.bins_x;
  for (int e = threadIdx.x; e < kBinCells * kCellFloats; e += kBinReduceBlock) {
    const int within = e / kCellFloats, c = e % kCellFloats;   // 8 consecutive lanes: the record of one cell, one request
    const double v = c < kCellFloats - 1 ? table[c * kBinCells + within] : (double)counts[within];
    if (v == 0.0) continue;
    const int cx = (bx << kBinShift) + (within & ((1 << kBinShift) - 1)), cy = (by << kBinShift) + (within >> kBinShift);
    unsafeAtomicAdd(&cells[((size_t)cy * cf_width + cx) * kCellFloats + c], v);
  }
}

// The same reduction without a floating-point atomic in LDS (round 4).  ds_add_f64 is what bounds the kernel above; 32-bit integer
// LDS atomics and plain LDS traffic are several times faster.  So a chunk of kSortChunk records is SORTED by cell in LDS -- a
// histogram by returning 32-bit atomics (the returned value is the record's rank within its cell), a scan of the 1024 counts, and
// the seven values of every record stored at start[cell] + rank -- and then every thread adds the records of ITS two cells, which
// now lie side by side, into binary64 sums it keeps in registers across the chunks of the slice.  At the end the sums go through
// the table layout of the kernel above into the global accumulators (one request per touched cell).  The order in which a cell's
// records are added is the order the atomics arrived in, as arbitrary as before; the definition at the head of this file makes
// the result independent of it.
#ifndef BAHIP_SORT_CHUNK
#define BAHIP_SORT_CHUNK 4096
#endif
#ifndef BAHIP_SORT_BLOCK
#define BAHIP_SORT_BLOCK 512
#endif
constexpr int kSortChunk = BAHIP_SORT_CHUNK;
constexpr int kSortBlock = BAHIP_SORT_BLOCK;
constexpr int kSortCells = kBinCells / kSortBlock;  // cells per thread
constexpr int kSortUnroll = kSortChunk / kSortBlock;
constexpr size_t kSortStartBytes = 4352;            // 1025 offsets, padded to 256 B
constexpr size_t kSortValueBytes = (size_t)(kCellFloats - 1) * kSortChunk * sizeof(float);
constexpr size_t kSortTableBytes = (size_t)kBinCells * kCellFloats * sizeof(double);
// (the flush table reuses the sort buffers)
constexpr size_t kSortLdsBytes = kSortStartBytes + kSortValueBytes > kSortTableBytes ? kSortStartBytes + kSortValueBytes : kSortTableBytes;
static_assert(kBinCells % kSortBlock == 0 && kSortChunk % kSortBlock == 0 && kSortBlock % 64 == 0 && kSortBlock <= 1024, "shape of the sorted reduction");
__global__ void __launch_bounds__(kSortBlock)
intrinsics_bin_reduce_sorted_kernel(IntrBins bins, int slices_per_bin, int cf_width, int cf_height, double* __restrict__ cells) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
  uint32_t* start = reinterpret_cast<uint32_t*>(sort_lds);                     // counts, then exclusive offsets; [kBinCells] = total
  float* vals = reinterpret_cast<float*>(sort_lds + kSortStartBytes);          // [kCellFloats - 1][kSortChunk], sorted by cell
  __shared__ uint32_t wave_total[kSortBlock / 64];
  const int buffer = blockIdx.x / slices_per_bin, slice = blockIdx.x % slices_per_bin, block = buffer / kBinSubs;
  const uint32_t count = min(bins.cursors[buffer], bins.capacity);
  const uint32_t begin = (uint32_t)slice * kBinSlice;
  if (begin >= count) return;
  const uint32_t end = min(count, begin + kBinSlice);
  const uint32_t* rec = bins.records + (size_t)buffer * kCellFloats * bins.capacity;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double acc[kSortCells][kCellFloats - 1];
  uint32_t observations[kSortCells];
#pragma unroll
  for (int h = 0; h < kSortCells; ++h) {
    observations[h] = 0u;
#pragma unroll
    for (int c = 0; c < kCellFloats - 1; ++c) acc[h][c] = 0.0;
  }

  for (uint32_t chunk = begin; chunk < end; chunk += kSortChunk) {
    const uint32_t chunk_n = min((uint32_t)kSortChunk, end - chunk);
    for (int e = tid; e <= kBinCells; e += kSortBlock) start[e] = 0u;
    uint32_t within[kSortUnroll], rank[kSortUnroll];
    float v[kSortUnroll][kCellFloats - 1];
#pragma unroll
    for (int u = 0; u < kSortUnroll; ++u) {
      const uint32_t r = min(chunk + (uint32_t)(tid + u * kSortBlock), end - 1);   // clamped: loaded, not used
      within[u] = rec[r];
#pragma unroll
      for (int c = 0; c < kCellFloats - 1; ++c) v[u][c] = __uint_as_float(rec[(size_t)(c + 1) * bins.capacity + r]);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kSortUnroll; ++u) {
      rank[u] = 0u;
      if ((uint32_t)(tid + u * kSortBlock) < chunk_n) rank[u] = atomicAdd(start + within[u], 1u);
    }
    __syncthreads();
    // exclusive scan of the 1024 counts: kSortCells per thread, an inclusive scan per wavefront, the wavefront totals through LDS
    uint32_t mine[kSortCells], sum_mine = 0u;
#pragma unroll
    for (int h = 0; h < kSortCells; ++h) { mine[h] = start[kSortCells * tid + h]; sum_mine += mine[h]; }
    uint32_t inclusive = sum_mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t other = __shfl_up(inclusive, d);
      if (lane >= d) inclusive += other;
    }
    if (lane == 63) wave_total[wave] = inclusive;
    __syncthreads();
    uint32_t base = 0u;
#pragma unroll
    for (int w = 0; w < kSortBlock / 64; ++w) base += w < wave ? wave_total[w] : 0u;
    uint32_t running = base + inclusive - sum_mine;
#pragma unroll
    for (int h = 0; h < kSortCells; ++h) { start[kSortCells * tid + h] = running; running += mine[h]; }
    if (tid == kSortBlock - 1) start[kBinCells] = running;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kSortUnroll; ++u) {
      if ((uint32_t)(tid + u * kSortBlock) < chunk_n) {
        const uint32_t pos = start[within[u]] + rank[u];
#pragma unroll
        for (int c = 0; c < kCellFloats - 1; ++c) vals[c * kSortChunk + pos] = v[u][c];
      }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < kSortCells; ++h) {
      const int cell = tid + h * kSortBlock;   // neighbouring lanes, neighbouring cells: their records lie side by side
      const uint32_t j0 = start[cell], j1 = start[cell + 1];
      for (uint32_t j = j0; j < j1; ++j) {
#pragma unroll
        for (int c = 0; c < kCellFloats - 1; ++c) acc[h][c] += (double)vals[c * kSortChunk + j];
      }
      observations[h] += j1 - j0;
    }
    __syncthreads();   // the next chunk overwrites the offsets and the sorted values
  }

  // the sums of this slice through the table of the kernel above: planes of kBinCells doubles, the counts as integers in plane 7
  double* table = reinterpret_cast<double*>(sort_lds);
  uint32_t* counts = reinterpret_cast<uint32_t*>(table + (kCellFloats - 1) * kBinCells);
#pragma unroll
  for (int h = 0; h < kSortCells; ++h) {
    const int cell = tid + h * kSortBlock;
#pragma unroll
    for (int c = 0; c < kCellFloats - 1; ++c) table[c * kBinCells + cell] = acc[h][c];
    counts[cell] = observations[h];
  }
  __syncthreads();
  const int bx = block % bins.bins_x, by = block / bins.bins_x;
  for (int e = tid; e < kBinCells * kCellFloats; e += kSortBlock) {
    const int within_block = e / kCellFloats, c = e % kCellFloats;   // 8 consecutive lanes: the record of one cell, one request
    const double value = c < kCellFloats - 1 ? table[c * kBinCells + within_block] : (double)counts[within_block];
    if (value == 0.0) continue;
    const int cx = (bx << kBinShift) + (within_block & ((1 << kBinShift) - 1)), cy = (by << kBinShift) + (within_block >> kBinShift);
    unsafeAtomicAdd(&cells[((size_t)cy * cf_width + cx) * kCellFloats + c], value);
  }
}

// Schur complement: B/kernel_opt_intrinsics.cu:266-350.  One thread per sparse cell.  This runs AFTER the multi-GPU
// all-reduce of the accumulators, on every rank.  The binary64 accumulators are rounded to binary32 here (cells_f: what the
// reference holds at this point); the 20 sums over the cells are formed without atomics - the xor butterfly per wavefront
// of 64 cells (wave_sum), one partial per wavefront, then intrinsics_finish_kernel adds the partials in wavefront order
// (binary32, starting from 0) and adds that total to the rounded global sum: defined, and restated by the oracle.
__global__ void __launch_bounds__(kIntrBlock)
intrinsics_schur_kernel(int S, float* __restrict__ partials /* [wavefronts][20] */, const double* __restrict__ cells_d,
                        float* __restrict__ cells) {
  const int cell = blockIdx.x * kIntrBlock + threadIdx.x;
  float part[20];
#pragma unroll
  for (int q = 0; q < 20; ++q) part[q] = 0.f;
  if (cell < S) {
#pragma unroll
    for (int c = 0; c < kCellFloats; ++c) cells[(size_t)cell * kCellFloats + c] = (float)cells_d[(size_t)cell * kCellFloats + c];
    const float D_inverse = 1.0f / cell_D(cells, cell);
    if (!(D_inverse < 1e12f)) {
      cell_D(cells, cell) = __builtin_nanf("");
    } else {
      const float D_inv_b2 = D_inverse * cell_b2(cells, cell);
      cell_D(cells, cell) = D_inv_b2;
      float Bc[kARows];
#pragma unroll
      for (int c = 0; c < kARows; ++c) Bc[c] = cell_B(cells, cell, c);
      int q = 0;
#pragma unroll
      for (int row = 0; row < kARows; ++row)
#pragma unroll
        for (int col = row; col < kARows; ++col) part[q++] = -1.f * (Bc[row] * D_inverse * Bc[col]);
#pragma unroll
      for (int c = 0; c < kARows; ++c) part[15 + c] = -1.f * (Bc[c] * D_inv_b2);
#pragma unroll
      for (int c = 0; c < kARows; ++c) cell_B(cells, cell, c) = D_inverse * Bc[c];
    }
  }
  const int lane = threadIdx.x & 63;
  float mine = 0.f;
#pragma unroll
  for (int q = 0; q < 20; ++q) {
    const float v = wave_sum(part[q]);
    if (lane == q) mine = v;
  }
  const int wave = (blockIdx.x * kIntrBlock + threadIdx.x) >> 6;
  if (lane < 20) partials[(size_t)wave * 20 + lane] = mine;
}
// glob_f[q] = (float)glob_d[q] for the 34 global sums [+ the Schur total of sum q < 20 when num_waves > 0].  The Schur total is a
// chain over the wavefronts' partials in order; the workgroup stages them through LDS in pieces (a chain of dependent global
// loads took 0.2 ms at 640 x 480), threads 0 .. 19 add.
constexpr int kFinishBlock = 256, kFinishPiece = 128;   // wavefront partials per piece
__global__ void __launch_bounds__(kFinishBlock)
intrinsics_finish_kernel(int num_waves, const float* __restrict__ partials, const double* __restrict__ glob_d, float* __restrict__ glob_f) {
  __shared__ float piece[kFinishPiece * 20];
  const int q = threadIdx.x;
  float total = 0.f;
  for (int first = 0; first < num_waves; first += kFinishPiece) {
    const int n = min(kFinishPiece, num_waves - first) * 20;
    for (int e = threadIdx.x; e < n; e += kFinishBlock) piece[e] = partials[(size_t)first * 20 + e];
    __syncthreads();
    if (q < 20) {
#pragma unroll 8
      for (int w = 0; w < n / 20; ++w) total += piece[w * 20 + q];
    }
    __syncthreads();
  }
  if (q >= 34) return;
  float value = (float)glob_d[q];
  if (q < 20 && num_waves > 0) value += total;
  glob_f[q] = value;
}

// Back-substitution: B/kernel_opt_intrinsics.cu:375-423
__global__ void __launch_bounds__(kIntrBlock)
intrinsics_solve_cells_kernel(Intrinsics in, int S, float* __restrict__ cells, const float* __restrict__ x1 /* 5 */, float* cfactor,
                              uint32_t cfactor_pitch) {
  const int cell = blockIdx.x * kIntrBlock + threadIdx.x;
  if (cell >= S) return;
  float offset = cell_D(cells, cell);
  if (offset != offset) {
    offset = 0;
  } else {
#pragma unroll
    for (int c = 0; c < kARows; ++c) offset -= cell_B(cells, cell, c) * x1[c];
  }
  const int y = cell / in.cf_width, x = cell - y * in.cf_width;
  float* p = pitched_ptr(cfactor, cfactor_pitch, y, x);
  float value = *p - offset;
  if (cell_obs(cells, cell) == 0.f) value = 0;
  *p = value;
}

int intrinsics_bin_count(const Intrinsics& in, int* bins_x_out) {
  const int bins_x = (in.cf_width + (1 << kBinShift) - 1) >> kBinShift, bins_y = (in.cf_height + (1 << kBinShift) - 1) >> kBinShift;
  if (bins_x_out) *bins_x_out = bins_x;
  return bins_x * bins_y * kBinSubs;   // append buffers
}
size_t intrinsics_bin_record_bytes() { return kCellFloats * sizeof(uint32_t); }
uint32_t intrinsics_sweep_positions(uint32_t surfels, const uint32_t* sched) {
  return sched_positions(xcd_padded_tiles((surfels + kIntrSweepBlock - 1) / kIntrSweepBlock), sched);
}
// The second kernel of the step: the binned per-cell records into the per-cell accumulators (nothing to do without bins).
// 0: the table of ds_add_f64 (round 3), 1: records sorted by cell in LDS, sums in registers (round 4)
static int g_intr_reduce_form = bahip_env_int("BAHIP_INTR_REDUCE_FORM", BAHIP_INTR_REDUCE_FORM_DEFAULT);
void set_intrinsics_reduce_form(int form) { g_intr_reduce_form = form < 0 ? BAHIP_INTR_REDUCE_FORM_DEFAULT : form; }
void launch_intrinsics_bin_reduce(hipStream_t st, bool depth, const Intrinsics& in, const SurfelsView& s, double* cells, const IntrBins& bins) {
  if (!s.size) return;
  if (depth && bins.capacity) {
    const int slices = (int)((bins.capacity + kBinSlice - 1) / kBinSlice);
    if (g_intr_reduce_form == 1) {
      // more than 64 KB of LDS per workgroup: an opt-in per device, as the persistent pose sweep's
      static bool raised[64] = {}, failed[64] = {};
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
      if (!raised[dev] && !failed[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&intrinsics_bin_reduce_sorted_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kSortLdsBytes) == hipSuccess) raised[dev] = true;
        else { failed[dev] = true; (void)hipGetLastError(); }
      }
      if (raised[dev]) {
        hipLaunchKernelGGL(intrinsics_bin_reduce_sorted_kernel, dim3((unsigned)(intrinsics_bin_count(in, nullptr) * slices)), dim3(kSortBlock),
                           kSortLdsBytes, st, bins, slices, in.cf_width, in.cf_height, cells);
        return;
      }
    }
    hipLaunchKernelGGL(intrinsics_bin_reduce_kernel, dim3((unsigned)(intrinsics_bin_count(in, nullptr) * slices)), dim3(kBinReduceBlock),
                       kBinCells * kCellFloats * sizeof(double), st, bins, slices, in.cf_width, in.cf_height, cells);
  }
}
size_t intrinsics_schur_partials(int S) { return 20 * (size_t)((S + kIntrBlock - 1) / kIntrBlock) * (kIntrBlock / 64); }
void launch_intrinsics_finish(hipStream_t st, bool schur, int S, const double* glob_d, const double* cells_d, float* glob_f, float* cells_f,
                              float* partials) {
  int waves = 0;
  if (schur) {
    const int blocks = (S + kIntrBlock - 1) / kIntrBlock;
    hipLaunchKernelGGL(intrinsics_schur_kernel, dim3(blocks), dim3(kIntrBlock), 0, st, S, partials, cells_d, cells_f);
    waves = blocks * (kIntrBlock / 64);
  }
  hipLaunchKernelGGL(intrinsics_finish_kernel, dim3(1), dim3(kFinishBlock), 0, st, waves, partials, glob_d, glob_f);
}
void launch_intrinsics_solve_cells(hipStream_t st, const Intrinsics& in, int S, float* cells, const float* x1, float* cfactor,
                                   uint32_t cfactor_pitch) {
  hipLaunchKernelGGL(intrinsics_solve_cells_kernel, dim3((S + kIntrBlock - 1) / kIntrBlock), dim3(kIntrBlock), 0, st, in, S, cells, x1,
                     cfactor, cfactor_pitch);
}

}  // namespace bahip
#endif   // !BAHIP_FAST_MATH
