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
__device__ __forceinline__ float& cell_B(float* cells, int cell, int c) { return cells[(size_t)cell * kCellFloats + c]; }
__device__ __forceinline__ float& cell_D(float* cells, int cell) { return cells[(size_t)cell * kCellFloats + 5]; }
__device__ __forceinline__ float& cell_b2(float* cells, int cell) { return cells[(size_t)cell * kCellFloats + 6]; }
__device__ __forceinline__ float& cell_obs(float* cells, int cell) { return cells[(size_t)cell * kCellFloats + 7]; }

// Layout of the accumulation scratch (floats): [0..14] A, [15..19] b1, [20..29] colour H, [30..33] colour b.
template <bool kDepth, bool kColor>
__global__ void __launch_bounds__(kIntrSweepBlock)
intrinsics_accumulate_kernel(Intrinsics in, const KfEntry* __restrict__ kfs, int num_kfs, SurfelsView s,
                             double* __restrict__ glob /* 34 */, double* __restrict__ cells /* S records of kCellFloats */) {
  __shared__ float xpose[64 * (kCellFloats + 1)];   // lane-major: 8 values + the cell index, stride 9 (conflict-free both ways)
  const uint32_t i = blockIdx.x * kIntrSweepBlock + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in_range = i < s.size;
  const uint32_t ii = in_range ? i : 0;
  const Vec3 gp = surfel_position(s, ii);
  const Vec3 gn = surfel_normal(s, ii);
  const float radius_sq = s.row(kSurfelRadiusSquared)[ii];
  const float d1 = s.row(kSurfelDescriptor1)[ii], d2 = s.row(kSurfelDescriptor2)[ii];
  const WaveBounds wb = wave_bounds(gp, in_range && (gp.x == gp.x));
  float acc[34];
#pragma unroll
  for (int q = 0; q < 34; ++q) acc[q] = 0.f;

  for_each_candidate(
      num_kfs, [&](int k) { return sphere_may_project_item(in, kfs[k].pose.F, wb); },
      [&](int k) {
        const float* F = kfs[k].pose.F;
        Assoc r;
        const bool associated = in_range && project_associate<false>(in, F, kfs[k].geom, gp, gn, &r, nullptr);
        if (!__any(associated)) return;
        float cv[kCellFloats];
#pragma unroll
        for (int c = 0; c < kCellFloats; ++c) cv[c] = 0.f;
        int cell = -1;
        if (associated) {
          const float nx = unp_nx(in, (float)r.px), ny = unp_ny(in, (float)r.py);
          if (kDepth) {
            // B/kernel_opt_intrinsics.cu:81-120
            const int sparse_px = r.px / in.cell, sparse_py = r.py / in.cell;
            const float cfactor = pitched_load(in.cfactor, in.cfactor_pitch, sparse_py, sparse_px);
            const float raw_inv_depth = 1.0f / (in.raw_to_float_depth * pitched_load(kfs[k].depth, kfs[k].depth_pitch, r.py, r.px));
            const float exp_inv_depth = expf(-in.a * raw_inv_depth);
            const float corrected = cfactor * exp_inv_depth + raw_inv_depth;
            if (fabsf(corrected) > 1e-4f) {
              const float dot = dot3(mk3(nx, ny, 1), r.nl);
              const float inv_std = depth_inv_stddev(nx, ny, r.depth, r.nl, in.baseline_fx);
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
                for (int col = row; col < kARows; ++col) acc[q++] += w * J[row] * J[col];
              const float wr = w * raw;
#pragma unroll
              for (int c = 0; c < kARows; ++c) acc[15 + c] += wr * J[c];
              cell = sparse_px + sparse_py * in.cf_width;
#pragma unroll
              for (int c = 0; c < kARows; ++c) cv[c] = w * J[c] * J[kARows];
              cv[5] = w * J[kARows] * J[kARows];
              cv[6] = w * raw * J[kARows];
              cv[7] = 1.0f;
            }
          }
          if (kColor) {
            float cx, cy;
            if (depth_to_color_pixel(in, r.pxx, r.pxy, &cx, &cy)) {
              DescEval e;
              eval_descriptor<true>(in, kfs[k].lumafp, F, gp, gn, radius_sq, cx, cy, d1, d2, &e);
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
                    for (int col = row; col < 4; ++col) acc[q++] += w * J[row] * J[col];
                  const float wr = w * raw;
#pragma unroll
                  for (int c = 0; c < 4; ++c) acc[30 + c] += wr * J[c];
                }
              }
            }
          }
        }
        if (kDepth) {
          // transpose through LDS: afterwards lanes 8 s .. 8 s + 7 of pass j carry the 8 values of the surfel in lane 8 j + s
          const unsigned long long contributing = __ballot(cell >= 0);
          if (contributing) {
            float* mine = xpose + lane * (kCellFloats + 1);
#pragma unroll
            for (int c = 0; c < kCellFloats; ++c) mine[c] = cv[c];
            mine[kCellFloats] = __int_as_float(cell);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              if (!((contributing >> (8 * j)) & 0xffull)) continue;   // wave-uniform
              const float* src = xpose + (8 * j + (lane >> 3)) * (kCellFloats + 1);
              const int c = __float_as_int(src[kCellFloats]);
              if (c >= 0) unsafeAtomicAdd(&cells[(size_t)c * kCellFloats + (lane & 7)], (double)src[lane & 7]);
            }
            __builtin_amdgcn_wave_barrier();   // the next candidate overwrites the buffer
          }
        }
      });

  float mine = 0.f;
#pragma unroll
  for (int q = 0; q < 34; ++q) {
    const float v = wave_sum(acc[q]);
    if (lane == q) mine = v;
  }
  if (lane < 34 && mine != 0.f) unsafeAtomicAdd(&glob[lane], (double)mine);
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
// glob_f[q] = (float)glob_d[q] for the 34 global sums [+ the Schur total of sum q < 20 when num_waves > 0]
__global__ void __launch_bounds__(64)
intrinsics_finish_kernel(int num_waves, const float* __restrict__ partials, const double* __restrict__ glob_d, float* __restrict__ glob_f) {
  const int q = threadIdx.x;
  if (q >= 34) return;
  float value = (float)glob_d[q];
  if (q < 20 && num_waves > 0) {
    float total = 0.f;
    for (int w = 0; w < num_waves; ++w) total += partials[(size_t)w * 20 + q];
    value += total;
  }
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

void launch_intrinsics_accumulate(hipStream_t st, bool depth, bool color, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                                  const SurfelsView& s, double* glob, double* cells) {
  if (!s.size) return;
  const dim3 grid((s.size + kIntrSweepBlock - 1) / kIntrSweepBlock), block(kIntrSweepBlock);
  if (depth && color) hipLaunchKernelGGL((intrinsics_accumulate_kernel<true, true>), grid, block, 0, st, in, kfs, num_kfs, s, glob, cells);
  else if (depth) hipLaunchKernelGGL((intrinsics_accumulate_kernel<true, false>), grid, block, 0, st, in, kfs, num_kfs, s, glob, cells);
  else hipLaunchKernelGGL((intrinsics_accumulate_kernel<false, true>), grid, block, 0, st, in, kfs, num_kfs, s, glob, cells);
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
  hipLaunchKernelGGL(intrinsics_finish_kernel, dim3(1), dim3(64), 0, st, waves, partials, glob_d, glob_f);
}
void launch_intrinsics_solve_cells(hipStream_t st, const Intrinsics& in, int S, float* cells, const float* x1, float* cfactor,
                                   uint32_t cfactor_pitch) {
  hipLaunchKernelGGL(intrinsics_solve_cells_kernel, dim3((S + kIntrBlock - 1) / kIntrBlock), dim3(kIntrBlock), 0, st, in, S, cells, x1,
                     cfactor, cfactor_pitch);
}

}  // namespace bahip
