// wave_reduce.h -- wave64 cross-lane reductions for gfx950 without LDS traffic.
//
// gfx950 has v_permlane32_swap / v_permlane16_swap (exchange half-waves / odd-even 16-lane rows of
// two registers) and DPP modifiers for every smaller stride, so a reduction never needs
// ds_bpermute.  Two forms:
//
//   wave_sum(v)            every lane receives the sum over the 64 lanes; same pairing (xor 32, 16,
//                          8, 4, 2, 1) and therefore the same rounding as the classic xor butterfly.
//   wave_reduce28(acc, …)  28 per-lane values -> 28 wave totals, spread over 28 lanes, with a
//                          "halving" butterfly: at stride s the two lanes of a pair split the value
//                          set between them, so the number of live values halves at every level
//                          (14+7+4+2+1+1 = 29 cross-lane adds instead of 28*6 = 168).  This is the
//                          reduction of the 21+6 pose normal-equation coefficients (the reference
//                          runs 27 serial CUB block reductions for them, B/gauss_newton.cuh:46-93).
#pragma once

#include <hip/hip_runtime.h>

namespace bahip {

namespace dpp {
constexpr int kQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kRowHalfMirror = 0x141; // lane i <-> 7-i inside each group of 8
constexpr int kRowRor8 = 0x128;       // lane i <-> i^8 inside each row of 16

template <int kCtrl>
__device__ __forceinline__ float mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), kCtrl, 0xf, 0xf, false));
}
}  // namespace dpp

// a <- sum of a over the lane pair (i, i^32) in lanes 0..31, sum of b over the pair in lanes 32..63.
__device__ __forceinline__ float halve32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// rows 0,2 (lane bit 4 clear): a summed over (i, i^16); rows 1,3: b summed over (i, i^16).
__device__ __forceinline__ float halve16(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// Generic DPP level: lanes with sel == false keep a, the others keep b; the partner's copy of the
// kept value is added.
template <int kCtrl>
__device__ __forceinline__ float halve_dpp(float a, float b, bool sel) {
  const float keep = sel ? b : a;
  const float send = sel ? a : b;
  return keep + dpp::mov<kCtrl>(send);
}

__device__ __forceinline__ float wave_sum(float v) {
  v = halve32(v, v);
  v = halve16(v, v);
  v += dpp::mov<dpp::kRowRor8>(v);
  v += __uint_as_float(__builtin_amdgcn_ds_swizzle(__float_as_uint(v), 0x101F));   // xor 4 (bit mode: and 0x1f, xor 4)
  v += dpp::mov<dpp::kQuadXor2>(v);
  v += dpp::mov<dpp::kQuadXor1>(v);
  return v;
}

// Index (0..27) of the total a lane holds after wave_reduce28, or -1 if the lane holds a duplicate.
__device__ __forceinline__ int wave_reduce28_slot(int lane) {
  const int b0 = lane & 1, b1 = (lane >> 1) & 1, b2 = (lane >> 2) & 1, b3 = (lane >> 3) & 1, b4 = (lane >> 4) & 1, b5 = (lane >> 5) & 1;
  const int i4 = b1 + 2 * b2;                       // index among the 4 values left after the stride-8 level
  if (b0 || (i4 == 3 && b3)) return -1;             // odd lanes mirror their even neighbour; value 3 of 7 had no partner
  const int i3 = (i4 < 3) ? i4 + 4 * b3 : 3;        // index among the 7 values left after the stride-16 level
  const int i2 = i3 + 7 * b4;                       // index among the 14 values left after the stride-32 level
  return i2 + 14 * b5;
}

// acc[0..27] per lane -> returns the wave total of acc[wave_reduce28_slot(lane)] (garbage where slot < 0).
__device__ __forceinline__ float wave_reduce28(const float (&acc)[28], int lane) {
  float r1[14], r2[7], r3[4], r4[2];
#pragma unroll
  for (int p = 0; p < 14; ++p) r1[p] = halve32(acc[p], acc[p + 14]);
#pragma unroll
  for (int p = 0; p < 7; ++p) r2[p] = halve16(r1[p], r1[p + 7]);
  const bool s3 = (lane & 8) != 0, s2 = (lane & 4) != 0, s1 = (lane & 2) != 0;
#pragma unroll
  for (int p = 0; p < 3; ++p) r3[p] = halve_dpp<dpp::kRowRor8>(r2[p], r2[p + 4], s3);
  r3[3] = r2[3] + dpp::mov<dpp::kRowRor8>(r2[3]);
#pragma unroll
  for (int p = 0; p < 2; ++p) r4[p] = halve_dpp<dpp::kRowHalfMirror>(r3[p], r3[p + 2], s2);
  const float r5 = halve_dpp<dpp::kQuadXor2>(r4[0], r4[1], s1);
  return r5 + dpp::mov<dpp::kQuadXor1>(r5);
}

// kCount = 8 or 16 per-lane values -> their wave totals with the same halving butterfly: lane (64 / kCount) * j receives
// the total of v[j] (every lane of that group of 64 / kCount lanes does).  10 / 17 cross-lane adds instead of 6 per value.
template <int kCount>
__device__ __forceinline__ float wave_reduce_small(const float (&v)[kCount], int lane) {
  static_assert(kCount == 8 || kCount == 16, "8 or 16 values");
  float r1[kCount / 2], r2[kCount / 4], r3[kCount / 8];
#pragma unroll
  for (int p = 0; p < kCount / 2; ++p) r1[p] = halve32(v[p], v[p + kCount / 2]);
#pragma unroll
  for (int p = 0; p < kCount / 4; ++p) r2[p] = halve16(r1[p], r1[p + kCount / 4]);
#pragma unroll
  for (int p = 0; p < kCount / 8; ++p) r3[p] = halve_dpp<dpp::kRowRor8>(r2[p], r2[p + kCount / 8], (lane & 8) != 0);
  float r4;
  if (kCount == 16) r4 = halve_dpp<dpp::kRowHalfMirror>(r3[0], r3[kCount / 8 - 1], (lane & 4) != 0);
  else r4 = r3[0] + dpp::mov<dpp::kRowHalfMirror>(r3[0]);
  r4 += dpp::mov<dpp::kQuadXor2>(r4);
  return r4 + dpp::mov<dpp::kQuadXor1>(r4);
}

}  // namespace bahip
