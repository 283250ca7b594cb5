// exact_sum.h -- exact (order-free) sums of binary32 values on the device.
//
// The reference merges the dense entries of its PCG vectors and its dot products with binary32 atomics in arbitrary order
// (B/kernel_pcg.cu:98-154), so its conjugate gradient differs from run to run.  Here such a sum is DEFINED as the exact sum of
// its binary32 terms, rounded once to binary64 (nearest, ties to even).  A finite binary32 is an integer multiple of 2^-149
// below 2^128 -- a 277-bit integer -- and an accumulator is 9 signed 64-bit limbs, limb j weighing 2^(32 j - 149).  A term
// m * 2^(p - 149), m < 2^24, is added as (m << (p & 31)) split into its low 32 bits (limb p >> 5) and the rest (next limb):
// every addend is below 2^32, so a limb takes 2^31 of them.  Limbs are integer sums: 64-bit integer atomics in any order,
// replicas that are folded later, and an integer all-reduce over GPUs all give the same limbs, hence the same bits.
// The oracle restates this in oracle_exact.c; both are held against Python's math.fsum (tests/test_*_exact_sum.py).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bahip {

constexpr int kExactLimbs = 9;
struct ExactCell { long long limb[kExactLimbs]; };   // 72 bytes
static_assert(sizeof(ExactCell) == 72, "ExactCell is exchanged as 9 int64");

struct ExactSplit {
  long long lo, hi;   // addends for limb and limb + 1
  int limb;           // -1: the value is zero, -2: not finite
};
__device__ __forceinline__ ExactSplit exact_split(float v) {
  const uint32_t bits = __float_as_uint(v);
  uint32_t e = (bits >> 23) & 0xffu;
  uint32_t m = bits & 0x7fffffu;
  ExactSplit s;
  s.lo = 0; s.hi = 0;
  if (e == 255u) { s.limb = -2; return s; }
  if (e) m |= 0x800000u; else e = 1u;   // denormals share the exponent of the smallest normal
  if (m == 0u) { s.limb = -1; return s; }
  const uint32_t p = e - 1u;            // the significand's LSB weighs 2^(p - 149)
  const unsigned long long w = (unsigned long long)m << (p & 31u);
  s.lo = (long long)(w & 0xffffffffull);
  s.hi = (long long)(w >> 32);
  if (bits >> 31) { s.lo = -s.lo; s.hi = -s.hi; }
  s.limb = (int)(p >> 5);
  return s;
}

__device__ __forceinline__ void limb_atomic_add(long long* limb, long long v) {
  __hip_atomic_fetch_add(limb, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // result unused: non-returning atomic
}

// cell += v, exactly.  A non-finite v raises the sticky flag instead (the sums then resolve to NaN, as a binary32 sum would).
__device__ __forceinline__ void exact_atomic_add(ExactCell* cell, float v, unsigned* invalid) {
  const ExactSplit s = exact_split(v);
  if (s.limb >= 0) {
    limb_atomic_add(&cell->limb[s.limb], s.lo);
    if (s.hi) limb_atomic_add(&cell->limb[s.limb + 1], s.hi);
  } else if (s.limb == -2) {
    atomicOr(invalid, 1u);
  }
}

// One part of cell += v from each of two neighbouring lanes that hold the same v (part 0: the low addend, part 1: the rest):
// one atomic instruction per value pair instead of two, issued as an asm statement the compiler's waitcnt pass does not track
// -- the sweeps issue these one candidate keyframe late, behind the next candidate's gathers (kernels_pose.hip says why: vmcnt
// retires in order, and an atomic in front of the gathers would put a second memory round trip into every candidate).
// Validated on gfx950 (the mnemonic is the gfx9 family's; later families call it global_atomic_add_u64).
__device__ __forceinline__ void exact_atomic_add_part_untracked(ExactCell* cell, float v, int part, unsigned* invalid) {
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__) && defined(__HIP_DEVICE_COMPILE__)
#error "exact_atomic_add_part_untracked: inline asm validated for the gfx9 family only"
#endif
  const ExactSplit s = exact_split(v);
  if (s.limb >= 0) {
    const long long addend = part ? s.hi : s.lo;
    long long* target = &cell->limb[s.limb + part];
    if (addend != 0) asm volatile("global_atomic_add_x2 %0, %1, off" ::"v"(target), "v"(addend) : "memory");
  } else if (s.limb == -2 && part == 0) {
    atomicOr(invalid, 1u);
  }
}

// The same into workgroup memory.  `limbs` is laid out [kExactLimbs][stride] with one column per thread, so the threads of a
// wavefront touch consecutive 8-byte words: a private accumulator per thread whose limb index is data dependent (registers
// cannot be indexed that way).
__device__ __forceinline__ void exact_lds_add(long long* limbs, int stride, int column, float v, unsigned* invalid) {
  const ExactSplit s = exact_split(v);
  if (s.limb >= 0) {
    limbs[s.limb * stride + column] += s.lo;
    if (s.hi) limbs[(s.limb + 1) * stride + column] += s.hi;
  } else if (s.limb == -2) {
    atomicOr(invalid, 1u);
  }
}

// The exact value of 9 limbs (already summed over replicas / ranks), rounded to binary64 (nearest, ties to even).
__device__ inline double exact_value(const long long (&limb)[kExactLimbs]) {
  // carry-normalise into 32-bit words; `carry` ends as the words above limb 8, sign included
  uint32_t w[kExactLimbs + 2];
  long long carry = 0;
#pragma unroll
  for (int j = 0; j < kExactLimbs; ++j) {
    const long long t = limb[j] + carry;
    w[j] = (uint32_t)((unsigned long long)t & 0xffffffffull);
    carry = t >> 32;   // arithmetic shift: floor
  }
  const bool negative = carry < 0;
  unsigned long long top = (unsigned long long)carry;
  if (negative) {   // two's complement -> magnitude
    unsigned long long c = 1;
#pragma unroll
    for (int j = 0; j < kExactLimbs; ++j) {
      const unsigned long long t = (unsigned long long)(uint32_t)~w[j] + c;
      w[j] = (uint32_t)t;
      c = t >> 32;
    }
    top = ~top + c;
  }
  w[kExactLimbs] = (uint32_t)top;
  w[kExactLimbs + 1] = (uint32_t)(top >> 32);
  // the leading word h, the two words below it, and whether anything non-zero lies further down -- without dynamic indexing
  uint32_t a = 0, b = 0, c3 = 0;   // w[h], w[h-1], w[h-2]
  int h = -1;
  bool sticky = false;
#pragma unroll
  for (int j = 0; j < kExactLimbs + 2; ++j)
    if (w[j] != 0u) h = j;
  if (h < 0) return 0.0;
#pragma unroll
  for (int j = 0; j < kExactLimbs + 2; ++j) {
    if (j == h) a = w[j];
    if (j == h - 1) b = w[j];
    if (j == h - 2) c3 = w[j];
    if (j < h - 2 && w[j] != 0u) sticky = true;
  }
  const int lz = __builtin_clz(a);
  const unsigned long long hi64 = ((unsigned long long)a << 32) | b;
  unsigned long long mant = hi64 << lz;
  if (lz) mant |= (unsigned long long)(c3 >> (32 - lz));
  sticky = sticky || ((uint32_t)(c3 << lz) != 0u);
  // mant has bit 63 set; its LSB weighs 2^(32 (h - 1) - 149 - lz)
  unsigned long long keep = mant >> 11;
  const uint32_t rem = (uint32_t)(mant & 0x7ffull);
  if (rem > 0x400u || (rem == 0x400u && (sticky || (keep & 1ull)))) ++keep;
  const int exponent = 32 * (h - 1) - 149 - lz + 11;      // -191 .. 214: a normal binary64 power of two
  const double scale = __longlong_as_double((long long)(1023 + exponent) << 52);
  const double value = (double)(long long)keep * scale;   // keep <= 2^53: the conversion and the product are exact
  return negative ? -value : value;
}

}  // namespace bahip
