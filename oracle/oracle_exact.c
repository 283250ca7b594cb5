/* oracle_exact.c -- exact (order-free) sums of binary32 values.
 * Test infrastructure only (see oracle.h).
 *
 * DEFINITION shared with the backend (badslam_amd/csrc/exact_sum.h).  The reference merges the dense entries of the PCG
 * vectors and its dot products with binary32 atomics in arbitrary order (B/kernel_pcg.cu:98-154: block reduction, then
 * atomicAdd), so its PCG is not reproducible run to run (SURVEY appendix B marks this FIX).  Here such a sum is the EXACT sum
 * of its binary32 terms, rounded once to binary64 (round to nearest, ties to even): a value that does not depend on the order
 * of the additions, on the launch shape, or on how the terms are spread over GPUs.
 *
 * Representation: a finite binary32 is an integer multiple of 2^-149 below 2^128, i.e. a 277-bit integer; an accumulator is
 * 9 signed 64-bit limbs, limb j carrying weight 2^(32 j - 149).  A term m * 2^(p-149) (m < 2^24) is added as
 * (m << (p & 31)) split into its low 32 bits -> limb p >> 5 and the rest -> the next limb; every addend is below 2^32 in
 * magnitude, so a limb holds 2^31 of them before it could overflow.  Limbs are plain integer sums: adding them with atomics,
 * in any grouping, or with an integer all-reduce gives the same limbs.
 *
 * This file is pinned independently of the backend: tests/test_cpu_exact_sum.py compares orc_exact_sum with Python's
 * math.fsum (exactly rounded by construction) on wide-range, cancelling and denormal inputs. */
#include "oracle_internal.h"

void orc_exact_add(orc_exact* cell, float v, int* invalid) {
  uint32_t bits;
  memcpy(&bits, &v, sizeof(bits));
  uint32_t e = (bits >> 23) & 0xffu;
  uint32_t m = bits & 0x7fffffu;
  if (e == 255u) {   /* NaN / Inf: sticky flag, the sum resolves to NaN */
    __atomic_store_n(invalid, 1, __ATOMIC_RELAXED);
    return;
  }
  if (e) m |= 0x800000u; else e = 1;   /* denormals share the exponent of the smallest normal */
  if (m == 0) return;
  const uint32_t p = e - 1u;           /* LSB weight 2^(p - 149) */
  const int limb = (int)(p >> 5);
  const uint64_t w = (uint64_t)m << (p & 31u);
  long long lo = (long long)(w & 0xffffffffull), hi = (long long)(w >> 32);
  if (bits >> 31) { lo = -lo; hi = -hi; }
  __atomic_fetch_add(&cell->limb[limb], lo, __ATOMIC_RELAXED);
  if (hi) __atomic_fetch_add(&cell->limb[limb + 1], hi, __ATOMIC_RELAXED);
}

void orc_exact_merge(orc_exact* dst, const orc_exact* src) {
  for (int j = 0; j < ORC_EXACT_LIMBS; ++j) __atomic_fetch_add(&dst->limb[j], src->limb[j], __ATOMIC_RELAXED);
}

/* The exact value of the accumulator, rounded to binary64 (nearest, ties to even). */
double orc_exact_value(const orc_exact* cell) {
  /* carry-normalise into 32-bit words; what is left in `carry` are the words above limb 8 (sign included) */
  uint32_t w[ORC_EXACT_LIMBS + 2];
  long long carry = 0;
  for (int j = 0; j < ORC_EXACT_LIMBS; ++j) {
    const long long t = cell->limb[j] + carry;
    w[j] = (uint32_t)((unsigned long long)t & 0xffffffffull);
    carry = t >> 32;   /* arithmetic: floor */
  }
  const int negative = carry < 0;
  unsigned long long top = (unsigned long long)carry;
  if (negative) {   /* two's complement -> magnitude */
    unsigned long long c = 1;
    for (int j = 0; j < ORC_EXACT_LIMBS; ++j) {
      const unsigned long long t = (unsigned long long)(uint32_t)~w[j] + c;
      w[j] = (uint32_t)t;
      c = t >> 32;
    }
    top = ~top + c;
  }
  w[ORC_EXACT_LIMBS] = (uint32_t)top;
  w[ORC_EXACT_LIMBS + 1] = (uint32_t)(top >> 32);
  int h = ORC_EXACT_LIMBS + 1;
  while (h >= 0 && w[h] == 0) --h;
  if (h < 0) return 0.0;
  const uint32_t w1 = h >= 1 ? w[h - 1] : 0u, w2 = h >= 2 ? w[h - 2] : 0u;
  int sticky = 0;
  for (int j = 0; j < h - 2; ++j) sticky |= (w[j] != 0);
  const int lz = __builtin_clz(w[h]);
  const uint64_t hi64 = ((uint64_t)w[h] << 32) | w1;
  uint64_t mant = hi64 << lz;
  if (lz) mant |= (uint64_t)(w2 >> (32 - lz));
  sticky |= ((uint32_t)(w2 << lz) != 0);   /* the bits of w2 that did not make it into mant */
  /* mant has bit 63 set; its LSB weighs 2^(32 (h - 1) - 149 - lz) */
  uint64_t keep = mant >> 11;
  const uint32_t rem = (uint32_t)(mant & 0x7ffu);
  if (rem > 0x400u || (rem == 0x400u && (sticky || (keep & 1u)))) ++keep;
  const int exponent = 32 * (h - 1) - 149 - lz + 11;
  const double value = ldexp((double)keep, exponent);   /* exact: keep <= 2^53, the result is a normal binary64 */
  return negative ? -value : value;
}

/* Test hook: the exactly rounded binary64 sum of n binary32 values (NaN if one of them is not finite). */
double orc_exact_sum(const float* values, size_t n) {
  orc_exact acc;
  int invalid = 0;
  memset(&acc, 0, sizeof(acc));
  for (size_t i = 0; i < n; ++i) orc_exact_add(&acc, values[i], &invalid);
  return invalid ? (double)NAN : orc_exact_value(&acc);
}

/* Test hooks for the multi-rank exchange (tests/test_cpu_multigpu_gloo.py): the limbs themselves.  A surfel-sharded run adds
 * its own terms into 9 int64 limbs, sums the limbs over the ranks with an integer all-reduce and resolves the total. */
void orc_exact_accumulate(const float* values, size_t n, long long limbs[ORC_EXACT_LIMBS], int* invalid) {
  orc_exact acc;
  memcpy(acc.limb, limbs, sizeof(acc.limb));
  for (size_t i = 0; i < n; ++i) orc_exact_add(&acc, values[i], invalid);
  memcpy(limbs, acc.limb, sizeof(acc.limb));
}
double orc_exact_resolve(const long long limbs[ORC_EXACT_LIMBS]) {
  orc_exact acc;
  memcpy(acc.limb, limbs, sizeof(acc.limb));
  return orc_exact_value(&acc);
}
