// ba_device.h -- device-side arithmetic of the direct BA backend (gfx950, wave64).
//
// Restates, for HIP, the residual / association arithmetic defined by the reference's device
// headers (B/ = applications/badslam/src/badslam/ in ETH3D/badslam):
//   B/cuda_matrix.cuh:37-141, B/surfel_projection.cuh:40-207, B/util.cuh:62-153,
//   B/util_nvcc_only.cuh:51-115, B/robust_weighting.cuh:39-86, B/cost_function.cuh:44-254,
//   B/surfel_projection_nvcc_only.cuh:48-127,302-511.
// gfx950 has no texture sampling path (tex2D is unavailable), so the bilinear, clamp-addressed,
// normalised-float colour fetch of B/keyframe.cc:67-73 is done in software on the luma byte.
//
// The sweeps do not read the keyframe images in the reference's pitch-linear layout but two derived
// "BA planes" per keyframe (kernels_preprocess.hip: pack_*_kernel), both made of 8x4-pixel tiles of
// 32-bit words so that one 128-byte line holds a compact pixel block:
//   geom   word(x, y) = raw depth u16 | packed normal u16 << 16            (one load per association test)
//   lumafp word(ix+1, iy+1) = the 2x2 clamp-addressed luma footprint whose top-left texel is (ix, iy),
//          ix in [-1, W], iy in [-1, H]: tl | tr << 8 | bl << 16 | br << 24 (one load per bilinear sample)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/badslam_hip.h"

namespace bahip {

constexpr uint16_t kInvalidDepthBit = 1 << 15;       // B/kernels.cuh:38
constexpr uint16_t kUnknownDepth = 65535;            // B/kernels.cuh:41
constexpr uint8_t kSurfelActiveFlag = 1;             // B/kernels.cuh:44
constexpr uint32_t kInvalidIndex = 4294967295u;      // B/kernels.cuh:54
constexpr float kCosNormalCompat = 0.76604f;         // B/kernels.cuh:58
constexpr uint32_t kDeletedSurfelBits = 0x7fffffffu; // CUDART_NAN_F, B/kernel_delete_surfels.cu:145

enum SurfelRow {  // B/kernels.cuh:69-88
  kSurfelX = 0, kSurfelY = 1, kSurfelZ = 2, kSurfelNormal = 3, kSurfelRadiusSquared = 4,
  kSurfelColor = 5, kSurfelDescriptor1 = 6, kSurfelDescriptor2 = 7, kSurfelAccum0 = 8
};

struct Vec3 { float x, y, z; };
__device__ __forceinline__ Vec3 mk3(float x, float y, float z) { Vec3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ Vec3 operator+(Vec3 a, Vec3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ Vec3 operator-(Vec3 a, Vec3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ Vec3 operator*(float m, Vec3 b) { return mk3(m * b.x, m * b.y, m * b.z); }
// Sums of products are written as explicit fused multiply-add chains, the same chains as the oracle's helpers
// (oracle_internal.h): -ffp-contract=off keeps the compiler from fusing anything else, so both sides round alike, and the
// kernels (VALU-issue bound) spend one instruction per term.  (The reference's nvcc build fuses at its own discretion.)
__device__ __forceinline__ float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot3(Vec3 a, Vec3 b) { return mad(a.z, b.z, mad(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float sqlen3(Vec3 a) { return mad(a.z, a.z, mad(a.y, a.y, a.x * a.x)); }
__device__ __forceinline__ float norm3(Vec3 a) { return sqrtf(sqlen3(a)); }
__device__ __forceinline__ Vec3 cross3(Vec3 a, Vec3 b) {
  return mk3(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y);
}

// ---- exact reciprocal and square root in few instructions ----------------------------------------------------------
// The sweeps are VALU-issue bound, and the compiler's IEEE sequences for 1.f / x (11 instructions: two v_div_scale, v_rcp,
// five fused multiply-adds, v_div_fmas, v_div_fixup) and sqrtf (16) spend most of them on operands that cannot occur here
// (denormal inputs / results, 2^-126 scalings).  On normal operands the hardware approximation (<= 1 ulp) followed by one
// fused Newton step IS the correctly rounded result (Markstein's theorem for the reciprocal; the sqrt keeps the compiler's
// own residual test over s - 1 ulp, s, s + 1 ulp).  tests/test_gpu_exact_math.py checks both against IEEE division / sqrtf
// for EVERY binary32 significand (2^23 values per binade) on the device, so parity with the oracle's `1.f / x` and
// `sqrtf(x)` stays bit for bit.  Domain: |x| and 1 / |x| normal (2^-126 < |x| < 2^126), or 0 / inf / NaN, which get the
// hardware's IEEE result.
// (Round 5 measured what an instruction of each class costs -- scripts/microbench/inst_cost.hip, profiles/r5_inst_cost.txt: a VOP2
// binary32 op on VGPRs 1.6 cycles of a SIMD, v_fma_f32 1.8, a conversion / compare / 24-bit multiply 2.6, three-operand integer
// ops, selects, v_med3 / v_min3, DPP forms and anything with an SGPR source 2.6 - 2.8, v_rcp / v_sqrt / v_permlane*_swap 5.1 -- and
// which respellings pay on the sweeps: see DESIGN.md section 5, "Round 5".)
__device__ __forceinline__ float rcp_exact(float x) {
#ifdef BAHIP_FAST_MATH
  return __builtin_amdgcn_rcpf(x);   // fast flavour: the hardware approximation (<= 1 ulp), what nvcc -use_fast_math gives the reference
#endif
  const float y0 = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, y0, 1.f);
  const float y1 = __builtin_fmaf(e, y0, y0);
  // x = 0, inf, NaN (y1 is NaN there): v_div_fixup_f32 substitutes the IEEE quotient 1 / x (inf, 0, NaN with the operand's sign) and
  // passes |y1| with the quotient's sign through otherwise -- one instruction instead of a compare and a select
  return __builtin_amdgcn_div_fixupf(y1, x, 1.f);
}
// x in [2^-96, 2^127) or 0 (the range in which the compiler's sequence does not rescale).
__device__ __forceinline__ float sqrt_exact(float x) {
#ifdef BAHIP_FAST_MATH
  return __builtin_amdgcn_sqrtf(x);   // fast flavour: v_sqrt_f32 (<= 1 ulp)
#endif
  const float s = __builtin_amdgcn_sqrtf(x);
  const float dn = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_dn = __builtin_fmaf(-dn, s, x);
  const float r_up = __builtin_fmaf(-up, s, x);
  float t = (0.f >= r_dn) ? dn : s;
  t = (0.f < r_up) ? up : t;
  return (x == 0.f) ? x : t;
}

// exp of a binary32 argument, defined by explicit binary32 operations (reduction by ln 2 in two pieces, Taylor polynomial of
// degree 7 on |r| <= ln 2 / 2 in fused multiply-adds, scaled by 2^k): the depth deformation exp(-a / depth) enters every
// residual once a != 0, and the device library's expf and glibc's differ in the last bit now and then -- with it defined (the
// oracle restates the same operations) runs that optimise the depth intrinsics stay comparable bit for bit, as sincos_det does
// for the pose updates; the weights of the preprocessing's bilateral filter use it too.  Within 2 ulp of the correctly rounded
// value (the reference's own expf is CUDA's, documented at 2 ulp); 14 instructions, no binary64 (a binary64 evaluation made the
// geometry sweep spill).
__device__ __forceinline__ float exp_det(float xf) {
#ifdef BAHIP_FAST_MATH
  return __builtin_amdgcn_exp2f(xf * 1.44269504f);   // fast flavour: v_exp_f32, as __expf
#endif
  if (!(xf == xf)) return xf;
  if (xf > 100.f) return __builtin_inff();
  if (xf < -110.f) return 0.f;
  const float k = __builtin_rintf(xf * 1.44269504f);                           /* nearest multiple of ln 2 */
  float r = __builtin_fmaf(-k, 0.693145752f, xf);                              /* ln 2 = 0.693145752 + 1.42860677e-06 */
  r = __builtin_fmaf(-k, 1.42860677e-06f, r);
  float p = 1.f / 5040.f;
  p = __builtin_fmaf(p, r, 1.f / 720.f);
  p = __builtin_fmaf(p, r, 1.f / 120.f);
  p = __builtin_fmaf(p, r, 1.f / 24.f);
  p = __builtin_fmaf(p, r, 1.f / 6.f);
  p = __builtin_fmaf(p, r, 0.5f);
  p = __builtin_fmaf(p, r, 1.f);
  p = __builtin_fmaf(p, r, 1.f);
  return __builtin_ldexpf(p, (int)k);
}

// Pose of one keyframe as the kernels consume it: frame_T_global (3x4 row-major) and
// global_R_frame (3x3 row-major), both cached whenever the pose is set (B/keyframe.h:160-172).
struct KfPose {
  float F[12];
  float GR[9];
};

// Device-side keyframe table entry.
struct KfEntry {
  const uint16_t* depth;
  const uint16_t* normals;
  const uint16_t* radius;
  const uint8_t* color;
  const uint32_t* geom;      // tiled depth | normal words (BA plane)
  const uint32_t* lumafp;    // tiled 2x2 luma footprints (BA plane)
  uint32_t depth_pitch, normals_pitch, radius_pitch, color_pitch;
  KfPose pose;
  float global_T_frame[7];   // Sophus layout: qx qy qz qw tx ty tz
  int32_t activation;
};

// One frame whose pose is being estimated (batched Gauss-Newton, kernels_pose.hip).
constexpr int kHbCoefficients = 28;   // 21 H (row-major upper triangle) + 6 b + 1 pad
// The pose normal equations are summed in FIXED POINT, order-free.  The 27 binary32 totals of one (64-surfel tile, keyframe)
// pair -- the result of the fixed halving tree of wave_reduce.h -- are converted to multiples of 2^-32 (round to nearest even;
// exact for every total of magnitude >= 2^-9) and added as two 64-bit integer limbs: the low 32 bits of that integer into limb
// 0 (weight 2^-32), the rest into limb 1 (weight 1).  Integer addition is associative, so the sums do not depend on the order
// of the atomics, on the launch shape (workgroup tables in LDS or global atomics, pose_parts), on how tiles are grouped, or on
// how the surfels are sharded over GPUs (an integer all-reduce is exact): H and b are deterministic, and the oracle computes
// the same bits (oracle_pose.c).  The reference merges float atomics in arbitrary order (B/gauss_newton.cuh:71,89; SURVEY
// appendix B marks this FIX).  Round 2 used one limb of weight 2^-16 (range 1.4e14, and a quantum above the binary32 ulp of a
// total below 128); with two limbs the quantum is 2.3e-10 and the range of a sum 9.2e18.  A tile total that is not finite or
// not below 2^52 = 4.5e15 in magnitude is NOT added: it raises a sticky flag in the counter record of the phase and the pose
// estimation fails with an error (the reference would carry a NaN into the solve).  Tile totals of 1e12 do occur (surfels seen
// at grazing angles from half a metre: 1 / sigma^2 ~ 1e8 per pair), sums of a keyframe reach 1e11 ... 1e14; a sum whose limb 1
// ends at 2^62 or beyond in magnitude -- it cannot jump over that band, every addend being below 2^52 -- is flagged as well by
// the solve kernel, so a wrapped sum would need more than 2^64 - 2^62 = 1.4e19 to go unnoticed.
typedef long long HbFixed;            // one limb
constexpr int kHbLimbs = 2;
constexpr int kHbStride = kHbCoefficients * kHbLimbs;   // int64 words per work item: [coefficient][limb]
struct HbSplit {
  long long lo, hi;
  bool valid;
};
__host__ __device__ __forceinline__ HbSplit hb_split(float v) {
  unsigned int bits;
  __builtin_memcpy(&bits, &v, sizeof(bits));
  unsigned int e = (bits >> 23) & 0xffu;
  unsigned int m = bits & 0x7fffffu;
  HbSplit r;
  r.lo = 0; r.hi = 0; r.valid = true;
  if (e) m |= 0x800000u; else e = 1u;
  // |v| = m * 2^(e - 150) = m * 2^s in units of 2^-32, s = e - 118; |v| < 2^52 <=> e - 127 < 52 <=> s <= 60
  const int s = (int)e - 118;
  if (s > 60) { r.valid = false; return r; }        // too large, infinite or NaN (e = 255)
  if (s >= 32) {
    r.hi = (long long)((unsigned long long)m << (s - 32));
  } else if (s >= 0) {
    const unsigned long long w = (unsigned long long)m << s;
    r.lo = (long long)(w & 0xffffffffull);
    r.hi = (long long)(w >> 32);
  } else if (s >= -25) {                            // below the quantum: round to nearest, ties to even
    const int sh = -s;
    unsigned int q = m >> sh;
    const unsigned int rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    r.lo = (long long)q;
  }
  if (bits >> 31) { r.lo = -r.lo; r.hi = -r.hi; }
  return r;
}
// The same limbs as sign and magnitudes, without a branch (device only): hb_split walks four exponent ranges, and the 27 totals of a
// tile lie in all of them, so a wavefront executes every branch -- ~35 vector instructions and eight execution-mask regions per
// candidate.  Here the split is float arithmetic, every step exact for |v| < 2^52:
//   hi  = trunc(v)                       (integer part: limb 1)
//   lo  = rint((v - hi) * 2^32)          (v - hi is exact; for |v| >= 2^-9 the product is an integer already, below that rint is
//                                         hb_split's round-to-nearest-even of m >> sh)
//   hi  = hh * 2^32 + hl, hh = trunc(hi * 2^-32), hl = hi - hh * 2^32   (both below 2^32: v_cvt_u32_f32 converts them)
// |lo| < 2^32 (below 2^23 in the rounding range), so the magnitudes are what hb_split returns and the sign is v's.  A sink adds the
// magnitudes with an add or a subtract instruction according to the sign (tests/test_gpu_kernels_vs_oracle.py compares value by value).
struct HbMagnitudes {
  uint32_t lo;          // |limb 0|
  uint32_t hi_lo, hi_hi;   // |limb 1| as two words
  bool negative, valid;
};
__device__ __forceinline__ HbMagnitudes hb_split_magnitudes(float v) {
  HbMagnitudes r;
  const float a = fabsf(v);
  r.valid = a < 4503599627370496.f;                       // 2^52; false for NaN
  r.negative = (__float_as_uint(v) >> 31) != 0u;
  const float hi = __builtin_truncf(a);
  const float lo = __builtin_rintf((a - hi) * 4294967296.f);
  const float hh = __builtin_truncf(hi * 2.3283064365386963e-10f);
  const float hl = __builtin_fmaf(-hh, 4294967296.f, hi);
  r.lo = (uint32_t)lo;
  r.hi_lo = (uint32_t)hl;
  r.hi_hi = (uint32_t)hh;
  return r;
}
// The value of a limb pair (carry-normalised first): binary64; H and b are its rounding to binary32.
__host__ __device__ __forceinline__ double hb_value(long long lo, long long hi) {
  hi += lo >> 32;                 // arithmetic shift: floor
  lo &= 0xffffffffll;
  return (double)hi + (double)lo * 2.3283064365386963e-10;   // 2^-32
}
struct PoseWork {
  float F[12];        // frame_T_global at the current linearisation point
  float T[7];         // global_T_frame estimate (Sophus layout)
  float T0[7];        // global_T_frame when the Gauss-Newton rounds began (decides afterwards whether the keyframe "moved")
  int32_t kf_index;   // entry of the frame table providing the images
  int32_t done;       // converged or iteration cap reached (or skipped)
  int32_t iterations;
  int32_t converged;
  int32_t moved;      // set when done (batched keyframe phase): log(T0^-1 * T) fails the convergence test
  int32_t skip;       // the accumulate sweep leaves the item out: done, or (keyframe sharding) the keyframe's images live on another rank
};
static_assert(sizeof(PoseWork) == 128, "PoseWork is read back as 32-word records");
// The two records after the last work item hold counters (as int32): [round] = work items still iterating after that
// Gauss-Newton round (round < BAHIP_MAX_POSE_ITERATIONS = 30), [kPoseCounterConverged] = keyframes that count as converged
// in the BA loop (inactive ones + those that did not move).  One device-to-host copy per round brings work items and counters.
constexpr int kPoseCounterConverged = 32;
// [kPoseCounterTicket]: workgroups of pose_solve_kernel that have finished (the last one publishes the counters to the host);
// [kPoseCounterSequence] (host copy only): the sequence number of the launch whose counters the host copy holds -- the host
// polls it instead of synchronising the stream and copying 256 bytes (capi_ba.hip: run_pose_rounds).
constexpr int kPoseCounterTicket = 33;
constexpr int kPoseCounterSequence = 34;
// [kPoseCounterInvalid]: sticky, raised by the accumulate kernel when a tile total could not be added (not finite, or 2^52 and
// beyond: hb_split) and by the solve kernel when a sum left the safe range; the host turns it into an error when the round's
// counters arrive.
constexpr long long kHbSumLimit = 1ll << 62;
constexpr int kPoseCounterInvalid = 35;
constexpr int kPoseTailRecords = 2;
// [kPoseCounterWorked]: workgroups of the current solve launch in which a work item took a Gauss-Newton step (cleared by the
// publishing workgroup; feeds the round count of the device-driven loop)
constexpr int kPoseCounterWorked = 36;
// Control words of the device-driven BA loop (capi_ba.hip: bahip_alternating_iterations).  kLoopStop: 0 = run, 1 = the loop has
// converged (B/direct_ba_alternating.cc:693-701: every keyframe counts as converged and iteration >= min_iterations - 1),
// 2 = a pose phase ran out of queued Gauss-Newton rounds with work items still iterating (the host continues it) -- every launch
// of the loop does nothing once it is non-zero.  The rest are totals since the host last cleared them.
constexpr int kLoopStop = 0, kLoopIterationsDone = 1, kLoopRounds = 2, kLoopSteps = 3, kLoopNotConverged = 4,
              kLoopInvalid = 5 /* a pose phase of the loop raised kPoseCounterInvalid */, kLoopWords = 8;
// Behind the counter records (device only): the indices of the work items still iterating after the latest Gauss-Newton
// round, in arbitrary order (pose_solve_kernel appends with the same atomic that counts them).  The later rounds of a phase
// sweep over this list instead of over all work items (a handful of entries instead of K).
__host__ __device__ constexpr size_t pose_work_records(size_t work_items) { return work_items + kPoseTailRecords + (work_items + 31) / 32; }

// Unknown-vector layout of the PCG scheme (B/direct_ba_pcg.cc:232-307): [6 per non-gauge keyframe |
// geom_stride per surfel | 5 + S depth intrinsics | 4 colour intrinsics].
struct PcgLayout {
  int optimize_poses, optimize_geometry, optimize_depth_intrinsics, optimize_color_intrinsics;
  int use_depth, use_desc;
  uint32_t surfel_start, depth_intr_start, a_index, color_intr_start, unknown_count;
  int geom_stride;   // 3 with descriptor residuals, else 1
  int gauge;         // keyframe whose pose is held fixed
  // The surfel block [head_lo, head_hi) of the unknown vector is local to a rank under surfel sharding; everything else (the
  // "dense head": poses and intrinsics) is replicated.  head_lo == head_hi == unknown_count when geometry is not optimised.
  uint32_t head_lo, head_hi;
  // Per-keyframe entry points (bahip_pcg_init / bahip_pcg_step1 with one keyframe, Route B): the sweep covers one keyframe
  // whose pose unknowns start at single_pose_index (optimize_poses already says whether they are unknowns at all), and the
  // surfel entries continue the chains the earlier calls left in the vectors instead of starting from zero.
  int single_keyframe;   // -1: the sweep covers the whole bound table
  uint32_t single_pose_index;
  int accumulate;
};

// Exact accumulators of one PCG solve (exact_sum.h; kernels_pcg.hip says what is summed where).  Replicated slots: sums that
// every tile / workgroup adds to -- 64 replicas each, folded when they are resolved.
struct ExactCell;
constexpr int kHotReplicas = 64;
enum : int {
  kHotA = 0,           // 9: global intrinsics entries of r (PCGInit) / g (PCGStep1)
  kHotB = 9,           // 9: ... of M
  kHotAlphaD = 18,     // pair part of alpha_d
  kHotEpsLocal = 19,   // epsilon terms of alpha_d over the local unknowns
  kHotExchanged1 = 20, // slots [0, 20) + the head cells are what a sharded run exchanges after a sweep
  kHotDotLocal = 20,   // alpha_n / beta_n over the local unknowns (exchanged after the dot-product kernel)
  kHotEpsHead = 21,    // the dense head's share (identical on every rank: not exchanged)
  kHotDotHead = 22,
  kHotSlots = 23
};
struct PcgExact {
  ExactCell* hot;        // slots [0, kHotExchanged1) x kHotReplicas
  ExactCell* head_a;     // one cell per dense-head unknown: r (PCGInit) / g (PCGStep1)
  ExactCell* head_b;     // ... M
  ExactCell* hot_tail;   // slots [kHotExchanged1, kHotSlots) x kHotReplicas
  unsigned* invalid;     // sticky: a non-finite term was added; every sum resolves to NaN
};

// Everything that is constant over a sweep; passed to kernels by value (kernarg -> SGPRs).
struct Intrinsics {
  // depth camera: corner-convention projector + centre-convention unprojector (B/surfel_projection.h:42-71)
  float fx, fy, cx, cy;
  float fx_inv, fy_inv, cx_inv, cy_inv;
  int width, height;
  // colour camera (PixelCornerProjector; PixelCenterProjector shares fx, fy)
  float cfx, cfy, ccx, ccy;
  int cwidth, cheight;
  // DepthToColorPixelCorner (B/surfel_projection.h:105-124)
  float d2c_fx, d2c_fy, d2c_cx, d2c_cy;
  // DepthParameters
  float a, raw_to_float_depth, baseline_fx;
  int cell;
  int cell_shift;   // log2(cell) if cell is a power of two, else -1 (host-side hint: shifts instead of integer division)
  uint32_t geom_skip, fp_skip;   // plane_strip_skip() of the geom / lumafp planes
  const float* cfactor;
  uint32_t cfactor_pitch;
  int cf_width, cf_height;
  // Interleaved partial sums per surfel in the normals / geometry passes (kernels_surfel.hip: tile_sums) -- part of the NUMERICAL
  // DEFINITION of those sums: 4 (default) or 8 (bahip_context_set_sum_classes; what keyframe sharding over 8 ranks needs)
  int sum_classes;
  // Side, in pixels, of the square tiles new surfels are numbered by (kernels_lifecycle.hip: tile_seq): 8 * cell (default: the 64
  // surfels of a wavefront form a compact patch) or, for the reference's row-major append order, a side that covers the whole image
  // (bahip_context_set_creation_order)
  int create_tile;
  // Arithmetic flavour of the sweeps (bahip_context_set_arithmetic; ba_launch.h: "Two arithmetic flavours"): 0 = the exact flavour that
  // shares every bit with the oracle, 1 = the fast flavour.  Read by the host-side dispatchers only; the kernels ignore it.
  int fast_math;
};

struct SurfelsView {
  float* data;
  uint32_t pitch;   // bytes
  uint8_t* active;
  uint32_t size;
  __device__ __forceinline__ float* row(int r) const { return reinterpret_cast<float*>(reinterpret_cast<char*>(data) + (size_t)r * pitch); }
};

// Keyframe sharding: the partial sums of the keyframe classes (4 or 8) of the normals / geometry passes (kernels_surfel.hip), the
// unit the ranks exchange.  data[(class * sums + q) * stride + surfel]; `owned` has bit c set when this rank visits class c.
struct ClassPartials {
  float* data;
  uint32_t stride;   // floats per plane (>= surfels, even: the planes travel as 64-bit integer words)
  uint32_t owned;
};

// A pointer read from a device table (KfEntry::geom, ...) is a generic pointer to the compiler, which then emits flat_load
// (checked against the LDS / scratch apertures, counted on vmcnt AND lgkmcnt).  Every such pointer here is a hipMalloc
// allocation: say so, and the gathers are global_load.
template <typename T>
__device__ __forceinline__ T load_global(const T* p) {
  if constexpr (__is_scalar(T)) return *(const __attribute__((address_space(1))) T*)(p);
  else return *p;   // class types (uchar4): no address-space-qualified copy constructor; not on a hot path
}
template <typename T>
__device__ __forceinline__ T pitched_load(const T* base, uint32_t pitch, int y, int x) {
  // 32-bit offset, 24-bit multiply (full rate; v_mul_lo_u32 is quarter rate): rows < 2^24, pitch < 2^24, images < 4 GiB
  return load_global(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (__umul24((uint32_t)y, pitch) + (uint32_t)x * (uint32_t)sizeof(T))));
}
template <typename T>
__device__ __forceinline__ T* pitched_ptr(T* base, uint32_t pitch, int y, int x) {
  return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (size_t)y * pitch + (size_t)x * sizeof(T));
}

// ---- BA planes: 8x4-pixel tiles of 32-bit words, one tile = one 128-byte line ------------------------------
// Memory order of the tiles: COLUMN STRIPS.  A strip is 8 pixels wide and as high as the (padded) image; inside a strip the rows
// follow each other (8 words = 32 bytes per row), so every aligned 128-byte line holds rows 4 k .. 4 k + 3 of the strip -- the same
// 8x4 tile per line as a row-major order of tiles, but the byte offset of pixel (x, y) is three terms,
//   (x >> 3) * strip_bytes + (y << 5) + ((x & 7) << 2)  =  (x << 2) + (x >> 3) * (strip_bytes - 32) + (y << 5),
// i.e. a shift, a 24-bit multiply-add and a shift-add (four vector instructions where the row-major tile index took eight), and as
// a 32-bit offset from the wave-uniform plane pointer it rides in the load's own address operand (no 64-bit add).  The sweeps
// compute four such addresses per (surfel, keyframe) pair.
constexpr uint32_t kPlaneTileW = 8, kPlaneTileH = 4;
__host__ __device__ __forceinline__ uint32_t plane_tiles_x(uint32_t width) { return (width + kPlaneTileW - 1) / kPlaneTileW; }
__host__ __device__ __forceinline__ uint32_t plane_tiles_y(uint32_t height) { return (height + kPlaneTileH - 1) / kPlaneTileH; }
// bytes from one strip to the next, minus the 32 bytes that `x << 2` has already advanced by then (Intrinsics::geom_skip, fp_skip)
__host__ __device__ __forceinline__ uint32_t plane_strip_skip(uint32_t height) { return plane_tiles_y(height) * kPlaneTileH * 32u - 32u; }
__device__ __forceinline__ uint32_t plane_byte_offset(uint32_t x, uint32_t y, uint32_t strip_skip) {
  return __umul24(x >> 3, strip_skip) + (x << 2) + (y << 5);   // 24-bit multiply: full rate; planes are far below 16 MiB
}
__device__ __forceinline__ uint32_t plane_load(const uint32_t* __restrict__ plane, uint32_t byte_offset) {
  return load_global(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(plane) + byte_offset));
}
// word index of pixel (x, y) (the packing kernels' view of the same order)
__host__ __device__ __forceinline__ uint32_t plane_word(uint32_t x, uint32_t y, uint32_t padded_height) {
  return (x >> 3) * (padded_height * 8u) + y * 8u + (x & 7u);
}

// ---- packing (B/util.cuh:121-153, B/util_nvcc_only.cuh:66-95) ------------------------------------
__device__ __forceinline__ float ten_bit_signed_to_float(uint32_t value) {
  const int32_t s = (int32_t)(value << 22) >> 22;  // sign-extend the low 10 bits
  return (float)s * (1.0f / 511.0f);
}
__device__ __forceinline__ Vec3 unpack_normal10(uint32_t v) {
  Vec3 n = mk3(ten_bit_signed_to_float(v), ten_bit_signed_to_float(v >> 10), ten_bit_signed_to_float(v >> 20));
  const float factor = 1.0f / norm3(n);
  return factor * n;
}
__device__ __forceinline__ uint32_t float_to_ten_bit_signed(float value) {
  const int16_t v = (int16_t)(value * 511.0f + ((value > 0) ? 0.5f : -0.5f));
  return 0x03ffu & (uint16_t)v;
}
__device__ __forceinline__ uint32_t pack_normal10(Vec3 n) {
  return float_to_ten_bit_signed(n.x) | (float_to_ten_bit_signed(n.y) << 10) | (float_to_ten_bit_signed(n.z) << 20);
}
__device__ __forceinline__ Vec3 unpack_normal8(uint16_t v) {
  Vec3 r;
  r.x = (float)(int8_t)(v & 0xff) * (1.0f / 127.0f);
  r.y = (float)(int8_t)(v >> 8) * (1.0f / 127.0f);
  const float z = 1 - r.x * r.x - r.y * r.y;
  r.z = -sqrt_exact((z > 0.f) ? z : 0.f);   // z is 0 or >= 2^-24 (x, y are multiples of 1 / 127): inside sqrt_exact's range
  return r;
}
__device__ __forceinline__ uint16_t pack_normal8(float x, float y) {
  const int8_t sx = (int8_t)(x * 127.0f + ((x > 0) ? 0.5f : -0.5f));
  const int8_t sy = (int8_t)(y * 127.0f + ((y > 0) ? 0.5f : -0.5f));
  return (uint16_t)((uint16_t)(uint8_t)sx | ((uint16_t)(uint8_t)sy << 8));
}

// ---- transforms ----------------------------------------------------------------------------------
__device__ __forceinline__ Vec3 transform34(const float* F, Vec3 p) {
  return mk3(mad(F[2], p.z, mad(F[1], p.y, mad(F[0], p.x, F[3]))),
             mad(F[6], p.z, mad(F[5], p.y, mad(F[4], p.x, F[7]))),
             mad(F[10], p.z, mad(F[9], p.y, mad(F[8], p.x, F[11]))));
}
__device__ __forceinline__ Vec3 rotate34(const float* F, Vec3 p) {
  return mk3(mad(F[2], p.z, mad(F[1], p.y, F[0] * p.x)),
             mad(F[6], p.z, mad(F[5], p.y, F[4] * p.x)),
             mad(F[10], p.z, mad(F[9], p.y, F[8] * p.x)));
}
__device__ __forceinline__ Vec3 mul33(const float* R, Vec3 p) {
  return mk3(mad(R[2], p.z, mad(R[1], p.y, R[0] * p.x)),
             mad(R[5], p.z, mad(R[4], p.y, R[3] * p.x)),
             mad(R[8], p.z, mad(R[7], p.y, R[6] * p.x)));
}

// ---- depth ---------------------------------------------------------------------------------------
// B/util.cuh:62-69
__device__ __forceinline__ float raw_to_calibrated_depth(float a, float cfactor, float raw_to_float_depth, uint16_t raw) {
  const float inv_depth = rcp_exact(raw_to_float_depth * raw);
  // a == 0 (wave-uniform: a kernel argument; the value of every run that has not optimised the depth intrinsics):
  // exp(-0 * inv_depth) is exactly 1 for every finite inv_depth, so the exponential is skipped -- the same bits; raw == 0
  // (inv_depth = inf, -0 * inf = NaN) keeps its NaN.
  if (a == 0.f) return rcp_exact(mad(cfactor, (raw == 0) ? __builtin_nanf("") : 1.f, inv_depth));
  return rcp_exact(mad(cfactor, exp_det(-a * inv_depth), inv_depth));
}
__device__ __forceinline__ float cfactor_at(const Intrinsics& in, int px, int py) {
  // px, py >= 0, so the shift is the same integer division
  if (in.cell_shift >= 0) return pitched_load(in.cfactor, in.cfactor_pitch, py >> in.cell_shift, px >> in.cell_shift);
  return pitched_load(in.cfactor, in.cfactor_pitch, py / in.cell, px / in.cell);
}
__device__ __forceinline__ float unp_nx(const Intrinsics& in, float px) { return mad(in.fx_inv, px, in.cx_inv); }
__device__ __forceinline__ float unp_ny(const Intrinsics& in, float py) { return mad(in.fy_inv, py, in.cy_inv); }
__device__ __forceinline__ Vec3 unproject(const Intrinsics& in, int x, int y, float depth) {
  return mk3(depth * mad(in.fx_inv, (float)x, in.cx_inv), depth * mad(in.fy_inv, (float)y, in.cy_inv), depth);
}
// B/cost_function.cuh:81-88
// the factor both functions below share: 0.1 |n . (nx, ny, 1)| depth^2
__device__ __forceinline__ float depth_sigma_factor(float nx, float ny, float depth, Vec3 nl) {
  return 0.1f * fabsf(mad(nl.y, ny, mad(nl.x, nx, nl.z))) * (depth * depth);
}
__device__ __forceinline__ float depth_stddev(float nx, float ny, float depth, Vec3 nl, float baseline_fx) {
  return (0.1f * fabsf(mad(nl.y, ny, mad(nl.x, nx, nl.z))) * (depth * depth)) * (1.f / baseline_fx);
}
__device__ __forceinline__ float depth_inv_stddev(float nx, float ny, float depth, Vec3 nl, float baseline_fx) {
  return baseline_fx / (0.1f * fabsf(mad(nl.y, ny, mad(nl.x, nx, nl.z))) * (depth * depth));
}

// ---- robust weights (B/robust_weighting.cuh:39-86; parameters B/cost_function.cuh:44-52,105-109) ---
__device__ __forceinline__ float tukey_weight10(float r) {
  if (fabsf(r) < 10.f) { const float q = r * 0.1f; const float t = 1.f - q * q; return t * t; }
  return 0.f;
}
__device__ __forceinline__ float huber_weight10(float r) {
  const float a = fabsf(r);
  if (!__any(a >= 10.f)) return 1.f;   // no outlier in the wave: skip the division (same values either way)
  return (a < 10.f) ? 1.f : (10.f / a);
}
__device__ __forceinline__ float depth_residual_weight(float r) { return 1.f * tukey_weight10(r); }
__device__ __forceinline__ float descriptor_residual_weight(float r) { return 1.f * 1e-2f * huber_weight10(r); }

// ---- association -----------------------------------------------------------------------------------
struct Assoc {
  Vec3 local;        // surfel position in the keyframe frame
  float inv_z;       // 1 / local.z (IEEE), shared by the projection and the descriptor Jacobians
  Vec3 nl;           // surfel normal in the keyframe frame
  float depth;       // calibrated depth of the associated pixel
  int px, py;
  float pxx, pxy;    // float pixel position, pixel-corner convention
  uint16_t normal_bits;   // packed measured normal of the associated pixel
  // what the depth residual shares with the association test (assoc_inv_std / assoc_unproject below): the unprojection
  // coordinates of the pixel and 0.1 |n . (nx, ny, 1)| depth^2, the factor of depth_stddev and depth_inv_stddev
  float nx, ny, sigma;
};

// B/surfel_projection_nvcc_only.cuh:332-359 with IsAssociatedWithPixel :48-127; the order of the
// rejection tests is the reference's.  A NaN position (deleted surfel) is rejected explicitly.
// `gp`, `gn`: global position and (decoded, renormalised) global normal of the surfel.
//
// The test is written in three phases so that a sweep can put every gather of a (surfel, keyframe) pair in flight at once:
//   project_surfel        arithmetic only: the position in the keyframe frame, the pixel, "projects into the image";
//   load_pixel_words      the packed depth + normal word of that pixel and the cfactor of its cell, from CLAMPED coordinates
//                         (always a valid address, so the loads are unconditional and sit in straight-line code; for a lane
//                         that projects into the image the clamp is the identity);
//   associate_from_words  the remaining tests on the loaded values.
// A wavefront spends most of its time waiting for gathers (61 % of the wave-cycles of the geometry sweep, PMC SQ_WAIT_ANY,
// profiles/r2_c_stall.txt): pixel word -> cfactor -> luma footprints used to be three dependent round trips per pair; issued
// together they are one.  project_associate() is the three phases in sequence (every other caller).
struct Projected {
  Vec3 local;        // surfel position in the keyframe frame
  float inv_z;       // 1 / local.z (IEEE)
  float pxx, pxy;    // float pixel position, pixel-corner convention
  int px, py;
  float fpx, fpy;    // (float)px, (float)py
  bool ok;           // z > 0 and the pixel lies in the image
};
__device__ __forceinline__ Projected project_surfel(const Intrinsics& in, const float* F, Vec3 gp) {
  Projected p;
  p.local.z = mad(F[10], gp.z, mad(F[9], gp.y, mad(F[8], gp.x, F[11])));
  p.local.x = mad(F[2], gp.z, mad(F[1], gp.y, mad(F[0], gp.x, F[3])));
  p.local.y = mad(F[6], gp.z, mad(F[5], gp.y, mad(F[4], gp.x, F[7])));
  p.inv_z = rcp_exact(p.local.z);   // == 1.f / z; one reciprocal shared by both coordinates (and by the Jacobians), as in the oracle
  p.pxx = mad(in.fx, p.local.x * p.inv_z, in.cx);
  p.pxy = mad(in.fy, p.local.y * p.inv_z, in.cy);
  p.px = (int)p.pxx;
  p.py = (int)p.pxy;
  p.fpx = (float)p.px;
  p.fpy = (float)p.py;
  // 0 <= pxx < (float)width implies (int)pxx < width: the reference's integer compares (B/surfel_projection_nvcc_only.cuh:343) add nothing
  p.ok = (p.local.z > 0.f) && (p.pxx >= 0.f) && (p.pxy >= 0.f) && (p.pxx < (float)in.width) && (p.pxy < (float)in.height);
  return p;
}
struct PixelWords {
  uint32_t geom;     // measured depth (low half) and packed measured normal (high half)
  float cfactor;
};
__device__ __forceinline__ PixelWords load_pixel_words(const Intrinsics& in, const uint32_t* __restrict__ geom, const Projected& p) {
  const int cx = min(max(p.px, 0), in.width - 1), cy = min(max(p.py, 0), in.height - 1);
  PixelWords w;
  w.geom = plane_load(geom, plane_byte_offset((uint32_t)cx, (uint32_t)cy, in.geom_skip));
  w.cfactor = cfactor_at(in, cx, cy);
  return w;
}
template <bool kFreeSpace>
__device__ __forceinline__ bool associate_from_words(const Intrinsics& in, const float* F, Vec3 gn, const Projected& p,
                                                     const PixelWords& w, Assoc* r, bool* free_space_violation) {
  if (!p.ok) return false;
  r->local = p.local; r->inv_z = p.inv_z; r->pxx = p.pxx; r->pxy = p.pxy; r->px = p.px; r->py = p.py;
  const uint16_t measured = (uint16_t)(w.geom & 0xffffu);
  if (measured & kInvalidDepthBit) return false;
  r->depth = raw_to_calibrated_depth(in.a, w.cfactor, in.raw_to_float_depth, measured);
  r->nl = rotate34(F, gn);
  r->nx = unp_nx(in, p.fpx);
  r->ny = unp_ny(in, p.fpy);
  r->sigma = depth_sigma_factor(r->nx, r->ny, r->depth, r->nl);
  const float thr = 10.f * (r->sigma * (1.f / in.baseline_fx));   // == 10 depth_stddev(nx, ny, depth, nl, baseline_fx)
  if (kFreeSpace) {
    const float diff = r->depth - r->local.z;
    if (diff > thr) { *free_space_violation = true; return false; }
    else if (diff < -thr) return false;
  } else {
    if (fabsf(r->local.z - r->depth) > thr) return false;
  }
  // the reference tests (1 / |p|) * dot(p, n) > 0; |p| > 0 here, so the sign test needs no normalisation (oracle: same)
  if (dot3(r->local, r->nl) > 0) return false;
  r->normal_bits = (uint16_t)(w.geom >> 16);
  const Vec3 m = unpack_normal8(r->normal_bits);
  if (dot3(r->nl, m) < kCosNormalCompat) return false;
  return true;
}
template <bool kFreeSpace>
__device__ __forceinline__ bool project_associate(const Intrinsics& in, const float* F, const uint32_t* __restrict__ geom,
                                                  Vec3 gp, Vec3 gn, Assoc* r, bool* free_space_violation) {
  const Projected p = project_surfel(in, F, gp);
  if (!p.ok) return false;
  const PixelWords w = load_pixel_words(in, geom, p);
  return associate_from_words<kFreeSpace>(in, F, gn, p, w, r, free_space_violation);
}

// The depth residual's inputs from what the association test already computed (the same expressions as
// depth_inv_stddev(unp_nx(px), unp_ny(py), depth, nl, baseline_fx) and unproject(px, py, depth), on the same values).
__device__ __forceinline__ float assoc_inv_std(const Intrinsics& in, const Assoc& r) { return in.baseline_fx / r.sigma; }
__device__ __forceinline__ Vec3 assoc_unproject(const Assoc& r) { return mk3(r.depth * r.nx, r.depth * r.ny, r.depth); }

// ---- colour sampling -------------------------------------------------------------------------------
// 2x2 luma footprint with top-left texel (ix, iy), ix in [-1, w], iy in [-1, h] (clamp addressing baked in).
struct Luma4 { float tl, tr, bl, br; };
__device__ __forceinline__ uint32_t luma_footprint_word(const Intrinsics& in, const uint32_t* __restrict__ lumafp, int ix, int iy) {
  return plane_load(lumafp, plane_byte_offset((uint32_t)(ix + 1), (uint32_t)(iy + 1), in.fp_skip));
}
__device__ __forceinline__ Luma4 unpack_luma_footprint(uint32_t word) {
  Luma4 t;
  t.tl = (float)(word & 0xffu) * (1.0f / 255.0f);
  t.tr = (float)((word >> 8) & 0xffu) * (1.0f / 255.0f);
  t.bl = (float)((word >> 16) & 0xffu) * (1.0f / 255.0f);
  t.br = (float)(word >> 24) * (1.0f / 255.0f);
  return t;
}
__device__ __forceinline__ Luma4 luma_footprint(const Intrinsics& in, const uint32_t* __restrict__ lumafp, int ix, int iy) {
  const uint32_t word = luma_footprint_word(in, lumafp, ix, iy);
  Luma4 t;
  t.tl = (float)(word & 0xffu) * (1.0f / 255.0f);
  t.tr = (float)((word >> 8) & 0xffu) * (1.0f / 255.0f);
  t.bl = (float)((word >> 16) & 0xffu) * (1.0f / 255.0f);
  t.br = (float)(word >> 24) * (1.0f / 255.0f);
  return t;
}
// -DBAHIP_QUANTIZED_BILINEAR_WEIGHTS (experiment build only, tests/tools/seed_study.sh): the two interpolation weights of a
// bilinear VALUE sample are rounded to 8 fractional bits, as CUDA's tex2D does in hardware (the reference's sampler; this
// backend and the oracle use exact binary32 weights).  The gradient taps are point samples lerped in float in
// the reference too and are not affected.
__device__ __forceinline__ float bilinear_weight(float w) {
#ifdef BAHIP_QUANTIZED_BILINEAR_WEIGHTS
  return floorf(w * 256.f + 0.5f) * (1.f / 256.f);
#else
  return w;
#endif
}
// Bilinear luma at unnormalised coords, clamp addressing, texel centres at +0.5 (B/keyframe.cc:67-73).
__device__ __forceinline__ float sample_luma(const Intrinsics& in, const uint32_t* lumafp, int w, int h, float x, float y) {
  float xb = x - 0.5f, yb = y - 0.5f;
  if (!(xb >= -1.f)) xb = -1.f;
  if (xb > (float)w) xb = (float)w;
  if (!(yb >= -1.f)) yb = -1.f;
  if (yb > (float)h) yb = (float)h;
  const float fx = floorf(xb), fy = floorf(yb);
  const float a = bilinear_weight(xb - fx), b = bilinear_weight(yb - fy);
  const Luma4 t = luma_footprint(in, lumafp, (int)fx, (int)fy);
  const float top = mad(a, t.tr - t.tl, t.tl);
  const float bot = mad(a, t.br - t.bl, t.bl);
  return mad(b, bot - top, top);
}
// Bilinear luma AND the gradient sample of DescriptorJacobianWrtProjectedPosition (B/cost_function.cuh:200-211) at the
// same point.  Away from the image border both use the same footprint (floor(x - 0.5) == trunc(max(0, x - 0.5))), which
// is fetched once; at the border the gradient's footprint is fetched separately, so the results are those of the
// reference's two separate evaluations in every case.
__device__ __forceinline__ void sample_luma_and_gradient(const Intrinsics& in, const uint32_t* lumafp, int w, int h, float x, float y,
                                                         float* value, float* dx, float* dy) {
  float xb = x - 0.5f, yb = y - 0.5f;
  if (!(xb >= -1.f)) xb = -1.f;
  if (xb > (float)w) xb = (float)w;
  if (!(yb >= -1.f)) yb = -1.f;
  if (yb > (float)h) yb = (float)h;
  const float fx = floorf(xb), fy = floorf(yb);
  const float a = xb - fx, b = yb - fy;
  const int ix = (int)fx, iy = (int)fy;
  Luma4 t = luma_footprint(in, lumafp, ix, iy);
  {
    const float qa = bilinear_weight(a), qb = bilinear_weight(b);
    const float top = mad(qa, t.tr - t.tl, t.tl);
    const float bot = mad(qa, t.br - t.bl, t.bl);
    *value = mad(qb, bot - top, top);
  }

  float mx = fmaxf(0.f, x - 0.5f), my = fmaxf(0.f, y - 0.5f);
  if (!(mx < (float)w)) mx = (float)w;
  if (!(my < (float)h)) my = (float)h;
  const int gx = (int)mx, gy = (int)my;
  const float tx = fmaxf(0.f, fminf(1.f, x - 0.5f - gx));
  const float ty = fmaxf(0.f, fminf(1.f, y - 0.5f - gy));
  if (gx != ix || gy != iy) t = luma_footprint(in, lumafp, gx, gy);
  *dx = mad(t.br - t.bl, ty, (t.tr - t.tl) * (1 - ty));
  *dy = mad(t.br - t.tr, tx, (t.bl - t.tl) * (1 - tx));
}

// The same for a point whose footprint lies inside the image (0 <= x - 0.5 < w, 0 <= y - 0.5 < h): none of the clamps
// above changes a value then (xb = x - 0.5, trunc(max(0, xb)) == floor(xb), and x - 0.5 - gx is the fraction a in [0, 1)),
// so they are not evaluated.  Same operations in the same order on the values that matter: identical results.
__device__ __forceinline__ bool luma_sample_is_interior(int w, int h, float x, float y) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  // (bitwise on purpose: `&&` made the compiler guard each comparison by the one before it -- a save-exec and a branch per test, six
  // of them per candidate for the three sample points -- where four compares and three scalar ANDs do)
  return (bool)((int)(xb >= 0.f) & (int)(xb < (float)w) & (int)(yb >= 0.f) & (int)(yb < (float)h));
}
// The footprint word of the sample at (x, y), from coordinates clamped into the plane: the word the interior sampler uses
// when the point is interior, some valid word otherwise (the caller then does not use it).  NaN coordinates clamp too.
__device__ __forceinline__ uint32_t luma_word_clamped(const Intrinsics& in, const uint32_t* __restrict__ lumafp, float x, float y) {
  // v_med3_f32: one instruction, and none to quiet a NaN first (a NaN coordinate gives the minimum of the bounds: valid)
  // No floor: for an interior point (0 <= x - 0.5 < w) the clamp is the identity and the conversion's truncation IS the floor; anywhere
  // else (-1 <= clamped < 0 truncates to 0 instead of -1) the result is still a valid word, which the caller does not use.
  const float ux = __builtin_amdgcn_fmed3f(x - 0.5f, -1.f, (float)in.cwidth);
  const float uy = __builtin_amdgcn_fmed3f(y - 0.5f, -1.f, (float)in.cheight);
  return luma_footprint_word(in, lumafp, (int)ux, (int)uy);
}
__device__ __forceinline__ void sample_luma_and_gradient_interior(uint32_t word, float x, float y, float* value, float* dx, float* dy) {
  const float xb = x - 0.5f, yb = y - 0.5f;
  // xb, yb >= 0 here: xb - floor(xb) is exact, which is what v_fract_f32 returns (its clamp below 1 never acts on an exact fraction)
  const float a = __builtin_amdgcn_fractf(xb), b = __builtin_amdgcn_fractf(yb);
  const Luma4 t = unpack_luma_footprint(word);
  {
    const float qa = bilinear_weight(a), qb = bilinear_weight(b);
    const float top = mad(qa, t.tr - t.tl, t.tl);
    const float bot = mad(qa, t.br - t.bl, t.bl);
    *value = mad(qb, bot - top, top);
  }
  *dx = mad(t.br - t.bl, b, (t.tr - t.tl) * (1 - b));
  *dy = mad(t.br - t.tr, a, (t.bl - t.tl) * (1 - a));
}

// B/surfel_projection.cuh:194-207
__device__ __forceinline__ bool depth_to_color_pixel(const Intrinsics& in, float pxx, float pxy, float* cx, float* cy) {
  *cx = mad(in.d2c_fx, pxx, in.d2c_cx);
  *cy = mad(in.d2c_fy, pxy, in.d2c_cy);
  return *cx >= 0 && *cy >= 0 && *cx < (float)in.cwidth && *cy < (float)in.cheight;   // for cx >= 0: (int)cx < w <=> cx < w (w an integer)
}

// B/cost_function.cuh:115-136.  The two tangent sample points gp + t1, gp + t2 depend on the surfel only, so the hot
// kernels compute them once per surfel (surfel_tangent_points) and project them per keyframe (project_tangents).
struct TangentPoints { Vec3 q1, q2; };
__device__ __forceinline__ TangentPoints surfel_tangent_points(Vec3 gp, Vec3 gn, float radius_sq) {
  Vec3 t1 = cross3(gn, (fabsf(gn.x) > 0.9f) ? mk3(0, 1, 0) : mk3(1, 0, 0));
  t1 = (2.0f * sqrtf(radius_sq / fmaxf(1e-12f, sqlen3(t1)))) * t1;
  Vec3 t2 = cross3(gn, t1);
  t2 = (2.0f * sqrtf(radius_sq / fmaxf(1e-12f, sqlen3(t2)))) * t2;
  TangentPoints tp;
  tp.q1 = gp + t1;
  tp.q2 = gp + t2;
  return tp;
}
__device__ __forceinline__ void project_tangents(const Intrinsics& in, const float* F, const TangentPoints& tp,
                                                 float* t1x, float* t1y, float* t2x, float* t2y) {
  const Vec3 l1 = transform34(F, tp.q1);
  const float inv_z1 = rcp_exact(l1.z);
  *t1x = mad(in.cfx, l1.x * inv_z1, in.ccx);
  *t1y = mad(in.cfy, l1.y * inv_z1, in.ccy);
  const Vec3 l2 = transform34(F, tp.q2);
  const float inv_z2 = rcp_exact(l2.z);
  *t2x = mad(in.cfx, l2.x * inv_z2, in.ccx);
  *t2y = mad(in.cfy, l2.y * inv_z2, in.ccy);
}
__device__ __forceinline__ void tangent_projections(const Intrinsics& in, const float* F, Vec3 gp, Vec3 gn, float radius_sq,
                                                    float* t1x, float* t1y, float* t2x, float* t2y) {
  project_tangents(in, F, surfel_tangent_points(gp, gn, radius_sq), t1x, t1y, t2x, t2y);
}

// Descriptor residuals (B/cost_function.cuh:140-156) and gradients (:191-254) of one pair.
struct DescEval {
  float r1, r2;       // raw residuals
  float gx1, gy1, gx2, gy2;
};
template <bool kWithGradient>
__device__ __forceinline__ void eval_descriptor(const Intrinsics& in, const uint32_t* lumafp, const float* F,
                                                const TangentPoints& tp, float cx, float cy, float d1, float d2, DescEval* e) {
  float t1x, t1y, t2x, t2y;
  project_tangents(in, F, tp, &t1x, &t1y, &t2x, &t2y);
  const int w = in.cwidth, h = in.cheight;
  if (kWithGradient) {
    float i0, i1, i2, cdx, cdy, adx, ady, bdx, bdy;
    if ((int)luma_sample_is_interior(w, h, cx, cy) & (int)luma_sample_is_interior(w, h, t1x, t1y) & (int)luma_sample_is_interior(w, h, t2x, t2y)) {
      sample_luma_and_gradient_interior(luma_word_clamped(in, lumafp, cx, cy), cx, cy, &i0, &cdx, &cdy);
      sample_luma_and_gradient_interior(luma_word_clamped(in, lumafp, t1x, t1y), t1x, t1y, &i1, &adx, &ady);
      sample_luma_and_gradient_interior(luma_word_clamped(in, lumafp, t2x, t2y), t2x, t2y, &i2, &bdx, &bdy);
    } else {
      sample_luma_and_gradient(in, lumafp, w, h, cx, cy, &i0, &cdx, &cdy);
      sample_luma_and_gradient(in, lumafp, w, h, t1x, t1y, &i1, &adx, &ady);
      sample_luma_and_gradient(in, lumafp, w, h, t2x, t2y, &i2, &bdx, &bdy);
    }
    e->r1 = mad(180.f, i1 - i0, -d1);
    e->r2 = mad(180.f, i2 - i0, -d2);
    e->gx1 = 180.f * (adx - cdx);
    e->gy1 = 180.f * (ady - cdy);
    e->gx2 = 180.f * (bdx - cdx);
    e->gy2 = 180.f * (bdy - cdy);
  } else {
    const float i0 = sample_luma(in, lumafp, w, h, cx, cy);
    const float i1 = sample_luma(in, lumafp, w, h, t1x, t1y);
    const float i2 = sample_luma(in, lumafp, w, h, t2x, t2y);
    e->r1 = mad(180.f, i1 - i0, -d1);
    e->r2 = mad(180.f, i2 - i0, -d2);
  }
}
// Split-phase form for the sweeps (see project_surfel): load_descriptor_words() needs only the projection, so its three
// gathers are in flight together with the pixel word; eval_descriptor_from_words() is eval_descriptor<true> on the loaded
// words (samples near the image border take the general sampler, which loads its footprints itself).
struct DescWords {
  float cx, cy, t1x, t1y, t2x, t2y;
  uint32_t w0, w1, w2;
  bool color_ok;     // depth_to_color_pixel succeeded (B/kernel_opt_pose.cu:303-353: nothing is added otherwise)
  bool interior;     // all three footprints lie inside the image
};
__device__ __forceinline__ DescWords load_descriptor_words(const Intrinsics& in, const uint32_t* __restrict__ lumafp, const float* F,
                                                           const TangentPoints& tp, const Projected& p) {
  DescWords d;
  d.color_ok = depth_to_color_pixel(in, p.pxx, p.pxy, &d.cx, &d.cy);
  project_tangents(in, F, tp, &d.t1x, &d.t1y, &d.t2x, &d.t2y);
  const int w = in.cwidth, h = in.cheight;
  d.interior = (bool)((int)luma_sample_is_interior(w, h, d.cx, d.cy) & (int)luma_sample_is_interior(w, h, d.t1x, d.t1y) &
                      (int)luma_sample_is_interior(w, h, d.t2x, d.t2y));
  d.w0 = luma_word_clamped(in, lumafp, d.cx, d.cy);
  d.w1 = luma_word_clamped(in, lumafp, d.t1x, d.t1y);
  d.w2 = luma_word_clamped(in, lumafp, d.t2x, d.t2y);
  return d;
}
__device__ __forceinline__ void eval_descriptor_from_words(const Intrinsics& in, const uint32_t* lumafp, const DescWords& d, float d1,
                                                           float d2, DescEval* e) {
  const int w = in.cwidth, h = in.cheight;
  float i0, i1, i2, cdx, cdy, adx, ady, bdx, bdy;
  // The interior form for every lane -- the three words were loaded from clamped coordinates, so they are valid memory for any lane,
  // and a lane whose footprints are not all inside the image computes values that are replaced below -- and then, behind a
  // wave-uniform test that practically never fires, the border form for the lanes that need it.  (As an if / else per lane the three
  // samples compiled into three diamonds with their save-exec pairs and nine register copies, on every candidate.)
  sample_luma_and_gradient_interior(d.w0, d.cx, d.cy, &i0, &cdx, &cdy);
  sample_luma_and_gradient_interior(d.w1, d.t1x, d.t1y, &i1, &adx, &ady);
  sample_luma_and_gradient_interior(d.w2, d.t2x, d.t2y, &i2, &bdx, &bdy);
  if (__builtin_amdgcn_ballot_w64(!d.interior) != 0ull) {
    if (!d.interior) {
      sample_luma_and_gradient(in, lumafp, w, h, d.cx, d.cy, &i0, &cdx, &cdy);
      sample_luma_and_gradient(in, lumafp, w, h, d.t1x, d.t1y, &i1, &adx, &ady);
      sample_luma_and_gradient(in, lumafp, w, h, d.t2x, d.t2y, &i2, &bdx, &bdy);
    }
  }
  e->r1 = mad(180.f, i1 - i0, -d1);
  e->r2 = mad(180.f, i2 - i0, -d2);
  e->gx1 = 180.f * (adx - cdx);
  e->gy1 = 180.f * (ady - cdy);
  e->gx2 = 180.f * (bdx - cdx);
  e->gy2 = 180.f * (bdy - cdy);
}
// "Every gather of this pair has arrived", placed after the association tests of a sweep.  An empty asm statement that reads
// the loaded words on every path: it keeps the compiler from sinking a gather into the branch that uses it (which would put
// the round trips back in series), and no load is pending at the loop's back edge (a pending one costs an s_waitcnt vmcnt(0)
// at the top of every following candidate).
__device__ __forceinline__ void gathers_arrived(const PixelWords& pix) { asm volatile("" ::"v"(pix.geom), "v"(pix.cfactor)); }
__device__ __forceinline__ void gathers_arrived(const PixelWords& pix, const DescWords& d) {
  asm volatile("" ::"v"(pix.geom), "v"(pix.cfactor), "v"(d.w0), "v"(d.w1), "v"(d.w2));
}
template <bool kWithGradient>
__device__ __forceinline__ void eval_descriptor(const Intrinsics& in, const uint32_t* lumafp, const float* F,
                                                Vec3 gp, Vec3 gn, float radius_sq, float cx, float cy, float d1, float d2,
                                                DescEval* e) {
  eval_descriptor<kWithGradient>(in, lumafp, F, surfel_tangent_points(gp, gn, radius_sq), cx, cy, d1, d2, e);
}

// ---- residual Jacobians (the same functions the oracle has, oracle_internal.h: jac_*; checked in isolation against golden
// vectors generated from the reference's derivation script, tests/golden/jacobians.json) ---------------------------------
// B/kernel_opt_pose.cu:88-93: nl = surfel normal, u = unprojected measurement, both in the keyframe frame.
__device__ __forceinline__ void jac_depth_pose(Vec3 nl, Vec3 u, float inv_std, float (&J)[6]) {
  J[0] = inv_std * nl.x;
  J[1] = inv_std * nl.y;
  J[2] = inv_std * nl.z;
  J[3] = inv_std * mad(nl.z, u.y, -(nl.y * u.z));
  J[4] = inv_std * mad(nl.x, u.z, -(nl.z * u.x));
  J[5] = inv_std * mad(nl.y, u.x, -(nl.x * u.y));
}
// B/kernel_opt_pose.cu:126-141: ls = surfel position in the keyframe frame, gx, gy = image gradient of the residual times fx, fy.
// inv_z = 1.f / ls.z: the hot kernels pass the reciprocal the projection already computed (Assoc::inv_z, the same value).
__device__ __forceinline__ void jac_descriptor_pose(Vec3 ls, float inv_z, float gx, float gy, float (&J)[6]) {
  const float z_sq = ls.z * ls.z, inv_z_sq = inv_z * inv_z, xy = ls.x * ls.y;
  J[0] = -gx * inv_z;
  J[1] = -gy * inv_z;
  J[2] = mad(ls.y, gy, ls.x * gx) * inv_z_sq;
  J[3] = mad(mad(ls.y, ls.y, z_sq), gy, xy * gx) * inv_z_sq;
  J[4] = -mad(mad(ls.x, ls.x, z_sq), gx, xy * gy) * inv_z_sq;
  J[5] = -mad(ls.x, gy, -(ls.y * gx)) * inv_z;
}
__device__ __forceinline__ void jac_descriptor_pose(Vec3 ls, float gx, float gy, float (&J)[6]) {
  jac_descriptor_pose(ls, 1.f / ls.z, gx, gy, J);
}
// B/kernel_opt_geometry.cu:170-190: rn = surfel normal, lp = surfel position in the keyframe frame, g = gradient per pixel.
__device__ __forceinline__ float jac_descriptor_surfel(Vec3 rn, Vec3 lp, float inv_z, float gx, float gy, float cfx, float cfy) {
  const float term1 = -cfx * mad(rn.x, lp.z, -(rn.z * lp.x));
  const float term2 = -cfy * mad(rn.y, lp.z, -(rn.z * lp.y));
  const float term3 = inv_z * inv_z;
  return -mad(gy, term2, gx * term1) * term3;
}
__device__ __forceinline__ float jac_descriptor_surfel(Vec3 rn, Vec3 lp, float gx, float gy, float cfx, float cfy) {
  return jac_descriptor_surfel(rn, lp, 1.f / lp.z, gx, gy, cfx, cfy);
}
// B/kernel_opt_intrinsics.cu:107-140: rows fx_inv, fy_inv, cx_inv, cy_inv, a, cfactor.
__device__ __forceinline__ void jac_depth_intrinsics(int px, int py, float depth, float inv_std, float n_dot_Frow0, float n_dot_Frow1,
                                                     float dot, float cfactor, float raw_inv_depth, float exp_inv_depth,
                                                     float corrected_inv_depth, float (&J)[6]) {
  const float jac_base = inv_std * dot * exp_inv_depth / (corrected_inv_depth * corrected_inv_depth);
  J[2] = inv_std * depth * n_dot_Frow0;
  J[3] = inv_std * depth * n_dot_Frow1;
  J[0] = px * J[2];
  J[1] = py * J[3];
  J[4] = cfactor * raw_inv_depth * jac_base;
  J[5] = -jac_base;
}
// B/kernel_opt_intrinsics.cu:176-199: rows fx, fy, cx, cy of the colour camera.
__device__ __forceinline__ void jac_descriptor_color_intrinsics(float gx, float gy, float nx, float ny, float (&J)[4]) {
  J[0] = gx * nx; J[1] = gy * ny; J[2] = gx; J[3] = gy;
}

// ---- surfel loads ----------------------------------------------------------------------------------
__device__ __forceinline__ Vec3 surfel_position(const SurfelsView& s, uint32_t i) {
  return mk3(s.row(kSurfelX)[i], s.row(kSurfelY)[i], s.row(kSurfelZ)[i]);
}
__device__ __forceinline__ Vec3 surfel_normal(const SurfelsView& s, uint32_t i) {
  return unpack_normal10(reinterpret_cast<const uint32_t*>(s.row(kSurfelNormal))[i]);
}

}  // namespace bahip

#include "wave_reduce.h"   // wave_sum, wave_reduce28
