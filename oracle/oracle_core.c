/* oracle_core.c -- SE3, packing, sampler, association and residuals of the CPU oracle.
 * Test infrastructure only (see oracle.h).  File:line citations refer to /root/reference,
 * B/ = applications/badslam/src/badslam/. */
#include "oracle_internal.h"

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * SE3 / SO3 following Sophus (libvis/third_party/sophus/sophus/so3.hpp, se3.hpp), binary32.
 * Quaternion storage order here: (x, y, z, w).
 * ---------------------------------------------------------------------------------------- */
#define SOPHUS_EPS_F 1e-5f /* common.hpp:144-148 */

void orc_se3_identity(orc_se3* T) {
  T->q[0] = T->q[1] = T->q[2] = 0.f; T->q[3] = 1.f;
  T->t[0] = T->t[1] = T->t[2] = 0.f;
}

/* Eigen quaternion product a*b (Hamilton). */
static void quat_mul(const float* a, const float* b, float* o) {
  const float ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const float bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

/* Eigen QuaternionBase::_transformVector */
static v3 quat_rotate(const float* q, v3 v) {
  v3 qv = v3_make(q[0], q[1], q[2]);
  v3 uv = v3_cross(qv, v);
  uv = v3_add(uv, uv);
  v3 c = v3_cross(qv, uv);
  return v3_make(v.x + q[3] * uv.x + c.x, v.y + q[3] * uv.y + c.y, v.z + q[3] * uv.z + c.z);
}

/* Eigen QuaternionBase::toRotationMatrix */
void orc_se3_rotation(const orc_se3* T, float r[9]) {
  const float x = T->q[0], y = T->q[1], z = T->q[2], w = T->q[3];
  const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  r[0] = 1 - (tyy + tzz); r[1] = txy - twz;       r[2] = txz + twy;
  r[3] = txy + twz;       r[4] = 1 - (txx + tzz); r[5] = tyz - twx;
  r[6] = txz - twy;       r[7] = tyz + twx;       r[8] = 1 - (txx + tyy);
}

void orc_se3_matrix3x4(const orc_se3* T, float m[12]) {
  float r[9];
  orc_se3_rotation(T, r);
  m[0] = r[0]; m[1] = r[1]; m[2] = r[2];  m[3] = T->t[0];
  m[4] = r[3]; m[5] = r[4]; m[6] = r[5];  m[7] = T->t[1];
  m[8] = r[6]; m[9] = r[7]; m[10] = r[8]; m[11] = T->t[2];
}

/* se3.hpp:203-207 and so3.hpp:215-233 */
void orc_se3_mul(const orc_se3* a, const orc_se3* b, orc_se3* out) {
  orc_se3 r;
  v3 rt = quat_rotate(a->q, v3_make(b->t[0], b->t[1], b->t[2]));
  r.t[0] = a->t[0] + rt.x; r.t[1] = a->t[1] + rt.y; r.t[2] = a->t[2] + rt.z;
  quat_mul(a->q, b->q, r.q);
  const float sq = r.q[0] * r.q[0] + r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3];
  if (sq != 1.0f) {
    const float f = 2.0f / (1.0f + sq);
    r.q[0] *= f; r.q[1] *= f; r.q[2] *= f; r.q[3] *= f;
  }
  *out = r;
}

/* se3.hpp:127-130 */
void orc_se3_inverse(const orc_se3* a, orc_se3* out) {
  orc_se3 r;
  r.q[0] = -a->q[0]; r.q[1] = -a->q[1]; r.q[2] = -a->q[2]; r.q[3] = a->q[3];
  v3 t = quat_rotate(r.q, v3_make(a->t[0] * -1.f, a->t[1] * -1.f, a->t[2] * -1.f));
  r.t[0] = t.x; r.t[1] = t.y; r.t[2] = t.z;
  *out = r;
}

static void mat3_mul(const float* a, const float* b, float* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      o[3 * i + j] = a[3 * i + 0] * b[0 + j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

/* sin / cos of a binary32 angle by explicit binary64 operations: the backend defines them this way (se3_device.h:
 * sincos_det) because device and host math libraries differ in the last bit of sinf / cosf; restated here operation by
 * operation (pi/2 split in two parts, Taylor polynomials on |r| <= pi/4, fused multiply-adds, one final rounding). */
/* exp as the backend defines it (ba_device.h: exp_det), restated operation by operation: binary32 reduction by ln 2 in two
 * pieces, Taylor polynomial of degree 7 in fused multiply-adds, ldexpf. */
float orc_exp(float xf) {
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

void orc_sincos(float xf, float* sin_out, float* cos_out) {
  const double x = (double)xf;
  const double k = rint(x * 0.63661977236758134308);
  double r = fma(-k, 1.57079632679489655800e+00, x);
  r = fma(-k, 6.12323399573676603587e-17, r);
  const double r2 = r * r;
  double sp = 1.0 / 6227020800.0;
  sp = fma(sp, r2, -1.0 / 39916800.0);
  sp = fma(sp, r2, 1.0 / 362880.0);
  sp = fma(sp, r2, -1.0 / 5040.0);
  sp = fma(sp, r2, 1.0 / 120.0);
  sp = fma(sp, r2, -1.0 / 6.0);
  sp = fma(sp * r2, r, r);
  double cp = -1.0 / 87178291200.0;
  cp = fma(cp, r2, 1.0 / 479001600.0);
  cp = fma(cp, r2, -1.0 / 3628800.0);
  cp = fma(cp, r2, 1.0 / 40320.0);
  cp = fma(cp, r2, -1.0 / 720.0);
  cp = fma(cp, r2, 1.0 / 24.0);
  cp = fma(cp, r2, -0.5);
  cp = fma(cp, r2, 1.0);
  const int quadrant = (int)((long long)k & 3);
  const double sv = (quadrant == 0) ? sp : (quadrant == 1) ? cp : (quadrant == 2) ? -sp : -cp;
  const double cv = (quadrant == 0) ? cp : (quadrant == 1) ? -sp : (quadrant == 2) ? -cp : sp;
  *sin_out = (float)sv;
  *cos_out = (float)cv;
}

/* atan as the backend defines it (se3_device.h: atan_det), operation by operation in binary64. */
float orc_atan(float xf) {
  double x = (double)xf;
  const int negative = x < 0.0;
  if (negative) x = -x;
  const int inverted = x > 1.0;
  if (inverted) x = 1.0 / x;
  const int shifted = x > 0.41421356237309503;
  if (shifted) x = (x - 1.0) / (x + 1.0);
  const double x2 = x * x;
  double p = 1.0 / 27.0;
  p = fma(p, -x2, 1.0 / 25.0);
  p = fma(p, -x2, 1.0 / 23.0);
  p = fma(p, -x2, 1.0 / 21.0);
  p = fma(p, -x2, 1.0 / 19.0);
  p = fma(p, -x2, 1.0 / 17.0);
  p = fma(p, -x2, 1.0 / 15.0);
  p = fma(p, -x2, 1.0 / 13.0);
  p = fma(p, -x2, 1.0 / 11.0);
  p = fma(p, -x2, 1.0 / 9.0);
  p = fma(p, -x2, 1.0 / 7.0);
  p = fma(p, -x2, 1.0 / 5.0);
  p = fma(p, -x2, 1.0 / 3.0);
  p = fma(p, -x2, 1.0);
  double r = p * x;
  if (shifted) r = 0.78539816339744830962 + r;
  if (inverted) r = 1.57079632679489661923 - r;
  return (float)(negative ? -r : r);
}

/* se3.hpp:293-313, so3.hpp:282-320 */
void orc_se3_exp(const float a[6], orc_se3* out) {
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = ox * ox + oy * oy + oz * oz;
  const float theta = sqrtf(theta_sq);
  const float half_theta = 0.5f * theta;
  float imag_factor, real_factor;
  if (theta < SOPHUS_EPS_F) {
    const float theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
    real_factor = 1.f - 0.5f * theta_sq + (float)(1.0 / 384.0) * theta_po4;
  } else {
    float sin_half_theta, cos_half_theta;
    orc_sincos(half_theta, &sin_half_theta, &cos_half_theta);
    imag_factor = sin_half_theta / theta;
    real_factor = cos_half_theta;
  }
  orc_se3 r;
  r.q[3] = real_factor; r.q[0] = imag_factor * ox; r.q[1] = imag_factor * oy; r.q[2] = imag_factor * oz;

  const float Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
  float Om2[9];
  mat3_mul(Om, Om, Om2);
  float V[9];
  if (theta < SOPHUS_EPS_F) {
    orc_se3_rotation(&r, V);
  } else {
    float sin_theta, cos_theta;
    orc_sincos(theta, &sin_theta, &cos_theta);
    const float c1 = (1.f - cos_theta) / theta_sq;
    const float c2 = (theta - sin_theta) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + c1 * Om[i] + c2 * Om2[i];
  }
  /* V * upsilon as a plain (unfused) matrix-vector product, like the rest of this SE(3) arithmetic (host-side Eigen /
   * Sophus code in the reference); m33_mul of oracle_internal.h is the fused chain of the device kernels and is NOT used here */
  r.t[0] = V[0] * a[0] + V[1] * a[1] + V[2] * a[2];
  r.t[1] = V[3] * a[0] + V[4] * a[1] + V[5] * a[2];
  r.t[2] = V[6] * a[0] + V[7] * a[1] + V[8] * a[2];
  *out = r;
}

/* se3.hpp:440-467, so3.hpp:421-465 */
void orc_se3_log(const orc_se3* T, float out[6]) {
  const float qx = T->q[0], qy = T->q[1], qz = T->q[2], w = T->q[3];
  const float squared_n = qx * qx + qy * qy + qz * qz;
  const float n = sqrtf(squared_n);
  float two_atan_nbyw_by_n;
  if (n < SOPHUS_EPS_F) {
    const float squared_w = w * w;
    two_atan_nbyw_by_n = 2.f / w - 2.f * squared_n / (w * squared_w);
  } else if (fabsf(w) < SOPHUS_EPS_F) {
    two_atan_nbyw_by_n = (w > 0.f) ? ((float)M_PI / n) : (-(float)M_PI / n);
  } else {
    two_atan_nbyw_by_n = 2.f * orc_atan(n / w) / n;
  }
  const float theta = two_atan_nbyw_by_n * n;
  const float ox = two_atan_nbyw_by_n * qx, oy = two_atan_nbyw_by_n * qy, oz = two_atan_nbyw_by_n * qz;
  const float Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
  float Om2[9];
  mat3_mul(Om, Om, Om2);
  float Vinv[9];
  float c;
  if (fabsf(theta) < SOPHUS_EPS_F) {
    c = (float)(1. / 12.);
  } else {
    const float half_theta = 0.5f * theta;
    float sin_half_theta, cos_half_theta;
    orc_sincos(half_theta, &sin_half_theta, &cos_half_theta);
    c = (1.f - theta * cos_half_theta / (2.f * sin_half_theta)) / (theta * theta);
  }
  for (int i = 0; i < 9; ++i) Vinv[i] = ((i % 4 == 0) ? 1.f : 0.f) - 0.5f * Om[i] + c * Om2[i];
  out[0] = Vinv[0] * T->t[0] + Vinv[1] * T->t[1] + Vinv[2] * T->t[2];
  out[1] = Vinv[3] * T->t[0] + Vinv[4] * T->t[1] + Vinv[5] * T->t[2];
  out[2] = Vinv[6] * T->t[0] + Vinv[7] * T->t[1] + Vinv[8] * T->t[2];
  out[3] = ox; out[4] = oy; out[5] = oz;
}

/* B/keyframe.h:160-165 with libvis/src/libvis/image_frame.h:84-94 (both directions cached) */
void orc_keyframe_set_global_T_frame(orc_keyframe* kf, const orc_se3* global_T_frame) {
  kf->global_T_frame = *global_T_frame;
  orc_se3 inv;
  orc_se3_inverse(global_T_frame, &inv);
  orc_se3_matrix3x4(&inv, kf->frame_T_global);
  orc_se3_rotation(global_T_frame, kf->global_R_frame);
}

/* ------------------------------------------------------------------------------------------
 * Packing helpers
 * ---------------------------------------------------------------------------------------- */
/* B/util_nvcc_only.cuh:66-84 */
static uint32_t small_float_to_ten_bit_signed(float value) {
  const int16_t v = (int16_t)(value * ((1 << 9) - 1) + ((value > 0) ? 0.5f : -0.5f));
  return 0x03ffu & (uint16_t)v;
}
static float ten_bit_signed_to_small_float(uint32_t value) {
  const uint16_t temp = (uint16_t)(((0x0200u & value) ? 0xfc00u : 0u) | (0x03ffu & value));
  int16_t s; memcpy(&s, &temp, 2);
  return s * (1.0f / ((1 << 9) - 1));
}
uint32_t orc_pack_normal10(float x, float y, float z) {
  return (small_float_to_ten_bit_signed(x) << 0) | (small_float_to_ten_bit_signed(y) << 10) |
         (small_float_to_ten_bit_signed(z) << 20);
}
/* B/util_nvcc_only.cuh:87-95 */
void orc_unpack_normal10(uint32_t value, float n[3]) {
  v3 v = v3_make(ten_bit_signed_to_small_float(value >> 0), ten_bit_signed_to_small_float(value >> 10),
                 ten_bit_signed_to_small_float(value >> 20));
  const float factor = 1.0f / v3_norm(v);
  n[0] = factor * v.x; n[1] = factor * v.y; n[2] = factor * v.z;
}
/* B/util.cuh:121-146 */
static int8_t small_float_to_eight_bit_signed(float value) {
  return (int8_t)(value * ((1 << 7) - 1) + ((value > 0) ? 0.5f : -0.5f));
}
uint16_t orc_pack_normal8(float x, float y) {
  return (uint16_t)(((uint16_t)(uint8_t)small_float_to_eight_bit_signed(x) << 0) |
                    ((uint16_t)(uint8_t)small_float_to_eight_bit_signed(y) << 8));
}
void orc_unpack_normal8(uint16_t value, float n[3]) {
  n[0] = (int8_t)(value & 0x00ff) * (1.0f / ((1 << 7) - 1));
  n[1] = (int8_t)((value & 0xff00) >> 8) * (1.0f / ((1 << 7) - 1));
  float z = 1 - n[0] * n[0] - n[1] * n[1];
  n[2] = -sqrtf((z > 0.f) ? z : 0.f);
}

/* IEEE binary16 <-> binary32, round-to-nearest-even (__float2half_rn / __half2float,
 * B/cuda_depth_processing.cu:355, B/kernel_create_surfels.cu:121). */
uint16_t orc_float_to_half(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x0200u : 0u));
  if (absx >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* rounds to inf (>= 65520) */
  if (absx < 0x33000001u) return (uint16_t)sign;              /* rounds to zero (<= 2^-25) */
  int32_t e = (int32_t)(absx >> 23) - 127;
  uint32_t m = (absx & 0x007fffffu) | 0x00800000u;
  int shift;
  uint32_t half_e;
  if (e < -14) { shift = 13 + (-14 - e); half_e = 0; } else { shift = 13; half_e = (uint32_t)(e + 15); }
  uint32_t q = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u);
  const uint32_t halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
  uint32_t r;
  if (half_e == 0) r = q; /* subnormal; q may carry into exponent 1 which is correct */
  else r = ((half_e - 1) << 10) + q; /* q includes the implicit bit (0x400) */
  return (uint16_t)(sign | r);
}
float orc_half_to_float(uint16_t h) {
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1fu;
  const uint32_t m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) { x = sign; }
    else {
      float f = (float)m * (1.0f / 16777216.0f); /* m * 2^-24 */
      memcpy(&x, &f, 4); x |= sign;
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f; memcpy(&f, &x, 4);
  return f;
}

/* Test hook: PixelCenterUnprojector::UnprojectPoint (B/surfel_projection.cuh:88-126) of pixel (x, y) at `depth`. */
void orc_unproject(const orc_camera* cam, int x, int y, float depth, float out[3]) {
  const unprojector u = make_unprojector(cam);
  const v3 p = unp_point(&u, x, y, depth);
  out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

/* B/util.cuh:62-69 */
float orc_raw_to_calibrated_depth(float a, float cfactor, float raw_to_float_depth, uint16_t measured_depth) {
  const float inv_depth = 1.0f / (raw_to_float_depth * measured_depth);
  return 1.f / mad(cfactor, orc_exp(-a * inv_depth), inv_depth);
}

/* ------------------------------------------------------------------------------------------
 * Colour sampler.  The reference reads luma through a cudaTextureObject_t with
 * cudaFilterModeLinear, cudaAddressModeClamp, cudaReadModeNormalizedFloat, unnormalised
 * coordinates (B/keyframe.cc:67-73): texel centres at integer + 0.5.  CUDA hardware uses 9-bit
 * fixed-point interpolation weights; this restatement uses exact binary32 weights (documented
 * delta).
 * ---------------------------------------------------------------------------------------- */
static inline float luma_texel(const uint8_t* rgba, int width, int height, int x, int y) {
  if (x < 0) x = 0; if (x > width - 1) x = width - 1;
  if (y < 0) y = 0; if (y > height - 1) y = height - 1;
  return (float)rgba[4 * ((size_t)y * width + x) + 3] * (1.0f / 255.0f);
}

float orc_sample_luma(const uint8_t* rgba, int width, int height, float x, float y) {
  float xb = x - 0.5f, yb = y - 0.5f;
  if (!(xb >= -1.f)) xb = -1.f; /* also catches NaN */
  if (xb > (float)width) xb = (float)width;
  if (!(yb >= -1.f)) yb = -1.f;
  if (yb > (float)height) yb = (float)height;
  const float fx = floorf(xb), fy = floorf(yb);
  const float a = xb - fx, b = yb - fy;
  const int ix = (int)fx, iy = (int)fy;
  const float tl = luma_texel(rgba, width, height, ix, iy);
  const float tr = luma_texel(rgba, width, height, ix + 1, iy);
  const float bl = luma_texel(rgba, width, height, ix, iy + 1);
  const float br = luma_texel(rgba, width, height, ix + 1, iy + 1);
  const float top = mad(a, tr - tl, tl);
  const float bot = mad(a, br - bl, bl);
  return mad(b, bot - top, top);
}

/* The same sampler on all four channels (tex2D<float4> of B/kernel_assign_colors.cu:80). */
void orc_sample_rgba(const uint8_t* rgba, int width, int height, float x, float y, float out[4]) {
  float xb = x - 0.5f, yb = y - 0.5f;
  if (!(xb >= -1.f)) xb = -1.f;
  if (xb > (float)width) xb = (float)width;
  if (!(yb >= -1.f)) yb = -1.f;
  if (yb > (float)height) yb = (float)height;
  const float fx = floorf(xb), fy = floorf(yb);
  const float a = xb - fx, b = yb - fy;
  int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
  if (x0 < 0) x0 = 0;
  if (x0 > width - 1) x0 = width - 1;
  if (x1 < 0) x1 = 0;
  if (x1 > width - 1) x1 = width - 1;
  if (y0 < 0) y0 = 0;
  if (y0 > height - 1) y0 = height - 1;
  if (y1 < 0) y1 = 0;
  if (y1 > height - 1) y1 = height - 1;
  for (int c = 0; c < 4; ++c) {
    const float tl = (float)rgba[4 * ((size_t)y0 * width + x0) + c] * (1.0f / 255.0f);
    const float tr = (float)rgba[4 * ((size_t)y0 * width + x1) + c] * (1.0f / 255.0f);
    const float bl = (float)rgba[4 * ((size_t)y1 * width + x0) + c] * (1.0f / 255.0f);
    const float br = (float)rgba[4 * ((size_t)y1 * width + x1) + c] * (1.0f / 255.0f);
    const float top = mad(a, tr - tl, tl);
    const float bot = mad(a, br - bl, bl);
    out[c] = mad(b, bot - top, top);
  }
}

/* B/cost_function.cuh:115-136 */
void orc_tangent_projections(v3 gp, v3 gn, float radius_sq, const float* F, const orc_camera* color_cam,
                             float t1_pxy[2], float t2_pxy[2]) {
  const float kTangentScaling = 2.0f;
  v3 t1 = v3_cross(gn, (fabsf(gn.x) > 0.9f) ? v3_make(0, 1, 0) : v3_make(1, 0, 0));
  t1 = v3_scale(kTangentScaling * sqrtf(radius_sq / fmaxf(1e-12f, v3_sqlen(t1))), t1);
  v3 l1 = m34_mul(F, v3_add(gp, t1));
  const float inv_z1 = 1.f / l1.z;
  t1_pxy[0] = mad(color_cam->fx, l1.x * inv_z1, color_cam->cx);
  t1_pxy[1] = mad(color_cam->fy, l1.y * inv_z1, color_cam->cy);
  v3 t2 = v3_cross(gn, t1);
  t2 = v3_scale(kTangentScaling * sqrtf(radius_sq / fmaxf(1e-12f, v3_sqlen(t2))), t2);
  v3 l2 = m34_mul(F, v3_add(gp, t2));
  const float inv_z2 = 1.f / l2.z;
  t2_pxy[0] = mad(color_cam->fx, l2.x * inv_z2, color_cam->cx);
  t2_pxy[1] = mad(color_cam->fy, l2.y * inv_z2, color_cam->cy);
}

/* B/cost_function.cuh:140-156 */
void orc_raw_descriptor_residual(const orc_keyframe* kf, const float c[2], const float t1[2], const float t2[2],
                                 float d1, float d2, float* r1, float* r2) {
  const float intensity = orc_sample_luma(kf->color, kf->color_width, kf->color_height, c[0], c[1]);
  const float t1_intensity = orc_sample_luma(kf->color, kf->color_width, kf->color_height, t1[0], t1[1]);
  const float t2_intensity = orc_sample_luma(kf->color, kf->color_width, kf->color_height, t2[0], t2[1]);
  *r1 = mad(180.f, t1_intensity - intensity, -d1);
  *r2 = mad(180.f, t2_intensity - intensity, -d2);
}

/* One sample point of B/cost_function.cuh:200-211.  The four taps sit exactly on texel centres,
 * so the bilinear sampler returns the (clamped) texel itself. */
static void point_gradient(const orc_keyframe* kf, float qx, float qy, float* dx, float* dy) {
  const int w = kf->color_width, h = kf->color_height;
  float mx = fmaxf(0.f, qx - 0.5f), my = fmaxf(0.f, qy - 0.5f);
  if (!(mx < (float)w)) mx = (float)w; /* keeps the int conversion defined; texel index clamps anyway */
  if (!(my < (float)h)) my = (float)h;
  const int ix = (int)mx, iy = (int)my;
  const float tx = fmaxf(0.f, fminf(1.f, qx - 0.5f - ix));
  const float ty = fmaxf(0.f, fminf(1.f, qy - 0.5f - iy));
  const float top_left = luma_texel(kf->color, w, h, ix, iy);
  const float top_right = luma_texel(kf->color, w, h, ix + 1, iy);
  const float bottom_left = luma_texel(kf->color, w, h, ix, iy + 1);
  const float bottom_right = luma_texel(kf->color, w, h, ix + 1, iy + 1);
  *dx = mad(bottom_right - bottom_left, ty, (top_right - top_left) * (1 - ty));
  *dy = mad(bottom_right - top_right, tx, (bottom_left - top_left) * (1 - tx));
}

/* B/cost_function.cuh:191-254 */
void orc_descriptor_gradient(const orc_keyframe* kf, const float c[2], const float t1[2], const float t2[2],
                             float g[4]) {
  float cdx, cdy, t1dx, t1dy, t2dx, t2dy;
  point_gradient(kf, c[0], c[1], &cdx, &cdy);
  point_gradient(kf, t1[0], t1[1], &t1dx, &t1dy);
  point_gradient(kf, t2[0], t2[1], &t2dx, &t2dy);
  g[0] = 180.f * (t1dx - cdx);
  g[1] = 180.f * (t1dy - cdy);
  g[2] = 180.f * (t2dx - cdx);
  g[3] = 180.f * (t2dy - cdy);
}

/* ------------------------------------------------------------------------------------------
 * Association: B/surfel_projection_nvcc_only.cuh:332-359 -> B/cuda_matrix.cuh:116-124,
 * B/util.cuh:102-118, B/surfel_projection_nvcc_only.cuh:48-127.  The order of the rejection
 * tests is the reference's.  FIX: a NaN position (deleted surfel) is
 * rejected explicitly instead of relying on float->int conversion of NaN.
 * ---------------------------------------------------------------------------------------- */
int orc_project_associate(const proj_params* p, uint32_t i, proj_result* r, int* free_space_violation) {
  if (free_space_violation) *free_space_violation = 0;
  if (i >= p->s->surfels_size) return 0;
  r->global_position = surfel_position(p->s, i);
  const float* F = p->F;
  const v3 g = r->global_position;
  r->local_position.z = mad(F[10], g.z, mad(F[9], g.y, mad(F[8], g.x, F[11])));
  if (!(r->local_position.z > 0.f)) return 0;
  r->local_position.x = mad(F[2], g.z, mad(F[1], g.y, mad(F[0], g.x, F[3])));
  r->local_position.y = mad(F[6], g.z, mad(F[5], g.y, mad(F[4], g.x, F[7])));

  /* one reciprocal shared by both coordinates (x * (1/z) instead of x / z; kernels: same) */
  const float inv_z = 1.f / r->local_position.z;
  r->pxx = mad(p->fx, r->local_position.x * inv_z, p->cx);
  r->pxy = mad(p->fy, r->local_position.y * inv_z, p->cy);
  if (!(r->pxx >= 0.f) || !(r->pxy >= 0.f) || !(r->pxx < (float)p->width) || !(r->pxy < (float)p->height)) return 0;
  r->px = (int)r->pxx;
  r->py = (int)r->pxy;
  if (r->px >= p->width || r->py >= p->height) return 0;

  const uint16_t measured_depth = p->depth[(size_t)r->py * p->width + r->px];
  if (measured_depth & ORC_INVALID_DEPTH_BIT) return 0;

  const float calibrated_depth = orc_raw_to_calibrated_depth(
      p->dp->a, cfactor_at(p->dp, r->px, r->py), p->dp->raw_to_float_depth, measured_depth);
  r->calibrated_depth = calibrated_depth;

  r->normal = surfel_normal(p->s, i);
  const v3 nl = m34_rotate(F, r->normal);

  const float stddev = depth_stddev(unp_nx(&p->unp, (float)r->px), unp_ny(&p->unp, (float)r->py),
                                    calibrated_depth, nl, p->dp->baseline_fx);
  const float thr = 10.f * stddev;
  if (free_space_violation) {
    const float depth_difference = calibrated_depth - r->local_position.z;
    if (depth_difference > thr) { *free_space_violation = 1; return 0; }
    else if (depth_difference < -thr) return 0;
  } else {
    if (fabsf(r->local_position.z - calibrated_depth) > thr) return 0;
  }

  /* The reference tests (1 / |p|) * dot(p, n) > 0 (B/surfel_projection_nvcc_only.cuh:107-111).  |p| > 0 here (z > 0), so the
   * sign is that of the dot product; the normalisation (a square root and a division per test) is left out on both sides,
   * oracle and kernels. */
  if (v3_dot(r->local_position, nl) > 0) return 0;

  float m[3];
  orc_unpack_normal8(p->normals[(size_t)r->py * p->width + r->px], m);
  if (v3_dot(nl, v3_make(m[0], m[1], m[2])) < ORC_COS_NORMAL_COMPAT) return 0;
  return 1;
}

/* ------------------------------------------------------------------------------------------
 * Pair evaluation for tests: residuals + Jacobians of one (surfel, keyframe) pair.
 * Pose Jacobians: B/kernel_opt_pose.cu:45-142; surfel Jacobians: B/kernel_opt_geometry.cu:119-230.
 * ---------------------------------------------------------------------------------------- */
int orc_evaluate_pair(const orc_camera* color_cam, const orc_camera* depth_cam, const orc_depth_params* dp,
                      const orc_keyframe* kf, const float F[12], const orc_surfels* s, uint32_t i,
                      orc_pair_eval* o) {
  memset(o, 0, sizeof(*o));
  proj_params p = make_proj_params(depth_cam, dp, s, kf, F);
  proj_result r;
  if (!orc_project_associate(&p, i, &r, NULL)) return 0;
  o->associated = 1; o->px = r.px; o->py = r.py; o->calibrated_depth = r.calibrated_depth;
  const v3 nl = m34_rotate(F, r.normal);
  const float inv_std = depth_inv_stddev(unp_nx(&p.unp, (float)r.px), unp_ny(&p.unp, (float)r.py),
                                         r.calibrated_depth, nl, dp->baseline_fx);
  const v3 u = unp_point(&p.unp, r.px, r.py, r.calibrated_depth);
  const float raw = inv_std * v3_dot(nl, v3_sub(u, r.local_position));
  o->depth_inv_stddev = inv_std;
  o->depth_residual = raw;
  o->depth_weight = depth_residual_weight(raw);
  jac_depth_pose(nl, u, inv_std, o->depth_jac_pose);
  o->depth_jac_surfel = -inv_std;

  depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
  float c[2];
  if (transform_depth_to_color(r.pxx, r.pxy, &d2c, &c[0], &c[1])) {
    o->color_valid = 1;
    float t1[2], t2[2];
    orc_tangent_projections(r.global_position, r.normal, srow(s, ORC_SURFEL_RADIUS_SQ)[i], F, color_cam, t1, t2);
    orc_raw_descriptor_residual(kf, c, t1, t2, srow(s, ORC_SURFEL_DESC1)[i], srow(s, ORC_SURFEL_DESC2)[i],
                                &o->desc_residual[0], &o->desc_residual[1]);
    o->desc_weight[0] = descriptor_residual_weight(o->desc_residual[0]);
    o->desc_weight[1] = descriptor_residual_weight(o->desc_residual[1]);
    orc_descriptor_gradient(kf, c, t1, t2, o->grad);
    const v3 ls = r.local_position;
    /* PixelCenterProjector fx, fy == camera fx, fy (B/surfel_projection.h:53-59) */
    for (int k = 0; k < 2; ++k) {
      jac_descriptor_pose(ls, o->grad[2 * k + 0] * color_cam->fx, o->grad[2 * k + 1] * color_cam->fy, o->desc_jac_pose[k]);
      o->desc_jac_surfel[k] = jac_descriptor_surfel(nl, ls, o->grad[2 * k + 0], o->grad[2 * k + 1], color_cam->fx, color_cam->fy);
    }
  }
  return 1;
}

/* orc_evaluate_pair for `count` surfel indices against one keyframe (OpenMP over the pairs); used by the sampled
 * per-pair parity test at full size. */
void orc_evaluate_pairs(const orc_camera* color_cam, const orc_camera* depth_cam, const orc_depth_params* dp,
                        const orc_keyframe* kf, const float F[12], const orc_surfels* s, const uint32_t* indices,
                        int count, orc_pair_eval* out) {
#pragma omp parallel for schedule(static)
  for (int t = 0; t < count; ++t) {
    if (indices[t] >= s->surfels_size) { memset(&out[t], 0, sizeof(out[t])); continue; }
    orc_evaluate_pair(color_cam, depth_cam, dp, kf, F, s, indices[t], &out[t]);
  }
}
