// se3_device.h -- binary32 SE(3) arithmetic laid out like Sophus' SE3f (unit quaternion x,y,z,w +
// translation = 7 floats), usable on host and device.  Formulas follow
// libvis/third_party/sophus/sophus/so3.hpp:215-233,282-320,421-465 and se3.hpp:127-130,203-207,
// 293-313,440-467 of the reference, and Eigen's quaternion product / toRotationMatrix /
// _transformVector, because the reference's pose update T <- T * exp(-x) runs through exactly
// these (B/direct_ba_alternating.cc:214).
#pragma once

#include <math.h>

// Usable from hipcc translation units (host + device) and from the plain-g++ host library.
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BAHIP_HD __host__ __device__ inline
#else
#define BAHIP_HD inline
#endif

namespace bahip {

constexpr float kSophusEpsilonF = 1e-5f;  // sophus/common.hpp:144-148

// sin and cos of a binary32 angle, defined by explicit binary64 operations (Cody-Waite reduction by pi/2, Taylor
// polynomials on |r| <= pi/4 evaluated with fused multiply-adds, result rounded to binary32): the exponential map runs on
// the device (pose_solve_kernel) and on the host, and math-library sinf / cosf differ between the two in the last bit.
// Defined this way every pose update is the same bits wherever it is computed (the oracle restates the same operations),
// which is what lets whole bundle-adjustment runs be compared bit for bit.  Error < 1 ulp of binary32 for |x| < 1e5.
BAHIP_HD void sincos_det(float xf, float* sin_out, float* cos_out) {
  const double x = (double)xf;
  const double k = __builtin_rint(x * 0.63661977236758134308);                 // nearest multiple of pi/2
  double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
  r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
  const double r2 = r * r;
  double sp = 1.0 / 6227020800.0;
  sp = __builtin_fma(sp, r2, -1.0 / 39916800.0);
  sp = __builtin_fma(sp, r2, 1.0 / 362880.0);
  sp = __builtin_fma(sp, r2, -1.0 / 5040.0);
  sp = __builtin_fma(sp, r2, 1.0 / 120.0);
  sp = __builtin_fma(sp, r2, -1.0 / 6.0);
  sp = __builtin_fma(sp * r2, r, r);
  double cp = -1.0 / 87178291200.0;
  cp = __builtin_fma(cp, r2, 1.0 / 479001600.0);
  cp = __builtin_fma(cp, r2, -1.0 / 3628800.0);
  cp = __builtin_fma(cp, r2, 1.0 / 40320.0);
  cp = __builtin_fma(cp, r2, -1.0 / 720.0);
  cp = __builtin_fma(cp, r2, 1.0 / 24.0);
  cp = __builtin_fma(cp, r2, -0.5);
  cp = __builtin_fma(cp, r2, 1.0);
  const int quadrant = (int)((long long)k & 3);
  const double sv = (quadrant == 0) ? sp : (quadrant == 1) ? cp : (quadrant == 2) ? -sp : -cp;
  const double cv = (quadrant == 0) ? cp : (quadrant == 1) ? -sp : (quadrant == 2) ? -cp : sp;
  *sin_out = (float)sv;
  *cos_out = (float)cv;
}

// atan of a binary32 argument by explicit binary64 operations (same reason as sincos_det: the logarithm map decides whether a
// keyframe "moved" -- on the device inside the BA loop and on the host -- and both must decide alike).  |x| > 1 is folded by
// atan(x) = pi/2 - atan(1/x), |x| > tan(pi/8) by atan(x) = pi/4 + atan((x - 1) / (x + 1)); odd Taylor polynomial to x^27 on
// the remaining |x| <= 0.4143 (truncation < 2e-12 relative), one final rounding.  Error < 1 ulp of binary32.
BAHIP_HD float atan_det(float xf) {
  double x = (double)xf;
  const bool negative = x < 0.0;
  if (negative) x = -x;
  const bool inverted = x > 1.0;
  if (inverted) x = 1.0 / x;
  const bool shifted = x > 0.41421356237309503;
  if (shifted) x = (x - 1.0) / (x + 1.0);
  const double x2 = x * x;
  double p = 1.0 / 27.0;
  p = __builtin_fma(p, -x2, 1.0 / 25.0);
  p = __builtin_fma(p, -x2, 1.0 / 23.0);
  p = __builtin_fma(p, -x2, 1.0 / 21.0);
  p = __builtin_fma(p, -x2, 1.0 / 19.0);
  p = __builtin_fma(p, -x2, 1.0 / 17.0);
  p = __builtin_fma(p, -x2, 1.0 / 15.0);
  p = __builtin_fma(p, -x2, 1.0 / 13.0);
  p = __builtin_fma(p, -x2, 1.0 / 11.0);
  p = __builtin_fma(p, -x2, 1.0 / 9.0);
  p = __builtin_fma(p, -x2, 1.0 / 7.0);
  p = __builtin_fma(p, -x2, 1.0 / 5.0);
  p = __builtin_fma(p, -x2, 1.0 / 3.0);
  p = __builtin_fma(p, -x2, 1.0);
  double r = p * x;
  if (shifted) r = 0.78539816339744830962 + r;
  if (inverted) r = 1.57079632679489661923 - r;
  return (float)(negative ? -r : r);
}

BAHIP_HD void quat_mul(const float* a, const float* b, float* o) {
  const float ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const float bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by + ay * bw + az * bx - ax * bz;
  o[2] = aw * bz + az * bw + ax * by - ay * bx;
}

BAHIP_HD void quat_rotate(const float* q, const float* v, float* o) {
  // uv = 2 * q.vec x v ; o = v + w*uv + q.vec x uv
  const float ux = 2.f * (q[1] * v[2] - q[2] * v[1]);
  const float uy = 2.f * (q[2] * v[0] - q[0] * v[2]);
  const float uz = 2.f * (q[0] * v[1] - q[1] * v[0]);
  o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

BAHIP_HD void se3_rotation(const float* T, float* r) {
  const float x = T[0], y = T[1], z = T[2], w = T[3];
  const float tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  r[0] = 1 - (tyy + tzz); r[1] = txy - twz;       r[2] = txz + twy;
  r[3] = txy + twz;       r[4] = 1 - (txx + tzz); r[5] = tyz - twx;
  r[6] = txz - twy;       r[7] = tyz + twx;       r[8] = 1 - (txx + tyy);
}

BAHIP_HD void se3_matrix3x4(const float* T, float* m) {
  float r[9];
  se3_rotation(T, r);
  m[0] = r[0]; m[1] = r[1]; m[2] = r[2];  m[3] = T[4];
  m[4] = r[3]; m[5] = r[4]; m[6] = r[5];  m[7] = T[5];
  m[8] = r[6]; m[9] = r[7]; m[10] = r[8]; m[11] = T[6];
}

BAHIP_HD void se3_mul(const float* a, const float* b, float* o) {
  float rt[3], q[4];
  quat_rotate(a, b + 4, rt);
  quat_mul(a, b, q);
  const float sq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sq != 1.0f) {
    const float f = 2.0f / (1.0f + sq);
    q[0] *= f; q[1] *= f; q[2] *= f; q[3] *= f;
  }
  o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
  o[4] = a[4] + rt[0]; o[5] = a[5] + rt[1]; o[6] = a[6] + rt[2];
}

BAHIP_HD void se3_inverse(const float* a, float* o) {
  float q[4] = {-a[0], -a[1], -a[2], a[3]};
  float nt[3] = {a[4] * -1.f, a[5] * -1.f, a[6] * -1.f};
  float t[3];
  quat_rotate(q, nt, t);
  o[0] = q[0]; o[1] = q[1]; o[2] = q[2]; o[3] = q[3];
  o[4] = t[0]; o[5] = t[1]; o[6] = t[2];
}

BAHIP_HD void mat3_mul(const float* a, const float* b, float* o) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      o[3 * i + j] = a[3 * i + 0] * b[0 + j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

BAHIP_HD void se3_exp(const float* a, float* o) {
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = ox * ox + oy * oy + oz * oz;
  const float theta = sqrtf(theta_sq);
  const float half_theta = 0.5f * theta;
  float imag_factor, real_factor;
  if (theta < kSophusEpsilonF) {
    const float theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5f - (float)(1.0 / 48.0) * theta_sq + (float)(1.0 / 3840.0) * theta_po4;
    real_factor = 1.f - 0.5f * theta_sq + (float)(1.0 / 384.0) * theta_po4;
  } else {
    float sin_half, cos_half;
    sincos_det(half_theta, &sin_half, &cos_half);
    imag_factor = sin_half / theta;
    real_factor = cos_half;
  }
  o[3] = real_factor; o[0] = imag_factor * ox; o[1] = imag_factor * oy; o[2] = imag_factor * oz;
  const float Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
  float Om2[9], V[9];
  mat3_mul(Om, Om, Om2);
  if (theta < kSophusEpsilonF) {
    se3_rotation(o, V);
  } else {
    float sin_theta, cos_theta;
    sincos_det(theta, &sin_theta, &cos_theta);
    const float c1 = (1.f - cos_theta) / theta_sq;
    const float c2 = (theta - sin_theta) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.f : 0.f) + c1 * Om[i] + c2 * Om2[i];
  }
  o[4] = V[0] * a[0] + V[1] * a[1] + V[2] * a[2];
  o[5] = V[3] * a[0] + V[4] * a[1] + V[5] * a[2];
  o[6] = V[6] * a[0] + V[7] * a[1] + V[8] * a[2];
}

BAHIP_HD void se3_log(const float* T, float* out) {
  const float qx = T[0], qy = T[1], qz = T[2], w = T[3];
  const float squared_n = qx * qx + qy * qy + qz * qz;
  const float n = sqrtf(squared_n);
  float two_atan_nbyw_by_n;
  if (n < kSophusEpsilonF) {
    const float squared_w = w * w;
    two_atan_nbyw_by_n = 2.f / w - 2.f * squared_n / (w * squared_w);
  } else if (fabsf(w) < kSophusEpsilonF) {
    two_atan_nbyw_by_n = (w > 0.f) ? (3.14159265358979323846f / n) : (-3.14159265358979323846f / n);
  } else {
    two_atan_nbyw_by_n = 2.f * atan_det(n / w) / n;
  }
  const float theta = two_atan_nbyw_by_n * n;
  const float ox = two_atan_nbyw_by_n * qx, oy = two_atan_nbyw_by_n * qy, oz = two_atan_nbyw_by_n * qz;
  const float Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
  float Om2[9];
  mat3_mul(Om, Om, Om2);
  float c;
  if (fabsf(theta) < kSophusEpsilonF) {
    c = (float)(1. / 12.);
  } else {
    const float half_theta = 0.5f * theta;
    float sin_half, cos_half;
    sincos_det(half_theta, &sin_half, &cos_half);
    c = (1.f - theta * cos_half / (2.f * sin_half)) / (theta * theta);
  }
  float Vinv[9];
  for (int i = 0; i < 9; ++i) Vinv[i] = ((i % 4 == 0) ? 1.f : 0.f) - 0.5f * Om[i] + c * Om2[i];
  out[0] = Vinv[0] * T[4] + Vinv[1] * T[5] + Vinv[2] * T[6];
  out[1] = Vinv[3] * T[4] + Vinv[4] * T[5] + Vinv[5] * T[6];
  out[2] = Vinv[6] * T[4] + Vinv[7] * T[5] + Vinv[8] * T[6];
  out[3] = ox; out[4] = oy; out[5] = oz;
}

}  // namespace bahip
