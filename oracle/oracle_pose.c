/* oracle_pose.c -- pose normal equations and the per-frame Gauss-Newton loop.
 * Test infrastructure only (see oracle.h). */
#include "oracle_internal.h"

/* One residual into (H, b): B/gauss_newton.cuh:59-92, row-major upper triangle.
 *
 * Definition of the sum (accumulate_double == 0).  The reference merges block partials with float atomics in arbitrary
 * order (B/gauss_newton.cuh:71,89), so its H, b are not reproducible run to run; SURVEY appendix B marks that FIX.  The
 * backend defines the sum so that it is reproducible AND independent of launch shape and multi-GPU sharding, and this
 * function restates exactly that definition:
 *   1. every surfel i accumulates its (up to three) residuals into 27 binary32 values with fused multiply-adds, depth
 *      residual first, then the two descriptor residuals (acc_residual below);
 *   2. the 64 surfels [64 t, 64 t + 64) of tile t are summed by a fixed binary tree (tile_tree: lane pairs at distance
 *      32, 16, 8, then 7 - i inside groups of 8, then distance 2, 1 -- the halving butterfly of wave_reduce.h);
 *   3. each of the 27 tile totals is converted to a multiple of 2^-32 (round to nearest even; exact for |total| >= 2^-9) and
 *      added as two 64-bit integer limbs -- the low 32 bits of that integer (weight 2^-32) and the rest (weight 1) --
 *      associative, hence order-free (hb_split below; round 2 used a single limb of weight 2^-16).  A total that is not
 *      finite or not below 2^52 in magnitude is not added and raises orc_pose_sum_invalid(), and so does a sum whose
 *      limb 1 ends at 2^62 or beyond;
 *   4. H, b = the carry-normalised limb pairs as binary64 (hb_value), rounded to binary32.
 * accumulate_double != 0 is the plain binary64 running sum in surfel order (an independent check of 1.-4.). */
static inline void acc_residual(float* acc, float raw, float w, const float* J) {
  int k = 0;
  for (int row = 0; row < 6; ++row) {
    const float wj = w * J[row];
    for (int col = row; col < 6; ++col, ++k) acc[k] = fmaf(wj, J[col], acc[k]);
  }
  const float wr = w * raw;
  for (int i = 0; i < 6; ++i) acc[21 + i] = fmaf(wr, J[i], acc[21 + i]);
}
static inline void add_residual_d(double* H, double* b, float raw, float w, const float* J) {
  int k = 0;
  for (int row = 0; row < 6; ++row)
    for (int col = row; col < 6; ++col) H[k++] += (double)(w * J[row] * J[col]);
  const float wr = w * raw;
  for (int i = 0; i < 6; ++i) b[i] += (double)(wr * J[i]);
}
/* Sum of 64 lane values in the order of the backend's wave64 reduction (wave_reduce.h: wave_reduce28). */
float orc_tile_tree_sum(const float x[64]) {
  float A[8];
  for (int c = 0; c < 8; ++c) {
    const float* v = x + c;   /* lanes c + 8 m, m = 0..7: v[8 m] */
    const float s04 = v[0] + v[32], s26 = v[16] + v[48], s15 = v[8] + v[40], s37 = v[24] + v[56];
    A[c] = (s04 + s26) + (s15 + s37);
  }
  const float B0 = A[0] + A[7], B1 = A[1] + A[6], B2 = A[2] + A[5], B3 = A[3] + A[4];
  return (B0 + B2) + (B1 + B3);
}
/* The backend's hb_split / hb_value (ba_device.h), restated. */
static int g_pose_sum_invalid = 0;
int orc_pose_sum_invalid(int reset) { const int v = g_pose_sum_invalid; if (reset) g_pose_sum_invalid = 0; return v; }
static int hb_split(float v, long long* lo_out, long long* hi_out) {
  uint32_t bits;
  memcpy(&bits, &v, sizeof(bits));
  uint32_t e = (bits >> 23) & 0xffu, m = bits & 0x7fffffu;
  long long lo = 0, hi = 0;
  if (e) m |= 0x800000u; else e = 1u;
  const int s = (int)e - 118;           /* |v| = m * 2^s in units of 2^-32 */
  if (s > 60) return 0;                 /* 2^52 and beyond, infinite, NaN */
  if (s >= 32) hi = (long long)((unsigned long long)m << (s - 32));
  else if (s >= 0) { const unsigned long long w = (unsigned long long)m << s; lo = (long long)(w & 0xffffffffull); hi = (long long)(w >> 32); }
  else if (s >= -25) {
    const int sh = -s;
    uint32_t q = m >> sh;
    const uint32_t rem = m & ((1u << sh) - 1u), half = 1u << (sh - 1);
    if (rem > half || (rem == half && (q & 1u))) ++q;
    lo = (long long)q;
  }
  if (bits >> 31) { lo = -lo; hi = -hi; }
  *lo_out = lo; *hi_out = hi;
  return 1;
}
/* test hook: the limb pair of one tile total; returns 0 if the value cannot be represented (not added, flag raised) */
int orc_pose_limbs(float v, long long out[2]) { out[0] = out[1] = 0; return hb_split(v, &out[0], &out[1]); }
static double hb_value(long long lo, long long hi) {
  hi += lo >> 32;
  lo &= 0xffffffffll;
  return (double)hi + (double)lo * 2.3283064365386963e-10;
}

/* B/kernel_opt_pose.cc:39-97, kernel B/kernel_opt_pose.cu:251-383. */
static uint32_t accumulate_pose_coeffs_impl(int use_depth, int use_desc, const orc_camera* color_cam,
                                            const orc_camera* depth_cam, const orc_depth_params* dp,
                                            const orc_keyframe* kf, const float F[12], const orc_surfels* s,
                                            float H[21], float b[6], float* residual_sum, int accumulate_double, long long* fixed_out /* [27][2] */) {
  proj_params p = make_proj_params(depth_cam, dp, s, kf, F);
  const depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
  long long fixed[27][2] = {{0}};
  double Hd[21] = {0}, bd[6] = {0};
  double cost = 0;
  uint32_t count = 0;
  for (uint32_t tile = 0; tile < s->surfels_size; tile += 64) {
    float lanes[27][64];
    int any = 0;
    memset(lanes, 0, sizeof(lanes));
    for (uint32_t lane = 0; lane < 64 && tile + lane < s->surfels_size; ++lane) {
      const uint32_t i = tile + lane;
      proj_result r;
      if (!orc_project_associate(&p, i, &r, NULL)) continue;
      ++count;
      any = 1;
      float acc[27] = {0};
      float J[6], raw;
      if (use_depth) {
        const v3 nl = m34_rotate(F, r.normal);
        const float inv_std = depth_inv_stddev(unp_nx(&p.unp, (float)r.px), unp_ny(&p.unp, (float)r.py),
                                               r.calibrated_depth, nl, dp->baseline_fx);
        const v3 u = unp_point(&p.unp, r.px, r.py, r.calibrated_depth);
        raw = inv_std * v3_dot(nl, v3_sub(u, r.local_position));
        jac_depth_pose(nl, u, inv_std, J);
        const float w = depth_residual_weight(raw);
        if (accumulate_double) add_residual_d(Hd, bd, raw, w, J); else acc_residual(acc, raw, w, J);
        cost += weighted_depth_residual(raw);
      }
      float c[2];
      /* B/kernel_opt_pose.cu:303-353: if the colour-pixel transform fails, nothing is added. */
      if (use_desc && transform_depth_to_color(r.pxx, r.pxy, &d2c, &c[0], &c[1])) {
        float t1[2], t2[2], raw1, raw2, g[4];
        orc_tangent_projections(r.global_position, r.normal, srow(s, ORC_SURFEL_RADIUS_SQ)[i], F, color_cam, t1, t2);
        orc_raw_descriptor_residual(kf, c, t1, t2, srow(s, ORC_SURFEL_DESC1)[i], srow(s, ORC_SURFEL_DESC2)[i], &raw1, &raw2);
        orc_descriptor_gradient(kf, c, t1, t2, g);
        const v3 ls = r.local_position;
        for (int k = 0; k < 2; ++k) {
          const float gx = g[2 * k + 0] * color_cam->fx;
          const float gy = g[2 * k + 1] * color_cam->fy;
          jac_descriptor_pose(ls, gx, gy, J);
          raw = k ? raw2 : raw1;
          const float w = descriptor_residual_weight(raw);
          if (accumulate_double) add_residual_d(Hd, bd, raw, w, J); else acc_residual(acc, raw, w, J);
        }
        cost += weighted_descriptor_residual(raw1);
      }
      for (int q = 0; q < 27; ++q) lanes[q][lane] = acc[q];
    }
    if (any && !accumulate_double)
      for (int q = 0; q < 27; ++q) {
        long long lo, hi;
        if (hb_split(orc_tile_tree_sum(lanes[q]), &lo, &hi)) { fixed[q][0] += lo; fixed[q][1] += hi; }
        else g_pose_sum_invalid = 1;
      }
  }
  if (!accumulate_double)
    for (int k = 0; k < 27; ++k) if (fixed[k][1] >= (1ll << 62) || fixed[k][1] <= -(1ll << 62)) g_pose_sum_invalid = 1;
  for (int k = 0; k < 21; ++k) H[k] = accumulate_double ? (float)Hd[k] : (float)hb_value(fixed[k][0], fixed[k][1]);
  for (int k = 0; k < 6; ++k) b[k] = accumulate_double ? (float)bd[k] : (float)hb_value(fixed[21 + k][0], fixed[21 + k][1]);
  if (residual_sum) *residual_sum = (float)cost;
  if (fixed_out) memcpy(fixed_out, fixed, sizeof(fixed));
  return count;
}

uint32_t orc_accumulate_pose_coeffs(int use_depth, int use_desc, const orc_camera* color_cam,
                                    const orc_camera* depth_cam, const orc_depth_params* dp,
                                    const orc_keyframe* kf, const float F[12], const orc_surfels* s,
                                    float H[21], float b[6], float* residual_sum, int accumulate_double) {
  return accumulate_pose_coeffs_impl(use_depth, use_desc, color_cam, depth_cam, dp, kf, F, s, H, b, residual_sum, accumulate_double, NULL);
}

/* The fixed-point limb pairs themselves, [27][2] (what the ranks of a surfel-sharded run exchange: an integer sum over shards
 * made of whole 64-surfel tiles IS the unsharded total), and their value. */
double orc_pose_limbs_value(long long lo, long long hi) { return hb_value(lo, hi); }
uint32_t orc_accumulate_pose_coeffs_fixed(int use_depth, int use_desc, const orc_camera* color_cam,
                                          const orc_camera* depth_cam, const orc_depth_params* dp,
                                          const orc_keyframe* kf, const float F[12], const orc_surfels* s, long long fixed[54]) {
  float H[21], b[6];
  return accumulate_pose_coeffs_impl(use_depth, use_desc, color_cam, depth_cam, dp, kf, F, s, H, b, NULL, 0, fixed);
}

/* B/convergence_analysis.h:43-51 */
int orc_is_scale1_pose_converged(const float x[6]) {
  const float translation_threshold = 1e-06f;
  const float rotation_threshold = 1e-07f;
  float sq = 0.f;
  for (int i = 0; i < 3; ++i) sq += x[i] * x[i];
  const float f = translation_threshold / rotation_threshold;
  for (int i = 3; i < 6; ++i) { const float v = x[i] * f; sq += v * v; }
  return sq < translation_threshold;
}

/* Eigen::LDLT (symmetric diagonal pivoting, lower) + solve with the pseudo-inverse rule on D
 * (Eigen/src/Cholesky/LDLT.h: |D_ii| <= numeric_limits<double>::min() -> 0).  n <= 8. */
void orc_ldlt_solve(int n, const double* Hfull, const double* b, double* x) {
  double A[64];
  int perm[8];
  for (int i = 0; i < n * n; ++i) A[i] = Hfull[i];
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    /* pivot: largest |diagonal| in the remaining block */
    int piv = k; double best = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) if (fabs(A[i * n + i]) > best) { best = fabs(A[i * n + i]); piv = i; }
    if (piv != k) {
      for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = t; }
      for (int j = 0; j < n; ++j) { double t = A[j * n + k]; A[j * n + k] = A[j * n + piv]; A[j * n + piv] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    const double d = A[k * n + k];
    if (fabs(d) > 2.2250738585072014e-308) {
      for (int i = k + 1; i < n; ++i) A[i * n + k] /= d;
      for (int i = k + 1; i < n; ++i)
        for (int j = k + 1; j <= i; ++j) {
          A[i * n + j] -= A[i * n + k] * d * A[j * n + k];
          A[j * n + i] = A[i * n + j];
        }
    } else {
      for (int i = k + 1; i < n; ++i) A[i * n + k] = 0;
    }
  }
  double y[8];
  for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
  for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) y[i] -= A[i * n + j] * y[j];
  for (int i = 0; i < n; ++i) { const double d = A[i * n + i]; y[i] = (fabs(d) > 2.2250738585072014e-308) ? y[i] / d : 0.0; }
  for (int i = n - 1; i >= 0; --i) for (int j = i + 1; j < n; ++j) y[i] -= A[j * n + i] * y[j];
  for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
}

/* B/direct_ba_alternating.cc:42-283 */
int orc_estimate_frame_pose(int use_depth, int use_desc, const orc_camera* color_cam,
                            const orc_camera* depth_cam, const orc_depth_params* dp,
                            const orc_keyframe* kf, const orc_se3* init, const orc_surfels* s,
                            orc_se3* out, int* converged_out) {
  orc_se3 est = *init;
  int converged = 0, iteration;
  for (iteration = 0; iteration < 30; ++iteration) {
    orc_se3 inv;
    orc_se3_inverse(&est, &inv);
    float F[12];
    orc_se3_matrix3x4(&inv, F);
    float H21[21], b6[6];
    if (s->surfels_size == 0) {
      memset(H21, 0, sizeof(H21)); memset(b6, 0, sizeof(b6));
    } else {
      orc_accumulate_pose_coeffs(use_depth, use_desc, color_cam, depth_cam, dp, kf, F, s, H21, b6, NULL, 0);
    }
    double H[36], bd[6], xd[6];
    int k = 0;
    for (int row = 0; row < 6; ++row)
      for (int col = row; col < 6; ++col) { H[row * 6 + col] = H21[k]; H[col * 6 + row] = H21[k]; ++k; }
    for (int i = 0; i < 6; ++i) bd[i] = b6[i];
    orc_ldlt_solve(6, H, bd, xd);
    float x[6], mx[6];
    for (int i = 0; i < 6; ++i) { x[i] = (float)xd[i]; mx[i] = -1.f * x[i]; }
    orc_se3 upd, next;
    orc_se3_exp(mx, &upd);
    orc_se3_mul(&est, &upd, &next);
    est = next;
    converged = orc_is_scale1_pose_converged(x);
    if (converged) { ++iteration; break; }
  }
  *out = est;
  if (converged_out) *converged_out = converged;
  return iteration;
}

/* ---- the Jacobian helpers of oracle_internal.h, exported for tests/test_cpu_golden_jacobians.py ---- */
void orc_jac_depth_pose(const float nl[3], const float u[3], float inv_std, float J[6]) {
  jac_depth_pose(v3_make(nl[0], nl[1], nl[2]), v3_make(u[0], u[1], u[2]), inv_std, J);
}
void orc_jac_descriptor_pose(const float ls[3], float gx, float gy, float J[6]) {
  jac_descriptor_pose(v3_make(ls[0], ls[1], ls[2]), gx, gy, J);
}
float orc_jac_descriptor_surfel(const float rn[3], const float lp[3], float gx, float gy, float cfx, float cfy) {
  return jac_descriptor_surfel(v3_make(rn[0], rn[1], rn[2]), v3_make(lp[0], lp[1], lp[2]), gx, gy, cfx, cfy);
}
void orc_jac_depth_intrinsics(int px, int py, float depth, float inv_std, float n_dot_Frow0, float n_dot_Frow1, float dot, float cfactor,
                              float raw_inv_depth, float exp_inv_depth, float corrected_inv_depth, float J[6]) {
  jac_depth_intrinsics(px, py, depth, inv_std, n_dot_Frow0, n_dot_Frow1, dot, cfactor, raw_inv_depth, exp_inv_depth, corrected_inv_depth, J);
}
void orc_jac_descriptor_color_intrinsics(float gx, float gy, float nx, float ny, float J[4]) {
  jac_descriptor_color_intrinsics(gx, gy, nx, ny, J);
}
