/* oracle_internal.h -- shared internals of the CPU oracle (test infrastructure; see oracle.h). */
#ifndef BADSLAM_ORACLE_INTERNAL_H_
#define BADSLAM_ORACLE_INTERNAL_H_

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef struct { float x, y, z; } v3;

/* ---- exact sums of binary32 values (oracle_exact.c) ---- */
#define ORC_EXACT_LIMBS 9
typedef struct { long long limb[ORC_EXACT_LIMBS]; } orc_exact;   /* limb j: weight 2^(32 j - 149) */
void orc_exact_add(orc_exact* cell, float v, int* invalid);      /* atomic on the limbs: callable from OpenMP regions */
void orc_exact_merge(orc_exact* dst, const orc_exact* src);
double orc_exact_value(const orc_exact* cell);                   /* the exact sum rounded to binary64, ties to even */

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(float m, v3 b) { return v3_make(m * b.x, m * b.y, m * b.z); }
/* B/cuda_util.cuh:47-92 */
/* Sums of products are explicit fused multiply-add chains, the same chains as the device helpers (badslam_amd/csrc/
 * ba_device.h); -ffp-contract=off keeps the compilers from fusing anything else, so both sides round alike.  (The reference's
 * nvcc build contracts a*b+c into fma at the compiler's discretion, so neither spelling is "the" reference rounding.) */
static inline float mad(float a, float b, float c) { return fmaf(a, b, c); }
static inline float v3_dot(v3 a, v3 b) { return mad(a.z, b.z, mad(a.y, b.y, a.x * b.x)); }
static inline float v3_sqlen(v3 a) { return mad(a.z, a.z, mad(a.y, a.y, a.x * a.x)); }
static inline float v3_norm(v3 a) { return sqrtf(v3_sqlen(a)); }
static inline v3 v3_cross(v3 a, v3 b) {
  return v3_make(a.y * b.z - b.y * a.z, b.x * a.z - a.x * b.z, a.x * b.y - b.x * a.y);
}

/* B/cuda_matrix.cuh:100-141; m is row-major 3x4 */
static inline v3 m34_mul(const float* m, v3 p) {
  return v3_make(mad(m[2], p.z, mad(m[1], p.y, mad(m[0], p.x, m[3]))),
                 mad(m[6], p.z, mad(m[5], p.y, mad(m[4], p.x, m[7]))),
                 mad(m[10], p.z, mad(m[9], p.y, mad(m[8], p.x, m[11]))));
}
static inline v3 m34_rotate(const float* m, v3 p) {
  return v3_make(mad(m[2], p.z, mad(m[1], p.y, m[0] * p.x)),
                 mad(m[6], p.z, mad(m[5], p.y, m[4] * p.x)),
                 mad(m[10], p.z, mad(m[9], p.y, m[8] * p.x)));
}
static inline v3 m33_mul(const float* m, v3 p) {
  return v3_make(mad(m[2], p.z, mad(m[1], p.y, m[0] * p.x)),
                 mad(m[5], p.z, mad(m[4], p.y, m[3] * p.x)),
                 mad(m[8], p.z, mad(m[7], p.y, m[6] * p.x)));
}

/* PixelCenterUnprojector, B/surfel_projection.cuh:88-126, built as in B/surfel_projection.h:61-71 */
typedef struct { float fx_inv, fy_inv, cx_inv, cy_inv; } unprojector;
static inline unprojector make_unprojector(const orc_camera* c) {
  unprojector u;
  u.fx_inv = 1.0f / c->fx;
  u.fy_inv = 1.0f / c->fy;
  const float cx_pixel_center = c->cx - 0.5f;
  const float cy_pixel_center = c->cy - 0.5f;
  u.cx_inv = -cx_pixel_center * u.fx_inv;
  u.cy_inv = -cy_pixel_center * u.fy_inv;
  return u;
}
static inline float unp_nx(const unprojector* u, float px) { return mad(u->fx_inv, px, u->cx_inv); }
static inline float unp_ny(const unprojector* u, float py) { return mad(u->fy_inv, py, u->cy_inv); }
static inline v3 unp_point(const unprojector* u, int x, int y, float depth) {
  return v3_make(depth * mad(u->fx_inv, (float)x, u->cx_inv), depth * mad(u->fy_inv, (float)y, u->cy_inv), depth);
}

/* DepthToColorPixelCorner, B/surfel_projection.h:105-124 */
typedef struct { float fx, fy, cx, cy; int width, height; } depth_to_color;
static inline depth_to_color make_depth_to_color(const orc_camera* depth_cam, const orc_camera* color_cam) {
  depth_to_color r;
  r.width = color_cam->width;
  r.height = color_cam->height;
  r.fx = color_cam->fx / depth_cam->fx;
  r.cx = -1 * color_cam->fx * depth_cam->cx / depth_cam->fx + color_cam->cx;
  r.fy = color_cam->fy / depth_cam->fy;
  r.cy = -1 * color_cam->fy * depth_cam->cy / depth_cam->fy + color_cam->cy;
  return r;
}
/* B/surfel_projection.cuh:194-207 */
static inline int transform_depth_to_color(float pxx, float pxy, const depth_to_color* d, float* cx, float* cy) {
  *cx = mad(d->fx, pxx, d->cx);
  *cy = mad(d->fy, pxy, d->cy);
  return *cx >= 0 && *cy >= 0 && (int)(*cx) < d->width && (int)(*cy) < d->height;
}

/* Surfel accessors, B/util_nvcc_only.cuh:51-115 */
static inline float* srow(const orc_surfels* s, int row) { return s->data + (size_t)row * s->capacity; }
static inline v3 surfel_position(const orc_surfels* s, uint32_t i) {
  return v3_make(srow(s, ORC_SURFEL_X)[i], srow(s, ORC_SURFEL_Y)[i], srow(s, ORC_SURFEL_Z)[i]);
}
static inline void surfel_set_position(orc_surfels* s, uint32_t i, v3 p) {
  srow(s, ORC_SURFEL_X)[i] = p.x; srow(s, ORC_SURFEL_Y)[i] = p.y; srow(s, ORC_SURFEL_Z)[i] = p.z;
}
static inline v3 surfel_normal(const orc_surfels* s, uint32_t i) {
  uint32_t v; memcpy(&v, &srow(s, ORC_SURFEL_NORMAL)[i], 4);
  float n[3]; orc_unpack_normal10(v, n);
  return v3_make(n[0], n[1], n[2]);
}
static inline void surfel_set_normal(orc_surfels* s, uint32_t i, v3 n) {
  uint32_t v = orc_pack_normal10(n.x, n.y, n.z);
  memcpy(&srow(s, ORC_SURFEL_NORMAL)[i], &v, 4);
}

/* Everything a per-keyframe surfel sweep needs (SurfelProjectionParameters, B/surfel_projection.cuh:152-180) */
typedef struct {
  const orc_surfels* s;
  const uint16_t* depth;
  const uint16_t* normals;
  int width, height;
  const orc_depth_params* dp;
  float fx, fy, cx, cy;         /* PixelCornerProjector of the depth camera */
  unprojector unp;
  const float* F;               /* frame_T_global, 3x4 row-major */
} proj_params;

static inline proj_params make_proj_params(const orc_camera* depth_cam, const orc_depth_params* dp,
                                           const orc_surfels* s, const orc_keyframe* kf, const float* F) {
  proj_params p;
  p.s = s; p.depth = kf->depth; p.normals = kf->normals; p.width = kf->width; p.height = kf->height;
  p.dp = dp; p.fx = depth_cam->fx; p.fy = depth_cam->fy; p.cx = depth_cam->cx; p.cy = depth_cam->cy;
  p.unp = make_unprojector(depth_cam); p.F = F;
  return p;
}

/* SurfelProjectionResult6, B/surfel_projection_nvcc_only.cuh:236-256 */
typedef struct {
  v3 global_position, local_position, normal /* global */;
  float calibrated_depth;
  int px, py;
  float pxx, pxy;   /* float pixel coordinates, pixel-corner convention */
} proj_result;

/* B/surfel_projection_nvcc_only.cuh:332-359 (+:48-127).  free_space_violation may be NULL. */
int orc_project_associate(const proj_params* p, uint32_t surfel_index, proj_result* r, int* free_space_violation);

/* Robust weights, B/robust_weighting.cuh:39-86; B/cost_function.cuh:44-52,95-98,105-109,177-185 */
static inline float tukey_weight(float r, float k) {
  if (fabsf(r) < k) { const float q = r * (1.f / k); const float t = 1.f - q * q; return t * t; }
  return 0.f;
}
static inline float tukey_residual(float r, float k) {
  if (fabsf(r) < k) { const float q = r * (1.f / k); const float t = 1.f - q * q; return (1 / 6.f) * k * k * (1 - t * t * t); }
  return (1 / 6.f) * k * k;
}
static inline float huber_weight(float r, float k) { const float a = fabsf(r); return (a < k) ? 1.f : (k / a); }
static inline float huber_residual(float r, float k) {
  const float a = fabsf(r); return (a < k) ? (0.5f * r * r) : (k * (a - 0.5f * k));
}
static inline float depth_residual_weight(float r) { return 1.f * tukey_weight(r, 1.f * 10.f); }
static inline float weighted_depth_residual(float r) { return 1.f * tukey_residual(r, 1.f * 10.f); }
static inline float descriptor_residual_weight(float r) { return 1.f * 1e-2f * huber_weight(r, 10.f); }
static inline float weighted_descriptor_residual(float r) { return 1.f * 1e-2f * huber_residual(r, 10.f); }

/* B/cost_function.cuh:81-88 */
static inline float depth_stddev(float nx, float ny, float depth, v3 nl, float baseline_fx) {
  return (0.1f * fabsf(mad(nl.y, ny, mad(nl.x, nx, nl.z))) * (depth * depth)) * (1.f / baseline_fx);
}
static inline float depth_inv_stddev(float nx, float ny, float depth, v3 nl, float baseline_fx) {
  return baseline_fx / (0.1f * fabsf(mad(nl.y, ny, mad(nl.x, nx, nl.z))) * (depth * depth));
}

/* B/cost_function.cuh:115-136 */
void orc_tangent_projections(v3 gp, v3 gn, float radius_sq, const float* F, const orc_camera* color_cam,
                             float t1[2], float t2[2]);
/* B/cost_function.cuh:140-156 */
void orc_raw_descriptor_residual(const orc_keyframe* kf, const float c[2], const float t1[2], const float t2[2],
                                 float d1, float d2, float* r1, float* r2);
void orc_sample_rgba(const uint8_t* rgba, int width, int height, float x, float y, float out[4]);
/* B/cost_function.cuh:191-254 */
void orc_descriptor_gradient(const orc_keyframe* kf, const float c[2], const float t1[2], const float t2[2],
                             float g[4]);

/* ---- residual Jacobians in isolation.  Used by the sweeps below and exported (orc_jac_*) so that they can be checked
 * against golden vectors generated from the reference's own derivation script
 * (applications/badslam/scripts/jacobians_derivation.py -> tests/golden/jacobians.json). ---- */
/* B/kernel_opt_pose.cu:88-93: d(depth residual)/d(pose delta); nl = surfel normal and u = unprojected measurement, both in
 * the keyframe frame. */
static inline void jac_depth_pose(v3 nl, v3 u, float inv_std, float J[6]) {
  J[0] = inv_std * nl.x;
  J[1] = inv_std * nl.y;
  J[2] = inv_std * nl.z;
  J[3] = inv_std * mad(nl.z, u.y, -(nl.y * u.z));
  J[4] = inv_std * mad(nl.x, u.z, -(nl.z * u.x));
  J[5] = inv_std * mad(nl.y, u.x, -(nl.x * u.y));
}
/* B/kernel_opt_pose.cu:126-141: d(descriptor residual)/d(pose delta); ls = surfel position in the keyframe frame,
 * gx, gy = image gradient of the residual times fx, fy of the colour camera. */
static inline void jac_descriptor_pose(v3 ls, float gx, float gy, float J[6]) {
  const float inv_z = 1.f / ls.z, z_sq = ls.z * ls.z, inv_z_sq = inv_z * inv_z, xy = ls.x * ls.y;
  J[0] = -gx * inv_z;
  J[1] = -gy * inv_z;
  J[2] = mad(ls.y, gy, ls.x * gx) * inv_z_sq;
  J[3] = mad(mad(ls.y, ls.y, z_sq), gy, xy * gx) * inv_z_sq;
  J[4] = -mad(mad(ls.x, ls.x, z_sq), gx, xy * gy) * inv_z_sq;
  J[5] = -mad(ls.x, gy, -(ls.y * gx)) * inv_z;
}
/* B/kernel_opt_geometry.cu:170-190: d(descriptor residual)/d(surfel offset along its normal); rn = surfel normal and lp =
 * surfel position in the keyframe frame, g = image gradient of the residual (per pixel). */
static inline float jac_descriptor_surfel(v3 rn, v3 lp, float gx, float gy, float cfx, float cfy) {
  const float term1 = -cfx * mad(rn.x, lp.z, -(rn.z * lp.x));
  const float term2 = -cfy * mad(rn.y, lp.z, -(rn.z * lp.y));
  const float inv_z = 1.f / lp.z, term3 = inv_z * inv_z;
  return -mad(gy, term2, gx * term1) * term3;
}
/* B/kernel_opt_intrinsics.cu:107-140: d(depth residual)/d(fx_inv, fy_inv, cx_inv, cy_inv, a, cfactor).  n_dot_Frow0/1 = global
 * surfel normal . first / second row of frame_R_global (= nl.x, nl.y up to rounding), dot = (nx, ny, 1) . nl. */
static inline void jac_depth_intrinsics(int px, int py, float depth, float inv_std, float n_dot_Frow0, float n_dot_Frow1, float dot,
                                        float cfactor, float raw_inv_depth, float exp_inv_depth, float corrected_inv_depth, float J[6]) {
  const float jac_base = inv_std * dot * exp_inv_depth / (corrected_inv_depth * corrected_inv_depth);
  J[2] = inv_std * depth * n_dot_Frow0;
  J[3] = inv_std * depth * n_dot_Frow1;
  J[0] = px * J[2];
  J[1] = py * J[3];
  J[4] = cfactor * raw_inv_depth * jac_base;
  J[5] = -jac_base;
}
/* B/kernel_opt_intrinsics.cu:176-199: d(descriptor residual)/d(fx, fy, cx, cy) of the colour camera; g = image gradient of the
 * residual, (nx, ny) = normalised image coordinates of the pixel. */
static inline void jac_descriptor_color_intrinsics(float gx, float gy, float nx, float ny, float J[4]) {
  J[0] = gx * nx; J[1] = gy * ny; J[2] = gx; J[3] = gy;
}

static inline float cfactor_at(const orc_depth_params* dp, int px, int py) {
  return dp->cfactor[(size_t)(py / dp->cell) * dp->cf_width + (px / dp->cell)];
}

#endif
