/* oracle_geometry.c -- surfel activation and the per-surfel geometry step.
 * Test infrastructure only (see oracle.h). */
#include "oracle_internal.h"

/* B/kernel_surfel_activation.cc:39-67, B/kernel_surfel_activation.cu:38-94 */
void orc_update_surfel_activation(const orc_camera* depth_cam, const orc_depth_params* dp,
                                  orc_keyframe* const* kfs, int num_kfs, uint32_t surfels_size,
                                  orc_surfels* s) {
  if (surfels_size == 0) return;
  orc_surfels view = *s;
  view.surfels_size = surfels_size;
#pragma omp parallel for schedule(static)
  for (uint32_t i = 0; i < surfels_size; ++i) {
    uint8_t flag = s->active[i] & (uint8_t)~ORC_SURFEL_ACTIVE_FLAG;
    for (int k = 0; k < num_kfs && !(flag & ORC_SURFEL_ACTIVE_FLAG); ++k) {
      const orc_keyframe* kf = kfs[k];
      if (!kf || kf->activation != ORC_KF_ACTIVE) continue;
      proj_params p = make_proj_params(depth_cam, dp, &view, kf, kf->frame_T_global);
      proj_result r;
      if (orc_project_associate(&p, i, &r, NULL)) flag = ORC_SURFEL_ACTIVE_FLAG;
    }
    s->active[i] = flag;
  }
}

/* DirectBA::AssignColors: B/kernel_assign_colors.cc:39-80, B/kernel_assign_colors.cu:41-125.  Every keyframe a surfel is
 * associated with (active or not) contributes the bilinear RGBA sample at the surfel's colour pixel, in keyframe order; the
 * mean becomes the surfel colour (rounded to nearest by + 0.5 and truncation).  Surfels seen by no keyframe keep theirs.  The
 * reference parks count and sums in accumulator rows 0-4 (scratch by contract); they stay untouched here. */
void orc_assign_colors(const orc_camera* color_cam, const orc_camera* depth_cam, const orc_depth_params* dp,
                       orc_keyframe* const* kfs, int num_kfs, orc_surfels* s) {
  const depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
#pragma omp parallel for schedule(static)
  for (uint32_t i = 0; i < s->surfels_size; ++i) {
    float count = 0.f, sum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < num_kfs; ++k) {
      const orc_keyframe* kf = kfs[k];
      if (!kf) continue;
      proj_params p = make_proj_params(depth_cam, dp, s, kf, kf->frame_T_global);
      proj_result r;
      if (!orc_project_associate(&p, i, &r, NULL)) continue;
      float cx, cy;
      if (!transform_depth_to_color(r.pxx, r.pxy, &d2c, &cx, &cy)) continue;
      float c[4];
      orc_sample_rgba(kf->color, kf->color_width, kf->color_height, cx, cy, c);
      count += 1.f;
      for (int q = 0; q < 4; ++q) sum[q] += c[q];
    }
    if (count > 0) {
      uint8_t* out = (uint8_t*)&srow(s, ORC_SURFEL_COLOR)[i];
      for (int q = 0; q < 4; ++q) out[q] = (uint8_t)(255.f * sum[q] / count + 0.5f);
    }
  }
}

/* Per-surfel sums over keyframes.  The reference adds the keyframes' contributions in keyframe order (one kernel launch
 * per keyframe).  The HIP path defines the sums as four interleaved partial sums - partial j takes the keyframes whose
 * index among the non-deleted keyframes is congruent to j modulo 4, in ascending order - combined as
 * ((p0 + p1) + p2) + p3 (kernels_surfel.hip: tile_sums).  The oracle follows that definition so
 * that the comparison can be bit-exact; in exact arithmetic it is the reference's sum. */
/* The class count is part of the definition: 4 by default, 8 on request (orc_set_sum_classes; kernels_surfel.hip: Intrinsics::
 * sum_classes, what keyframe sharding over 8 ranks needs): p0 + p1 + ... in ascending order of the class. */
#define ORC_SPLIT 8            /* room for either */
static int g_sum_classes = 4;
void orc_set_sum_classes(int classes) { if (classes == 4 || classes == 8) g_sum_classes = classes; }
int orc_get_sum_classes(void) { return g_sum_classes; }
static inline float combine4(const float p[ORC_SPLIT]) {
  float t = p[0] + p[1];
  for (int c = 2; c < g_sum_classes; ++c) t += p[c];
  return t;
}

/* Normals pass shared by B/kernel_opt_geometry.cc:39-77 and :108-134; kernels
 * B/kernel_opt_geometry.cu:82-101 (reset), :527-553 (accumulate), :577-597 (update). */
void orc_update_surfel_normals(const orc_camera* depth_cam, const orc_depth_params* dp,
                               orc_keyframe* const* kfs, int num_kfs, orc_surfels* s) {
  if (s->surfels_size == 0) return;
  float* a0 = srow(s, ORC_SURFEL_ACCUM0 + 0); float* a1 = srow(s, ORC_SURFEL_ACCUM0 + 1);
  float* a2 = srow(s, ORC_SURFEL_ACCUM0 + 2); float* a3 = srow(s, ORC_SURFEL_ACCUM0 + 3);
#pragma omp parallel for schedule(static)
  for (uint32_t i = 0; i < s->surfels_size; ++i) {
    if (!(s->active[i] & ORC_SURFEL_ACTIVE_FLAG)) continue;
    float part[4][ORC_SPLIT] = {{0}};
    int bound_index = -1;   /* index among the non-deleted keyframes = index in the HIP keyframe table */
    for (int k = 0; k < num_kfs; ++k) {
      const orc_keyframe* kf = kfs[k];
      if (!kf) continue;
      ++bound_index;
      if (kf->activation == ORC_KF_INACTIVE) continue;
      proj_params p = make_proj_params(depth_cam, dp, s, kf, kf->frame_T_global);
      proj_result r;
      if (!orc_project_associate(&p, i, &r, NULL)) continue;
      float m[3];
      orc_unpack_normal8(kf->normals[(size_t)r.py * kf->width + r.px], m);
      const v3 g = m33_mul(kf->global_R_frame, v3_make(m[0], m[1], m[2]));
      const int j = bound_index % g_sum_classes;
      part[0][j] += g.x; part[1][j] += g.y; part[2][j] += g.z; part[3][j] += 1.f;
    }
    a0[i] = combine4(part[0]); a1[i] = combine4(part[1]); a2[i] = combine4(part[2]); a3[i] = combine4(part[3]);
    const float count = a3[i];
    if (count >= 1) surfel_set_normal(s, i, v3_scale(1.f / count, v3_make(a0[i], a1[i], a2[i])));
  }
}

/* B/kernel_opt_geometry.cc:80-201 */
void orc_optimize_geometry_iteration(int use_depth, int use_desc, const orc_camera* color_cam,
                                     const orc_camera* depth_cam, const orc_depth_params* dp,
                                     orc_keyframe* const* kfs, int num_kfs, orc_surfels* s) {
  if (s->surfels_size == 0) return;
  orc_update_surfel_normals(depth_cam, dp, kfs, num_kfs, s);

  const depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
  float* acc[9];
  for (int k = 0; k < 9; ++k) acc[k] = srow(s, ORC_SURFEL_ACCUM0 + k);

  if (!use_desc) {
    /* depth only: B/kernel_opt_geometry.cu:381-399 (reset 0..1), :417-460, :487-508 */
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < s->surfels_size; ++i) {
      if (!(s->active[i] & ORC_SURFEL_ACTIVE_FLAG)) continue;
      float part[2][ORC_SPLIT] = {{0}};
      int bound_index = -1;
      for (int k = 0; k < num_kfs; ++k) {
        const orc_keyframe* kf = kfs[k];
        if (!kf) continue;
        ++bound_index;
        if (kf->activation == ORC_KF_INACTIVE) continue;
        const int j = bound_index % g_sum_classes;
        proj_params p = make_proj_params(depth_cam, dp, s, kf, kf->frame_T_global);
        proj_result r;
        if (!orc_project_associate(&p, i, &r, NULL)) continue;
        const v3 rn = m34_rotate(kf->frame_T_global, r.normal);
        const float inv_std = depth_inv_stddev(unp_nx(&p.unp, (float)r.px), unp_ny(&p.unp, (float)r.py),
                                               r.calibrated_depth, rn, dp->baseline_fx);
        const float depth_jacobian = -inv_std;
        const v3 u = unp_point(&p.unp, r.px, r.py, r.calibrated_depth);
        const float raw = inv_std * v3_dot(rn, v3_sub(u, r.local_position));
        const float w = depth_residual_weight(raw);
        const float weighted_jacobian = w * depth_jacobian;
        part[0][j] = mad(weighted_jacobian, depth_jacobian, part[0][j]);
        part[1][j] = mad(weighted_jacobian, raw, part[1][j]);
      }
      acc[0][i] = combine4(part[0]); acc[1][i] = combine4(part[1]);
      const float Hs = acc[0][i];
      if (Hs > 1e-6f) {
        const v3 gp = surfel_position(s, i);
        const float t = -1.f * acc[1][i] / Hs;
        const v3 n = surfel_normal(s, i);
        surfel_set_position(s, i, v3_add(gp, v3_scale(t, n)));
      }
    }
    return;
  }

  /* joint position + descriptors: B/kernel_opt_geometry.cu:41-63 (reset 0..8), :119-230, :273-353 */
#pragma omp parallel for schedule(static)
  for (uint32_t i = 0; i < s->surfels_size; ++i) {
    if (!(s->active[i] & ORC_SURFEL_ACTIVE_FLAG)) continue;
    float part[9][ORC_SPLIT] = {{0}};
    int bound_index = -1;
    for (int k = 0; k < num_kfs; ++k) {
      const orc_keyframe* kf = kfs[k];
      if (!kf) continue;
      ++bound_index;
      if (kf->activation == ORC_KF_INACTIVE) continue;
      const int j = bound_index % g_sum_classes;
      proj_params p = make_proj_params(depth_cam, dp, s, kf, kf->frame_T_global);
      proj_result r;
      if (!orc_project_associate(&p, i, &r, NULL)) continue;
      const float* F = kf->frame_T_global;
      const v3 rn = m34_rotate(F, r.normal);
      if (use_depth) {
        const float inv_std = depth_inv_stddev(unp_nx(&p.unp, (float)r.px), unp_ny(&p.unp, (float)r.py),
                                               r.calibrated_depth, rn, dp->baseline_fx);
        const float depth_jacobian = -inv_std;
        const v3 u = unp_point(&p.unp, r.px, r.py, r.calibrated_depth);
        const float raw = inv_std * v3_dot(rn, v3_sub(u, r.local_position));
        const float w = depth_residual_weight(raw);
        part[0][j] = mad(w * depth_jacobian, depth_jacobian, part[0][j]);
        part[6][j] = mad(w * raw, depth_jacobian, part[6][j]);
      }
      float c[2];
      if (transform_depth_to_color(r.pxx, r.pxy, &d2c, &c[0], &c[1])) {
        float t1[2], t2[2], raw1, raw2, g[4];
        orc_tangent_projections(r.global_position, r.normal, srow(s, ORC_SURFEL_RADIUS_SQ)[i], F, color_cam, t1, t2);
        orc_raw_descriptor_residual(kf, c, t1, t2, srow(s, ORC_SURFEL_DESC1)[i], srow(s, ORC_SURFEL_DESC2)[i], &raw1, &raw2);
        orc_descriptor_gradient(kf, c, t1, t2, g);
        const v3 lp = r.local_position;
        const float jp1 = jac_descriptor_surfel(rn, lp, g[0], g[1], color_cam->fx, color_cam->fy);
        const float jp2 = jac_descriptor_surfel(rn, lp, g[2], g[3], color_cam->fx, color_cam->fy);
        const float jd = -1.f;
        const float w1 = descriptor_residual_weight(raw1);
        const float wr1 = w1 * raw1;
        const float w2 = descriptor_residual_weight(raw2);
        const float wr2 = w2 * raw2;
        part[0][j] = mad(w2 * jp2, jp2, mad(w1 * jp1, jp1, part[0][j]));
        part[1][j] += w1 * jp1 * jd;
        part[3][j] += w1 * jd * jd;
        part[6][j] = mad(wr2, jp2, mad(wr1, jp1, part[6][j]));
        part[7][j] += wr1 * jd;
        part[2][j] += w2 * jp2 * jd;
        part[5][j] += w2 * jd * jd;
        part[8][j] += wr2 * jd;
      }
    }
    for (int q = 0; q < 9; ++q) acc[q][i] = combine4(part[q]);
    /* B/kernel_opt_geometry.cu:273-353 */
    float H00 = acc[0][i], H01 = acc[1][i], H02 = acc[2][i], H11 = acc[3][i], H12 = acc[4][i], H22 = acc[5][i];
    const float kEpsilon = 1e-6f;
    H00 += kEpsilon; H11 += kEpsilon; H22 += kEpsilon;
    H00 = sqrtf(H00);
    H01 = H01 / H00;
    H11 = sqrtf(H11 - H01 * H01);
    H02 = H02 / H00;
    H12 = (H12 - H02 * H01) / H11;
    H22 = sqrtf(H22 - H02 * H02 - H12 * H12);
    const float b0 = acc[6][i], b1 = acc[7][i], b2 = acc[8][i];
    const float y0 = b0 / H00;
    const float y1 = (b1 - H01 * y0) / H11;
    const float y2 = (b2 - H02 * y0 - H12 * y1) / H22;
    const float x2 = y2 / H22;
    const float x1 = (y1 - H12 * x2) / H11;
    const float x0 = (y0 - H02 * x2 - H01 * x1) / H00;
    if (x0 != 0) {
      const v3 gp = surfel_position(s, i);
      const v3 n = surfel_normal(s, i);
      surfel_set_position(s, i, v3_sub(gp, v3_scale(x0, n)));
    }
    if (x1 != 0) {
      float d = srow(s, ORC_SURFEL_DESC1)[i];
      d -= x1;
      srow(s, ORC_SURFEL_DESC1)[i] = fmaxf(-180.f, fminf(180.f, d));
    }
    if (x2 != 0) {
      float d = srow(s, ORC_SURFEL_DESC2)[i];
      d -= x2;
      srow(s, ORC_SURFEL_DESC2)[i] = fmaxf(-180.f, fminf(180.f, d));
    }
  }
}
