/* oracle_intrinsics.c -- depth intrinsics + deformation (Schur complement) and colour intrinsics.
 * Test infrastructure only (see oracle.h).
 * Follows B/kernel_opt_intrinsics.cc:39-281 and B/kernel_opt_intrinsics.cu:47-448. */
#include "oracle_internal.h"

#define K_A_ROWS 5

static inline void add_h_b(int n, float* H, float* b, float raw, float w, const float* J) {
  int k = 0;
  for (int row = 0; row < n; ++row)
    for (int col = row; col < n; ++col) H[k++] += w * J[row] * J[col];
  const float wr = w * raw;
  for (int i = 0; i < n; ++i) b[i] += wr * J[i];
}

void orc_optimize_intrinsics(int optimize_depth_intrinsics, int optimize_color_intrinsics,
                             orc_keyframe* const* kfs, int num_kfs, const orc_camera* color_cam,
                             const orc_camera* depth_cam, orc_depth_params* dp, const orc_surfels* s,
                             orc_camera* out_color_cam, orc_camera* out_depth_cam, float* out_a) {
  *out_color_cam = *color_cam;
  *out_depth_cam = *depth_cam;
  *out_a = dp->a;
  if (s->surfels_size == 0) return;
  const unprojector unp = make_unprojector(depth_cam);
  const depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
  const int S = dp->cf_width * dp->cf_height;

  float A[15] = {0}, b1[K_A_ROWS] = {0}, color_H[10] = {0}, color_b[4] = {0};
  float* B = (float*)calloc((size_t)K_A_ROWS * S, sizeof(float));
  float* D = (float*)calloc((size_t)S, sizeof(float));
  float* b2 = (float*)calloc((size_t)S, sizeof(float));
  uint32_t* obs = (uint32_t*)calloc((size_t)S, sizeof(uint32_t));

  for (int k = 0; k < num_kfs; ++k) {
    const orc_keyframe* kf = kfs[k];
    if (!kf) continue;
    const float* F = kf->frame_T_global;
    proj_params p = make_proj_params(depth_cam, dp, s, kf, F);
    for (uint32_t i = 0; i < s->surfels_size; ++i) {
      proj_result r;
      if (!orc_project_associate(&p, i, &r, NULL)) continue;
      const float nx = unp_nx(&unp, (float)r.px), ny = unp_ny(&unp, (float)r.py);
      if (optimize_depth_intrinsics) {
        const int sparse_px = r.px / dp->cell, sparse_py = r.py / dp->cell;
        const float cfactor = dp->cfactor[(size_t)sparse_py * dp->cf_width + sparse_px];
        const float raw_inv_depth = 1.0f / (dp->raw_to_float_depth * kf->depth[(size_t)r.py * kf->width + r.px]);
        const float exp_inv_depth = expf(-dp->a * raw_inv_depth);
        const float corrected_inv_depth = cfactor * exp_inv_depth + raw_inv_depth;
        if (fabsf(corrected_inv_depth) > 1e-4f) {
          const v3 nl = m34_rotate(F, r.normal);
          const float dot = v3_dot(v3_make(nx, ny, 1), nl);
          const float inv_std = depth_inv_stddev(nx, ny, r.calibrated_depth, nl, dp->baseline_fx);
          float J[K_A_ROWS + 1];
          jac_depth_intrinsics(r.px, r.py, r.calibrated_depth, inv_std, v3_dot(r.normal, v3_make(F[0], F[1], F[2])),
                               v3_dot(r.normal, v3_make(F[4], F[5], F[6])), dot, cfactor, raw_inv_depth, exp_inv_depth,
                               corrected_inv_depth, J);
          const v3 u = v3_make(r.calibrated_depth * nx, r.calibrated_depth * ny, r.calibrated_depth);
          const float raw = inv_std * v3_dot(nl, v3_sub(u, r.local_position));
          const float w = depth_residual_weight(raw);
          const int cell = sparse_px + sparse_py * dp->cf_width;
          add_h_b(K_A_ROWS, A, b1, raw, w, J);
          for (int q = 0; q < K_A_ROWS; ++q) B[(size_t)q * S + cell] += w * J[q] * J[K_A_ROWS];
          D[cell] += w * J[K_A_ROWS] * J[K_A_ROWS];
          b2[cell] += w * raw * J[K_A_ROWS];
          obs[cell] += 1;
        }
      }
      if (optimize_color_intrinsics) {
        float c[2];
        if (transform_depth_to_color(r.pxx, r.pxy, &d2c, &c[0], &c[1])) {
          float t1[2], t2[2], g[4], raw1, raw2;
          orc_tangent_projections(r.global_position, r.normal, srow(s, ORC_SURFEL_RADIUS_SQ)[i], F, color_cam, t1, t2);
          orc_descriptor_gradient(kf, c, t1, t2, g);
          orc_raw_descriptor_residual(kf, c, t1, t2, srow(s, ORC_SURFEL_DESC1)[i], srow(s, ORC_SURFEL_DESC2)[i], &raw1, &raw2);
          float J1[4], J2[4];
          jac_descriptor_color_intrinsics(g[0], g[1], nx, ny, J1);
          jac_descriptor_color_intrinsics(g[2], g[3], nx, ny, J2);
          /* validity flag is "residual != 0" (B/kernel_opt_intrinsics.cu:200-215) */
          if (raw1 != 0) add_h_b(4, color_H, color_b, raw1, descriptor_residual_weight(raw1), J1);
          if (raw2 != 0) add_h_b(4, color_H, color_b, raw2, descriptor_residual_weight(raw2), J2);
        }
      }
    }
  }

  if (optimize_depth_intrinsics) {
    /* Schur complement, B/kernel_opt_intrinsics.cu:266-350 */
    for (int cell = 0; cell < S; ++cell) {
      const float D_inverse = 1.0f / D[cell];
      if (!(D_inverse < 1e12f)) { D[cell] = NAN; continue; }
      const float D_inv_b2 = D_inverse * b2[cell];
      D[cell] = D_inv_b2;
      int index = 0;
      for (int row = 0; row < K_A_ROWS; ++row)
        for (int col = row; col < K_A_ROWS; ++col)
          A[index++] += -1.f * (B[(size_t)row * S + cell] * D_inverse * B[(size_t)col * S + cell]);
      for (int row = 0; row < K_A_ROWS; ++row) b1[row] += -1.f * (B[(size_t)row * S + cell] * D_inv_b2);
      for (int row = 0; row < K_A_ROWS; ++row) B[(size_t)row * S + cell] = D_inverse * B[(size_t)row * S + cell];
    }
    float M[K_A_ROWS][K_A_ROWS];
    int index = 0;
    for (int row = 0; row < K_A_ROWS; ++row)
      for (int col = row; col < K_A_ROWS; ++col) { M[row][col] = A[index]; M[col][row] = A[index]; ++index; }
    float rhs[K_A_ROWS];
    for (int i = 0; i < K_A_ROWS; ++i) rhs[i] = b1[i];
    const float kAPriorWeight = 10;
    M[4][4] += kAPriorWeight * kAPriorWeight;
    rhs[4] += kAPriorWeight * kAPriorWeight * dp->a;
    double Md[K_A_ROWS * K_A_ROWS], rd[K_A_ROWS], xd[K_A_ROWS];
    for (int i = 0; i < K_A_ROWS; ++i) { rd[i] = rhs[i]; for (int j = 0; j < K_A_ROWS; ++j) Md[i * K_A_ROWS + j] = M[i][j]; }
    orc_ldlt_solve(K_A_ROWS, Md, rd, xd);
    float x1[K_A_ROWS];
    for (int i = 0; i < K_A_ROWS; ++i) x1[i] = (float)xd[i];
    const float new_fx = 1.0f / (unp.fx_inv - x1[0]);
    const float new_fy = 1.0f / (unp.fy_inv - x1[1]);
    const float new_cx = -(new_fx * (unp.cx_inv - x1[2])) + 0.5f;
    const float new_cy = -(new_fy * (unp.cy_inv - x1[3])) + 0.5f;
    out_depth_cam->fx = new_fx; out_depth_cam->fy = new_fy; out_depth_cam->cx = new_cx; out_depth_cam->cy = new_cy;
    *out_a = dp->a - x1[4];
    /* B/kernel_opt_intrinsics.cu:375-423 */
    for (int cell = 0; cell < S; ++cell) {
      float offset = D[cell];
      if (isnan(offset)) offset = 0;
      else for (int row = 0; row < K_A_ROWS; ++row) offset -= B[(size_t)row * S + cell] * x1[row];
      float cfactor = dp->cfactor[cell] - offset;
      if (obs[cell] == 0) cfactor = 0;
      dp->cfactor[cell] = cfactor;
    }
  }
  if (optimize_color_intrinsics) {
    double Md[16], rd[4], xd[4];
    int index = 0;
    for (int row = 0; row < 4; ++row)
      for (int col = row; col < 4; ++col) { Md[row * 4 + col] = color_H[index]; Md[col * 4 + row] = color_H[index]; ++index; }
    for (int i = 0; i < 4; ++i) rd[i] = color_b[i];
    orc_ldlt_solve(4, Md, rd, xd);
    out_color_cam->fx = color_cam->fx - (float)xd[0];
    out_color_cam->fy = color_cam->fy - (float)xd[1];
    out_color_cam->cx = color_cam->cx - (float)xd[2];
    out_color_cam->cy = color_cam->cy - (float)xd[3];
  }
  free(B); free(D); free(b2); free(obs);
}
