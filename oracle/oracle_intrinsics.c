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

/* Sum of 64 values in the order of the classic xor butterfly (32, 16, 8, 4, 2, 1), binary32: the backend's wave_sum
 * (wave_reduce.h).  Every lane of the butterfly ends with the same value; lane 0's is returned. */
float orc_wave_xor_sum(const float x[64]) {
  float v[64], t[64];
  memcpy(v, x, sizeof(v));
  for (int s = 32; s >= 1; s >>= 1) {
    for (int i = 0; i < 64; ++i) t[i] = v[i] + v[i ^ s];
    memcpy(v, t, sizeof(v));
  }
  return v[0];
}

/* DEFINITION of the sums (shared with the backend, kernels_intrinsics.hip): the terms are binary32, the reference's
 * expressions; the 34 global sums are per-surfel binary32 chains over the keyframes in ascending order, then the xor
 * butterfly over the 64 surfels of a tile, then binary64 over the tiles; the per-cell sums are binary64, pair by pair; both
 * are rounded to binary32 before the Schur complement.  (The reference adds binary32 atomics in arbitrary order,
 * B/kernel_opt_intrinsics.cu:217-262.) */
/* The accumulation alone: glob[34] (slots 0..14 A, 15..19 b1, 20..29 colour H, 30..33 colour b) and cells[8 S] (per sparse
 * cell: B0..B4, D, b2, observation count), both binary64, ADDED to what the arrays hold.  This is what a surfel-sharded run
 * sums over its ranks (BAHIP_SUM_F64) before the Schur complement. */
void orc_intrinsics_accumulate(int optimize_depth_intrinsics, int optimize_color_intrinsics,
                               orc_keyframe* const* kfs, int num_kfs, const orc_camera* color_cam,
                               const orc_camera* depth_cam, const orc_depth_params* dp, const orc_surfels* s,
                               double glob[34], double* cells) {
  const unprojector unp = make_unprojector(depth_cam);
  const depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
  proj_params* pp = (proj_params*)calloc((size_t)(num_kfs > 0 ? num_kfs : 1), sizeof(proj_params));
  for (int k = 0; k < num_kfs; ++k)
    if (kfs[k]) pp[k] = make_proj_params(depth_cam, dp, s, kfs[k], kfs[k]->frame_T_global);

  for (uint32_t tile = 0; tile < s->surfels_size; tile += 64) {
    float lanes[34][64];
    memset(lanes, 0, sizeof(lanes));
    for (uint32_t lane = 0; lane < 64 && tile + lane < s->surfels_size; ++lane) {
      const uint32_t i = tile + lane;
      float acc[34] = {0};
      float* A = acc; float* b1 = acc + 15; float* color_H = acc + 20; float* color_b = acc + 30;
      for (int k = 0; k < num_kfs; ++k) {
        const orc_keyframe* kf = kfs[k];
        if (!kf) continue;
        const float* F = kf->frame_T_global;
        proj_result r;
        if (!orc_project_associate(&pp[k], i, &r, NULL)) continue;
        const float nx = unp_nx(&unp, (float)r.px), ny = unp_ny(&unp, (float)r.py);
        if (optimize_depth_intrinsics) {
          const int sparse_px = r.px / dp->cell, sparse_py = r.py / dp->cell;
          const float cfactor = dp->cfactor[(size_t)sparse_py * dp->cf_width + sparse_px];
          const float raw_inv_depth = 1.0f / (dp->raw_to_float_depth * kf->depth[(size_t)r.py * kf->width + r.px]);
          const float exp_inv_depth = orc_exp(-dp->a * raw_inv_depth);
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
            double* cell = cells + (size_t)8 * (sparse_px + sparse_py * dp->cf_width);
            add_h_b(K_A_ROWS, A, b1, raw, w, J);
            for (int q = 0; q < K_A_ROWS; ++q) cell[q] += (double)(w * J[q] * J[K_A_ROWS]);
            cell[5] += (double)(w * J[K_A_ROWS] * J[K_A_ROWS]);
            cell[6] += (double)(w * raw * J[K_A_ROWS]);
            cell[7] += 1.0;
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
      for (int q = 0; q < 34; ++q) lanes[q][lane] = acc[q];
    }
    for (int q = 0; q < 34; ++q) glob[q] += (double)orc_wave_xor_sum(lanes[q]);
  }
  free(pp);
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
  const int S = dp->cf_width * dp->cf_height;
  double glob[34] = {0};
  double* cells = (double*)calloc((size_t)8 * S, sizeof(double));
  orc_intrinsics_accumulate(optimize_depth_intrinsics, optimize_color_intrinsics, kfs, num_kfs, color_cam, depth_cam, dp, s, glob, cells);

  float A[15], b1[K_A_ROWS], color_H[10], color_b[4];
  for (int q = 0; q < 15; ++q) A[q] = (float)glob[q];
  for (int q = 0; q < K_A_ROWS; ++q) b1[q] = (float)glob[15 + q];
  for (int q = 0; q < 10; ++q) color_H[q] = (float)glob[20 + q];
  for (int q = 0; q < 4; ++q) color_b[q] = (float)glob[30 + q];
  float* B = (float*)calloc((size_t)K_A_ROWS * S, sizeof(float));
  float* D = (float*)calloc((size_t)S, sizeof(float));
  float* b2 = (float*)calloc((size_t)S, sizeof(float));
  uint32_t* obs = (uint32_t*)calloc((size_t)S, sizeof(uint32_t));
  for (int cell = 0; cell < S; ++cell) {
    for (int q = 0; q < K_A_ROWS; ++q) B[(size_t)q * S + cell] = (float)cells[(size_t)8 * cell + q];
    D[cell] = (float)cells[(size_t)8 * cell + 5];
    b2[cell] = (float)cells[(size_t)8 * cell + 6];
    obs[cell] = (uint32_t)cells[(size_t)8 * cell + 7];
  }
  free(cells);

  if (optimize_depth_intrinsics) {
    /* Schur complement, B/kernel_opt_intrinsics.cu:266-350 */
    /* The 20 sums over the cells: xor butterfly over each group of 64 consecutive cells, the group totals added in order
     * (binary32, from 0), that total added to the rounded accumulator -- the backend's definition (intrinsics_schur_kernel). */
    float total[20] = {0};
    for (int base = 0; base < S; base += 64) {
      float part[20][64];
      memset(part, 0, sizeof(part));
      for (int lane = 0; lane < 64 && base + lane < S; ++lane) {
        const int cell = base + lane;
        const float D_inverse = 1.0f / D[cell];
        if (!(D_inverse < 1e12f)) { D[cell] = NAN; continue; }
        const float D_inv_b2 = D_inverse * b2[cell];
        D[cell] = D_inv_b2;
        int index = 0;
        for (int row = 0; row < K_A_ROWS; ++row)
          for (int col = row; col < K_A_ROWS; ++col)
            part[index++][lane] = -1.f * (B[(size_t)row * S + cell] * D_inverse * B[(size_t)col * S + cell]);
        for (int row = 0; row < K_A_ROWS; ++row) part[15 + row][lane] = -1.f * (B[(size_t)row * S + cell] * D_inv_b2);
        for (int row = 0; row < K_A_ROWS; ++row) B[(size_t)row * S + cell] = D_inverse * B[(size_t)row * S + cell];
      }
      for (int q = 0; q < 20; ++q) total[q] += orc_wave_xor_sum(part[q]);
    }
    for (int q = 0; q < 15; ++q) A[q] += total[q];
    for (int q = 0; q < K_A_ROWS; ++q) b1[q] += total[15 + q];
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
