/* oracle_pcg.c -- the matrix-free PCG Gauss-Newton scheme.
 * Test infrastructure only (see oracle.h).
 * Follows B/direct_ba_pcg.cc:43-819 (driver) and B/kernel_pcg.cu:44-1389 (kernels).
 * PCGScalar = float (B/kernels.cuh:62).  Sums that the reference forms with block reductions +
 * atomics are formed here sequentially in (keyframe, surfel) order. */
#include "oracle_internal.h"

/* The file is compiled twice (oracle/Makefile): as is -- PCGScalar = float, the reference's -- and with -DORC_PCG_DOUBLE,
 * which keeps the per-pair terms in binary32 but holds the vectors r, M, g, p, delta and every scalar of the conjugate
 * gradient recurrence in binary64 (entry points orc_bundle_adjustment_pcg_f64 / orc_pcg_assemble_f64).  The binary64
 * flavour is what the binary32 solvers -- this one and the backend's -- are measured against in
 * tests/test_gpu_directba_vs_oracle.py: it shows how far a binary32 conjugate gradient is from the solution of its own
 * linear system, i.e. how close two correct binary32 implementations can be expected to agree. */
#ifdef ORC_PCG_DOUBLE
typedef double pcg_real;
#define PCG_SQRT(x) sqrt(x)
#define orc_bundle_adjustment_pcg orc_bundle_adjustment_pcg_f64
#define orc_pcg_assemble orc_pcg_assemble_f64
#else
typedef float pcg_real;
#define PCG_SQRT(x) sqrtf(x)
#endif

static const float kDiagEpsilon = 1e-8f;     /* B/kernel_pcg.cu:44 */
static const float kAPriorWeight = 10.f;     /* B/kernel_pcg.cu:48 */
#define INVALID_UNKNOWN 0xffffffffu

typedef struct {
  int optimize_poses, optimize_geometry, optimize_depth_intrinsics, optimize_color_intrinsics;
  int use_depth, use_desc;
  uint32_t surfel_start, depth_intr_start, a_index, color_intr_start, unknown_count;
  int geom_stride;   /* 3 with descriptor residuals, else 1 */
} pcg_layout;

/* Everything PCGInit / PCGStep1 derive for one associated (surfel, keyframe) pair. */
typedef struct {
  /* depth residual */
  float raw, w, inv_std;
  float Jpose[6];
  float Jgeom;
  int di_valid;
  float Jdi[5];
  float Jcf;
  uint32_t cf_index;
  float corrected_inv_depth;
  /* descriptor residuals */
  int color_ok;
  float raw1, raw2, w1, w2;
  float Jp1[6], Jp2[6];
  float Jg1, Jg2;
  float Jci1[4], Jci2[4];
} pair_terms;

static void eval_pair_terms(const pcg_layout* L, const orc_camera* color_cam, const orc_camera* depth_cam,
                            const orc_depth_params* dp, const orc_keyframe* kf, const proj_params* p,
                            const depth_to_color* d2c, const orc_surfels* s, uint32_t i, const proj_result* r,
                            pair_terms* t) {
  const float* F = kf->frame_T_global;
  const v3 rn = m34_rotate(F, r->normal);
  memset(t, 0, sizeof(*t));
  /* All Jacobians through the jac_* helpers of oracle_internal.h (the golden-vector-tested ones), like the backend. */
  if (L->use_depth) {
    const float nx = unp_nx(&p->unp, (float)r->px), ny = unp_ny(&p->unp, (float)r->py);
    t->inv_std = depth_inv_stddev(nx, ny, r->calibrated_depth, rn, dp->baseline_fx);
    const v3 u = unp_point(&p->unp, r->px, r->py, r->calibrated_depth);
    t->raw = t->inv_std * v3_dot(rn, v3_sub(u, r->local_position));
    t->w = depth_residual_weight(t->raw);
    t->Jgeom = -t->inv_std;
    jac_depth_pose(rn, u, t->inv_std, t->Jpose);
    if (L->optimize_depth_intrinsics) {
      const int sparse_px = r->px / dp->cell, sparse_py = r->py / dp->cell;
      const float cfactor = dp->cfactor[(size_t)sparse_py * dp->cf_width + sparse_px];
      const float raw_inv_depth = 1.0f / (dp->raw_to_float_depth * kf->depth[(size_t)r->py * kf->width + r->px]);
      const float exp_inv_depth = expf(-dp->a * raw_inv_depth);
      const float corrected = cfactor * exp_inv_depth + raw_inv_depth;
      t->corrected_inv_depth = corrected;
      t->di_valid = !(fabsf(corrected) < 1e-4f);
      const float dot = v3_dot(v3_make(nx, ny, 1), rn);
      float Jdi[6];   /* fx_inv, fy_inv, cx_inv, cy_inv, a, cfactor: B/kernel_pcg.cu:258-303 */
      jac_depth_intrinsics(r->px, r->py, r->calibrated_depth, t->inv_std, v3_dot(r->normal, v3_make(F[0], F[1], F[2])),
                           v3_dot(r->normal, v3_make(F[4], F[5], F[6])), dot, cfactor, raw_inv_depth, exp_inv_depth, corrected, Jdi);
      for (int c = 0; c < 5; ++c) t->Jdi[c] = Jdi[c];
      t->Jcf = Jdi[5];
      t->cf_index = L->depth_intr_start + 5 + sparse_px + sparse_py * dp->cf_width;
    }
  }
  if (L->use_desc) {
    float c[2];
    t->color_ok = transform_depth_to_color(r->pxx, r->pxy, d2c, &c[0], &c[1]);
    if (t->color_ok) {
      float t1[2], t2[2], g[4];
      orc_tangent_projections(r->global_position, r->normal, srow(s, ORC_SURFEL_RADIUS_SQ)[i], F, color_cam, t1, t2);
      orc_raw_descriptor_residual(kf, c, t1, t2, srow(s, ORC_SURFEL_DESC1)[i], srow(s, ORC_SURFEL_DESC2)[i], &t->raw1, &t->raw2);
      orc_descriptor_gradient(kf, c, t1, t2, g);
      t->w1 = descriptor_residual_weight(t->raw1);
      t->w2 = descriptor_residual_weight(t->raw2);
      const v3 lp = r->local_position;
      t->Jg1 = jac_descriptor_surfel(rn, lp, g[0], g[1], color_cam->fx, color_cam->fy);
      t->Jg2 = jac_descriptor_surfel(rn, lp, g[2], g[3], color_cam->fx, color_cam->fy);
      /* B/kernel_pcg.cu:366-369: gradients pre-multiplied with the colour focal lengths */
      jac_descriptor_pose(lp, g[0] * color_cam->fx, g[1] * color_cam->fy, t->Jp1);
      jac_descriptor_pose(lp, g[2] * color_cam->fx, g[3] * color_cam->fy, t->Jp2);
      if (L->optimize_color_intrinsics) {
        const float nx = unp_nx(&p->unp, (float)r->px), ny = unp_ny(&p->unp, (float)r->py);
        jac_descriptor_color_intrinsics(g[0], g[1], nx, ny, t->Jci1);
        jac_descriptor_color_intrinsics(g[2], g[3], nx, ny, t->Jci2);
      }
    }
  }
}

static inline void sum_r_m(pcg_real* r, pcg_real* M, uint32_t idx, float J, float w, float raw) {
  const pcg_real wj = (pcg_real)w * J;
  r[idx] += -1 * wj * raw;
  M[idx] += J * wj;
}
static inline void sum_r_m2(pcg_real* r, pcg_real* M, uint32_t idx, float J1, float w1, float raw1, float J2, float w2, float raw2) {
  const pcg_real wj1 = (pcg_real)w1 * J1, wj2 = (pcg_real)w2 * J2;
  r[idx] += -1 * wj1 * raw1 + -1 * wj2 * raw2;
  M[idx] += J1 * wj1 + J2 * wj2;
}

/* B/kernel_pcg.cu:179-541, one keyframe */
static void pcg_init_kf(const pcg_layout* L, uint32_t pose_index, int optimize_pose_of_kf, const orc_camera* color_cam,
                        const orc_camera* depth_cam, const orc_depth_params* dp, const orc_keyframe* kf,
                        const orc_surfels* s, pcg_real* r_, pcg_real* M_) {
  proj_params p = make_proj_params(depth_cam, dp, s, kf, kf->frame_T_global);
  const depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
  for (uint32_t i = 0; i < s->surfels_size; ++i) {
    proj_result pr;
    if (!orc_project_associate(&p, i, &pr, NULL)) continue;
    pair_terms t;
    eval_pair_terms(L, color_cam, depth_cam, dp, kf, &p, &d2c, s, i, &pr, &t);
    int visible = 1;
    const uint32_t gi = L->surfel_start + (uint32_t)L->geom_stride * i;
    if (L->use_depth) {
      if (L->optimize_geometry) {
        r_[gi] -= t.Jgeom * t.w * t.raw;
        M_[gi] += t.Jgeom * t.w * t.Jgeom;
      }
      if (optimize_pose_of_kf) for (int c = 0; c < 6; ++c) sum_r_m(r_, M_, pose_index + c, t.Jpose[c], t.w, t.raw);
      if (L->optimize_depth_intrinsics) {
        if (!t.di_valid) visible = 0;   /* B/kernel_pcg.cu:272-274: also hides the descriptor part */
        if (visible) {
          sum_r_m(r_, M_, L->depth_intr_start + 2, t.Jdi[2], t.w, t.raw);
          sum_r_m(r_, M_, L->depth_intr_start + 3, t.Jdi[3], t.w, t.raw);
          sum_r_m(r_, M_, L->depth_intr_start + 0, t.Jdi[0], t.w, t.raw);
          sum_r_m(r_, M_, L->depth_intr_start + 1, t.Jdi[1], t.w, t.raw);
          sum_r_m(r_, M_, L->depth_intr_start + 4, t.Jdi[4], t.w, t.raw);
          sum_r_m(r_, M_, t.cf_index, t.Jcf, t.w, t.raw);
        }
      }
    }
    if (L->use_desc) {
      visible = visible && t.color_ok;
      if (!visible) continue;
      if (L->optimize_geometry) {
        r_[gi + 0] -= t.Jg1 * t.w1 * t.raw1 + t.Jg2 * t.w2 * t.raw2;
        M_[gi + 0] += t.Jg1 * t.w1 * t.Jg1 + t.Jg2 * t.w2 * t.Jg2;
        r_[gi + 1] -= -1.f * t.w1 * t.raw1 + 0.f * t.w2 * t.raw2;
        M_[gi + 1] += -1.f * t.w1 * -1.f + 0.f * t.w2 * 0.f;
        r_[gi + 2] -= 0.f * t.w1 * t.raw1 + -1.f * t.w2 * t.raw2;
        M_[gi + 2] += 0.f * t.w1 * 0.f + -1.f * t.w2 * -1.f;
      }
      if (optimize_pose_of_kf)
        for (int c = 0; c < 6; ++c) sum_r_m2(r_, M_, pose_index + c, t.Jp1[c], t.w1, t.raw1, t.Jp2[c], t.w2, t.raw2);
      if (L->optimize_color_intrinsics)
        for (int c = 0; c < 4; ++c) sum_r_m2(r_, M_, L->color_intr_start + c, t.Jci1[c], t.w1, t.raw1, t.Jci2[c], t.w2, t.raw2);
    }
  }
}

/* B/kernel_pcg.cu:646-1026, one keyframe: g += J^T W J p, alpha_d += p^T J^T W J p */
static void pcg_step1_kf(const pcg_layout* L, uint32_t pose_index, int optimize_pose_of_kf, const orc_camera* color_cam,
                         const orc_camera* depth_cam, const orc_depth_params* dp, const orc_keyframe* kf,
                         const orc_surfels* s, const pcg_real* p_, pcg_real* g_, pcg_real* alpha_d) {
  proj_params p = make_proj_params(depth_cam, dp, s, kf, kf->frame_T_global);
  const depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
  for (uint32_t i = 0; i < s->surfels_size; ++i) {
    proj_result pr;
    if (!orc_project_associate(&p, i, &pr, NULL)) continue;
    pair_terms t;
    eval_pair_terms(L, color_cam, depth_cam, dp, kf, &p, &d2c, s, i, &pr, &t);
    const uint32_t gi = L->surfel_start + (uint32_t)L->geom_stride * i;
    if (L->use_depth) {
      pcg_real sum = 0;
      if (L->optimize_geometry) sum += t.Jgeom * p_[gi];
      if (optimize_pose_of_kf) for (int c = 0; c < 6; ++c) sum += t.Jpose[c] * p_[pose_index + c];
      const int di = L->optimize_depth_intrinsics && t.di_valid;
      if (di) {
        sum += t.Jdi[2] * p_[L->depth_intr_start + 2];
        sum += t.Jdi[3] * p_[L->depth_intr_start + 3];
        sum += t.Jdi[0] * p_[L->depth_intr_start + 0];
        sum += t.Jdi[1] * p_[L->depth_intr_start + 1];
        sum += t.Jdi[4] * p_[L->depth_intr_start + 4];
        sum += t.Jcf * p_[t.cf_index];
      }
      *alpha_d += sum * t.w * sum;
      sum *= t.w;
      if (L->optimize_geometry) g_[gi] += t.Jgeom * sum;
      if (optimize_pose_of_kf) for (int c = 0; c < 6; ++c) g_[pose_index + c] += t.Jpose[c] * sum;
      if (di) {
        for (int c = 0; c < 5; ++c) g_[L->depth_intr_start + c] += t.Jdi[c] * sum;
        g_[t.cf_index] += t.Jcf * sum;
      }
    }
    if (L->use_desc) {
      if (!t.color_ok) continue;
      pcg_real sum1 = 0, sum2 = 0;
      if (L->optimize_geometry) {
        pcg_real pv = p_[gi + 0];
        sum1 += t.Jg1 * pv; sum2 += t.Jg2 * pv;
        pv = p_[gi + 1]; sum1 += -1.f * pv;
        pv = p_[gi + 2]; sum2 += -1.f * pv;
      }
      if (optimize_pose_of_kf)
        for (int c = 0; c < 6; ++c) { const pcg_real pv = p_[pose_index + c]; sum1 += t.Jp1[c] * pv; sum2 += t.Jp2[c] * pv; }
      if (L->optimize_color_intrinsics)
        for (int c = 0; c < 4; ++c) { const pcg_real pv = p_[L->color_intr_start + c]; sum1 += t.Jci1[c] * pv; sum2 += t.Jci2[c] * pv; }
      *alpha_d += sum1 * t.w1 * sum1 + sum2 * t.w2 * sum2;
      sum1 *= t.w1; sum2 *= t.w2;
      if (L->optimize_geometry) {
        g_[gi + 0] += t.Jg1 * sum1 + t.Jg2 * sum2;
        g_[gi + 1] += -1.f * sum1 + 0.f * sum2;
        g_[gi + 2] += 0.f * sum1 + -1.f * sum2;
      }
      if (optimize_pose_of_kf) for (int c = 0; c < 6; ++c) g_[pose_index + c] += t.Jp1[c] * sum1 + t.Jp2[c] * sum2;
      if (L->optimize_color_intrinsics)
        for (int c = 0; c < 4; ++c) g_[L->color_intr_start + c] += t.Jci1[c] * sum1 + t.Jci2[c] * sum2;
    }
  }
}

static inline float prior_at(const pcg_layout* L, uint32_t idx) {
  return (idx == L->a_index) ? (kAPriorWeight * kAPriorWeight) : 0.f;
}

/* Test hook: assembles r = -J^T W F and M = diag(J^T W J) (PCGInit over all keyframes, without the
 * prior on a) for the current state.  Returns the unknown count; writes at most `capacity` entries. */
uint32_t orc_pcg_assemble(orc_ba_state* st, const orc_ba_options* opt, float* r_out, float* M_out, uint32_t capacity) {
  const orc_surfels* s = st->surfels;
  const int K = st->num_kfs;
  const int S = st->dp.cf_width * st->dp.cf_height;
  pcg_layout L;
  memset(&L, 0, sizeof(L));
  L.use_depth = opt->use_depth_residuals; L.use_desc = opt->use_descriptor_residuals;
  L.optimize_poses = opt->optimize_poses; L.optimize_geometry = opt->optimize_geometry;
  L.optimize_depth_intrinsics = opt->optimize_depth_intrinsics && L.use_depth;
  L.optimize_color_intrinsics = opt->optimize_color_intrinsics && L.use_desc;
  L.geom_stride = L.use_desc ? 3 : 1;
  uint32_t cur = 0;
  if (L.optimize_poses) cur += 6u * (uint32_t)(K - 1);
  L.surfel_start = INVALID_UNKNOWN;
  if (L.optimize_geometry) { L.surfel_start = cur; cur += (uint32_t)L.geom_stride * s->surfels_size; }
  L.depth_intr_start = INVALID_UNKNOWN; L.a_index = INVALID_UNKNOWN;
  if (L.optimize_depth_intrinsics) { L.depth_intr_start = cur; cur += 5u + (uint32_t)S; L.a_index = L.depth_intr_start + 4; }
  L.color_intr_start = INVALID_UNKNOWN;
  if (L.optimize_color_intrinsics) { L.color_intr_start = cur; cur += 4; }
  L.unknown_count = cur;
  pcg_real* r_ = (pcg_real*)calloc(cur ? cur : 1, sizeof(pcg_real));
  pcg_real* M_ = (pcg_real*)calloc(cur ? cur : 1, sizeof(pcg_real));
  const int gauge = (opt->pcg_gauge_keyframe >= 0 && opt->pcg_gauge_keyframe < K) ? opt->pcg_gauge_keyframe : 0;
  for (int k = 0; k < K; ++k) {
    const uint32_t pi = (k == gauge) ? INVALID_UNKNOWN : ((k < gauge) ? 6u * (uint32_t)k : 6u * (uint32_t)(k - 1));
    pcg_init_kf(&L, pi, (k == gauge) ? 0 : L.optimize_poses, &st->color_cam, &st->depth_cam, &st->dp, st->kfs[k], s, r_, M_);
  }
  const uint32_t n = cur < capacity ? cur : capacity;
  for (uint32_t u = 0; u < n; ++u) { r_out[u] = (float)r_[u]; M_out[u] = (float)M_[u]; }
  free(r_); free(M_);
  return cur;
}

void orc_bundle_adjustment_pcg(orc_ba_state* st, const orc_ba_options* opt, orc_ba_stats* stats) {
  orc_surfels* s = st->surfels;
  memset(stats, 0, sizeof(*stats));
  const int K = st->num_kfs;
  pcg_layout L;
  memset(&L, 0, sizeof(L));
  L.use_depth = opt->use_depth_residuals; L.use_desc = opt->use_descriptor_residuals;
  L.optimize_poses = opt->optimize_poses; L.optimize_geometry = opt->optimize_geometry;
  L.optimize_depth_intrinsics = opt->optimize_depth_intrinsics && L.use_depth;   /* B/direct_ba.cc:427-434 */
  L.optimize_color_intrinsics = opt->optimize_color_intrinsics && L.use_desc;
  L.geom_stride = L.use_desc ? 3 : 1;
  for (int k = 0; k < K; ++k) if (!st->kfs[k]) return;   /* B/direct_ba_pcg.cc:138-143 */
  const int S = st->dp.cf_width * st->dp.cf_height;

  /* B/direct_ba_pcg.cc:152-158: end tasks of the previous block; shares the alternating scheme's helper
   * through a zero-iteration call. */
  if (!opt->increase_ba_iteration_count && st->ba_iteration_count != st->last_ba_iteration_count) {
    orc_ba_options o2 = *opt; o2.max_iterations = 0; o2.increase_ba_iteration_count = 0;
    orc_ba_stats tmp;
    orc_bundle_adjustment_alternating(st, &o2, &tmp);
  }

  pcg_real *r_ = NULL, *M_ = NULL, *delta = NULL, *g_ = NULL, *p_ = NULL;
  size_t allocated = 0;
  int n_new = 0;
  int* new_kfs = (int*)malloc(sizeof(int) * (K ? K : 1));

  for (int iteration = 0; iteration < opt->max_iterations; ++iteration) {
    stats->iterations_done += 1;
    /* --- surfel creation (B/direct_ba_pcg.cc:184-206) --- */
    n_new = 0;
    if (opt->optimize_geometry && opt->do_surfel_updates) {
      for (int k = 0; k < K; ++k) {
        orc_keyframe* kf = st->kfs[k];
        if (kf->activation == ORC_KF_ACTIVE && kf->last_active_in_ba_iteration != st->ba_iteration_count) {
          kf->last_active_in_ba_iteration = st->ba_iteration_count;
          int* all = NULL; const int* covis; int n_covis;
          if (st->covis_lists) { covis = st->covis_lists[k]; n_covis = st->covis_counts[k]; }
          else {
            all = (int*)malloc(sizeof(int) * K); n_covis = 0;
            for (int c = 0; c < K; ++c) if (c != k) all[n_covis++] = c;
            covis = all;
          }
          orc_create_surfels_for_keyframe(1, opt->min_observation_count, &st->color_cam, &st->depth_cam, &st->dp, kf, st->kfs,
                                          covis, n_covis, s, st->supporting);
          free(all);
          new_kfs[n_new++] = k;
        } else if (kf->activation == ORC_KF_COVIS_ACTIVE && kf->last_covis_in_ba_iteration != st->ba_iteration_count) {
          kf->last_covis_in_ba_iteration = st->ba_iteration_count;
        }
      }
    }
    memset(s->active, ORC_SURFEL_ACTIVE_FLAG, s->surfels_size);
    if (opt->optimize_geometry) orc_update_surfel_normals(&st->depth_cam, &st->dp, st->kfs, K, s);

    /* --- unknown layout (B/direct_ba_pcg.cc:232-307) --- */
    uint32_t cur = 0;
    const uint32_t kf_unknowns = L.optimize_poses ? 6u * (uint32_t)(K - 1) : 0u;
    if (L.optimize_poses) cur += kf_unknowns;
    L.surfel_start = INVALID_UNKNOWN;
    if (L.optimize_geometry) { L.surfel_start = cur; cur += (uint32_t)L.geom_stride * s->surfels_size; }
    L.depth_intr_start = INVALID_UNKNOWN; L.a_index = INVALID_UNKNOWN;
    if (L.optimize_depth_intrinsics) { L.depth_intr_start = cur; cur += 5u + (uint32_t)S; L.a_index = L.depth_intr_start + 4; }
    L.color_intr_start = INVALID_UNKNOWN;
    if (L.optimize_color_intrinsics) { L.color_intr_start = cur; cur += 4; }
    L.unknown_count = cur;
    const uint32_t U = cur;
    if (U > allocated) {
      free(r_); free(M_); free(delta); free(g_); free(p_);
      allocated = U + 1024;
      r_ = (pcg_real*)malloc(sizeof(pcg_real) * allocated); M_ = (pcg_real*)malloc(sizeof(pcg_real) * allocated);
      delta = (pcg_real*)malloc(sizeof(pcg_real) * allocated); g_ = (pcg_real*)malloc(sizeof(pcg_real) * allocated);
      p_ = (pcg_real*)malloc(sizeof(pcg_real) * allocated);
    }
    memset(r_, 0, sizeof(pcg_real) * U); memset(M_, 0, sizeof(pcg_real) * U);

    const int gauge = (opt->pcg_gauge_keyframe >= 0 && opt->pcg_gauge_keyframe < K) ? opt->pcg_gauge_keyframe : 0;
#define KF_POSE_INDEX(id) ((id) == gauge ? INVALID_UNKNOWN : ((id) < gauge ? 6u * (uint32_t)(id) : 6u * (uint32_t)((id) - 1)))

    for (int k = 0; k < K; ++k)
      pcg_init_kf(&L, KF_POSE_INDEX(k), (k == gauge) ? 0 : L.optimize_poses, &st->color_cam, &st->depth_cam, &st->dp,
                  st->kfs[k], s, r_, M_);

    /* PCGInit2, B/kernel_pcg.cu:565-600 */
    pcg_real alpha_n = 0, alpha_d = 0, beta_n = 0;
    for (uint32_t u = 0; u < U; ++u) {
      g_[u] = 0;
      const pcg_real r_value = r_[u] + ((u == L.a_index) ? (-kAPriorWeight * kAPriorWeight * st->dp.a) : 0);
      const pcg_real p_value = r_value / (M_[u] + kDiagEpsilon + prior_at(&L, u));
      p_[u] = p_value;
      delta[u] = 0;
      alpha_n += r_value * p_value;
    }

    pcg_real prev_r_norm = INFINITY;
    int no_improvement = 0;
    for (int step = 0; step < opt->pcg_max_inner_iterations; ++step) {
      stats->pcg_inner_steps_total += 1;
      alpha_d = 0;
      if (step > 0) {
        const pcg_real tmp = alpha_n; alpha_n = beta_n; beta_n = tmp;
        memset(g_, 0, sizeof(pcg_real) * U);
      }
      for (int k = 0; k < K; ++k) {
        pcg_step1_kf(&L, KF_POSE_INDEX(k), (k == gauge) ? 0 : L.optimize_poses, &st->color_cam, &st->depth_cam, &st->dp,
                     st->kfs[k], s, p_, g_, &alpha_d);
        /* AddAlphaDEpsilonTerms runs once per keyframe (B/kernel_pcg.cu:1102-1112): reproduced */
        if (s->surfels_size > 0)
          for (uint32_t u = 0; u < U; ++u) alpha_d += (kDiagEpsilon + prior_at(&L, u)) * p_[u] * p_[u];
      }
      /* PCGStep2, B/kernel_pcg.cu:1117-1158 */
      beta_n = 0;
      const pcg_real alpha = (alpha_d >= 1e-35f) ? (alpha_n / alpha_d) : 0;
      for (uint32_t u = 0; u < U; ++u) {
        const pcg_real p_value = p_[u];
        delta[u] += alpha * p_value;
        pcg_real r_value = r_[u];
        r_value -= alpha * (g_[u] + (kDiagEpsilon + prior_at(&L, u)) * p_value);
        r_[u] = r_value;
        const pcg_real z_value = r_value / (M_[u] + kDiagEpsilon + prior_at(&L, u));
        g_[u] = z_value;
        beta_n += z_value * r_value;
      }
      const pcg_real r_norm = PCG_SQRT(beta_n);
      if (r_norm < prev_r_norm - 1e-3f) no_improvement = 0;
      else if (++no_improvement >= 3) break;
      prev_r_norm = r_norm;
      if (step < opt->pcg_max_inner_iterations - 1) {
        /* PCGStep3, B/kernel_pcg.cu:1212-1226 */
        const pcg_real beta = (alpha_n >= 1e-35f) ? (beta_n / alpha_n) : 0;
        for (uint32_t u = 0; u < U; ++u) p_[u] = g_[u] + beta * p_[u];
      }
    }

    /* --- apply the update (B/direct_ba_pcg.cc:551-642) --- */
    int num_converged = 0;
    if (L.optimize_poses) {
      for (int k = 0; k < K; ++k) {
        if (k == gauge) { ++num_converged; continue; }
        orc_se3 d, next;
        float step6[6];
        for (int c = 0; c < 6; ++c) step6[c] = (float)delta[KF_POSE_INDEX(k) + c];
        orc_se3_exp(step6, &d);
        orc_se3_mul(&st->kfs[k]->global_T_frame, &d, &next);
        orc_keyframe_set_global_T_frame(st->kfs[k], &next);
        float lg[6];
        orc_se3_log(&d, lg);
        if (orc_is_scale1_pose_converged(lg)) ++num_converged;
      }
    }
    if (L.optimize_geometry) {
      for (uint32_t i = 0; i < s->surfels_size; ++i) {
        const uint32_t gi = L.surfel_start + (uint32_t)L.geom_stride * i;
        const float tt = (float)delta[gi];
        if (tt != 0) surfel_set_position(s, i, v3_add(surfel_position(s, i), v3_scale(tt, surfel_normal(s, i))));
        if (L.use_desc) {
          float d1 = srow(s, ORC_SURFEL_DESC1)[i]; d1 += (float)delta[gi + 1];
          srow(s, ORC_SURFEL_DESC1)[i] = fmaxf(-180.f, fminf(180.f, d1));
          float d2 = srow(s, ORC_SURFEL_DESC2)[i]; d2 += (float)delta[gi + 2];
          srow(s, ORC_SURFEL_DESC2)[i] = fmaxf(-180.f, fminf(180.f, d2));
        }
      }
    }
    if (L.optimize_depth_intrinsics) {
      const pcg_real* b = &delta[L.depth_intr_start];
      const double old_fx_inv = 1. / st->depth_cam.fx, old_fy_inv = 1. / st->depth_cam.fy;
      const double old_cx_pc = st->depth_cam.cx - 0.5, old_cy_pc = st->depth_cam.cy - 0.5;
      const double old_cx_inv = -old_cx_pc * old_fx_inv, old_cy_inv = -old_cy_pc * old_fy_inv;
      const double new_fx = 1. / (old_fx_inv + b[0]), new_fy = 1. / (old_fy_inv + b[1]);
      const double new_cx = -(new_fx * (old_cx_inv + b[2])) + 0.5, new_cy = -(new_fy * (old_cy_inv + b[3])) + 0.5;
      st->depth_cam.fx = (float)new_fx; st->depth_cam.fy = (float)new_fy;
      st->depth_cam.cx = (float)new_cx; st->depth_cam.cy = (float)new_cy;
      st->dp.a += (float)b[4];
      for (int c = 0; c < S; ++c) st->dp.cfactor[c] += (float)delta[L.depth_intr_start + 5 + c];
    }
    if (L.optimize_color_intrinsics) {
      const pcg_real* b = &delta[L.color_intr_start];
      st->color_cam.fx = (float)(st->color_cam.fx + b[0]); st->color_cam.fy = (float)(st->color_cam.fy + b[1]);
      st->color_cam.cx = (float)(st->color_cam.cx + b[2]); st->color_cam.cy = (float)(st->color_cam.cy + b[3]);
    }
    /* --- merge + compaction (B/direct_ba_pcg.cc:651-690) --- */
    if (opt->do_surfel_updates) {
      for (int j = 0; j < n_new; ++j)
        orc_determine_supporting_surfels(1, opt->surfel_merge_dist_factor, &st->depth_cam, &st->dp, st->kfs[new_kfs[j]], s, st->supporting);
      if (n_new > 0) orc_compact_surfels(s);
    }
    if (iteration >= opt->min_iterations - 1 && (num_converged == K || !L.optimize_poses)) { stats->converged = 1; break; }
  }
  free(r_); free(M_); free(delta); free(g_); free(p_);

  if (opt->increase_ba_iteration_count) {
    /* PerformBASchemeEndTasks + ++ba_iteration_count_ (B/direct_ba_pcg.cc:800-812) via the shared helper */
    orc_ba_options o2 = *opt; o2.max_iterations = 0; o2.increase_ba_iteration_count = 1;
    orc_ba_stats tmp;
    orc_bundle_adjustment_alternating(st, &o2, &tmp);
  } else if (opt->do_surfel_updates) {
    /* B/direct_ba_pcg.cc:775-812: the merge + compaction of the last iteration is repeated */
    for (int j = 0; j < n_new; ++j)
      orc_determine_supporting_surfels(1, opt->surfel_merge_dist_factor, &st->depth_cam, &st->dp, st->kfs[new_kfs[j]], s, st->supporting);
    if (n_new > 0) orc_compact_surfels(s);
  }
  free(new_kfs);
}
