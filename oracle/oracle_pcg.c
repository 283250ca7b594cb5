/* oracle_pcg.c -- the matrix-free PCG Gauss-Newton scheme.
 * Test infrastructure only (see oracle.h).
 * Follows B/direct_ba_pcg.cc:43-819 (driver) and B/kernel_pcg.cu:44-1389 (kernels).
 * PCGScalar = float (B/kernels.cuh:62).
 *
 * DEFINITION of the sums (shared with the backend, kernels_pcg.hip).  The reference forms every dense entry of r, M and g and
 * every dot product with block reductions + binary32 atomics in arbitrary order (B/kernel_pcg.cu:98-154), so its conjugate
 * gradient is not reproducible run to run (FIX).  Here:
 *   - a surfel's own entries are binary32 chains over the keyframes in ascending order (as in the reference, whose launches
 *     are ordered by keyframe);
 *   - the 6 pose entries of a keyframe, the 5 + 4 global intrinsics entries and the pair part of alpha_d: per (64-surfel tile,
 *     keyframe) the per-surfel binary32 contributions of that keyframe are added by the fixed tree of orc_tile_tree_sum, and
 *     these (tile, keyframe) totals are summed EXACTLY (oracle_exact.c) -- so the sums do not depend on whether the
 *     keyframes are swept in one pass or one call per keyframe (B/direct_ba_pcg.cc does the latter);
 *   - per-cell cfactor entries: the per-pair binary32 terms summed exactly;
 *   - dot products over the unknowns (alpha_n, beta_n, the epsilon terms of alpha_d): the binary32 products summed exactly;
 * every exact sum is rounded once to binary64 and from there to binary32 where it is stored in a PCGScalar.  Exact sums do
 * not depend on the order or grouping of their terms, so the backend's atomics, launch shapes and surfel sharding over GPUs
 * (an integer all-reduce of the limbs) produce the same bits, and so does the OpenMP loop below. */
#include "oracle_internal.h"

typedef float pcg_real;

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
      const float exp_inv_depth = orc_exp(-dp->a * raw_inv_depth);
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


/* ---- the dense head: every unknown outside the surfel block ---- */
enum { HOT_A = 0 /* 9: r or g of the global intrinsics */, HOT_B = 9 /* 9: M */, HOT_ALPHA_D = 18, HOT_SLOTS = 19 };
typedef struct {
  uint32_t lo, hi, count;   /* head = [0, lo) + [hi, U) */
  orc_exact* a;             /* r (init) / g (step 1) */
  orc_exact* b;             /* M (init) */
  orc_exact hot[HOT_SLOTS];
  int invalid;
} pcg_head;
static uint32_t head_index(const pcg_head* H, uint32_t u) { return u < H->lo ? u : H->lo + (u - H->hi); }
static void head_setup(pcg_head* H, const pcg_layout* L, uint32_t surfels_size) {
  const uint32_t U = L->unknown_count;
  H->lo = L->optimize_geometry ? L->surfel_start : U;
  H->hi = L->optimize_geometry ? L->surfel_start + (uint32_t)L->geom_stride * surfels_size : U;
  H->count = H->lo + (U - H->hi);
  H->a = (orc_exact*)calloc(H->count ? H->count : 1, sizeof(orc_exact));
  H->b = (orc_exact*)calloc(H->count ? H->count : 1, sizeof(orc_exact));
  memset(H->hot, 0, sizeof(H->hot));
  H->invalid = 0;
}
static void head_free(pcg_head* H) { free(H->a); free(H->b); }
static float exact_f32(const orc_exact* c, int invalid) { return invalid ? NAN : (float)orc_exact_value(c); }
/* slot of a global intrinsics unknown among HOT_A.. (0..4 depth, 5..8 colour), or -1 */
static int intrinsics_slot(const pcg_layout* L, uint32_t u) {
  if (L->optimize_depth_intrinsics && u >= L->depth_intr_start && u < L->depth_intr_start + 5) return (int)(u - L->depth_intr_start);
  if (L->optimize_color_intrinsics && u >= L->color_intr_start && u < L->color_intr_start + 4) return 5 + (int)(u - L->color_intr_start);
  return -1;
}
/* head accumulators -> the head entries of the PCGScalar vectors; the accumulators are cleared */
static void head_resolve(pcg_head* H, const pcg_layout* L, pcg_real* va, pcg_real* vb) {
  const uint32_t U = L->unknown_count;
  for (uint32_t u = 0; u < U; ++u) {
    if (u >= H->lo && u < H->hi) { u = H->hi - 1; continue; }
    const uint32_t h = head_index(H, u);
    const int slot = intrinsics_slot(L, u);
    va[u] = exact_f32(slot >= 0 ? &H->hot[HOT_A + slot] : &H->a[h], H->invalid);
    if (vb) vb[u] = exact_f32(slot >= 0 ? &H->hot[HOT_B + slot] : &H->b[h], H->invalid);
  }
  memset(H->a, 0, sizeof(orc_exact) * (H->count ? H->count : 1));
  memset(H->b, 0, sizeof(orc_exact) * (H->count ? H->count : 1));
  memset(H->hot, 0, sizeof(orc_exact) * HOT_ALPHA_D);
}

typedef struct {
  const pcg_layout* L;
  int K, gauge;
  orc_keyframe* const* kfs;
  const orc_camera* color_cam; const orc_camera* depth_cam;
  const orc_depth_params* dp;
  const orc_surfels* s;
  proj_params* pp;          /* one per keyframe */
  depth_to_color d2c;
} pcg_sweep;
static uint32_t kf_pose_index(const pcg_sweep* w, int k) {   /* B/direct_ba_pcg.cc:329-337 */
  if (k == w->gauge) return INVALID_UNKNOWN;
  return (k < w->gauge) ? 6u * (uint32_t)k : 6u * (uint32_t)(k - 1);
}

/* PCGInit over all keyframes (B/kernel_pcg.cu:179-541): r -= J^T W F, M += diag(J^T W J). */
static void pcg_init_sweep(const pcg_sweep* w, pcg_real* r_, pcg_real* M_, pcg_head* H) {
  const pcg_layout* L = w->L;
  const orc_surfels* s = w->s;
  const long tiles = (long)((s->surfels_size + 63u) / 64u);
#pragma omp parallel for schedule(dynamic, 8)
  for (long tile_index = 0; tile_index < tiles; ++tile_index) {
    const uint32_t tile = (uint32_t)tile_index * 64u;
    float gr[3][64], gM[3][64];
    memset(gr, 0, sizeof(gr)); memset(gM, 0, sizeof(gM));
    for (int k = 0; k < w->K; ++k) {
      const orc_keyframe* kf = w->kfs[k];
      const int pose_kf = L->optimize_poses && k != w->gauge;
      float pr[6][64], pM[6][64], ir[9][64], iM[9][64];   /* this keyframe's terms */
      int any = 0;
      memset(pr, 0, sizeof(pr)); memset(pM, 0, sizeof(pM)); memset(ir, 0, sizeof(ir)); memset(iM, 0, sizeof(iM));
      for (uint32_t lane = 0; lane < 64 && tile + lane < s->surfels_size; ++lane) {
        const uint32_t i = tile + lane;
        proj_result pres;
        if (!orc_project_associate(&w->pp[k], i, &pres, NULL)) continue;
        any = 1;
        pair_terms t;
        eval_pair_terms(L, w->color_cam, w->depth_cam, w->dp, kf, &w->pp[k], &w->d2c, s, i, &pres, &t);
        int visible = 1;
        if (L->use_depth) {
          if (L->optimize_geometry) {
            gr[0][lane] -= t.Jgeom * t.w * t.raw;
            gM[0][lane] += t.Jgeom * t.w * t.Jgeom;
          }
          if (pose_kf)
            for (int c = 0; c < 6; ++c) { const float wj = t.w * t.Jpose[c]; pr[c][lane] += -1 * wj * t.raw; pM[c][lane] += t.Jpose[c] * wj; }
          if (L->optimize_depth_intrinsics) {
            if (!t.di_valid) visible = 0;   /* B/kernel_pcg.cu:272-274: also hides the descriptor part */
            if (visible) {
              for (int c = 0; c < 5; ++c) { const float wj = t.w * t.Jdi[c]; ir[c][lane] += -1 * wj * t.raw; iM[c][lane] += t.Jdi[c] * wj; }
              const float wj = t.w * t.Jcf;
              const uint32_t h = head_index(H, t.cf_index);
              orc_exact_add(&H->a[h], -1 * wj * t.raw, &H->invalid);
              orc_exact_add(&H->b[h], t.Jcf * wj, &H->invalid);
            }
          }
        }
        if (L->use_desc && visible && t.color_ok) {
          if (L->optimize_geometry) {
            gr[0][lane] -= t.Jg1 * t.w1 * t.raw1 + t.Jg2 * t.w2 * t.raw2;
            gM[0][lane] += t.Jg1 * t.w1 * t.Jg1 + t.Jg2 * t.w2 * t.Jg2;
            gr[1][lane] -= -1.f * t.w1 * t.raw1 + 0.f * t.w2 * t.raw2;
            gM[1][lane] += -1.f * t.w1 * -1.f + 0.f * t.w2 * 0.f;
            gr[2][lane] -= 0.f * t.w1 * t.raw1 + -1.f * t.w2 * t.raw2;
            gM[2][lane] += 0.f * t.w1 * 0.f + -1.f * t.w2 * -1.f;
          }
          if (pose_kf)
            for (int c = 0; c < 6; ++c) {
              const float wj1 = t.w1 * t.Jp1[c], wj2 = t.w2 * t.Jp2[c];
              pr[c][lane] += -1 * wj1 * t.raw1 + -1 * wj2 * t.raw2;
              pM[c][lane] += t.Jp1[c] * wj1 + t.Jp2[c] * wj2;
            }
          if (L->optimize_color_intrinsics)
            for (int c = 0; c < 4; ++c) {
              const float wj1 = t.w1 * t.Jci1[c], wj2 = t.w2 * t.Jci2[c];
              ir[5 + c][lane] += -1 * wj1 * t.raw1 + -1 * wj2 * t.raw2;
              iM[5 + c][lane] += t.Jci1[c] * wj1 + t.Jci2[c] * wj2;
            }
        }
      }
      if (pose_kf && any) {
        const uint32_t base = kf_pose_index(w, k);   /* pose unknowns come first: head index == unknown index */
        for (int c = 0; c < 6; ++c) {
          orc_exact_add(&H->a[base + c], orc_tile_tree_sum(pr[c]), &H->invalid);
          orc_exact_add(&H->b[base + c], orc_tile_tree_sum(pM[c]), &H->invalid);
        }
      }
      if (any && (L->optimize_depth_intrinsics || L->optimize_color_intrinsics))
        for (int q = 0; q < 9; ++q) {
          if ((q < 5) ? !L->optimize_depth_intrinsics : !L->optimize_color_intrinsics) continue;
          orc_exact_add(&H->hot[HOT_A + q], orc_tile_tree_sum(ir[q]), &H->invalid);
          orc_exact_add(&H->hot[HOT_B + q], orc_tile_tree_sum(iM[q]), &H->invalid);
        }
    }
    for (uint32_t lane = 0; lane < 64 && tile + lane < s->surfels_size; ++lane) {
      if (!L->optimize_geometry) break;
      const uint32_t gi = L->surfel_start + (uint32_t)L->geom_stride * (tile + lane);
      for (int c = 0; c < L->geom_stride; ++c) { r_[gi + c] = gr[c][lane]; M_[gi + c] = gM[c][lane]; }
    }
  }
  head_resolve(H, L, r_, M_);
}

/* PCGStep1 over all keyframes (B/kernel_pcg.cu:646-1026): g = J^T W J p; returns the pair part of alpha_d = p^T J^T W J p
 * as its exactly rounded binary64 value (the epsilon terms are added by the caller). */
static double pcg_step1_sweep(const pcg_sweep* w, const pcg_real* p_, pcg_real* g_, pcg_head* H) {
  const pcg_layout* L = w->L;
  const orc_surfels* s = w->s;
  const long tiles = (long)((s->surfels_size + 63u) / 64u);
  float pdi[5] = {0, 0, 0, 0, 0}, pci[4] = {0, 0, 0, 0};
  if (L->optimize_depth_intrinsics) for (int c = 0; c < 5; ++c) pdi[c] = p_[L->depth_intr_start + c];
  if (L->optimize_color_intrinsics) for (int c = 0; c < 4; ++c) pci[c] = p_[L->color_intr_start + c];
#pragma omp parallel for schedule(dynamic, 8)
  for (long tile_index = 0; tile_index < tiles; ++tile_index) {
    const uint32_t tile = (uint32_t)tile_index * 64u;
    float gs[3][64];
    memset(gs, 0, sizeof(gs));
    for (int k = 0; k < w->K; ++k) {
      float gia[9][64], ad[64];   /* this keyframe's terms */
      memset(gia, 0, sizeof(gia)); memset(ad, 0, sizeof(ad));
      const orc_keyframe* kf = w->kfs[k];
      const int pose_kf = L->optimize_poses && k != w->gauge;
      const uint32_t base = kf_pose_index(w, k);
      float pp6[6] = {0, 0, 0, 0, 0, 0};
      if (pose_kf) for (int c = 0; c < 6; ++c) pp6[c] = p_[base + c];
      float gpose[6][64];
      int any = 0;
      memset(gpose, 0, sizeof(gpose));
      for (uint32_t lane = 0; lane < 64 && tile + lane < s->surfels_size; ++lane) {
        const uint32_t i = tile + lane;
        proj_result pres;
        if (!orc_project_associate(&w->pp[k], i, &pres, NULL)) continue;
        any = 1;
        pair_terms t;
        eval_pair_terms(L, w->color_cam, w->depth_cam, w->dp, kf, &w->pp[k], &w->d2c, s, i, &pres, &t);
        const uint32_t gi = L->optimize_geometry ? L->surfel_start + (uint32_t)L->geom_stride * i : 0u;
        float ps[3] = {0, 0, 0};
        if (L->optimize_geometry) for (int c = 0; c < L->geom_stride; ++c) ps[c] = p_[gi + c];
        if (L->use_depth) {
          float sum = 0;
          if (L->optimize_geometry) sum += t.Jgeom * ps[0];
          if (pose_kf) for (int c = 0; c < 6; ++c) sum += t.Jpose[c] * pp6[c];
          const int di = L->optimize_depth_intrinsics && t.di_valid;
          if (di) {
            sum += t.Jdi[2] * pdi[2];
            sum += t.Jdi[3] * pdi[3];
            sum += t.Jdi[0] * pdi[0];
            sum += t.Jdi[1] * pdi[1];
            sum += t.Jdi[4] * pdi[4];
            sum += t.Jcf * p_[t.cf_index];
          }
          ad[lane] += sum * t.w * sum;
          sum *= t.w;
          if (L->optimize_geometry) gs[0][lane] += t.Jgeom * sum;
          if (pose_kf) for (int c = 0; c < 6; ++c) gpose[c][lane] += t.Jpose[c] * sum;
          if (di) {
            for (int c = 0; c < 5; ++c) gia[c][lane] += t.Jdi[c] * sum;
            orc_exact_add(&H->a[head_index(H, t.cf_index)], t.Jcf * sum, &H->invalid);
          }
        }
        if (L->use_desc && t.color_ok) {
          float sum1 = 0, sum2 = 0;
          if (L->optimize_geometry) {
            sum1 += t.Jg1 * ps[0]; sum2 += t.Jg2 * ps[0];
            sum1 += -1.f * ps[1];
            sum2 += -1.f * ps[2];
          }
          if (pose_kf) for (int c = 0; c < 6; ++c) { sum1 += t.Jp1[c] * pp6[c]; sum2 += t.Jp2[c] * pp6[c]; }
          if (L->optimize_color_intrinsics) for (int c = 0; c < 4; ++c) { sum1 += t.Jci1[c] * pci[c]; sum2 += t.Jci2[c] * pci[c]; }
          ad[lane] += sum1 * t.w1 * sum1 + sum2 * t.w2 * sum2;
          sum1 *= t.w1; sum2 *= t.w2;
          if (L->optimize_geometry) {
            gs[0][lane] += t.Jg1 * sum1 + t.Jg2 * sum2;
            gs[1][lane] += -1.f * sum1 + 0.f * sum2;
            gs[2][lane] += 0.f * sum1 + -1.f * sum2;
          }
          if (pose_kf) for (int c = 0; c < 6; ++c) gpose[c][lane] += t.Jp1[c] * sum1 + t.Jp2[c] * sum2;
          if (L->optimize_color_intrinsics) for (int c = 0; c < 4; ++c) gia[5 + c][lane] += t.Jci1[c] * sum1 + t.Jci2[c] * sum2;
        }
      }
      if (pose_kf && any)
        for (int c = 0; c < 6; ++c) orc_exact_add(&H->a[base + c], orc_tile_tree_sum(gpose[c]), &H->invalid);
      if (any) {
        for (int q = 0; q < 9; ++q) {
          if ((q < 5) ? !L->optimize_depth_intrinsics : !L->optimize_color_intrinsics) continue;
          orc_exact_add(&H->hot[HOT_A + q], orc_tile_tree_sum(gia[q]), &H->invalid);
        }
        orc_exact_add(&H->hot[HOT_ALPHA_D], orc_tile_tree_sum(ad), &H->invalid);
      }
    }
    for (uint32_t lane = 0; lane < 64 && tile + lane < s->surfels_size; ++lane) {
      if (!L->optimize_geometry) break;
      const uint32_t gi = L->surfel_start + (uint32_t)L->geom_stride * (tile + lane);
      for (int c = 0; c < L->geom_stride; ++c) g_[gi + c] = gs[c][lane];
    }
  }
  const double pairs = H->invalid ? (double)NAN : orc_exact_value(&H->hot[HOT_ALPHA_D]);
  memset(&H->hot[HOT_ALPHA_D], 0, sizeof(orc_exact));
  head_resolve(H, L, g_, NULL);
  return pairs;
}

static inline float prior_at(const pcg_layout* L, uint32_t idx) {
  return (idx == L->a_index) ? (kAPriorWeight * kAPriorWeight) : 0.f;
}
/* sum_u (epsilon + prior_u) p_u^2, exact (AddAlphaDEpsilonTerms, B/kernel_pcg.cu:1028-1050) */
static double pcg_eps_terms(const pcg_layout* L, const pcg_real* p_) {
  orc_exact acc; int invalid = 0;
  memset(&acc, 0, sizeof(acc));
  for (uint32_t u = 0; u < L->unknown_count; ++u) {
    const float pv = p_[u];
    orc_exact_add(&acc, (kDiagEpsilon + prior_at(L, u)) * pv * pv, &invalid);
  }
  return invalid ? (double)NAN : orc_exact_value(&acc);
}

static void make_layout(pcg_layout* L, const orc_ba_options* opt, int K, uint32_t surfels_size, int S) {
  memset(L, 0, sizeof(*L));
  L->use_depth = opt->use_depth_residuals; L->use_desc = opt->use_descriptor_residuals;
  L->optimize_poses = opt->optimize_poses; L->optimize_geometry = opt->optimize_geometry;
  L->optimize_depth_intrinsics = opt->optimize_depth_intrinsics && L->use_depth;   /* B/direct_ba.cc:427-434 */
  L->optimize_color_intrinsics = opt->optimize_color_intrinsics && L->use_desc;
  L->geom_stride = L->use_desc ? 3 : 1;
  /* unknown layout (B/direct_ba_pcg.cc:232-307) */
  uint32_t cur = 0;
  if (L->optimize_poses) cur += 6u * (uint32_t)(K - 1);
  L->surfel_start = INVALID_UNKNOWN;
  if (L->optimize_geometry) { L->surfel_start = cur; cur += (uint32_t)L->geom_stride * surfels_size; }
  L->depth_intr_start = INVALID_UNKNOWN; L->a_index = INVALID_UNKNOWN;
  if (L->optimize_depth_intrinsics) { L->depth_intr_start = cur; cur += 5u + (uint32_t)S; L->a_index = L->depth_intr_start + 4; }
  L->color_intr_start = INVALID_UNKNOWN;
  if (L->optimize_color_intrinsics) { L->color_intr_start = cur; cur += 4; }
  L->unknown_count = cur;
}
static void sweep_setup(pcg_sweep* w, const pcg_layout* L, orc_ba_state* st, int gauge) {
  w->L = L; w->K = st->num_kfs; w->gauge = gauge; w->kfs = st->kfs;
  w->color_cam = &st->color_cam; w->depth_cam = &st->depth_cam; w->dp = &st->dp; w->s = st->surfels;
  w->pp = (proj_params*)calloc((size_t)(w->K > 0 ? w->K : 1), sizeof(proj_params));
  for (int k = 0; k < w->K; ++k) w->pp[k] = make_proj_params(&st->depth_cam, &st->dp, st->surfels, st->kfs[k], st->kfs[k]->frame_T_global);
  w->d2c = make_depth_to_color(&st->depth_cam, &st->color_cam);
}

/* Test hook: assembles r = -J^T W F and M = diag(J^T W J) (PCGInit over all keyframes, without the
 * prior on a) for the current state.  Returns the unknown count; writes at most `capacity` entries. */
uint32_t orc_pcg_assemble(orc_ba_state* st, const orc_ba_options* opt, float* r_out, float* M_out, uint32_t capacity) {
  const orc_surfels* s = st->surfels;
  const int K = st->num_kfs;
  pcg_layout L;
  make_layout(&L, opt, K, s->surfels_size, st->dp.cf_width * st->dp.cf_height);
  const uint32_t cur = L.unknown_count;
  pcg_real* r_ = (pcg_real*)calloc(cur ? cur : 1, sizeof(pcg_real));
  pcg_real* M_ = (pcg_real*)calloc(cur ? cur : 1, sizeof(pcg_real));
  const int gauge = (opt->pcg_gauge_keyframe >= 0 && opt->pcg_gauge_keyframe < K) ? opt->pcg_gauge_keyframe : 0;
  pcg_sweep w; pcg_head H;
  sweep_setup(&w, &L, st, gauge);
  head_setup(&H, &L, s->surfels_size);
  pcg_init_sweep(&w, r_, M_, &H);
  head_free(&H); free(w.pp);
  const uint32_t n = cur < capacity ? cur : capacity;
  for (uint32_t u = 0; u < n; ++u) { r_out[u] = r_[u]; M_out[u] = M_[u]; }
  free(r_); free(M_);
  return cur;
}

void orc_bundle_adjustment_pcg(orc_ba_state* st, const orc_ba_options* opt, orc_ba_stats* stats) {
  orc_surfels* s = st->surfels;
  memset(stats, 0, sizeof(*stats));
  const int K = st->num_kfs;
  pcg_layout L;
  make_layout(&L, opt, K, 0, 0);
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
          const uint32_t size_before = s->surfels_size;
          orc_create_surfels_for_keyframe(1, opt->min_observation_count, &st->color_cam, &st->depth_cam, &st->dp, kf, st->kfs,
                                          covis, n_covis, s, st->supporting);
          st->unsorted_surfels += s->surfels_size - size_before;
          free(all);
          new_kfs[n_new++] = k;
        } else if (kf->activation == ORC_KF_COVIS_ACTIVE && kf->last_covis_in_ba_iteration != st->ba_iteration_count) {
          kf->last_covis_in_ba_iteration = st->ba_iteration_count;
        }
      }
    }
    memset(s->active, ORC_SURFEL_ACTIVE_FLAG, s->surfels_size);
    if (opt->optimize_geometry) orc_update_surfel_normals(&st->depth_cam, &st->dp, st->kfs, K, s);

    make_layout(&L, opt, K, s->surfels_size, S);
    const uint32_t U = L.unknown_count;
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
    pcg_sweep w; pcg_head H;
    sweep_setup(&w, &L, st, gauge);
    head_setup(&H, &L, s->surfels_size);
    pcg_init_sweep(&w, r_, M_, &H);

    /* PCGInit2, B/kernel_pcg.cu:565-600 */
    pcg_real alpha_n = 0, alpha_d = 0, beta_n = 0;
    {
      orc_exact acc; int invalid = 0;
      memset(&acc, 0, sizeof(acc));
      for (uint32_t u = 0; u < U; ++u) {
        g_[u] = 0;
        const pcg_real r_value = r_[u] + ((u == L.a_index) ? (-kAPriorWeight * kAPriorWeight * st->dp.a) : 0);
        const pcg_real p_value = r_value / (M_[u] + kDiagEpsilon + prior_at(&L, u));
        p_[u] = p_value;
        delta[u] = 0;
        orc_exact_add(&acc, r_value * p_value, &invalid);
      }
      alpha_n = exact_f32(&acc, invalid);
    }

    double prev_r_norm = INFINITY;
    int no_improvement = 0;
    for (int step = 0; step < opt->pcg_max_inner_iterations; ++step) {
      stats->pcg_inner_steps_total += 1;
      if (step > 0) { const pcg_real tmp = alpha_n; alpha_n = beta_n; beta_n = tmp; }
      memset(g_, 0, sizeof(pcg_real) * U);
      const double pairs = pcg_step1_sweep(&w, p_, g_, &H);
      /* AddAlphaDEpsilonTerms runs once per keyframe (B/kernel_pcg.cu:1102-1112): the term enters K times -- reproduced */
      const double eps_terms = (s->surfels_size > 0) ? pcg_eps_terms(&L, p_) : 0.0;
      alpha_d = (pcg_real)(pairs + (double)K * eps_terms);
      /* PCGStep2, B/kernel_pcg.cu:1117-1158 */
      const pcg_real alpha = (alpha_d >= 1e-35f) ? (alpha_n / alpha_d) : 0;
      {
        orc_exact acc; int invalid = 0;
        memset(&acc, 0, sizeof(acc));
        for (uint32_t u = 0; u < U; ++u) {
          const pcg_real p_value = p_[u];
          delta[u] += alpha * p_value;
          pcg_real r_value = r_[u];
          r_value -= alpha * (g_[u] + (kDiagEpsilon + prior_at(&L, u)) * p_value);
          r_[u] = r_value;
          const pcg_real z_value = r_value / (M_[u] + kDiagEpsilon + prior_at(&L, u));
          g_[u] = z_value;
          orc_exact_add(&acc, z_value * r_value, &invalid);
        }
        beta_n = exact_f32(&acc, invalid);
      }
      /* B/direct_ba_pcg.cc:441-456: PCGScalar r_norm = sqrt(beta_n); the comparison is evaluated in double */
      const pcg_real r_norm = sqrtf(beta_n);
      if ((double)r_norm < prev_r_norm - 1e-3) no_improvement = 0;
      else if (++no_improvement >= 3) break;
      prev_r_norm = (double)r_norm;
      if (step < opt->pcg_max_inner_iterations - 1) {
        /* PCGStep3, B/kernel_pcg.cu:1212-1226 */
        const pcg_real beta = (alpha_n >= 1e-35f) ? (beta_n / alpha_n) : 0;
        for (uint32_t u = 0; u < U; ++u) p_[u] = g_[u] + beta * p_[u];
      }
    }
    head_free(&H); free(w.pp);

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
      if (n_new > 0) { st->unsorted_surfels += s->surfels_size - s->surfel_count; orc_compact_surfels(s); orc_sort_after_in_loop_compaction(st); }
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
    if (n_new > 0) { st->unsorted_surfels += s->surfels_size - s->surfel_count; orc_compact_surfels(s); orc_sort_after_in_loop_compaction(st); }
  }
  free(new_kfs);
}
