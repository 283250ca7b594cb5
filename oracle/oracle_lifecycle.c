/* oracle_lifecycle.c -- surfel creation, merging, deletion + radius update, compaction.
 * Test infrastructure only (see oracle.h).
 *
 * Determinism: the reference resolves per-cell ownership with atomicCAS, so the winner among
 * competing pixels / surfels is arbitrary (B/kernel_create_surfels.cu:56-68,
 * B/kernel_supporting_surfels.cu:61).  The oracle (and the HIP path) fix the winner: lowest
 * linear pixel index for creation, ascending surfel index for the supporting-surfel slots. */
#include "oracle_internal.h"

static const uint32_t kDeletedBits = 0x7fffffffu; /* CUDART_NAN_F, B/kernel_delete_surfels.cu:145-146 */
static inline int is_deleted(const orc_surfels* s, uint32_t i) {
  uint32_t b; memcpy(&b, &srow(s, ORC_SURFEL_X)[i], 4);
  return b == kDeletedBits;
}
static inline void mark_deleted(orc_surfels* s, uint32_t i) {
  memcpy(&srow(s, ORC_SURFEL_X)[i], &kDeletedBits, 4);
}

/* B/kernel_supporting_surfels.cc:40-110, kernel B/kernel_supporting_surfels.cu:45-97 */
void orc_determine_supporting_surfels(int merge, float merge_dist_factor, const orc_camera* depth_cam,
                                      const orc_depth_params* dp, const orc_keyframe* kf,
                                      orc_surfels* s, uint32_t* supporting) {
  const size_t plane = (size_t)kf->width * kf->height;
  for (size_t i = 0; i < ORC_MERGE_BUFFER_COUNT * plane; ++i) supporting[i] = ORC_INVALID_INDEX;
  if (s->surfels_size == 0) return;
  const float cell_merge_dist_squared = dp->cell * dp->cell * merge_dist_factor * merge_dist_factor;
  const float cos_thr = ORC_COS_NORMAL_COMPAT;
  proj_params p = make_proj_params(depth_cam, dp, s, kf, kf->frame_T_global);
  uint32_t deleted = 0;
  for (uint32_t i = 0; i < s->surfels_size; ++i) {
    proj_result r;
    if (!orc_project_associate(&p, i, &r, NULL)) continue;
    const int cx = r.px / dp->cell, cy = r.py / dp->cell;
    for (int b = 0; b < ORC_MERGE_BUFFER_COUNT; ++b) {
      uint32_t* slot = &supporting[b * plane + (size_t)cy * kf->width + cx];
      const uint32_t sup_index = *slot;
      if (sup_index == ORC_INVALID_INDEX) { *slot = i; break; }
      if (merge) {
        const v3 sup_normal = surfel_normal(s, sup_index);
        const v3 this_normal = surfel_normal(s, i);
        if (v3_dot(sup_normal, this_normal) > cos_thr) {
          const v3 sp = surfel_position(s, sup_index), tp = surfel_position(s, i);
          const float min_radius_sq = fminf(srow(s, ORC_SURFEL_RADIUS_SQ)[sup_index], srow(s, ORC_SURFEL_RADIUS_SQ)[i]);
          const v3 d = v3_sub(sp, tp);
          if (d.x * d.x + d.y * d.y + d.z * d.z < min_radius_sq * cell_merge_dist_squared) {
            mark_deleted(s, i);
            deleted += 1; /* NOTE: like the reference, no break: later slots are still visited
                             (with a NaN position the distance test can no longer pass). */
          }
        }
      }
    }
  }
  if (merge) s->surfel_count -= deleted;
}

/* B/kernel_create_surfels.cu:91-160 */
static void create_new_surfel(int x, int y, uint32_t surfel_index, const orc_camera* color_cam,
                              const orc_camera* depth_cam, const orc_depth_params* dp, const orc_keyframe* kf,
                              orc_surfels* s) {
  const unprojector unp = make_unprojector(depth_cam);
  const depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
  float G[12];
  orc_se3_matrix3x4(&kf->global_T_frame, G);
  const size_t idx = (size_t)y * kf->width + x;
  const float calibrated_depth = orc_raw_to_calibrated_depth(dp->a, cfactor_at(dp, x, y), dp->raw_to_float_depth, kf->depth[idx]);
  const v3 gp = m34_mul(G, unp_point(&unp, x, y, calibrated_depth));
  surfel_set_position(s, surfel_index, gp);
  float m[3];
  orc_unpack_normal8(kf->normals[idx], m);
  const v3 gn = m34_rotate(G, v3_make(m[0], m[1], m[2]));
  surfel_set_normal(s, surfel_index, gn);
  const float radius_sq = orc_half_to_float(kf->radius[idx]);
  srow(s, ORC_SURFEL_RADIUS_SQ)[surfel_index] = radius_sq;
  float c[2];
  transform_depth_to_color(x + 0.5f, y + 0.5f, &d2c, &c[0], &c[1]);
  /* colour: bilinear RGB sample, truncated to u8 (B/kernel_create_surfels.cu:124-155) */
  uint8_t rgba[4] = {0, 0, 0, 0};
  {
    /* per-channel bilinear fetch with the same sampler geometry as orc_sample_luma */
    float xb = c[0] - 0.5f, yb = c[1] - 0.5f;
    const int w = kf->color_width, h = kf->color_height;
    if (!(xb >= -1.f)) xb = -1.f; if (xb > (float)w) xb = (float)w;
    if (!(yb >= -1.f)) yb = -1.f; if (yb > (float)h) yb = (float)h;
    const float fx = floorf(xb), fy = floorf(yb), a = xb - fx, b = yb - fy;
    int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    if (x0 < 0) x0 = 0; if (x0 > w - 1) x0 = w - 1; if (x1 < 0) x1 = 0; if (x1 > w - 1) x1 = w - 1;
    if (y0 < 0) y0 = 0; if (y0 > h - 1) y0 = h - 1; if (y1 < 0) y1 = 0; if (y1 > h - 1) y1 = h - 1;
    for (int ch = 0; ch < 3; ++ch) {
      const float tl = kf->color[4 * ((size_t)y0 * w + x0) + ch] * (1.0f / 255.0f);
      const float tr = kf->color[4 * ((size_t)y0 * w + x1) + ch] * (1.0f / 255.0f);
      const float bl = kf->color[4 * ((size_t)y1 * w + x0) + ch] * (1.0f / 255.0f);
      const float br = kf->color[4 * ((size_t)y1 * w + x1) + ch] * (1.0f / 255.0f);
      const float top = tl + a * (tr - tl), bot = bl + a * (br - bl);
      rgba[ch] = (uint8_t)(255.f * (top + b * (bot - top)));
    }
  }
  memcpy(&srow(s, ORC_SURFEL_COLOR)[surfel_index], rgba, 4);
  float t1[2], t2[2], d1, d2;
  orc_tangent_projections(gp, v3_make(gn.x, gn.y, gn.z), radius_sq, kf->frame_T_global, color_cam, t1, t2);
  orc_raw_descriptor_residual(kf, c, t1, t2, 0, 0, &d1, &d2);
  srow(s, ORC_SURFEL_DESC1)[surfel_index] = d1;
  srow(s, ORC_SURFEL_DESC2)[surfel_index] = d2;
}

/* Pixel-defined surfel association, B/surfel_projection_nvcc_only.cuh:131-231, used by
 * CountObservationsForNewSurfels (B/kernel_create_surfels.cu:213-276). */
static int pixel_surfel_associated(v3 lp, v3 nl, const orc_keyframe* covis, const orc_depth_params* dp,
                                   const unprojector* unp, int px, int py, int* fsv) {
  const size_t idx = (size_t)py * covis->width + px;
  const uint16_t raw = covis->depth[idx];
  if (raw & ORC_INVALID_DEPTH_BIT) return 0;
  const float d = orc_raw_to_calibrated_depth(dp->a, cfactor_at(dp, px, py), dp->raw_to_float_depth, raw);
  const float thr = 10.f * depth_stddev(unp_nx(unp, (float)px), unp_ny(unp, (float)py), d, nl, dp->baseline_fx);
  const float diff = d - lp.z;
  if (diff > thr) { *fsv = 1; return 0; }
  else if (diff < -thr) return 0;
  if (v3_dot(lp, nl) > 0) return 0;   /* sign of (1 / |p|) * dot(p, n), B/surfel_projection_nvcc_only.cuh:216-220; see orc_project_associate */
  float m[3];
  orc_unpack_normal8(covis->normals[idx], m);
  if (v3_dot(nl, v3_make(m[0], m[1], m[2])) < ORC_COS_NORMAL_COMPAT) return 0;
  return 1;
}

/* B/direct_ba.cc:340-405 + B/kernel_create_surfels.cc:40-183 */
uint32_t orc_create_surfels_for_keyframe(int filter_new_surfels, int min_observation_count,
                                         const orc_camera* color_cam, const orc_camera* depth_cam,
                                         const orc_depth_params* dp, const orc_keyframe* kf,
                                         orc_keyframe* const* kfs, const int* covis, int n_covis,
                                         orc_surfels* s, uint32_t* supporting) {
  orc_determine_supporting_surfels(0, 0, depth_cam, dp, kf, s, supporting);
  const int W = kf->width, H = kf->height;
  uint8_t* flag = (uint8_t*)calloc((size_t)W * H, 1);
  /* B/kernel_create_surfels.cu:41-75: first qualifying pixel (row-major) claims the cell. */
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) {
      if (!(x >= 1 && y >= 1 && x < W - 1 && y < H - 1)) continue;
      if (kf->depth[(size_t)y * W + x] & ORC_INVALID_DEPTH_BIT) continue;
      uint32_t* slot = &supporting[(size_t)(y / dp->cell) * W + (x / dp->cell)];
      if (*slot == ORC_INVALID_INDEX) { *slot = 0; flag[(size_t)y * W + x] = 1; }
    }
  }
  if (filter_new_surfels) {
    const unprojector unp = make_unprojector(depth_cam);
    /* covis_T_frame = covis.frame_T_global * kf.global_T_frame (B/direct_ba.cc:359-365); one per co-visible keyframe */
    float* rel_M = (float*)malloc(sizeof(float) * 12 * (n_covis ? n_covis : 1));
    for (int c = 0; c < n_covis; ++c) {
      orc_se3 cinv, rel;
      orc_se3_inverse(&kfs[covis[c]]->global_T_frame, &cinv);
      orc_se3_mul(&cinv, &kf->global_T_frame, &rel);
      orc_se3_matrix3x4(&rel, rel_M + 12 * c);
    }
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y) {
      for (int x = 0; x < W; ++x) {
        const size_t idx = (size_t)y * W + x;
        if (!flag[idx]) continue;
        uint32_t observations = 1, violations = 0;
        const float cd = orc_raw_to_calibrated_depth(dp->a, cfactor_at(dp, x, y), dp->raw_to_float_depth, kf->depth[idx]);
        const v3 input_pos = unp_point(&unp, x, y, cd);
        float m[3];
        orc_unpack_normal8(kf->normals[idx], m);
        for (int c = 0; c < n_covis; ++c) {
          const orc_keyframe* ck = kfs[covis[c]];
          const float* M = rel_M + 12 * c;
          v3 lp;
          lp.z = M[8] * input_pos.x + M[9] * input_pos.y + M[10] * input_pos.z + M[11];
          if (!(lp.z > 0.f)) continue;
          lp.x = M[0] * input_pos.x + M[1] * input_pos.y + M[2] * input_pos.z + M[3];
          lp.y = M[4] * input_pos.x + M[5] * input_pos.y + M[6] * input_pos.z + M[7];
          const float pxx = depth_cam->fx * (lp.x / lp.z) + depth_cam->cx;
          const float pxy = depth_cam->fy * (lp.y / lp.z) + depth_cam->cy;
          if (!(pxx >= 0.f) || !(pxy >= 0.f) || !(pxx < (float)ck->width) || !(pxy < (float)ck->height)) continue;
          const v3 nl = m34_rotate(M, v3_make(m[0], m[1], m[2]));
          int fsv = 0;
          if (pixel_surfel_associated(lp, nl, ck, dp, &unp, (int)pxx, (int)pxy, &fsv)) observations += 1;
          else if (fsv) violations += 1;
        }
        if (observations < (uint32_t)min_observation_count || violations > observations) flag[idx] = 0;
      }
    }
    free(rel_M);
  }
  uint32_t count = 0;
  for (size_t i = 0; i < (size_t)W * H; ++i) count += flag[i];
  if (count == 0 || s->surfels_size + count > s->capacity) { free(flag); return 0; }
  /* Append order.  The reference numbers new surfels by a prefix sum over the row-major pixel
   * index (B/kernel_create_surfels.cu:357-390); the order is not observable in its results.  This
   * backend numbers them tile-major -- tiles of 8x8 sparse cells, row-major inside a tile -- so that
   * 64 consecutive surfels (one wavefront) form a compact patch ("surfel order"). */
  uint32_t next = s->surfels_size;
  const int TP = 8 * dp->cell;
  const int tiles_x = (W + TP - 1) / TP, tiles_y = (H + TP - 1) / TP;
  for (int ty = 0; ty < tiles_y; ++ty)
    for (int tx = 0; tx < tiles_x; ++tx)
      for (int ly = 0; ly < TP; ++ly)
        for (int lx = 0; lx < TP; ++lx) {
          const int x = tx * TP + lx, y = ty * TP + ly;
          if (x >= W || y >= H) continue;
          if (flag[(size_t)y * W + x]) create_new_surfel(x, y, next++, color_cam, depth_cam, dp, kf, s);
        }
  free(flag);
  s->surfels_size += count;
  s->surfel_count += count;
  return count;
}

/* B/kernel_delete_surfels.cc:40-120 (update_radii = true) */
void orc_delete_surfels_and_update_radii(int min_observation_count, const orc_camera* depth_cam,
                                         const orc_depth_params* dp, orc_keyframe* const* kfs,
                                         int num_kfs, orc_surfels* s) {
  if (s->surfels_size == 0) return;
  float* a0 = srow(s, ORC_SURFEL_ACCUM0 + 0); float* a1 = srow(s, ORC_SURFEL_ACCUM0 + 1);
  float* a2 = srow(s, ORC_SURFEL_ACCUM0 + 2);
  uint32_t deleted = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : deleted)
  for (uint32_t i = 0; i < s->surfels_size; ++i) {
    a0[i] = 0; a1[i] = 0; a2[i] = INFINITY;
    for (int k = 0; k < num_kfs; ++k) {
      const orc_keyframe* kf = kfs[k];
      if (!kf) continue;
      proj_params p = make_proj_params(depth_cam, dp, s, kf, kf->frame_T_global);
      proj_result r;
      int fsv = 0;
      if (orc_project_associate(&p, i, &r, &fsv)) {
        a0[i] += 1.f;
        const float measured = orc_half_to_float(kf->radius[(size_t)r.py * kf->width + r.px]);
        a2[i] = fminf(a2[i], measured);
      } else if (fsv) {
        a1[i] += 1.f;
      }
    }
    const float observation_count = a0[i];
    if (observation_count < min_observation_count || a1[i] > observation_count) {
      if (!is_deleted(s, i)) { mark_deleted(s, i); deleted += 1; }
    } else {
      srow(s, ORC_SURFEL_RADIUS_SQ)[i] = a2[i];
    }
  }
  s->surfel_count -= deleted;
}

/* B/kernel_compact_surfels.cu:159-279: the j-th free slot (ascending) receives the j-th valid
 * surfel counted from the end, if that moves it to a smaller index.  Only the 8 data rows (and
 * the active flag) move. */
void orc_compact_surfels(orc_surfels* s) {
  if (s->surfels_size == s->surfel_count) return;
  const uint32_t n = s->surfels_size;
  const uint32_t free_spot_count = n - s->surfel_count;
  uint32_t* free_spots = (uint32_t*)malloc(sizeof(uint32_t) * (free_spot_count ? free_spot_count : 1));
  uint8_t* invalid = (uint8_t*)malloc(n ? n : 1); /* snapshot (kSurfelAccum2 flags in the reference) */
  uint32_t nf = 0;
  for (uint32_t i = 0; i < n; ++i) {
    invalid[i] = (uint8_t)is_deleted(s, i);
    if (invalid[i] && nf < free_spot_count) free_spots[nf++] = i;
  }
  uint32_t reverse_index = 0;
  for (uint32_t ii = n; ii-- > 0;) {
    if (invalid[ii]) continue;
    if (reverse_index < free_spot_count) {
      const uint32_t dst = free_spots[reverse_index];
      if (dst < ii) {
        for (int row = 0; row < ORC_SURFEL_DATA_ATTRS; ++row) srow(s, row)[dst] = srow(s, row)[ii];
        if (s->active) s->active[dst] = s->active[ii];
      }
    }
    ++reverse_index;
  }
  free(free_spots);
  free(invalid);
  s->surfels_size = s->surfel_count;
}

/* ---- spatial order (no counterpart in the reference; mirrors kernels_lifecycle.hip: sort_surfels_spatially) ---- */
static inline uint64_t spread21(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
typedef struct { uint64_t key; uint32_t idx; } sort_item;
static int cmp_sort_item(const void* a, const void* b) {
  const sort_item* x = (const sort_item*)a; const sort_item* y = (const sort_item*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);   /* ties keep their order: a stable sort */
}
void orc_sort_surfels_spatially(orc_surfels* s, float grid_cell_size) {
  const uint32_t n = s->surfels_size;
  if (n < 2) return;
  const float inv_cell = 1.0f / grid_cell_size;
  sort_item* items = (sort_item*)malloc(sizeof(sort_item) * n);
  for (uint32_t i = 0; i < n; ++i) {
    const v3 p = surfel_position(s, i);
    uint64_t key = ~0ull;
    if (p.x == p.x) {
      const float c[3] = {p.x, p.y, p.z};
      uint64_t q[3];
      for (int a = 0; a < 3; ++a) {
        float g = floorf(c[a] * inv_cell) + 1048576.f;
        g = fminf(fmaxf(g, 0.f), 2097151.f);
        q[a] = (uint64_t)g;
      }
      key = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    }
    items[i].key = key; items[i].idx = i;
  }
  qsort(items, n, sizeof(sort_item), cmp_sort_item);
  float* tmp = (float*)malloc(sizeof(float) * n);
  for (int row = 0; row < ORC_SURFEL_DATA_ATTRS; ++row) {
    float* r = srow(s, row);
    for (uint32_t i = 0; i < n; ++i) tmp[i] = r[items[i].idx];
    memcpy(r, tmp, sizeof(float) * n);
  }
  if (s->active) {
    uint8_t* t8 = (uint8_t*)tmp;
    for (uint32_t i = 0; i < n; ++i) t8[i] = s->active[items[i].idx];
    memcpy(s->active, t8, n);
  }
  free(tmp);
  free(items);
}
