/* oracle_preprocess.c -- keyframe construction from raw depth + RGB (B/keyframe.cc:81-158).
 * Test infrastructure only (see oracle.h). */
#include "oracle_internal.h"

/* B/cuda_image_processing.cu:165-175 */
void orc_compute_brightness(const uint8_t* rgb, int width, int height, uint8_t* rgba) {
  for (size_t i = 0; i < (size_t)width * height; ++i) {
    const uint8_t r = rgb[3 * i + 0], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
    /* evaluated as the fused chain a CUDA/HIP compiler emits for this expression (fmad contraction) */
    const uint8_t intensity = (uint8_t)(fmaf(0.114f, (float)b, fmaf(0.587f, (float)g, 0.299f * (float)r)) + 0.5f);
    rgba[4 * i + 0] = r; rgba[4 * i + 1] = g; rgba[4 * i + 2] = b; rgba[4 * i + 3] = intensity;
  }
}

/* B/cuda_depth_processing.cu:42-128 */
void orc_bilateral_filter_and_depth_cutoff(float sigma_xy, float sigma_value, float radius_factor, uint16_t max_depth,
                                           float raw_to_float_depth, const uint16_t* in_depth, int width, int height,
                                           uint16_t* out_depth) {
  const float denom_xy = 2.0f * sigma_xy * sigma_xy, denom_value = 2.0f * sigma_value * sigma_value;
  const int radius = (int)(radius_factor * sigma_xy + 0.5f);
  const int radius_squared = radius * radius;
#pragma omp parallel for schedule(static)
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const uint16_t center_value = in_depth[(size_t)y * width + x];
      if (center_value == 0 || center_value > max_depth) { out_depth[(size_t)y * width + x] = ORC_UNKNOWN_DEPTH; continue; }
      const float inv_center_value = 1.0f / (raw_to_float_depth * center_value);
      float sum = 0, weight = 0;
      const int min_y = y - radius > 0 ? y - radius : 0, max_y = y + radius < height - 1 ? y + radius : height - 1;
      const int min_x = x - radius > 0 ? x - radius : 0, max_x = x + radius < width - 1 ? x + radius : width - 1;
      for (int sy = min_y; sy <= max_y; ++sy) {
        const int dy = sy - y;
        for (int sx = min_x; sx <= max_x; ++sx) {
          const int dx = sx - x;
          const int grid_distance_squared = dx * dx + dy * dy;
          if (grid_distance_squared > radius_squared) continue;
          const uint16_t sample = in_depth[(size_t)sy * width + sx];
          if (sample == 0) continue;
          const float inv_sample = 1.0f / (raw_to_float_depth * sample);
          float value_distance_squared = inv_center_value - inv_sample;
          value_distance_squared *= value_distance_squared;
          const float w = orc_exp(-grid_distance_squared / denom_xy + -value_distance_squared / denom_value);
          sum += w * inv_sample;
          weight += w;
        }
      }
      out_depth[(size_t)y * width + x] = (weight == 0) ? ORC_UNKNOWN_DEPTH : (uint16_t)(1.0f / (raw_to_float_depth * sum / weight));
    }
}

static inline float calib_at(const orc_depth_params* dp, int x, int y, uint16_t raw) {
  return orc_raw_to_calibrated_depth(dp->a, cfactor_at(dp, x, y), dp->raw_to_float_depth, raw);
}

/* B/cuda_depth_processing.cu:134-264 */
void orc_compute_normals(const orc_camera* cam, const orc_depth_params* dp, const uint16_t* in_depth,
                         uint16_t* out_depth, uint16_t* out_normals) {
  const int W = cam->width, H = cam->height;
  const unprojector unp = make_unprojector(cam);
  const uint16_t zero_normal = orc_pack_normal8(0, 0);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) {
      const size_t idx = (size_t)y * W + x;
      if (x < 1 || y < 1 || x >= W - 1 || y >= H - 1) {
        out_depth[idx] = ORC_UNKNOWN_DEPTH; out_normals[idx] = zero_normal; continue;
      }
      const uint16_t center_raw = in_depth[idx];
      if (center_raw & ORC_INVALID_DEPTH_BIT) {
        out_depth[idx] = ORC_UNKNOWN_DEPTH; out_normals[idx] = zero_normal; continue;
      }
      const uint16_t right_raw = in_depth[idx + 1], left_raw = in_depth[idx - 1];
      const uint16_t bottom_raw = in_depth[idx + W], top_raw = in_depth[idx - W];
      if ((right_raw | left_raw | bottom_raw | top_raw) & ORC_INVALID_DEPTH_BIT) {
        out_depth[idx] = ORC_UNKNOWN_DEPTH; out_normals[idx] = zero_normal; continue;
      }
      const float center_depth = calib_at(dp, x, y, center_raw);
      const float left_depth = calib_at(dp, x - 1, y, left_raw);
      const float top_depth = calib_at(dp, x, y - 1, top_raw);
      const float right_depth = calib_at(dp, x + 1, y, right_raw);
      const float bottom_depth = calib_at(dp, x, y + 1, bottom_raw);
      const v3 left_point = unp_point(&unp, x - 1, y, left_depth);
      const v3 top_point = unp_point(&unp, x, y - 1, top_depth);
      const v3 right_point = unp_point(&unp, x + 1, y, right_depth);
      const v3 bottom_point = unp_point(&unp, x, y + 1, bottom_depth);
      const v3 center_point = unp_point(&unp, x, y, center_depth);

      const float kRatioThresholdSquared = 2.f * 2.f;
      const float left_dist_sq = v3_sqlen(v3_sub(left_point, center_point));
      const float right_dist_sq = v3_sqlen(v3_sub(right_point, center_point));
      const float left_right_ratio = left_dist_sq / right_dist_sq;
      v3 left_to_right;
      if (left_right_ratio < kRatioThresholdSquared && left_right_ratio > 1.f / kRatioThresholdSquared) {
        left_to_right = v3_sub(right_point, left_point);
      } else if (left_dist_sq < right_dist_sq) {
        left_to_right = v3_sub(center_point, left_point);
      } else {
        left_to_right = v3_sub(right_point, center_point);
      }
      const float bottom_dist_sq = v3_sqlen(v3_sub(bottom_point, center_point));
      const float top_dist_sq = v3_sqlen(v3_sub(top_point, center_point));
      const float bottom_top_ratio = bottom_dist_sq / top_dist_sq;
      v3 bottom_to_top;
      if (bottom_top_ratio < kRatioThresholdSquared && bottom_top_ratio > 1.f / kRatioThresholdSquared) {
        bottom_to_top = v3_sub(top_point, bottom_point);
      } else if (bottom_dist_sq < top_dist_sq) {
        bottom_to_top = v3_sub(center_point, bottom_point);
      } else {
        bottom_to_top = v3_sub(top_point, center_point);
      }
      v3 normal = v3_cross(left_to_right, bottom_to_top);
      const float length = v3_norm(normal);
      if (!(length > 1e-6f)) {
        normal = v3_make(0, 0, -1);
      } else {
        const float inv_length = ((unp.fy_inv < 0) ? -1.0f : 1.0f) / length;
        normal.x *= inv_length;
        normal.y *= inv_length;
      }
      out_normals[idx] = orc_pack_normal8(normal.x, normal.y);
      out_depth[idx] = in_depth[idx];
    }
  }
}

/* B/cuda_depth_processing.cu:289-360 (min_neighbors_for_radius_computation = 4) */
void orc_compute_point_radii(const orc_camera* cam, float raw_to_float_depth, const uint16_t* depth,
                             uint16_t* radius, uint16_t* out_depth) {
  const int W = cam->width, H = cam->height;
  const unprojector u = make_unprojector(cam);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y) {
    for (int x = 0; x < W; ++x) {
      const size_t idx = (size_t)y * W + x;
      const uint16_t depth_u16 = depth[idx];
      if (depth_u16 & ORC_INVALID_DEPTH_BIT) { out_depth[idx] = ORC_UNKNOWN_DEPTH; continue; }
      const float d = raw_to_float_depth * depth_u16;
      const v3 local_position = v3_make(d * (u.fx_inv * x + u.cx_inv), d * (u.fy_inv * y + u.cy_inv), d);
      int neighbor_count = 0;
      float min_dist_sq = INFINITY;
      for (int dy = y - 1; dy < y + 2; ++dy) {
        for (int dx = x - 1; dx < x + 2; ++dx) {
          if ((dx != x && dy != y) || (dx == x && dy == y)) continue;
          /* valid pixels are never on the border (ComputeNormals invalidates it), so no bounds issue;
           * guard anyway for robustness against hand-made inputs */
          if (dx < 0 || dy < 0 || dx >= W || dy >= H) continue;
          const uint16_t d_depth = depth[(size_t)dy * W + dx];
          if (d_depth & ORC_INVALID_DEPTH_BIT) continue;
          ++neighbor_count;
          const float dd = raw_to_float_depth * d_depth;
          const v3 other = v3_make(dd * (u.fx_inv * dx + u.cx_inv), dd * (u.fy_inv * dy + u.cy_inv), dd);
          const float dist_sq = v3_sqlen(v3_sub(other, local_position));
          if (dist_sq < min_dist_sq) min_dist_sq = dist_sq;
        }
      }
      const int valid = neighbor_count >= 4;
      radius[idx] = orc_float_to_half(valid ? min_dist_sq : 0);
      out_depth[idx] = valid ? depth_u16 : ORC_UNKNOWN_DEPTH;
    }
  }
}

/* B/cuda_depth_processing.cu:391-465: min initialised to +inf, max to 0 (init buffer,
 * B/cuda_depth_processing.cu ComputeMinMaxDepthCUDA_InitializeBuffers). */
void orc_compute_min_max_depth(const uint16_t* depth, int width, int height, float raw_to_float_depth,
                               float* min_depth, float* max_depth) {
  float mn = INFINITY, mx = 0.f;
  for (size_t i = 0; i < (size_t)width * height; ++i) {
    if (depth[i] & ORC_INVALID_DEPTH_BIT) continue;
    const float d = raw_to_float_depth * depth[i];
    if (d < mn) mn = d;
    if (d > mx) mx = d;
  }
  *min_depth = mn; *max_depth = mx;
}

/* B/keyframe.cc:81-158.  Order matters: normals are computed from the raw upload into a
 * temporary depth image (which receives the border/neighbour invalidation); radii + isolated
 * pixel removal read that temporary and write the final depth; min/max is taken over the
 * temporary. */
void orc_keyframe_from_images(orc_keyframe* kf, const orc_camera* depth_cam, const orc_depth_params* dp,
                              const uint16_t* depth_image, const uint8_t* rgb_image,
                              const orc_se3* global_T_frame) {
  const int W = depth_cam->width, H = depth_cam->height;
  kf->width = W; kf->height = H;
  orc_compute_brightness(rgb_image, kf->color_width, kf->color_height, kf->color);
  uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)W * H);
  orc_compute_normals(depth_cam, dp, depth_image, tmp, kf->normals);
  memset(kf->radius, 0, sizeof(uint16_t) * (size_t)W * H);
  orc_compute_point_radii(depth_cam, dp->raw_to_float_depth, tmp, kf->radius, kf->depth);
  orc_compute_min_max_depth(tmp, W, H, dp->raw_to_float_depth, &kf->min_depth, &kf->max_depth);
  free(tmp);
  orc_keyframe_set_global_T_frame(kf, global_T_frame);
  kf->activation = ORC_KF_ACTIVE;
  kf->last_active_in_ba_iteration = -1;
  kf->last_covis_in_ba_iteration = -1;
}
