/* oracle_ba.c -- the alternating bundle-adjustment scheme and the full cost evaluation.
 * Test infrastructure only (see oracle.h).
 * Follows B/direct_ba_alternating.cc:285-738 and B/direct_ba.cc:549-653. */
#include "oracle_internal.h"

/* B/direct_ba.cc:549-564 */
static void determine_covisible_active(orc_ba_state* st) {
  for (int k = 0; k < st->num_kfs; ++k) {
    orc_keyframe* kf = st->kfs[k];
    if (!kf || kf->activation != ORC_KF_ACTIVE) continue;
    if (st->covis_lists) {
      for (int c = 0; c < st->covis_counts[k]; ++c) {
        orc_keyframe* other = st->kfs[st->covis_lists[k][c]];
        if (other && other->activation == ORC_KF_INACTIVE) other->activation = ORC_KF_COVIS_ACTIVE;
      }
    } else {
      for (int c = 0; c < st->num_kfs; ++c) {
        orc_keyframe* other = st->kfs[c];
        if (c != k && other && other->activation == ORC_KF_INACTIVE) other->activation = ORC_KF_COVIS_ACTIVE;
      }
    }
  }
}

/* ours (host/direct_ba.cc: SortAfterInLoopCompaction): the compaction inside the loop is followed by the Morton reorder once the
 * surfels out of order -- appended, or moved into holes from the end of the buffer -- amount to one per 64-surfel tile */
void orc_sort_after_in_loop_compaction(orc_ba_state* st) {
  orc_surfels* s = st->surfels;
  if (st->spatial_sort_cell > 0.f && s->surfels_size > 1 && (uint64_t)st->unsorted_surfels * 64 >= (uint64_t)s->surfels_size) {
    orc_sort_surfels_spatially(s, st->spatial_sort_cell);
    st->unsorted_surfels = 0;
  }
}

/* B/direct_ba.cc:566-653 */
static void perform_ba_scheme_end_tasks(orc_ba_state* st, const orc_ba_options* opt) {
  orc_surfels* s = st->surfels;
  if (opt->do_surfel_updates) {
    for (int k = 0; k < st->num_kfs; ++k) {
      orc_keyframe* kf = st->kfs[k];
      if (!kf) continue;
      if (kf->last_active_in_ba_iteration == st->ba_iteration_count)
        orc_determine_supporting_surfels(1, opt->surfel_merge_dist_factor, &st->depth_cam, &st->dp, kf, s, st->supporting);
    }
  }
  orc_delete_surfels_and_update_radii(opt->min_observation_count, &st->depth_cam, &st->dp, st->kfs, st->num_kfs, s);
  /* B/direct_ba.cc:619: compaction without the active-flag buffer */
  uint8_t* active = s->active;
  s->active = NULL;
  st->unsorted_surfels += s->surfels_size - s->surfel_count;   /* the holes compaction fills with surfels from the end */
  orc_compact_surfels(s);
  s->active = active;
  /* ours (host/direct_ba.cc: PerformBASchemeEndTasks): surfels were appended or moved since the buffer was last in Morton
   * order -> reorder, so that a caller of the reference's API gets the spatially coherent buffer the sweeps are fast on */
  if (st->spatial_sort_cell > 0.f && st->unsorted_surfels > 0 && s->surfels_size > 1) orc_sort_surfels_spatially(s, st->spatial_sort_cell);
  if (st->spatial_sort_cell > 0.f) st->unsorted_surfels = 0;
}

void orc_bundle_adjustment_alternating(orc_ba_state* st, const orc_ba_options* opt, orc_ba_stats* stats) {
  orc_surfels* s = st->surfels;
  memset(stats, 0, sizeof(*stats));
  const int use_depth = opt->use_depth_residuals, use_desc = opt->use_descriptor_residuals;
  int optimize_depth_intrinsics = opt->optimize_depth_intrinsics && use_depth;   /* B/direct_ba.cc:427-434 */
  int optimize_color_intrinsics = opt->optimize_color_intrinsics && use_desc;

  const int fixed_ba_iteration_count = st->ba_iteration_count;
  if (!opt->increase_ba_iteration_count && fixed_ba_iteration_count != st->last_ba_iteration_count) {
    st->last_ba_iteration_count = fixed_ba_iteration_count;
    perform_ba_scheme_end_tasks(st, opt);
  }

  const int fixed_active_keyframe_set = opt->window_start > 0 || opt->window_end > 0;
  const int full_window = (opt->window_start == 0 && opt->window_end == st->num_kfs - 1);
  memset(s->active, 0, s->surfels_size);

  int* kfs_with_new_surfels = (int*)malloc(sizeof(int) * (st->num_kfs ? st->num_kfs : 1));

  for (int iteration = 0; iteration < opt->max_iterations; ++iteration) {
    stats->iterations_done += 1;
    if (fixed_active_keyframe_set) {
      for (int k = 0; k < st->num_kfs; ++k) {
        if (!st->kfs[k]) continue;
        st->kfs[k]->activation = (k >= opt->window_start && k <= opt->window_end) ? ORC_KF_ACTIVE : ORC_KF_INACTIVE;
      }
      determine_covisible_active(st);
    }

    /* --- surfel creation --- */
    int n_new_kfs = 0;
    const uint32_t old_surfels_size = s->surfels_size;
    if (opt->optimize_geometry && opt->do_surfel_updates) {
      for (int k = 0; k < st->num_kfs; ++k) {
        orc_keyframe* kf = st->kfs[k];
        if (!kf) continue;
        if (kf->activation == ORC_KF_ACTIVE && kf->last_active_in_ba_iteration != fixed_ba_iteration_count) {
          kf->last_active_in_ba_iteration = fixed_ba_iteration_count;
          kfs_with_new_surfels[n_new_kfs++] = k;
        } else if (kf->activation == ORC_KF_COVIS_ACTIVE && kf->last_covis_in_ba_iteration != fixed_ba_iteration_count) {
          kf->last_covis_in_ba_iteration = fixed_ba_iteration_count;
        }
      }
      for (int j = 0; j < n_new_kfs; ++j) {
        const int k = kfs_with_new_surfels[j];
        int* all = NULL; const int* covis; int n_covis;
        if (st->covis_lists) { covis = st->covis_lists[k]; n_covis = st->covis_counts[k]; }
        else {
          all = (int*)malloc(sizeof(int) * st->num_kfs); n_covis = 0;
          for (int c = 0; c < st->num_kfs; ++c) if (c != k && st->kfs[c]) all[n_covis++] = c;
          covis = all;
        }
        const uint32_t size_before = s->surfels_size;
        orc_create_surfels_for_keyframe(1, opt->min_observation_count, &st->color_cam, &st->depth_cam, &st->dp,
                                        st->kfs[k], st->kfs, covis, n_covis, s, st->supporting);
        st->unsorted_surfels += s->surfels_size - size_before;
        free(all);
      }
    }

    /* --- surfel activation --- */
    if (opt->optimize_geometry && s->surfels_size > old_surfels_size)
      memset(s->active + old_surfels_size, ORC_SURFEL_ACTIVE_FLAG, s->surfels_size - old_surfels_size);
    if (!full_window) memset(s->active, ORC_SURFEL_ACTIVE_FLAG, old_surfels_size);
    else orc_update_surfel_activation(&st->depth_cam, &st->dp, st->kfs, st->num_kfs, old_surfels_size, s);

    /* --- geometry --- */
    if (opt->optimize_geometry)
      orc_optimize_geometry_iteration(use_depth, use_desc, &st->color_cam, &st->depth_cam, &st->dp, st->kfs, st->num_kfs, s);

    /* --- surfel merge + compaction --- */
    if (opt->do_surfel_updates) {
      for (int j = 0; j < n_new_kfs; ++j) {
        orc_keyframe* kf = st->kfs[kfs_with_new_surfels[j]];
        if (!kf) continue;
        orc_determine_supporting_surfels(1, opt->surfel_merge_dist_factor, &st->depth_cam, &st->dp, kf, s, st->supporting);
      }
      if (n_new_kfs > 0) { st->unsorted_surfels += s->surfels_size - s->surfel_count; orc_compact_surfels(s); orc_sort_after_in_loop_compaction(st); }
    }

    /* --- poses --- */
    int num_converged = 0;
    if (opt->optimize_poses) {
      /* Every keyframe is estimated against the same (frozen) surfels, so the estimates are independent: they are
       * computed in parallel (one thread per keyframe, each summing in surfel order as before) and applied in
       * keyframe order. */
      int max_steps = 0;
      orc_se3* ests = (orc_se3*)malloc(sizeof(orc_se3) * (st->num_kfs ? st->num_kfs : 1));
      int* steps_of = (int*)calloc(st->num_kfs ? st->num_kfs : 1, sizeof(int));
#pragma omp parallel for schedule(dynamic, 1)
      for (int k = 0; k < st->num_kfs; ++k) {
        orc_keyframe* kf = st->kfs[k];
        if (!kf || kf->activation == ORC_KF_INACTIVE) continue;
        steps_of[k] = orc_estimate_frame_pose(use_depth, use_desc, &st->color_cam, &st->depth_cam, &st->dp, kf,
                                              &kf->global_T_frame, s, &ests[k], NULL);
      }
      for (int k = 0; k < st->num_kfs; ++k) {
        orc_keyframe* kf = st->kfs[k];
        if (!kf || kf->activation == ORC_KF_INACTIVE) { ++num_converged; continue; }
        const orc_se3 est = ests[k];
        const int steps = steps_of[k];
        stats->pose_gn_steps_total += steps;
        if (steps > max_steps) max_steps = steps;
        /* pose_difference = frame_T_global(old) * global_T_frame(new) */
        orc_se3 old_inv, diff;
        orc_se3_inverse(&kf->global_T_frame, &old_inv);
        orc_se3_mul(&old_inv, &est, &diff);
        float lg[6];
        orc_se3_log(&diff, lg);
        const int frame_moved = !orc_is_scale1_pose_converged(lg);
        orc_keyframe_set_global_T_frame(kf, &est);
        if (frame_moved) kf->activation = ORC_KF_ACTIVE;
        else { kf->activation = ORC_KF_INACTIVE; ++num_converged; }
      }
      free(ests); free(steps_of);
      stats->pose_gn_rounds_max_sum += max_steps;
    }

    /* --- intrinsics --- */
    if (optimize_depth_intrinsics || optimize_color_intrinsics) {
      orc_camera out_color, out_depth; float out_a;
      orc_optimize_intrinsics(optimize_depth_intrinsics, optimize_color_intrinsics, st->kfs, st->num_kfs,
                              &st->color_cam, &st->depth_cam, &st->dp, s, &out_color, &out_depth, &out_a);
      if (s->surfels_size > 0) {
        if (optimize_color_intrinsics) st->color_cam = out_color;
        if (optimize_depth_intrinsics) { st->depth_cam = out_depth; st->dp.a = out_a; }
      }
    }

    /* --- convergence --- */
    if (iteration >= opt->min_iterations - 1 && (num_converged == st->num_kfs || !opt->optimize_poses)) {
      stats->converged = 1;
      break;
    }
    determine_covisible_active(st);
  }
  free(kfs_with_new_surfels);

  if (opt->increase_ba_iteration_count) {
    perform_ba_scheme_end_tasks(st, opt);
    st->ba_iteration_count += 1;
  }
}

/* Full robust cost over all (keyframe, surfel) pairs; OpenMP over surfels.  This is the
 * "CPU cost-evaluation path" used as cpu_baseline in bench.py. */
double orc_evaluate_cost(int use_depth, int use_desc, const orc_camera* color_cam,
                         const orc_camera* depth_cam, const orc_depth_params* dp,
                         orc_keyframe* const* kfs, int num_kfs, const orc_surfels* s,
                         uint64_t* num_residuals) {
  const depth_to_color d2c = make_depth_to_color(depth_cam, color_cam);
  double total = 0;
  uint64_t count = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : total, count)
  for (uint32_t i = 0; i < s->surfels_size; ++i) {
    for (int k = 0; k < num_kfs; ++k) {
      const orc_keyframe* kf = kfs[k];
      if (!kf) continue;
      const float* F = kf->frame_T_global;
      proj_params p = make_proj_params(depth_cam, dp, s, kf, F);
      proj_result r;
      if (!orc_project_associate(&p, i, &r, NULL)) continue;
      if (use_depth) {
        const v3 nl = m34_rotate(F, r.normal);
        const float inv_std = depth_inv_stddev(unp_nx(&p.unp, (float)r.px), unp_ny(&p.unp, (float)r.py),
                                               r.calibrated_depth, nl, dp->baseline_fx);
        const v3 u = unp_point(&p.unp, r.px, r.py, r.calibrated_depth);
        const float raw = inv_std * v3_dot(nl, v3_sub(u, r.local_position));
        total += weighted_depth_residual(raw);
        count += 1;
      }
      if (use_desc) {
        float c[2];
        if (!transform_depth_to_color(r.pxx, r.pxy, &d2c, &c[0], &c[1])) continue;
        float t1[2], t2[2], raw1, raw2;
        orc_tangent_projections(r.global_position, r.normal, srow(s, ORC_SURFEL_RADIUS_SQ)[i], F, color_cam, t1, t2);
        orc_raw_descriptor_residual(kf, c, t1, t2, srow(s, ORC_SURFEL_DESC1)[i], srow(s, ORC_SURFEL_DESC2)[i], &raw1, &raw2);
        total += weighted_descriptor_residual(raw1) + weighted_descriptor_residual(raw2);
        count += 2;
      }
    }
  }
  if (num_residuals) *num_residuals = count;
  return total;
}
