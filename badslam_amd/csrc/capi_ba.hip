// capi_ba.hip -- the stages of the alternating scheme behind the C boundary: activation, geometry step, the batched Gauss-Newton rounds of
// the pose phase (queued ahead of the host), and the device-driven loop (bahip_alternating_iterations).
#include "capi_internal.h"

using namespace bahip;
using namespace bahip_capi;

namespace bahip_capi {
// Waits until pose_solve_kernel has published `sequence` in the host copy of the counter records.  Polling a word of mapped
// host memory costs a microsecond where hipStreamSynchronize + a 256-byte copy cost 25.  If the word does not show up within
// two seconds (a runtime that does not map the allocation coherently), fall back to synchronising and copying.
int wait_for_pose_sequence(bahip_context* ctx, PoseWork* host_work, const PoseWork* dev_work, int num_work, int sequence) {
  PoseWork* host_tail = host_work + num_work;
  volatile int* published = reinterpret_cast<volatile int*>(host_tail) + kPoseCounterSequence;
  if (!ctx->poll_disabled) {
    const auto start = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
      if (*published == sequence) { std::atomic_thread_fence(std::memory_order_acquire); return 0; }
      if ((spin & 0xfff) == 0xfff && std::chrono::steady_clock::now() - start > std::chrono::seconds(2)) break;
    }
  }
  // not seen within two seconds (or polling is off): wait for the stream.  If the word is there afterwards the launch was
  // merely slow and polling stays on; if it is not, this system does not show the kernel's stores to the host: copy, and
  // stop polling for this context.
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (*published == sequence) { std::atomic_thread_fence(std::memory_order_acquire); return 0; }
  if (!ctx->poll_disabled) {
    ctx->poll_disabled = true;
    fprintf(stderr, "badslam_hip: the pose counters were not published to host memory; falling back to stream synchronisation\n");
  }
  // the finished work items were written to the same mapped memory by the solve kernel: bring the whole record range over,
  // not only the counters, or the poses read after the phase would be stale
  HIP_TRY(hipMemcpy(host_work, dev_work, sizeof(PoseWork) * ((size_t)num_work + kPoseTailRecords), hipMemcpyDeviceToHost));
  return 0;
}

// Batched Gauss-Newton rounds over `num_work` work items already initialised on the device.
//
// Rounds are queued AHEAD of the host (round 4): a later round's accumulate launch reads the number of work items still
// iterating from the counter the previous round's solve kernel left on the device (and does nothing when it is zero), so a
// batch of rounds -- accumulate, exchange, solve each -- goes out without the host in between, and the host waits once per
// batch, for the last solve's sequence number.  The batch size follows the previous phase on the same table (*rounds_hint):
// in the steady state of a BA loop a phase needs one or two rounds and costs one host reaction instead of one per round.
// A round queued in vain costs two near-empty launches (and, sharded, an exchange of zeros); results do not depend on the batch
// size (tests run 1, the default and 4).
// the launch that ends a pose phase of the device-driven loop also sets up the next iteration (kernels_pose.hip: pose_solve_begin_kernel)
// (off by default: measured SLOWER in round 4 -- 583 against 598 BA iterations/s, 0.420 against 0.411 ms on an eighth of the cloud:
// sixteen wavefronts on one compute unit take longer over the set-up, and over a real solve, than the launch they save)
int g_fused_iteration_begin = bahip_env_int("BAHIP_FUSED_ITERATION_BEGIN", 0);
int g_pose_rounds_ahead = bahip_env_int("BAHIP_POSE_ROUNDS_AHEAD", 0);
int run_pose_rounds(bahip_context* ctx, bool use_depth, bool use_desc, const KfEntry* dev_frames, KfEntry* dev_frames_rw,
                    PoseWork* dev_work, HbFixed* dev_Hb, int num_work, const SurfelsView& s, int write_back, int update_activation,
                    PoseWork* host_work /* page-locked, num_work + kPoseTailRecords records */, int* rounds_out,
                    bool schedule /* a phase over the keyframe table: its first round counts the candidates per tile and the
                    run order of the following sweeps is rebuilt from them */, int* rounds_hint,
                    int first_round, int first_iterating /* continue a phase whose rounds [0, first_round) have run (the
                    device-driven loop hands over a phase that needs more rounds than it had queued) */,
                    const PoseLoopControl* loop_stats /* keeps the loop's totals going (never ends a phase) */) {
  int rounds = 0;
  int iterating = first_round > 0 ? first_iterating : num_work;
  const int* counters = reinterpret_cast<const int*>(host_work + num_work);
  const int* dev_counters = reinterpret_cast<const int*>(dev_work + num_work);
  if (ensure_tile_bounds(ctx, s.size)) return 1;
  const uint32_t padded_tiles = pose_padded_tiles(s.size);
  // (costs drift slowly -- poses move by millimetres, keyframes come one at a time -- so the census and the order kernel (one
  // workgroup: 0.16 ms at 47 k tiles) are spent on every 32nd phase only, and whenever the grid has changed)
  constexpr int kSchedulePhases = 32;
  schedule = schedule && g_tile_order_enabled && s.size > 0 && !ctx->tile_order_unavailable_for(padded_tiles) &&
             (ctx->tile_order_tiles != padded_tiles || ++ctx->phases_since_schedule >= kSchedulePhases);
  if (schedule && ensure_tile_schedule(ctx, padded_tiles)) return 1;
  static const bool host_timing = getenv("BADSLAM_HOST_TIMING") != nullptr;   // diagnostics: where a pose round's wall time goes
  static double t_launch = 0, t_wait = 0; static long n_rounds = 0;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // process-wide and increasing: page-locked memory is recycled between contexts, and a word left behind by an earlier
  // context must never equal a sequence number somebody is going to wait for
  static std::atomic<int> g_pose_sequence{0};
  const int wanted_ahead = g_pose_rounds_ahead > 0 ? g_pose_rounds_ahead : std::max(1, std::min(rounds_hint ? *rounds_hint : 1, 4));
  int round = first_round;
  while (round < BAHIP_MAX_POSE_ITERATIONS && iterating > 0) {
    const double t0 = host_timing ? now() : 0;
    int batch = std::min(wanted_ahead, BAHIP_MAX_POSE_ITERATIONS - round);
    if (batch > 1 && !pose_round_can_be_queued_ahead(s.size, round == 0 ? num_work : iterating, ctx->dev_tile_counters != nullptr)) batch = 1;
    int sequence = 0;
    StageTimer& acc_timer = ctx->timers[2];
    for (int ahead = 0; ahead < batch; ++ahead) {
      const int r = round + ahead;
      // `iterating`: what the host knows -- exact for the first round of the batch, an upper bound for the rounds queued ahead
      // (the list only shrinks), which read the exact count from the device
      timer_begin(ctx, 2, r == 0, ahead == 0 ? iterating : 0);
      launch_pose_accumulate(ctx->stream, use_depth, use_desc, ctx->in, dev_frames, dev_work, num_work, s, dev_Hb, ctx->dev_tile_bounds,
                             /*stored_bounds*/ r > 0, /*num_listed*/ iterating, ctx->dev_tile_counters, &ctx->pose_parity,
                             (schedule && r == 0) ? ctx->dev_tile_cost : nullptr, tile_order_for(ctx, s.size),
                             ahead > 0 ? dev_counters + (r - 1) : nullptr);
      timer_end(ctx, 2);
      CHECK_LAUNCH();
      if (schedule && r == 0) {
        if (launch_tile_order(ctx->stream, ctx->dev_tile_cost, padded_tiles, ctx->dev_tile_order)) {
          ctx->tile_order_tiles = padded_tiles;
          ctx->phases_since_schedule = 0;
          CHECK_LAUNCH();
        } else {
          // more runs than the order kernel handles: remember it, so that the census is not taken again for this grid
          ctx->tile_order_unavailable_tiles = padded_tiles;
          HIP_TRY(hipMemsetAsync(ctx->dev_tile_cost, 0, sizeof(uint32_t) * padded_tiles, ctx->stream));
        }
      }
      // integer sum over the ranks: exact, so a sharded run produces the H, b of the unsharded one bit for bit
      // (keyframe sharding: the ranks hold disjoint keyframes and all surfels, so the sum completes each rank's table -- the
      // "all-reduce of pose Hessians" of BASELINE configs[3]; a single frame outside the table is complete on every rank)
      if (!(kf_sharded(ctx) && dev_frames == ctx->dev_frame1) && reduce_over_ranks(ctx, dev_Hb, (size_t)num_work * kHbStride, BAHIP_SUM_I64)) return 1;
      timer_begin(ctx, 3, r == 0);
      sequence = ++g_pose_sequence;
      launch_pose_solve(ctx->stream, dev_work, num_work, dev_Hb, dev_frames_rw, write_back, update_activation, r, host_work, sequence, loop_stats);
      timer_end(ctx, 3);
      CHECK_LAUNCH();
    }
    // No stream synchronisation and no copy: the solve kernel writes finished work items and, last, the counters and its
    // launch's sequence number into host_work (mapped, coherent host memory); the host polls the sequence number of the
    // batch's last solve.
    const double t1 = host_timing ? now() : 0;
    if (wait_for_pose_sequence(ctx, host_work, dev_work, num_work, sequence)) return 1;
    if (counters[kPoseCounterInvalid])
      return fail("pose normal equations: a tile total was not finite or reached 2^52 (hb_split), or a sum left the fixed-point range; the "
                  "surfels or images hold non-finite values", __FILE__, __LINE__);
    if (host_timing) {
      t_launch += t1 - t0; t_wait += now() - t1;
      n_rounds += batch;
      if (n_rounds % 30 < batch) fprintf(stderr, "[pose rounds, us per round] enqueue %.1f | wait %.1f\n", t_launch / n_rounds, t_wait / n_rounds);
    }
    // which of the batch's rounds had work: round r did iff something was still iterating after round r - 1
    int executed = 0;
    for (int ahead = 0; ahead < batch && iterating > 0; ++ahead) {
      if (ahead > 0 && timer_on(ctx, 2)) acc_timer.units += iterating;   // the keyframes that launch swept (known only now)
      ++executed;
      iterating = counters[round + ahead];
    }
    // the launches queued in vain are not launches of the sweep: their event pairs (the last ones recorded) are dropped, so
    // that launch counts and average durations keep describing launches that did work
    if (timer_on(ctx, 2) && executed < batch) acc_timer.used = std::max(0, acc_timer.used - (batch - executed));
    if (timer_on(ctx, 3) && executed < batch) ctx->timers[3].used = std::max(0, ctx->timers[3].used - (batch - executed));
    rounds += executed;
    round += batch;
  }
  if (rounds_hint) *rounds_hint = rounds;
  if (rounds_out) *rounds_out = rounds;
  return 0;
}

// Keyframe-sharded geometry step: three launches, the class partials of the normals pass and of the position pass summed over
// the ranks in between (as 64-bit integers: a rank's partials are zero where another rank's are not, so bit patterns survive).
int geometry_keyframe_sharded(bahip_context* ctx, bool use_depth, bool use_desc, const SurfelsView& v, long long activate_count) {
  REQUIRE(is_sharded(ctx), "keyframe sharding needs an all-reduce hook or an RCCL communicator");
  if (v.size == 0) return 0;
  const int nn = geometry_normals_sums(activate_count >= 0), np = geometry_position_sums(use_desc);
  const size_t stride = ((size_t)v.size + 63) & ~(size_t)63;
  // class c (the keyframes k with k % classes == c) lives on rank c % world: world divides classes, both powers of two
  const int classes = ctx->sum_classes;
  const size_t normals_floats = (size_t)classes * nn * stride, position_floats = (size_t)classes * np * stride;
  if (grow_device(&ctx->kf_partials, &ctx->kf_partials_capacity, normals_floats + position_floats, 0, "the class partials of the geometry step")) return 1;
  uint32_t owned = 0;
  for (int c = 0; c < classes; ++c) if ((c & (ctx->kf_world - 1)) == ctx->kf_rank) owned |= 1u << c;
  const ClassPartials cpn{ctx->kf_partials, (uint32_t)stride, owned}, cpp{ctx->kf_partials + normals_floats, (uint32_t)stride, owned};
  HIP_TRY(hipMemsetAsync(ctx->kf_partials, 0, sizeof(float) * (normals_floats + position_floats), ctx->stream));
  launch_geometry_phase(ctx->stream, 1, use_depth, use_desc, ctx->in, ctx->dev_kfs, ctx->num_kfs, v, activate_count, cpn, cpp);
  CHECK_LAUNCH();
  if (reduce_over_ranks(ctx, cpn.data, normals_floats / 2, BAHIP_SUM_I64)) return 1;
  launch_geometry_phase(ctx->stream, 2, use_depth, use_desc, ctx->in, ctx->dev_kfs, ctx->num_kfs, v, activate_count, cpn, cpp);
  CHECK_LAUNCH();
  if (reduce_over_ranks(ctx, cpp.data, position_floats / 2, BAHIP_SUM_I64)) return 1;
  launch_geometry_phase(ctx->stream, 3, use_depth, use_desc, ctx->in, ctx->dev_kfs, ctx->num_kfs, v, activate_count, cpn, cpp);
  CHECK_LAUNCH();
  return 0;
}

}  // namespace bahip_capi

extern "C" {
// ---- stages ---------------------------------------------------------------------------------------------
int bahip_update_surfel_activation(bahip_context* ctx, const bahip_surfels* surfels, uint32_t surfels_size) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(surfels->active != nullptr, "activation needs the active-surfel buffer");
  if (kf_sharded(ctx)) {
    // a surfel is active iff a kActive keyframe of ANY rank sees it: one hit word per surfel, summed over the ranks
    REQUIRE(is_sharded(ctx), "keyframe sharding needs an all-reduce hook or an RCCL communicator");
    if (surfels_size == 0) return 0;
    const size_t words = ((size_t)surfels_size + 63) & ~(size_t)63;
    if (grow_device(&ctx->kf_partials, &ctx->kf_partials_capacity, words, 0, "the activation hit words")) return 1;
    uint32_t* hits = reinterpret_cast<uint32_t*>(ctx->kf_partials);
    HIP_TRY(hipMemsetAsync(hits, 0, sizeof(uint32_t) * words, ctx->stream));
    timer_begin(ctx, 0, true);
    launch_activation_hits(ctx->stream, ctx->in, ctx->dev_kfs, ctx->num_kfs, make_view(surfels), surfels_size, ctx->kf_rank, ctx->kf_world, hits);
    CHECK_LAUNCH();
    if (reduce_over_ranks(ctx, hits, words / 2, BAHIP_SUM_I64)) return 1;
    launch_activation_from_hits(ctx->stream, make_view(surfels), surfels_size, hits);
    timer_end(ctx, 0);
    CHECK_LAUNCH();
    return 0;
  }
  timer_begin(ctx, 0, true);
  launch_activation(ctx->stream, ctx->in, ctx->dev_kfs, ctx->num_kfs, make_view(surfels), surfels_size);
  timer_end(ctx, 0);
  CHECK_LAUNCH();
  return 0;
}

int bahip_assign_colors(bahip_context* ctx, const bahip_surfels* surfels) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE_NO_KF_SHARDING("bahip_assign_colors");
  launch_assign_colors(ctx->stream, ctx->in, ctx->dev_kfs, ctx->num_kfs, make_view(surfels));
  CHECK_LAUNCH();
  return 0;
}

int bahip_update_surfel_normals(bahip_context* ctx, const bahip_surfels* surfels) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(surfels->active != nullptr, "normals update needs the active-surfel buffer");
  REQUIRE_NO_KF_SHARDING("bahip_update_surfel_normals (a stage of the PCG scheme)");
  launch_normals(ctx->stream, ctx->in, ctx->dev_kfs, ctx->num_kfs, make_view(surfels));
  CHECK_LAUNCH();
  return 0;
}

int bahip_optimize_geometry_iteration(bahip_context* ctx, int use_depth, int use_desc, const bahip_surfels* surfels) {
  ctx->lifecycle_bounds_tiles = 0;   // positions change or surfels move: a batch's tile bounds end here
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(use_depth || use_desc, "at least one residual type must be enabled");   // B/kernel_opt_geometry.cc:91
  REQUIRE(surfels->active != nullptr, "geometry optimisation needs the active-surfel buffer");
  timer_begin(ctx, 1, true);
  if (kf_sharded(ctx)) {
    if (geometry_keyframe_sharded(ctx, use_depth != 0, use_desc != 0, make_view(surfels), -1)) return 1;
  } else {
    launch_geometry(ctx->stream, use_depth != 0, use_desc != 0, ctx->in, ctx->dev_kfs, ctx->num_kfs, make_view(surfels), -1,
                    tile_order_for(ctx, surfels->surfels_size));
  }
  timer_end(ctx, 1);
  CHECK_LAUNCH();
  return 0;
}

int bahip_update_activation_and_optimize_geometry(bahip_context* ctx, int use_depth, int use_desc, const bahip_surfels* surfels,
                                                  uint32_t activation_surfels_size) {
  ctx->lifecycle_bounds_tiles = 0;
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(use_depth || use_desc, "at least one residual type must be enabled");   // B/kernel_opt_geometry.cc:91
  REQUIRE(surfels->active != nullptr, "geometry optimisation needs the active-surfel buffer");
  REQUIRE(activation_surfels_size <= surfels->surfels_size, "activation range exceeds surfels_size");
  timer_begin(ctx, 1, true);
  if (kf_sharded(ctx)) {
    if (geometry_keyframe_sharded(ctx, use_depth != 0, use_desc != 0, make_view(surfels), (long long)activation_surfels_size)) return 1;
  } else {
    launch_geometry(ctx->stream, use_depth != 0, use_desc != 0, ctx->in, ctx->dev_kfs, ctx->num_kfs, make_view(surfels),
                    (long long)activation_surfels_size, tile_order_for(ctx, surfels->surfels_size));
  }
  timer_end(ctx, 1);
  CHECK_LAUNCH();
  return 0;
}

int bahip_accumulate_pose_estimation_coeffs(bahip_context* ctx, int use_depth, int use_desc, const bahip_frame* frame,
                                            const float frame_T_global[12], const bahip_surfels* surfels, float* H, float* b) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(use_depth || use_desc, "at least one residual type must be enabled");   // B/kernel_opt_pose.cc:58
  REQUIRE(surfels->surfels_size > 0, "AccumulatePoseEstimationCoeffs is only intended for surfels_size > 0");  // :61
  KfEntry e;
  if (make_entry(ctx, *frame, 0, &e)) return 1;
  PoseWork w[1 + kPoseTailRecords] = {};   // the work item and its (zeroed) counter records
  memcpy(w[0].F, frame_T_global, 12 * sizeof(float));
  w[0].kf_index = 0;
  HIP_TRY(hipMemcpyAsync(ctx->dev_frame1, &e, sizeof(e), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->dev_work1, w, sizeof(w), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->dev_Hb1, 0, sizeof(HbFixed) * kHbStride, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ensure_tile_bounds(ctx, surfels->surfels_size)) return 1;
  launch_pose_accumulate(ctx->stream, use_depth != 0, use_desc != 0, ctx->in, ctx->dev_frame1, ctx->dev_work1, 1,
                         make_view(surfels), ctx->dev_Hb1, ctx->dev_tile_bounds, /*stored_bounds*/ false, /*num_listed*/ 0,
                         ctx->dev_tile_counters, &ctx->pose_parity);
  CHECK_LAUNCH();
  // (keyframe sharding: every rank holds all surfels, a single frame's equations are complete on each)
  if (!kf_sharded(ctx) && reduce_over_ranks(ctx, ctx->dev_Hb1, kHbStride, BAHIP_SUM_I64)) return 1;
  HbFixed* fixed = reinterpret_cast<HbFixed*>(ctx->pinned_f);   // 56 x 8 bytes of the 128-float pinned buffer
  HIP_TRY(hipMemcpyAsync(fixed, ctx->dev_Hb1, sizeof(HbFixed) * kHbStride, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  // the sweep's "not representable" flag travels in the row's unused 28th coefficient (kernels_pose.hip: pose_invalid_word), summed
  // over the ranks like the rest of the row
  if (fixed[27 * kHbLimbs] != 0)
    return fail("pose normal equations: a tile total was not finite or reached 2^52 (hb_split)", __FILE__, __LINE__);
  for (int c = 0; c < 21; ++c) H[c] = (float)hb_value(fixed[c * kHbLimbs], fixed[c * kHbLimbs + 1]);
  for (int c = 0; c < 6; ++c) b[c] = (float)hb_value(fixed[(21 + c) * kHbLimbs], fixed[(21 + c) * kHbLimbs + 1]);
  return 0;
}

int bahip_estimate_frame_pose(bahip_context* ctx, int use_depth, int use_desc, const bahip_frame* frame,
                              const float init[7], const bahip_surfels* surfels, float out[7], int* iterations_done,
                              int* converged) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(use_depth || use_desc, "at least one residual type must be enabled");
  KfEntry e;
  if (make_entry(ctx, *frame, 0, &e)) return 1;
  PoseWork w[1 + kPoseTailRecords] = {};   // the work item and its (zeroed) counter records
  memcpy(w[0].T, init, 7 * sizeof(float));
  memcpy(w[0].T0, init, 7 * sizeof(float));
  float inv[7];
  se3_inverse(init, inv);
  se3_matrix3x4(inv, w[0].F);
  HIP_TRY(hipMemcpyAsync(ctx->dev_frame1, &e, sizeof(e), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->dev_work1, w, sizeof(w), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->dev_Hb1, 0, sizeof(HbFixed) * kHbStride, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  // surfels_size == 0: H = b = 0 -> x = 0 -> converged after one step (B/direct_ba_alternating.cc:148-151)
  if (run_pose_rounds(ctx, use_depth != 0, use_desc != 0, ctx->dev_frame1, ctx->dev_frame1, ctx->dev_work1, ctx->dev_Hb1, 1,
                      make_view(surfels), /*write_back*/ 0, /*update_activation*/ 0, ctx->pinned_work1, nullptr, false, &ctx->rounds_hint_frame)) return 1;
  const PoseWork& result = ctx->pinned_work1[0];
  memcpy(out, result.T, 7 * sizeof(float));
  if (iterations_done) *iterations_done = result.iterations;
  if (converged) *converged = result.converged;
  return 0;
}

static int estimate_keyframe_poses_impl(bahip_context* ctx, int use_depth, int use_desc, const bahip_surfels* surfels,
                                       float* global_T_frame_out, int* iterations_done, int* converged, int* rounds_out,
                                       bool update_activation, int* moved_out, int* num_converged_out) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(use_depth || use_desc, "at least one residual type must be enabled");
  const int K = ctx->num_kfs;
  if (rounds_out) *rounds_out = 0;
  if (num_converged_out) *num_converged_out = 0;
  if (K == 0) return 0;
  if (ensure_work(ctx, K)) return 1;
  REQUIRE(!kf_sharded(ctx) || is_sharded(ctx), "keyframe sharding needs an all-reduce hook or an RCCL communicator");
  launch_pose_init_from_keyframes(ctx->stream, ctx->dev_kfs, K, ctx->dev_work, ctx->dev_Hb, ctx->pinned_work, ctx->kf_rank, ctx->kf_world);
  CHECK_LAUNCH();
  if (run_pose_rounds(ctx, use_depth != 0, use_desc != 0, ctx->dev_kfs, ctx->dev_kfs, ctx->dev_work, ctx->dev_Hb, K,
                      make_view(surfels), /*write_back*/ 1, update_activation ? 1 : 0, ctx->pinned_work, rounds_out, /*schedule*/ true,
                      &ctx->rounds_hint_table)) return 1;
  const PoseWork* hw = ctx->pinned_work;
  const int* counters = reinterpret_cast<const int*>(hw + K);
  for (int k = 0; k < K; ++k) {
    if (hw[k].iterations > 0) fill_pose(&ctx->host_kfs[k], hw[k].T);
    if (global_T_frame_out) memcpy(global_T_frame_out + 7 * k, ctx->host_kfs[k].global_T_frame, 7 * sizeof(float));
    if (iterations_done) iterations_done[k] = hw[k].iterations;
    if (converged) converged[k] = hw[k].converged;
    if (moved_out) moved_out[k] = update_activation ? hw[k].moved : 0;
  }
  if (update_activation && num_converged_out) *num_converged_out = counters[kPoseCounterConverged];
  return 0;
}

int bahip_estimate_keyframe_poses(bahip_context* ctx, int use_depth, int use_desc, const bahip_surfels* surfels,
                                  float* global_T_frame_out, int* iterations_done, int* converged, int* rounds_out) {
  return estimate_keyframe_poses_impl(ctx, use_depth, use_desc, surfels, global_T_frame_out, iterations_done, converged, rounds_out,
                                      false, nullptr, nullptr);
}

int bahip_estimate_keyframe_poses_and_update_activation(bahip_context* ctx, int use_depth, int use_desc, const bahip_surfels* surfels,
                                                        float* global_T_frame_out, int* iterations_done, int* converged, int* moved,
                                                        int* rounds_out, int* num_converged_out) {
  return estimate_keyframe_poses_impl(ctx, use_depth, use_desc, surfels, global_T_frame_out, iterations_done, converged, rounds_out,
                                      true, moved, num_converged_out);
}

// ---- the alternating loop, driven by the device (include/badslam_hip.h) ---------------------------------------------------------
}  // extern "C"
namespace bahip_capi {
int g_device_loop_enabled = bahip_env_int("BAHIP_DEVICE_LOOP", 1) != 0 ? 1 : 0;
}  // namespace bahip_capi
namespace {
constexpr int kLoopLogSlots = 4096;
}
extern "C" {
int bahip_debug_set_device_loop(int enabled) { g_device_loop_enabled = enabled ? 1 : 0; return 0; }
static std::atomic<long long> g_loop_calls_handled{0}, g_loop_calls_declined{0};
int bahip_debug_alternating_loop_calls(long long* handled_out, long long* declined_out) {
  if (handled_out) *handled_out = g_loop_calls_handled.load();
  if (declined_out) *declined_out = g_loop_calls_declined.load();
  return 0;
}
int bahip_debug_set_pcg_lds_form(int mode) { set_pcg_lds_form(mode); return 0; }
int bahip_alternating_iterations(bahip_context* ctx, const bahip_alternating_options* opt, const bahip_surfels* surfels,
                                 float* global_T_frame_out, int* activation_out, int* handled_out, int* iterations_done_out,
                                 int* converged_out, int* pose_rounds_out, int* pose_steps_out, int* not_converged_out) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(opt != nullptr && handled_out != nullptr, "bahip_alternating_iterations: NULL argument");
  ctx->lifecycle_bounds_tiles = 0;
  REQUIRE(opt->use_depth_residuals || opt->use_descriptor_residuals, "at least one residual type must be enabled");
  const int K = ctx->num_kfs;
  *handled_out = 0;
  if (iterations_done_out) *iterations_done_out = 0;
  if (converged_out) *converged_out = 0;
  if (pose_rounds_out) *pose_rounds_out = 0;
  if (pose_steps_out) *pose_steps_out = 0;
  if (not_converged_out) *not_converged_out = 0;
  static const bool say_why = getenv("BADSLAM_HOST_TIMING") != nullptr;
  auto now_us = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_enter = say_why ? now_us() : 0;
  double t_queued = 0, t_waited = 0;
  if (say_why)
    fprintf(stderr, "[bahip_alternating_iterations] enabled %d K %d kf_sharded %d max_it %d queued-ahead %d hook %d\n", g_device_loop_enabled, K,
            (int)kf_sharded(ctx), opt->max_iterations, (int)pose_round_can_be_queued_ahead(surfels->surfels_size, K, true), ctx->allreduce != nullptr ? 1 : 0);
  if (!g_device_loop_enabled || K == 0 || kf_sharded(ctx) || opt->max_iterations <= 0 || !pose_round_can_be_queued_ahead(surfels->surfels_size, K, true) ||
      ctx->allreduce != nullptr)
    ++g_loop_calls_declined;   // (bahip_debug_alternating_loop_calls: a caller that expects the device loop can check that it got it)
  if (!g_device_loop_enabled || K == 0 || kf_sharded(ctx) || opt->max_iterations <= 0 || !pose_round_can_be_queued_ahead(surfels->surfels_size, K, true)) return 0;
  // With a HOST all-reduce hook every queued round is a stream synchronisation plus a host collective -- also the rounds queued
  // behind the iteration that ended the loop, which exchange zeros: the host loop, which knows when to stop, serves
  // that configuration.  The native RCCL path (collectives enqueued on the stream) keeps the device-driven loop.
  if (ctx->allreduce != nullptr) return 0;
  REQUIRE(surfels->active != nullptr, "the alternating loop needs the active-surfel buffer");
  REQUIRE(ctx->have_covisibility && (int)ctx->covis_offsets.size() == K + 1, "bahip_set_covisibility must follow bahip_set_keyframes");
  REQUIRE(!opt->fixed_window || (int)ctx->window.size() == K, "bahip_set_activation_window must follow bahip_set_keyframes");
  REQUIRE(opt->activation_surfels_size <= surfels->surfels_size, "activation range exceeds surfels_size");
  if (ensure_work(ctx, K)) return 1;
  if (!ctx->dev_loop_ctl || !ctx->host_loop_ctl) {   // both or neither: a call that got only the first must not leave it behind
    if (!ctx->dev_loop_ctl) HIP_TRY(hipMalloc(&ctx->dev_loop_ctl, sizeof(int) * kLoopWords));
    if (hipHostMalloc(&ctx->host_loop_ctl, sizeof(int) * (kLoopWords + kLoopLogSlots), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      ctx->host_loop_ctl = nullptr;
      hipFree(ctx->dev_loop_ctl);
      ctx->dev_loop_ctl = nullptr;
      return fail("hipHostMalloc of the loop control words failed", __FILE__, __LINE__);
    }
  }
  const SurfelsView sv = make_view(surfels);
  if (ensure_tile_bounds(ctx, sv.size)) return 1;
  const bool use_depth = opt->use_depth_residuals != 0, use_desc = opt->use_descriptor_residuals != 0;
  hipStream_t st = ctx->stream;
  HIP_TRY(hipMemsetAsync(ctx->dev_loop_ctl, 0, sizeof(int) * kLoopWords, st));
  memset(ctx->host_loop_ctl, 0, sizeof(int) * kLoopWords);
  if (ctx->profiling == 1) for (int stage = 1; stage <= 3; ++stage) { ctx->timers[stage].used = 0; ctx->timers[stage].units = 0; }   // "the last call"
  const int* stop = ctx->dev_loop_ctl + kLoopStop;
  const int* dev_counters = reinterpret_cast<const int*>(ctx->dev_work + K);
  const int* counters = reinterpret_cast<const int*>(ctx->pinned_work + K);
  const int* csr = ctx->dev_covis_csr;
  static std::atomic<int> g_loop_sequence{1 << 30};   // disjoint from run_pose_rounds' numbers (which count up from 1)
  const uint32_t padded_tiles = pose_padded_tiles(sv.size);
  // Rounds queued per pose phase.  A phase right after something changed (a new keyframe, a loop closure) needs three or four
  // Gauss-Newton rounds, the phases behind it fewer, the steady state one or two: the first phase queued here gets what the
  // phases at the end of the previous call needed (rounds_hint_table) or what the phase handed to the host just took, every
  // following phase one round less, down to the steady-state floor.  A round queued in vain costs two near-empty launches and
  // their dependencies (~20 us, and an exchange of zeros when sharded); a phase with too few rounds costs one host reaction.
  const bool rounds_forced = g_pose_rounds_ahead > 0;
  int rounds_ahead = rounds_forced ? g_pose_rounds_ahead : std::max(1, std::min(ctx->rounds_hint_table, 4));
  int rounds_floor = rounds_forced ? rounds_ahead : std::min(rounds_ahead, 2);
  std::vector<int> queued_rounds;    // per queued iteration of the current batch
  int last_needed[2] = {0, 0};       // rounds the last two completed phases needed
  int it = 0, done_before = 0, rounds_before = 0;
  bool converged = false;
  // Under surfel sharding every rank must queue the SAME rounds (each is a collective): the schedule may depend on nothing but what
  // all ranks hold alike -- the loop's control words on the device, identical everywhere because the sums are exchanged and the
  // solve is replicated.  The per-round log in mapped host memory is not used then (whether a rank can poll it, poll_disabled, is a
  // property of that rank's runtime: ADVICE r4, ranks that differed in it would have queued different numbers of collectives).
  const bool rank_invariant_schedule = is_sharded(ctx);
  while (it < opt->max_iterations) {
    // heavy work first (wave_cull.h): the first phase queued here takes the census when one is due
    constexpr int kSchedulePhases = 32;
    bool schedule = g_tile_order_enabled && sv.size > 0 && !ctx->tile_order_unavailable_for(padded_tiles) &&
                    (ctx->tile_order_tiles != padded_tiles || ++ctx->phases_since_schedule >= kSchedulePhases);
    if (schedule && ensure_tile_schedule(ctx, padded_tiles)) return 1;
    StageTimer& acc_timer = ctx->timers[2];
    const int acc_mark = acc_timer.used;
    int log_slot = 0, sequence = 0;
    bool begun_by_previous = false;
    queued_rounds.clear();
    for (int i = it; i < opt->max_iterations; ++i) {
      const int phase_rounds = std::max(rounds_floor, rounds_ahead - (i - it));
      queued_rounds.push_back(phase_rounds);
      // window / propagation (which closes iteration i - 1, B/direct_ba_alternating.cc:703-709) and the pose phase's work items
      // (done already by the launch that ended iteration i - 1's pose phase when that launch could take it along: begun_by_previous)
      const int begin_mode = opt->fixed_window ? 1 : (i > 0 ? 2 : 0);
      const bool begun = begun_by_previous ||
                         launch_iteration_begin(st, ctx->dev_kfs, K, begin_mode, ctx->dev_window, csr, csr + K + 1, ctx->dev_work, ctx->dev_Hb, ctx->pinned_work, stop);
      begun_by_previous = false;
      if (!begun) {
        if (begin_mode == 1) launch_window_activation(st, ctx->dev_kfs, K, ctx->dev_window, csr, csr + K + 1, stop);
        else if (begin_mode == 2) launch_propagate_covisible(st, ctx->dev_kfs, K, csr, csr + K + 1, stop);
      }
      timer_begin(ctx, 1, true);
      launch_geometry(st, use_depth, use_desc, ctx->in, ctx->dev_kfs, K, sv, opt->activate_in_geometry ? (long long)opt->activation_surfels_size : -1,
                      tile_order_for(ctx, sv.size), stop);
      timer_end(ctx, 1);
      if (!begun) launch_pose_init_from_keyframes(st, ctx->dev_kfs, K, ctx->dev_work, ctx->dev_Hb, ctx->pinned_work, 0, 1, stop);
      CHECK_LAUNCH();
      for (int r = 0; r < phase_rounds; ++r) {
        timer_begin(ctx, 2, false, 0);
        launch_pose_accumulate(st, use_depth, use_desc, ctx->in, ctx->dev_kfs, ctx->dev_work, K, sv, ctx->dev_Hb, ctx->dev_tile_bounds,
                               /*stored_bounds*/ r > 0, /*num_listed: upper bound*/ K, ctx->dev_tile_counters, &ctx->pose_parity,
                               (schedule && r == 0) ? ctx->dev_tile_cost : nullptr, tile_order_for(ctx, sv.size),
                               r > 0 ? dev_counters + (r - 1) : nullptr, stop);
        timer_end(ctx, 2);
        CHECK_LAUNCH();
        if (schedule && r == 0) {
          if (launch_tile_order(st, ctx->dev_tile_cost, padded_tiles, ctx->dev_tile_order)) {
            ctx->tile_order_tiles = padded_tiles;
            ctx->phases_since_schedule = 0;
            CHECK_LAUNCH();
          } else {
            ctx->tile_order_unavailable_tiles = padded_tiles;
            HIP_TRY(hipMemsetAsync(ctx->dev_tile_cost, 0, sizeof(uint32_t) * padded_tiles, st));
          }
          schedule = false;
        }
        if (reduce_over_ranks(ctx, ctx->dev_Hb, (size_t)K * kHbStride, BAHIP_SUM_I64)) return 1;
        PoseLoopControl loop;
        loop.ctl = ctx->dev_loop_ctl; loop.host_ctl = ctx->host_loop_ctl;
        loop.phase_end = r == phase_rounds - 1 ? 1 : 0;
        loop.publish = (i == opt->max_iterations - 1 && r == phase_rounds - 1) ? 1 : 0;   // the launch the host waits for (below) publishes for all
        loop.iteration = i; loop.min_iterations = opt->min_iterations;
        loop.round_log = log_slot < kLoopLogSlots ? ctx->host_loop_ctl + kLoopWords : nullptr;
        loop.log_slot = log_slot++;
        if (loop.phase_end && begun && g_fused_iteration_begin && K <= 1024 && i + 1 < opt->max_iterations) {
          loop.next_mode = opt->fixed_window ? 1 : 2;
          loop.in_window = ctx->dev_window; loop.covis_offsets = csr; loop.covis_indices = csr + K + 1;
          begun_by_previous = true;
        }
        timer_begin(ctx, 3, false);
        sequence = ++g_loop_sequence;
        launch_pose_solve(st, ctx->dev_work, K, ctx->dev_Hb, ctx->dev_kfs, /*write_back*/ 1, /*update_activation*/ 1, r, ctx->pinned_work, sequence, &loop);
        timer_end(ctx, 3);
        CHECK_LAUNCH();
      }
    }
    if (say_why && t_queued == 0) t_queued = now_us();
    if (wait_for_pose_sequence(ctx, ctx->pinned_work, ctx->dev_work, K, sequence)) return 1;
    if (say_why && t_waited == 0) t_waited = now_us();
    if (ctx->poll_disabled) HIP_TRY(hipMemcpy(ctx->host_loop_ctl, ctx->dev_loop_ctl, sizeof(int) * kLoopWords, hipMemcpyDeviceToHost));
    if (counters[kPoseCounterInvalid] || ctx->host_loop_ctl[kLoopInvalid])
      return fail("pose normal equations: a tile total was not finite or reached 2^52 (hb_split), or a sum left the fixed-point range; the "
                  "surfels or images hold non-finite values", __FILE__, __LINE__);
    const int* ctl = ctx->host_loop_ctl;
    // the stage timers describe launches that did work: the log says how many work items every queued round iterated
    if (timer_on(ctx, 2)) {
      const int* log = ctl + kLoopWords;
      for (int j = 0; j < log_slot && acc_mark + j < acc_timer.used; ++j) {
        if (j < kLoopLogSlots && !ctx->poll_disabled) {
          if (log[j] == 0) acc_timer.skip[acc_mark + j] = 1;
          else acc_timer.units += log[j];
        }
      }
    }
    const int completed = ctl[kLoopIterationsDone] - done_before;
    done_before = ctl[kLoopIterationsDone];
    const int rounds_now = ctl[kLoopRounds];
    if (rank_invariant_schedule) {
      if (completed > 0) {   // rounds with work per completed phase, rounded up (the control words: the same on every rank)
        last_needed[0] = last_needed[1];
        last_needed[1] = std::max(1, (rounds_now - rounds_before + completed - 1) / completed);
      }
    } else if (!ctx->poll_disabled) {
      // rounds the completed phases needed: the log holds the work items every queued round iterated
      const int* log = ctl + kLoopWords;
      int slot = 0;
      for (int j = 0; j < completed && j < (int)queued_rounds.size(); ++j) {
        int needed = 0;
        for (int r = 0; r < queued_rounds[j] && slot + r < kLoopLogSlots; ++r) if (log[slot + r] > 0) needed = r + 1;
        slot += queued_rounds[j];
        if (slot > kLoopLogSlots) break;
        last_needed[0] = last_needed[1];
        last_needed[1] = std::max(1, needed);
      }
    }
    rounds_before = rounds_now;
    const int handed_over_rounds = completed < (int)queued_rounds.size() ? queued_rounds[completed] : rounds_ahead;
    it += completed;
    if (ctl[kLoopStop] == 1) { converged = true; break; }
    if (ctl[kLoopStop] == 2) {
      // iteration `it`'s pose phase has work items left after the rounds queued for it: the host finishes it round by round,
      // applies the loop's stopping rule itself, and queues what is left with more rounds per phase
      HIP_TRY(hipMemsetAsync(ctx->dev_loop_ctl + kLoopStop, 0, sizeof(int), st));
      PoseLoopControl totals;
      totals.ctl = ctx->dev_loop_ctl; totals.host_ctl = ctx->host_loop_ctl;
      int more_rounds = 0;
      if (run_pose_rounds(ctx, use_depth, use_desc, ctx->dev_kfs, ctx->dev_kfs, ctx->dev_work, ctx->dev_Hb, K, sv, 1, 1, ctx->pinned_work, &more_rounds,
                          false, nullptr, handed_over_rounds, counters[handed_over_rounds - 1], &totals)) return 1;
      const bool all_converged = counters[kPoseCounterConverged] == K;
      const bool ends_loop = it >= opt->min_iterations - 1 && all_converged;
      it += 1;
      last_needed[0] = last_needed[1];
      last_needed[1] = handed_over_rounds + more_rounds;
      // the next phase gets what this one took (it decays from there)
      if (!rounds_forced) { rounds_ahead = std::max(1, std::min(handed_over_rounds + more_rounds, 8)); rounds_floor = std::min(rounds_ahead, 2); }
      if (ends_loop) { converged = true; break; }
      continue;
    }
    break;   // every queued iteration ran
  }
  // the table after the last pose phase: poses and activations
  HIP_TRY(hipMemcpyAsync(ctx->host_kfs.data(), ctx->dev_kfs, sizeof(KfEntry) * K, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ctx->host_loop_ctl, ctx->dev_loop_ctl, sizeof(int) * kLoopWords, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int k = 0; k < K; ++k) {
    if (global_T_frame_out) memcpy(global_T_frame_out + 7 * k, ctx->host_kfs[k].global_T_frame, 7 * sizeof(float));
    if (activation_out) activation_out[k] = ctx->host_kfs[k].activation;
  }
  const int rounds_total = ctx->host_loop_ctl[kLoopRounds];
  if (last_needed[1] > 0) ctx->rounds_hint_table = std::max(last_needed[0], last_needed[1]);
  else if (it > 0) ctx->rounds_hint_table = std::max(1, (rounds_total + it - 1) / it);
  *handled_out = 1;
  ++g_loop_calls_handled;
  if (say_why)
    fprintf(stderr, "[bahip_alternating_iterations, us] queueing %.0f | first wait %.0f | rest (hand-overs, read-back) %.0f | iterations %d\n",
            t_queued - t_enter, t_waited - t_queued, now_us() - t_waited, it);
  if (iterations_done_out) *iterations_done_out = it;
  if (converged_out) *converged_out = converged ? 1 : 0;
  if (pose_rounds_out) *pose_rounds_out = rounds_total;
  if (pose_steps_out) *pose_steps_out = ctx->host_loop_ctl[kLoopSteps];
  if (not_converged_out) *not_converged_out = ctx->host_loop_ctl[kLoopNotConverged];
  return 0;
}

int bahip_set_covisibility(bahip_context* ctx, const int* offsets, const int* indices, int num_keyframes) {
  REQUIRE(num_keyframes == ctx->num_kfs, "bahip_set_covisibility: list count differs from the bound keyframes");
  REQUIRE(offsets != nullptr && offsets[0] == 0, "bahip_set_covisibility: offsets must start at 0");
  const int K = num_keyframes, total = offsets[K];
  REQUIRE(total >= 0 && (total == 0 || indices != nullptr), "bahip_set_covisibility: bad lists");
  for (int k = 0; k < K; ++k) REQUIRE(offsets[k + 1] >= offsets[k], "bahip_set_covisibility: offsets must be non-decreasing");
  for (int j = 0; j < total; ++j) REQUIRE(indices[j] >= 0 && indices[j] < K, "bahip_set_covisibility: keyframe index out of range");
  ctx->covis_offsets.assign(offsets, offsets + K + 1);
  ctx->covis_indices.assign(indices, indices + total);
  const size_t need = (size_t)K + 1 + (size_t)total;
  if (need > ctx->covis_csr_capacity) {
    int* grown = nullptr;
    HIP_TRY(hipMalloc(&grown, sizeof(int) * (need + 1024)));
    hipFree(ctx->dev_covis_csr);   // only the CSR buffer is re-grown here (tile bounds and window have their own grow paths)
    ctx->dev_covis_csr = grown;
    ctx->covis_csr_capacity = need + 1024;
  }
  if (stage_upload(&ctx->stage_covis, ctx->dev_covis_csr, ctx->covis_offsets.data(), sizeof(int) * (size_t)(K + 1), ctx->stream,
                   ctx->covis_indices.data(), sizeof(int) * (size_t)total, ctx->dev_covis_csr + K + 1)) return 1;   // (no host wait: capi_internal.h)
  ctx->have_covisibility = true;
  return 0;
}

int bahip_set_activation_window(bahip_context* ctx, const uint8_t* in_window, int num_keyframes) {
  REQUIRE(num_keyframes == ctx->num_kfs && (in_window != nullptr || num_keyframes == 0), "bahip_set_activation_window: one flag per bound keyframe");
  ctx->window.assign(in_window, in_window + (in_window ? num_keyframes : 0));
  if ((size_t)num_keyframes > ctx->window_capacity) {
    uint8_t* grown = nullptr;
    HIP_TRY(hipMalloc(&grown, (size_t)num_keyframes + 256));
    hipFree(ctx->dev_window);
    ctx->dev_window = grown;
    ctx->window_capacity = (size_t)num_keyframes + 256;
  }
  if (num_keyframes && stage_upload(&ctx->stage_window, ctx->dev_window, ctx->window.data(), (size_t)num_keyframes, ctx->stream)) return 1;
  return 0;
}

int bahip_apply_activation_window(bahip_context* ctx) {
  REQUIRE((int)ctx->window.size() == ctx->num_kfs, "bahip_set_activation_window must follow bahip_set_keyframes");
  const int K = ctx->num_kfs;
  REQUIRE(ctx->have_covisibility && (int)ctx->covis_offsets.size() == K + 1, "bahip_set_covisibility must follow bahip_set_keyframes");
  launch_window_activation(ctx->stream, ctx->dev_kfs, K, ctx->dev_window, ctx->dev_covis_csr, ctx->dev_covis_csr + K + 1);
  CHECK_LAUNCH();
  return 0;
}

int bahip_propagate_covisible_activation(bahip_context* ctx) {
  REQUIRE(ctx->have_covisibility && (int)ctx->covis_offsets.size() == ctx->num_kfs + 1,
          "bahip_set_covisibility must follow bahip_set_keyframes before the activation can be propagated");
  const int K = ctx->num_kfs;
  // (the activation field of the host-side copy of the table is "as bound": only the device table follows the state machine)
  launch_propagate_covisible(ctx->stream, ctx->dev_kfs, K, ctx->dev_covis_csr, ctx->dev_covis_csr + K + 1);
  CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
