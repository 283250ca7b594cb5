// capi_solvers.hip -- the intrinsics step of the alternating scheme and the PCG scheme (one outer iteration, and stage by stage for callers
// of B/kernels.h:397-491) behind the C boundary.
#include "capi_internal.h"

using namespace bahip;
using namespace bahip_capi;

extern "C" {
int bahip_optimize_intrinsics(bahip_context* ctx, int optimize_depth, int optimize_color, const bahip_surfels* surfels,
                              bahip_camera* out_color_camera, bahip_camera* out_depth_camera, float* out_a) {
  REQUIRE_NO_KF_SHARDING("bahip_optimize_intrinsics");
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(optimize_depth || optimize_color, "at least one of depth / colour intrinsics must be optimised");  // :55
  *out_color_camera = ctx->color_cam;
  *out_depth_camera = ctx->depth_cam;
  *out_a = ctx->dp.a;
  if (surfels->surfels_size == 0 && !is_sharded(ctx)) return 0;   // a rank with an empty shard still takes part in the exchange
  const int S = ctx->in.cf_width * ctx->in.cf_height;
  if (S > ctx->intr_capacity) {
    const int cap = S + 1024;
    float* grown = nullptr;
    // doubles first (8-byte aligned): glob_d[64] | cells_d[8 cap] | then floats: glob_f[64] | cells_f[8 cap] | Schur partials
    HIP_TRY(hipMalloc(&grown, sizeof(double) * (64 + 8 * (size_t)cap) + sizeof(float) * (64 + 8 * (size_t)cap + intrinsics_schur_partials(cap))));
    hipFree(ctx->intr_scratch);
    ctx->intr_scratch = grown;
    ctx->intr_capacity = cap;
  }
  double* glob_d = reinterpret_cast<double*>(ctx->intr_scratch);   // 34 sums
  double* cells_d = glob_d + 64;                                   // S records {B0..B4, D, b2, observation count}
  float* glob = reinterpret_cast<float*>(cells_d + 8 * (size_t)ctx->intr_capacity);   // the 34 sums rounded (+ Schur); x1 at [40..44]
  float* cells = glob + 64;
  float* partials = cells + 8 * (size_t)ctx->intr_capacity;
  // Append buffers for the per-cell records (kernels_intrinsics.hip).  A record exists per associated pair with a depth residual (52 M
  // at the bench size, 1.0 G at BASELINE configs[4]: 53 GB when one set of buffers had to hold them all, round 4).  On a LARGE cloud the
  // sweep therefore runs in SLICES of its schedule: slice p appends to buffer set p & 1, and the reduction of that set is queued on a
  // second stream while slice p + 1 sweeps into the other set, so the buffers hold the records of one slice at a time (configs[4]: 30 GB
  // in 8 slices where one set took 53 GB, the same 36 ms per step).  Measured in round 5 (profiles/r5_intrinsics_slices.txt): the
  // reduction does NOT overlap the next slice's sweep -- the sweep's wavefronts hold 504 of a SIMD's 512 vector registers, a workgroup of
  // the reduction finds no room until the sweep drains -- and at the bench size slices cost time (1.99 ms in one piece, 2.33-2.57 ms in
  // 2-5 slices: every slice ends in its own tail), so small clouds keep one slice.  A call that finds the buffers too small still gives
  // the same result: the records that do not fit go out as atomics.  The sums do not depend on the slicing (binary64 sums of
  // binary32 terms: kernels_intrinsics.hip "DEFINITION").
  IntrBins bins{nullptr, nullptr, 0, 1};
  const int num_bins = intrinsics_bin_count(ctx->in, &bins.bins_x);
  const uint32_t* sched = tile_order_for(ctx, surfels->surfels_size);
  const uint32_t positions = intrinsics_sweep_positions(surfels->surfels_size, sched);
  // one slice below 131072 tiles (8.4 M surfels); beyond, slices of >= 32768 positions, at most 8; multiples of 8 keep the position -> XCD deal
  int slices = ctx->intr_slices_forced > 0 ? ctx->intr_slices_forced
                                           : (positions < 131072u ? 1 : (int)std::min<uint32_t>(8u, positions / 32768u));
  if (!optimize_depth) slices = 1;   // no records, nothing to overlap
  const uint32_t per_slice = ((positions + (uint32_t)slices - 1) / (uint32_t)slices + 7u) & ~7u;
  const int sets = slices > 1 ? 2 : 1;
  if (optimize_depth) {
    if (num_bins != ctx->intr_bin_count || sets != ctx->intr_bin_sets) {
      hipFree(ctx->intr_bin_cursors); hipHostFree(ctx->intr_bin_counts_host); hipFree(ctx->intr_bin_records);
      ctx->intr_bin_cursors = nullptr; ctx->intr_bin_counts_host = nullptr; ctx->intr_bin_records = nullptr;
      ctx->intr_bin_capacity = 0; ctx->intr_bin_wanted = 0;
      HIP_TRY(hipMalloc(&ctx->intr_bin_cursors, sizeof(uint32_t) * (size_t)num_bins * 2));
      HIP_TRY(hipHostMalloc(&ctx->intr_bin_counts_host, sizeof(uint32_t) * (size_t)num_bins * kIntrMaxSlices));   // one row of counts per slice
      ctx->intr_bin_count = num_bins;
      ctx->intr_bin_sets = sets;
    }
    // Keep what there is unless the previous call OVERFLOWED it (intr_bin_wanted is raised only then).  Round 4, configs[4]: sized
    // as "the previous call's largest count + 25 %" the request crept up by 64 records per call while the poses converged, and each
    // time 53 GB of record buffers were freed and allocated again -- 1.5 to 2.5 s per reallocation, in whichever call it fell
    // (gpurun_out/r4_call30: 3.4 BA iterations/s with one of them inside the timed call, 14.4 without).
    uint64_t want = std::max<uint64_t>(ctx->intr_bin_capacity, ctx->intr_bin_wanted);
    if (!want) want = (uint64_t)surfels->surfels_size * (uint64_t)std::min(ctx->num_kfs, 16) * 2 / ((uint64_t)num_bins * (uint64_t)slices) + 4096;
    if (ctx->intr_bin_forced >= 0) want = (uint64_t)ctx->intr_bin_forced;
    const uint64_t limit = (96ull << 30) / (intrinsics_bin_record_bytes() * (uint64_t)num_bins * (uint64_t)sets);   // at most 96 GB of records
    want = std::min(want, limit);
    if (want > ctx->intr_bin_capacity || (ctx->intr_bin_forced >= 0 && want != ctx->intr_bin_capacity)) {
      static const bool host_timing = getenv("BADSLAM_HOST_TIMING") != nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      hipFree(ctx->intr_bin_records);
      ctx->intr_bin_records = nullptr; ctx->intr_bin_capacity = 0;
      const uint64_t cap = (want + 63) / 64 * 64;
      if (cap) HIP_TRY(hipMalloc(&ctx->intr_bin_records, intrinsics_bin_record_bytes() * cap * (uint64_t)num_bins * (uint64_t)sets));
      ctx->intr_bin_capacity = (uint32_t)cap;
      if (host_timing)
        fprintf(stderr, "[intrinsics record buffers] %d set(s) x %d buffers x %llu records = %.2f GB (re)allocated in %.1f ms (%d slices)\n", sets, num_bins,
                (unsigned long long)cap, (double)(intrinsics_bin_record_bytes() * cap * (uint64_t)num_bins * (uint64_t)sets) / 1e9,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), slices);
    }
    bins.cursors = ctx->intr_bin_cursors; bins.records = ctx->intr_bin_records; bins.capacity = ctx->intr_bin_capacity;
  }
  hipStream_t st = ctx->stream;
  timer_begin(ctx, 4, true);
  HIP_TRY(hipMemsetAsync(glob_d, 0, sizeof(double) * (64 + 8 * (size_t)S), st));
  const SurfelsView sv = make_view(surfels);
  if (slices == 1) {
    if (bins.capacity) HIP_TRY(hipMemsetAsync(bins.cursors, 0, sizeof(uint32_t) * (size_t)num_bins, st));
    timer_begin(ctx, 6, true);
    launch_intrinsics_accumulate(st, optimize_depth != 0, optimize_color != 0, ctx->in, ctx->dev_kfs, ctx->num_kfs, sv, glob_d, cells_d, bins, sched);
    timer_end(ctx, 6);
    timer_begin(ctx, 7, true);
    launch_intrinsics_bin_reduce(st, optimize_depth != 0, ctx->in, sv, cells_d, bins);
    timer_end(ctx, 7);
    CHECK_LAUNCH();
    if (bins.capacity) HIP_TRY(hipMemcpyAsync(ctx->intr_bin_counts_host, bins.cursors, sizeof(uint32_t) * (size_t)num_bins, hipMemcpyDeviceToHost, st));
  } else {
    // second stream (higher priority: its workgroups take the slots the sweep's finished wavefronts free) and the events that order the
    // two; created on first use
    if (!ctx->intr_aux_stream) {
      int least = 0, greatest = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
      HIP_TRY(hipStreamCreateWithPriority(&ctx->intr_aux_stream, hipStreamNonBlocking, greatest));
      for (int e = 0; e < 4; ++e) HIP_TRY(hipEventCreateWithFlags(&ctx->intr_events[e], hipEventDisableTiming));
    }
    hipStream_t aux = ctx->intr_aux_stream;
    hipEvent_t* swept = ctx->intr_events;        // [set]: the slice's records are complete
    hipEvent_t* reduced = ctx->intr_events + 2;  // [set]: the set's records have been added up, the buffers are free
    timer_begin(ctx, 6, true);
    for (int p = 0; p < slices; ++p) {
      const int set = p & 1;
      IntrBins mine = bins;
      if (bins.capacity) {
        mine.cursors = bins.cursors + (size_t)set * num_bins;
        mine.records = bins.records + (size_t)set * num_bins * (intrinsics_bin_record_bytes() / sizeof(uint32_t)) * bins.capacity;
      }
      if (p >= 2) HIP_TRY(hipStreamWaitEvent(st, reduced[set], 0));
      if (mine.capacity) HIP_TRY(hipMemsetAsync(mine.cursors, 0, sizeof(uint32_t) * (size_t)num_bins, st));
      launch_intrinsics_accumulate(st, true, optimize_color != 0, ctx->in, ctx->dev_kfs, ctx->num_kfs, sv, glob_d, cells_d, mine, sched,
                                   (uint32_t)p * per_slice, per_slice);
      CHECK_LAUNCH();
      HIP_TRY(hipEventRecord(swept[set], st));
      HIP_TRY(hipStreamWaitEvent(aux, swept[set], 0));
      launch_intrinsics_bin_reduce(aux, true, ctx->in, sv, cells_d, mine);
      CHECK_LAUNCH();
      if (mine.capacity)
        HIP_TRY(hipMemcpyAsync(ctx->intr_bin_counts_host + (size_t)p * num_bins, mine.cursors, sizeof(uint32_t) * (size_t)num_bins, hipMemcpyDeviceToHost, aux));
      HIP_TRY(hipEventRecord(reduced[set], aux));
    }
    HIP_TRY(hipStreamWaitEvent(st, reduced[0], 0));
    HIP_TRY(hipStreamWaitEvent(st, reduced[1], 0));
    timer_end(ctx, 6);   // (sweeps and reductions together: the reductions have no events of their own on this stream)
  }
  if (reduce_over_ranks(ctx, glob_d, 64 + 8 * (size_t)S, BAHIP_SUM_F64)) return 1;
  launch_intrinsics_finish(ctx->stream, optimize_depth != 0, S, glob_d, cells_d, glob, cells, partials);
  CHECK_LAUNCH();
  timer_end(ctx, 4);   // the sweep and the Schur complement; the 5x5 / 4x4 solves and the cfactor update that follow are tiny
  HIP_TRY(hipMemcpyAsync(ctx->pinned_f, glob, 34 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  {
    static const bool host_timing = getenv("BADSLAM_HOST_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    const double waited = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (host_timing && waited > 100.0) fprintf(stderr, "[intrinsics step] waited %.1f ms for the stream (capacity %u per buffer)\n", waited, bins.capacity);
  }
  ctx->intr_bin_rows = bins.capacity ? slices : 0;
  if (bins.capacity) {
    uint32_t most = 0;
    for (size_t b = 0; b < (size_t)num_bins * (size_t)slices; ++b) most = std::max(most, ctx->intr_bin_counts_host[b]);
    if (most > bins.capacity) ctx->intr_bin_wanted = std::max(ctx->intr_bin_wanted, most + most / 4 + 1024);   // the next call regrows
    ctx->intr_bin_last_overflow = most > bins.capacity ? 1 : 0;
  }
  const float* g = ctx->pinned_f;
  if (optimize_depth) {
    double M[25], rhs[5], x[5];
    int q = 0;
    for (int row = 0; row < 5; ++row)
      for (int col = row; col < 5; ++col) { M[row * 5 + col] = g[q]; M[col * 5 + row] = g[q]; ++q; }
    for (int c = 0; c < 5; ++c) rhs[c] = g[15 + c];
    // weak prior pulling a towards zero (B/kernel_opt_intrinsics.cc:153-158); added in binary32 like the reference
    constexpr float kAPriorWeight = 10;
    M[24] = (double)((float)M[24] + kAPriorWeight * kAPriorWeight);
    rhs[4] = (double)((float)rhs[4] + kAPriorWeight * kAPriorWeight * ctx->dp.a);
    ldlt_solve_sym<5>(M, rhs, x);
    float x1[5];
    for (int c = 0; c < 5; ++c) x1[c] = (float)x[c];
    const float new_fx = 1.0f / (ctx->in.fx_inv - x1[0]);
    const float new_fy = 1.0f / (ctx->in.fy_inv - x1[1]);
    out_depth_camera->fx = new_fx;
    out_depth_camera->fy = new_fy;
    out_depth_camera->cx = -(new_fx * (ctx->in.cx_inv - x1[2])) + 0.5f;
    out_depth_camera->cy = -(new_fy * (ctx->in.cy_inv - x1[3])) + 0.5f;
    *out_a = ctx->dp.a - x1[4];
    memcpy(ctx->pinned_f + 40, x1, sizeof(x1));
    HIP_TRY(hipMemcpyAsync(glob + 40, ctx->pinned_f + 40, sizeof(x1), hipMemcpyHostToDevice, ctx->stream));
    launch_intrinsics_solve_cells(ctx->stream, ctx->in, S, cells, glob + 40, ctx->dp.cfactor, ctx->dp.cfactor_pitch_bytes);
    CHECK_LAUNCH();
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  if (optimize_color) {
    double M[16], rhs[4], x[4];
    int q = 20;
    for (int row = 0; row < 4; ++row)
      for (int col = row; col < 4; ++col) { M[row * 4 + col] = g[q]; M[col * 4 + row] = g[q]; ++q; }
    for (int c = 0; c < 4; ++c) rhs[c] = g[30 + c];
    ldlt_solve_sym<4>(M, rhs, x);
    out_color_camera->fx = ctx->color_cam.fx - (float)x[0];
    out_color_camera->fy = ctx->color_cam.fy - (float)x[1];
    out_color_camera->cx = ctx->color_cam.cx - (float)x[2];
    out_color_camera->cy = ctx->color_cam.cy - (float)x[3];
  }
  return 0;
}
// One outer Gauss-Newton iteration of the PCG scheme: B/direct_ba_pcg.cc:229-646.
static int ensure_pcg_exact(bahip_context* ctx, uint32_t head_count) {
  const size_t need = pcg_exact_cells(head_count);
  if (need <= ctx->pcg_exact_capacity && ctx->pcg_exact) return 0;
  void* grown = nullptr;
  HIP_TRY(hipMalloc(&grown, sizeof(ExactCell) * (need + need / 8)));
  hipFree(ctx->pcg_exact);   // (pcg_stage_ctl is an allocation of its own, 64 bytes, and stays: ADVICE r3 -- it was freed here and used afterwards)
  ctx->pcg_exact = grown;
  ctx->pcg_exact_capacity = need + need / 8;
  return 0;
}
int bahip_pcg_iteration(bahip_context* ctx, const bahip_pcg_options* opt, const bahip_surfels* surfels,
                        bahip_camera* out_color_camera, bahip_camera* out_depth_camera, float* out_a, int* inner_steps_out,
                        int* num_converged_out) {
  REQUIRE_NO_KF_SHARDING("bahip_pcg_iteration");
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  const bool sharded = is_sharded(ctx);   // (exact sums need no rank count: every rank adds its terms, the limbs are summed)
  ctx->pcg_stage_head = 0xffffffffu;      // the accumulators are re-used: a stage-by-stage caller has to call bahip_pcg_begin again
  const int K = ctx->num_kfs;
  REQUIRE(K >= 1, "PCG needs at least one keyframe");
  const uint32_t N = surfels->surfels_size;
  const int S = ctx->in.cf_width * ctx->in.cf_height;
  PcgLayout L{};
  L.use_depth = opt->use_depth_residuals; L.use_desc = opt->use_descriptor_residuals;
  L.optimize_poses = opt->optimize_poses; L.optimize_geometry = opt->optimize_geometry;
  L.optimize_depth_intrinsics = opt->optimize_depth_intrinsics; L.optimize_color_intrinsics = opt->optimize_color_intrinsics;
  L.geom_stride = L.use_desc ? 3 : 1;
  L.gauge = (opt->gauge_keyframe >= 0 && opt->gauge_keyframe < K) ? opt->gauge_keyframe : 0;
  uint32_t cur = 0;
  const uint32_t kInvalid = 0xffffffffu;
  if (L.optimize_poses) cur += 6u * (uint32_t)(K - 1);
  L.surfel_start = kInvalid;
  if (L.optimize_geometry) { L.surfel_start = cur; cur += (uint32_t)L.geom_stride * N; }
  L.depth_intr_start = kInvalid; L.a_index = kInvalid;
  if (L.optimize_depth_intrinsics) { L.depth_intr_start = cur; cur += 5u + (uint32_t)S; L.a_index = L.depth_intr_start + 4; }
  L.color_intr_start = kInvalid;
  if (L.optimize_color_intrinsics) { L.color_intr_start = cur; cur += 4; }
  L.unknown_count = cur;
  L.head_lo = L.optimize_geometry ? L.surfel_start : cur;
  L.head_hi = L.optimize_geometry ? L.surfel_start + (uint32_t)L.geom_stride * N : cur;
  L.single_keyframe = -1; L.single_pose_index = kInvalid; L.accumulate = 0;
  const size_t U = cur;
  const uint32_t head_count = L.head_lo + (L.unknown_count - L.head_hi);
  *out_color_camera = ctx->color_cam; *out_depth_camera = ctx->depth_cam; *out_a = ctx->dp.a;
  if (inner_steps_out) *inner_steps_out = 0;
  if (num_converged_out) *num_converged_out = 0;

  if (U == 0 && !sharded) {   // nothing to solve for (e.g. one keyframe = the gauge, no surfels): every pose counts as converged
    if (num_converged_out) *num_converged_out = K;
    return 0;
  }
  if (U > ctx->pcg_capacity || ctx->pcg_buf == nullptr) {   // lazy (re-)allocation like B/direct_ba_pcg.cc:255-268
    const size_t cap = (U + U / 8 + 4096 + 3) & ~(size_t)3;   // multiple of 4: the scalar block behind the vectors stays 16-byte aligned
    float* grown = nullptr;
    HIP_TRY(hipMalloc(&grown, sizeof(float) * (5 * cap + 16)));
    hipFree(ctx->pcg_buf);
    ctx->pcg_buf = grown;
    ctx->pcg_capacity = cap;
  }
  if (ensure_pcg_exact(ctx, head_count)) return 1;
  const PcgExact ex = pcg_exact_view(ctx->pcg_exact, head_count);
  const size_t cap = ctx->pcg_capacity;
  float* r_ = ctx->pcg_buf; float* M_ = r_ + cap; float* delta = M_ + cap; float* g_ = delta + cap; float* p_ = g_ + cap;
  float* sc = p_ + cap;   // [0] alpha_n / beta_n (swapped), [1] alpha_d, [2] beta_n / alpha_n
  int i_an = 0, i_bn = 2;
  const SurfelsView sv = make_view(surfels);
  hipStream_t st = ctx->stream;
  // a rank with an empty shard launches no sweep, so it must provide zeros for the entries a sweep would have written
  HIP_TRY(hipMemsetAsync(sc, 0, sizeof(float) * 16, st));
  HIP_TRY(hipMemsetAsync(ctx->pcg_exact, 0, sizeof(ExactCell) * pcg_exact_cells(head_count), st));
  // what a sharded run exchanges: the limbs, as int64 -- an exact sum, so sharded == unsharded bit for bit
  const size_t x1_init = ((size_t)kHotExchanged1 * kHotReplicas + 2 * (size_t)head_count) * kExactLimbs;
  const size_t x1_step = ((size_t)kHotExchanged1 * kHotReplicas + (size_t)head_count) * kExactLimbs;
  const size_t x2 = ((size_t)kHotReplicas + 1) * kExactLimbs;   // the sticky flag's cell + slot 20, from ex.invalid on
  void* const x2_from = ex.invalid;
  // heavy work first (wave_cull.h): the init sweep takes the census when there is no schedule for this grid yet (a PCG-only
  // caller never runs the pose sweep that usually provides it), the inner steps use it
  const uint32_t padded_tiles = pose_padded_tiles(sv.size);
  const bool census = g_tile_order_enabled && sv.size > 0 && ctx->tile_order_tiles != padded_tiles && !ctx->tile_order_unavailable_for(padded_tiles);
  if (census && ensure_tile_schedule(ctx, padded_tiles)) return 1;
  launch_pcg_init(st, L, ex, ctx->in, ctx->dev_kfs, K, sv, r_, M_, census ? ctx->dev_tile_cost : nullptr, tile_order_for(ctx, sv.size));
  CHECK_LAUNCH();
  if (census) {
    if (launch_tile_order(st, ctx->dev_tile_cost, padded_tiles, ctx->dev_tile_order)) {
      ctx->tile_order_tiles = padded_tiles;
      ctx->phases_since_schedule = 0;
      CHECK_LAUNCH();
    } else {
      ctx->tile_order_unavailable_tiles = padded_tiles;
      HIP_TRY(hipMemsetAsync(ctx->dev_tile_cost, 0, sizeof(uint32_t) * padded_tiles, st));
    }
  }
  const uint32_t* sched = tile_order_for(ctx, sv.size);
  if (sharded && reduce_over_ranks(ctx, ex.hot, x1_init, BAHIP_SUM_I64)) return 1;
  launch_pcg_resolve_init(st, L, ex, r_, M_);
  CHECK_LAUNCH();
  launch_pcg_init2(st, L, ex, ctx->dp.a, r_, M_, delta, g_, p_);
  CHECK_LAUNCH();
  if (sharded && reduce_over_ranks(ctx, x2_from, x2, BAHIP_SUM_I64)) return 1;

  // Inner loop: the stopping rule runs on the device (pcg_control_kernel), so steps are queued in groups without a host
  // round trip per step; kernels queued after the stop return at once.  The host only looks at `stop` between groups.
  void* ctl = sc + 8;   // PcgControl lives in the scalar block (16 floats)
  if (pcg_control_bytes() > sizeof(float) * 8) return fail("PcgControl does not fit behind the scalars", __FILE__, __LINE__);
  launch_pcg_control_init(st, ex, ctl, sc + i_an);
  CHECK_LAUNCH();
  // AddAlphaDEpsilonTerms runs once per keyframe in the reference (B/kernel_pcg.cu:1102-1112), and not at all without surfels
  const double eps_repeat = (N > 0 || sharded) ? (double)K : 0.0;
  constexpr int kStepsPerGroup = 6;
  int steps = 0;
  for (int step = 0; step < opt->max_inner_iterations; ++step) {
    if (step > 0) { const int t = i_an; i_an = i_bn; i_bn = t; }
    timer_begin(ctx, 5, step == 0);
    launch_pcg_step1(st, L, ex, ctx->in, ctx->dev_kfs, K, sv, p_, g_, ctl, sched, ctx->dev_tile_counters, &ctx->pose_parity);
    timer_end(ctx, 5);
    CHECK_LAUNCH();
    if (sharded && reduce_over_ranks(ctx, ex.hot, x1_step, BAHIP_SUM_I64)) return 1;   // g head, intrinsics entries, alpha_d terms
    launch_pcg_resolve_step1(st, L, ex, g_, sc + 1, eps_repeat, ctl);
    CHECK_LAUNCH();
    launch_pcg_step2(st, L, ex, r_, M_, delta, g_, p_, sc + i_an, sc + 1, ctl);
    CHECK_LAUNCH();
    if (sharded && reduce_over_ranks(ctx, x2_from, x2, BAHIP_SUM_I64)) return 1;
    launch_pcg_control(st, ex, ctl, sc + i_bn);
    CHECK_LAUNCH();
    if (step < opt->max_inner_iterations - 1) {
      launch_pcg_step3(st, L, ex, g_, p_, sc + i_an, sc + i_bn, ctl);
      CHECK_LAUNCH();
    }
    if ((step + 1) % kStepsPerGroup == 0 || step == opt->max_inner_iterations - 1) {
      HIP_TRY(hipMemcpyAsync(ctx->pinned_i, ctl, 24, hipMemcpyDeviceToHost, st));
      HIP_TRY(hipMemcpyAsync(ctx->pinned_i + 8, ex.invalid, sizeof(unsigned), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      steps = ctx->pinned_i[4];
      // the sticky flag has been through exchange 2 of this step: every rank reads the same value here and fails alike
      if (ctx->pinned_i[8])
        return fail("PCG scheme: a non-finite term was added to the exact sums (on this rank or on another one); the surfels or images hold "
                    "non-finite values", __FILE__, __LINE__);
      if (ctx->pinned_i[3]) break;   // stop
    }
  }
  if (inner_steps_out) *inner_steps_out = steps;

  // ---- apply the update (B/direct_ba_pcg.cc:551-642) ----
  int num_converged = 0;
  if (L.optimize_poses) {
    std::vector<float> d(6 * (size_t)(K > 1 ? K - 1 : 1), 0.f);
    if (K > 1) HIP_TRY(hipMemcpy(d.data(), delta, sizeof(float) * 6 * (K - 1), hipMemcpyDeviceToHost));
    for (int k = 0; k < K; ++k) {
      if (k == L.gauge) { ++num_converged; continue; }
      const float* dk = &d[6 * (size_t)(k < L.gauge ? k : k - 1)];
      float upd[7], next[7], lg[6];
      se3_exp(dk, upd);
      se3_mul(ctx->host_kfs[k].global_T_frame, upd, next);
      fill_pose(&ctx->host_kfs[k], next);
      se3_log(upd, lg);
      float sq = 0.f;
      for (int c = 0; c < 3; ++c) sq += lg[c] * lg[c];
      for (int c = 3; c < 6; ++c) { const float v = lg[c] * 10.f; sq += v * v; }
      if (sq < 1e-06f) ++num_converged;
    }
    // (re-uploads the table "as bound" apart from the poses: nothing in the PCG scheme reads the activation field)
    HIP_TRY(hipMemcpyAsync(ctx->dev_kfs, ctx->host_kfs.data(), sizeof(KfEntry) * K, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  if (num_converged_out) *num_converged_out = num_converged;
  if (L.optimize_geometry) {
    ctx->lifecycle_bounds_tiles = 0;   // positions change: the tile bounds of an open lifecycle batch end here
    launch_pcg_update_surfels(st, L, sv, delta);
    CHECK_LAUNCH();
  }
  if (L.optimize_depth_intrinsics) {
    float b[5];
    HIP_TRY(hipMemcpy(b, delta + L.depth_intr_start, sizeof(b), hipMemcpyDeviceToHost));
    const double old_fx_inv = 1. / ctx->depth_cam.fx, old_fy_inv = 1. / ctx->depth_cam.fy;
    const double old_cx_pc = ctx->depth_cam.cx - 0.5, old_cy_pc = ctx->depth_cam.cy - 0.5;
    const double old_cx_inv = -old_cx_pc * old_fx_inv, old_cy_inv = -old_cy_pc * old_fy_inv;
    const double new_fx = 1. / (old_fx_inv + b[0]), new_fy = 1. / (old_fy_inv + b[1]);
    out_depth_camera->fx = (float)new_fx;
    out_depth_camera->fy = (float)new_fy;
    out_depth_camera->cx = (float)(-(new_fx * (old_cx_inv + b[2])) + 0.5);
    out_depth_camera->cy = (float)(-(new_fy * (old_cy_inv + b[3])) + 0.5);
    *out_a = ctx->dp.a + b[4];
    launch_pcg_update_cfactors(st, ctx->in, L.depth_intr_start + 5, delta, ctx->dp.cfactor, ctx->dp.cfactor_pitch_bytes);
    CHECK_LAUNCH();
  }
  if (L.optimize_color_intrinsics) {
    float b[4];
    HIP_TRY(hipMemcpy(b, delta + L.color_intr_start, sizeof(b), hipMemcpyDeviceToHost));
    out_color_camera->fx = (float)(ctx->color_cam.fx + b[0]);
    out_color_camera->fy = (float)(ctx->color_cam.fy + b[1]);
    out_color_camera->cx = (float)(ctx->color_cam.cx + b[2]);
    out_color_camera->cy = (float)(ctx->color_cam.cy + b[3]);
  }
  HIP_TRY(hipStreamSynchronize(st));
  return 0;
}

// ---- the PCG scheme stage by stage (B/kernels.h:397-491) ----------------------------------------------------------------------
}  // extern "C"
namespace {
constexpr uint32_t kNoUnknown = 0xffffffffu;
PcgLayout stage_layout(const bahip_pcg_layout* in, uint32_t surfels_size) {
  PcgLayout L{};
  L.use_depth = in->use_depth_residuals; L.use_desc = in->use_descriptor_residuals;
  L.optimize_poses = in->optimize_poses; L.optimize_geometry = in->optimize_geometry;
  L.optimize_depth_intrinsics = in->optimize_depth_intrinsics; L.optimize_color_intrinsics = in->optimize_color_intrinsics;
  L.geom_stride = L.use_desc ? 3 : 1;
  L.gauge = -1;
  L.surfel_start = L.optimize_geometry ? in->surfel_unknown_start_index : kNoUnknown;
  L.depth_intr_start = L.optimize_depth_intrinsics ? in->depth_intrinsics_unknown_start_index : kNoUnknown;
  L.a_index = L.optimize_depth_intrinsics ? in->depth_intrinsics_unknown_start_index + 4 : kNoUnknown;
  L.color_intr_start = L.optimize_color_intrinsics ? in->color_intrinsics_unknown_start_index : kNoUnknown;
  L.unknown_count = in->unknown_count;
  L.head_lo = L.optimize_geometry ? L.surfel_start : L.unknown_count;
  L.head_hi = L.optimize_geometry ? L.surfel_start + (uint32_t)L.geom_stride * surfels_size : L.unknown_count;
  L.single_keyframe = -1; L.single_pose_index = kNoUnknown; L.accumulate = 0;
  return L;
}
uint32_t head_count_of(const PcgLayout& L) { return L.head_lo + (L.unknown_count - L.head_hi); }
// One keyframe as a one-entry table on the device (the slot the single-frame pose entry points use).
int stage_keyframe(bahip_context* ctx, const bahip_frame* frame, const float frame_T_global[12]) {
  KfEntry e;
  if (make_entry(ctx, *frame, 0, &e)) return 1;
  memcpy(e.pose.F, frame_T_global, 12 * sizeof(float));
  HIP_TRY(hipMemcpyAsync(ctx->dev_frame1, &e, sizeof(e), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));   // e lives on this stack frame
  return 0;
}
int stage_ready(bahip_context* ctx, const PcgLayout& L) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(ctx->pcg_exact != nullptr && ctx->pcg_stage_head == head_count_of(L), "bahip_pcg_begin was not called for this layout");
  return 0;
}
}  // namespace
extern "C" {

int bahip_pcg_begin(bahip_context* ctx, const bahip_pcg_layout* layout, uint32_t surfels_size) {
  REQUIRE_NO_KF_SHARDING("bahip_pcg_begin");
  const PcgLayout L = stage_layout(layout, surfels_size);
  const uint32_t head = head_count_of(L);
  if (ensure_pcg_exact(ctx, head)) return 1;
  // the control block the stage kernels look at: never stopped (the caller owns the inner loop)
  HIP_TRY(hipMemsetAsync(ctx->pcg_exact, 0, sizeof(ExactCell) * pcg_exact_cells(head), ctx->stream));
  if (!ctx->pcg_stage_ctl) HIP_TRY(hipMalloc(&ctx->pcg_stage_ctl, 64));
  HIP_TRY(hipMemsetAsync(ctx->pcg_stage_ctl, 0, 64, ctx->stream));
  ctx->pcg_stage_head = head;
  ctx->pcg_stage_step1_calls = 0;
  return 0;
}

int bahip_pcg_init(bahip_context* ctx, const bahip_pcg_layout* layout, const bahip_frame* frame, const float frame_T_global[12],
                   uint32_t kf_pose_unknown_index, int optimize_pose_of_keyframe, const bahip_surfels* surfels, float* pcg_r, float* pcg_M) {
  PcgLayout L = stage_layout(layout, surfels->surfels_size);
  if (stage_ready(ctx, L) || stage_keyframe(ctx, frame, frame_T_global)) return 1;
  L.single_keyframe = 0; L.single_pose_index = kf_pose_unknown_index; L.accumulate = 1;
  L.optimize_poses = L.optimize_poses && optimize_pose_of_keyframe;
  launch_pcg_init(ctx->stream, L, pcg_exact_view(ctx->pcg_exact, ctx->pcg_stage_head), ctx->in, ctx->dev_frame1, 1, make_view(surfels), pcg_r, pcg_M);
  CHECK_LAUNCH();
  return 0;
}

int bahip_pcg_init2(bahip_context* ctx, const bahip_pcg_layout* layout, uint32_t surfels_size, float a, float* pcg_r, float* pcg_M,
                    float* pcg_delta, float* pcg_g, float* pcg_p, float* pcg_alpha_n) {
  const PcgLayout L = stage_layout(layout, surfels_size);
  if (stage_ready(ctx, L)) return 1;
  const PcgExact ex = pcg_exact_view(ctx->pcg_exact, ctx->pcg_stage_head);
  launch_pcg_resolve_init(ctx->stream, L, ex, pcg_r, pcg_M);
  launch_pcg_init2(ctx->stream, L, ex, a, pcg_r, pcg_M, pcg_delta, pcg_g, pcg_p);
  launch_pcg_control_init(ctx->stream, ex, ctx->pcg_stage_ctl, pcg_alpha_n);
  CHECK_LAUNCH();
  ctx->pcg_stage_step1_calls = 0;
  return 0;
}

int bahip_pcg_step1(bahip_context* ctx, const bahip_pcg_layout* layout, const bahip_frame* frame, const float frame_T_global[12],
                    uint32_t kf_pose_unknown_index, int optimize_pose_of_keyframe, const bahip_surfels* surfels, const float* pcg_p,
                    float* pcg_g) {
  PcgLayout L = stage_layout(layout, surfels->surfels_size);
  if (stage_ready(ctx, L) || stage_keyframe(ctx, frame, frame_T_global)) return 1;
  L.single_keyframe = 0; L.single_pose_index = kf_pose_unknown_index; L.accumulate = 1;
  L.optimize_poses = L.optimize_poses && optimize_pose_of_keyframe;
  launch_pcg_step1(ctx->stream, L, pcg_exact_view(ctx->pcg_exact, ctx->pcg_stage_head), ctx->in, ctx->dev_frame1, 1, make_view(surfels), pcg_p, pcg_g,
                   ctx->pcg_stage_ctl);
  CHECK_LAUNCH();
  if (surfels->surfels_size > 0) ctx->pcg_stage_step1_calls += 1;   // AddAlphaDEpsilonTerms runs in every PCGStep1CUDA call with surfels
  return 0;
}

int bahip_pcg_step2(bahip_context* ctx, const bahip_pcg_layout* layout, uint32_t surfels_size, float* pcg_r, const float* pcg_M,
                    float* pcg_delta, float* pcg_g, const float* pcg_p, const float* pcg_alpha_n, float* pcg_alpha_d, float* pcg_beta_n) {
  const PcgLayout L = stage_layout(layout, surfels_size);
  if (stage_ready(ctx, L)) return 1;
  const PcgExact ex = pcg_exact_view(ctx->pcg_exact, ctx->pcg_stage_head);
  // the epsilon terms of alpha_d from the p this step works with (bahip_pcg_iteration folds them into the kernels that
  // produce p; here p is the caller's): whatever an earlier stage left in those two slots is dropped first
  HIP_TRY(hipMemsetAsync(ex.hot + (size_t)kHotEpsLocal * kHotReplicas, 0, sizeof(ExactCell) * kHotReplicas, ctx->stream));
  HIP_TRY(hipMemsetAsync(ex.hot_tail + (size_t)(kHotEpsHead - kHotExchanged1) * kHotReplicas, 0, sizeof(ExactCell) * kHotReplicas, ctx->stream));
  launch_pcg_eps_terms(ctx->stream, L, ex, pcg_p);
  launch_pcg_resolve_step1(ctx->stream, L, ex, pcg_g, pcg_alpha_d, (double)ctx->pcg_stage_step1_calls, ctx->pcg_stage_ctl);
  launch_pcg_step2(ctx->stream, L, ex, pcg_r, pcg_M, pcg_delta, pcg_g, pcg_p, pcg_alpha_n, pcg_alpha_d, ctx->pcg_stage_ctl);
  launch_pcg_control(ctx->stream, ex, ctx->pcg_stage_ctl, pcg_beta_n);
  // the stage API never stops on its own: clear what the control kernel decided
  HIP_TRY(hipMemsetAsync(ctx->pcg_stage_ctl, 0, 64, ctx->stream));
  CHECK_LAUNCH();
  ctx->pcg_stage_step1_calls = 0;
  return 0;
}

int bahip_pcg_step3(bahip_context* ctx, const bahip_pcg_layout* layout, uint32_t surfels_size, const float* pcg_g, float* pcg_p,
                    const float* pcg_alpha_n, const float* pcg_beta_n) {
  const PcgLayout L = stage_layout(layout, surfels_size);
  if (stage_ready(ctx, L)) return 1;
  launch_pcg_step3(ctx->stream, L, pcg_exact_view(ctx->pcg_exact, ctx->pcg_stage_head), pcg_g, pcg_p, pcg_alpha_n, pcg_beta_n, ctx->pcg_stage_ctl);
  CHECK_LAUNCH();
  return 0;
}

int bahip_update_surfels_from_pcg_delta(bahip_context* ctx, const bahip_surfels* surfels, int use_descriptor_residuals,
                                        uint32_t surfel_unknown_start_index, const float* pcg_delta) {
  ctx->lifecycle_bounds_tiles = 0;   // positions change: the tile bounds of an open lifecycle batch end here
  PcgLayout L{};
  L.surfel_start = surfel_unknown_start_index;
  L.geom_stride = use_descriptor_residuals ? 3 : 1;
  launch_pcg_update_surfels(ctx->stream, L, make_view(surfels), pcg_delta);
  CHECK_LAUNCH();
  return 0;
}

int bahip_update_cfactors_from_pcg_delta(bahip_context* ctx, uint32_t cfactor_unknown_start_index, const float* pcg_delta) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  launch_pcg_update_cfactors(ctx->stream, ctx->in, cfactor_unknown_start_index, pcg_delta, ctx->dp.cfactor, ctx->dp.cfactor_pitch_bytes);
  CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
