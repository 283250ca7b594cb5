// capi_debug.hip -- test hooks and experiment switches of the C boundary (bahip_debug_*).
#include "capi_internal.h"

using namespace bahip;
using namespace bahip_capi;

extern "C" {
// ---- test hook ------------------------------------------------------------------------------------------------------
int bahip_debug_evaluate_pairs(bahip_context* ctx, const bahip_frame* frame, const float frame_T_global[12],
                               const bahip_surfels* surfels, const uint32_t* surfel_indices_host, int count, float* out_host) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  if (count <= 0) return 0;
  KfEntry e;
  if (make_entry(ctx, *frame, 0, &e)) return 1;
  memcpy(e.pose.F, frame_T_global, 12 * sizeof(float));
  DevMem idx, out;
  HIP_TRY(hipMalloc(&idx.p, sizeof(uint32_t) * count));
  HIP_TRY(hipMalloc(&out.p, sizeof(float) * 40 * count));
  HIP_TRY(hipMemcpy(idx.p, surfel_indices_host, sizeof(uint32_t) * count, hipMemcpyHostToDevice));
  launch_evaluate_pairs(ctx->stream, ctx->in, e, make_view(surfels), idx.as<uint32_t>(), count, out.as<float>());
  CHECK_LAUNCH();
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemcpy(out_host, out.p, sizeof(float) * 40 * count, hipMemcpyDeviceToHost));
  return 0;
}

int bahip_debug_exact_sum(bahip_context* ctx, const float* values_host, size_t count, int mode, double* out_host) {
  REQUIRE(out_host != nullptr && (values_host != nullptr || count == 0) && (mode == 0 || mode == 1), "bahip_debug_exact_sum: bad arguments");
  DevMem values, cells, out;
  HIP_TRY(hipMalloc(&values.p, sizeof(float) * (count ? count : 1)));
  HIP_TRY(hipMalloc(&cells.p, sizeof(ExactCell) * pcg_exact_cells(0)));
  HIP_TRY(hipMalloc(&out.p, sizeof(double)));
  if (count) HIP_TRY(hipMemcpy(values.p, values_host, sizeof(float) * count, hipMemcpyHostToDevice));
  HIP_TRY(hipMemsetAsync(cells.p, 0, sizeof(ExactCell) * pcg_exact_cells(0), ctx->stream));
  launch_exact_sum_debug(ctx->stream, pcg_exact_view(cells.p, 0), values.as<float>(), count, mode, out.as<double>());
  CHECK_LAUNCH();
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemcpy(out_host, out.p, sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

int bahip_debug_read_pcg_vector(bahip_context* ctx, int which, size_t offset, size_t count, float* out_host) {
  REQUIRE(ctx->pcg_buf != nullptr, "no PCG iteration has run on this context");
  REQUIRE(which >= 0 && which < 5 && offset + count <= ctx->pcg_capacity, "PCG vector range out of bounds");
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemcpy(out_host, ctx->pcg_buf + (size_t)which * ctx->pcg_capacity + offset, sizeof(float) * count, hipMemcpyDeviceToHost));
  return 0;
}

int bahip_debug_set_pose_lds_items(int items) {
  if (items < 0) return fail("bahip_debug_set_pose_lds_items: items must be >= 0", __FILE__, __LINE__, hipSuccess);
  set_pose_lds_items(items);
  return 0;
}
int bahip_debug_set_pose_lds_shape(int waves, int parts_shift) {
  if (waves < 0 || waves > 16 || parts_shift < -1 || parts_shift > 3) return fail("bahip_debug_set_pose_lds_shape: waves 0 .. 16, parts_shift -1 .. 3", __FILE__, __LINE__, hipSuccess);
  set_pose_lds_waves(waves);
  set_pose_lds_parts_shift(parts_shift);
  return 0;
}
int bahip_debug_set_intrinsics_reduce_form(int form) {
  if (form < -1 || form > 1) return fail("bahip_debug_set_intrinsics_reduce_form: 0, 1 or -1 (the default)", __FILE__, __LINE__, hipSuccess);
  set_intrinsics_reduce_form(form);
  return 0;
}

int bahip_debug_set_fused_iteration_begin(int enabled) {
  g_fused_iteration_begin = enabled ? 1 : 0;
  return 0;
}

int bahip_debug_set_pose_rounds_ahead(int rounds) {
  if (rounds < 0 || rounds > BAHIP_MAX_POSE_ITERATIONS) return fail("bahip_debug_set_pose_rounds_ahead: 0 .. BAHIP_MAX_POSE_ITERATIONS", __FILE__, __LINE__, hipSuccess);
  g_pose_rounds_ahead = rounds;
  return 0;
}
int bahip_debug_pose_form_launches(long long* global_form, long long* lds_form, int reset) {
  long long n[2];
  pose_form_launches(n, reset != 0);
  if (global_form) *global_form = n[0];
  if (lds_form) *lds_form = n[1];
  return 0;
}
int bahip_debug_pcg_step1_form_launches(long long* tile_form, long long* lds_form) {
  long long n[2];
  pcg_step1_form_launches(n);
  if (tile_form) *tile_form = n[0];
  if (lds_form) *lds_form = n[1];
  return 0;
}
int bahip_debug_pose_kernel_dispatches(long long* dispatches_out) {
  if (dispatches_out) *dispatches_out = pose_kernel_dispatches();
  return 0;
}
int bahip_debug_read_tile_schedule(bahip_context* ctx, uint32_t* padded_tiles_out, uint32_t* words_out, size_t max_words) {
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *padded_tiles_out = ctx->tile_order_tiles;
  if (ctx->tile_order_tiles == 0 || words_out == nullptr) return 0;
  const size_t words = std::min(max_words, tile_schedule_words(ctx->tile_order_tiles));
  HIP_TRY(hipMemcpy(words_out, ctx->dev_tile_order, sizeof(uint32_t) * words, hipMemcpyDeviceToHost));
  return 0;
}

int bahip_debug_set_tile_order(int enabled) {
  g_tile_order_enabled = enabled ? 1 : 0;
  return 0;
}

int bahip_debug_set_append_groups(int groups) {
  REQUIRE(groups >= 0, "bahip_debug_set_append_groups: groups must be >= 0");
  set_append_groups_limit(groups);
  return 0;
}

int bahip_debug_set_pose_form(int form) {
  REQUIRE(form == 0 || form == 1 || form == 2, "pose form must be 0 (automatic), 1 (one tile per wavefront, global atomics) or 2 (persistent, LDS table)");
  set_pose_form(form);
  return 0;
}

int bahip_debug_set_launch_shapes(int tile_waves, int pose_parts) {
  REQUIRE(tile_waves == 0 || tile_waves == 1 || tile_waves == 4 || tile_waves == 5, "tile_waves must be 0 (automatic), 1, 4 or 5 (the geometry step's hybrid shape)");
  REQUIRE(pose_parts == 0 || pose_parts == 1 || pose_parts == 2 || pose_parts == 4 || pose_parts == 8, "pose_parts must be 0, 1, 2, 4 or 8");
  set_tile_waves(tile_waves);
  set_pose_parts(pose_parts);
  return 0;
}

int bahip_debug_geometry_hybrid_launches(long long* launches_out) { if (launches_out) *launches_out = geometry_hybrid_launches(); return 0; }

int bahip_debug_jacobian(bahip_context* ctx, int kind, const float* in, int n_in, float* out, int n_out) {
  REQUIRE(kind >= 0 && kind <= 4 && n_in > 0 && n_in <= 16 && n_out > 0 && n_out <= 8, "bahip_debug_jacobian: bad arguments");
  DevMem d_in, d_out;
  HIP_TRY(hipMalloc(&d_in.p, 16 * sizeof(float)));
  HIP_TRY(hipMalloc(&d_out.p, 8 * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(d_in.p, in, n_in * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  launch_jacobian_debug(ctx->stream, kind, d_in.as<float>(), d_out.as<float>());
  CHECK_LAUNCH();
  HIP_TRY(hipMemcpyAsync(out, d_out.p, n_out * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}

int bahip_debug_read_pattern(bahip_context* ctx, size_t bytes, int pattern, int repeats) {
  REQUIRE(bytes >= 4096 && (pattern == 0 || pattern == 1) && repeats >= 1, "bahip_debug_read_pattern: bad arguments");
  uint32_t* buf = nullptr;
  HIP_TRY(hipMalloc(&buf, bytes + 4));
  hipError_t e = hipMemsetAsync(buf, 0, bytes + 4, ctx->stream);
  for (int r = 0; r < repeats && e == hipSuccess; ++r) {
    launch_read_pattern(ctx->stream, buf, bytes / 4, pattern, buf + bytes / 4);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(buf);
  if (e != hipSuccess) return fail("bahip_debug_read_pattern", __FILE__, __LINE__, e);
  return 0;
}

int bahip_debug_exact_math(bahip_context* ctx, int kind, const float* in, float* out, size_t n) {
  REQUIRE(kind >= 0 && kind <= 5, "bahip_debug_exact_math: kind must be 0 (reciprocal), 1 (square root), 2 (sin), 3 (cos), 4 (atan) or 5 (exp)");
  if (n == 0) return 0;
  float *d_in = nullptr, *d_out = nullptr;
  HIP_TRY(hipMalloc(&d_in, n * sizeof(float)));
  if (hipMalloc(&d_out, n * sizeof(float)) != hipSuccess) { hipFree(d_in); return fail("hipMalloc failed", __FILE__, __LINE__); }
  hipError_t e = hipMemcpyAsync(d_in, in, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) { launch_exact_math_debug(ctx->stream, kind, d_in, d_out, n); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d_in); hipFree(d_out);
  if (e != hipSuccess) return fail("bahip_debug_exact_math", __FILE__, __LINE__, e);
  return 0;
}

int bahip_debug_pose_limbs(bahip_context* ctx, const float* values_host, size_t count, long long* out_host) {
  if (count == 0) return 0;
  DevMem in, out;
  HIP_TRY(hipMalloc(&in.p, sizeof(float) * count));
  HIP_TRY(hipMalloc(&out.p, sizeof(long long) * 3 * count));
  HIP_TRY(hipMemcpy(in.p, values_host, sizeof(float) * count, hipMemcpyHostToDevice));
  launch_pose_limbs_debug(ctx->stream, in.as<float>(), out.as<long long>(), count);
  CHECK_LAUNCH();
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemcpy(out_host, out.p, sizeof(long long) * 3 * count, hipMemcpyDeviceToHost));
  return 0;
}

int bahip_debug_pose_step(bahip_context* ctx, const float* H21_b6, const float* global_T_frame, float* out_25) {
  float *d_in = nullptr, *d_out = nullptr;
  HIP_TRY(hipMalloc(&d_in, 34 * sizeof(float)));
  if (hipMalloc(&d_out, 25 * sizeof(float)) != hipSuccess) { hipFree(d_in); return fail("hipMalloc failed", __FILE__, __LINE__); }
  float in[34];
  memcpy(in, H21_b6, 27 * sizeof(float));
  memcpy(in + 27, global_T_frame, 7 * sizeof(float));
  hipError_t e = hipMemcpyAsync(d_in, in, sizeof(in), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) { launch_pose_step_debug(ctx->stream, d_in, d_out); e = hipGetLastError(); }
  if (e == hipSuccess) e = hipMemcpyAsync(out_25, d_out, 25 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  hipFree(d_in); hipFree(d_out);
  if (e != hipSuccess) return fail("bahip_debug_pose_step", __FILE__, __LINE__, e);
  return 0;
}

int bahip_debug_wave_reduce(bahip_context* ctx, const float* in_64x28, float* out_80) {
  DevMem d_in, d_out;
  HIP_TRY(hipMalloc(&d_in.p, 64 * 28 * sizeof(float)));
  HIP_TRY(hipMalloc(&d_out.p, 80 * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(d_in.p, in_64x28, 64 * 28 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemsetAsync(d_out.p, 0xff, 80 * sizeof(float), ctx->stream));
  launch_wave_reduce_debug(ctx->stream, d_in.as<float>(), d_out.as<float>());
  CHECK_LAUNCH();
  HIP_TRY(hipMemcpyAsync(out_80, d_out.p, 80 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}

int bahip_debug_count_pairs(bahip_context* ctx, const bahip_surfels* surfels, uint64_t* counts_out) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  DevMem d;
  HIP_TRY(hipMalloc(&d.p, 4 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d.p, 0, 4 * sizeof(unsigned long long), ctx->stream));
  launch_count_pairs(ctx->stream, ctx->in, ctx->dev_kfs, ctx->num_kfs, make_view(surfels), d.as<unsigned long long>());
  CHECK_LAUNCH();
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipMemcpy(counts_out, d.p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}

int bahip_debug_set_intrinsics_bin_capacity(bahip_context* ctx, int records_per_block) {
  ctx->intr_bin_forced = records_per_block;
  return 0;
}
int bahip_debug_set_intrinsics_slices(bahip_context* ctx, int slices) {
  ctx->intr_slices_forced = slices > 0 ? std::min(slices, kIntrMaxSlices) : 0;
  return 0;
}
int bahip_debug_intrinsics_bin_stats(bahip_context* ctx, uint32_t* capacity_out, uint32_t* most_out, uint64_t* total_out) {
  uint32_t most = 0; uint64_t total = 0;
  for (size_t b = 0; b < (size_t)ctx->intr_bin_count * (size_t)std::max(ctx->intr_bin_rows, 0) && ctx->intr_bin_counts_host; ++b) {
    most = std::max(most, ctx->intr_bin_counts_host[b]); total += ctx->intr_bin_counts_host[b];
  }
  if (capacity_out) *capacity_out = ctx->intr_bin_capacity;
  if (most_out) *most_out = most;
  if (total_out) *total_out = total;
  return 0;
}

}  // extern "C"
