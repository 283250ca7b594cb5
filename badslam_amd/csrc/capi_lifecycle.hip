// capi_lifecycle.hip -- the surfel lifecycle behind the C boundary: supporting surfels and merging, creation (one keyframe or a batch on
// the device), deletion + radius update, compaction, the spatial reorder.
#include "capi_internal.h"

using namespace bahip;
using namespace bahip_capi;

// the creation chain (create_chain_kernel): on unless BAHIP_CREATION_CHAIN=0 / bahip_debug_set_creation_chain(0); how many batches took it
static int g_creation_chain_enabled = bahip_env_int("BAHIP_CREATION_CHAIN", 1);
static long long g_creation_chain_batches = 0;
// a merge batch by cell lists (merge_pairs_kernel): on unless BAHIP_MERGE_CELLS=0 / bahip_debug_set_merge_cells(0)
static int g_merge_cells_enabled = bahip_env_int("BAHIP_MERGE_CELLS", 1);
static long long g_merge_cells_batches = 0;

extern "C" {
int bahip_debug_set_creation_chain(int enabled) { g_creation_chain_enabled = enabled ? 1 : 0; return 0; }
int bahip_debug_set_merge_cells(int enabled) { g_merge_cells_enabled = enabled ? 1 : 0; return 0; }
int bahip_debug_merge_cells_batches(long long* batches_out) { if (batches_out) *batches_out = g_merge_cells_batches; return 0; }
int bahip_debug_creation_chain_batches(long long* batches_out) { if (batches_out) *batches_out = g_creation_chain_batches; return 0; }
// ---- lifecycle ---------------------------------------------------------------------------------------------
static int supporting_view(uint32_t* const* supporting, uint32_t pitch, SupportingView* v) {
  for (int b = 0; b < BAHIP_MERGE_BUFFER_COUNT; ++b) {
    if (!supporting[b]) return 1;
    v->b[b] = supporting[b];
  }
  v->pitch = pitch;
  return 0;
}

// What the per-keyframe sweeps of an open lifecycle batch may skip for the frame with this frame_T_global (ba_launch.h: LifecycleCull).
// The bounds hold for the buffer they were taken from, while it only grows; the list is found by the frame's 12 coefficients.
static LifecycleCull lifecycle_cull_for(const bahip_context* ctx, const bahip_surfels* surfels, const float* frame_T_global) {
  LifecycleCull cull;
  if (!ctx->lifecycle_bounds_tiles || ctx->lifecycle_bounds_data != surfels->data || (uint64_t)ctx->lifecycle_bounds_tiles * 64 > surfels->surfels_size)
    return cull;
  cull.spheres = ctx->dev_lifecycle_bounds;
  cull.tiles = ctx->lifecycle_bounds_tiles;
  const size_t n = ctx->lifecycle_list_counts.size();
  for (size_t f = 0; f < n; ++f) {
    if (memcmp(&ctx->lifecycle_frames[12 * f], frame_T_global, 12 * sizeof(float)) == 0) {
      cull.list = ctx->dev_lifecycle_lists + ctx->lifecycle_list_offsets[f];
      cull.list_count = ctx->lifecycle_list_counts[f];
      break;
    }
  }
  return cull;
}

static int determine_supporting_impl(bahip_context* ctx, int merge, float merge_dist_factor, const KfEntry& e,
                                     const bahip_surfels* surfels, const SupportingView& sup, uint32_t* merged_count_out) {
  // The reference clears full-resolution planes (B/kernel_supporting_surfels.cc:58-60); only the
  // sparse-cell region is ever addressed, so clearing that region is equivalent.
  // Inside a lifecycle batch that knows its frames (the merge pass of a BA iteration, the end tasks: one call per keyframe, back to
  // back) the planes belong to the backend, and a merge call leaves them empty (merge_apply_kernel): the fill launch is needed for the
  // first keyframe of the batch only.
  const bool backend_owns_planes = merge && ctx->lifecycle_bounds_tiles != 0 && !ctx->lifecycle_frames.empty();
  const bool planes_known_empty = backend_owns_planes && ctx->supporting_planes_empty == sup.b[0];
  ctx->supporting_planes_empty = nullptr;
  if (!planes_known_empty) {
    launch_supporting_fill(ctx->stream, sup, ctx->in.cf_width, ctx->in.cf_height);
    CHECK_LAUNCH();
  }
  if (merged_count_out) *merged_count_out = 0;
  if (surfels->surfels_size == 0) return 0;
  const SurfelsView s = make_view(surfels);
  const LifecycleCull cull = lifecycle_cull_for(ctx, surfels, e.pose.F);
  launch_supporting_insert(ctx->stream, ctx->in, e, s, sup, cull);
  CHECK_LAUNCH();
  if (merge) {
    const float cell = (float)ctx->in.cell;
    const float cell_merge_dist_sq = cell * cell * merge_dist_factor * merge_dist_factor;
    // per-surfel decision flags live in accum row 0 (scratch by contract, B/kernels.cuh:78-90)
    uint32_t* flags = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(surfels->data) + (size_t)kSurfelAccum0 * surfels->pitch_bytes);
    uint32_t* cell_of = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(surfels->data) + (size_t)(kSurfelAccum0 + 1) * surfels->pitch_bytes);
    if (merged_count_out) {
      HIP_TRY(hipMemsetAsync(ctx->dev_counter, 0, sizeof(int), ctx->stream));
      launch_merge(ctx->stream, ctx->in, e, s, sup, cell_merge_dist_sq, kCosNormalCompat, flags, cell_of, backend_owns_planes, reinterpret_cast<uint32_t*>(ctx->dev_counter), cull);
      CHECK_LAUNCH();
      HIP_TRY(hipMemcpyAsync(ctx->pinned_i, ctx->dev_counter, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      *merged_count_out = (uint32_t)ctx->pinned_i[0];
    } else {
      // deferred count: a batch of keyframes merges without a read-back and a stream synchronisation per keyframe; the total
      // waits in dev_counter[3] for bahip_take_merged_count
      launch_merge(ctx->stream, ctx->in, e, s, sup, cell_merge_dist_sq, kCosNormalCompat, flags, cell_of, backend_owns_planes, reinterpret_cast<uint32_t*>(ctx->dev_counter) + 3, cull);
      CHECK_LAUNCH();
    }
    if (backend_owns_planes) ctx->supporting_planes_empty = sup.b[0];
  }
  return 0;
}

int bahip_determine_supporting_surfels(bahip_context* ctx, int merge, float merge_dist_factor, const bahip_frame* frame,
                                       const float frame_T_global[12], const bahip_surfels* surfels,
                                       uint32_t* const* supporting, uint32_t supporting_pitch, uint32_t* merged_count_out) {
  REQUIRE_NO_KF_SHARDING("bahip_determine_supporting_surfels");
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  SupportingView sup;
  REQUIRE(supporting_view(supporting, supporting_pitch, &sup) == 0, "supporting-surfel planes missing");
  KfEntry e;
  if (make_entry(ctx, *frame, 0, &e)) return 1;
  memcpy(e.pose.F, frame_T_global, 12 * sizeof(float));
  return determine_supporting_impl(ctx, merge, merge_dist_factor, e, surfels, sup, merged_count_out);
}

// The merges of a batch of keyframes, pipelined: per keyframe two dependent launches -- [apply of the previous keyframe beside the insert
// of this one], decide -- instead of three (kernels_lifecycle.hip: merge_apply_insert_kernel).  The keyframes alternate between the
// caller's supporting planes and a second set the context owns; both end empty.
int bahip_merge_surfels_for_keyframes(bahip_context* ctx, float merge_dist_factor, const bahip_frame* frames, const float* frame_T_global_3x4,
                                      int num_frames, const bahip_surfels* surfels, uint32_t* const* supporting, uint32_t supporting_pitch,
                                      uint32_t* merged_count_out) {
  REQUIRE_NO_KF_SHARDING("bahip_merge_surfels_for_keyframes");
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(num_frames >= 0 && (num_frames == 0 || (frames != nullptr && frame_T_global_3x4 != nullptr)) && surfels != nullptr,
          "bahip_merge_surfels_for_keyframes: NULL argument");
  SupportingView sup[2];
  REQUIRE(supporting_view(supporting, supporting_pitch, &sup[0]) == 0, "supporting-surfel planes missing");
  if (merged_count_out) *merged_count_out = 0;
  hipStream_t st = ctx->stream;
  // By cell lists (round 6; kernels_lifecycle.hip: merge_pairs_kernel): when the open lifecycle batch knows every frame of this call and
  // the frames' BA planes can all be held at once (the association sweep reads them together), the associated (surfel, frame) pairs are
  // grouped by (frame, cell) up front and each frame costs ONE launch of one thread per pair.
  bool by_cells = false;
  if (g_merge_cells_enabled && num_frames > 0 && surfels->surfels_size > 0) {
    const bool bounds_valid = ctx->lifecycle_bounds_tiles != 0 && ctx->lifecycle_bounds_data == surfels->data &&
                              (uint64_t)ctx->lifecycle_bounds_tiles * 64 <= surfels->surfels_size && !ctx->lifecycle_list_counts.empty();
    const uint32_t bounded_tiles = ctx->lifecycle_bounds_tiles;
    const uint32_t all_tiles = (surfels->surfels_size + 63u) / 64u, tail = all_tiles > bounded_tiles ? all_tiles - bounded_tiles : 0u;
    std::vector<MergeBatchFrame> table((size_t)num_frames);
    bool known = bounds_valid;
    uint64_t positions = 0;
    uint32_t max_positions = 0;
    for (int j = 0; j < num_frames && known; ++j) {
      const float* F = frame_T_global_3x4 + 12 * (size_t)j;
      const size_t listed = ctx->lifecycle_list_counts.size();
      size_t f = 0;
      while (f < listed && memcmp(&ctx->lifecycle_frames[12 * f], F, 12 * sizeof(float)) != 0) ++f;
      // (the association sweep reads the BA planes of all frames at once: a frame handed over without them gets a packing slot of its
      // own -- for a small batch; a long one without planes takes the pipelined path, which re-packs ONE slot frame after frame)
      if (f == listed || (frames[j].planes == nullptr && num_frames > 64)) { known = false; break; }
      if (make_entry(ctx, frames[j], frames[j].planes ? 0 : (size_t)j + 1, &table[j].entry)) return 1;
      memcpy(table[j].entry.pose.F, F, 12 * sizeof(float));
      table[j].list_offset = ctx->lifecycle_list_offsets[f];
      table[j].list_count = ctx->lifecycle_list_counts[f];
      table[j].pair_offset = (uint32_t)positions;
      table[j].pad_ = 0;
      positions += (uint64_t)table[j].list_count + tail;
      max_positions = std::max(max_positions, table[j].list_count + tail);
    }
    const size_t cells = (size_t)ctx->in.cf_width * (size_t)ctx->in.cf_height;
    const size_t entries = (size_t)num_frames * cells + 1;
    if (known && positions * 64 < ((uint64_t)1 << 31) && entries < ((size_t)1 << 31)) {
      auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
      const size_t sweep = 64 * (size_t)positions;   // lanes of all sweep positions: an upper bound of the pairs
      const size_t table_bytes = align(sizeof(MergeBatchFrame) * (size_t)num_frames), entry_bytes = align(sizeof(uint32_t) * entries),
                   word_bytes = align(sizeof(uint32_t) * sweep), first_bytes = align(sizeof(uint32_t) * ((size_t)num_frames + 1)),
                   temp_bytes = align(merge_batch_scan_temp_bytes(entries));
      const size_t need = table_bytes + 2 * entry_bytes + 3 * word_bytes + 2 * word_bytes + first_bytes + temp_bytes;
      if (need > ctx->merge_batch_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        hipFree(ctx->dev_merge_batch); ctx->dev_merge_batch = nullptr; ctx->merge_batch_bytes = 0;
        HIP_TRY(hipMalloc(&ctx->dev_merge_batch, need + need / 4));
        ctx->merge_batch_bytes = need + need / 4;
      }
      char* p = static_cast<char*>(ctx->dev_merge_batch);
      MergeBatchFrame* dev_table = reinterpret_cast<MergeBatchFrame*>(p); p += table_bytes;
      uint32_t* counts = reinterpret_cast<uint32_t*>(p); p += entry_bytes;
      uint32_t* offsets = reinterpret_cast<uint32_t*>(p); p += entry_bytes;
      uint32_t* pair_cells = reinterpret_cast<uint32_t*>(p); p += word_bytes;
      uint32_t* pair_ranks = reinterpret_cast<uint32_t*>(p); p += word_bytes;
      uint32_t* members = reinterpret_cast<uint32_t*>(p); p += word_bytes;
      void* member_cell = p; p += 2 * word_bytes;
      uint32_t* frame_first = reinterpret_cast<uint32_t*>(p); p += first_bytes;
      void* scan_temp = p;
      HIP_TRY(hipMemcpyAsync(dev_table, table.data(), sizeof(MergeBatchFrame) * (size_t)num_frames, hipMemcpyHostToDevice, st));
      const SurfelsView s = make_view(surfels);
      HIP_TRY(launch_merge_batch_lists(st, ctx->in, dev_table, num_frames, max_positions, s, ctx->dev_lifecycle_lists, bounded_tiles, counts, offsets, pair_cells,
                                       pair_ranks, members, member_cell, frame_first, scan_temp, temp_bytes));
      // per-surfel words "deleted at step" in accum row 0 (scratch by contract, B/kernels.cuh:78-90): ~0 = not deleted by this batch
      uint32_t* deleted_at = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(surfels->data) + (size_t)kSurfelAccum0 * surfels->pitch_bytes);
      HIP_TRY(hipMemsetAsync(deleted_at, 0xff, sizeof(uint32_t) * (size_t)surfels->surfels_size, st));
      std::vector<uint32_t> first((size_t)num_frames + 1);
      HIP_TRY(hipMemcpyAsync(first.data(), frame_first, sizeof(uint32_t) * first.size(), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));   // `table` is pageable and goes out of scope; the grids below come from `first`
      const float cell = (float)ctx->in.cell;
      const float cell_merge_dist_sq = cell * cell * merge_dist_factor * merge_dist_factor;
      uint32_t* counter = reinterpret_cast<uint32_t*>(ctx->dev_counter) + 3;   // the deferred count of bahip_take_merged_count
      for (int j = 0; j < num_frames; ++j) {
        launch_merge_pairs(st, s, members, member_cell, first[(size_t)j], first[(size_t)j + 1], (uint32_t)j, deleted_at, cell_merge_dist_sq,
                           kCosNormalCompat);
        CHECK_LAUNCH();
      }
      launch_merge_batch_apply(st, s, deleted_at, counter);
      CHECK_LAUNCH();
      // (the planes are not used; the batch's contract is that they end empty)
      if (ctx->supporting_planes_empty != sup[0].b[0]) launch_supporting_fill(st, sup[0], ctx->in.cf_width, ctx->in.cf_height);
      ctx->supporting_planes_empty = sup[0].b[0];
      by_cells = true;
      ++g_merge_cells_batches;
    }
  }
  if (!by_cells && num_frames > 0 && surfels->surfels_size > 0) {
    // the second set of planes: same pitch, the sparse-cell region's rows
    const size_t plane_bytes = (size_t)supporting_pitch * (size_t)ctx->in.cf_height;
    if (plane_bytes > ctx->merge_planes_bytes) {
      for (int b = 0; b < BAHIP_MERGE_BUFFER_COUNT; ++b) {
        hipFree(ctx->merge_planes[b]);
        ctx->merge_planes[b] = nullptr;
      }
      ctx->merge_planes_bytes = 0;
      for (int b = 0; b < BAHIP_MERGE_BUFFER_COUNT; ++b) HIP_TRY(hipMalloc(&ctx->merge_planes[b], plane_bytes));
      ctx->merge_planes_bytes = plane_bytes;
    }
    for (int b = 0; b < BAHIP_MERGE_BUFFER_COUNT; ++b) sup[1].b[b] = ctx->merge_planes[b];
    sup[1].pitch = supporting_pitch;
    if (ctx->supporting_planes_empty != sup[0].b[0]) launch_supporting_fill(st, sup[0], ctx->in.cf_width, ctx->in.cf_height);
    launch_supporting_fill(st, sup[1], ctx->in.cf_width, ctx->in.cf_height);
    ctx->supporting_planes_empty = nullptr;
    const float cell = (float)ctx->in.cell;
    const float cell_merge_dist_sq = cell * cell * merge_dist_factor * merge_dist_factor;
    // per-surfel decision words and cells live in accum rows 0 and 1 (scratch by contract, B/kernels.cuh:78-90); the decision words
    // start cleared: an insert sweep reads them for surfels no decide sweep of this batch has visited
    uint32_t* flags = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(surfels->data) + (size_t)kSurfelAccum0 * surfels->pitch_bytes);
    uint32_t* cell_of = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(surfels->data) + (size_t)(kSurfelAccum0 + 1) * surfels->pitch_bytes);
    HIP_TRY(hipMemsetAsync(flags, 0, sizeof(uint32_t) * (size_t)surfels->surfels_size, st));
    uint32_t* counter = reinterpret_cast<uint32_t*>(ctx->dev_counter) + 3;   // the deferred count of bahip_take_merged_count
    const SurfelsView s = make_view(surfels);
    std::vector<KfEntry> entries((size_t)num_frames);
    std::vector<LifecycleCull> culls((size_t)num_frames);
    for (int j = 0; j <= num_frames; ++j) {
      if (j < num_frames) {
        // (here, not up front: frames handed over without BA planes share ONE packing slot of the context, re-packed on the stream by
        // make_entry -- behind keyframe j - 1's decide sweep, the last reader of the previous packing; an apply sweep reads no image)
        if (make_entry(ctx, frames[j], 0, &entries[j])) return 1;
        memcpy(entries[j].pose.F, frame_T_global_3x4 + 12 * (size_t)j, 12 * sizeof(float));
        culls[j] = lifecycle_cull_for(ctx, surfels, entries[j].pose.F);
      }
      // [apply of keyframe j - 1] beside [insert of keyframe j]
      launch_merge_apply_insert(st, ctx->in, j > 0 ? &entries[j - 1] : nullptr, j < num_frames ? &entries[j] : nullptr, s, flags, cell_of,
                                sup[(j + 1) & 1], sup[j & 1], counter, j > 0 ? culls[j - 1] : LifecycleCull(), j < num_frames ? culls[j] : LifecycleCull());
      CHECK_LAUNCH();
      if (j < num_frames) {
        launch_merge_decide(st, ctx->in, entries[j], s, sup[j & 1], cell_merge_dist_sq, kCosNormalCompat, flags, cell_of, culls[j]);
        CHECK_LAUNCH();
      }
    }
    ctx->supporting_planes_empty = sup[0].b[0];   // (every apply sweep left its set empty)
  }
  if (merged_count_out) return bahip_take_merged_count(ctx, merged_count_out);
  return 0;
}

int bahip_lifecycle_batch_begin(bahip_context* ctx, const bahip_surfels* surfels) {
  REQUIRE(surfels != nullptr, "bahip_lifecycle_batch_begin: NULL argument");
  ctx->lifecycle_bounds_tiles = 0;
  ctx->supporting_planes_empty = nullptr;
  ctx->lifecycle_frames.clear(); ctx->lifecycle_list_offsets.clear(); ctx->lifecycle_list_counts.clear();
  const uint32_t tiles = surfels->surfels_size / 64;   // whole tiles only: what is appended later starts in the tile behind them
  if (tiles == 0) return 0;
  if (tiles > ctx->lifecycle_bounds_capacity) {
    void* grown = nullptr;
    const size_t capacity = (size_t)tiles + tiles / 4 + 1024;
    HIP_TRY(hipMalloc(&grown, capacity * 16));   // WaveBounds: four floats
    hipFree(ctx->dev_lifecycle_bounds);
    ctx->dev_lifecycle_bounds = grown;
    ctx->lifecycle_bounds_capacity = capacity;
  }
  launch_lifecycle_bounds(ctx->stream, make_view(surfels), tiles, ctx->dev_lifecycle_bounds);
  CHECK_LAUNCH();
  ctx->lifecycle_bounds_tiles = tiles;
  ctx->lifecycle_bounds_data = surfels->data;
  return 0;
}

int bahip_lifecycle_batch_set_frames(bahip_context* ctx, const float* frame_T_global_3x4, int num_frames) {
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(num_frames >= 0 && (num_frames == 0 || frame_T_global_3x4 != nullptr), "bahip_lifecycle_batch_set_frames: NULL argument");
  ctx->lifecycle_frames.clear(); ctx->lifecycle_list_offsets.clear(); ctx->lifecycle_list_counts.clear();
  const uint32_t tiles = ctx->lifecycle_bounds_tiles;
  if (tiles == 0 || num_frames == 0) return 0;   // no batch open (or an empty cloud): the sweeps take everything
  hipStream_t st = ctx->stream;
  if ((size_t)num_frames > ctx->lifecycle_frames_capacity) {
    float* F = nullptr; uint32_t* cursors = nullptr;
    const size_t capacity = (size_t)num_frames + 64;
    if (hipMalloc(&F, capacity * 12 * sizeof(float)) != hipSuccess || hipMalloc(&cursors, 2 * capacity * sizeof(uint32_t)) != hipSuccess) {
      hipFree(F); hipFree(cursors);
      return fail("allocation of the lifecycle batch's frame table failed", __FILE__, __LINE__);
    }
    hipFree(ctx->dev_lifecycle_frames); hipFree(ctx->dev_lifecycle_cursors);
    ctx->dev_lifecycle_frames = F; ctx->dev_lifecycle_cursors = cursors;
    ctx->lifecycle_frames_capacity = capacity;
  }
  uint32_t* cursors = ctx->dev_lifecycle_cursors;
  std::vector<uint32_t> counts(num_frames), starts(num_frames);
  const size_t total = (size_t)num_frames * tiles;   // room for every tile in every frame's list: one pass, no counting pass, one host wait
  if (total > ctx->lifecycle_lists_capacity) {
    uint32_t* lists = nullptr;
    const size_t capacity = total + total / 4 + 4096;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMalloc(&lists, capacity * sizeof(uint32_t)));
    hipFree(ctx->dev_lifecycle_lists);
    ctx->dev_lifecycle_lists = lists;
    ctx->lifecycle_lists_capacity = capacity;
  }
  HIP_TRY(hipMemcpyAsync(ctx->dev_lifecycle_frames, frame_T_global_3x4, (size_t)num_frames * 12 * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(cursors, 0, (size_t)num_frames * sizeof(uint32_t), st));
  launch_lifecycle_visible_tiles(st, ctx->in, ctx->dev_lifecycle_frames, num_frames, ctx->dev_lifecycle_bounds, tiles, cursors, ctx->dev_lifecycle_lists);
  CHECK_LAUNCH();
  HIP_TRY(hipMemcpyAsync(counts.data(), cursors, (size_t)num_frames * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));   // (also: frame_T_global_3x4 may be pageable memory of the caller)
  for (int f = 0; f < num_frames; ++f) starts[f] = (uint32_t)((size_t)f * tiles);
  ctx->lifecycle_frames.assign(frame_T_global_3x4, frame_T_global_3x4 + (size_t)num_frames * 12);
  ctx->lifecycle_list_offsets = starts;
  ctx->lifecycle_list_counts = counts;
  return 0;
}

int bahip_lifecycle_batch_set_keyframes(bahip_context* ctx, const int* keyframe_indices, int num_keyframes) {
  REQUIRE(num_keyframes >= 0 && (num_keyframes == 0 || keyframe_indices != nullptr), "bahip_lifecycle_batch_set_keyframes: NULL argument");
  std::vector<float> F(12 * (size_t)num_keyframes);
  for (int j = 0; j < num_keyframes; ++j) {
    REQUIRE(keyframe_indices[j] >= 0 && keyframe_indices[j] < ctx->num_kfs, "keyframe index out of range");
    memcpy(&F[12 * (size_t)j], ctx->host_kfs[keyframe_indices[j]].pose.F, 12 * sizeof(float));
  }
  return bahip_lifecycle_batch_set_frames(ctx, F.data(), num_keyframes);
}

int bahip_lifecycle_batch_end(bahip_context* ctx) {
  ctx->lifecycle_bounds_tiles = 0;
  ctx->supporting_planes_empty = nullptr;
  return 0;
}

int bahip_take_merged_count(bahip_context* ctx, uint32_t* merged_count_out) {
  REQUIRE(merged_count_out != nullptr, "bahip_take_merged_count: NULL argument");
  HIP_TRY(hipMemcpyAsync(ctx->pinned_i, ctx->dev_counter + 3, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemsetAsync(ctx->dev_counter + 3, 0, sizeof(int), ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *merged_count_out = (uint32_t)ctx->pinned_i[0];
  return 0;
}

int bahip_create_surfels_for_keyframe(bahip_context* ctx, int keyframe_index, int filter_new_surfels, int min_observation_count,
                                      const int* covis, int n_covis, const bahip_surfels* surfels, uint32_t* const* supporting,
                                      uint32_t supporting_pitch, uint32_t* new_surfel_count_out) {
  REQUIRE_NO_KF_SHARDING("bahip_create_surfels_for_keyframe");
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(keyframe_index >= 0 && keyframe_index < ctx->num_kfs, "keyframe index out of range");
  SupportingView sup;
  REQUIRE(supporting_view(supporting, supporting_pitch, &sup) == 0, "supporting-surfel planes missing");
  const KfEntry& e = ctx->host_kfs[keyframe_index];
  *new_surfel_count_out = 0;
  if (determine_supporting_impl(ctx, 0, 0.f, e, surfels, sup, nullptr)) return 1;
  const size_t px = create_padded_count(ctx->in);   // tile-major sequence, padded to whole tiles
  if (ensure_px(ctx, px, px > surfels->capacity ? px : surfels->capacity)) return 1;
  HIP_TRY(hipMemsetAsync(ctx->dev_flags, 0, px, ctx->stream));
  launch_create_flag(ctx->stream, ctx->in, e, sup, ctx->dev_flags);
  CHECK_LAUNCH();
  if (filter_new_surfels && n_covis > 0) {
    if (n_covis > ctx->covis_capacity) {
      int* idx = nullptr; float* T = nullptr;
      const int cap = n_covis + 64;
      if (hipMalloc(&idx, sizeof(int) * cap) != hipSuccess || hipMalloc(&T, sizeof(float) * 12 * cap) != hipSuccess) {
        hipFree(idx); hipFree(T);
        return fail("allocation of the co-visibility scratch failed", __FILE__, __LINE__);
      }
      hipFree(ctx->dev_covis); hipFree(ctx->dev_covis_T);
      ctx->dev_covis = idx; ctx->dev_covis_T = T;
      ctx->covis_capacity = cap;
    }
    std::vector<float> rel(12 * (size_t)n_covis);
    for (int c = 0; c < n_covis; ++c) {
      REQUIRE(covis[c] >= 0 && covis[c] < ctx->num_kfs, "co-visibility index out of range");
      // covis_T_frame = covis.frame_T_global * keyframe.global_T_frame (B/direct_ba.cc:359-365)
      float cinv[7], prod[7];
      se3_inverse(ctx->host_kfs[covis[c]].global_T_frame, cinv);
      se3_mul(cinv, e.global_T_frame, prod);
      se3_matrix3x4(prod, &rel[12 * c]);
    }
    HIP_TRY(hipMemcpyAsync(ctx->dev_covis, covis, sizeof(int) * n_covis, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->dev_covis_T, rel.data(), sizeof(float) * 12 * n_covis, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    launch_create_filter(ctx->stream, ctx->in, e, ctx->dev_kfs, ctx->dev_covis, ctx->dev_covis_T, n_covis,
                         min_observation_count, ctx->dev_flags);
    CHECK_LAUNCH();
  } else if (filter_new_surfels) {
    // no co-visible keyframe: every candidate has exactly one observation
    if (1 < min_observation_count) HIP_TRY(hipMemsetAsync(ctx->dev_flags, 0, px, ctx->stream));
  }
  HIP_TRY(scan_flags_inclusive(ctx->stream, ctx->scan_temp, ctx->scan_temp_bytes, ctx->dev_flags, ctx->dev_indices, (int)px));
  HIP_TRY(hipMemcpyAsync(ctx->pinned_i, ctx->dev_indices + (px - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const uint32_t count = (uint32_t)ctx->pinned_i[0];
  if (count == 0) return 0;
  if ((uint64_t)surfels->surfels_size + count > surfels->capacity) {
    // soft failure in the reference: logs and returns without creating (B/kernel_create_surfels.cc:162-165); the caller asks
    // bahip_context_take_capacity_exceeded() to tell this from "no new surfels"
    g_last_error = "Maximum surfel count exceeded! Retry with a higher max_surfel_count.";
    ctx->capacity_exceeded = true;
    return 0;
  }
  launch_create_append(ctx->stream, ctx->in, e, ctx->dev_flags, ctx->dev_indices, surfels->surfels_size, make_view(surfels));
  CHECK_LAUNCH();
  *new_surfel_count_out = count;
  return 0;
}

// A batch of keyframes creating surfels, one after the other as the reference does (each sees what the ones before it appended,
// B/direct_ba_alternating.cc:389-425), but without the host in between: the cloud's size lives on the device for the duration of
// the batch, the co-visibility lists and relative poses of all keyframes go up front in one copy, and the host reads the final size
// once.  Same kernels on the same data in the same order as n calls of bahip_create_surfels_for_keyframe.
int bahip_create_surfels_for_keyframes(bahip_context* ctx, const int* keyframe_indices, int num_keyframes, int filter_new_surfels,
                                       int min_observation_count, const int* covis_offsets, const int* covis_indices,
                                       const bahip_surfels* surfels, uint32_t* const* supporting, uint32_t supporting_pitch,
                                       uint32_t* new_surfel_count_out) {
  REQUIRE_NO_KF_SHARDING("bahip_create_surfels_for_keyframes");
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  REQUIRE(keyframe_indices != nullptr && covis_offsets != nullptr && new_surfel_count_out != nullptr && num_keyframes >= 0,
          "bahip_create_surfels_for_keyframes: NULL argument");
  SupportingView sup;
  REQUIRE(supporting_view(supporting, supporting_pitch, &sup) == 0, "supporting-surfel planes missing");
  *new_surfel_count_out = 0;
  if (num_keyframes == 0) return 0;
  const int total_covis = covis_offsets[num_keyframes];
  REQUIRE(total_covis == 0 || covis_indices != nullptr, "bahip_create_surfels_for_keyframes: co-visibility indices missing");
  for (int j = 0; j < num_keyframes; ++j) {
    REQUIRE(keyframe_indices[j] >= 0 && keyframe_indices[j] < ctx->num_kfs, "keyframe index out of range");
    REQUIRE(covis_offsets[j] <= covis_offsets[j + 1], "co-visibility offsets must ascend");
  }
  const size_t px = create_padded_count(ctx->in);
  if (ensure_px(ctx, px, px > surfels->capacity ? px : surfels->capacity)) return 1;
  hipStream_t st = ctx->stream;
  if (filter_new_surfels && total_covis > 0) {
    if (total_covis > ctx->covis_capacity) {
      int* idx = nullptr; float* T = nullptr;
      const int cap = total_covis + 64;
      if (hipMalloc(&idx, sizeof(int) * cap) != hipSuccess || hipMalloc(&T, sizeof(float) * 12 * cap) != hipSuccess) {
        hipFree(idx); hipFree(T);
        return fail("allocation of the co-visibility scratch failed", __FILE__, __LINE__);
      }
      hipFree(ctx->dev_covis); hipFree(ctx->dev_covis_T);
      ctx->dev_covis = idx; ctx->dev_covis_T = T;
      ctx->covis_capacity = cap;
    }
    std::vector<float> rel(12 * (size_t)total_covis);
    for (int j = 0; j < num_keyframes; ++j) {
      const KfEntry& e = ctx->host_kfs[keyframe_indices[j]];
      for (int c = covis_offsets[j]; c < covis_offsets[j + 1]; ++c) {
        REQUIRE(covis_indices[c] >= 0 && covis_indices[c] < ctx->num_kfs, "co-visibility index out of range");
        // covis_T_frame = covis.frame_T_global * keyframe.global_T_frame (B/direct_ba.cc:359-365)
        float cinv[7], prod[7];
        se3_inverse(ctx->host_kfs[covis_indices[c]].global_T_frame, cinv);
        se3_mul(cinv, e.global_T_frame, prod);
        se3_matrix3x4(prod, &rel[12 * (size_t)c]);
      }
    }
    HIP_TRY(hipMemcpyAsync(ctx->dev_covis, covis_indices, sizeof(int) * total_covis, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(ctx->dev_covis_T, rel.data(), sizeof(float) * 12 * (size_t)total_covis, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // `rel` is pageable and goes out of scope
  }
  // The cloud's size lives on the device between the keyframes of the batch, in TWO cells: a keyframe's launches read one, its append
  // writes the other (kernels_lifecycle.hip: create_append_fused_kernel); [6] = the sticky "capacity exceeded" flag.
  uint32_t* size_cell[2] = {reinterpret_cast<uint32_t*>(ctx->dev_counter) + 4, reinterpret_cast<uint32_t*>(ctx->dev_counter) + 5};
  uint32_t* exceeded_on_device = reinterpret_cast<uint32_t*>(ctx->dev_counter) + 6;
  ctx->pinned_i[2] = (int)surfels->surfels_size; ctx->pinned_i[3] = (int)surfels->surfels_size; ctx->pinned_i[4] = 0;
  HIP_TRY(hipMemcpyAsync(size_cell[0], ctx->pinned_i + 2, 3 * sizeof(int), hipMemcpyHostToDevice, st));
  // scratch of the fused appends in the (otherwise unused) index vector: one tagged word per slice of the flag sequence
  const int groups = create_append_groups();
  REQUIRE((size_t)groups <= px, "bahip_create_surfels_for_keyframes: flag sequence shorter than the append's scratch");
  uint32_t* group_words = ctx->dev_indices;
  HIP_TRY(hipMemsetAsync(group_words, 0, sizeof(uint32_t) * (size_t)groups, st));
  // the flag kernel writes every in-image entry of the flag sequence for every keyframe; the padding of the tile-major sequence is
  // cleared once per batch
  HIP_TRY(hipMemsetAsync(ctx->dev_flags, 0, px, st));
  const uint32_t cells = (uint32_t)ctx->in.cf_width * (uint32_t)ctx->in.cf_height;   // a keyframe appends at most one surfel per sparse cell
  // The chain (round 6; kernels_lifecycle.hip: create_chain_kernel): when the open lifecycle batch knows every keyframe of this call
  // (their visible-tile lists exist), keyframes 0 .. n - 2 cost ONE launch each behind three launches for the whole batch; the last
  // keyframe goes the old way below, so that the caller's supporting planes end as a one-keyframe call leaves them.
  int chained = 0;
  if (num_keyframes >= 2 && g_creation_chain_enabled) {
    std::vector<CreateBatchItem> items((size_t)num_keyframes - 1);
    uint32_t max_list = 0, bounded_tiles = 0;
    bool known = true;
    const bool bounds_valid = ctx->lifecycle_bounds_tiles != 0 && ctx->lifecycle_bounds_data == surfels->data &&
                              (uint64_t)ctx->lifecycle_bounds_tiles * 64 <= surfels->surfels_size && !ctx->lifecycle_list_counts.empty();
    known = bounds_valid;
    bounded_tiles = ctx->lifecycle_bounds_tiles;
    for (int j = 0; j + 1 < num_keyframes && known; ++j) {
      const float* F = ctx->host_kfs[keyframe_indices[j]].pose.F;
      const size_t frames = ctx->lifecycle_list_counts.size();
      size_t f = 0;
      while (f < frames && memcmp(&ctx->lifecycle_frames[12 * f], F, 12 * sizeof(float)) != 0) ++f;
      if (f == frames) { known = false; break; }
      items[j].kf_index = keyframe_indices[j];
      items[j].list_offset = ctx->lifecycle_list_offsets[f];
      items[j].list_count = ctx->lifecycle_list_counts[f];
      items[j].covis_offset = covis_offsets[j];
      items[j].n_covis = covis_offsets[j + 1] - covis_offsets[j];
      max_list = std::max(max_list, items[j].list_count);
    }
    const size_t n = (size_t)num_keyframes - 1, occupancy_bytes = n * cells, candidates_bytes = n * px;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // one block: occupancy, candidates, the scan over them, the compact list's cells and records (at most one candidate per cell and
    // keyframe), the items, the list position of every keyframe's first candidate, the library's temporary
    const size_t columns = n * cells, scan_temp_bytes = known ? create_batch_scan_temp_bytes(n * px) : 0;
    const size_t need = align(occupancy_bytes) + align(candidates_bytes) + align(sizeof(uint32_t) * n * px) + align(sizeof(uint32_t) * columns) +
                        align(sizeof(float) * columns * kSurfelAccum0) + align(sizeof(CreateBatchItem) * n) + align(sizeof(uint32_t) * (n + 1)) + align(scan_temp_bytes);
    if (known && need <= ((size_t)16 << 30) && sizeof(float) * columns < ((size_t)1 << 32) && n * px < ((size_t)1 << 31)) {   // (the scan counts in int)
      if (need > ctx->create_batch_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        hipFree(ctx->dev_create_batch); ctx->dev_create_batch = nullptr; ctx->create_batch_bytes = 0;
        HIP_TRY(hipMalloc(&ctx->dev_create_batch, need + need / 4));
        ctx->create_batch_bytes = need + need / 4;
      }
      char* p = static_cast<char*>(ctx->dev_create_batch);
      uint8_t* occupancy = reinterpret_cast<uint8_t*>(p); p += align(occupancy_bytes);
      uint8_t* candidates = reinterpret_cast<uint8_t*>(p); p += align(candidates_bytes);
      uint32_t* scan = reinterpret_cast<uint32_t*>(p); p += align(sizeof(uint32_t) * n * px);
      uint32_t* cand_cell = reinterpret_cast<uint32_t*>(p); p += align(sizeof(uint32_t) * columns);
      SurfelsView records;
      records.data = reinterpret_cast<float*>(p); records.pitch = (uint32_t)(sizeof(float) * columns); records.active = nullptr; records.size = (uint32_t)columns;
      p += align(sizeof(float) * columns * kSurfelAccum0);
      CreateBatchItem* dev_items = reinterpret_cast<CreateBatchItem*>(p); p += align(sizeof(CreateBatchItem) * n);
      uint32_t* first_of_item = reinterpret_cast<uint32_t*>(p); p += align(sizeof(uint32_t) * (n + 1));
      void* scan_temp = p;
      HIP_TRY(hipMemcpyAsync(dev_items, items.data(), n * sizeof(CreateBatchItem), hipMemcpyHostToDevice, st));
      HIP_TRY(hipMemsetAsync(occupancy, 0, occupancy_bytes, st));
      HIP_TRY(hipMemsetAsync(candidates, 0, candidates_bytes, st));
      const SurfelsView cloud_at_begin = make_view(surfels);
      HIP_TRY(launch_create_batch_prepare(st, ctx->in, ctx->dev_kfs, dev_items, (int)n, max_list, cloud_at_begin, ctx->dev_lifecycle_lists, bounded_tiles, occupancy,
                                          candidates, filter_new_surfels != 0, ctx->dev_covis, ctx->dev_covis_T, min_observation_count, scan, scan_temp,
                                          scan_temp_bytes, cand_cell, records, first_of_item));
      std::vector<uint32_t> first(n + 1);
      HIP_TRY(hipMemcpyAsync(first.data(), first_of_item, sizeof(uint32_t) * (n + 1), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));   // `items` is pageable and goes out of scope; the chain's grids come from `first`
      bahip_surfels whole = *surfels;
      whole.surfels_size = surfels->capacity;   // (the chain addresses rows by index; sizes are read on the device)
      const SurfelsView s = make_view(&whole);
      for (int j = 0; j + 1 < num_keyframes; ++j) {
        const uint32_t tag = (uint32_t)(j % 255) + 1u;
        if (j > 0 && tag == 1u) HIP_TRY(hipMemsetAsync(group_words, 0, sizeof(uint32_t) * (size_t)groups, st));   // the tags start over
        const bool has_next = j + 2 < num_keyframes;   // (the last keyframe takes the old path: it looks at the cloud itself)
        const uint32_t appended_bound = (uint32_t)std::min<uint64_t>((uint64_t)surfels->capacity - surfels->surfels_size, (uint64_t)first[(size_t)j]);
        launch_create_chain(st, ctx->in, has_next ? &ctx->host_kfs[keyframe_indices[j + 1]] : nullptr, cand_cell, records, first[(size_t)j], first[(size_t)j + 1],
                            occupancy + (size_t)j * cells, has_next ? occupancy + (size_t)(j + 1) * cells : nullptr, s, (uint32_t)surfels->surfels_size,
                            size_cell[j & 1], size_cell[(j & 1) ^ 1], (uint32_t)surfels->capacity, exceeded_on_device, group_words, tag, appended_bound);
        CHECK_LAUNCH();
      }
      chained = num_keyframes - 1;
      ++g_creation_chain_batches;
    }
  }
  for (int j = chained; j < num_keyframes; ++j) {
    const KfEntry& e = ctx->host_kfs[keyframe_indices[j]];
    const uint32_t* size_in = size_cell[j & 1];
    // what the cloud can hold by now at most: the grid of the sweep; the size itself is read on the device
    bahip_surfels bound = *surfels;
    bound.surfels_size = (uint32_t)std::min<uint64_t>(surfels->capacity, (uint64_t)surfels->surfels_size + (uint64_t)j * cells);
    const SurfelsView s = make_view(&bound);
    ctx->supporting_planes_empty = nullptr;
    // the planes are filled once per batch: every keyframe but the last leaves them empty behind its flag pass (one thread per cell reads
    // the cell and resets it); after the last one they hold what a one-keyframe call leaves -- the lists and the claim marks
    if (j == chained) launch_supporting_fill(st, sup, ctx->in.cf_width, ctx->in.cf_height);
    launch_supporting_insert(st, ctx->in, e, s, sup, lifecycle_cull_for(ctx, surfels, e.pose.F), size_in);
    launch_create_flag(st, ctx->in, e, sup, ctx->dev_flags, j + 1 < num_keyframes);
    const int n_covis = covis_offsets[j + 1] - covis_offsets[j];
    if (filter_new_surfels && n_covis > 0) {
      launch_create_filter(st, ctx->in, e, ctx->dev_kfs, ctx->dev_covis + covis_offsets[j], ctx->dev_covis_T + 12 * (size_t)covis_offsets[j], n_covis,
                           min_observation_count, ctx->dev_flags);
    } else if (filter_new_surfels) {
      if (1 < min_observation_count) HIP_TRY(hipMemsetAsync(ctx->dev_flags, 0, px, st));   // no co-visible keyframe: one observation
    }
    bound.surfels_size = surfels->capacity;   // (the append addresses rows by index; the view's size is not looked at)
    const uint32_t tag = (uint32_t)(j % 255) + 1u;
    if (j > 0 && tag == 1u) HIP_TRY(hipMemsetAsync(group_words, 0, sizeof(uint32_t) * (size_t)groups, st));   // the tags start over
    launch_create_append_fused(st, ctx->in, e, ctx->dev_flags, make_view(&bound), size_in, size_cell[(j & 1) ^ 1], (uint32_t)surfels->capacity,
                               exceeded_on_device, group_words, tag);
    CHECK_LAUNCH();
  }
  HIP_TRY(hipMemcpyAsync(ctx->pinned_i + 2, size_cell[num_keyframes & 1], sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipMemcpyAsync(ctx->pinned_i + 3, exceeded_on_device, sizeof(int), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  *new_surfel_count_out = (uint32_t)ctx->pinned_i[2] - surfels->surfels_size;
  if (ctx->pinned_i[3]) {
    g_last_error = "Maximum surfel count exceeded! Retry with a higher max_surfel_count.";
    ctx->capacity_exceeded = true;
  }
  return 0;
}

int bahip_delete_surfels_and_update_radii(bahip_context* ctx, int min_observation_count, const bahip_surfels* surfels,
                                          uint32_t* deleted_count_out) {
  REQUIRE_NO_KF_SHARDING("bahip_delete_surfels_and_update_radii");
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics not called");
  *deleted_count_out = 0;
  if (surfels->surfels_size == 0) return 0;
  HIP_TRY(hipMemsetAsync(ctx->dev_counter, 0, sizeof(int), ctx->stream));
  launch_delete_update(ctx->stream, ctx->in, ctx->dev_kfs, ctx->num_kfs, make_view(surfels), min_observation_count,
                       reinterpret_cast<uint32_t*>(ctx->dev_counter));
  CHECK_LAUNCH();
  HIP_TRY(hipMemcpyAsync(ctx->pinned_i, ctx->dev_counter, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *deleted_count_out = (uint32_t)ctx->pinned_i[0];
  return 0;
}

int bahip_compact_surfels(bahip_context* ctx, uint32_t surfel_count, const bahip_surfels* surfels) {
  ctx->lifecycle_bounds_tiles = 0;   // positions change or surfels move: a batch's tile bounds end here
  if (surfels->surfels_size == surfel_count) return 0;
  REQUIRE(surfel_count < surfels->surfels_size, "surfel_count larger than surfels_size");
  ctx->tile_order_tiles = 0;         // surfels move to other tiles: the run order of the sweeps is rebuilt by the next pose phase
  if (ensure_px(ctx, 1, surfels->capacity)) return 1;
  char* base = reinterpret_cast<char*>(surfels->data);
  // scratch rows as in the reference: accum2 = invalid flags, accum0 = ranks, accum3 = free-spot list
  uint32_t* invalid = reinterpret_cast<uint32_t*>(base + (size_t)(kSurfelAccum0 + 2) * surfels->pitch_bytes);
  uint32_t* free_rank = reinterpret_cast<uint32_t*>(base + (size_t)(kSurfelAccum0 + 0) * surfels->pitch_bytes);
  uint32_t* free_list = reinterpret_cast<uint32_t*>(base + (size_t)(kSurfelAccum0 + 3) * surfels->pitch_bytes);
  HIP_TRY(launch_compact(ctx->stream, make_view(surfels), invalid, free_rank, free_list, surfel_count, ctx->scan_temp, ctx->scan_temp_bytes));
  return 0;
}

int bahip_sort_surfels_spatially(bahip_context* ctx, const bahip_surfels* surfels, float grid_cell_size) {
  ctx->lifecycle_bounds_tiles = 0;   // positions change or surfels move: a batch's tile bounds end here
  REQUIRE(grid_cell_size > 0.f, "grid_cell_size must be positive");
  ctx->tile_order_tiles = 0;         // every surfel changes its tile: the run order of the sweeps is rebuilt by the next pose phase
  const float inv_cell = 1.0f / grid_cell_size;
  if (surfels->surfels_size >= 2) {
    const size_t need = sort_scratch_bytes(surfels->surfels_size);
    if (need > ctx->sort_scratch_bytes) {
      HIP_TRY(hipStreamSynchronize(ctx->stream));   // (a sort still reading the old scratch)
      hipFree(ctx->dev_sort_scratch); ctx->dev_sort_scratch = nullptr; ctx->sort_scratch_bytes = 0;
      HIP_TRY(hipMalloc(&ctx->dev_sort_scratch, need + need / 4));
      ctx->sort_scratch_bytes = need + need / 4;
    }
    HIP_TRY(sort_surfels_spatially(ctx->stream, make_view(surfels), inv_cell, ctx->dev_sort_scratch, ctx->sort_scratch_bytes));
  }
  return 0;
}

// B/kernel_opt_intrinsics.cc:39-281
}  // extern "C"
