// capi.hip -- the extern "C" boundary declared in include/badslam_hip.h: context, allocation, streams, preprocessing entry points, scene
// binding and the stage timers (the other groups of entry points: capi_internal.h).
// Owns only scratch (device keyframe table, pose work items, scan temp, counters); every image
// and the surfel buffer are borrowed from the caller.
#include "capi_internal.h"

using namespace bahip;
using namespace bahip_capi;

namespace bahip_capi {

thread_local std::string g_last_error;

int fail(const char* what, const char* file, int line, hipError_t e) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s failed at %s:%d: %s", what, file, line, hipGetErrorString(e));
  else snprintf(buf, sizeof(buf), "%s (%s:%d)", what, file, line);
  g_last_error = buf;
  return 1;
}

Intrinsics make_intrinsics(const bahip_camera& cc, const bahip_camera& dc, const bahip_depth_params& dp) {
  Intrinsics in{};
  in.fx = dc.fx; in.fy = dc.fy; in.cx = dc.cx; in.cy = dc.cy;
  // B/surfel_projection.h:61-71
  in.fx_inv = 1.0f / dc.fx;
  in.fy_inv = 1.0f / dc.fy;
  const float cx_pixel_center = dc.cx - 0.5f, cy_pixel_center = dc.cy - 0.5f;
  in.cx_inv = -cx_pixel_center * in.fx_inv;
  in.cy_inv = -cy_pixel_center * in.fy_inv;
  in.width = dc.width; in.height = dc.height;
  in.cfx = cc.fx; in.cfy = cc.fy; in.ccx = cc.cx; in.ccy = cc.cy;
  in.cwidth = cc.width; in.cheight = cc.height;
  // B/surfel_projection.h:105-124
  in.d2c_fx = cc.fx / dc.fx;
  in.d2c_cx = -1 * cc.fx * dc.cx / dc.fx + cc.cx;
  in.d2c_fy = cc.fy / dc.fy;
  in.d2c_cy = -1 * cc.fy * dc.cy / dc.fy + cc.cy;
  in.a = dp.a; in.raw_to_float_depth = dp.raw_to_float_depth; in.baseline_fx = dp.baseline_fx;
  in.cell = dp.sparse_surfel_cell_size;
  in.cell_shift = -1;
  if (in.cell > 0 && (in.cell & (in.cell - 1)) == 0) { in.cell_shift = 0; while ((1 << in.cell_shift) < in.cell) ++in.cell_shift; }
  in.cfactor = dp.cfactor; in.cfactor_pitch = dp.cfactor_pitch_bytes;
  in.cf_width = dp.cfactor_width; in.cf_height = dp.cfactor_height;
  in.geom_skip = plane_strip_skip(dc.height);
  in.fp_skip = plane_strip_skip(cc.height + 2);
  in.sum_classes = 4;   // (the context's choice is written over this: bahip_set_intrinsics, bahip_context_set_sum_classes)
  in.create_tile = 8 * dp.sparse_surfel_cell_size;   // (likewise: bahip_context_set_creation_order)
  return in;
}

void fill_pose(KfEntry* e, const float* global_T_frame) {
  for (int c = 0; c < 7; ++c) e->global_T_frame[c] = global_T_frame[c];
  float inv[7];
  se3_inverse(global_T_frame, inv);
  se3_matrix3x4(inv, e->pose.F);
  se3_rotation(global_T_frame, e->pose.GR);
}

int planes_alloc(int width, int height, int cwidth, int cheight, bahip_frame_planes** out) {
  bahip_frame_planes* p = new bahip_frame_planes();
  p->width = width; p->height = height; p->cwidth = cwidth; p->cheight = cheight;
  const size_t gwords = (size_t)plane_tiles_x(width) * plane_tiles_y(height) * 32;
  const size_t fwords = (size_t)plane_tiles_x(cwidth + 2) * plane_tiles_y(cheight + 2) * 32;
  if (hipMalloc(&p->geom, gwords * sizeof(uint32_t)) != hipSuccess || hipMalloc(&p->lumafp, fwords * sizeof(uint32_t)) != hipSuccess) {
    hipFree(p->geom); hipFree(p->lumafp); delete p;
    return fail("hipMalloc of frame planes failed", __FILE__, __LINE__);
  }
  *out = p;
  return 0;
}
void planes_free(bahip_frame_planes* p) {
  if (!p) return;
  hipFree(p->geom); hipFree(p->lumafp);
  delete p;
}

KfEntry raw_entry(const bahip_frame& f) {
  KfEntry e{};
  e.depth = f.depth; e.normals = f.normals; e.radius = f.radius; e.color = f.color;
  e.depth_pitch = f.depth_pitch_bytes; e.normals_pitch = f.normals_pitch_bytes;
  e.radius_pitch = f.radius_pitch_bytes; e.color_pitch = f.color_pitch_bytes;
  return e;
}

// Frame table entry for `f`.  The sweeps read the tiled BA planes; a caller that maintains them (Keyframe does) passes
// them in f.planes, otherwise they are packed here, on the context stream, into library-owned planes (`slot`).
int make_entry(bahip_context* ctx, const bahip_frame& f, size_t slot, KfEntry* out) {
  KfEntry e = raw_entry(f);
  const bahip_frame_planes* p = f.planes;
  REQUIRE(ctx->have_intrinsics, "bahip_set_intrinsics must precede any call that takes frames");
  const int w = ctx->in.width, h = ctx->in.height, cw = ctx->in.cwidth, ch = ctx->in.cheight;
  if (!p) {
    if (ctx->auto_planes.size() <= slot) ctx->auto_planes.resize(slot + 1, nullptr);
    bahip_frame_planes*& mine = ctx->auto_planes[slot];
    if (mine && (mine->width != w || mine->height != h || mine->cwidth != cw || mine->cheight != ch)) { planes_free(mine); mine = nullptr; }
    if (!mine && planes_alloc(w, h, cw, ch, &mine)) return 1;
    launch_pack_planes(ctx->stream, e, w, h, cw, ch, mine->geom, mine->lumafp);
    CHECK_LAUNCH();
    p = mine;
  }
  REQUIRE(p->width == w && p->height == h && p->cwidth == cw && p->cheight == ch, "frame planes do not match the camera image sizes");
  e.geom = p->geom; e.lumafp = p->lumafp;
  *out = e;
  return 0;
}

SurfelsView make_view(const bahip_surfels* s) {
  SurfelsView v;
  v.data = s->data; v.pitch = s->pitch_bytes; v.active = s->active; v.size = s->surfels_size;
  return v;
}

// Grow-on-demand for library-owned scratch: the new block is allocated FIRST and swapped in on success, so a failed grow
// leaves pointer and capacity as they were (no dangling pointer behind an unchanged capacity, no double free at destroy).

int ensure_work(bahip_context* ctx, int n) {
  if (n <= ctx->work_capacity) return 0;
  const int cap = n + 64;
  // allocate first, swap on success: a failed grow leaves the context as it was
  PoseWork* work = nullptr; HbFixed* hb = nullptr; PoseWork* pinned = nullptr;
  const size_t records = pose_work_records((size_t)cap);
  if (hipMalloc(&work, sizeof(PoseWork) * records) != hipSuccess || hipMalloc(&hb, sizeof(HbFixed) * kHbStride * cap) != hipSuccess ||
      hipHostMalloc(&pinned, sizeof(PoseWork) * records, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
    hipFree(work); hipFree(hb); if (pinned) hipHostFree(pinned);
    return fail("allocation of the pose work items failed", __FILE__, __LINE__);
  }
  hipFree(ctx->dev_work); hipFree(ctx->dev_Hb);
  if (ctx->pinned_work) hipHostFree(ctx->pinned_work);
  ctx->dev_work = work; ctx->dev_Hb = hb; ctx->pinned_work = pinned;
  ctx->work_capacity = cap;
  return 0;
}

int ensure_px(bahip_context* ctx, size_t px, size_t scan_n) {
  if (px > ctx->px_capacity) {
    uint8_t* flags = nullptr; uint32_t* indices = nullptr;
    if (hipMalloc(&flags, px) != hipSuccess || hipMalloc(&indices, px * sizeof(uint32_t)) != hipSuccess) {
      hipFree(flags); hipFree(indices);
      return fail("allocation of the new-surfel flag / index vectors failed", __FILE__, __LINE__);
    }
    hipFree(ctx->dev_flags); hipFree(ctx->dev_indices);
    ctx->dev_flags = flags; ctx->dev_indices = indices;
    ctx->px_capacity = px;
  }
  const size_t need = scan_temp_bytes(scan_n);
  if (need > ctx->scan_temp_bytes) {
    void* temp = nullptr;
    if (hipMalloc(&temp, need) != hipSuccess) return fail("allocation of the scan scratch failed", __FILE__, __LINE__);
    hipFree(ctx->scan_temp);
    ctx->scan_temp = temp;
    ctx->scan_temp_bytes = need;
  }
  return 0;
}

// profiling == 3: like 2 (cumulative), but only stage 2 (the pose-accumulate launches) is timed

void timer_begin(bahip_context* ctx, int stage, bool first, int units) {
  if (!timer_on(ctx, stage)) return;
  StageTimer& t = ctx->timers[stage];
  if (first && ctx->profiling == 1) { t.used = 0; t.units = 0; }
  t.units += units;
  if ((int)t.ev.size() < 2 * (t.used + 1)) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    t.ev.push_back(a); t.ev.push_back(b);
    t.skip.push_back(0);
  }
  t.skip[t.used] = 0;
  hipEventRecord(t.ev[2 * t.used], ctx->stream);
}
void timer_end(bahip_context* ctx, int stage) {
  if (!timer_on(ctx, stage)) return;
  StageTimer& t = ctx->timers[stage];
  hipEventRecord(t.ev[2 * t.used + 1], ctx->stream);
  t.used += 1;
}


int ensure_tile_bounds(bahip_context* ctx, uint32_t surfels) {
  const size_t need = pose_tile_bounds_bytes(surfels);
  if (need <= ctx->tile_bounds_bytes) return 0;
  void* grown = nullptr;
  HIP_TRY(hipMalloc(&grown, need + need / 4));
  hipFree(ctx->dev_tile_bounds);
  ctx->dev_tile_bounds = grown;
  ctx->tile_bounds_bytes = need + need / 4;
  return 0;
}

int g_tile_order_enabled = bahip_env_int("BAHIP_TILE_ORDER", 1);
int ensure_tile_schedule(bahip_context* ctx, uint32_t padded_tiles) {
  if (padded_tiles <= ctx->tile_schedule_capacity) return 0;
  const size_t cap = (size_t)padded_tiles + padded_tiles / 4;
  uint32_t* cost = nullptr; uint32_t* order = nullptr;
  if (hipMalloc(&cost, sizeof(uint32_t) * cap) != hipSuccess || hipMalloc(&order, sizeof(uint32_t) * tile_schedule_words((uint32_t)cap)) != hipSuccess) {
    hipFree(cost); hipFree(order);
    return fail("allocation of the tile schedule failed", __FILE__, __LINE__);
  }
  if (hipMemsetAsync(cost, 0, sizeof(uint32_t) * cap, ctx->stream) != hipSuccess) {
    hipFree(cost); hipFree(order);
    return fail("clearing the tile census failed", __FILE__, __LINE__);
  }
  hipFree(ctx->dev_tile_cost); hipFree(ctx->dev_tile_order);
  ctx->dev_tile_cost = cost; ctx->dev_tile_order = order;
  ctx->tile_schedule_capacity = cap;
  ctx->tile_order_tiles = 0;
  return 0;
}
// The schedule for a sweep over `surfels` surfels, or NULL (none built for this grid size yet, or switched off).
const uint32_t* tile_order_for(const bahip_context* ctx, uint32_t surfels) {
  return (g_tile_order_enabled && ctx->tile_order_tiles != 0 && ctx->tile_order_tiles == pose_padded_tiles(surfels)) ? ctx->dev_tile_order : nullptr;
}


}  // namespace bahip_capi

extern "C" {

const char* bahip_last_error(void) { return g_last_error.c_str(); }

int bahip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int bahip_context_create(bahip_context** out, void* hip_stream) {
  REQUIRE(out != nullptr, "bahip_context_create: out is NULL");
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  REQUIRE(n > 0, "bahip_context_create: no HIP device (the HIP backend has no CPU fallback)");
  bahip_context* ctx = new bahip_context();
  ctx->stream = static_cast<hipStream_t>(hip_stream);
  if (const char* e = getenv("BAHIP_ARITHMETIC")) ctx->arithmetic = (strcmp(e, "fast") == 0 || strcmp(e, "1") == 0) ? BAHIP_ARITHMETIC_FAST : BAHIP_ARITHMETIC_EXACT;
  ctx->in.fast_math = ctx->arithmetic;
  if (const char* e = getenv("BAHIP_INTR_SLICES")) ctx->intr_slices_forced = std::min(std::max(atoi(e), 0), kIntrMaxSlices);   // experiments: slices of the intrinsics sweep
  const bool ok = hipMalloc(&ctx->dev_counter, 16 * sizeof(int)) == hipSuccess && hipMemset(ctx->dev_counter, 0, 16 * sizeof(int)) == hipSuccess &&
                  hipHostMalloc(&ctx->pinned_i, 16 * sizeof(int)) == hipSuccess &&
                  hipHostMalloc(&ctx->pinned_f, 128 * sizeof(float)) == hipSuccess &&
                  hipMalloc(&ctx->dev_tile_counters, 16 * sizeof(uint32_t)) == hipSuccess &&
                  hipMemset(ctx->dev_tile_counters, 0, 16 * sizeof(uint32_t)) == hipSuccess &&
                  hipMalloc(&ctx->dev_frame1, sizeof(KfEntry)) == hipSuccess &&
                  hipMalloc(&ctx->dev_work1, sizeof(PoseWork) * pose_work_records(1)) == hipSuccess &&
                  hipHostMalloc(&ctx->pinned_work1, sizeof(PoseWork) * (1 + kPoseTailRecords), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
                  hipMalloc(&ctx->dev_Hb1, sizeof(HbFixed) * kHbStride) == hipSuccess;
  if (!ok) {
    bahip_context_destroy(ctx);   // frees whatever was allocated (hipFree(nullptr) is a no-op)
    return fail("bahip_context_create: allocation of the context scratch failed", __FILE__, __LINE__);
  }
  *out = ctx;
  return 0;
}

void bahip_context_destroy(bahip_context* ctx) {
  if (!ctx) return;
  hipStreamSynchronize(ctx->stream);
#ifdef BAHIP_COUNT_CANDIDATES
  bahip::pose_counters_dump();
#endif
#ifdef BAHIP_TILE_TIMELINE
  if (const char* dir = getenv("BAHIP_TIMELINE_DIR")) {
    bahip::geometry_timeline_dump((std::string(dir) + "/geometry_timeline.bin").c_str());
    bahip::pose_timeline_dump((std::string(dir) + "/pose_timeline.bin").c_str());
  }
#endif
  if (ctx->pinned_work) hipHostFree(ctx->pinned_work);
  hipFree(ctx->dev_kfs); hipFree(ctx->dev_work); hipFree(ctx->dev_Hb);
  hipFree(ctx->dev_frame1); hipFree(ctx->dev_work1); hipFree(ctx->dev_Hb1); hipFree(ctx->dev_tile_counters);
  hipFree(ctx->dev_counter);
  if (ctx->pinned_i) hipHostFree(ctx->pinned_i);
  if (ctx->pinned_f) hipHostFree(ctx->pinned_f);
  if (ctx->pinned_work1) hipHostFree(ctx->pinned_work1);
  stage_free(&ctx->stage_kfs); stage_free(&ctx->stage_covis); stage_free(&ctx->stage_window);
  hipFree(ctx->dev_flags); hipFree(ctx->dev_indices); hipFree(ctx->scan_temp);
  hipFree(ctx->dev_merge_batch);
  hipFree(ctx->dev_sort_scratch);
  hipFree(ctx->dev_create_batch);
  hipFree(ctx->dev_covis); hipFree(ctx->dev_covis_T); hipFree(ctx->dev_covis_csr); hipFree(ctx->dev_tile_bounds); hipFree(ctx->dev_lifecycle_bounds); hipFree(ctx->dev_lifecycle_frames); hipFree(ctx->dev_lifecycle_cursors); hipFree(ctx->dev_lifecycle_lists); hipFree(ctx->dev_window);
  hipFree(ctx->intr_bin_cursors); hipFree(ctx->intr_bin_records); hipHostFree(ctx->intr_bin_counts_host);
  if (ctx->intr_aux_stream) { hipStreamDestroy(ctx->intr_aux_stream); for (hipEvent_t e : ctx->intr_events) if (e) hipEventDestroy(e); }
  hipFree(ctx->intr_scratch); hipFree(ctx->pcg_buf); hipFree(ctx->pcg_exact); hipFree(ctx->pcg_stage_ctl); hipFree(ctx->kf_partials);
  for (int b = 0; b < BAHIP_MERGE_BUFFER_COUNT; ++b) hipFree(ctx->merge_planes[b]);
  hipFree(ctx->dev_tile_cost); hipFree(ctx->dev_tile_order);
  hipFree(ctx->dev_loop_ctl);
  if (ctx->host_loop_ctl) hipHostFree(ctx->host_loop_ctl);
  rccl_destroy_communicator(ctx);
  for (bahip_frame_planes* p : ctx->auto_planes) planes_free(p);
  for (auto& t : ctx->timers) for (auto e : t.ev) hipEventDestroy(e);
  delete ctx;
}

int bahip_context_synchronize(bahip_context* ctx) {
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return 0;
}

int bahip_context_take_capacity_exceeded(bahip_context* ctx) {
  const int flag = ctx->capacity_exceeded ? 1 : 0;
  ctx->capacity_exceeded = false;
  return flag;
}

int bahip_malloc_pitch(void** ptr, size_t* pitch_bytes, size_t width_bytes, size_t height) {
  // Rows padded to 256 B (what hipMallocPitch would give), one plain allocation.
  const size_t pitch = (width_bytes + 255) & ~size_t(255);
  HIP_TRY(hipMalloc(ptr, pitch * (height ? height : 1)));
  *pitch_bytes = pitch;
  return 0;
}

int bahip_free(void* ptr) {
  HIP_TRY(hipFree(ptr));
  return 0;
}

int bahip_memcpy_2d(bahip_context* ctx, void* dst, size_t dst_pitch, const void* src, size_t src_pitch,
                    size_t width_bytes, size_t height, int kind) {
  const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height, k, ctx->stream));
  if (kind != 3) HIP_TRY(hipStreamSynchronize(ctx->stream));  // pageable host memory: keep the call self-contained
  return 0;
}

int bahip_memset_2d(bahip_context* ctx, void* dst, size_t pitch, int value, size_t width_bytes, size_t height) {
  HIP_TRY(hipMemset2DAsync(dst, pitch, value, width_bytes, height, ctx->stream));
  return 0;
}

// ---- stream-level helpers for the host-side CUDABuffer<T> (no context needed) ------------------------
int bahip_context_set_stream(bahip_context* ctx, void* hip_stream) {
  hipStream_t next = static_cast<hipStream_t>(hip_stream);
  if (next != ctx->stream) {
    // the tables of the bound scene went out on the previous stream without a host wait (UploadStage): work on the new stream is
    // ordered behind those copies
    for (UploadStage* stage : {&ctx->stage_kfs, &ctx->stage_covis, &ctx->stage_window})
      if (stage->pending && stage->done) HIP_TRY(hipStreamWaitEvent(next, stage->done, 0));
  }
  ctx->stream = next;
  return 0;
}
int bahip_stream_create(void** out) {
  hipStream_t s;
  HIP_TRY(hipStreamCreate(&s));
  *out = s;
  return 0;
}
int bahip_stream_destroy(void* s) {
  HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(s)));
  return 0;
}
int bahip_stream_synchronize(void* s) {
  HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(s)));
  return 0;
}
int bahip_memcpy_2d_async(void* stream, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes,
                          size_t height, int kind) {
  const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height, k, static_cast<hipStream_t>(stream)));
  return 0;
}
int bahip_memcpy_async(void* stream, void* dst, const void* src, size_t bytes, int kind) {
  const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, k, static_cast<hipStream_t>(stream)));
  return 0;
}
int bahip_host_alloc(void** ptr, size_t bytes) {
  REQUIRE(ptr != nullptr, "bahip_host_alloc: NULL argument");
  HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
  return 0;
}
int bahip_host_free(void* ptr) {
  if (ptr) HIP_TRY(hipHostFree(ptr));
  return 0;
}
int bahip_host_is_pinned(const void* ptr, size_t bytes) {
  if (!ptr) return 0;
  hipPointerAttribute_t first{}, last{};
  if (hipPointerGetAttributes(&first, ptr) != hipSuccess) { (void)hipGetLastError(); return 0; }   // unknown to the runtime: pageable
  if (first.type != hipMemoryTypeHost) return 0;
  if (bytes > 1) {
    if (hipPointerGetAttributes(&last, static_cast<const char*>(ptr) + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return 0; }
    if (last.type != hipMemoryTypeHost) return 0;
  }
  return 1;
}
int bahip_memset_async(void* stream, void* dst, int value, size_t bytes) {
  HIP_TRY(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(stream)));
  return 0;
}

}  // extern "C"
namespace {
template <typename T>
__global__ void fill_2d_kernel(T* data, uint32_t pitch, T value, int width, int height) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x < width && y < height) *reinterpret_cast<T*>(reinterpret_cast<char*>(data) + (size_t)y * pitch + (size_t)x * sizeof(T)) = value;
}
}  // namespace
extern "C" {

// CUDABuffer<T>::Clear(value, stream) (libvis/src/libvis/cuda/cuda_buffer.cu:41-60): every element := value.
int bahip_fill_2d(void* stream, void* data, size_t pitch_bytes, int elem_bytes, uint32_t value_bits, int width, int height) {
  if (width <= 0 || height <= 0) return 0;
  const dim3 grid((width + 63) / 64, (height + 3) / 4), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (elem_bytes == 1) hipLaunchKernelGGL(fill_2d_kernel<uint8_t>, grid, block, 0, st, (uint8_t*)data, (uint32_t)pitch_bytes, (uint8_t)value_bits, width, height);
  else if (elem_bytes == 2) hipLaunchKernelGGL(fill_2d_kernel<uint16_t>, grid, block, 0, st, (uint16_t*)data, (uint32_t)pitch_bytes, (uint16_t)value_bits, width, height);
  else if (elem_bytes == 4) hipLaunchKernelGGL(fill_2d_kernel<uint32_t>, grid, block, 0, st, (uint32_t*)data, (uint32_t)pitch_bytes, value_bits, width, height);
  else return fail("bahip_fill_2d: element size must be 1, 2 or 4 bytes", __FILE__, __LINE__);
  CHECK_LAUNCH();
  return 0;
}

// ---- preprocessing ---------------------------------------------------------------------------------
int bahip_bilateral_filtering_and_depth_cutoff(bahip_context* ctx, float sigma_xy, float sigma_value, float radius_factor,
                                               uint16_t max_depth, float raw_to_float_depth, const uint16_t* in_depth, uint32_t in_pitch,
                                               uint16_t* out_depth, uint32_t out_pitch, int width, int height) {
  REQUIRE(in_depth != out_depth, "bilateral filtering cannot run in place");
  REQUIRE(launch_bilateral_filter(ctx->stream, sigma_xy, sigma_value, radius_factor, max_depth, raw_to_float_depth, in_depth, in_pitch,
                                  out_depth, out_pitch, width, height) == 0,
          "bilateral filter radius (radius_factor * sigma_xy) must be in [0, 8] pixels");
  CHECK_LAUNCH();
  return 0;
}

int bahip_compute_brightness(bahip_context* ctx, const uint8_t* rgb, uint32_t rgb_pitch, uint8_t* rgba, uint32_t rgba_pitch,
                             int width, int height) {
  launch_brightness(ctx->stream, rgb, rgb_pitch, rgba, rgba_pitch, width, height);
  CHECK_LAUNCH();
  return 0;
}

int bahip_compute_normals(bahip_context* ctx, const bahip_camera* cam, const bahip_depth_params* dp, const uint16_t* in_depth,
                          uint32_t in_pitch, uint16_t* out_depth, uint32_t out_pitch, uint16_t* out_normals,
                          uint32_t normals_pitch) {
  const Intrinsics in = make_intrinsics(*cam, *cam, *dp);
  launch_normals_from_depth(ctx->stream, in, in_depth, in_pitch, out_depth, out_pitch, out_normals, normals_pitch);
  CHECK_LAUNCH();
  return 0;
}

int bahip_compute_point_radii_and_remove_isolated_pixels(bahip_context* ctx, const bahip_camera* cam, float raw_to_float_depth,
                                                         const uint16_t* depth, uint32_t depth_pitch, uint16_t* radius,
                                                         uint32_t radius_pitch, uint16_t* out_depth, uint32_t out_pitch) {
  bahip_depth_params dp{};
  dp.sparse_surfel_cell_size = 1;
  const Intrinsics in = make_intrinsics(*cam, *cam, dp);
  launch_point_radii(ctx->stream, in, raw_to_float_depth, depth, depth_pitch, radius, radius_pitch, out_depth, out_pitch);
  CHECK_LAUNCH();
  return 0;
}

int bahip_compute_min_max_depth(bahip_context* ctx, const uint16_t* depth, uint32_t depth_pitch, int width, int height,
                                float raw_to_float_depth, float* min_depth, float* max_depth) {
  // init: min = +inf, max = 0 (B/cuda_depth_processing.cu ComputeMinMaxDepthCUDA_InitializeBuffers)
  const float init[2] = {__builtin_huge_valf(), 0.f};
  memcpy(ctx->pinned_f, init, sizeof(init));
  HIP_TRY(hipMemcpyAsync(ctx->dev_counter + 1, ctx->pinned_f, 2 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  launch_min_max_depth(ctx->stream, depth, depth_pitch, width, height, raw_to_float_depth, ctx->dev_counter + 1);
  CHECK_LAUNCH();
  HIP_TRY(hipMemcpyAsync(ctx->pinned_f, ctx->dev_counter + 1, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  *min_depth = ctx->pinned_f[0];
  *max_depth = ctx->pinned_f[1];
  return 0;
}

// ---- scene binding -----------------------------------------------------------------------------------
int bahip_set_intrinsics(bahip_context* ctx, const bahip_camera* color_camera, const bahip_camera* depth_camera,
                         const bahip_depth_params* dp) {
  REQUIRE(dp->sparse_surfel_cell_size >= 1, "sparse_surfel_cell_size must be >= 1");
  ctx->color_cam = *color_camera; ctx->depth_cam = *depth_camera; ctx->dp = *dp;
  ctx->in = make_intrinsics(*color_camera, *depth_camera, *dp);
  ctx->in.sum_classes = ctx->sum_classes;
  ctx->in.fast_math = ctx->arithmetic;
  if (ctx->row_major_creation) ctx->in.create_tile = std::max(ctx->in.width, ctx->in.height);
  ctx->have_intrinsics = true;
  return 0;
}

int bahip_context_set_arithmetic(bahip_context* ctx, int arithmetic) {
  REQUIRE(ctx != nullptr, "bahip_context_set_arithmetic: NULL context");
  REQUIRE(arithmetic == BAHIP_ARITHMETIC_EXACT || arithmetic == BAHIP_ARITHMETIC_FAST, "bahip_context_set_arithmetic: BAHIP_ARITHMETIC_EXACT or BAHIP_ARITHMETIC_FAST");
  ctx->arithmetic = arithmetic;
  ctx->in.fast_math = arithmetic;
  return 0;
}
int bahip_context_get_arithmetic(bahip_context* ctx) { return ctx ? ctx->arithmetic : -1; }

int bahip_context_set_creation_order(bahip_context* ctx, int row_major) {
  ctx->row_major_creation = row_major != 0;
  if (ctx->have_intrinsics) ctx->in.create_tile = ctx->row_major_creation ? std::max(ctx->in.width, ctx->in.height) : 8 * ctx->in.cell;
  return 0;
}

int bahip_context_surfels_rearranged(bahip_context* ctx) {
  REQUIRE(ctx != nullptr, "bahip_context_surfels_rearranged: NULL context");
  ctx->tile_order_tiles = 0;          // the next pose phase counts the candidates per tile again and rebuilds the run order
  ctx->lifecycle_bounds_tiles = 0;    // and an open lifecycle batch's tile bounds describe the previous arrangement
  return 0;
}

int bahip_frame_planes_create(bahip_context* ctx, int depth_width, int depth_height, int color_width, int color_height,
                              bahip_frame_planes** out) {
  REQUIRE(ctx != nullptr && out != nullptr, "bahip_frame_planes_create: NULL argument");
  REQUIRE(depth_width > 0 && depth_height > 0 && color_width > 0 && color_height > 0, "image sizes must be positive");
  return planes_alloc(depth_width, depth_height, color_width, color_height, out);
}

int bahip_frame_planes_update(bahip_context* ctx, bahip_frame_planes* planes, const bahip_frame* frame) {
  REQUIRE(planes != nullptr && frame != nullptr, "bahip_frame_planes_update: NULL argument");
  REQUIRE(frame->depth && frame->normals && frame->color, "bahip_frame_planes_update needs depth, normals and colour");
  launch_pack_planes(ctx->stream, raw_entry(*frame), planes->width, planes->height, planes->cwidth, planes->cheight, planes->geom,
                     planes->lumafp);
  CHECK_LAUNCH();
  return 0;
}

void bahip_frame_planes_destroy(bahip_frame_planes* planes) { planes_free(planes); }

}  // extern "C"
int stage_upload(UploadStage* stage, void* dev_dst, const void* src, size_t bytes, hipStream_t stream, const void* src2, size_t bytes2, void* dev_dst2) {
  if (stage->pending) {   // the previous upload through this stage still owns the buffer (it has normally completed long ago)
    HIP_TRY(hipEventSynchronize(stage->done));
    stage->pending = false;
  }
  const size_t total = bytes + bytes2;
  if (total == 0) return 0;
  if (total > stage->capacity) {
    void* grown = nullptr;
    HIP_TRY(hipHostMalloc(&grown, total + total / 4 + 4096));
    if (stage->pinned) hipHostFree(stage->pinned);
    stage->pinned = grown;
    stage->capacity = total + total / 4 + 4096;
  }
  if (!stage->done) HIP_TRY(hipEventCreateWithFlags(&stage->done, hipEventDisableTiming));
  char* p = static_cast<char*>(stage->pinned);
  if (bytes) { memcpy(p, src, bytes); HIP_TRY(hipMemcpyAsync(dev_dst, p, bytes, hipMemcpyHostToDevice, stream)); }
  if (bytes2) { memcpy(p + bytes, src2, bytes2); HIP_TRY(hipMemcpyAsync(dev_dst2, p + bytes, bytes2, hipMemcpyHostToDevice, stream)); }
  HIP_TRY(hipEventRecord(stage->done, stream));
  stage->pending = true;
  return 0;
}
void stage_free(UploadStage* stage) {
  if (stage->done) { hipEventSynchronize(stage->done); hipEventDestroy(stage->done); }
  if (stage->pinned) hipHostFree(stage->pinned);
  *stage = UploadStage{};
}
extern "C" {

int bahip_set_keyframes(bahip_context* ctx, const bahip_keyframe* keyframes, int num_keyframes) {
  REQUIRE(num_keyframes >= 0, "negative keyframe count");
  ctx->host_kfs.resize(num_keyframes);
  for (int k = 0; k < num_keyframes; ++k) {
    KfEntry e{};   // keyframe sharding: the images of a keyframe that lives on another rank are not looked at (null pointers)
    if (kf_owned(ctx, k) && make_entry(ctx, keyframes[k].frame, 1 + (size_t)k, &e)) return 1;
    fill_pose(&e, keyframes[k].global_T_frame);
    e.activation = keyframes[k].activation;
    ctx->host_kfs[k] = e;
  }
  if (num_keyframes > ctx->kfs_capacity) {
    size_t cap = (size_t)ctx->kfs_capacity;
    if (grow_device(&ctx->dev_kfs, &cap, (size_t)num_keyframes, 64, "the keyframe table")) return 1;
    ctx->kfs_capacity = (int)cap;
  }
  ctx->num_kfs = num_keyframes;
  ctx->have_covisibility = false;   // lists refer to the previous binding
  ctx->window.clear();
  // (through a page-locked stage of the context: no host wait; everything that reads the table is ordered behind the copy on this stream)
  if (num_keyframes > 0 && stage_upload(&ctx->stage_kfs, ctx->dev_kfs, ctx->host_kfs.data(), sizeof(KfEntry) * num_keyframes, ctx->stream)) return 1;
  return 0;
}

int bahip_get_keyframe_poses(bahip_context* ctx, float* out, int num_keyframes) {
  REQUIRE(num_keyframes <= ctx->num_kfs, "more poses requested than keyframes bound");
  if (num_keyframes == 0) return 0;
  HIP_TRY(hipMemcpyAsync(ctx->host_kfs.data(), ctx->dev_kfs, sizeof(KfEntry) * ctx->num_kfs, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < num_keyframes; ++k) memcpy(out + 7 * k, ctx->host_kfs[k].global_T_frame, 7 * sizeof(float));
  return 0;
}

// ---- instrumentation ------------------------------------------------------------------------------------------------
int bahip_set_profiling(bahip_context* ctx, int enabled) {
  ctx->profiling = enabled;
  for (StageTimer& t : ctx->timers) { t.used = 0; t.units = 0; }
  return 0;
}

int bahip_last_stage_time_ms(bahip_context* ctx, int stage, float* ms_out, int* launches_out) {
  REQUIRE(stage >= 0 && stage < 8, "stage out of range");
  StageTimer& t = ctx->timers[stage];
  float total = 0.f;
  int launches = 0;
  for (int i = 0; i < t.used; ++i) {
    if (t.skip[i]) continue;
    HIP_TRY(hipEventSynchronize(t.ev[2 * i + 1]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t.ev[2 * i], t.ev[2 * i + 1]));
    total += ms;
    ++launches;
  }
  *ms_out = total;
  if (launches_out) *launches_out = launches;
  return 0;
}

int bahip_stage_work_units(bahip_context* ctx, int stage, long long* units_out) {
  REQUIRE(stage >= 0 && stage < 8 && units_out != nullptr, "stage out of range");
  *units_out = ctx->timers[stage].units;
  return 0;
}

}  // extern "C"
