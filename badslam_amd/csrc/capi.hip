// capi.hip -- the extern "C" boundary declared in include/badslam_hip.h.
// Owns only scratch (device keyframe table, pose work items, scan temp, counters); every image
// and the surfel buffer are borrowed from the caller.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <thread>

#include <string>
#include <vector>

#include "ba_launch.h"
#include "exact_sum.h"
#include "ldlt.h"
#include "se3_device.h"

using namespace bahip;

namespace {

thread_local std::string g_last_error;

int fail(const char* what, const char* file, int line, hipError_t e = hipSuccess) {
  char buf[512];
  if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s failed at %s:%d: %s", what, file, line, hipGetErrorString(e));
  else snprintf(buf, sizeof(buf), "%s (%s:%d)", what, file, line);
  g_last_error = buf;
  return 1;
}

#define HIP_TRY(expr)                                                  \
  do {                                                                 \
    hipError_t _e = (expr);                                            \
    if (_e != hipSuccess) return fail(#expr, __FILE__, __LINE__, _e);  \
  } while (0)
#define REQUIRE(cond, msg)                                             \
  do {                                                                 \
    if (!(cond)) return fail(msg, __FILE__, __LINE__);                 \
  } while (0)
#define CHECK_LAUNCH() HIP_TRY(hipGetLastError())

// Scratch of the test hooks: freed on every return path.
struct DevMem {
  void* p = nullptr;
  ~DevMem() { if (p) hipFree(p); }
  template <typename T> T* as() const { return static_cast<T*>(p); }
};

struct StageTimer {
  std::vector<hipEvent_t> ev;   // pairs (start, stop)
  std::vector<char> skip;       // per pair: not a launch that did work (queued ahead in vain): left out of sums and counts
  int used = 0;                 // number of pairs used by the last call (mode 1) / since set_profiling (mode 2)
  long long units = 0;          // work units of those launches (stage 2: keyframes still iterating)
};

}  // namespace

// Tiled BA planes of one frame (ba_device.h).  Opaque to the C API.
struct bahip_frame_planes {
  uint32_t* geom = nullptr;
  uint32_t* lumafp = nullptr;
  int width = 0, height = 0, cwidth = 0, cheight = 0;
};

struct bahip_context {
  hipStream_t stream = nullptr;
  bool have_intrinsics = false;
  bahip_camera color_cam{}, depth_cam{};
  bahip_depth_params dp{};
  Intrinsics in{};

  std::vector<KfEntry> host_kfs;
  KfEntry* dev_kfs = nullptr;
  int kfs_capacity = 0;
  int num_kfs = 0;

  PoseWork* dev_work = nullptr;
  HbFixed* dev_Hb = nullptr;      // pose normal equations in fixed point (ba_device.h: HbFixed)
  int work_capacity = 0;
  KfEntry* dev_frame1 = nullptr;   // single-frame table for EstimateFramePose / AccumulatePoseEstimationCoeffs
  PoseWork* dev_work1 = nullptr;
  PoseWork* pinned_work1 = nullptr;
  HbFixed* dev_Hb1 = nullptr;
  uint32_t* dev_tile_counters = nullptr;   // persistent pose sweep: two sets of 8 tile counters (kernels_pose.hip)
  int pose_parity = 0;                     // the set the next persistent launch draws from

  int* dev_counter = nullptr;      // [0] generic counter, [1..2] min/max depth bits
  int* pinned_i = nullptr;         // 16 ints
  float* pinned_f = nullptr;       // 128 floats

  uint8_t* dev_flags = nullptr;    // W*H new-surfel flags
  uint32_t* dev_indices = nullptr; // W*H scan output
  size_t px_capacity = 0;
  void* scan_temp = nullptr;
  size_t scan_temp_bytes = 0;
  int* dev_covis = nullptr;
  float* dev_covis_T = nullptr;
  int covis_capacity = 0;
  // co-visibility lists of the bound keyframes (CSR over bound indices), for the device-side activation state machine
  std::vector<int> covis_offsets, covis_indices;
  int* dev_covis_csr = nullptr;    // offsets (K + 1) followed by the indices
  size_t covis_csr_capacity = 0;
  bool capacity_exceeded = false;  // last bahip_create_surfels_for_keyframe did not fit (bahip_context_take_capacity_exceeded)
  bool have_covisibility = false;
  std::vector<uint8_t> window;     // per bound keyframe: inside the fixed active window (bahip_set_activation_window)
  uint8_t* dev_window = nullptr;
  size_t window_capacity = 0;
  PoseWork* pinned_work = nullptr;   // read-back of the pose work items + their counter records (page-locked)
  const void* supporting_planes_empty = nullptr;   // the supporting planes (by their first plane) that the last merge call left empty
  bool row_major_creation = false;   // new surfels of a keyframe appended in row-major pixel order (the reference's) instead of tile-major
  bool poll_disabled = false;        // the host copy of the pose counters is not updated by the kernel on this system: synchronise instead
  // lifecycle batch (bahip_lifecycle_batch_begin): bounding spheres of the cloud's whole tiles, for the per-keyframe sweeps of a batch
  void* dev_lifecycle_bounds = nullptr;
  size_t lifecycle_bounds_capacity = 0;   // tiles
  uint32_t lifecycle_bounds_tiles = 0;    // 0: no batch open
  const void* lifecycle_bounds_data = nullptr;   // the surfel buffer they describe
  // ... and, when the batch knows its frames (bahip_lifecycle_batch_set_frames), which of those tiles each frame can see
  std::vector<float> lifecycle_frames;           // 12 floats per frame: frame_T_global as given
  std::vector<uint32_t> lifecycle_list_offsets, lifecycle_list_counts;
  float* dev_lifecycle_frames = nullptr;
  uint32_t* dev_lifecycle_cursors = nullptr;     // [2 * capacity]: cursors, offsets
  size_t lifecycle_frames_capacity = 0;
  uint32_t* dev_lifecycle_lists = nullptr;
  size_t lifecycle_lists_capacity = 0;
  void* dev_tile_bounds = nullptr;   // bounding sphere per 64-surfel tile, written by the first pose round of a phase
  size_t tile_bounds_bytes = 0;
  // heavy work first (wave_cull.h: scheduled_tile): candidates per tile counted by the first pose round of a phase over the
  // keyframe table (or by the PCG init sweep), and the schedule built from them, valid for grids of tile_order_tiles (padded)
  // tiles (0: none yet)
  uint32_t* dev_tile_cost = nullptr;
  uint32_t* dev_tile_order = nullptr;
  size_t tile_schedule_capacity = 0;   // tiles
  uint32_t tile_order_tiles = 0;
  int phases_since_schedule = 0;       // the schedule is rebuilt when the grid changes and every kSchedulePhases-th phase
  uint32_t tile_order_unavailable_tiles = 0;   // a grid the order kernel cannot schedule (too many runs): no census for it again
  bool tile_order_unavailable_for(uint32_t padded_tiles) const { return padded_tiles != 0 && tile_order_unavailable_tiles == padded_tiles; }
  int* dev_loop_ctl = nullptr;     // device-driven BA loop (bahip_alternating_iterations): kLoopWords control words ...
  int* host_loop_ctl = nullptr;    // ... their mapped host copy, followed by kLoopLogSlots words of per-round log
  int rounds_hint_table = 1, rounds_hint_frame = 1;   // Gauss-Newton rounds the previous pose phase took (keyframe table / single frame)

  float* intr_scratch = nullptr;   // intrinsics step: (64 + 8 S) doubles, then (64 + 8 S) floats + Schur partials
  int intr_capacity = 0;
  // append buffers of the intrinsics sweep's per-cell records (ba_launch.h: IntrBins), sized from the previous call's counts
  uint32_t* intr_bin_cursors = nullptr;   // device, intr_bin_count words
  uint32_t* intr_bin_records = nullptr;
  uint32_t* intr_bin_counts_host = nullptr;   // pinned copy of the cursors after the sweep
  int intr_bin_count = 0;
  uint32_t intr_bin_capacity = 0;         // records per block the buffers hold
  uint32_t intr_bin_wanted = 0;           // records per block the next call should have room for (0: estimate)
  int intr_bin_forced = -1;
  int intr_bin_last_overflow = 0;         // did the last call have records that did not fit?               // bahip_debug_set_intrinsics_bin_capacity: >= 0 fixes the capacity (0: no binning)

  float* pcg_buf = nullptr;        // PCG vectors r, M, delta, g, p (5 * pcg_capacity floats) + 16 scalars
  size_t pcg_capacity = 0;
  void* pcg_exact = nullptr;       // exact accumulators of the PCG solve (ExactCell[pcg_exact_capacity]; kernels_pcg.hip)
  size_t pcg_exact_capacity = 0;
  void* pcg_stage_ctl = nullptr;   // stage API (bahip_pcg_begin ...): a control block that never stops, the head size the
  uint32_t pcg_stage_head = 0;     // accumulators were set up for, and the bahip_pcg_step1 calls since the last step 2
  int pcg_stage_step1_calls = 0;
  int world = 0;                   // ranks of the RCCL communicator (0 = none)
  int kf_rank = 0, kf_world = 1;   // keyframe sharding (bahip_context_set_keyframe_sharding): keyframe k lives on rank k % kf_world (1, 2, 4 or 8)
  int sum_classes = 4;             // interleaved partial sums per surfel of the normals / geometry passes: 4 or 8 (bahip_context_set_sum_classes)
  float* kf_partials = nullptr;    // class partials of the geometry step (normals, then position) / hit words of the activation
  size_t kf_partials_capacity = 0; // floats
  long long exchange_calls = 0;    // sums over the ranks requested since the last reset (bahip_exchange_stats), and their bytes
  long long exchange_bytes = 0;

  // planes packed by the library itself for frames handed over without bahip_frame.planes:
  // slot 0 = the single frame of the per-frame entry points, slot 1 + k = bound keyframe k
  std::vector<bahip_frame_planes*> auto_planes;

  bahip_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  void* rccl_comm = nullptr;       // ncclComm_t created by bahip_context_init_rccl (native all-reduce on ctx->stream)

  int profiling = 0;               // 0 off, 1 last call of each stage, 2 cumulative since bahip_set_profiling
  StageTimer timers[8];              // 0 activation, 1 geometry, 2 pose accumulate, 3 pose solve, 4 intrinsics (whole step), 5 PCG step-1 sweep,
                                     // 6 intrinsics sweep alone, 7 intrinsics reduction of the binned records alone
};

namespace {

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
template <typename T>
int grow_device(T** ptr, size_t* capacity, size_t need, size_t slack, const char* what) {
  if (need <= *capacity && *ptr) return 0;
  T* grown = nullptr;
  const size_t cap = need + slack;
  if (hipMalloc(&grown, sizeof(T) * cap) != hipSuccess) {
    char buf[160];
    snprintf(buf, sizeof(buf), "hipMalloc of %zu bytes for %s failed", sizeof(T) * cap, what);
    g_last_error = buf;
    return 1;
  }
  hipFree(*ptr);
  *ptr = grown;
  *capacity = cap;
  return 0;
}

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
inline bool timer_on(const bahip_context* ctx, int stage) { return ctx->profiling && (ctx->profiling != 3 || stage == 2); }

void timer_begin(bahip_context* ctx, int stage, bool first, int units = 1) {
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

int reduce_over_ranks(bahip_context* ctx, void* buffer, size_t count, int dtype);
inline bool kf_sharded(const bahip_context* ctx) { return ctx->kf_world > 1; }

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

static int g_tile_order_enabled = [] { const char* e = getenv("BAHIP_TILE_ORDER"); return e ? atoi(e) : 1; }();
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
int g_fused_iteration_begin = [] { const char* e = getenv("BAHIP_FUSED_ITERATION_BEGIN"); return e ? atoi(e) : 0; }();
int g_pose_rounds_ahead = [] { const char* e = getenv("BAHIP_POSE_ROUNDS_AHEAD"); return e ? atoi(e) : 0; }();
int run_pose_rounds(bahip_context* ctx, bool use_depth, bool use_desc, const KfEntry* dev_frames, KfEntry* dev_frames_rw,
                    PoseWork* dev_work, HbFixed* dev_Hb, int num_work, const SurfelsView& s, int write_back, int update_activation,
                    PoseWork* host_work /* page-locked, num_work + kPoseTailRecords records */, int* rounds_out,
                    bool schedule = false /* a phase over the keyframe table: its first round counts the candidates per tile and the
                    run order of the following sweeps is rebuilt from them */, int* rounds_hint = nullptr,
                    int first_round = 0, int first_iterating = -1 /* continue a phase whose rounds [0, first_round) have run (the
                    device-driven loop hands over a phase that needs more rounds than it had queued) */,
                    const PoseLoopControl* loop_stats = nullptr /* keeps the loop's totals going (never ends a phase) */) {
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

// ---- RCCL, loaded on first use ---------------------------------------------------------------------------------------
// The prototypes below restate the four RCCL entry points used (rccl.h: ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
// ncclCommDestroy, ncclGetErrorString); ncclFloat = 7, ncclInt64 = 4, ncclSum = 0 in every NCCL / RCCL release.
struct RcclId { char internal[BAHIP_RCCL_UNIQUE_ID_BYTES]; };
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId /* ncclUniqueId, by value */, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int load_rccl() {
  if (g_rccl.handle) return 0;
  void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail("librccl.so could not be loaded (multi-GPU needs RCCL)", __FILE__, __LINE__);
  g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
  g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
    return fail("librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy", __FILE__, __LINE__);
  g_rccl.handle = h;
  return 0;
}
int rccl_fail(const char* what, int rc) {
  char buf[256];
  snprintf(buf, sizeof(buf), "%s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
  g_last_error = buf;
  return 1;
}
int rccl_allreduce(bahip_context* ctx, void* buffer, size_t count, int dtype) {
  const int nccl_type = dtype == BAHIP_SUM_I64 ? 4 /* ncclInt64 */ : dtype == BAHIP_SUM_F64 ? 8 /* ncclDouble */ : 7 /* ncclFloat */;
  const int rc = g_rccl.AllReduce(buffer, buffer, count, nccl_type, 0 /* ncclSum */, ctx->rccl_comm, ctx->stream);
  return rc == 0 ? 0 : rccl_fail("ncclAllReduce", rc);
}

// Element-wise sum of a device buffer over all ranks, in place, ordered on the context's stream: the caller's hook if one is
// installed (it overrides: a caller that installs a hook after bahip_context_init_rccl wants the hook), else the native RCCL
// path if a communicator exists, else nothing (single GPU).
int reduce_over_ranks(bahip_context* ctx, void* buffer, size_t count, int dtype) {
  if (count == 0) return 0;
  if (ctx->allreduce || ctx->rccl_comm) {
    ctx->exchange_calls += 1;
    ctx->exchange_bytes += (long long)count * (dtype == BAHIP_SUM_F32 ? 4 : 8);
  }
  if (ctx->allreduce) {
    if (ctx->allreduce(buffer, count, dtype, ctx->stream, ctx->allreduce_user) != 0) return fail("all-reduce hook failed", __FILE__, __LINE__);
    return 0;
  }
  if (ctx->rccl_comm) return rccl_allreduce(ctx, buffer, count, dtype);
  return 0;
}
inline bool is_sharded(const bahip_context* ctx) { return ctx->allreduce != nullptr || ctx->rccl_comm != nullptr; }
inline bool kf_owned(const bahip_context* ctx, int k) { return (k & (ctx->kf_world - 1)) == ctx->kf_rank; }
#define REQUIRE_NO_KF_SHARDING(what) \
  REQUIRE(!kf_sharded(ctx), what " is not available under keyframe sharding (its per-surfel sums run over all keyframes in order): use surfel sharding")

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

}  // namespace

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
  hipFree(ctx->dev_flags); hipFree(ctx->dev_indices); hipFree(ctx->scan_temp);
  hipFree(ctx->dev_covis); hipFree(ctx->dev_covis_T); hipFree(ctx->dev_covis_csr); hipFree(ctx->dev_tile_bounds); hipFree(ctx->dev_lifecycle_bounds); hipFree(ctx->dev_lifecycle_frames); hipFree(ctx->dev_lifecycle_cursors); hipFree(ctx->dev_lifecycle_lists); hipFree(ctx->dev_window);
  hipFree(ctx->intr_bin_cursors); hipFree(ctx->intr_bin_records); hipHostFree(ctx->intr_bin_counts_host);
  hipFree(ctx->intr_scratch); hipFree(ctx->pcg_buf); hipFree(ctx->pcg_exact); hipFree(ctx->pcg_stage_ctl); hipFree(ctx->kf_partials);
  hipFree(ctx->dev_tile_cost); hipFree(ctx->dev_tile_order);
  hipFree(ctx->dev_loop_ctl);
  if (ctx->host_loop_ctl) hipHostFree(ctx->host_loop_ctl);
  if (ctx->rccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->rccl_comm);
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

int bahip_context_is_sharded(bahip_context* ctx) { return (ctx->allreduce != nullptr || ctx->rccl_comm != nullptr) ? 1 : 0; }

int bahip_context_set_allreduce(bahip_context* ctx, bahip_allreduce_fn fn, void* user) {
  ctx->allreduce = fn;
  ctx->allreduce_user = user;
  return 0;
}

int bahip_context_set_sum_classes(bahip_context* ctx, int classes) {
  REQUIRE(classes == 4 || classes == 8, "the per-surfel sums of the normals / geometry passes are defined over 4 or 8 keyframe classes");
  REQUIRE(ctx->kf_world <= classes, "keyframe sharding over more ranks than classes");
  ctx->sum_classes = classes;
  ctx->in.sum_classes = classes;
  return 0;
}

int bahip_context_set_keyframe_sharding(bahip_context* ctx, int rank, int world) {
  REQUIRE(world == 1 || world == 2 || world == 4 || world == 8, "keyframe sharding: world must be 1, 2, 4 or 8 (a rank holds whole keyframe classes)");
  REQUIRE(world <= ctx->sum_classes, "keyframe sharding over 8 ranks needs the 8-class definition of the per-surfel sums: bahip_context_set_sum_classes(ctx, 8) "
                                     "first (on the single-GPU run it is compared with as well: the class count is part of the sums' definition)");
  REQUIRE(rank >= 0 && rank < world, "keyframe sharding: rank out of range");
  ctx->kf_rank = rank; ctx->kf_world = world;
  return 0;
}

int bahip_rccl_get_unique_id(char unique_id_out[BAHIP_RCCL_UNIQUE_ID_BYTES]) {
  REQUIRE(unique_id_out != nullptr, "bahip_rccl_get_unique_id: NULL argument");
  if (load_rccl()) return 1;
  RcclId id;
  const int rc = g_rccl.GetUniqueId(&id);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  memcpy(unique_id_out, id.internal, BAHIP_RCCL_UNIQUE_ID_BYTES);
  return 0;
}

int bahip_context_init_rccl(bahip_context* ctx, const char unique_id[BAHIP_RCCL_UNIQUE_ID_BYTES], int rank, int world_size) {
  REQUIRE(ctx != nullptr && unique_id != nullptr, "bahip_context_init_rccl: NULL argument");
  REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "bahip_context_init_rccl: rank / world_size out of range");
  if (load_rccl()) return 1;
  if (ctx->rccl_comm) { g_rccl.CommDestroy(ctx->rccl_comm); ctx->rccl_comm = nullptr; }
  RcclId id;
  memcpy(id.internal, unique_id, BAHIP_RCCL_UNIQUE_ID_BYTES);
  void* comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, world_size, id, rank);
  if (rc != 0) return rccl_fail("ncclCommInitRank", rc);
  ctx->rccl_comm = comm;
  ctx->world = world_size;
  return 0;
}

// The first exchange of a run, as a probe: every rank contributes 1 through whatever transport the context uses (the hook or the
// native RCCL communicator), on the context's stream, and the host waits for the sum with a time limit.  A multi-rank job whose
// collective cannot complete (a rank that never arrived, a fabric that does not come up) otherwise hangs in the first BA
// iteration without a word; this returns an error that says which exchange it was and how long it waited.
int bahip_context_count_ranks(bahip_context* ctx, int timeout_ms, int* ranks_out) {
  REQUIRE(ctx != nullptr && ranks_out != nullptr, "bahip_context_count_ranks: NULL argument");
  *ranks_out = 1;
  if (!is_sharded(ctx)) return 0;
  DevMem word;
  HIP_TRY(hipMalloc(&word.p, sizeof(long long)));
  const long long one = 1;
  HIP_TRY(hipMemcpyAsync(word.p, &one, sizeof(one), hipMemcpyHostToDevice, ctx->stream));
  if (reduce_over_ranks(ctx, word.p, 1, BAHIP_SUM_I64)) return 1;
  hipEvent_t done;
  HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(done, ctx->stream));
  const auto start = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t state = hipEventQuery(done);
    if (state == hipSuccess) break;
    if (state != hipErrorNotReady) { hipEventDestroy(done); return fail("the probe exchange failed on the device", __FILE__, __LINE__); }
    if (timeout_ms > 0 && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count() > timeout_ms) {
      // (the event and the buffer are left alone: the collective may still own them)
      word.p = nullptr;
      return fail(ctx->allreduce ? "the first all-reduce (hook transport) did not complete within the time limit: not every rank reached it"
                                 : "the first ncclAllReduce (native RCCL transport over xGMI) did not complete within the time limit: not every rank "
                                   "reached it, or the communicator's links did not come up (NCCL_DEBUG=INFO shows the ring)", __FILE__, __LINE__);
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  HIP_TRY(hipEventDestroy(done));
  long long seen = 0;
  HIP_TRY(hipMemcpy(&seen, word.p, sizeof(seen), hipMemcpyDeviceToHost));
  *ranks_out = (int)seen;
  return 0;
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
  ctx->stream = static_cast<hipStream_t>(hip_stream);
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
  if (ctx->row_major_creation) ctx->in.create_tile = std::max(ctx->in.width, ctx->in.height);
  ctx->have_intrinsics = true;
  return 0;
}

int bahip_context_set_creation_order(bahip_context* ctx, int row_major) {
  ctx->row_major_creation = row_major != 0;
  if (ctx->have_intrinsics) ctx->in.create_tile = ctx->row_major_creation ? std::max(ctx->in.width, ctx->in.height) : 8 * ctx->in.cell;
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
  if (num_keyframes > 0) {
    HIP_TRY(hipMemcpyAsync(ctx->dev_kfs, ctx->host_kfs.data(), sizeof(KfEntry) * num_keyframes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // host_kfs is pageable
  }
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
namespace {
constexpr int kLoopLogSlots = 4096;
int g_device_loop_enabled = [] { const char* e = getenv("BAHIP_DEVICE_LOOP"); return (e && atoi(e) == 0) ? 0 : 1; }();
}
int bahip_debug_set_device_loop(int enabled) { g_device_loop_enabled = enabled ? 1 : 0; return 0; }
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
    if (wait_for_pose_sequence(ctx, ctx->pinned_work, ctx->dev_work, K, sequence)) return 1;
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
  HIP_TRY(hipMemcpyAsync(ctx->dev_covis_csr, ctx->covis_offsets.data(), sizeof(int) * (K + 1), hipMemcpyHostToDevice, ctx->stream));
  if (total) HIP_TRY(hipMemcpyAsync(ctx->dev_covis_csr + K + 1, ctx->covis_indices.data(), sizeof(int) * total, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));   // the vectors are pageable
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
  if (num_keyframes) {
    HIP_TRY(hipMemcpyAsync(ctx->dev_window, ctx->window.data(), num_keyframes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
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
  uint32_t* offsets = ctx->dev_lifecycle_cursors + ctx->lifecycle_frames_capacity;
  std::vector<uint32_t> counts(num_frames), starts(num_frames);
  HIP_TRY(hipMemcpyAsync(ctx->dev_lifecycle_frames, frame_T_global_3x4, (size_t)num_frames * 12 * sizeof(float), hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemsetAsync(cursors, 0, (size_t)num_frames * sizeof(uint32_t), st));
  launch_lifecycle_visible_tiles(st, ctx->in, ctx->dev_lifecycle_frames, num_frames, ctx->dev_lifecycle_bounds, tiles, nullptr, cursors, nullptr);
  CHECK_LAUNCH();
  HIP_TRY(hipMemcpyAsync(counts.data(), cursors, (size_t)num_frames * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  size_t total = 0;
  for (int f = 0; f < num_frames; ++f) { starts[f] = (uint32_t)total; total += counts[f]; }
  if (total > ctx->lifecycle_lists_capacity) {
    uint32_t* lists = nullptr;
    const size_t capacity = total + total / 4 + 4096;
    HIP_TRY(hipMalloc(&lists, capacity * sizeof(uint32_t)));
    hipFree(ctx->dev_lifecycle_lists);
    ctx->dev_lifecycle_lists = lists;
    ctx->lifecycle_lists_capacity = capacity;
  }
  if (total > 0) {
    HIP_TRY(hipMemcpyAsync(offsets, starts.data(), (size_t)num_frames * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(cursors, 0, (size_t)num_frames * sizeof(uint32_t), st));
    launch_lifecycle_visible_tiles(st, ctx->in, ctx->dev_lifecycle_frames, num_frames, ctx->dev_lifecycle_bounds, tiles, offsets, cursors, ctx->dev_lifecycle_lists);
    CHECK_LAUNCH();
    HIP_TRY(hipStreamSynchronize(st));   // `starts` is pageable and goes out of scope
  }
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
  for (int j = 0; j < num_keyframes; ++j) {
    const KfEntry& e = ctx->host_kfs[keyframe_indices[j]];
    const uint32_t* size_in = size_cell[j & 1];
    // what the cloud can hold by now at most: the grid of the sweep; the size itself is read on the device
    bahip_surfels bound = *surfels;
    bound.surfels_size = (uint32_t)std::min<uint64_t>(surfels->capacity, (uint64_t)surfels->surfels_size + (uint64_t)j * cells);
    const SurfelsView s = make_view(&bound);
    ctx->supporting_planes_empty = nullptr;
    launch_supporting_fill(st, sup, ctx->in.cf_width, ctx->in.cf_height);
    launch_supporting_insert(st, ctx->in, e, s, sup, lifecycle_cull_for(ctx, surfels, e.pose.F), size_in);
    launch_create_flag(st, ctx->in, e, sup, ctx->dev_flags);
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
  const float inv_cell = 1.0f / grid_cell_size;
  HIP_TRY(sort_surfels_spatially(ctx->stream, make_view(surfels), inv_cell));
  return 0;
}

// B/kernel_opt_intrinsics.cc:39-281
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
  // Append buffers for the per-cell records (kernels_intrinsics.hip).  Their size follows the demand the previous call saw
  // (+ 25 %); before the first call it is an estimate, and a call that finds them too small still gives the same result: the
  // records that do not fit go out as atomics.
  IntrBins bins{nullptr, nullptr, 0, 1};
  const int num_bins = intrinsics_bin_count(ctx->in, &bins.bins_x);
  if (optimize_depth) {
    if (num_bins != ctx->intr_bin_count) {
      hipFree(ctx->intr_bin_cursors); hipHostFree(ctx->intr_bin_counts_host); hipFree(ctx->intr_bin_records);
      ctx->intr_bin_cursors = nullptr; ctx->intr_bin_counts_host = nullptr; ctx->intr_bin_records = nullptr;
      ctx->intr_bin_capacity = 0; ctx->intr_bin_wanted = 0;
      HIP_TRY(hipMalloc(&ctx->intr_bin_cursors, sizeof(uint32_t) * (size_t)num_bins));
      HIP_TRY(hipHostMalloc(&ctx->intr_bin_counts_host, sizeof(uint32_t) * (size_t)num_bins));
      ctx->intr_bin_count = num_bins;
    }
    // Keep what there is unless the previous call OVERFLOWED it (intr_bin_wanted is raised only then).  Round 4, configs[4]: sized
    // as "the previous call's largest count + 25 %" the request crept up by 64 records per call while the poses converged, and each
    // time 53 GB of record buffers were freed and allocated again -- 1.5 to 2.5 s per reallocation, in whichever call it fell
    // (gpurun_out/r4_call30: 3.4 BA iterations/s with one of them inside the timed call, 14.4 without).
    uint64_t want = std::max<uint64_t>(ctx->intr_bin_capacity, ctx->intr_bin_wanted);
    if (!want) want = (uint64_t)surfels->surfels_size * (uint64_t)std::min(ctx->num_kfs, 16) * 2 / (uint64_t)num_bins + 4096;
    if (ctx->intr_bin_forced >= 0) want = (uint64_t)ctx->intr_bin_forced;
    const uint64_t limit = (96ull << 30) / (intrinsics_bin_record_bytes() * (uint64_t)num_bins);   // at most 96 GB of records
    want = std::min(want, limit);
    if (want > ctx->intr_bin_capacity || (ctx->intr_bin_forced >= 0 && want != ctx->intr_bin_capacity)) {
      static const bool host_timing = getenv("BADSLAM_HOST_TIMING") != nullptr;
      const auto t0 = std::chrono::steady_clock::now();
      hipFree(ctx->intr_bin_records);
      ctx->intr_bin_records = nullptr; ctx->intr_bin_capacity = 0;
      const uint64_t cap = (want + 63) / 64 * 64;
      if (cap) HIP_TRY(hipMalloc(&ctx->intr_bin_records, intrinsics_bin_record_bytes() * cap * (uint64_t)num_bins));
      ctx->intr_bin_capacity = (uint32_t)cap;
      if (host_timing)
        fprintf(stderr, "[intrinsics record buffers] %d buffers x %llu records = %.2f GB (re)allocated in %.1f ms\n", num_bins, (unsigned long long)cap,
                (double)(intrinsics_bin_record_bytes() * cap * (uint64_t)num_bins) / 1e9,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    bins.cursors = ctx->intr_bin_cursors; bins.records = ctx->intr_bin_records; bins.capacity = ctx->intr_bin_capacity;
  }
  timer_begin(ctx, 4, true);
  HIP_TRY(hipMemsetAsync(glob_d, 0, sizeof(double) * (64 + 8 * (size_t)S), ctx->stream));
  if (bins.capacity) HIP_TRY(hipMemsetAsync(bins.cursors, 0, sizeof(uint32_t) * (size_t)num_bins, ctx->stream));
  timer_begin(ctx, 6, true);
  launch_intrinsics_accumulate(ctx->stream, optimize_depth != 0, optimize_color != 0, ctx->in, ctx->dev_kfs, ctx->num_kfs,
                               make_view(surfels), glob_d, cells_d, bins, tile_order_for(ctx, surfels->surfels_size));
  timer_end(ctx, 6);
  timer_begin(ctx, 7, true);
  launch_intrinsics_bin_reduce(ctx->stream, optimize_depth != 0, ctx->in, make_view(surfels), cells_d, bins);
  timer_end(ctx, 7);
  CHECK_LAUNCH();
  if (bins.capacity)
    HIP_TRY(hipMemcpyAsync(ctx->intr_bin_counts_host, bins.cursors, sizeof(uint32_t) * (size_t)num_bins, hipMemcpyDeviceToHost, ctx->stream));
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
  if (bins.capacity) {
    uint32_t most = 0;
    for (int b = 0; b < num_bins; ++b) most = std::max(most, ctx->intr_bin_counts_host[b]);
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

int bahip_debug_set_pose_form(int form) {
  REQUIRE(form == 0 || form == 1 || form == 2, "pose form must be 0 (automatic), 1 (one tile per wavefront, global atomics) or 2 (persistent, LDS table)");
  set_pose_form(form);
  return 0;
}

int bahip_debug_set_launch_shapes(int tile_waves, int pose_parts) {
  REQUIRE(tile_waves == 0 || tile_waves == 1 || tile_waves == 4, "tile_waves must be 0 (automatic), 1 or 4");
  REQUIRE(pose_parts == 0 || pose_parts == 1 || pose_parts == 2 || pose_parts == 4 || pose_parts == 8, "pose_parts must be 0, 1, 2, 4 or 8");
  set_tile_waves(tile_waves);
  set_pose_parts(pose_parts);
  return 0;
}

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

namespace {
float* host_row(const SurfelsView& v, int row) { return reinterpret_cast<float*>(reinterpret_cast<char*>(v.data) + (size_t)row * v.pitch); }
// number of surfels of a cloud of `total` that the chunk-cyclic partition gives to `rank`
uint32_t shard_size_of(uint32_t total, int rank, int world, uint32_t chunk) {
  const uint64_t stride = (uint64_t)chunk * (uint64_t)world;
  const uint64_t full = total / stride, rest = total % stride;
  const uint64_t begin = (uint64_t)rank * chunk;
  const uint64_t tail = rest > begin ? (rest - begin < chunk ? rest - begin : chunk) : 0;
  return (uint32_t)(full * chunk + tail);
}
}  // namespace

int bahip_gather_surfel_shards(bahip_context* ctx, const bahip_surfels* shard, uint32_t shard_surfel_count, int rank, int world, uint32_t chunk,
                               bahip_surfels* cloud, uint32_t* cloud_surfels_size_out, uint32_t* cloud_surfel_count_out) {
  REQUIRE(world >= 1 && rank >= 0 && rank < world && chunk > 0 && chunk % 64 == 0, "bahip_gather_surfel_shards: bad partition (chunks are whole 64-surfel tiles)");
  REQUIRE(world <= 64, "bahip_gather_surfel_shards: at most 64 ranks");
  hipStream_t st = ctx->stream;
  // every rank's (size, count): a sum over the ranks of a table that is zero except for the own row
  long long table[128] = {0};
  table[2 * rank] = shard->surfels_size; table[2 * rank + 1] = shard_surfel_count;
  DevMem dev_table;
  HIP_TRY(hipMalloc(&dev_table.p, sizeof(table)));
  HIP_TRY(hipMemcpyAsync(dev_table.p, table, sizeof(table), hipMemcpyHostToDevice, st));
  if (reduce_over_ranks(ctx, dev_table.p, 2 * (size_t)world, BAHIP_SUM_I64)) return 1;
  HIP_TRY(hipMemcpyAsync(table, dev_table.p, sizeof(table), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  uint64_t total = 0, count = 0;
  for (int r = 0; r < world; ++r) { total += (uint64_t)table[2 * r]; count += (uint64_t)table[2 * r + 1]; }
  REQUIRE(is_sharded(ctx) || world == 1, "bahip_gather_surfel_shards: world > 1 needs a communicator or an all-reduce hook");
  REQUIRE(total <= cloud->capacity, "bahip_gather_surfel_shards: the cloud buffer is too small for the union of the shards");
  REQUIRE(cloud->capacity % 8 == 0, "bahip_gather_surfel_shards: the cloud's capacity must be a multiple of 8 (rows travel as 64-bit words)");
  for (int r = 0; r < world; ++r)
    REQUIRE((uint64_t)table[2 * r] == shard_size_of((uint32_t)total, r, world, chunk),
            "bahip_gather_surfel_shards: the shards are not the chunk-cyclic partition of one cloud");
  cloud->surfels_size = (uint32_t)total;
  const SurfelsView sv = make_view(shard), cv = make_view(cloud);
  const size_t words = ((size_t)total + 1) / 2;   // int64 words per data row (rows start 8-byte aligned: pitched allocations)
  for (int row = 0; row < kSurfelAccum0; ++row) HIP_TRY(hipMemsetAsync(host_row(cv, row), 0, words * 8, st));
  if (cv.active) HIP_TRY(hipMemsetAsync(cv.active, 0, ((size_t)total + 7) / 8 * 8, st));
  launch_shard_to_cloud(st, sv, cv, (uint32_t)rank, (uint32_t)world, chunk);
  CHECK_LAUNCH();
  for (int row = 0; row < kSurfelAccum0; ++row)
    if (reduce_over_ranks(ctx, host_row(cv, row), words, BAHIP_SUM_I64)) return 1;
  if (cv.active && reduce_over_ranks(ctx, cv.active, ((size_t)total + 7) / 8, BAHIP_SUM_I64)) return 1;
  if (cloud_surfels_size_out) *cloud_surfels_size_out = (uint32_t)total;
  if (cloud_surfel_count_out) *cloud_surfel_count_out = (uint32_t)count;
  return 0;
}

int bahip_extract_surfel_shard(bahip_context* ctx, const bahip_surfels* cloud, int rank, int world, uint32_t chunk, bahip_surfels* shard,
                               uint32_t* shard_surfels_size_out) {
  REQUIRE(world >= 1 && rank >= 0 && rank < world && chunk > 0 && chunk % 64 == 0, "bahip_extract_surfel_shard: bad partition");
  const uint32_t mine = shard_size_of(cloud->surfels_size, rank, world, chunk);
  REQUIRE(mine <= shard->capacity, "bahip_extract_surfel_shard: the shard buffer is too small");
  shard->surfels_size = mine;
  launch_cloud_to_shard(ctx->stream, make_view(cloud), make_view(shard), (uint32_t)rank, (uint32_t)world, chunk);
  CHECK_LAUNCH();
  if (shard_surfels_size_out) *shard_surfels_size_out = mine;
  return 0;
}

int bahip_debug_set_intrinsics_bin_capacity(bahip_context* ctx, int records_per_block) {
  ctx->intr_bin_forced = records_per_block;
  return 0;
}
int bahip_debug_intrinsics_bin_stats(bahip_context* ctx, uint32_t* capacity_out, uint32_t* most_out, uint64_t* total_out) {
  uint32_t most = 0; uint64_t total = 0;
  for (int b = 0; b < ctx->intr_bin_count && ctx->intr_bin_counts_host; ++b) { most = std::max(most, ctx->intr_bin_counts_host[b]); total += ctx->intr_bin_counts_host[b]; }
  if (capacity_out) *capacity_out = ctx->intr_bin_capacity;
  if (most_out) *most_out = most;
  if (total_out) *total_out = total;
  return 0;
}

int bahip_exchange_stats(bahip_context* ctx, long long* calls_out, long long* bytes_out, int reset) {
  if (calls_out) *calls_out = ctx->exchange_calls;
  if (bytes_out) *bytes_out = ctx->exchange_bytes;
  if (reset) { ctx->exchange_calls = 0; ctx->exchange_bytes = 0; }
  return 0;
}

int bahip_stage_work_units(bahip_context* ctx, int stage, long long* units_out) {
  REQUIRE(stage >= 0 && stage < 8 && units_out != nullptr, "stage out of range");
  *units_out = ctx->timers[stage].units;
  return 0;
}

}  // extern "C"
