// capi_internal.h -- what the translation units of the C boundary (include/badslam_hip.h) share: the context, the error helpers and the
// helpers one unit defines and the others call.  Not part of the ABI.
//   capi.hip            context, allocation, streams, preprocessing entry points, scene binding, stage timers
//   capi_rccl.hip       transports of a multi-GPU run: the RCCL loader, the reduction over the ranks, shard gather / extract
//   capi_ba.hip         the stages of the alternating scheme, the Gauss-Newton rounds, the device-driven loop
//   capi_lifecycle.hip  supporting surfels, merging, creation (single and batched), deletion, compaction, spatial order
//   capi_solvers.hip    the intrinsics step and the PCG scheme (whole iteration and stage by stage)
//   capi_debug.hip      test hooks and experiment switches
#pragma once

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

namespace bahip_capi {
using namespace bahip;

extern thread_local std::string g_last_error;   // bahip_last_error() (capi.hip)

int fail(const char* what, const char* file, int line, hipError_t e = hipSuccess);   // sets the error text, returns 1

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


}  // namespace bahip_capi
using bahip_capi::DevMem;
using bahip_capi::StageTimer;
using namespace bahip;

// Tiled BA planes of one frame (ba_device.h).  Opaque to the C API.
struct bahip_frame_planes {
  uint32_t* geom = nullptr;
  uint32_t* lumafp = nullptr;
  int width = 0, height = 0, cwidth = 0, cheight = 0;
};

// A small host-to-device upload without a host wait: the bytes are copied into a page-locked buffer the context owns and go out with
// hipMemcpyAsync; the event says when the buffer may be overwritten (checked -- normally long over -- by the next upload through the
// same stage).  Scene binding used to synchronise the stream three times per BundleAdjustment call for its three tables.
struct UploadStage {
  void* pinned = nullptr;
  size_t capacity = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
};
int stage_upload(UploadStage* stage, void* dev_dst, const void* src, size_t bytes, hipStream_t stream,
                 const void* src2 = nullptr, size_t bytes2 = 0, void* dev_dst2 = nullptr);
void stage_free(UploadStage* stage);

constexpr int kIntrMaxSlices = 16;   // slices of the intrinsics sweep (capi_solvers.hip): automatic up to 8, forced up to 16 (half the record memory again)
struct bahip_context {
  UploadStage stage_kfs, stage_covis, stage_window;
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
  // a creation batch as a chain of one launch per keyframe (bahip_create_surfels_for_keyframes; kernels_lifecycle.hip: create_chain_kernel):
  // per keyframe of the batch the bytes "cell occupied" and "pixel would create a surfel", and the batch's item table
  void* dev_merge_batch = nullptr;    // bahip_merge_surfels_for_keyframes by cell lists: frame table, counts, offsets, pair cells, members, scan temporary
  size_t merge_batch_bytes = 0;
  void* dev_sort_scratch = nullptr;   // bahip_sort_surfels_spatially: keys, indices, a dense copy of the data rows, the library's temporary
  size_t sort_scratch_bytes = 0;
  void* dev_create_batch = nullptr;   // occupancy, candidates, their scan, the compact candidate list with its records, the item table
  size_t create_batch_bytes = 0;
  uint32_t* merge_planes[BAHIP_MERGE_BUFFER_COUNT] = {};   // the second set of supporting planes of a pipelined merge batch (bahip_merge_surfels_for_keyframes)
  size_t merge_planes_bytes = 0;
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
  int intr_bin_sets = 0;                  // buffer sets allocated: 2 when the sweep runs in slices (one is reduced while the other fills)
  int intr_bin_rows = 0;                  // rows of counts (one per slice) the last call left in intr_bin_counts_host
  int intr_slices_forced = 0;             // bahip_debug_set_intrinsics_slices: > 0 fixes the number of slices of the sweep
  hipStream_t intr_aux_stream = nullptr;  // the reductions of a sliced sweep run here
  hipEvent_t intr_events[4] = {nullptr, nullptr, nullptr, nullptr};
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
  int arithmetic = 0;              // BAHIP_ARITHMETIC_EXACT / _FAST: flavour of the sweeps (bahip_context_set_arithmetic), mirrored in in.fast_math
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

namespace bahip_capi {

// ---- defined in capi.hip --------------------------------------------------------------------------------------------------------
Intrinsics make_intrinsics(const bahip_camera& cc, const bahip_camera& dc, const bahip_depth_params& dp);
void fill_pose(KfEntry* e, const float* global_T_frame);
int planes_alloc(int width, int height, int cwidth, int cheight, bahip_frame_planes** out);
void planes_free(bahip_frame_planes* p);
KfEntry raw_entry(const bahip_frame& f);
int make_entry(bahip_context* ctx, const bahip_frame& f, size_t slot, KfEntry* out);
SurfelsView make_view(const bahip_surfels* s);
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
int ensure_work(bahip_context* ctx, int n);
int ensure_px(bahip_context* ctx, size_t px, size_t scan_n);
inline bool timer_on(const bahip_context* ctx, int stage) { return ctx->profiling && (ctx->profiling != 3 || stage == 2); }
void timer_begin(bahip_context* ctx, int stage, bool first, int units = 1);
void timer_end(bahip_context* ctx, int stage);
inline bool kf_sharded(const bahip_context* ctx) { return ctx->kf_world > 1; }
int ensure_tile_bounds(bahip_context* ctx, uint32_t surfels);
extern int g_tile_order_enabled;
int ensure_tile_schedule(bahip_context* ctx, uint32_t padded_tiles);
const uint32_t* tile_order_for(const bahip_context* ctx, uint32_t surfels);

// ---- defined in capi_ba.hip -----------------------------------------------------------------------------------------------------
int wait_for_pose_sequence(bahip_context* ctx, PoseWork* host_work, const PoseWork* dev_work, int num_work, int sequence);
extern int g_fused_iteration_begin, g_pose_rounds_ahead, g_device_loop_enabled;
int run_pose_rounds(bahip_context* ctx, bool use_depth, bool use_desc, const KfEntry* dev_frames, KfEntry* dev_frames_rw,
                    PoseWork* dev_work, HbFixed* dev_Hb, int num_work, const SurfelsView& s, int write_back, int update_activation,
                    PoseWork* host_work /* page-locked, num_work + kPoseTailRecords records */, int* rounds_out,
                    bool schedule = false /* a phase over the keyframe table: its first round counts the candidates per tile and the
                    run order of the following sweeps is rebuilt from them */, int* rounds_hint = nullptr,
                    int first_round = 0, int first_iterating = -1 /* continue a phase whose rounds [0, first_round) have run (the
                    device-driven loop hands over a phase that needs more rounds than it had queued) */,
                    const PoseLoopControl* loop_stats = nullptr /* keeps the loop's totals going (never ends a phase) */);
int geometry_keyframe_sharded(bahip_context* ctx, bool use_depth, bool use_desc, const SurfelsView& v, long long activate_count);

// ---- defined in capi_rccl.hip ---------------------------------------------------------------------------------------------------
int reduce_over_ranks(bahip_context* ctx, void* buffer, size_t count, int dtype);
void rccl_destroy_communicator(bahip_context* ctx);   // (context destruction)
inline bool is_sharded(const bahip_context* ctx) { return ctx->allreduce != nullptr || ctx->rccl_comm != nullptr; }
inline bool kf_owned(const bahip_context* ctx, int k) { return (k & (ctx->kf_world - 1)) == ctx->kf_rank; }
#define REQUIRE_NO_KF_SHARDING(what) \
  REQUIRE(!kf_sharded(ctx), what " is not available under keyframe sharding (its per-surfel sums run over all keyframes in order): use surfel sharding")

}  // namespace bahip_capi
