// ba_launch.h -- host-callable launch wrappers shared between the kernel translation units and
// the C-ABI layer (capi*.hip; capi_internal.h lists the units).
#pragma once

#include "ba_device.h"
#include <cstdlib>

// An integer switch from the environment, read once at load time.  A plain function on purpose: these switches used to be initialised by
// lambdas at namespace scope, and hipcc gave two such lambdas in two openings of one namespace of one file the SAME mangled name (host
// and device passes number them per context) -- the linker kept one body, and `g_device_loop_enabled` was initialised by the lambda of
// BAHIP_FUSED_ITERATION_BEGIN (default 0): the device-driven loop silently declined every call (round 5, found in the kernel trace).
inline int bahip_env_int(const char* name, int fallback) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : fallback;
}

// ---- Two arithmetic flavours of the sweeps ----------------------------------------------------------------------------------------
// The sweeps over (surfel, keyframe) pairs -- activation, normals / geometry step, pose normal equations, intrinsics sweep, the PCG
// init and step-1 sweeps -- exist twice in the library, compiled from the same source:
//   exact  IEEE reciprocal / square root / division in few instructions (ba_device.h: rcp_exact, sqrt_exact), defined exp / sin / cos,
//          no contraction beyond the spelled fused multiply-adds: every bit is the oracle's (the checker's flavour, and the default);
//   fast   the hardware's v_rcp_f32 / v_sqrt_f32 / v_exp_f32 (<= 1 ulp), contraction at the compiler's discretion, denormals flushed
//          -- the arithmetic the reference itself ships with (B/../CMakeLists.txt:74: nvcc -use_fast_math), held to the reference's
//          kernels by tolerance (tests/test_gpu_fast_flavour.py), selected per context by bahip_context_set_arithmetic.
// What does NOT change with the flavour: the order of every sum (fixed trees, fixed-point pose sums, exact PCG sums, binary64
// intrinsics accumulation), so the fast flavour is as deterministic and as shard-invariant as the exact one; the rejection order of the
// association tests; preprocessing, lifecycle, the pose solve and the PCG vector kernels (compiled once, exact).
// Mechanics: a kernel translation unit wraps its sweeps in BAHIP_FLAVOURED_BEGIN / _END (namespace bahip::exact or bahip::fast); it
// is compiled once per flavour (Makefile: *_fast.o with -DBAHIP_FAST_MATH and the device-side flags); everything else in the unit
// sits under #ifndef BAHIP_FAST_MATH in plain bahip::, including the dispatchers that carry the public names below and pick the
// flavour from Intrinsics::fast_math.
#ifdef BAHIP_FAST_MATH
#define BAHIP_FLAVOUR fast
#else
#define BAHIP_FLAVOUR exact
#endif
#define BAHIP_FLAVOURED_BEGIN namespace bahip { namespace BAHIP_FLAVOUR {
#define BAHIP_FLAVOURED_END } }

namespace bahip {

struct SupportingView {
  uint32_t* b[BAHIP_MERGE_BUFFER_COUNT];
  uint32_t pitch;
};

// kernels_preprocess.hip
int launch_bilateral_filter(hipStream_t stream, float sigma_xy, float sigma_value, float radius_factor, uint16_t max_depth,
                            float raw_to_float_depth, const uint16_t* in, uint32_t in_pitch, uint16_t* out, uint32_t out_pitch, int w, int h);
void launch_brightness(hipStream_t stream, const uint8_t* rgb, uint32_t rgb_pitch, uint8_t* rgba, uint32_t rgba_pitch, int w, int h);
void launch_normals_from_depth(hipStream_t stream, const Intrinsics& in, const uint16_t* in_depth, uint32_t in_pitch,
                               uint16_t* out_depth, uint32_t out_pitch, uint16_t* out_normals, uint32_t normals_pitch);
void launch_point_radii(hipStream_t stream, const Intrinsics& in, float raw_to_float_depth, const uint16_t* depth,
                        uint32_t depth_pitch, uint16_t* radius, uint32_t radius_pitch, uint16_t* out_depth, uint32_t out_pitch);
void launch_min_max_depth(hipStream_t stream, const uint16_t* depth, uint32_t depth_pitch, int w, int h,
                          float raw_to_float_depth, int* result);
void launch_pack_planes(hipStream_t stream, const KfEntry& frame, int width, int height, int cwidth, int cheight, uint32_t* geom,
                        uint32_t* lumafp);

void launch_read_pattern(hipStream_t stream, const uint32_t* data, size_t words, int pattern, uint32_t* sink);

// kernels_surfel.hip
void launch_activation(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                       uint32_t surfels_size);
void launch_assign_colors(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s);
void launch_normals(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s);
void launch_geometry(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs,
                     int num_kfs, const SurfelsView& s, long long activate_count = -1,
                     const uint32_t* sched = nullptr /* heavy work first (wave_cull.h: scheduled_tile) */,
                     const int* stop = nullptr /* device word: non-zero = the launch does nothing (device-driven BA loop) */);

// keyframe sharding (kernels_surfel.hip: geometry_step, kPhase): one phase of the geometry step over this rank's keyframe classes
int geometry_normals_sums(bool activate);     // sums per class and surfel of the normals pass (cpn) ...
int geometry_position_sums(bool use_desc);    // ... and of the position pass (cpp)
void launch_geometry_phase(hipStream_t stream, int phase, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                           const SurfelsView& s, long long activate_count, const ClassPartials& cpn, const ClassPartials& cpp);
void launch_activation_hits(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s, uint32_t surfels_size,
                            int kf_rank, int kf_world, uint32_t* hits);
void launch_activation_from_hits(hipStream_t stream, const SurfelsView& s, uint32_t surfels_size, const uint32_t* hits);

void launch_count_pairs(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                        unsigned long long* counts);

// kernels_pose.hip
size_t pose_tile_bounds_bytes(uint32_t surfels);   // size of the per-tile bounding-sphere buffer launch_pose_accumulate needs
void launch_pose_accumulate(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* frames,
                            const void* work, int num_work, const SurfelsView& s, HbFixed* Hb, void* tile_bounds, bool stored_bounds,
                            int num_listed /* stored_bounds: entries of the list of work items still iterating */,
                            uint32_t* tile_counters /* 16 words, zero before the first launch; NULL: never the persistent LDS form */,
                            int* parity_inout /* which half of tile_counters the next persistent launch draws from */,
                            uint32_t* tile_cost = nullptr /* one word per tile, zero before a first round: += the candidates the tile visits */,
                            const uint32_t* sched = nullptr /* heavy work first (wave_cull.h: scheduled_tile) */,
                            const int* listed_count = nullptr /* device word holding num_listed (a round queued before the host
                            knows it; num_listed is then an upper bound and sizes the LDS table); 0 there: the launch does nothing */,
                            const int* stop = nullptr /* device word: non-zero = the launch does nothing (device-driven BA loop) */);
// May a later round over at most num_items work items be queued ahead (launch_pose_accumulate with listed_count)?
bool pose_round_can_be_queued_ahead(uint32_t surfels, int num_items, bool have_tile_counters);
uint32_t pose_padded_tiles(uint32_t surfels);   // tiles of the (padded) grid the sweeps run over: the length of tile_cost
size_t tile_schedule_words(uint32_t padded_tiles);   // words of a schedule for such a grid
// sched := heavy tiles + runs by descending cost (wave_cull.h; clears tile_cost); false if there are more runs than the kernel
// handles (sched untouched)
bool launch_tile_order(hipStream_t stream, uint32_t* tile_cost, uint32_t padded_tiles, uint32_t* sched);
// The device-driven BA loop (capi_ba.hip: bahip_alternating_iterations): the control words the last solve launch of a pose phase
// updates and every launch of the loop looks at (ba_device.h: kLoop*), and what that solve launch needs to decide.
struct PoseLoopControl {
  int* ctl = nullptr;          // device, kLoopWords ints; ctl + kLoopStop is the `stop` word of the other launches
  int* host_ctl = nullptr;     // mapped host copy, written by the publishing workgroup before the sequence number
  int phase_end = 0;           // this launch is the last queued round of its phase: it decides
  int iteration = 0;           // index of the BA iteration the phase belongs to (for min_iterations)
  int min_iterations = 0;
  int* round_log = nullptr;    // mapped host memory: [log_slot] = work items that iterated in this round (for the stage timers)
  int log_slot = 0;
  // 0: this launch writes nothing to the host copies (counters, control words, sequence number) and needs no system-scope fence -- the
  // launches of a queued loop that the host does not wait for (round 6: every solve launch used to publish, ~5 us of posted PCIe writes
  // and two system-scope fences each, 2.7 launches per iteration); the launch whose sequence number the host polls publishes for all
  int publish = 1;
  // >= 0 (with phase_end, at most 1024 work items): when the phase is complete and the loop goes on, this launch also runs the top
  // of the next iteration -- mode 1: activation window + propagation, 2: propagation -- and sets up its work items
  // (kernels_pose.hip: iteration_begin_body); the caller then queues no launch_iteration_begin for that iteration
  int next_mode = -1;
  const uint8_t* in_window = nullptr;
  const int* covis_offsets = nullptr;
  const int* covis_indices = nullptr;
};
void launch_pose_solve(hipStream_t stream, void* work, int num_work, HbFixed* Hb, KfEntry* frames, int write_back,
                       int update_activation, int round, void* host_out,
                       int sequence /* published to the host copy of the counters when the launch is complete */,
                       const PoseLoopControl* loop = nullptr);
void launch_pose_init_from_keyframes(hipStream_t stream, const KfEntry* frames, int num_kfs, void* work, HbFixed* Hb, void* host_out,
                                     int kf_rank = 0, int kf_world = 1 /* keyframe sharding: the sweep skips keyframes of other ranks */,
                                     const int* stop = nullptr);

void launch_window_activation(hipStream_t stream, KfEntry* frames, int num_kfs, const uint8_t* in_window, const int* offsets,
                              const int* indices, const int* stop = nullptr);   // window activation + co-visible propagation
void launch_propagate_covisible(hipStream_t stream, KfEntry* frames, int num_kfs, const int* offsets, const int* indices,
                                const int* stop = nullptr);
// window (mode 1) / propagation (mode 2) / nothing (mode 0), then the work items of the pose phase, in one launch; false (and
// nothing launched) beyond 1024 keyframes
bool launch_iteration_begin(hipStream_t stream, KfEntry* frames, int num_kfs, int mode, const uint8_t* in_window, const int* offsets, const int* indices,
                            void* work, HbFixed* Hb, void* host_out, const int* stop);

void set_tile_waves(int waves);   // 0 = automatic; 1 | 4 wavefronts per surfel tile in the normals / geometry passes; 5 = the hybrid shape of the geometry step
long long geometry_hybrid_launches();   // launches of the geometry step in the hybrid shape so far (both flavours)
void set_pose_lds_waves(int waves);        // test hook: wavefronts per workgroup of the LDS form (0: 16)
void set_pose_lds_parts_shift(int shift);  // test hook: 2^shift wavefronts share a tile's work items in the LDS form (-1: from the grid size)
void set_pose_lds_items(int items);   // test hook: slices of that many work items per launch of the LDS form (0: as many as the table holds)
void pcg_step1_form_launches(long long out[2]);   // launches of the PCG step-1 sweep: [0] one tile per wavefront, [1] persistent LDS form
long long pose_kernel_dispatches();   // kernel dispatches of the accumulate sweep since the process started
void pose_form_launches(long long out[2], bool reset);   // launches of either form since the last reset (bench: which kernel to name)
void set_pose_form(int form);     // 0 = automatic; 1 = one tile per wavefront + global atomics; 2 = persistent workgroups with the normal equations in LDS
void set_pose_parts(int parts);   // 0 = automatic; 1 | 2 | 4 | 8 wavefronts share a tile's keyframes in the pose kernel
void launch_jacobian_debug(hipStream_t stream, int kind, const float* in, float* out);
void launch_pose_step_debug(hipStream_t stream, const float* in, float* out);
void launch_exact_math_debug(hipStream_t stream, int kind, const float* in, float* out, size_t n);
void launch_pose_limbs_debug(hipStream_t stream, const float* in, long long* out, size_t n);
void launch_wave_reduce_debug(hipStream_t stream, const float* in, float* out);
void launch_evaluate_pairs(hipStream_t stream, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s,
                           const uint32_t* indices, int count, float* out);

// kernels_intrinsics.hip
// Intrinsics step (kernels_intrinsics.hip).  glob_d / cells_d: binary64 accumulators (34 sums; S records of 8: B0..B4, D, b2,
// observation count) -- what a multi-GPU run sums over the ranks; glob_f / cells_f: their binary32 roundings after the
// Schur complement (glob_f also carries x1 at [40..44] for the back-substitution).
// Append buffers of the intrinsics sweep's per-cell records (kernels_intrinsics.hip): several per block of 32 x 32 sparse cells, each
// 8 planes of `capacity` words; cursors[buffer] counts the records appended (and keeps counting when the buffer is full).
// capacity == 0: no binning, every record goes out as atomics.
struct IntrBins {
  uint32_t* cursors;
  uint32_t* records;
  uint32_t capacity;
  int bins_x;
};
int intrinsics_bin_count(const Intrinsics& in, int* bins_x_out);
size_t intrinsics_bin_record_bytes();
void launch_intrinsics_accumulate(hipStream_t st, bool depth, bool color, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                                  const SurfelsView& s, double* glob, double* cells, const IntrBins& bins, const uint32_t* sched,
                                  uint32_t position_begin = 0, uint32_t position_count = 0 /* 0: all positions of the schedule */);
uint32_t intrinsics_sweep_positions(uint32_t surfels, const uint32_t* sched);   // positions of a full sweep (what the slices cut)
void launch_intrinsics_bin_reduce(hipStream_t st, bool depth, const Intrinsics& in, const SurfelsView& s, double* cells, const IntrBins& bins);
void set_intrinsics_reduce_form(int form);   // 0: LDS table of ds_add_f64, 1: records sorted by cell in LDS, sums in registers
size_t intrinsics_schur_partials(int S);   // floats of scratch launch_intrinsics_finish needs
void launch_intrinsics_finish(hipStream_t st, bool schur, int S, const double* glob_d, const double* cells_d, float* glob_f, float* cells_f,
                              float* partials);
void launch_intrinsics_solve_cells(hipStream_t st, const Intrinsics& in, int S, float* cells_f, const float* x1, float* cfactor,
                                   uint32_t cfactor_pitch);

#ifdef BAHIP_TILE_TIMELINE
void geometry_timeline_dump(const char* path);   // experiment build only (kernels_surfel.hip, kernels_pose.hip; scripts/tile_timeline.py)
void pose_timeline_dump(const char* path);
#endif
#ifdef BAHIP_COUNT_CANDIDATES
void pose_counters_dump();   // experiment build only (kernels_pose.hip)
#endif
// kernels_pcg.hip
// `ex`: the exact accumulators of the solve (pcg_exact_cells(head_count) cells of 72 bytes, zeroed once; every kernel that
// resolves a sum clears what it read).  ctl: device-side inner-loop control block (pcg_control_bytes(); stopping rule of
// B/direct_ba_pcg.cc:427-456 evaluated on the device).
size_t pcg_exact_cells(uint32_t head_count);
PcgExact pcg_exact_view(void* buffer, uint32_t head_count);
size_t pcg_control_bytes();
void launch_pcg_init(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                     const SurfelsView& s, float* r, float* M,
                     uint32_t* tile_cost = nullptr /* census for the schedule, as in launch_pose_accumulate */, const uint32_t* sched = nullptr);
void launch_pcg_resolve_init(hipStream_t st, const PcgLayout& L, const PcgExact& ex, float* r, float* M);
void launch_pcg_init2(hipStream_t st, const PcgLayout& L, const PcgExact& ex, float a, const float* r, const float* M, float* delta, float* g,
                      float* p);
void launch_pcg_control_init(hipStream_t st, const PcgExact& ex, void* ctl, float* alpha_n);
void launch_pcg_step1(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* kfs, int num_kfs,
                      const SurfelsView& s, const float* p, float* g, const void* ctl, const uint32_t* sched = nullptr,
                      uint32_t* tile_counters = nullptr /* the context's two sets of eight tile counters: allows the persistent LDS form */,
                      int* parity_inout = nullptr);
void set_pcg_lds_form(int mode);   // test hook: 0 = always the one-tile-per-wavefront form of the step-1 sweep, 1 = automatic, 2 = the LDS form whenever the table fits
void launch_pcg_eps_terms(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const float* p);
void launch_pcg_resolve_step1(hipStream_t st, const PcgLayout& L, const PcgExact& ex, float* g, float* alpha_d, double eps_repeat, const void* ctl);
void launch_pcg_step2(hipStream_t st, const PcgLayout& L, const PcgExact& ex, float* r, const float* M, float* delta, float* g, const float* p,
                      const float* alpha_n, const float* alpha_d, const void* ctl);
void launch_pcg_control(hipStream_t st, const PcgExact& ex, void* ctl, float* beta_n);
void launch_pcg_step3(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const float* g, float* p, const float* alpha_n, const float* beta_n,
                      const void* ctl);
void launch_exact_sum_debug(hipStream_t st, const PcgExact& ex, const float* values, size_t n, int mode, double* out);
void launch_pcg_update_surfels(hipStream_t st, const PcgLayout& L, const SurfelsView& s, const float* delta);
void launch_pcg_update_cfactors(hipStream_t st, const Intrinsics& in, uint32_t start, const float* delta, float* cfactor, uint32_t pitch);

// kernels_lifecycle.hip
void launch_supporting_fill(hipStream_t st, const SupportingView& sup, int w, int h);
void launch_lifecycle_bounds(hipStream_t st, const SurfelsView& s, uint32_t tiles, void* spheres);   // bounding spheres of tiles [0, tiles)
// What a per-keyframe sweep of a lifecycle batch may skip (kernels_lifecycle.hip: LifecycleBounds): the bounding spheres of the
// tiles [0, tiles) taken at the start of the batch, and -- when the batch knows its frames -- the list of those tiles this frame sees.
struct LifecycleCull {
  const void* spheres = nullptr;
  uint32_t tiles = 0;
  const uint32_t* list = nullptr;
  uint32_t list_count = 0;
};
// cursors[f] (zeroed) <- the number of bounded tiles frame f can see, lists[f * tiles + ...] <- those tiles (no particular order)
void launch_lifecycle_visible_tiles(hipStream_t st, const Intrinsics& in, const float* frames_F, int num_frames, const void* spheres, uint32_t tiles,
                                    uint32_t* cursors, uint32_t* lists);
void launch_supporting_insert(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s, const SupportingView& sup,
                              const LifecycleCull& cull = LifecycleCull(),
                              const uint32_t* size_on_device = nullptr /* a creation batch: min(*size_on_device, s.size) surfels exist */);
void launch_merge(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s, const SupportingView& sup,
                  float cell_merge_dist_sq, float cos_thr, uint32_t* flags, uint32_t* cell_of, bool empty_the_planes, uint32_t* deleted_count,
                  const LifecycleCull& cull = LifecycleCull());
void launch_merge_decide(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s, const SupportingView& sup,
                         float cell_merge_dist_sq, float cos_thr, uint32_t* flags, uint32_t* cell_of, const LifecycleCull& cull);
// one launch: the apply sweep of apply_frame (NULL: none) beside the insert sweep of insert_frame (NULL: none); flags must have been
// cleared before the batch's first decide (kernels_lifecycle.hip: merge_apply_insert_kernel)
void launch_merge_apply_insert(hipStream_t st, const Intrinsics& in, const KfEntry* apply_frame, const KfEntry* insert_frame, const SurfelsView& s,
                               const uint32_t* flags, const uint32_t* cell_of, const SupportingView& apply_sup, const SupportingView& insert_sup,
                               uint32_t* deleted_count, const LifecycleCull& apply_cull, const LifecycleCull& insert_cull);
// a merge batch by cell lists (kernels_lifecycle.hip: merge_pairs_kernel): the associated (surfel, frame) pairs grouped by (frame, cell)
// up front -- counts and offsets [num_frames * cells + 1] words; pair_cells and pair_ranks [64 * total sweep positions] words; members
// (words) and member_cell (uint2) [pairs <= 64 * total sweep positions] --, then one launch per frame over its pairs
// (frame_first[j] = offsets[j * cells], read back by the caller) and one launch that writes the deleted markers.  deleted_at: one word per surfel, ~0 before.
struct MergeBatchFrame {
  KfEntry entry;
  uint32_t list_offset, list_count;   // its visible bounded tiles in `lists`
  uint32_t pair_offset;               // its first sweep position (wavefront) in pair_cells / pair_ranks
  uint32_t pad_;
};
hipError_t launch_merge_batch_lists(hipStream_t st, const Intrinsics& in, const MergeBatchFrame* frames, int num_frames, uint32_t max_positions, const SurfelsView& s,
                                    const uint32_t* lists, uint32_t bounded_tiles, uint32_t* counts, uint32_t* offsets, uint32_t* pair_cells, uint32_t* pair_ranks,
                                    uint32_t* members, void* member_cell, uint32_t* frame_first /* [num_frames + 1] */, void* scan_temp, size_t scan_temp_bytes);
size_t merge_batch_scan_temp_bytes(size_t entries);
void launch_merge_pairs(hipStream_t st, const SurfelsView& s, const uint32_t* members, const void* member_cell, uint32_t first_pair,
                        uint32_t end_pair, uint32_t step, uint32_t* deleted_at, float cell_merge_dist_sq, float cos_thr);
void launch_merge_batch_apply(hipStream_t st, const SurfelsView& s, const uint32_t* deleted_at, uint32_t* deleted_count);   // writes the markers, adds their number
// a creation batch: scan + append at *size_in + the new size into *size_out (or *capacity_exceeded raised and nothing appended) in one
// launch; group_words: create_append_groups() words, cleared before tag 1 and whenever a tag (1 .. 255) would repeat
int create_append_groups();
void set_append_groups_limit(int groups);   // test hook: the fused append's grid as on a device that holds at most `groups` of its workgroups (0: ask the device)
void launch_create_append_fused(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const uint8_t* flags, const SurfelsView& s,
                                const uint32_t* size_in, uint32_t* size_out, uint32_t capacity, uint32_t* capacity_exceeded,
                                uint32_t* group_words, uint32_t tag);
size_t create_padded_count(const Intrinsics& in);   // length of the tile-major flag / index vectors
// a creation batch whose keyframes do not wait for each other's sweeps (kernels_lifecycle.hip: create_chain_kernel).  Up front, for all
// keyframes of the batch: occupancy [n][cells] bytes (cleared by the caller) <- the cloud at the batch's begin, candidates [n][padded]
// bytes (cleared by the caller) <- the pixels that would create a surfel, filtered; then one launch per keyframe of the chain.
struct CreateBatchItem {
  int kf_index;                       // bound keyframe
  uint32_t list_offset, list_count;   // its visible bounded tiles in `lists`
  int covis_offset, n_covis;          // its slice of the batch's co-visibility lists
};
// scan: [num_items * padded] words; cand_cell: one word, records: kSurfelAccum0 rows (a view whose pitch is the row length), per list
// position -- at most one candidate per sparse cell and keyframe; first_of_item: [num_items + 1] words (read back by the caller)
size_t create_batch_scan_temp_bytes(size_t entries);
hipError_t launch_create_batch_prepare(hipStream_t st, const Intrinsics& in, const KfEntry* kfs, const CreateBatchItem* items, int num_items, uint32_t max_list_count,
                                       const SurfelsView& cloud_at_begin, const uint32_t* lists, uint32_t bounded_tiles, uint8_t* occupancy, uint8_t* candidates,
                                       bool filter_new_surfels, const int* covis, const float* covis_T_frame, int min_obs, uint32_t* scan, void* scan_temp,
                                       size_t scan_temp_bytes, uint32_t* cand_cell, const SurfelsView& records, uint32_t* first_of_item);
void launch_create_chain(hipStream_t st, const Intrinsics& in, const KfEntry* next_frame, const uint32_t* cand_cell, const SurfelsView& records, uint32_t first,
                         uint32_t end_of_frame, const uint8_t* occupancy, uint8_t* next_occupancy, const SurfelsView& s, uint32_t batch_begin_size,
                         const uint32_t* size_in, uint32_t* size_out, uint32_t capacity, uint32_t* capacity_exceeded, uint32_t* group_words, uint32_t tag,
                         uint32_t appended_bound);
void launch_create_flag(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const SupportingView& sup, uint8_t* flags, bool leave_planes_empty = false);
void launch_create_filter(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const KfEntry* kfs, const int* covis,
                          const float* covis_T_frame, int n_covis, int min_obs, uint8_t* flags);
void launch_create_append(hipStream_t st, const Intrinsics& in, const KfEntry& frame, const uint8_t* flags,
                          const uint32_t* indices, uint32_t surfels_size, const SurfelsView& s);
void launch_delete_update(hipStream_t st, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,
                          int min_obs, uint32_t* deleted_count);
void launch_shard_to_cloud(hipStream_t st, const SurfelsView& shard, const SurfelsView& cloud, uint32_t rank, uint32_t world, uint32_t chunk);
void launch_cloud_to_shard(hipStream_t st, const SurfelsView& cloud, const SurfelsView& shard, uint32_t rank, uint32_t world, uint32_t chunk);
size_t sort_scratch_bytes(uint32_t n);
hipError_t sort_surfels_spatially(hipStream_t st, const SurfelsView& s, float inv_cell, void* scratch, size_t scratch_bytes);   // stream-ordered, no host wait
size_t scan_temp_bytes(size_t n);
hipError_t scan_flags_inclusive(hipStream_t st, void* temp, size_t temp_bytes, const uint8_t* flags, uint32_t* out, int n);
hipError_t scan_u32_exclusive(hipStream_t st, void* temp, size_t temp_bytes, const uint32_t* in, uint32_t* out, int n);
hipError_t launch_compact(hipStream_t st, const SurfelsView& s, uint32_t* invalid, uint32_t* free_rank, uint32_t* free_list,
                          uint32_t surfel_count, void* temp, size_t temp_bytes);


// The flavoured launchers (same parameters as the public names above, no defaults: only the dispatchers call them) and the test
// hooks / counters that live with them, declared in both flavour namespaces.
#define BAHIP_FLAVOURED_DECLARATIONS                                                                                                          \
  /* kernels_surfel.hip */                                                                                                                    \
  void launch_activation(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s, uint32_t surfels_size); \
  void launch_normals(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s);                       \
  void launch_geometry(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs, int num_kfs,              \
                       const SurfelsView& s, long long activate_count, const uint32_t* sched, const int* stop);                              \
  void launch_geometry_phase(hipStream_t stream, int phase, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* kfs,          \
                             int num_kfs, const SurfelsView& s, long long activate_count, const ClassPartials& cpn, const ClassPartials& cpp); \
  void launch_activation_hits(hipStream_t stream, const Intrinsics& in, const KfEntry* kfs, int num_kfs, const SurfelsView& s,                \
                              uint32_t surfels_size, int kf_rank, int kf_world, uint32_t* hits);                                              \
  void set_tile_waves(int waves);                                                                                                             \
  long long geometry_hybrid_launches();                                                                                                       \
  /* kernels_pose.hip */                                                                                                                      \
  void launch_pose_accumulate(hipStream_t stream, bool use_depth, bool use_desc, const Intrinsics& in, const KfEntry* frames,                 \
                              const void* work, int num_work, const SurfelsView& s, HbFixed* Hb, void* tile_bounds, bool stored_bounds,       \
                              int num_listed, uint32_t* tile_counters, int* parity_inout, uint32_t* tile_cost, const uint32_t* sched,         \
                              const int* listed_count, const int* stop);                                                                      \
  bool pose_round_can_be_queued_ahead(uint32_t surfels, int num_items, bool have_tile_counters);                                              \
  void set_pose_lds_waves(int waves);                                                                                                         \
  void set_pose_lds_parts_shift(int shift);                                                                                                   \
  void set_pose_lds_items(int items);                                                                                                         \
  long long pose_kernel_dispatches();                                                                                                         \
  void pose_form_launches(long long out[2], bool reset);                                                                                      \
  void set_pose_form(int form);                                                                                                               \
  void set_pose_parts(int parts);                                                                                                             \
  void launch_evaluate_pairs(hipStream_t stream, const Intrinsics& in, const KfEntry& frame, const SurfelsView& s, const uint32_t* indices,   \
                             int count, float* out);                                                                                          \
  /* kernels_intrinsics.hip */                                                                                                                \
  void launch_intrinsics_accumulate(hipStream_t st, bool depth, bool color, const Intrinsics& in, const KfEntry* kfs, int num_kfs,            \
                                    const SurfelsView& s, double* glob, double* cells, const IntrBins& bins, const uint32_t* sched,           \
                                    uint32_t position_begin, uint32_t position_count);                                                        \
  /* kernels_pcg.hip */                                                                                                                       \
  void launch_pcg_init(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* kfs, int num_kfs,         \
                       const SurfelsView& s, float* r, float* M, uint32_t* tile_cost, const uint32_t* sched);                                 \
  void launch_pcg_step1(hipStream_t st, const PcgLayout& L, const PcgExact& ex, const Intrinsics& in, const KfEntry* kfs, int num_kfs,        \
                        const SurfelsView& s, const float* p, float* g, const void* ctl, const uint32_t* sched, uint32_t* tile_counters,      \
                        int* parity_inout);                                                                                                   \
  void set_pcg_lds_form(int mode);                                                                                                            \
  void pcg_step1_form_launches(long long out[2]);
namespace exact { BAHIP_FLAVOURED_DECLARATIONS }
namespace fast { BAHIP_FLAVOURED_DECLARATIONS }
#define BAHIP_PICK(in, call) do { if ((in).fast_math) fast::call; else exact::call; } while (0)

}  // namespace bahip
