/*
 * badslam_hip.h -- C ABI of the MI355X-native direct bundle-adjustment backend.
 *
 * This is the drop-in boundary for the BA hot path of ETH3D/badslam: every entry point below
 * replaces one of the free functions that the reference's DirectBA class calls
 * (applications/badslam/src/badslam/kernels.h:94-495, abbreviated B/kernels.h) or one of its
 * keyframe preprocessing calls (B/keyframe.cc:96-144).  Plain pointers and sizes only, no C++
 * types.  All device pointers are BORROWED: the caller owns every allocation (the reference's
 * CUDABuffer objects own theirs, B/direct_ba.h:441-507); the library only owns the scratch held
 * inside a bahip_context.
 *
 * Conventions
 *   - return value 0 = success; non-zero = failure, message via bahip_last_error() (the C++ shim
 *     turns non-zero into LOG(FATAL), matching the reference's CUDA_CHECK() semantics,
 *     libvis/src/libvis/cuda/cuda_util.h:35-49).
 *   - "stream" is the hipStream_t a bahip_context was created on; every call is asynchronous on
 *     that stream unless documented as synchronising (those that return host values).
 *   - images are pitched row-major: element (y, x) of type T at (char*)data + y*pitch_bytes + x*sizeof(T),
 *     exactly libvis CUDABuffer_<T> (libvis/src/libvis/cuda/cuda_buffer.cuh:44-119).
 *   - the surfel buffer is the reference's 17-row SoA (B/kernels.cuh:69-93): row r, surfel i at
 *     (float*)((char*)data + r*pitch_bytes) + i.
 *   - cameras are PinholeCamera4f parameter vectors in the pixel-corner convention.
 *   - poses are SE3f stored like Sophus: unit quaternion (x, y, z, w) then translation = 7 floats.
 */
#ifndef BADSLAM_HIP_H_
#define BADSLAM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BAHIP_SURFEL_ATTRIBUTE_COUNT 17   /* B/kernels.cuh:93 */
#define BAHIP_MERGE_BUFFER_COUNT 3        /* B/kernels.cuh:51 */
#define BAHIP_MAX_POSE_ITERATIONS 30      /* B/direct_ba_alternating.cc:130 */

enum { BAHIP_KF_ACTIVE = 0, BAHIP_KF_COVISIBLE_ACTIVE = 1, BAHIP_KF_INACTIVE = 2 };  /* B/keyframe.h:54-67 */

typedef struct bahip_camera {       /* libvis PinholeCamera4f */
  float fx, fy, cx, cy;
  int32_t width, height;
} bahip_camera;

typedef struct bahip_depth_params { /* B/surfel_projection.cuh:129-149 */
  float a;
  float raw_to_float_depth;
  float baseline_fx;
  int32_t sparse_surfel_cell_size;
  float* cfactor;                   /* device, ((H-1)/cell+1) x ((W-1)/cell+1) floats */
  uint32_t cfactor_pitch_bytes;
  int32_t cfactor_width, cfactor_height;
} bahip_depth_params;

/* Tiled copies ("BA planes") of a frame's depth+normals and of its luma, the layout the surfel sweeps
 * read (DESIGN.md "HBM layout").  Opaque; owned by whoever owns the images (vis::Keyframe does). */
typedef struct bahip_frame_planes bahip_frame_planes;

typedef struct bahip_frame {        /* the four images a Keyframe owns, B/keyframe.h:227-231 */
  uint16_t* depth;   uint32_t depth_pitch_bytes;
  uint16_t* normals; uint32_t normals_pitch_bytes;
  uint16_t* radius;  uint32_t radius_pitch_bytes;
  uint8_t* color;    uint32_t color_pitch_bytes;   /* uchar4 (R, G, B, luma) */
  /* Optional.  NULL: the library packs the planes itself on every call that receives this frame (correct, but a
   * K-image repack per bahip_set_keyframes).  Non-NULL: must have been refreshed with bahip_frame_planes_update
   * after the last change of depth / normals / color. */
  const bahip_frame_planes* planes;
} bahip_frame;

typedef struct bahip_keyframe {     /* one entry of the keyframe list handed to the *_CUDA functions */
  bahip_frame frame;
  float global_T_frame[7];          /* qx qy qz qw tx ty tz */
  int32_t activation;               /* BAHIP_KF_*; a deleted (null) keyframe is simply left out */
} bahip_keyframe;

typedef struct bahip_surfels {      /* B/direct_ba.cc:122-123 */
  float* data;
  uint32_t pitch_bytes;
  uint8_t* active;                  /* may be NULL where the reference passes none */
  uint32_t surfels_size;
  uint32_t capacity;
} bahip_surfels;

typedef struct bahip_context bahip_context;

/* ---- library / context ---------------------------------------------------------------------- */
const char* bahip_last_error(void);
int bahip_device_count(void);
/* hip_stream: a hipStream_t (NULL = the legacy default stream). */
int bahip_context_create(bahip_context** out, void* hip_stream);
void bahip_context_destroy(bahip_context* ctx);
int bahip_context_synchronize(bahip_context* ctx);
/* Multi-GPU surfel sharding (one process per GPU, DESIGN.md "Multi-GPU"): the per-keyframe pose normal equations, the
 * intrinsics accumulators and the dense head of the PCG vectors are summed element-wise over all ranks, in place.
 *
 * Native path: bahip_context_init_rccl creates an RCCL communicator for this context (rank 0 obtains the 128-byte id with
 * bahip_rccl_get_unique_id and distributes it by any means, e.g. a torch.distributed broadcast) and every sum becomes one
 * ncclAllReduce on the context's stream: no host round trip, ordered with the kernels around it.  librccl.so is loaded
 * on first use (dlopen), so a single-GPU process never touches it.
 *
 * Hook path: any other transport (a hook, when installed, takes precedence over the communicator).  The hook receives the stream the producers of `device_buffer` were queued on and must
 * order the reduction after them and before later work on that stream (e.g. torch.cuda.ExternalStream(stream) around
 * dist.all_reduce).  dtype: BAHIP_SUM_F32 (count floats), BAHIP_SUM_I64 (count int64_t: the pose normal equations are
 * summed in fixed point, which makes a sharded run bit-identical to the unsharded one) or BAHIP_SUM_F64 (count doubles: the
 * accumulators of the intrinsics step).  NULL (default) = single GPU. */
/* 1 if sums go over several ranks (a hook or an RCCL communicator is installed), else 0. */
int bahip_context_is_sharded(bahip_context* ctx);
enum { BAHIP_SUM_F32 = 0, BAHIP_SUM_I64 = 1, BAHIP_SUM_F64 = 2 };
typedef int (*bahip_allreduce_fn)(void* device_buffer, size_t count, int dtype, void* hip_stream, void* user);
int bahip_context_set_allreduce(bahip_context* ctx, bahip_allreduce_fn fn, void* user);
/* Multi-GPU KEYFRAME sharding (BASELINE configs[3]: "sharded by keyframe, RCCL all-reduce of pose Hessians"; the loop it
 * partitions: B/kernel_opt_geometry.cc:108-200, B/kernel_opt_pose.cc:67-96, B/kernel_surfel_activation.cc:53-66).  Every rank
 * holds ALL surfels and the images of its own keyframes only: bound keyframe k lives on rank k % world, world = 1, 2, 4 or 8
 * (the per-surfel sums are defined as C interleaved partial sums over the keyframe classes k % C, kernels_surfel.hip, so whole
 * classes per rank reproduce the unsharded bits: C = 4 by default, which serves 2 and 4 ranks; 8 ranks -- BASELINE configs[3] as
 * written -- need C = 8, bahip_context_set_sum_classes, which is then also the definition the single-GPU run it is compared with
 * must use).  bahip_set_keyframes still
 * takes all keyframes -- poses and activation states are replicated -- but the image pointers of keyframes that live elsewhere
 * are not looked at (pass NULL).  With it
 *   - bahip_update_surfel_activation sums one hit word per surfel over the ranks (BAHIP_SUM_I64, N / 2 words);
 *   - bahip_optimize_geometry_iteration / bahip_update_activation_and_optimize_geometry run in three launches with two
 *     exchanges of the class partials (bit patterns as BAHIP_SUM_I64: 4 x 5 and 4 x 8 binary32 values per surfel);
 *   - bahip_estimate_keyframe_poses* sweep this rank's keyframes, sum the fixed-point normal equations over the ranks
 *     (K x 56 int64 per Gauss-Newton round, the exchange surfel sharding makes too) and solve every pose on every rank;
 *   - everything ends with the bits of the unsharded run on every rank.
 * The intrinsics step, the PCG scheme and the surfel lifecycle are refused in this mode (their per-surfel chains run over all
 * keyframes in order; use surfel sharding).  A hook or an RCCL communicator must be installed when world > 1.  Surfel and
 * keyframe sharding exclude each other on one context. */
int bahip_context_set_keyframe_sharding(bahip_context* ctx, int rank, int world);
/* The number of interleaved partial sums (keyframe classes) the per-surfel sums of the normals and geometry passes are DEFINED
 * over: 4 (default) or 8.  In exact arithmetic both are the reference's sum (B/kernel_opt_geometry.cu: keyframe after keyframe);
 * in binary32 they differ in the last bits like any reordering.  The oracle takes the same parameter (orc_set_sum_classes), and
 * both class counts are held against it bit for bit.  Takes effect at once (no re-binding needed). */
int bahip_context_set_sum_classes(bahip_context* ctx, int classes);
/* Order in which the surfels a keyframe creates are appended: 0 (default) = by tiles of 8 x 8 sparse cells of the creating
 * keyframe, row-major inside a tile -- the 64 surfels of a wavefront form a compact patch, which is what the sweeps' culling wants;
 * 1 = row-major over the whole image, the reference's order (B/kernel_create_surfels.cu:357-390).  The same surfels either way;
 * their indices differ inside each keyframe's block, and with them which of two mergeable surfels survives (the lower index:
 * B/kernel_supporting_surfels.cu:60-86).  Takes effect for the creations that follow. */
int bahip_context_set_creation_order(bahip_context* ctx, int row_major);
/* Arithmetic flavour of the sweeps over (surfel, keyframe) pairs -- activation, normals / geometry step, pose normal equations,
 * intrinsics sweep, PCG init and step-1 sweeps -- of this context:
 *   BAHIP_ARITHMETIC_EXACT (default)  correctly rounded reciprocal / square root / division, defined exp, no contraction beyond the
 *                                     spelled fused multiply-adds: every bit is the CPU oracle's (what the parity tests hold);
 *   BAHIP_ARITHMETIC_FAST             v_rcp_f32 / v_sqrt_f32 / v_exp_f32 (<= 1 ulp), contraction at the compiler's discretion, binary32
 *                                     denormals flushed -- the arithmetic of the reference's own build (nvcc -use_fast_math,
 *                                     applications/badslam/CMakeLists.txt:74); held to the reference's kernels by tolerance
 *                                     (tests/test_gpu_fast_flavour.py: poses within 1e-5 m, association flips <= 0.1 %).
 * The order of every sum is the same in both (fixed trees, fixed-point pose sums, exact PCG sums), so either flavour is
 * deterministic and invariant under surfel / keyframe sharding; preprocessing, the surfel lifecycle, the pose solve and the PCG
 * vector kernels exist once (exact).  Takes effect at once.  The environment variable BAHIP_ARITHMETIC=fast|exact sets the default
 * of contexts created afterwards (A/B runs of unmodified callers). */
enum { BAHIP_ARITHMETIC_EXACT = 0, BAHIP_ARITHMETIC_FAST = 1 };
int bahip_context_set_arithmetic(bahip_context* ctx, int arithmetic);
int bahip_context_get_arithmetic(bahip_context* ctx);
/* A hint, for callers that write the surfel buffer themselves (an upload, a permutation of their own): which surfels share a tile has
 * changed, so the run order the sweeps derive from a census of the tiles ("heavy work first", rebuilt on its own only every 32nd pose
 * phase or when the number of tiles changes) is stale -- the next pose phase takes the census again.  The backend's own movers
 * (bahip_compact_surfels, bahip_sort_surfels_spatially) do this themselves.  Results never depend on the run order. */
int bahip_context_surfels_rearranged(bahip_context* ctx);
#define BAHIP_RCCL_UNIQUE_ID_BYTES 128
int bahip_rccl_get_unique_id(char unique_id_out[BAHIP_RCCL_UNIQUE_ID_BYTES]);
int bahip_context_init_rccl(bahip_context* ctx, const char unique_id[BAHIP_RCCL_UNIQUE_ID_BYTES], int rank, int world_size);
/* The first exchange of a run, as a probe: every rank contributes 1 through the context's transport (hook or native RCCL) on the
 * context's stream; the host waits for the sum at most timeout_ms (<= 0: for ever) and returns the number of ranks that took
 * part.  A job whose first collective cannot complete fails here with a message instead of hanging inside the first BA iteration.
 * Every rank must call it at the same point.  1 rank / no transport: *ranks_out = 1. */
int bahip_context_count_ranks(bahip_context* ctx, int timeout_ms, int* ranks_out);

/* Device memory helpers (what libvis CUDABuffer does with cudaMallocPitch / cudaMemcpy2DAsync,
 * libvis/src/libvis/cuda/cuda_buffer_inl.h:36-186). */
int bahip_malloc_pitch(void** ptr, size_t* pitch_bytes, size_t width_bytes, size_t height);
int bahip_free(void* ptr);
int bahip_memcpy_2d(bahip_context* ctx, void* dst, size_t dst_pitch, const void* src, size_t src_pitch,
                    size_t width_bytes, size_t height, int kind /* 1 = H2D, 2 = D2H, 3 = D2D */);
int bahip_memset_2d(bahip_context* ctx, void* dst, size_t pitch, int value, size_t width_bytes, size_t height);

/* Stream-level helpers used by the host-side CUDABuffer<T> (libvis/src/libvis/cuda/cuda_buffer_inl.h:
 * UploadAsync / DownloadAsync / UploadPartAsync / DownloadPartAsync / Clear / SetTo all take the
 * stream per call).  kind: 1 = H2D, 2 = D2H, 3 = D2D.  All asynchronous on `hip_stream`. */
int bahip_context_set_stream(bahip_context* ctx, void* hip_stream);
int bahip_stream_create(void** hip_stream_out);
int bahip_stream_destroy(void* hip_stream);
int bahip_stream_synchronize(void* hip_stream);
int bahip_memcpy_2d_async(void* hip_stream, void* dst, size_t dst_pitch, const void* src, size_t src_pitch,
                          size_t width_bytes, size_t height, int kind);
int bahip_memcpy_async(void* hip_stream, void* dst, const void* src, size_t bytes, int kind);
int bahip_memset_async(void* hip_stream, void* dst, int value, size_t bytes);
/* Page-locked host memory, and the question CUDABuffer<T>::UploadAsync / DownloadAsync ask before they return: a transfer to or from
 * page-locked memory is left in flight on the stream, as cudaMemcpy2DAsync leaves it (libvis/src/libvis/cuda/cuda_buffer_inl.h:73-90);
 * for pageable memory the call waits, which is what the CUDA runtime does there too (the copy is staged, a download has completed
 * when the call returns).  bahip_host_is_pinned: 1 if both ends of [ptr, ptr + bytes) are page-locked host memory
 * (bahip_host_alloc, hipHostMalloc, hipHostRegister), else 0. */
int bahip_host_alloc(void** ptr, size_t bytes);
int bahip_host_free(void* ptr);
int bahip_host_is_pinned(const void* ptr, size_t bytes);
/* CUDABuffer<T>::Clear(value, stream) (libvis/src/libvis/cuda/cuda_buffer.cu:41-60) for 1/2/4-byte T. */
int bahip_fill_2d(void* hip_stream, void* data, size_t pitch_bytes, int elem_bytes, uint32_t value_bits, int width, int height);

/* ---- frame preprocessing (BadSlam::PreprocessFrame, B/bad_slam.cc:643-765) --------------------- */
/* B/cuda_depth_processing.cu:42-128 BilateralFilteringAndDepthCutoffCUDA: raw depth 0 or > max_depth -> 65535 (unknown),
 * else the bilateral filter in inverse depth over a disc of radius int(radius_factor * sigma_xy + 0.5) <= 8 pixels.
 * Not in place. */
int bahip_bilateral_filtering_and_depth_cutoff(bahip_context* ctx, float sigma_xy, float sigma_value, float radius_factor,
                                               uint16_t max_depth, float raw_to_float_depth, const uint16_t* in_depth,
                                               uint32_t in_pitch_bytes, uint16_t* out_depth, uint32_t out_pitch_bytes, int width,
                                               int height);

/* ---- keyframe preprocessing (Keyframe ctor #2, B/keyframe.cc:96-144) ------------------------- */
/* B/cuda_image_processing.cu:165-193 ComputeBrightnessCUDA: uchar3 RGB -> uchar4 (R,G,B,luma). */
int bahip_compute_brightness(bahip_context* ctx, const uint8_t* rgb, uint32_t rgb_pitch_bytes,
                             uint8_t* rgba, uint32_t rgba_pitch_bytes, int width, int height);
/* B/cuda_depth_processing.cu:134-287 ComputeNormalsCUDA. */
int bahip_compute_normals(bahip_context* ctx, const bahip_camera* depth_camera, const bahip_depth_params* dp,
                          const uint16_t* in_depth, uint32_t in_pitch, uint16_t* out_depth, uint32_t out_pitch,
                          uint16_t* out_normals, uint32_t normals_pitch);
/* B/cuda_depth_processing.cu:289-388 ComputePointRadiiAndRemoveIsolatedPixelsCUDA. */
int bahip_compute_point_radii_and_remove_isolated_pixels(
    bahip_context* ctx, const bahip_camera* depth_camera, float raw_to_float_depth,
    const uint16_t* depth, uint32_t depth_pitch, uint16_t* radius, uint32_t radius_pitch,
    uint16_t* out_depth, uint32_t out_pitch);
/* B/cuda_depth_processing.cu:391-465 ComputeMinMaxDepthCUDA (synchronises; results on host). */
int bahip_compute_min_max_depth(bahip_context* ctx, const uint16_t* depth, uint32_t depth_pitch, int width,
                                int height, float raw_to_float_depth, float* min_depth, float* max_depth);

/* ---- scene binding ----------------------------------------------------------------------------
 * The reference passes cameras, DepthParameters and the keyframe vector to every *_CUDA call.
 * Here they are bound once per BA call and kept in a device-side keyframe table so that one
 * launch can sweep all keyframes (the per-keyframe launch granularity of B/kernel_*.cc is what
 * this backend removes). */
int bahip_set_intrinsics(bahip_context* ctx, const bahip_camera* color_camera, const bahip_camera* depth_camera,
                         const bahip_depth_params* dp);
/* Maintenance, like compaction: reorders surfels [0, surfels_size) along a Morton curve over a world grid of
 * `grid_cell_size` metres (stable; deleted surfels last).  Moves the 8 data rows and the active flags.  Surfels that
 * one image region shows become neighbours in the buffer, which is what the sweeps' L2 behaviour wants;
 * no result depends on the order.  Not part of the reference: call it when convenient (after keyframes were added).
 * Round 6: ordered on the context's stream like a kernel launch -- no allocation and no host wait (a scratch block of the context
 * holds keys, indices and a dense copy of the rows; it grows when the cloud does). */
int bahip_sort_surfels_spatially(bahip_context* ctx, const bahip_surfels* surfels, float grid_cell_size);

/* BA planes of one frame: create for the given image sizes, refresh from the frame's images (on the context
 * stream), destroy.  The Keyframe constructor (B/keyframe.cc:81-158) is where the reference derives normals, radii
 * and luma from its inputs; the planes are one more derived product of the same step. */
int bahip_frame_planes_create(bahip_context* ctx, int depth_width, int depth_height, int color_width, int color_height,
                              bahip_frame_planes** out);
int bahip_frame_planes_update(bahip_context* ctx, bahip_frame_planes* planes, const bahip_frame* frame);
void bahip_frame_planes_destroy(bahip_frame_planes* planes);

int bahip_set_keyframes(bahip_context* ctx, const bahip_keyframe* keyframes, int num_keyframes);
/* Read back the poses of the bound keyframes (7 floats each); synchronises. */
int bahip_get_keyframe_poses(bahip_context* ctx, float* global_T_frame_out, int num_keyframes);

/* ---- optimisation stages ---------------------------------------------------------------------- */
/* B/kernels.h UpdateSurfelActivationCUDA (B/kernel_surfel_activation.cc:39-67): surfels
 * [0, surfels_size) become active iff associated with >= 1 keyframe whose activation is kActive. */
int bahip_update_surfel_activation(bahip_context* ctx, const bahip_surfels* surfels, uint32_t surfels_size);
/* B/kernels.h:301-308 AssignColorsCUDA (B/kernel_assign_colors.cc:39-80), behind DirectBA::AssignColors
 * (B/direct_ba.cc:456-459): the colour of every surfel becomes the mean of the bilinear RGBA samples at its colour pixel
 * in all bound keyframes it is associated with (any activation); surfels seen by none keep their colour. */
int bahip_assign_colors(bahip_context* ctx, const bahip_surfels* surfels);
/* B/kernels.h UpdateSurfelNormalsCUDA (B/kernel_opt_geometry.cc:39-77). */
int bahip_update_surfel_normals(bahip_context* ctx, const bahip_surfels* surfels);
/* B/kernels.h OptimizeGeometryIterationCUDA (B/kernel_opt_geometry.cc:80-201). */
int bahip_optimize_geometry_iteration(bahip_context* ctx, int use_depth_residuals, int use_descriptor_residuals,
                                      const bahip_surfels* surfels);
/* UpdateSurfelActivationCUDA for surfels [0, activation_surfels_size) followed by OptimizeGeometryIterationCUDA, as ONE
 * sweep: a surfel is active iff it is associated with a kActive keyframe, and the normals pass of the geometry step tests
 * exactly those associations anyway (B/direct_ba_alternating.cc:441-487 calls the two back to back).  Same flags and the same
 * surfels, bit for bit, as the two separate calls. */
int bahip_update_activation_and_optimize_geometry(bahip_context* ctx, int use_depth_residuals, int use_descriptor_residuals,
                                                  const bahip_surfels* surfels, uint32_t activation_surfels_size);
/* B/kernels.h AccumulatePoseEstimationCoeffsCUDA (B/kernel_opt_pose.cc:39-97): one frame, one
 * linearisation point.  H = 21 floats (row-major upper triangle), b = 6 floats, on the host;
 * synchronises like the reference does (B/kernel_opt_pose.cc:94-96). */
int bahip_accumulate_pose_estimation_coeffs(bahip_context* ctx, int use_depth_residuals,
                                            int use_descriptor_residuals, const bahip_frame* frame,
                                            const float frame_T_global[12], const bahip_surfels* surfels,
                                            float* H, float* b);
/* DirectBA::EstimateFramePose (B/direct_ba_alternating.cc:42-283) for an arbitrary frame:
 * <= 30 Gauss-Newton steps, each = accumulate + double-precision LDLT + T <- T*exp(-x), entirely
 * on the device; one read-back at the end.  Outputs on the host. */
int bahip_estimate_frame_pose(bahip_context* ctx, int use_depth_residuals, int use_descriptor_residuals,
                              const bahip_frame* frame, const float global_T_frame_initial[7],
                              const bahip_surfels* surfels, float global_T_frame_out[7],
                              int* iterations_done, int* converged);
/* The pose phase of one alternating-BA iteration (B/direct_ba_alternating.cc:545-577): every
 * bound keyframe that is not kInactive gets its own EstimateFramePose, all keyframes batched per
 * Gauss-Newton round (they are mutually independent: surfels are frozen during this phase).
 * Updated poses stay in the device keyframe table (fetch with bahip_get_keyframe_poses) and are
 * also returned here.  iterations_done / converged: per keyframe (0 / 1 for skipped ones).
 * rounds_out: number of batched rounds (= max iterations over keyframes). */
int bahip_estimate_keyframe_poses(bahip_context* ctx, int use_depth_residuals, int use_descriptor_residuals,
                                  const bahip_surfels* surfels, float* global_T_frame_out,
                                  int* iterations_done, int* converged, int* rounds_out);

/* The same pose phase, followed on the device by the activation update of B/direct_ba_alternating.cc:556-577: a keyframe
 * whose pose moved (log(old^-1 * new) fails the convergence test of B/convergence_analysis.h:43-51) becomes kActive, one
 * that did not becomes kInactive.  The device keyframe table (and its host mirror inside the context) then carries the new
 * poses AND activations, so the next iteration's sweeps need no bahip_set_keyframes.  moved: per bound keyframe 1 / 0 (0 for
 * keyframes that were kInactive); num_converged_out: keyframes that were kInactive or did not move. */
int bahip_estimate_keyframe_poses_and_update_activation(bahip_context* ctx, int use_depth_residuals, int use_descriptor_residuals,
                                                        const bahip_surfels* surfels, float* global_T_frame_out,
                                                        int* iterations_done, int* converged, int* moved, int* rounds_out,
                                                        int* num_converged_out);
/* The whole loop of the alternating scheme (B/direct_ba_alternating.cc:345-718) over poses and geometry for a FIXED surfel set,
 * driven by the device: up to max_iterations iterations -- [activation window | co-visible propagation], activation + geometry
 * sweep, pose phase -- are queued back to back; the last solve launch of every pose phase evaluates the loop's stopping rule
 * (every keyframe counts as converged and iteration >= min_iterations - 1, :693-701) and raises a device word that turns
 * every launch queued behind it into a no-op.  The host waits ONCE, for the last launch, instead of after every Gauss-Newton
 * round.  A pose phase that needs more rounds than were queued for it raises the word too; the host then finishes that phase
 * round by round and queues the rest again -- results never depend on how many rounds were queued.  Needs bahip_set_keyframes,
 * bahip_set_covisibility and (fixed_window) bahip_set_activation_window; the device keyframe table carries the new poses and
 * activations afterwards (without the co-visible propagation that follows an iteration that did not end the loop: the caller
 * applies DetermineCovisibleActiveKeyframes itself when *converged_out == 0).
 * *handled_out = 0 and nothing done when the configuration is not covered (keyframe sharding, more keyframes than one launch of
 * the pose sums takes): the caller then drives the iterations through the stage functions above.
 * global_T_frame_out: 7 floats per bound keyframe; activation_out: BAHIP_KF_* per bound keyframe; not_converged_out: pose
 * estimations that hit BAHIP_MAX_POSE_ITERATIONS. */
typedef struct {
  int use_depth_residuals, use_descriptor_residuals;
  int fixed_window;              /* apply the bound activation window at the top of every iteration */
  int activate_in_geometry;      /* 1: UpdateSurfelActivation is part of the geometry sweep (full window); 0: flags as they are */
  uint32_t activation_surfels_size;
  int min_iterations, max_iterations;
} bahip_alternating_options;
int bahip_alternating_iterations(bahip_context* ctx, const bahip_alternating_options* options, const bahip_surfels* surfels,
                                 float* global_T_frame_out, int* activation_out, int* handled_out, int* iterations_done_out,
                                 int* converged_out, int* pose_rounds_out, int* pose_steps_out, int* not_converged_out);
/* Co-visibility lists of the bound keyframes in CSR form over bound indices (offsets: num_keyframes + 1 entries); call after
 * bahip_set_keyframes.  bahip_propagate_covisible_activation is DirectBA::DetermineCovisibleActiveKeyframes
 * (B/direct_ba.cc:549-564) on the device table: kInactive keyframes co-visible with a kActive one become kCovisibleActive.
 * Asynchronous on the context stream. */
int bahip_set_covisibility(bahip_context* ctx, const int* offsets, const int* indices, int num_keyframes);
int bahip_propagate_covisible_activation(bahip_context* ctx);
/* Fixed active window of the alternating scheme (B/direct_ba_alternating.cc:353-371): in_window[k] != 0 for the bound
 * keyframes inside it.  bahip_apply_activation_window sets those kActive, all others kInactive, and then propagates the
 * co-visible activation -- the top of every iteration, on the device table; asynchronous. */
int bahip_set_activation_window(bahip_context* ctx, const uint8_t* in_window, int num_keyframes);
int bahip_apply_activation_window(bahip_context* ctx);

/* ---- surfel lifecycle --------------------------------------------------------------------------- */
/* B/kernels.h DetermineSupportingSurfelsCUDA / ...AndMergeSurfelsCUDA
 * (B/kernel_supporting_surfels.cc:112-165).  supporting: BAHIP_MERGE_BUFFER_COUNT device planes
 * of height x width u32 with the given pitch.  merged_count_out (host, may be NULL when
 * merge == 0) receives the number of surfels deleted by merging; synchronises when merging.
 * Deterministic: slots are claimed by the lowest surfel index (atomicMin), not by arrival. */
int bahip_determine_supporting_surfels(bahip_context* ctx, int merge, float merge_dist_factor,
                                       const bahip_frame* frame, const float frame_T_global[12],
                                       const bahip_surfels* surfels, uint32_t* const* supporting,
                                       uint32_t supporting_pitch_bytes, uint32_t* merged_count_out);
/* merged_count_out == NULL in bahip_determine_supporting_surfels(merge = 1, ...) defers the count: no read-back and no stream
 * synchronisation per keyframe; the surfels merged by all such calls since the last bahip_take_merged_count are returned (and
 * the counter cleared) here -- one synchronisation per batch of keyframes instead of one per keyframe. */
int bahip_take_merged_count(bahip_context* ctx, uint32_t* merged_count_out);
/* The merges of a BATCH of keyframes in one call (DirectBA's merge pass of a BA iteration and of the end tasks,
 * B/direct_ba_alternating.cc:491-540, B/direct_ba.cc:566-620): bahip_determine_supporting_surfels(merge = 1) for frames[0],
 * frames[1], ... in this order, with the same deletions -- but two dependent launches per keyframe instead of three: the sweep that
 * applies keyframe j's decisions runs beside the one that inserts keyframe j + 1's surfels (they alternate between the caller's planes
 * and a second set the context owns).  frame_T_global_3x4: 12 floats per frame.  Inside bahip_lifecycle_batch_begin / _set_frames
 * each keyframe's sweeps run over the tiles it can see.  Both sets of planes end EMPTY (the lists are not an output of a merge batch).
 * Round 6, second form: when the lifecycle batch knows every frame of the call, the planes are not used at all -- which surfels share
 * a sparse cell of frame j does not change while a batch merges, so the members of every (frame, cell) are listed up front for all
 * frames at once and every frame costs ONE launch of one thread per cell (the three lowest indices still alive, the decisions of
 * B/kernel_supporting_surfels.cu:60-86 in ascending order); the same deletions (bahip_debug_set_merge_cells selects the form).
 * merged_count_out: NULL defers the count to bahip_take_merged_count. */
int bahip_merge_surfels_for_keyframes(bahip_context* ctx, float merge_dist_factor, const bahip_frame* frames, const float* frame_T_global_3x4,
                                      int num_frames, const bahip_surfels* surfels, uint32_t* const* supporting, uint32_t supporting_pitch_bytes,
                                      uint32_t* merged_count_out);
/* A batch of keyframes creating or merging surfels on one cloud (the BA loop's creation pass, its merge pass, the merges of the
 * end tasks): bahip_lifecycle_batch_begin takes the bounding spheres of the cloud's 64-surfel tiles once; until
 * bahip_lifecycle_batch_end the per-keyframe sweeps of bahip_determine_supporting_surfels / bahip_create_surfels_for_keyframe over
 * the same buffer skip, per wavefront, the tiles the keyframe cannot see.  Results do not depend on the bracket (the test is the
 * conservative one of the BA sweeps); surfels may be appended and marked deleted inside it, not moved: compaction, the spatial
 * sort and the geometry step end the batch by themselves.  Ours: the reference sweeps the whole cloud per keyframe
 * (B/kernel_supporting_surfels.cu:36-60). */
int bahip_lifecycle_batch_begin(bahip_context* ctx, const bahip_surfels* surfels);
/* Optional, after _begin: the frames of the batch -- frame_T_global as 3x4 row-major matrices, exactly the 12 coefficients the
 * per-keyframe calls will be given (bahip_determine_supporting_surfels), or bound keyframe indices (bahip_create_surfels_for_
 * keyframe[s]).  The batch then knows which tiles each frame can see and that frame's sweeps run over those tiles only (a launch of a
 * few hundred workgroups instead of one over the whole cloud).  A frame that is not found sweeps with the per-tile test as before.
 * From here to _end the supporting planes passed to the MERGING calls (bahip_determine_supporting_surfels with merge != 0) belong to
 * the backend: every such call leaves them empty (all slots kInvalidIndex) instead of holding the keyframe's lists, which spares the
 * next keyframe of the batch its fill launch; the surfel buffer and the merged counts are what they are without the bracket. */
int bahip_lifecycle_batch_set_frames(bahip_context* ctx, const float* frame_T_global_3x4, int num_frames);
int bahip_lifecycle_batch_set_keyframes(bahip_context* ctx, const int* keyframe_indices, int num_keyframes);
int bahip_lifecycle_batch_end(bahip_context* ctx);
/* B/kernels.h CreateSurfelsForKeyframeCUDA (B/kernel_create_surfels.cc:40-183), including the
 * DetermineSupportingSurfelsCUDA call that DirectBA::CreateSurfelsForKeyframe issues first
 * (B/direct_ba.cc:345-355).  keyframe_index selects the bound keyframe; covis / n_covis: indices
 * of bound keyframes used for the new-surfel outlier filter.  new_surfel_count_out on the host;
 * synchronises.  Deterministic: the lowest linear pixel index wins a sparse cell. */
int bahip_create_surfels_for_keyframe(bahip_context* ctx, int keyframe_index, int filter_new_surfels,
                                      int min_observation_count, const int* covis, int n_covis,
                                      const bahip_surfels* surfels, uint32_t* const* supporting,
                                      uint32_t supporting_pitch_bytes, uint32_t* new_surfel_count_out);
/* A batch of keyframes, in the order given, each seeing what the ones before it appended -- n calls of
 * bahip_create_surfels_for_keyframe without the host in between (the cloud's size stays on the device during the batch; one host
 * wait at the end instead of two per keyframe).  covis_offsets[num_keyframes + 1] / covis_indices: the co-visibility lists of the
 * keyframes, concatenated, as indices into the bound keyframe list.  new_surfel_count_out: the surfels all of them appended; the
 * caller adds it to surfels_size.  A keyframe that does not fit the capacity creates nothing and raises the flag of
 * bahip_context_take_capacity_exceeded, the others go on (B/kernel_create_surfels.cc:162-165 per keyframe).
 * Round 6: inside a lifecycle batch that knows the keyframes (bahip_lifecycle_batch_set_keyframes) the keyframes do not wait for each
 * other's sweeps: what the cloud at the batch's begin occupies, which pixel of a free cell would create a surfel and whether it passes
 * the filter are found for ALL keyframes up front (three launches); each keyframe but the last then costs ONE launch -- its candidates
 * whose cell is still free are counted, scanned and appended, what it appends is pushed into the next keyframe's occupancy, what the
 * batch appended earlier is pulled into it.  The last keyframe takes the four-launch path, so the supporting planes end as a
 * one-keyframe call leaves them.  The same surfels at the same indices (bahip_debug_set_creation_chain selects the form). */
int bahip_create_surfels_for_keyframes(bahip_context* ctx, const int* keyframe_indices, int num_keyframes, int filter_new_surfels,
                                       int min_observation_count, const int* covis_offsets, const int* covis_indices,
                                       const bahip_surfels* surfels, uint32_t* const* supporting, uint32_t supporting_pitch,
                                       uint32_t* new_surfel_count_out);
/* 1 if the last bahip_create_surfels_for_keyframe on this context created nothing because the surfels would not have fit
 * into `capacity` (the reference's soft failure, B/kernel_create_surfels.cc:162-165); reading clears the flag. */
int bahip_context_take_capacity_exceeded(bahip_context* ctx);
/* B/kernels.h DeleteSurfelsAndUpdateRadiiCUDA (B/kernel_delete_surfels.cc:40-120); synchronises. */
int bahip_delete_surfels_and_update_radii(bahip_context* ctx, int min_observation_count,
                                          const bahip_surfels* surfels, uint32_t* deleted_count_out);
/* B/kernels.h CompactSurfelsCUDA (B/kernel_compact_surfels.cu:159-279): surfel_count valid
 * surfels end up in [0, surfel_count).  surfels->active may be NULL. */
int bahip_compact_surfels(bahip_context* ctx, uint32_t surfel_count, const bahip_surfels* surfels);

/* ---- intrinsics (B/kernels.h OptimizeIntrinsicsCUDA, B/kernel_opt_intrinsics.cc:39-281) --------- */
int bahip_optimize_intrinsics(bahip_context* ctx, int optimize_depth_intrinsics, int optimize_color_intrinsics,
                              const bahip_surfels* surfels, bahip_camera* out_color_camera,
                              bahip_camera* out_depth_camera, float* out_a);

/* ---- PCG solver (B/direct_ba_pcg.cc:229-646 over B/kernels.h PCG*CUDA) -------------------------- */
typedef struct bahip_pcg_options {
  int optimize_poses, optimize_geometry, optimize_depth_intrinsics, optimize_color_intrinsics;
  int use_depth_residuals, use_descriptor_residuals;
  int max_inner_iterations;     /* 30 */
  int gauge_keyframe;           /* index of the keyframe held fixed (reference: rand() % K) */
} bahip_pcg_options;
/* One outer Gauss-Newton iteration of the PCG scheme: builds and solves the full normal
 * equations matrix-free, applies the update to poses (device table), surfels, intrinsics and
 * cfactors.  Outputs new intrinsics on the host; inner_steps_out = PCG steps used. */
int bahip_pcg_iteration(bahip_context* ctx, const bahip_pcg_options* opt, const bahip_surfels* surfels,
                        bahip_camera* out_color_camera, bahip_camera* out_depth_camera, float* out_a,
                        int* inner_steps_out, int* num_converged_out /* keyframes whose pose update is below the
                        convergence threshold, gauge keyframe included (B/direct_ba_pcg.cc:556-575) */);

/* ---- PCG solver, stage by stage (B/kernels.h:397-491): the entry points a caller that keeps the reference's own PCG driver
 * (B/direct_ba_pcg.cc:229-646) binds -- one bahip_ function per *CUDA function, same call sequence:
 *     r = M = 0; bahip_pcg_begin; for each keyframe bahip_pcg_init; bahip_pcg_init2;
 *     per inner step: g = 0 (from the second step on); for each keyframe bahip_pcg_step1; bahip_pcg_step2; read beta_n;
 *     bahip_pcg_step3; ... ; bahip_update_surfels_from_pcg_delta; bahip_update_cfactors_from_pcg_delta.
 * All vectors (unknown_count floats) and the three scalars are DEVICE pointers owned by the caller (the reference's
 * CUDABuffer_<PCGScalar>).  What the reference adds to the vectors with float atomics -- the entries of the poses, of the
 * intrinsics and of the cfactor cells, and the three dot products -- is summed exactly in accumulators inside the context
 * (badslam_amd/csrc/exact_sum.h) and written by the next stage that reads it: bahip_pcg_init2 writes those entries of r and
 * M, bahip_pcg_step2 those of g and *alpha_d (so r, M, g hold only their per-surfel entries between the keyframe loop and the
 * stage that follows it).  The per-surfel entries are read-modify-written per keyframe call exactly like the reference's.
 * Same bits as bahip_pcg_iteration, which runs the keyframe loops as single sweeps. */
typedef struct bahip_pcg_layout {   /* unknown layout (B/direct_ba_pcg.cc:232-307) */
  int optimize_poses, optimize_geometry, optimize_depth_intrinsics, optimize_color_intrinsics;
  int use_depth_residuals, use_descriptor_residuals;
  uint32_t unknown_count;
  uint32_t surfel_unknown_start_index;              /* valid if optimize_geometry */
  uint32_t depth_intrinsics_unknown_start_index;    /* valid if optimize_depth_intrinsics; a_unknown_index = this + 4 */
  uint32_t color_intrinsics_unknown_start_index;    /* valid if optimize_color_intrinsics */
} bahip_pcg_layout;
/* Sizes and clears the context's accumulators for a system with this layout over surfels_size surfels; before the first
 * bahip_pcg_init of an outer iteration. */
int bahip_pcg_begin(bahip_context* ctx, const bahip_pcg_layout* layout, uint32_t surfels_size);
/* PCGInitCUDA (B/kernels.h:397-416) for one keyframe: r -= J^T W F, M += diag(J^T W J).  kf_pose_unknown_index: first of the
 * keyframe's 6 pose unknowns; optimize_pose_of_keyframe = 0 for the gauge keyframe. */
int bahip_pcg_init(bahip_context* ctx, const bahip_pcg_layout* layout, const bahip_frame* frame, const float frame_T_global[12],
                   uint32_t kf_pose_unknown_index, int optimize_pose_of_keyframe, const bahip_surfels* surfels, float* pcg_r, float* pcg_M);
/* PCGInit2CUDA (B/kernels.h:418-428) */
int bahip_pcg_init2(bahip_context* ctx, const bahip_pcg_layout* layout, uint32_t surfels_size, float a, float* pcg_r, float* pcg_M,
                    float* pcg_delta, float* pcg_g, float* pcg_p, float* pcg_alpha_n);
/* PCGStep1CUDA (B/kernels.h:430-453) for one keyframe: g += J^T W J p; the keyframe's share of alpha_d is kept in the context
 * until bahip_pcg_step2 (pcg_alpha_d is not touched here). */
int bahip_pcg_step1(bahip_context* ctx, const bahip_pcg_layout* layout, const bahip_frame* frame, const float frame_T_global[12],
                    uint32_t kf_pose_unknown_index, int optimize_pose_of_keyframe, const bahip_surfels* surfels, const float* pcg_p,
                    float* pcg_g);
/* PCGStep2CUDA (B/kernels.h:455-466); writes *pcg_alpha_d (the sum over the bahip_pcg_step1 calls since the last step 2,
 * epsilon terms included once per call like the reference's AddAlphaDEpsilonTerms) and *pcg_beta_n. */
int bahip_pcg_step2(bahip_context* ctx, const bahip_pcg_layout* layout, uint32_t surfels_size, float* pcg_r, const float* pcg_M,
                    float* pcg_delta, float* pcg_g, const float* pcg_p, const float* pcg_alpha_n, float* pcg_alpha_d, float* pcg_beta_n);
/* PCGStep3CUDA (B/kernels.h:468-474) */
int bahip_pcg_step3(bahip_context* ctx, const bahip_pcg_layout* layout, uint32_t surfels_size, const float* pcg_g, float* pcg_p,
                    const float* pcg_alpha_n, const float* pcg_beta_n);
/* UpdateSurfelsFromPCGDeltaCUDA (B/kernels.h:484-490), UpdateCFactorsFromPCGDeltaCUDA (B/kernels.h:492-496; the cfactor image is
 * the one of bahip_set_intrinsics) */
int bahip_update_surfels_from_pcg_delta(bahip_context* ctx, const bahip_surfels* surfels, int use_descriptor_residuals,
                                        uint32_t surfel_unknown_start_index, const float* pcg_delta);
int bahip_update_cfactors_from_pcg_delta(bahip_context* ctx, uint32_t cfactor_unknown_start_index, const float* pcg_delta);

/* ---- test hook --------------------------------------------------------------------------------------
 * Per-pair evaluation with the production device functions (association, the three raw residuals,
 * weights, pose Jacobians, image gradients) for `count` surfel indices against one frame; 40 floats
 * per index, layout documented in badslam_amd/csrc/kernels_pose.hip.  Host in / host out. */
int bahip_debug_evaluate_pairs(bahip_context* ctx, const bahip_frame* frame, const float frame_T_global[12],
                               const bahip_surfels* surfels, const uint32_t* surfel_indices, int count, float* out);

/* Reads `count` entries of one of the PCG vectors left by the last bahip_pcg_iteration
 * (which: 0 = r, 1 = M, 2 = delta, 3 = g, 4 = p).  With max_inner_iterations = 0 that call only
 * assembles r = -J^T W F and M = diag(J^T W J) and applies no update. */
int bahip_debug_read_pcg_vector(bahip_context* ctx, int which, size_t offset, size_t count, float* out);

/* The exact sum of `count` binary32 values, rounded once to binary64 (nearest, ties to even; NaN if a value is not finite),
 * through the device code the PCG scheme's dense sums and dot products use (badslam_amd/csrc/exact_sum.h).  mode 0: every
 * term with 64-bit integer atomics on the accumulator's limbs; mode 1: per-thread limb columns in workgroup memory, folded
 * per workgroup (the path of the per-unknown kernels).  Tests compare it with math.fsum. */
int bahip_debug_exact_sum(bahip_context* ctx, const float* values, size_t count, int mode, double* out);

/* Work census of one sweep of the bound keyframes over the surfels: counts[0] = (wavefront, keyframe)
 * candidates left by frustum culling, [1] = of those with >= 1 association, [2] = associated
 * (surfel, keyframe) pairs, [3] = pairs projecting into the image. */
/* Launch shapes of the surfel sweeps, process-wide; 0 = chosen from the surfel count (default).  tile_waves (1 | 4):
 * wavefronts per 64-surfel tile in the normals / geometry passes - results are bit-identical for both (the per-surfel
 * sums are defined as four interleaved partial sums).  pose_parts (1 | 2 | 4 | 8): wavefronts sharing a
 * tile's keyframes in the pose kernel (sums merged by float atomics in any case).
 * tile_waves = 5: the geometry step's HYBRID shape (round 6; the default below 8192 tiles once the sweeps have a run order with its
 * heavy list): the heavy tiles take four wavefronts (a keyframe class each), every other tile one -- in one launch; same bits.
 * bahip_debug_geometry_hybrid_launches: how many geometry steps have run in that shape so far. */
int bahip_debug_set_launch_shapes(int tile_waves, int pose_parts);
int bahip_debug_geometry_hybrid_launches(long long* launches_out);
/* Form of the pose sweep, process-wide; results are bit-identical for all of them.  0 = chosen from the sizes (default);
 * 1 = one wavefront per surfel tile, tile totals added to the normal equations with global 64-bit integer atomics (the only form
 * for shards and for more work items than fit the table); 2 = persistent workgroups, one per compute unit, that keep the normal
 * equations of every work item in LDS and flush them once (whenever the table fits 128 KB: up to 292 work items). */
int bahip_debug_set_pose_form(int form);
/* Test / experiment hook: 0 = the sweeps take their surfel tiles in buffer order; 1 (default; BAHIP_TILE_ORDER=0 in the
 * environment switches it off too) = heavy work first, from the candidate counts of an earlier pose phase (wave_cull.h:
 * scheduled_tile).  A scheduling hint: results are bit-identical either way. */
int bahip_debug_set_tile_order(int enabled);
/* The schedule in use (test hook): *padded_tiles_out = the grid size it is valid for (0: none yet); words_out (may be NULL)
 * receives up to max_words of it: [0] heavy tiles, [8 .. 8 + 1024) their list, then one tile per regular position (padded_tiles
 * words, a permutation of the tiles), then one flag per tile (non-zero = in the heavy list). */
int bahip_debug_read_tile_schedule(bahip_context* ctx, uint32_t* padded_tiles_out, uint32_t* words_out, size_t max_words);
/* The intrinsics sweep appends its per-cell records to buffers that keep their size unless a call overflows them (the next call
 * then takes that call's largest demand + 25 %; the first call an estimate) (kernels_intrinsics.hip); records that do not fit go
 * out as atomics, with the same result.  records_per_block >= 0 fixes the size (0: no buffers, < 0:
 * automatic again) -- for the tests of the overflow path and for A/B timing.  bahip_debug_intrinsics_bin_stats: capacity and the
 * largest / total demand of the last call. */
int bahip_debug_set_intrinsics_bin_capacity(bahip_context* ctx, int records_per_block);
int bahip_debug_intrinsics_bin_stats(bahip_context* ctx, uint32_t* capacity_out, uint32_t* most_out, uint64_t* total_out);
/* The intrinsics step's sweep runs in slices of its schedule when the cloud is large (the per-pair records of a slice are reduced on a
 * second stream while the next slice sweeps; two buffer sets of one slice's records each): 1 .. 16 fixes the number of slices (tests on
 * small scenes), 0 = by the size of the sweep (default).  The sums do not depend on it. */
int bahip_debug_set_intrinsics_slices(bahip_context* ctx, int slices);
/* The creation batch's scan + append launch (bahip_create_surfels_for_keyframes) runs a grid handshake and therefore never launches more
 * workgroups than the device holds at once (occupancy x compute units, at most 256).  groups > 0 lowers that limit (tests: the path a
 * partitioned or masked device takes), 0 restores it.  The created surfels do not depend on it. */
int bahip_debug_set_append_groups(int groups);
/* The creation batch as a chain of one launch per keyframe (bahip_create_surfels_for_keyframes inside a lifecycle batch that knows the
 * keyframes): 1 (default; environment BAHIP_CREATION_CHAIN=0 switches it off) or 0 = four launches per keyframe as in round 5; the same
 * surfels either way.  ..._batches: how many calls have taken the chain so far (tests assert the route). */
int bahip_debug_set_creation_chain(int enabled);
int bahip_debug_creation_chain_batches(long long* batches_out);
/* A merge batch by cell lists (bahip_merge_surfels_for_keyframes inside a lifecycle batch that knows the frames, every frame with its
 * BA planes): 1 (default; BAHIP_MERGE_CELLS=0 switches it off) or 0 = the pipelined insert / decide / apply sweeps of round 6's first
 * version; the same deletions either way.  ..._batches: how many calls have gone by cell lists. */
int bahip_debug_set_merge_cells(int enabled);
int bahip_debug_merge_cells_batches(long long* batches_out);
/* The LDS form holds the normal equations of at most 292 work items; longer lists are cut into slices, one launch each.  items > 0
 * makes the slices that small (tests: 200 keyframes in slices of 64), 0 restores the default. */
int bahip_debug_set_pose_lds_items(int items);
/* Shape of the LDS form (test hooks; 0 / -1 restore the defaults): wavefronts per workgroup (1 .. 16, default 16), and
 * parts_shift: 2^parts_shift wavefronts share a tile's work items (0 .. 3; -1: chosen from the grid size -- small grids, i.e.
 * shards of a multi-GPU run, split their tiles).  Integer sums: every shape gives the same bits. */
int bahip_debug_set_pose_lds_shape(int waves, int parts_shift);
/* Gauss-Newton rounds queued ahead per host wait (bahip_estimate_*: the later rounds of a phase read the number of work items
 * still iterating from device memory and do nothing when it is zero, so the host need not wait for a round before it queues
 * the next): 0 = as many as the previous phase needed (default; in bahip_alternating_iterations: what the last phases needed,
 * one less per following iteration down to two), n >= 1 = exactly n (1: wait after every round, the round-3 behaviour).  Results
 * do not depend on it. */
int bahip_debug_set_pose_rounds_ahead(int rounds);
/* 1: in bahip_alternating_iterations the launch that ends an iteration's pose phase also opens the next iteration (activation
 * window / propagation, work items) when the keyframe table has at most 1024 entries; 0 (default -- the fused launch measured
 * slower): a launch of its own does.  Results do not depend on it. */
int bahip_debug_set_fused_iteration_begin(int enabled);
/* How the binned per-cell records of the intrinsics step are added (kernels_intrinsics.hip): 0 = into a table in LDS by
 * binary64 LDS atomics, 1 = sorted by cell in LDS and added by the thread that owns the cell, -1 = the default.  Same sums (binary64 sums of
 * binary32 terms, rounded to binary32 afterwards). */
int bahip_debug_set_intrinsics_reduce_form(int form);
/* 0: bahip_alternating_iterations reports "not handled" and callers drive the loop through the stage functions, one host wait
 * per Gauss-Newton round (BAHIP_DEVICE_LOOP=0 in the environment does the same); 1 (default): the device-driven loop.  Same bits. */
int bahip_debug_set_device_loop(int enabled);
/* Calls of bahip_alternating_iterations since the process started that the device-driven loop handled / declined (handled_out = 0:
 * switched off, keyframe sharding, a host all-reduce hook, more work items than one launch of the pose sweep takes).  bench.py prints
 * both; the test suite checks that the default configuration is handled. */
int bahip_debug_alternating_loop_calls(long long* handled_out, long long* declined_out);
/* 0: the step-1 sweep of the PCG scheme always runs one tile per wavefront with global atomics on the exact accumulators; 1
 * (default): persistent workgroups that keep the pose block of the dense head in LDS when it fits (up to ~295 keyframes) and
 * the grid fills the chip; 2: that form whenever the table fits (tests on small scenes).  Exact (integer) sums: the same bits. */
int bahip_debug_set_pcg_lds_form(int mode);
/* launches of the pose accumulation in either form since the last reset (process-wide); bench.py names the dominant kernel by it */
int bahip_debug_pose_form_launches(long long* global_form, long long* lds_form, int reset);
/* Kernel dispatches of the pose accumulate sweep since the process started (each slice of a sliced launch counts; never reset).
 * A profile of a bench run uses it to pick the dispatches of the timed region out of rocprofv3's per-dispatch rows. */
int bahip_debug_pose_kernel_dispatches(long long* dispatches_out);
/* launches of the PCG scheme's step-1 sweep by form since the process started: one tile per wavefront with global atomics on the
 * exact accumulators / persistent workgroups with the pose block of the dense head in LDS (bench.py names the kernel it measured) */
int bahip_debug_pcg_step1_form_launches(long long* tile_form, long long* lds_form);
/* The fixed-point representation of a tile total of the pose normal equations (badslam_amd/csrc/ba_device.h: hb_split):
 * out[3 i .. 3 i + 2] = limb 0 (weight 2^-32), limb 1 (weight 1), valid (0: not finite or 2^52 and beyond -- such a total is
 * not added and fails the pose estimation). */
int bahip_debug_pose_limbs(bahip_context* ctx, const float* values, size_t count, long long* out);
/* The residual Jacobian functions of the kernels (ba_device.h: jac_*) on explicit inputs, for the golden vectors of
 * tests/golden/jacobians.json.  kind: 0 depth/pose (in: nl[3] u[3] inv_std; out 6), 1 descriptor/pose (ls[3] gx gy; 6),
 * 2 descriptor/surfel (rn[3] lp[3] gx gy cfx cfy; 1), 3 depth/intrinsics (px py depth inv_std n.Frow0 n.Frow1 dot cfactor
 * raw_inv_depth exp_inv_depth corrected_inv_depth; 6), 4 descriptor/colour intrinsics (gx gy nx ny; 4). */
int bahip_debug_jacobian(bahip_context* ctx, int kind, const float* in, int n_in, float* out, int n_out);
/* One wavefront: in = 64 lanes x 28 floats; out[0..27] = totals from the halving reduction used by the pose kernel
 * (wave_reduce.h), out[28..55] = the same totals from the xor-butterfly wave_sum, out[56..63] / out[64..79] = totals of the
 * first 8 / 16 columns from the small halving reductions the PCG sweeps use (wave_reduce_small<8>, <16>). */
int bahip_debug_wave_reduce(bahip_context* ctx, const float* in_64x28, float* out_80);
/* One Gauss-Newton pose update with the device code of the pose solve (binary64 LDLT of the binary32 H (21) | b (6), x as
 * binary32, T <- T * exp(-x) with the defined sin / cos): out = x[6] | T_next[7] | frame_T_global(T_next)[12]. */
/* Counter calibration (tooling): reads a zero-filled device buffer of `bytes` with 4-byte loads, `repeats` launches of
 * read_pattern_kernel.  pattern 0: every dword once, coalesced; 1: one dword per 128-byte line.  Run under
 * `rocprofv3 --pmc FETCH_SIZE` to learn what the counter reports for a known amount of data at this access width
 * (scripts/profile_round.sh). */
int bahip_debug_read_pattern(bahip_context* ctx, size_t bytes, int pattern, int repeats);
/* rcp_exact (kind 0) / sqrt_exact (kind 1) of the device code on n host values (ba_device.h: the few-instruction exact
 * reciprocal and square root the sweeps use instead of the compiler's IEEE sequences); checked exhaustively by the tests.
 * kind 2 / 3 / 4: the defined sin / cos / atan of the SE(3) exponential and logarithm (se3_device.h). */
int bahip_debug_exact_math(bahip_context* ctx, int kind, const float* in, float* out, size_t n);
int bahip_debug_pose_step(bahip_context* ctx, const float* H21_b6, const float* global_T_frame, float* out_25);
int bahip_debug_count_pairs(bahip_context* ctx, const bahip_surfels* surfels, uint64_t* counts_out);

/* ---- instrumentation ---------------------------------------------------------------------------- */
/* Time (ms, hipEvent on the context stream) and launch count of the kernels issued by the last
 * call of each stage; used by bench.py for the roofline line.  stage: 0 activation, 1 geometry,
 * 2 pose accumulate, 3 pose solve, 4 intrinsics (accumulation sweep + reduction of the binned records + Schur complement),
 * 5 the step-1 sweeps of the PCG scheme, 6 the intrinsics sweep alone, 7 the reduction of the binned records alone. */
int bahip_last_stage_time_ms(bahip_context* ctx, int stage, float* ms_out, int* launches_out);
/* enabled: 0 off; 1 = the counters cover the last call of each stage; 2 = cumulative since this call; 3 = cumulative,
 * stage 2 only (two event records per pose round instead of ten per BA iteration). */
int bahip_set_profiling(bahip_context* ctx, int enabled);
/* Work units of the launches counted above (stage 2: sum over launches of the keyframes still
 * iterating in that Gauss-Newton round; other stages: launches). */
int bahip_stage_work_units(bahip_context* ctx, int stage, long long* units_out);
/* ---- surfel shards <-> the whole cloud ------------------------------------------------------------------------------------------
 * Under surfel sharding rank r of `world` owns every world-th chunk of `chunk` consecutive surfels of ONE cloud (local surfel l is
 * global surfel ((l / chunk) * world + r) * chunk + l % chunk).  The per-surfel work of an iteration needs no other rank's
 * surfels; the surfel LIFECYCLE does -- whether a pixel is already supported, which surfels of a cell merge, where compaction
 * moves the last surfels -- and is not worth a distributed algorithm: it runs a few times per BundleAdjustment call.  For those
 * phases every rank assembles the whole cloud, runs the unsharded lifecycle code on it (identical work, identical result on
 * every rank) and takes its shard back out, so a sharded run with surfel updates ends with the bits of the unsharded run.
 *
 * bahip_gather_surfel_shards: cloud := the union of all ranks' shards.  `cloud` must have room for the sum of the shard sizes;
 * rows 0 .. 7 and the active flags travel (one int64 all-reduce per row: bit patterns are preserved), the scratch rows do not.
 * Returns the cloud's size and surfel count (the sums over the ranks) and checks that the shard sizes are those of the
 * chunk-cyclic partition.  bahip_extract_surfel_shard: shard := this rank's surfels of a cloud of cloud->surfels_size; returns
 * the shard's size.  Without a communicator / hook (one GPU) both are plain copies. */
int bahip_gather_surfel_shards(bahip_context* ctx, const bahip_surfels* shard, uint32_t shard_surfel_count, int rank, int world, uint32_t chunk,
                               bahip_surfels* cloud, uint32_t* cloud_surfels_size_out, uint32_t* cloud_surfel_count_out);
int bahip_extract_surfel_shard(bahip_context* ctx, const bahip_surfels* cloud, int rank, int world, uint32_t chunk, bahip_surfels* shard,
                               uint32_t* shard_surfels_size_out);

/* Multi-GPU accounting: how many sums over the ranks this context has requested (through the hook or RCCL) since the last
 * reset, and the bytes of the buffers summed (per rank, one direction).  A single-GPU context reports zeros. */
int bahip_exchange_stats(bahip_context* ctx, long long* calls_out, long long* bytes_out, int reset);

#ifdef __cplusplus
}
#endif
#endif  /* BADSLAM_HIP_H_ */
