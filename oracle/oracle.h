/*
 * oracle.h -- CPU restatement ("oracle") of the BAD SLAM direct bundle-adjustment hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load liboracle.so.  The product path (badslam_amd/) never
 * includes, links or calls anything in this directory.
 *
 * Parity pinning: the reference (ETH3D/badslam) ships no golden vectors, fixtures or
 * known-answer files for this path and cannot be compiled as a whole here (CUDA + Eigen + Qt + ...).
 * The oracle is pinned by (0) THE REFERENCE'S OWN FUNCTIONS: the device-math headers of the path
 * (B/surfel_projection_nvcc_only.cuh, B/cost_function.cuh, B/robust_weighting.cuh, B/util.cuh, ...)
 * compile for the host through a stand-in cuda_runtime.h (oracle/ref_shim/, oracle/Makefile ->
 * oracle/_ref/libbadslam_ref.so, sources read where they lie under /root/reference), and
 * tests/test_cpu_oracle_vs_reference.py compares association, residuals, weights, gradients and the
 * cost of ~4e5 (surfel, keyframe) pairs with them (measured deltas: DESIGN.md section 6); and whole
 * KERNELS of the reference -- B/kernel_surfel_activation.cu, kernel_opt_geometry.cu, kernel_assign_colors.cu,
 * kernel_delete_surfels.cu, kernel_supporting_surfels.cu, kernel_create_surfels.cu, and kernel_opt_pose.cu,
 * kernel_pcg.cu, kernel_opt_intrinsics.cu with their block votes and block sums modelled, and
 * cuda_depth_processing.cu, cuda_image_processing.cu (keyframe preprocessing), kernel_compact_surfels.cu
 * -- run on the host through a stand-in launcher (oracle/ref_shim/ref_kernels.cc, ref_preprocess.cc) in the
 * order of the reference's host drivers, stage against stage with the oracle (same test file);
 * the outputs of those kernels for a small scene are also COMMITTED as a golden file
 * (tests/make_golden_reference_kernels.py -> tests/golden/reference_kernels.npz), replayed by the
 * oracle (tests/test_cpu_golden_reference.py) and by the HIP path (tests/test_gpu_golden_reference.py)
 * on machines that have neither the reference nor the library built from it;
 * (1) the reference's own closed-loop test criteria restated in
 * tests/test_oracle_*_closed_loop.py (applications/badslam/src/badslam/test/ *.cc tolerances; all
 * twelve of them also run against the HIP path, badslam_amd/host/test_directba.cc),
 * (2) its residual Jacobians checked against golden vectors generated HERE by importing the
 * reference's own derivation script (applications/badslam/scripts/jacobians_derivation.py ->
 * scripts/make_golden_jacobians.py -> tests/golden/jacobians.json), (3) golden VALUES of the
 * building blocks the same script defines -- depth calibration, projection, unprojection, bilinear
 * weights, the rotation of the exponential map -- (scripts/make_golden_functions.py ->
 * tests/golden/functions.json) and (4) central finite differences of the oracle's own residual
 * functions.  Where the arithmetic is DEFINED here rather than taken from the reference (summation
 * orders, fixed point, exact sums, defined sin / cos / atan / exp: DESIGN.md section 3), the definitions are
 * shared with the kernels, and (0) and (1) are what ties them to the reference.
 *
 * All arithmetic is IEEE binary32 unless stated (compiled with -ffp-contract=off); the small
 * dense solves are done in binary64 exactly where the reference does so (Eigen LDLT on
 * H.cast<double>()).
 *
 * Paths cited below are relative to /root/reference/; B/ = applications/badslam/src/badslam/.
 */
#ifndef BADSLAM_ORACLE_H_
#define BADSLAM_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants: B/kernels.cuh:38-93, B/cost_function.cuh:44-52,105-109 ---- */
#define ORC_INVALID_DEPTH_BIT 0x8000u
#define ORC_UNKNOWN_DEPTH 65535u
#define ORC_SURFEL_ACTIVE_FLAG 1u
#define ORC_MERGE_BUFFER_COUNT 3
#define ORC_INVALID_INDEX 4294967295u
#define ORC_COS_NORMAL_COMPAT 0.76604f

enum {
  ORC_SURFEL_X = 0, ORC_SURFEL_Y = 1, ORC_SURFEL_Z = 2, ORC_SURFEL_NORMAL = 3,
  ORC_SURFEL_RADIUS_SQ = 4, ORC_SURFEL_COLOR = 5, ORC_SURFEL_DESC1 = 6, ORC_SURFEL_DESC2 = 7,
  ORC_SURFEL_ACCUM0 = 8, /* ... ACCUM8 = 16 */
  ORC_SURFEL_DATA_ATTRS = 8, ORC_SURFEL_ATTRS = 17
};

enum { ORC_KF_ACTIVE = 0, ORC_KF_COVIS_ACTIVE = 1, ORC_KF_INACTIVE = 2 };

/* PinholeCamera4f: fx, fy, cx, cy in the pixel-CORNER convention (libvis/src/libvis/camera.h:1740). */
typedef struct {
  float fx, fy, cx, cy;
  int32_t width, height;
} orc_camera;

/* B/surfel_projection.cuh:129-149 (DepthParameters). cfactor is dense row-major,
 * cf_height x cf_width = ((H-1)/cell+1) x ((W-1)/cell+1), indexed (py/cell, px/cell). */
typedef struct {
  float a;
  float raw_to_float_depth;
  float baseline_fx;
  int32_t cell;
  float* cfactor;
  int32_t cf_width, cf_height;
} orc_depth_params;

/* SE3f stored like Sophus: unit quaternion (x,y,z,w) + translation. */
typedef struct {
  float q[4];
  float t[3];
} orc_se3;

/* B/keyframe.h:50-237.  Images are dense row-major. color is RGBA u8 with A = luma. */
typedef struct {
  int32_t width, height;             /* depth image size */
  int32_t color_width, color_height;
  uint16_t* depth;
  uint16_t* normals;
  uint16_t* radius;
  uint8_t* color;
  orc_se3 global_T_frame;
  float frame_T_global[12];          /* row-major 3x4, cached on every pose set */
  float global_R_frame[9];           /* row-major 3x3 */
  int32_t activation;
  float min_depth, max_depth;
  int32_t id;
  int32_t last_active_in_ba_iteration;
  int32_t last_covis_in_ba_iteration;
} orc_keyframe;

/* Surfel SoA: ORC_SURFEL_ATTRS rows of `capacity` floats; row r at data + r*capacity.
 * B/direct_ba.cc:122, B/kernels.cuh:69-93. */
typedef struct {
  float* data;
  uint8_t* active;
  uint32_t capacity;
  uint32_t surfels_size;
  uint32_t surfel_count;
} orc_surfels;

/* ---- SE3 (libvis/third_party/sophus/sophus/se3.hpp:293-313,440-467; so3.hpp:282-320,421-465) ---- */
void orc_se3_identity(orc_se3* T);
/* sin / cos as the backend defines them (binary64 polynomial evaluation, se3_device.h: sincos_det) */
void orc_sincos(float x, float* sin_out, float* cos_out);
/* exp as the backend defines it (binary32 reduction + polynomial in fused multiply-adds, ba_device.h: exp_det): the depth
 * deformation exp(-a / depth) and the bilateral filter's weights */
float orc_exp(float x);
float orc_atan(float x);
void orc_se3_exp(const float tangent[6], orc_se3* out);
void orc_se3_log(const orc_se3* T, float tangent[6]);
void orc_se3_mul(const orc_se3* a, const orc_se3* b, orc_se3* out);
void orc_se3_inverse(const orc_se3* a, orc_se3* out);
void orc_se3_matrix3x4(const orc_se3* T, float m[12]);
void orc_se3_rotation(const orc_se3* T, float r[9]);
/* B/keyframe.h:160-172 */
void orc_keyframe_set_global_T_frame(orc_keyframe* kf, const orc_se3* global_T_frame);

/* ---- frame preprocessing ahead of the keyframe (BadSlam::PreprocessFrame, B/bad_slam.cc:697-706) ---- */
/* B/cuda_depth_processing.cu:42-128 BilateralFilteringAndDepthCutoffCUDA (radius = int(radius_factor * sigma_xy + 0.5)) */
void orc_bilateral_filter_and_depth_cutoff(float sigma_xy, float sigma_value, float radius_factor, uint16_t max_depth,
                                           float raw_to_float_depth, const uint16_t* in_depth, int width, int height,
                                           uint16_t* out_depth);

/* ---- keyframe preprocessing (B/keyframe.cc:81-158) ---- */
/* B/cuda_image_processing.cu:165-175 */
void orc_compute_brightness(const uint8_t* rgb, int width, int height, uint8_t* rgba);
/* B/cuda_depth_processing.cu:134-264 */
void orc_compute_normals(const orc_camera* cam, const orc_depth_params* dp, const uint16_t* in_depth,
                         uint16_t* out_depth, uint16_t* out_normals);
/* B/cuda_depth_processing.cu:289-360 */
void orc_compute_point_radii(const orc_camera* cam, float raw_to_float_depth, const uint16_t* depth,
                             uint16_t* radius, uint16_t* out_depth);
/* B/cuda_depth_processing.cu:391-465 */
void orc_compute_min_max_depth(const uint16_t* depth, int width, int height, float raw_to_float_depth,
                               float* min_depth, float* max_depth);
/* Whole ctor #2: fills kf->depth/normals/radius/color (caller-allocated) from raw inputs. */
void orc_keyframe_from_images(orc_keyframe* kf, const orc_camera* depth_cam, const orc_depth_params* dp,
                              const uint16_t* depth_image, const uint8_t* rgb_image,
                              const orc_se3* global_T_frame);

/* ---- residual Jacobians in isolation (oracle_internal.h: jac_*), checked against tests/golden/jacobians.json ---- */
void orc_jac_depth_pose(const float nl[3], const float u[3], float inv_std, float J[6]);                    /* B/kernel_opt_pose.cu:88-93 */
void orc_jac_descriptor_pose(const float ls[3], float gx, float gy, float J[6]);                            /* B/kernel_opt_pose.cu:126-141 */
float orc_jac_descriptor_surfel(const float rn[3], const float lp[3], float gx, float gy, float cfx, float cfy);   /* B/kernel_opt_geometry.cu:170-190 */
void orc_jac_depth_intrinsics(int px, int py, float depth, float inv_std, float n_dot_Frow0, float n_dot_Frow1, float dot, float cfactor,
                              float raw_inv_depth, float exp_inv_depth, float corrected_inv_depth, float J[6]);      /* B/kernel_opt_intrinsics.cu:107-140 */
void orc_jac_descriptor_color_intrinsics(float gx, float gy, float nx, float ny, float J[4]);              /* B/kernel_opt_intrinsics.cu:176-199 */

/* ---- elementary pieces exposed for unit tests ---- */
/* Software restatement of the clamp-addressed bilinear normalized-float sampler
 * (B/keyframe.cc:67-73); returns luma in [0,1]. */
float orc_sample_luma(const uint8_t* rgba, int width, int height, float x, float y);
/* B/util.cuh:62-69 */
float orc_raw_to_calibrated_depth(float a, float cfactor, float raw_to_float_depth, uint16_t raw);
/* test hook: pixel-centre unprojection of pixel (x, y) at `depth` (B/surfel_projection.cuh:88-126) */
void orc_unproject(const orc_camera* cam, int x, int y, float depth, float out[3]);
uint32_t orc_pack_normal10(float x, float y, float z);       /* B/util_nvcc_only.cuh:66-84 */
void orc_unpack_normal10(uint32_t v, float n[3]);           /* B/util_nvcc_only.cuh:87-95 (renormalised) */
uint16_t orc_pack_normal8(float x, float y);                /* B/util.cuh:121-135 */
void orc_unpack_normal8(uint16_t v, float n[3]);            /* B/util.cuh:138-146 */
uint16_t orc_float_to_half(float f);
float orc_half_to_float(uint16_t h);

/* Per (surfel, keyframe) evaluation used by tests for Jacobian checks: evaluates association
 * and, if associated, the three raw residuals, weights and pose / surfel Jacobians.
 * Returns 1 if associated. out layout documented in oracle_core.c. */
typedef struct {
  int32_t associated;
  int32_t px, py;
  int32_t color_valid;
  float calibrated_depth;
  float depth_residual, depth_weight, depth_inv_stddev;
  float depth_jac_pose[6];
  float depth_jac_surfel;
  float desc_residual[2], desc_weight[2];
  float desc_jac_pose[2][6];
  float desc_jac_surfel[2];
  float grad[4]; /* grad_x_1, grad_y_1, grad_x_2, grad_y_2 */
} orc_pair_eval;
int orc_evaluate_pair(const orc_camera* color_cam, const orc_camera* depth_cam, const orc_depth_params* dp,
                      const orc_keyframe* kf, const float frame_T_global[12],
                      const orc_surfels* s, uint32_t surfel_index, orc_pair_eval* out);
void orc_evaluate_pairs(const orc_camera* color_cam, const orc_camera* depth_cam, const orc_depth_params* dp,
                        const orc_keyframe* kf, const float frame_T_global[12], const orc_surfels* s,
                        const uint32_t* surfel_indices, int count, orc_pair_eval* out);

/* ---- pose optimisation ---- */
/* B/kernel_opt_pose.cc:39-97 + B/kernel_opt_pose.cu:251-383 + B/gauss_newton.cuh:46-93.
 * H: 21 floats (row-major upper triangle), b: 6 floats.  accumulate_double == 0: the backend's definition of the sum
 * (per-surfel fma chains, fixed 64-surfel tile tree, fixed-point integer totals in two limbs; oracle_pose.c), which is what
 * orc_estimate_frame_pose uses; != 0: plain binary64 running sum in surfel order.  Returns the number of associated surfels. */
float orc_tile_tree_sum(const float lane_values[64]);
/* the classic xor butterfly (32, 16, 8, 4, 2, 1) in binary32: the backend's wave_sum */
float orc_wave_xor_sum(const float lane_values[64]);
uint32_t orc_accumulate_pose_coeffs(int use_depth, int use_desc, const orc_camera* color_cam,
                                    const orc_camera* depth_cam, const orc_depth_params* dp,
                                    const orc_keyframe* kf, const float frame_T_global[12],
                                    const orc_surfels* s, float H[21], float b[6],
                                    float* residual_sum, int accumulate_double);
uint32_t orc_accumulate_pose_coeffs_fixed(int use_depth, int use_desc, const orc_camera* color_cam,
                                          const orc_camera* depth_cam, const orc_depth_params* dp,
                                          const orc_keyframe* kf, const float frame_T_global[12],
                                          const orc_surfels* s, long long fixed_out[54] /* [27][2] limb pairs */);
/* value of a limb pair; and whether a tile total could not be added since the last reset (not finite, or 2^52 and beyond) */
double orc_pose_limbs_value(long long lo, long long hi);
int orc_pose_limbs(float v, long long out[2]);
int orc_pose_sum_invalid(int reset);
/* B/direct_ba_alternating.cc:42-283.  Returns number of GN iterations done; *converged set. */
int orc_estimate_frame_pose(int use_depth, int use_desc, const orc_camera* color_cam,
                            const orc_camera* depth_cam, const orc_depth_params* dp,
                            const orc_keyframe* kf, const orc_se3* global_T_frame_init,
                            const orc_surfels* s, orc_se3* global_T_frame_out, int* converged);
/* B/convergence_analysis.h:43-51 */
int orc_is_scale1_pose_converged(const float x[6]);
/* Solve H x = b like Eigen's H.cast<double>().selfadjointView<Upper>().ldlt().solve(b) for n<=8. */
void orc_ldlt_solve(int n, const double* H_full_rowmajor, const double* b, double* x);

/* ---- surfel activation, geometry ---- */
/* DirectBA::AssignColors (B/direct_ba.cc:456-459, B/kernel_assign_colors.cc:39-80) */
void orc_assign_colors(const orc_camera* color_cam, const orc_camera* depth_cam, const orc_depth_params* dp,
                       orc_keyframe* const* kfs, int num_kfs, orc_surfels* s);
/* B/kernel_surfel_activation.cc:39-67 */
void orc_update_surfel_activation(const orc_camera* depth_cam, const orc_depth_params* dp,
                                  orc_keyframe* const* kfs, int num_kfs, uint32_t surfels_size,
                                  orc_surfels* s);
/* B/kernel_opt_geometry.cc:39-77 */
void orc_update_surfel_normals(const orc_camera* depth_cam, const orc_depth_params* dp,
                               orc_keyframe* const* kfs, int num_kfs, orc_surfels* s);
/* B/kernel_opt_geometry.cc:80-201 */
void orc_optimize_geometry_iteration(int use_depth, int use_desc, const orc_camera* color_cam,
                                     const orc_camera* depth_cam, const orc_depth_params* dp,
                                     orc_keyframe* const* kfs, int num_kfs, orc_surfels* s);

/* ---- surfel lifecycle ---- */
/* B/kernel_supporting_surfels.cc:40-165.  supporting: ORC_MERGE_BUFFER_COUNT planes of
 * height*width u32 (full resolution allocation like the reference, indexed (py/cell, px/cell)).
 * Deterministic restatement: surfels claim cells in ascending surfel index (the reference's
 * atomicCAS winner is arbitrary). */
void orc_determine_supporting_surfels(int merge, float merge_dist_factor, const orc_camera* depth_cam,
                                      const orc_depth_params* dp, const orc_keyframe* kf,
                                      orc_surfels* s, uint32_t* supporting);
/* B/direct_ba.cc:340-405 + B/kernel_create_surfels.cc:40-183.  covis: indices into kfs of the
 * keyframes co-visible with kf (only used when filter_new_surfels).  Deterministic: within a
 * sparse cell the pixel with the lowest linear index wins.  Returns the number created. */
uint32_t orc_create_surfels_for_keyframe(int filter_new_surfels, int min_observation_count,
                                         const orc_camera* color_cam, const orc_camera* depth_cam,
                                         const orc_depth_params* dp, const orc_keyframe* kf,
                                         orc_keyframe* const* kfs, const int* covis, int n_covis,
                                         orc_surfels* s, uint32_t* supporting);
/* B/kernel_delete_surfels.cc:40-120 */
void orc_delete_surfels_and_update_radii(int min_observation_count, const orc_camera* depth_cam,
                                         const orc_depth_params* dp, orc_keyframe* const* kfs,
                                         int num_kfs, orc_surfels* s);
/* B/kernel_compact_surfels.cu:159-279 */
void orc_compact_surfels(orc_surfels* s);
/* kernels_lifecycle.hip: sort_surfels_spatially (ours, not the reference's): stable Morton order over a world grid */
void orc_sort_surfels_spatially(orc_surfels* s, float grid_cell_size);

/* ---- intrinsics (B/kernel_opt_intrinsics.cc:39-281) ---- */
/* The accumulators of the intrinsics step alone, binary64, ADDED to glob[34] / cells[8 * cf_width * cf_height] (per sparse
 * cell: B0..B4, D, b2, observation count): what a surfel-sharded run sums over its ranks. */
void orc_intrinsics_accumulate(int optimize_depth_intrinsics, int optimize_color_intrinsics,
                               orc_keyframe* const* kfs, int num_kfs, const orc_camera* color_cam,
                               const orc_camera* depth_cam, const orc_depth_params* dp, const orc_surfels* s,
                               double glob[34], double* cells);
void orc_optimize_intrinsics(int optimize_depth_intrinsics, int optimize_color_intrinsics,
                             orc_keyframe* const* kfs, int num_kfs, const orc_camera* color_cam,
                             const orc_camera* depth_cam, orc_depth_params* dp, const orc_surfels* s,
                             orc_camera* out_color_cam, orc_camera* out_depth_cam, float* out_a);

/* The per-surfel sums of the normals / geometry passes are defined over 4 (default) or 8 interleaved keyframe classes
 * (oracle_geometry.c; the HIP side: bahip_context_set_sum_classes).  Process-wide. */
void orc_set_sum_classes(int classes);
int orc_get_sum_classes(void);

/* ---- BA drivers ---- */
typedef struct {
  int use_depth_residuals, use_descriptor_residuals;
  int optimize_depth_intrinsics, optimize_color_intrinsics;
  int do_surfel_updates, optimize_poses, optimize_geometry;
  int min_iterations, max_iterations;
  int window_start, window_end;
  int increase_ba_iteration_count;
  int min_observation_count;
  float surfel_merge_dist_factor;
  int pcg_max_inner_iterations;
  int pcg_gauge_keyframe;       /* -1: keyframe 0 (the reference uses rand() % K, B/direct_ba_pcg.cc:328) */
} orc_ba_options;

typedef struct {
  int iterations_done;
  int converged;
  int pose_gn_steps_total;      /* sum over keyframes and iterations of GN steps in EstimateFramePose */
  int pose_gn_rounds_max_sum;   /* sum over iterations of the max GN steps of any keyframe */
  int pcg_inner_steps_total;
} orc_ba_stats;

/* B/direct_ba_alternating.cc:285-738 (alternating scheme).  covis_lists/covis_counts give the
 * co-visibility list of each keyframe (may be NULL: treated as fully connected).
 * ba_iteration_count / last_ba_iteration_count mirror DirectBA members. */
typedef struct {
  orc_camera color_cam, depth_cam;
  orc_depth_params dp;
  orc_keyframe** kfs;
  int num_kfs;
  const int* const* covis_lists;
  const int* covis_counts;
  orc_surfels* surfels;
  uint32_t* supporting;     /* ORC_MERGE_BUFFER_COUNT * height * width */
  int ba_iteration_count, last_ba_iteration_count;
  /* Spatial order of the surfel buffer (ours, badslam_amd/host/direct_ba.cc: PerformBASchemeEndTasks; the reference has no
   * counterpart): surfels appended or moved by a compaction since the buffer was last put in Morton order, and the grid cell
   * of that order in metres (0: the end tasks never reorder).  Mirrors DirectBA::unsorted_surfels_ / spatial_sort_cell_size_. */
  uint32_t unsorted_surfels;
  float spatial_sort_cell;
} orc_ba_state;

void orc_sort_after_in_loop_compaction(orc_ba_state* st);   /* oracle_ba.c: the reorder behind the loop's compaction (ours) */
void orc_bundle_adjustment_alternating(orc_ba_state* st, const orc_ba_options* opt, orc_ba_stats* stats);
/* B/direct_ba_pcg.cc:43-819 */
void orc_bundle_adjustment_pcg(orc_ba_state* st, const orc_ba_options* opt, orc_ba_stats* stats);

/* Full cost evaluation (all residuals over all keyframe x surfel pairs, no Jacobians):
 * the "CPU cost-evaluation path" timed as cpu_baseline.  Returns the summed robust cost. */
double orc_evaluate_cost(int use_depth, int use_desc, const orc_camera* color_cam,
                         const orc_camera* depth_cam, const orc_depth_params* dp,
                         orc_keyframe* const* kfs, int num_kfs, const orc_surfels* s,
                         uint64_t* num_residuals);

int orc_num_threads(void);

/* The exactly rounded (nearest, ties to even) binary64 sum of n binary32 values; NaN if one of them is not finite.  This is
 * the definition of every dense-block sum and dot product of the PCG scheme (oracle_exact.c). */
double orc_exact_sum(const float* values, size_t n);
/* The accumulator itself: 9 int64 limbs (limb j weighs 2^(32 j - 149)); `invalid` is set when a value is not finite. */
void orc_exact_accumulate(const float* values, size_t n, long long limbs[9], int* invalid);
double orc_exact_resolve(const long long limbs[9]);

#ifdef __cplusplus
}
#endif
#endif
