/*
 * badslam_directba.h -- flat C view of the C++ host classes vis::DirectBA / vis::Keyframe
 * (badslam_amd/host/direct_ba.h, keyframe.h), for bindings that cannot consume C++ (the Python
 * test / bench harness via ctypes).  One function per public method of the reference class that
 * the BA path uses (applications/badslam/src/badslam/direct_ba.h:73-388); argument order and
 * meaning are the reference's.  Poses are 7 floats (qx qy qz qw tx ty tz), cameras 4 floats
 * (fx fy cx cy, pixel-corner convention).  Return 0 = ok.
 */
#ifndef BADSLAM_DIRECTBA_H_
#define BADSLAM_DIRECTBA_H_

#include <stddef.h>
#include <stdint.h>

#include "badslam_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dba_handle dba_handle;

/* DirectBA::DirectBA (direct_ba.h:73-88; render window and anchor pose omitted) */
dba_handle* dba_create(int max_surfel_count, float raw_to_float_depth, float baseline_fx, int sparse_surfel_cell_size,
                       float surfel_merge_dist_factor, int min_observation_count_while_bootstrapping_1,
                       int min_observation_count_while_bootstrapping_2, int min_observation_count,
                       int width, int height, const float color_camera[4], const float depth_camera[4],
                       int use_depth_residuals, int use_descriptor_residuals);
void dba_destroy(dba_handle* h);

/* Keyframe ctor #2 (keyframe.cc:81-158) + DirectBA::AddKeyframe; returns the keyframe id. */
int dba_add_keyframe(dba_handle* h, void* hip_stream, const uint16_t* depth_image, const uint8_t* rgb_image,
                     const float global_T_frame[7]);
int dba_keyframe_count(dba_handle* h);
int dba_get_keyframe_pose(dba_handle* h, int keyframe_id, float global_T_frame[7]);
int dba_set_keyframe_pose(dba_handle* h, int keyframe_id, const float global_T_frame[7]);
int dba_get_keyframe_activation(dba_handle* h, int keyframe_id);
/* Keyframe image access (tests poke depth / normals in place like the reference's do):
 * which: 0 depth u16, 1 normals u16, 2 radius u16, 3 colour rgba u8x4.  Dense host arrays. */
int dba_download_keyframe_image(dba_handle* h, void* hip_stream, int keyframe_id, int which, void* out);
int dba_upload_keyframe_image(dba_handle* h, void* hip_stream, int keyframe_id, int which, const void* in);
int dba_delete_keyframe(dba_handle* h, int keyframe_id);
/* DirectBA::MergeKeyframes(stream, loop_detector = NULL, approx_merge_count) (B/direct_ba.h:98-101, B/direct_ba.cc:251-338): deletes
 * up to approx_merge_count keyframes whose neighbours in the sequence are closest (never keyframe 0); dba_keyframe_exists: 1 while
 * the slot of that id still holds a keyframe.  dba_export_point_count: DirectBA::ExportToPointCloud (B/direct_ba.h:175), size only. */
int dba_merge_keyframes(dba_handle* h, void* hip_stream, int approx_merge_count);
int dba_keyframe_exists(dba_handle* h, int keyframe_id);
int dba_export_point_count(dba_handle* h, void* hip_stream, unsigned* count_out);

/* DirectBA::CreateSurfelsForKeyframe */
int dba_create_surfels_for_keyframe(dba_handle* h, void* hip_stream, int filter_new_surfels, int keyframe_id);
/* DirectBA::EstimateFramePose against the images of keyframe `keyframe_id` */
int dba_estimate_frame_pose(dba_handle* h, void* hip_stream, int keyframe_id, const float init[7], float out[7]);
/* DirectBA::BundleAdjustment */
int dba_bundle_adjustment(dba_handle* h, void* hip_stream, int optimize_depth_intrinsics, int optimize_color_intrinsics,
                          int do_surfel_updates, int optimize_poses, int optimize_geometry, int min_iterations,
                          int max_iterations, int use_pcg, int active_keyframe_window_start, int active_keyframe_window_end,
                          int increase_ba_iteration_count, int* iterations_done, int* converged,
                          int pcg_max_inner_iterations);

/* accessors */
uint32_t dba_surfel_count(dba_handle* h);
uint32_t dba_surfels_size(dba_handle* h);
int dba_set_surfel_count(dba_handle* h, uint32_t surfel_count, uint32_t surfels_size);
/* DirectBA::SortSurfelsSpatially (ours: Morton order of the surfel buffer over a world grid of grid_cell_size metres) */
int dba_sort_surfels_spatially(dba_handle* h, void* hip_stream, float grid_cell_size);
/* DirectBA::SetBatchedCreation (ours): 1 (default) = the surfel creations of a BA iteration and the merges of a merge pass go to the
 * backend as one batch each (bahip_create_surfels_for_keyframes, bahip_merge_surfels_for_keyframes), 0 = keyframe by keyframe with the
 * host in between.  Same surfels. */
int dba_set_batched_creation(dba_handle* h, int enabled);
/* DirectBA::SetSpatialSortCellSize: the grid cell PerformBASchemeEndTasks re-establishes that order with whenever surfels were
 * appended or moved since the last reorder (default 0.02 m; 0 = never: the reference's surfel order stays observable);
 * dba_unsorted_surfels: how many were appended / moved since then */
int dba_set_spatial_sort_cell_size(dba_handle* h, float grid_cell_size);
uint32_t dba_unsorted_surfels(dba_handle* h);
/* surfels()->Download/UploadPartAsync of `rows` attribute rows x `count` surfels starting at row 0 */
int dba_download_surfels(dba_handle* h, void* hip_stream, int rows, uint32_t count, float* out);
int dba_upload_surfels(dba_handle* h, void* hip_stream, int rows, uint32_t count, const float* in);
int dba_get_cameras(dba_handle* h, float color_camera[4], float depth_camera[4], float* a);
int dba_set_cameras(dba_handle* h, const float color_camera[4], const float depth_camera[4], float a);
int dba_cfactor_size(dba_handle* h, int* width, int* height);
int dba_download_cfactor(dba_handle* h, void* hip_stream, float* out);
int dba_clear_cfactor(dba_handle* h, void* hip_stream);
int dba_set_pcg_gauge_keyframe(dba_handle* h, int keyframe_id);
/* DirectBA::SetSurfelSharding: this object holds rank `rank`'s chunk-cyclic shard of one surfel cloud (bahip_gather_surfel_shards) */
int dba_set_surfel_sharding(dba_handle* h, int rank, int world, uint32_t chunk);
/* DirectBA::SetSumClasses: 4 (default) or 8 interleaved keyframe classes in the definition of the per-surfel sums
 * (bahip_context_set_sum_classes) */
int dba_set_sum_classes(dba_handle* h, int classes);
/* DirectBA::SetRowMajorCreation (ours): 1 = the surfels a keyframe creates are appended in the reference's row-major pixel order
 * (B/kernel_create_surfels.cu:357-390), 0 (default) = tile-major (bahip_context_set_creation_order) */
int dba_set_row_major_creation(dba_handle* h, int enabled);
/* DirectBA::SetFastArithmetic (ours): 1 = the fast arithmetic flavour of the sweeps, 0 (default) = the exact one (bahip_context_set_arithmetic) */
int dba_set_fast_arithmetic(dba_handle* h, int enabled);
/* DirectBA::SetKeyframeSharding: this object holds all surfels and the images of the keyframes k with k % world == rank
 * (bahip_context_set_keyframe_sharding; world = 1, 2, 4, or 8 after dba_set_sum_classes(h, 8)) */
int dba_set_keyframe_sharding(dba_handle* h, int rank, int world);
/* DirectBA::SetBAIterationCount / SetLastBAIterationCount (direct_ba.h:362-366) */
int dba_set_ba_iteration_counts(dba_handle* h, int ba_iteration_count, int last_ba_iteration_count);
int dba_last_stats(dba_handle* h, int* pose_rounds, int* pose_steps, int* pcg_inner_steps);
/* backend context of this DirectBA (for bahip_context_set_allreduce, bahip_set_profiling, ...) */
bahip_context* dba_backend_context(dba_handle* h);
/* Borrowed views for callers that drive individual bahip_* stages on a DirectBA-owned scene (the stage-level parity
 * tests): the images (and BA planes) of one keyframe, the surfel buffer, and DirectBA::BindScene (cameras, depth
 * parameters and the keyframe list as the backend sees them; bound index = position among the non-deleted keyframes). */
int dba_keyframe_frame(dba_handle* h, int keyframe_id, bahip_frame* out);
int dba_surfels_struct(dba_handle* h, bahip_surfels* out);
int dba_bind_scene(dba_handle* h, void* hip_stream);
/* Keyframe::co_visibility_list() (keyframe ids); returns the list length (entries beyond `capacity` are not written), -1 for a bad id. */
int dba_keyframe_covisibility(dba_handle* h, int keyframe_id, int* out, int capacity);

#ifdef __cplusplus
}
#endif
#endif
