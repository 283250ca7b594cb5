"""ctypes declarations for include/badslam_hip.h (the C ABI of the HIP backend).

There is no CPU fallback: if libbadslam_hip.so is missing or cannot be loaded this module raises,
and every entry point returns non-zero (-> BackendError) when no HIP device is present.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BADSLAM_LIB_DIR: load the backend from another build directory (A/B timing of two builds on one GPU box)
LIB_PATH = os.path.join(os.environ.get("BADSLAM_LIB_DIR") or os.path.join(_HERE, "lib"), "libbadslam_hip.so")

SURFEL_ATTRIBUTE_COUNT = 17
MERGE_BUFFER_COUNT = 3
KF_ACTIVE, KF_COVISIBLE_ACTIVE, KF_INACTIVE = 0, 1, 2
ARITHMETIC_EXACT, ARITHMETIC_FAST = 0, 1   # bahip_context_set_arithmetic

# surfel rows, applications/badslam/src/badslam/kernels.cuh:69-88
SURFEL_X, SURFEL_Y, SURFEL_Z, SURFEL_NORMAL, SURFEL_RADIUS_SQUARED, SURFEL_COLOR, SURFEL_DESCRIPTOR1, SURFEL_DESCRIPTOR2 = range(8)
SURFEL_ACCUM0 = 8


class BackendError(RuntimeError):
    pass


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class DepthParams(C.Structure):
    _fields_ = [("a", C.c_float), ("raw_to_float_depth", C.c_float), ("baseline_fx", C.c_float),
                ("sparse_surfel_cell_size", C.c_int32), ("cfactor", C.c_void_p),
                ("cfactor_pitch_bytes", C.c_uint32), ("cfactor_width", C.c_int32), ("cfactor_height", C.c_int32)]


class Frame(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("depth_pitch_bytes", C.c_uint32),
                ("normals", C.c_void_p), ("normals_pitch_bytes", C.c_uint32),
                ("radius", C.c_void_p), ("radius_pitch_bytes", C.c_uint32),
                ("color", C.c_void_p), ("color_pitch_bytes", C.c_uint32),
                ("planes", C.c_void_p)]   # NULL: the library packs the BA planes itself


class Keyframe(C.Structure):
    _fields_ = [("frame", Frame), ("global_T_frame", C.c_float * 7), ("activation", C.c_int32)]


class Surfels(C.Structure):
    _fields_ = [("data", C.c_void_p), ("pitch_bytes", C.c_uint32), ("active", C.c_void_p),
                ("surfels_size", C.c_uint32), ("capacity", C.c_uint32)]


class PCGLayout(C.Structure):
    _fields_ = [("optimize_poses", C.c_int), ("optimize_geometry", C.c_int), ("optimize_depth_intrinsics", C.c_int),
                ("optimize_color_intrinsics", C.c_int), ("use_depth_residuals", C.c_int), ("use_descriptor_residuals", C.c_int),
                ("unknown_count", C.c_uint32), ("surfel_unknown_start_index", C.c_uint32),
                ("depth_intrinsics_unknown_start_index", C.c_uint32), ("color_intrinsics_unknown_start_index", C.c_uint32)]


class AlternatingOptions(C.Structure):
    _fields_ = [("use_depth_residuals", C.c_int), ("use_descriptor_residuals", C.c_int), ("fixed_window", C.c_int),
                ("activate_in_geometry", C.c_int), ("activation_surfels_size", C.c_uint32), ("min_iterations", C.c_int),
                ("max_iterations", C.c_int)]


class PCGOptions(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("optimize_poses", "optimize_geometry", "optimize_depth_intrinsics",
                                       "optimize_color_intrinsics", "use_depth_residuals", "use_descriptor_residuals",
                                       "max_inner_iterations", "gauge_keyframe")]


SUM_F32, SUM_I64, SUM_F64 = 0, 1, 2
RCCL_UNIQUE_ID_BYTES = 128
# int fn(void* device_buffer, size_t count, int dtype, void* hip_stream, void* user)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)

# name -> (restype, argtypes); every symbol declared in include/badslam_hip.h
SIGNATURES = {
    "bahip_last_error": (C.c_char_p, []),
    "bahip_device_count": (C.c_int, []),
    "bahip_context_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p]),
    "bahip_context_destroy": (None, [C.c_void_p]),
    "bahip_context_synchronize": (C.c_int, [C.c_void_p]),
    "bahip_context_take_capacity_exceeded": (C.c_int, [C.c_void_p]),
    "bahip_context_is_sharded": (C.c_int, [C.c_void_p]),
    "bahip_context_surfels_rearranged": (C.c_int, [C.c_void_p]),
    "bahip_context_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "bahip_context_set_keyframe_sharding": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "bahip_context_set_sum_classes": (C.c_int, [C.c_void_p, C.c_int]),
    "bahip_context_set_creation_order": (C.c_int, [C.c_void_p, C.c_int]),
    "bahip_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "bahip_host_free": (C.c_int, [C.c_void_p]),
    "bahip_host_is_pinned": (C.c_int, [C.c_void_p, C.c_size_t]),
    "bahip_merge_surfels_for_keyframes": (C.c_int, [C.c_void_p, C.c_float, C.POINTER(Frame), C.POINTER(C.c_float), C.c_int, C.POINTER(Surfels),
                                                   C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]),
    "bahip_context_set_arithmetic": (C.c_int, [C.c_void_p, C.c_int]),
    "bahip_context_get_arithmetic": (C.c_int, [C.c_void_p]),
    "bahip_debug_set_tile_order": (C.c_int, [C.c_int]),
    "bahip_debug_read_tile_schedule": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_size_t]),
    "bahip_rccl_get_unique_id": (C.c_int, [C.c_char_p]),
    "bahip_context_init_rccl": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "bahip_context_count_ranks": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "bahip_malloc_pitch": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t]),
    "bahip_free": (C.c_int, [C.c_void_p]),
    "bahip_memcpy_2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]),
    "bahip_memset_2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_size_t, C.c_size_t]),
    "bahip_context_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bahip_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "bahip_stream_destroy": (C.c_int, [C.c_void_p]),
    "bahip_stream_synchronize": (C.c_int, [C.c_void_p]),
    "bahip_memcpy_2d_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]),
    "bahip_memcpy_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "bahip_memset_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "bahip_fill_2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_int, C.c_int]),
    "bahip_bilateral_filtering_and_depth_cutoff": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint16, C.c_float,
                                                              C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "bahip_compute_brightness": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int]),
    "bahip_compute_normals": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(DepthParams), C.c_void_p, C.c_uint32,
                                        C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "bahip_compute_point_radii_and_remove_isolated_pixels": (
        C.c_int, [C.c_void_p, C.POINTER(Camera), C.c_float, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "bahip_compute_min_max_depth": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_float,
                                              C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bahip_set_intrinsics": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Camera), C.POINTER(DepthParams)]),
    "bahip_set_keyframes": (C.c_int, [C.c_void_p, C.POINTER(Keyframe), C.c_int]),
    "bahip_get_keyframe_poses": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "bahip_update_surfel_activation": (C.c_int, [C.c_void_p, C.POINTER(Surfels), C.c_uint32]),
    "bahip_assign_colors": (C.c_int, [C.c_void_p, C.POINTER(Surfels)]),
    "bahip_update_surfel_normals": (C.c_int, [C.c_void_p, C.POINTER(Surfels)]),
    "bahip_optimize_geometry_iteration": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Surfels)]),
    "bahip_update_activation_and_optimize_geometry": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Surfels), C.c_uint32]),
    "bahip_accumulate_pose_estimation_coeffs": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Frame), C.POINTER(C.c_float),
                                                          C.POINTER(Surfels), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bahip_estimate_frame_pose": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Frame), C.POINTER(C.c_float),
                                            C.POINTER(Surfels), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bahip_estimate_keyframe_poses": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Surfels), C.POINTER(C.c_float),
                                                C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bahip_estimate_keyframe_poses_and_update_activation": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Surfels), C.POINTER(C.c_float),
                                                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bahip_alternating_iterations": (C.c_int, [C.c_void_p, C.POINTER(AlternatingOptions), C.POINTER(Surfels), C.POINTER(C.c_float),
                                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bahip_set_covisibility": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "bahip_propagate_covisible_activation": (C.c_int, [C.c_void_p]),
    "bahip_set_activation_window": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]),
    "bahip_apply_activation_window": (C.c_int, [C.c_void_p]),
    "bahip_determine_supporting_surfels": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.POINTER(Frame), C.POINTER(C.c_float),
                                                     C.POINTER(Surfels), C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]),
    "bahip_take_merged_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "bahip_lifecycle_batch_begin": (C.c_int, [C.c_void_p, C.POINTER(Surfels)]),
    "bahip_lifecycle_batch_end": (C.c_int, [C.c_void_p]),
    "bahip_lifecycle_batch_set_frames": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "bahip_lifecycle_batch_set_keyframes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int]),
    "bahip_create_surfels_for_keyframes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                     C.POINTER(Surfels), C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]),
    "bahip_create_surfels_for_keyframe": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int,
                                                    C.POINTER(Surfels), C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]),
    "bahip_delete_surfels_and_update_radii": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Surfels), C.POINTER(C.c_uint32)]),
    "bahip_sort_surfels_spatially": (C.c_int, [C.c_void_p, C.POINTER(Surfels), C.c_float]),
    "bahip_compact_surfels": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(Surfels)]),
    "bahip_optimize_intrinsics": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Surfels), C.POINTER(Camera),
                                            C.POINTER(Camera), C.POINTER(C.c_float)]),
    "bahip_pcg_iteration": (C.c_int, [C.c_void_p, C.POINTER(PCGOptions), C.POINTER(Surfels), C.POINTER(Camera),
                                      C.POINTER(Camera), C.POINTER(C.c_float), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bahip_pcg_begin": (C.c_int, [C.c_void_p, C.POINTER(PCGLayout), C.c_uint32]),
    "bahip_pcg_init": (C.c_int, [C.c_void_p, C.POINTER(PCGLayout), C.POINTER(Frame), C.POINTER(C.c_float), C.c_uint32, C.c_int,
                                 C.POINTER(Surfels), C.c_void_p, C.c_void_p]),
    "bahip_pcg_init2": (C.c_int, [C.c_void_p, C.POINTER(PCGLayout), C.c_uint32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "bahip_pcg_step1": (C.c_int, [C.c_void_p, C.POINTER(PCGLayout), C.POINTER(Frame), C.POINTER(C.c_float), C.c_uint32, C.c_int,
                                  C.POINTER(Surfels), C.c_void_p, C.c_void_p]),
    "bahip_pcg_step2": (C.c_int, [C.c_void_p, C.POINTER(PCGLayout), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "bahip_pcg_step3": (C.c_int, [C.c_void_p, C.POINTER(PCGLayout), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bahip_update_surfels_from_pcg_delta": (C.c_int, [C.c_void_p, C.POINTER(Surfels), C.c_int, C.c_uint32, C.c_void_p]),
    "bahip_update_cfactors_from_pcg_delta": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "bahip_debug_evaluate_pairs": (C.c_int, [C.c_void_p, C.POINTER(Frame), C.POINTER(C.c_float), C.POINTER(Surfels),
                                             C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_float)]),
    "bahip_debug_read_pcg_vector": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_float)]),
    "bahip_gather_surfel_shards": (C.c_int, [C.c_void_p, C.POINTER(Surfels), C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.POINTER(Surfels),
                                             C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "bahip_extract_surfel_shard": (C.c_int, [C.c_void_p, C.POINTER(Surfels), C.c_int, C.c_int, C.c_uint32, C.POINTER(Surfels), C.POINTER(C.c_uint32)]),
    "bahip_debug_set_intrinsics_bin_capacity": (C.c_int, [C.c_void_p, C.c_int]),
    "bahip_debug_intrinsics_bin_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "bahip_debug_set_intrinsics_slices": (C.c_int, [C.c_void_p, C.c_int]),
    "bahip_exchange_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int]),
    "bahip_debug_exact_sum": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "bahip_debug_count_pairs": (C.c_int, [C.c_void_p, C.POINTER(Surfels), C.POINTER(C.c_uint64)]),
    "bahip_debug_set_launch_shapes": (C.c_int, [C.c_int, C.c_int]),
    "bahip_debug_set_pose_form": (C.c_int, [C.c_int]),
    "bahip_debug_set_append_groups": (C.c_int, [C.c_int]),
    "bahip_debug_geometry_hybrid_launches": (C.c_int, [C.POINTER(C.c_longlong)]),
    "bahip_debug_set_creation_chain": (C.c_int, [C.c_int]),
    "bahip_debug_creation_chain_batches": (C.c_int, [C.POINTER(C.c_longlong)]),
    "bahip_debug_set_merge_cells": (C.c_int, [C.c_int]),
    "bahip_debug_merge_cells_batches": (C.c_int, [C.POINTER(C.c_longlong)]),
    "bahip_debug_set_pose_lds_items": (C.c_int, [C.c_int]),
    "bahip_debug_set_pose_lds_shape": (C.c_int, [C.c_int, C.c_int]),
    "bahip_debug_set_fused_iteration_begin": (C.c_int, [C.c_int]),
    "bahip_debug_set_intrinsics_reduce_form": (C.c_int, [C.c_int]),
    "bahip_debug_set_pose_rounds_ahead": (C.c_int, [C.c_int]),
    "bahip_debug_set_device_loop": (C.c_int, [C.c_int]),
    "bahip_debug_alternating_loop_calls": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "bahip_debug_set_pcg_lds_form": (C.c_int, [C.c_int]),
    "bahip_debug_pose_form_launches": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int]),
    "bahip_debug_pose_kernel_dispatches": (C.c_int, [C.POINTER(C.c_longlong)]),
    "bahip_debug_pcg_step1_form_launches": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "bahip_debug_pose_limbs": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_longlong)]),
    "bahip_debug_jacobian": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int]),
    "bahip_debug_read_pattern": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "bahip_debug_exact_math": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_size_t]),
    "bahip_debug_pose_step": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bahip_debug_wave_reduce": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "bahip_last_stage_time_ms": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "bahip_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "bahip_frame_planes_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "bahip_frame_planes_update": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Frame)]),
    "bahip_frame_planes_destroy": (None, [C.c_void_p]),
    "bahip_stage_work_units": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]),
}

_lib = None


def load():
    """dlopen the HIP backend and attach prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BackendError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            if os.environ.get("BADSLAM_LIB_DIR") and not hasattr(lib, name):
                continue              # an older build under A/B timing (scripts/ab_bench.sh) may lack newer entry points
            fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise BackendError(load().bahip_last_error().decode("utf-8", "replace"))
