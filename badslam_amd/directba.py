"""Python view of the C++ host classes vis::DirectBA / vis::Keyframe (badslam_amd/host), through the
flat C header include/badslam_directba.h.  Method names follow the reference's DirectBA
(applications/badslam/src/badslam/direct_ba.h:73-388).  Nothing here computes: every call ends in
the HIP backend; if the libraries are missing or no GPU is present, construction raises."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
# BADSLAM_LIB_DIR: load the backend from another build directory (A/B timing of two builds on one GPU box)
HOST_LIB_PATH = os.path.join(os.environ.get("BADSLAM_LIB_DIR") or os.path.join(_HERE, "lib"), "libbadslam_host.so")

_F7 = C.c_float * 7
_F4 = C.c_float * 4
_lib = None


def _load():
    global _lib
    if _lib is None:
        capi.load()   # raises if the HIP library is missing
        if not os.path.exists(HOST_LIB_PATH):
            raise capi.BackendError(f"{HOST_LIB_PATH} not found: run __graft_entry__.build()")
        L = C.CDLL(HOST_LIB_PATH)
        L.dba_create.restype = C.c_void_p
        L.dba_create.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_int]
        L.dba_destroy.argtypes = [C.c_void_p]
        L.dba_add_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
        L.dba_keyframe_count.argtypes = [C.c_void_p]
        L.dba_get_keyframe_pose.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.dba_set_keyframe_pose.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
        L.dba_get_keyframe_activation.argtypes = [C.c_void_p, C.c_int]
        L.dba_download_keyframe_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dba_upload_keyframe_image.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dba_delete_keyframe.argtypes = [C.c_void_p, C.c_int]
        L.dba_create_surfels_for_keyframe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.dba_estimate_frame_pose.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.dba_bundle_adjustment.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 11 + [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        L.dba_surfel_count.restype = C.c_uint32
        L.dba_surfel_count.argtypes = [C.c_void_p]
        L.dba_surfels_size.restype = C.c_uint32
        L.dba_surfels_size.argtypes = [C.c_void_p]
        L.dba_set_surfel_count.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.dba_download_surfels.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        L.dba_upload_surfels.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
        L.dba_get_cameras.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.dba_set_cameras.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float]
        L.dba_cfactor_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.dba_download_cfactor.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.dba_clear_cfactor.argtypes = [C.c_void_p, C.c_void_p]
        L.dba_set_surfel_sharding.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_uint32]
        L.dba_set_keyframe_sharding.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.dba_set_pcg_gauge_keyframe.argtypes = [C.c_void_p, C.c_int]
        L.dba_set_ba_iteration_counts.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.dba_last_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.dba_backend_context.restype = C.c_void_p
        L.dba_backend_context.argtypes = [C.c_void_p]
        L.dba_keyframe_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(capi.Frame)]
        L.dba_surfels_struct.argtypes = [C.c_void_p, C.POINTER(capi.Surfels)]
        L.dba_bind_scene.argtypes = [C.c_void_p, C.c_void_p]
        L.dba_keyframe_covisibility.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        _lib = L
    return _lib


class _BackendContext:
    """Duck-types lowlevel.Context for helpers that need (lib, handle)."""

    def __init__(self, handle):
        self.lib = capi.load()
        self.handle = C.c_void_p(handle)

    def synchronize(self):
        capi.check(self.lib.bahip_context_synchronize(self.handle))


class DirectBA:
    def __init__(self, max_surfel_count, raw_to_float_depth, baseline_fx, sparse_surfel_cell_size, width, height, color_camera,
                 depth_camera, use_depth_residuals=True, use_descriptor_residuals=True, surfel_merge_dist_factor=0.8,
                 min_observation_count_while_bootstrapping_1=2, min_observation_count_while_bootstrapping_2=2,
                 min_observation_count=2, stream=None):
        self.L = _load()
        cc = _F4(*[float(v) for v in color_camera])
        dc = _F4(*[float(v) for v in depth_camera])
        self.h = self.L.dba_create(int(max_surfel_count), float(raw_to_float_depth), float(baseline_fx), int(sparse_surfel_cell_size),
                                   float(surfel_merge_dist_factor), int(min_observation_count_while_bootstrapping_1),
                                   int(min_observation_count_while_bootstrapping_2), int(min_observation_count), int(width),
                                   int(height), cc, dc, int(use_depth_residuals), int(use_descriptor_residuals))
        if not self.h:
            raise capi.BackendError("DirectBA needs a HIP device (no CPU fallback)")
        self.width, self.height, self.stream = int(width), int(height), stream
        self.capacity = int(max_surfel_count)

    def close(self):
        if self.h:
            self.L.dba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def backend_context(self):
        return _BackendContext(self.L.dba_backend_context(self.h))

    # borrowed views for driving single bahip_* stages on this scene (stage-level parity tests)
    def keyframe_frame(self, k):
        f = capi.Frame()
        assert self.L.dba_keyframe_frame(self.h, int(k), C.byref(f)) == 0
        return f

    def surfels_struct(self):
        s = capi.Surfels()
        assert self.L.dba_surfels_struct(self.h, C.byref(s)) == 0
        return s

    def keyframe_covisibility(self, k):
        cap = max(1, self.keyframe_count())
        out = (C.c_int * cap)()
        n = self.L.dba_keyframe_covisibility(self.h, int(k), out, cap)
        assert 0 <= n <= cap
        return [int(out[i]) for i in range(n)]

    def BindScene(self):
        assert self.L.dba_bind_scene(self.h, self.stream) == 0

    # -- keyframes --
    def AddKeyframe(self, depth_u16, rgb_u8, global_T_frame):
        d = np.ascontiguousarray(depth_u16, np.uint16)
        c = np.ascontiguousarray(rgb_u8, np.uint8)
        return self.L.dba_add_keyframe(self.h, self.stream, d.ctypes.data, c.ctypes.data, _F7(*[float(v) for v in global_T_frame]))

    def keyframe_count(self):
        return self.L.dba_keyframe_count(self.h)

    def keyframe_pose(self, k):
        out = _F7()
        assert self.L.dba_get_keyframe_pose(self.h, k, out) == 0
        return np.array(list(out), np.float64)

    def set_keyframe_pose(self, k, pose):
        assert self.L.dba_set_keyframe_pose(self.h, k, _F7(*[float(v) for v in pose])) == 0

    def keyframe_activation(self, k):
        return self.L.dba_get_keyframe_activation(self.h, k)

    def keyframe_image(self, k, which):
        names = {"depth": (0, np.uint16, 1), "normals": (1, np.uint16, 1), "radius": (2, np.uint16, 1), "color": (3, np.uint8, 4)}
        idx, dt, ch = names[which]
        out = np.empty((self.height, self.width) if ch == 1 else (self.height, self.width, ch), dt)
        assert self.L.dba_download_keyframe_image(self.h, self.stream, k, idx, out.ctypes.data) == 0
        return out

    def upload_keyframe_image(self, k, which, array):
        names = {"depth": (0, np.uint16), "normals": (1, np.uint16), "radius": (2, np.uint16), "color": (3, np.uint8)}
        idx, dt = names[which]
        a = np.ascontiguousarray(array, dt)
        assert self.L.dba_upload_keyframe_image(self.h, self.stream, k, idx, a.ctypes.data) == 0

    # -- surfels --
    def CreateSurfelsForKeyframe(self, k, filter_new_surfels=False):
        assert self.L.dba_create_surfels_for_keyframe(self.h, self.stream, int(filter_new_surfels), int(k)) == 0

    def surfel_count(self):
        return int(self.L.dba_surfel_count(self.h))

    def surfels_size(self):
        return int(self.L.dba_surfels_size(self.h))

    def SortSurfelsSpatially(self, grid_cell_size=0.02):
        self.L.dba_sort_surfels_spatially.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        assert self.L.dba_sort_surfels_spatially(self.h, self.stream, float(grid_cell_size)) == 0

    def SetBatchedCreation(self, enabled):
        self.L.dba_set_batched_creation.argtypes = [C.c_void_p, C.c_int]
        assert self.L.dba_set_batched_creation(self.h, int(bool(enabled))) == 0

    def SetSpatialSortCellSize(self, grid_cell_size):
        self.L.dba_set_spatial_sort_cell_size.argtypes = [C.c_void_p, C.c_float]
        assert self.L.dba_set_spatial_sort_cell_size(self.h, float(grid_cell_size)) == 0

    def unsorted_surfels(self):
        self.L.dba_unsorted_surfels.restype = C.c_uint32
        self.L.dba_unsorted_surfels.argtypes = [C.c_void_p]
        return int(self.L.dba_unsorted_surfels(self.h))

    def SetSurfelCount(self, surfel_count, surfels_size):
        self.L.dba_set_surfel_count(self.h, int(surfel_count), int(surfels_size))

    def download_surfels(self, rows=8, count=None):
        n = self.surfels_size() if count is None else int(count)
        out = np.zeros((rows, n), np.float32)
        if n:
            assert self.L.dba_download_surfels(self.h, self.stream, rows, n, out.ctypes.data) == 0
        return out

    def upload_surfels(self, data):
        a = np.ascontiguousarray(data, np.float32)
        assert self.L.dba_upload_surfels(self.h, self.stream, a.shape[0], a.shape[1], a.ctypes.data) == 0
        self.SetSurfelCount(a.shape[1], a.shape[1])

    # -- optimisation --
    def EstimateFramePose(self, k, init_pose):
        out = _F7()
        assert self.L.dba_estimate_frame_pose(self.h, self.stream, int(k), _F7(*[float(v) for v in init_pose]), out) == 0
        return np.array(list(out), np.float64)

    def BundleAdjustment(self, optimize_depth_intrinsics=False, optimize_color_intrinsics=False, do_surfel_updates=False,
                         optimize_poses=True, optimize_geometry=True, min_iterations=1, max_iterations=1, use_pcg=False,
                         active_keyframe_window_start=0, active_keyframe_window_end=None, increase_ba_iteration_count=True,
                         pcg_max_inner_iterations=30):
        if active_keyframe_window_end is None:
            active_keyframe_window_end = self.keyframe_count() - 1
        done, conv = C.c_int(), C.c_int()
        assert self.L.dba_bundle_adjustment(self.h, self.stream, int(optimize_depth_intrinsics), int(optimize_color_intrinsics),
                                            int(do_surfel_updates), int(optimize_poses), int(optimize_geometry), int(min_iterations),
                                            int(max_iterations), int(use_pcg), int(active_keyframe_window_start),
                                            int(active_keyframe_window_end), int(increase_ba_iteration_count), C.byref(done),
                                            C.byref(conv), int(pcg_max_inner_iterations)) == 0
        return done.value, bool(conv.value)

    def last_stats(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.L.dba_last_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(pose_rounds=a.value, pose_steps=b.value, pcg_inner_steps=c.value)

    # -- intrinsics --
    def cameras(self):
        cc, dc, a = _F4(), _F4(), C.c_float()
        self.L.dba_get_cameras(self.h, cc, dc, C.byref(a))
        return np.array(list(cc)), np.array(list(dc)), a.value

    def set_cameras(self, color_camera, depth_camera, a=0.0):
        self.L.dba_set_cameras(self.h, _F4(*[float(v) for v in color_camera]), _F4(*[float(v) for v in depth_camera]), float(a))

    def cfactor(self):
        w, h = C.c_int(), C.c_int()
        self.L.dba_cfactor_size(self.h, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.float32)
        self.L.dba_download_cfactor(self.h, self.stream, out.ctypes.data)
        return out

    def set_ba_iteration_counts(self, ba_iteration_count, last_ba_iteration_count):
        self.L.dba_set_ba_iteration_counts(self.h, int(ba_iteration_count), int(last_ba_iteration_count))

    def SetSurfelSharding(self, rank, world, chunk=1024):
        """This object holds rank `rank`'s chunk-cyclic shard of one cloud; lifecycle phases run on the gathered cloud."""
        assert self.L.dba_set_surfel_sharding(self.h, int(rank), int(world), int(chunk)) == 0

    def MergeKeyframes(self, approx_merge_count=10):
        self.L.dba_merge_keyframes.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        assert self.L.dba_merge_keyframes(self.h, self.stream, int(approx_merge_count)) == 0

    def keyframe_exists(self, k):
        self.L.dba_keyframe_exists.argtypes = [C.c_void_p, C.c_int]
        return bool(self.L.dba_keyframe_exists(self.h, int(k)))

    def ExportToPointCloud_count(self):
        n = C.c_uint()
        self.L.dba_export_point_count.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
        assert self.L.dba_export_point_count(self.h, self.stream, C.byref(n)) == 0
        return int(n.value)

    def SetRowMajorCreation(self, enabled):
        self.L.dba_set_row_major_creation.argtypes = [C.c_void_p, C.c_int]
        assert self.L.dba_set_row_major_creation(self.h, int(bool(enabled))) == 0

    def SetFastArithmetic(self, enabled):
        self.L.dba_set_fast_arithmetic.argtypes = [C.c_void_p, C.c_int]
        assert self.L.dba_set_fast_arithmetic(self.h, int(bool(enabled))) == 0

    def SetSumClasses(self, classes):
        self.L.dba_set_sum_classes.argtypes = [C.c_void_p, C.c_int]
        assert self.L.dba_set_sum_classes(self.h, int(classes)) == 0

    def SetKeyframeSharding(self, rank, world):
        """This object holds all surfels and sweeps the keyframes k with k % world == rank (world = 2, 4 or 8: whole keyframe classes of the per-surfel sums; alternating scheme only)."""
        assert self.L.dba_set_keyframe_sharding(self.h, int(rank), int(world)) == 0

    def set_pcg_gauge_keyframe(self, k):
        self.L.dba_set_pcg_gauge_keyframe(self.h, int(k))
