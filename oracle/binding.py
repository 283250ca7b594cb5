"""ctypes binding of liboracle.so -- the CPU restatement of the reference BA path.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, from __graft_entry__.smoke() and from the
cpu_baseline leg of bench.py; the product package (badslam_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

SURFEL_ATTRS = 17
KF_ACTIVE, KF_COVIS_ACTIVE, KF_INACTIVE = 0, 1, 2


def build(force=False):
    """Compile oracle/*.c -> liboracle.so with the committed Makefile."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class DepthParams(C.Structure):
    _fields_ = [("a", C.c_float), ("raw_to_float_depth", C.c_float), ("baseline_fx", C.c_float),
                ("cell", C.c_int32), ("cfactor", C.POINTER(C.c_float)),
                ("cf_width", C.c_int32), ("cf_height", C.c_int32)]


class SE3(C.Structure):
    _fields_ = [("q", C.c_float * 4), ("t", C.c_float * 3)]

    @staticmethod
    def from_array(a):
        s = SE3()
        a = np.asarray(a, dtype=np.float32)
        for i in range(4):
            s.q[i] = a[i]
        for i in range(3):
            s.t[i] = a[4 + i]
        return s

    def to_array(self):
        return np.array(list(self.q) + list(self.t), dtype=np.float64)


class Keyframe(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32),
                ("color_width", C.c_int32), ("color_height", C.c_int32),
                ("depth", C.POINTER(C.c_uint16)), ("normals", C.POINTER(C.c_uint16)),
                ("radius", C.POINTER(C.c_uint16)), ("color", C.POINTER(C.c_uint8)),
                ("global_T_frame", SE3), ("frame_T_global", C.c_float * 12),
                ("global_R_frame", C.c_float * 9), ("activation", C.c_int32),
                ("min_depth", C.c_float), ("max_depth", C.c_float), ("id", C.c_int32),
                ("last_active_in_ba_iteration", C.c_int32), ("last_covis_in_ba_iteration", C.c_int32)]


class Surfels(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_float)), ("active", C.POINTER(C.c_uint8)),
                ("capacity", C.c_uint32), ("surfels_size", C.c_uint32), ("surfel_count", C.c_uint32)]


class PairEval(C.Structure):
    _fields_ = [("associated", C.c_int32), ("px", C.c_int32), ("py", C.c_int32), ("color_valid", C.c_int32),
                ("calibrated_depth", C.c_float),
                ("depth_residual", C.c_float), ("depth_weight", C.c_float), ("depth_inv_stddev", C.c_float),
                ("depth_jac_pose", C.c_float * 6), ("depth_jac_surfel", C.c_float),
                ("desc_residual", C.c_float * 2), ("desc_weight", C.c_float * 2),
                ("desc_jac_pose", (C.c_float * 6) * 2), ("desc_jac_surfel", C.c_float * 2),
                ("grad", C.c_float * 4)]


class BAOptions(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "use_depth_residuals", "use_descriptor_residuals", "optimize_depth_intrinsics",
        "optimize_color_intrinsics", "do_surfel_updates", "optimize_poses", "optimize_geometry",
        "min_iterations", "max_iterations", "window_start", "window_end",
        "increase_ba_iteration_count", "min_observation_count")] + [
        ("surfel_merge_dist_factor", C.c_float), ("pcg_max_inner_iterations", C.c_int),
        ("pcg_gauge_keyframe", C.c_int)]


class BAStats(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("iterations_done", "converged", "pose_gn_steps_total",
                                       "pose_gn_rounds_max_sum", "pcg_inner_steps_total")]


class BAState(C.Structure):
    _fields_ = [("color_cam", Camera), ("depth_cam", Camera), ("dp", DepthParams),
                ("kfs", C.POINTER(C.POINTER(Keyframe))), ("num_kfs", C.c_int),
                ("covis_lists", C.POINTER(C.POINTER(C.c_int))), ("covis_counts", C.POINTER(C.c_int)),
                ("surfels", C.POINTER(Surfels)), ("supporting", C.POINTER(C.c_uint32)),
                ("ba_iteration_count", C.c_int), ("last_ba_iteration_count", C.c_int),
                ("unsorted_surfels", C.c_uint32), ("spatial_sort_cell", C.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_sample_luma.restype = C.c_float
        L.orc_bilateral_filter_and_depth_cutoff.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint16, C.c_float, C.c_void_p,
                                                            C.c_int, C.c_int, C.c_void_p]
        L.orc_bilateral_filter_and_depth_cutoff.restype = None
        L.orc_sample_luma.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        L.orc_raw_to_calibrated_depth.restype = C.c_float
        L.orc_raw_to_calibrated_depth.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint16]
        L.orc_pack_normal10.restype = C.c_uint32
        L.orc_pack_normal10.argtypes = [C.c_float] * 3
        L.orc_pack_normal8.restype = C.c_uint16
        L.orc_pack_normal8.argtypes = [C.c_float] * 2
        L.orc_float_to_half.restype = C.c_uint16
        L.orc_float_to_half.argtypes = [C.c_float]
        L.orc_half_to_float.restype = C.c_float
        L.orc_half_to_float.argtypes = [C.c_uint16]
        L.orc_accumulate_pose_coeffs.restype = C.c_uint32
        L.orc_estimate_frame_pose.restype = C.c_int
        L.orc_evaluate_pair.restype = C.c_int
        L.orc_create_surfels_for_keyframe.restype = C.c_uint32
        L.orc_evaluate_cost.restype = C.c_double
        L.orc_is_scale1_pose_converged.restype = C.c_int
        L.orc_num_threads.restype = C.c_int
        _lib = L
    return _lib


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def bilateral_filter_and_depth_cutoff(depth_u16, sigma_xy, sigma_value, radius_factor, max_depth, raw_to_float_depth):
    import numpy as np
    d = np.ascontiguousarray(depth_u16, np.uint16)
    out = np.empty_like(d)
    lib().orc_bilateral_filter_and_depth_cutoff(sigma_xy, sigma_value, radius_factor, int(max_depth), raw_to_float_depth, d.ctypes.data,
                                                d.shape[1], d.shape[0], out.ctypes.data)
    return out


def exact_sum(values):
    """The exactly rounded binary64 sum of binary32 values (oracle_exact.c: the definition of the PCG scheme's dense sums)."""
    v = np.ascontiguousarray(values, np.float32)
    L = lib()
    L.orc_exact_sum.restype = C.c_double
    return float(L.orc_exact_sum(_ptr(v, C.c_float), C.c_size_t(v.size)))


def exact_limbs(values, limbs=None):
    """Adds binary32 values into 9 int64 limbs (a new accumulator if limbs is None); returns the limbs."""
    v = np.ascontiguousarray(values, np.float32)
    out = np.zeros(9, np.int64) if limbs is None else np.ascontiguousarray(limbs, np.int64).copy()
    bad = C.c_int(0)
    lib().orc_exact_accumulate(_ptr(v, C.c_float), C.c_size_t(v.size), _ptr(out, C.c_longlong), C.byref(bad))
    if bad.value:
        raise FloatingPointError("non-finite term in an exact sum")
    return out


def exact_resolve(limbs):
    L = lib()
    L.orc_exact_resolve.restype = C.c_double
    return float(L.orc_exact_resolve(_ptr(np.ascontiguousarray(limbs, np.int64), C.c_longlong)))


def make_camera(params, width, height):
    p = np.asarray(params, dtype=np.float32)
    return Camera(float(p[0]), float(p[1]), float(p[2]), float(p[3]), int(width), int(height))


# --- SE3 wrappers -------------------------------------------------------------------------------
def se3_exp(xi):
    out = SE3()
    lib().orc_se3_exp((C.c_float * 6)(*[float(v) for v in xi]), C.byref(out))
    return out


def se3_log(T):
    out = (C.c_float * 6)()
    lib().orc_se3_log(C.byref(T), out)
    return np.array(list(out), dtype=np.float64)


def se3_mul(a, b):
    out = SE3()
    lib().orc_se3_mul(C.byref(a), C.byref(b), C.byref(out))
    return out


def se3_inverse(a):
    out = SE3()
    lib().orc_se3_inverse(C.byref(a), C.byref(out))
    return out


def se3_matrix3x4(a):
    out = (C.c_float * 12)()
    lib().orc_se3_matrix3x4(C.byref(a), out)
    return np.array(list(out), dtype=np.float32)


class OracleBA:
    """Host-memory scene (keyframes + surfels + intrinsics) driven through the oracle.
    Mirrors the DirectBA usage pattern of the reference tests."""

    def __init__(self, max_surfel_count, raw_to_float_depth, baseline_fx, cell, color_cam, depth_cam,
                 use_depth_residuals=True, use_descriptor_residuals=True, surfel_merge_dist_factor=0.8,
                 min_observation_count=2):
        self.L = lib()
        self.color_cam, self.depth_cam = color_cam, depth_cam
        W, H = depth_cam.width, depth_cam.height
        self.cf_w, self.cf_h = (W - 1) // cell + 1, (H - 1) // cell + 1
        self.cfactor = np.zeros((self.cf_h, self.cf_w), dtype=np.float32)
        self.dp = DepthParams(0.0, raw_to_float_depth, baseline_fx, cell, _ptr(self.cfactor, C.c_float), self.cf_w, self.cf_h)
        self.surfel_data = np.zeros((SURFEL_ATTRS, max_surfel_count), dtype=np.float32)
        self.active = np.zeros(max_surfel_count, dtype=np.uint8)
        self.surfels = Surfels(_ptr(self.surfel_data, C.c_float), _ptr(self.active, C.c_uint8), max_surfel_count, 0, 0)
        self.supporting = np.zeros(3 * W * H, dtype=np.uint32)
        self.keyframes = []
        self._kf_arrays = []
        self.use_depth, self.use_desc = int(use_depth_residuals), int(use_descriptor_residuals)
        self.merge_factor = surfel_merge_dist_factor
        self.min_observation_count = min_observation_count
        self.ba_iteration_count, self.last_ba_iteration_count = 0, -1
        # spatial order of the surfel buffer, as vis::DirectBA keeps it: surfels appended / moved since the last Morton reorder,
        # and the grid cell PerformBASchemeEndTasks reorders with (0: never -- the reference's behaviour and the default here;
        # tests that compare with vis::DirectBA, whose default is 0.02, set it)
        self.unsorted_surfels, self.spatial_sort_cell = 0, 0.0
        self.covis = None  # list of lists, or None for fully connected

    # -- keyframes --
    def add_keyframe(self, depth_u16, rgb_u8, global_T_frame):
        W, H = self.depth_cam.width, self.depth_cam.height
        arrs = dict(depth=np.zeros((H, W), np.uint16), normals=np.zeros((H, W), np.uint16),
                    radius=np.zeros((H, W), np.uint16),
                    color=np.zeros((self.color_cam.height, self.color_cam.width, 4), np.uint8),
                    raw=np.ascontiguousarray(depth_u16, dtype=np.uint16),
                    rgb=np.ascontiguousarray(rgb_u8, dtype=np.uint8))
        kf = Keyframe()
        kf.width, kf.height = W, H
        kf.color_width, kf.color_height = self.color_cam.width, self.color_cam.height
        kf.depth, kf.normals = _ptr(arrs["depth"], C.c_uint16), _ptr(arrs["normals"], C.c_uint16)
        kf.radius, kf.color = _ptr(arrs["radius"], C.c_uint16), _ptr(arrs["color"], C.c_uint8)
        T = global_T_frame if isinstance(global_T_frame, SE3) else SE3.from_array(global_T_frame)
        self.L.orc_keyframe_from_images(C.byref(kf), C.byref(self.depth_cam), C.byref(self.dp),
                                        _ptr(arrs["raw"], C.c_uint16), _ptr(arrs["rgb"], C.c_uint8), C.byref(T))
        kf.id = len(self.keyframes)
        self.keyframes.append(kf)
        self._kf_arrays.append(arrs)
        return kf

    def add_preprocessed_keyframe(self, depth, normals, radius, color_rgba, global_T_frame, min_depth=0.0, max_depth=0.0):
        """A keyframe from already preprocessed images (what Keyframe ctor #1, B/keyframe.cc:35-79, takes): used to hand the
        oracle the very images the HIP path works on (preprocessing parity has its own tests)."""
        W, H = self.depth_cam.width, self.depth_cam.height
        arrs = dict(depth=np.ascontiguousarray(depth, np.uint16).reshape(H, W), normals=np.ascontiguousarray(normals, np.uint16).reshape(H, W),
                    radius=np.ascontiguousarray(radius, np.uint16).reshape(H, W),
                    color=np.ascontiguousarray(color_rgba, np.uint8).reshape(self.color_cam.height, self.color_cam.width, 4))
        kf = Keyframe()
        kf.width, kf.height = W, H
        kf.color_width, kf.color_height = self.color_cam.width, self.color_cam.height
        kf.depth, kf.normals = _ptr(arrs["depth"], C.c_uint16), _ptr(arrs["normals"], C.c_uint16)
        kf.radius, kf.color = _ptr(arrs["radius"], C.c_uint16), _ptr(arrs["color"], C.c_uint8)
        kf.activation = KF_ACTIVE
        kf.min_depth, kf.max_depth = float(min_depth), float(max_depth)
        kf.last_active_in_ba_iteration = kf.last_covis_in_ba_iteration = -1
        T = global_T_frame if isinstance(global_T_frame, SE3) else SE3.from_array(global_T_frame)
        self.L.orc_keyframe_set_global_T_frame(C.byref(kf), C.byref(T))
        kf.id = len(self.keyframes)
        self.keyframes.append(kf)
        self._kf_arrays.append(arrs)
        return kf

    def kf_arrays(self, i):
        return self._kf_arrays[i]

    def set_pose(self, i, global_T_frame):
        T = global_T_frame if isinstance(global_T_frame, SE3) else SE3.from_array(global_T_frame)
        self.L.orc_keyframe_set_global_T_frame(C.byref(self.keyframes[i]), C.byref(T))

    def pose(self, i):
        return self.keyframes[i].global_T_frame.to_array()

    def _kf_ptr_array(self):
        arr = (C.POINTER(Keyframe) * len(self.keyframes))()
        for i, kf in enumerate(self.keyframes):
            if i not in getattr(self, "deleted", ()):      # a deleted keyframe is a NULL entry of the list (B/direct_ba.cc:251-283)
                arr[i] = C.pointer(kf)
        return arr

    def delete_keyframe(self, i):
        self.deleted = set(getattr(self, "deleted", ())) | {int(i)}

    # -- surfels --
    @property
    def surfels_size(self):
        return int(self.surfels.surfels_size)

    def create_surfels_for_keyframe(self, i, filter_new_surfels=False, covis=None):
        kfs = self._kf_ptr_array()
        if covis is None:
            covis = [j for j in range(len(self.keyframes)) if j != i]
        cv = (C.c_int * max(1, len(covis)))(*covis)
        created = int(self.L.orc_create_surfels_for_keyframe(
            int(filter_new_surfels), int(self.min_observation_count), C.byref(self.color_cam), C.byref(self.depth_cam),
            C.byref(self.dp), C.byref(self.keyframes[i]), kfs, cv, len(covis), C.byref(self.surfels),
            _ptr(self.supporting, C.c_uint32)))
        self.unsorted_surfels += created
        return created

    def determine_supporting_surfels(self, i, merge=False):
        """orc_determine_supporting_surfels for keyframe i at its current pose; returns the three planes restricted to the
        sparse-cell grid.  With merge=True surfels are deleted (NaN x) and surfel_count drops."""
        self.L.orc_determine_supporting_surfels.restype = None
        self.L.orc_determine_supporting_surfels(int(merge), C.c_float(self.merge_factor), C.byref(self.depth_cam), C.byref(self.dp),
                                                C.byref(self.keyframes[i]), C.byref(self.surfels), _ptr(self.supporting, C.c_uint32))
        W, H = self.depth_cam.width, self.depth_cam.height
        return self.supporting.reshape(3, H, W)[:, :self.cf_h, :self.cf_w].copy()

    def delete_surfels_and_update_radii(self, min_observation_count=None):
        self.L.orc_delete_surfels_and_update_radii.restype = None
        before = int(self.surfels.surfel_count)
        self.L.orc_delete_surfels_and_update_radii(int(self.min_observation_count if min_observation_count is None else min_observation_count),
                                                   C.byref(self.depth_cam), C.byref(self.dp), self._kf_ptr_array(), len(self.keyframes),
                                                   C.byref(self.surfels))
        return before - int(self.surfels.surfel_count)

    def compact_surfels(self):
        self.L.orc_compact_surfels.restype = None
        self.L.orc_compact_surfels(C.byref(self.surfels))

    # -- pose --
    def accumulate_pose_coeffs(self, i, frame_T_global=None, accumulate_double=False):
        kf = self.keyframes[i]
        F = (C.c_float * 12)(*(list(kf.frame_T_global) if frame_T_global is None else [float(v) for v in frame_T_global]))
        H = (C.c_float * 21)()
        b = (C.c_float * 6)()
        cost = C.c_float()
        n = self.L.orc_accumulate_pose_coeffs(self.use_depth, self.use_desc, C.byref(self.color_cam), C.byref(self.depth_cam),
                                              C.byref(self.dp), C.byref(kf), F, C.byref(self.surfels), H, b, C.byref(cost),
                                              int(accumulate_double))
        return np.array(list(H)), np.array(list(b)), int(n), float(cost.value)

    def accumulate_pose_coeffs_fixed(self, i, frame_T_global=None):
        """The 27 fixed-point totals of the defined pose sum as (27, 2) int64 limb pairs: limb 0 weighs 2^-32, limb 1 weighs 1."""
        kf = self.keyframes[i]
        F = (C.c_float * 12)(*(list(kf.frame_T_global) if frame_T_global is None else [float(v) for v in frame_T_global]))
        fixed = (C.c_longlong * 54)()
        self.L.orc_accumulate_pose_coeffs_fixed.restype = C.c_uint32
        self.L.orc_accumulate_pose_coeffs_fixed(self.use_depth, self.use_desc, C.byref(self.color_cam), C.byref(self.depth_cam),
                                                C.byref(self.dp), C.byref(kf), F, C.byref(self.surfels), fixed)
        return np.array(list(fixed), dtype=np.int64).reshape(27, 2)

    @staticmethod
    def pose_limbs_value(limbs):
        """Binary64 value of (..., 2) limb pairs (carry-normalised): what H and b are rounded from."""
        L = lib()
        L.orc_pose_limbs_value.restype = C.c_double
        flat = np.asarray(limbs, np.int64).reshape(-1, 2)
        return np.array([L.orc_pose_limbs_value(C.c_longlong(int(lo)), C.c_longlong(int(hi))) for lo, hi in flat]).reshape(np.shape(limbs)[:-1])

    def pose_sum_invalid(self, reset=True):
        return bool(self.L.orc_pose_sum_invalid(int(reset)))

    def estimate_frame_pose(self, i, init):
        T = init if isinstance(init, SE3) else SE3.from_array(init)
        out = SE3()
        conv = C.c_int()
        its = self.L.orc_estimate_frame_pose(self.use_depth, self.use_desc, C.byref(self.color_cam), C.byref(self.depth_cam),
                                             C.byref(self.dp), C.byref(self.keyframes[i]), C.byref(T), C.byref(self.surfels),
                                             C.byref(out), C.byref(conv))
        return out, int(its), bool(conv.value)

    def evaluate_pair(self, i, surfel_index, frame_T_global=None):
        kf = self.keyframes[i]
        F = (C.c_float * 12)(*(list(kf.frame_T_global) if frame_T_global is None else [float(v) for v in frame_T_global]))
        out = PairEval()
        ok = self.L.orc_evaluate_pair(C.byref(self.color_cam), C.byref(self.depth_cam), C.byref(self.dp), C.byref(kf), F,
                                      C.byref(self.surfels), C.c_uint32(surfel_index), C.byref(out))
        return bool(ok), out

    # word offsets of orc_pair_eval (37 four-byte words, no padding)
    PAIR_WORDS = 37
    PAIR_FIELDS = dict(associated=(0, 1), px=(1, 1), py=(2, 1), color_valid=(3, 1), calibrated_depth=(4, 1), depth_residual=(5, 1),
                       depth_weight=(6, 1), depth_inv_stddev=(7, 1), depth_jac_pose=(8, 6), depth_jac_surfel=(14, 1),
                       desc_residual=(15, 2), desc_weight=(17, 2), desc_jac_pose=(19, 12), desc_jac_surfel=(31, 2), grad=(33, 4))

    def evaluate_pairs(self, i, surfel_indices, frame_T_global=None):
        """orc_evaluate_pair for many surfel indices against keyframe i; returns a (count, 37) uint32 array of raw
        orc_pair_eval words (view as float32 where the field is a float; PAIR_FIELDS gives (offset, length))."""
        assert C.sizeof(PairEval) == 4 * self.PAIR_WORDS
        kf = self.keyframes[i]
        F = (C.c_float * 12)(*(list(kf.frame_T_global) if frame_T_global is None else [float(v) for v in frame_T_global]))
        idx = np.ascontiguousarray(surfel_indices, dtype=np.uint32)
        out = np.zeros((len(idx), self.PAIR_WORDS), np.uint32)
        self.L.orc_evaluate_pairs.restype = None
        self.L.orc_evaluate_pairs(C.byref(self.color_cam), C.byref(self.depth_cam), C.byref(self.dp), C.byref(kf), F,
                                  C.byref(self.surfels), _ptr(idx, C.c_uint32), C.c_int(len(idx)), out.ctypes.data_as(C.c_void_p))
        return out

    # -- geometry / activation --
    def update_surfel_activation(self):
        self.L.orc_update_surfel_activation(C.byref(self.depth_cam), C.byref(self.dp), self._kf_ptr_array(),
                                            len(self.keyframes), C.c_uint32(self.surfels_size), C.byref(self.surfels))

    def assign_colors(self):
        self.L.orc_assign_colors.restype = None
        self.L.orc_assign_colors(C.byref(self.color_cam), C.byref(self.depth_cam), C.byref(self.dp), self._kf_ptr_array(),
                                 len(self.keyframes), C.byref(self.surfels))

    def optimize_geometry_iteration(self):
        self.L.orc_optimize_geometry_iteration(self.use_depth, self.use_desc, C.byref(self.color_cam), C.byref(self.depth_cam),
                                               C.byref(self.dp), self._kf_ptr_array(), len(self.keyframes),
                                               C.byref(self.surfels))

    def sort_surfels_spatially(self, grid_cell_size=0.02):
        self.L.orc_sort_surfels_spatially.argtypes = [C.c_void_p, C.c_float]
        self.L.orc_sort_surfels_spatially.restype = None
        self.L.orc_sort_surfels_spatially(C.byref(self.surfels), float(grid_cell_size))
        self.unsorted_surfels = 0

    def update_surfel_normals(self):
        self.L.orc_update_surfel_normals(C.byref(self.depth_cam), C.byref(self.dp), self._kf_ptr_array(), len(self.keyframes),
                                         C.byref(self.surfels))

    def intrinsics_accumulators(self, optimize_depth=True, optimize_color=True):
        """The binary64 accumulators of the intrinsics step for the current surfel set: (glob[34], cells[S, 8])."""
        S = self.cf_w * self.cf_h
        glob, cells = np.zeros(34, np.float64), np.zeros((S, 8), np.float64)
        self.L.orc_intrinsics_accumulate.restype = None
        self.L.orc_intrinsics_accumulate(int(optimize_depth), int(optimize_color), self._kf_ptr_array(), len(self.keyframes),
                                         C.byref(self.color_cam), C.byref(self.depth_cam), C.byref(self.dp), C.byref(self.surfels),
                                         _ptr(glob, C.c_double), _ptr(cells, C.c_double))
        return glob, cells

    def optimize_intrinsics(self, optimize_depth, optimize_color, apply=True):
        cc, dc, a = Camera(), Camera(), C.c_float()
        self.L.orc_optimize_intrinsics(int(optimize_depth), int(optimize_color), self._kf_ptr_array(), len(self.keyframes),
                                       C.byref(self.color_cam), C.byref(self.depth_cam), C.byref(self.dp), C.byref(self.surfels),
                                       C.byref(cc), C.byref(dc), C.byref(a))
        if apply and self.surfels_size > 0:
            if optimize_color:
                self.color_cam = cc
            if optimize_depth:
                self.depth_cam = dc
                self.dp.a = a.value
        return cc, dc, a.value

    def evaluate_cost(self):
        n = C.c_uint64()
        c = self.L.orc_evaluate_cost(self.use_depth, self.use_desc, C.byref(self.color_cam), C.byref(self.depth_cam),
                                     C.byref(self.dp), self._kf_ptr_array(), len(self.keyframes), C.byref(self.surfels),
                                     C.byref(n))
        return float(c), int(n.value)

    def pcg_assemble(self, optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=False,
                     optimize_color_intrinsics=False, gauge_keyframe=0):
        """r = -J^T W F and M = diag(J^T W J) of the PCG scheme for the current state (test hook); dense entries are exact sums
        of the per-tile / per-pair binary32 terms (oracle_pcg.c, oracle_exact.c)."""
        opt = BAOptions(self.use_depth, self.use_desc, int(optimize_depth_intrinsics), int(optimize_color_intrinsics), 0,
                        int(optimize_poses), int(optimize_geometry), 1, 1, 0, len(self.keyframes) - 1, 0,
                        int(self.min_observation_count), float(self.merge_factor), 30, int(gauge_keyframe))
        kfs = self._kf_ptr_array()
        st = BAState()
        st.color_cam, st.depth_cam, st.dp = self.color_cam, self.depth_cam, self.dp
        st.kfs, st.num_kfs = kfs, len(self.keyframes)
        st.surfels = C.pointer(self.surfels)
        st.supporting = _ptr(self.supporting, C.c_uint32)
        cap = 6 * len(self.keyframes) + 3 * self.surfels_size + 5 + self.cf_w * self.cf_h + 4
        r = np.zeros(cap, np.float32)
        M = np.zeros(cap, np.float32)
        fn = self.L.orc_pcg_assemble
        fn.restype = C.c_uint32
        U = fn(C.byref(st), C.byref(opt), _ptr(r, C.c_float), _ptr(M, C.c_float), C.c_uint32(cap))
        return r[:U], M[:U]

    # -- BA --
    def bundle_adjustment(self, optimize_depth_intrinsics=False, optimize_color_intrinsics=False,
                          do_surfel_updates=False, optimize_poses=True, optimize_geometry=True,
                          min_iterations=1, max_iterations=1, use_pcg=False, window_start=0, window_end=None,
                          increase_ba_iteration_count=True, pcg_max_inner_iterations=30, pcg_gauge_keyframe=-1):
        if window_end is None:
            window_end = len(self.keyframes) - 1
        opt = BAOptions(self.use_depth, self.use_desc, int(optimize_depth_intrinsics), int(optimize_color_intrinsics),
                        int(do_surfel_updates), int(optimize_poses), int(optimize_geometry), int(min_iterations),
                        int(max_iterations), int(window_start), int(window_end), int(increase_ba_iteration_count),
                        int(self.min_observation_count), float(self.merge_factor), int(pcg_max_inner_iterations),
                        int(pcg_gauge_keyframe))
        kfs = self._kf_ptr_array()
        st = BAState()
        st.color_cam, st.depth_cam, st.dp = self.color_cam, self.depth_cam, self.dp
        st.kfs, st.num_kfs = kfs, len(self.keyframes)
        keep = []
        if self.covis is not None:
            lists = (C.POINTER(C.c_int) * len(self.covis))()
            counts = (C.c_int * len(self.covis))()
            for k, l in enumerate(self.covis):
                a = (C.c_int * max(1, len(l)))(*l)
                keep.append(a)
                lists[k] = C.cast(a, C.POINTER(C.c_int))
                counts[k] = len(l)
            st.covis_lists, st.covis_counts = lists, counts
        st.surfels = C.pointer(self.surfels)
        st.supporting = _ptr(self.supporting, C.c_uint32)
        st.ba_iteration_count, st.last_ba_iteration_count = self.ba_iteration_count, self.last_ba_iteration_count
        st.unsorted_surfels, st.spatial_sort_cell = int(self.unsorted_surfels), float(self.spatial_sort_cell)
        stats = BAStats()
        fn = self.L.orc_bundle_adjustment_pcg if use_pcg else self.L.orc_bundle_adjustment_alternating
        fn(C.byref(st), C.byref(opt), C.byref(stats))
        self.color_cam, self.depth_cam = st.color_cam, st.depth_cam
        self.dp.a = st.dp.a
        self.ba_iteration_count, self.last_ba_iteration_count = st.ba_iteration_count, st.last_ba_iteration_count
        self.unsorted_surfels = int(st.unsorted_surfels)
        return stats
