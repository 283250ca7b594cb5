"""ctypes binding of oracle/_ref/libbadslam_ref.so: the REFERENCE's own device-math headers compiled for the host
(oracle/Makefile, oracle/ref_shim/).  Test infrastructure only (see oracle.h): tests/ use it to measure how far the oracle's
restatement is from the reference's code, pair by pair."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libbadslam_ref.so")
REFERENCE_HEADERS = "/root/reference/applications/badslam/src/badslam/cost_function.cuh"
_lib = None


class RefScene(C.Structure):
    _fields_ = [("depth_cam", C.c_float * 4), ("color_cam", C.c_float * 4), ("width", C.c_int), ("height", C.c_int),
                ("color_width", C.c_int), ("color_height", C.c_int), ("a", C.c_float), ("raw_to_float_depth", C.c_float),
                ("baseline_fx", C.c_float), ("cell", C.c_int), ("cfactor", C.POINTER(C.c_float)), ("cf_width", C.c_int),
                ("cf_height", C.c_int), ("depth", C.POINTER(C.c_uint16)), ("normals", C.POINTER(C.c_uint16)),
                ("rgba", C.POINTER(C.c_uint8)), ("frame_T_global", C.c_float * 12), ("surfel_rows", C.POINTER(C.c_float)),
                ("capacity", C.c_uint32), ("surfels_size", C.c_uint32), ("quantize_texture_weights", C.c_int)]


def available():
    return os.path.exists(LIB_PATH) or os.path.exists(REFERENCE_HEADERS)


def lib():
    """Loads the library; builds it first where the reference's sources are present (the build container)."""
    global _lib
    if _lib is None:
        if os.path.exists(REFERENCE_HEADERS):
            subprocess.check_call(["make", "-s", "-C", _HERE, "_ref/libbadslam_ref.so"])
        L = C.CDLL(LIB_PATH)
        for name in ("ref_raw_to_calibrated_depth", "ref_tukey_weight", "ref_tukey_residual", "ref_huber_weight", "ref_huber_residual", "ref_sample_luma"):
            getattr(L, name).restype = C.c_float
        L.ref_raw_to_calibrated_depth.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint16]
        for name in ("ref_tukey_weight", "ref_tukey_residual", "ref_huber_weight", "ref_huber_residual"):
            getattr(L, name).argtypes = [C.c_float, C.c_float]
        L.ref_sample_luma.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
        L.ref_image_space_normal_to_u16.restype = C.c_uint16
        L.ref_image_space_normal_to_u16.argtypes = [C.c_float, C.c_float]
        L.ref_u16_to_image_space_normal.argtypes = [C.c_uint16, C.POINTER(C.c_float)]
        L.ref_pack_surfel_normal.restype = C.c_uint32
        L.ref_pack_surfel_normal.argtypes = [C.c_float, C.c_float, C.c_float]
        L.ref_unpack_surfel_normal.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        L.ref_evaluate_pairs.restype = None
        _lib = L
    return _lib


def _scene(orc, keyframe_index, quantize_texture_weights):
    from oracle import binding as ob
    arrs = orc.kf_arrays(keyframe_index)
    kf = orc.keyframes[keyframe_index]
    sc = RefScene()
    for name, cam in (("depth_cam", orc.depth_cam), ("color_cam", orc.color_cam)):
        getattr(sc, name)[:] = [cam.fx, cam.fy, cam.cx, cam.cy]
    sc.width, sc.height = orc.depth_cam.width, orc.depth_cam.height
    sc.color_width, sc.color_height = orc.color_cam.width, orc.color_cam.height
    sc.a, sc.raw_to_float_depth, sc.baseline_fx, sc.cell = orc.dp.a, orc.dp.raw_to_float_depth, orc.dp.baseline_fx, orc.dp.cell
    sc.cfactor, sc.cf_width, sc.cf_height = ob._ptr(orc.cfactor, C.c_float), orc.cf_w, orc.cf_h
    sc.depth, sc.normals = ob._ptr(arrs["depth"], C.c_uint16), ob._ptr(arrs["normals"], C.c_uint16)
    sc.rgba = ob._ptr(arrs["color"], C.c_uint8)
    sc.frame_T_global[:] = list(kf.frame_T_global)
    sc.surfel_rows, sc.capacity, sc.surfels_size = ob._ptr(orc.surfel_data, C.c_float), orc.surfel_data.shape[1], orc.surfels_size
    sc.quantize_texture_weights = int(quantize_texture_weights)
    return sc


def evaluate_pairs(orc, keyframe_index, surfel_indices, quantize_texture_weights=False):
    """ref_evaluate_pairs on the scene an oracle.binding.OracleBA holds: the same keyframe images, cfactor image, cameras and
    surfel rows go to the reference's functions.  Returns a (count, 37) uint32 array laid out like OracleBA.evaluate_pairs."""
    from oracle import binding as ob
    L = lib()
    sc = _scene(orc, keyframe_index, quantize_texture_weights)
    idx = np.ascontiguousarray(surfel_indices, dtype=np.uint32)
    out = np.zeros((len(idx), ob.OracleBA.PAIR_WORDS), np.uint32)
    L.ref_evaluate_pairs(C.byref(sc), ob._ptr(idx, C.c_uint32), C.c_int(len(idx)), out.ctypes.data_as(C.c_void_p))
    return out


def evaluate_cost(orc, keyframe_indices=None, quantize_texture_weights=False):
    """Sum of the robust costs of every associated (surfel, keyframe) pair by the reference's functions (ref_evaluate_cost,
    OpenMP over the surfels, one keyframe per call).  Returns (cost, number of residuals)."""
    L = lib()
    L.ref_evaluate_cost.restype = C.c_double
    total, count = 0.0, 0
    for k in (range(len(orc.keyframes)) if keyframe_indices is None else keyframe_indices):
        sc = _scene(orc, k, quantize_texture_weights)
        n = C.c_ulonglong()
        total += float(L.ref_evaluate_cost(C.byref(sc), C.byref(n), int(orc.use_depth), int(orc.use_desc)))
        count += int(n.value)
    return total, count


# ---- whole kernels of the reference (oracle/ref_shim/ref_kernels.cc) ---------------------------------------------------------
class RefBaKeyframe(C.Structure):
    _fields_ = [("depth", C.POINTER(C.c_uint16)), ("normals", C.POINTER(C.c_uint16)), ("rgba", C.POINTER(C.c_uint8)),
                ("radius", C.POINTER(C.c_uint16)), ("frame_T_global", C.c_float * 12), ("global_T_frame", C.c_float * 12),
                ("global_R_frame", C.c_float * 9), ("activation", C.c_int32), ("pad", C.c_int32)]


class RefBaScene(C.Structure):
    _fields_ = [("depth_cam", C.c_float * 4), ("color_cam", C.c_float * 4), ("width", C.c_int), ("height", C.c_int),
                ("color_width", C.c_int), ("color_height", C.c_int), ("a", C.c_float), ("raw_to_float_depth", C.c_float),
                ("baseline_fx", C.c_float), ("cell", C.c_int), ("cfactor", C.POINTER(C.c_float)), ("cf_width", C.c_int),
                ("cf_height", C.c_int), ("surfel_rows", C.POINTER(C.c_float)), ("capacity", C.c_uint32), ("surfels_size", C.c_uint32),
                ("active", C.POINTER(C.c_uint8)), ("quantize_texture_weights", C.c_int), ("num_keyframes", C.c_int),
                ("keyframes", C.POINTER(RefBaKeyframe))]


class ReferenceKernels:
    """The reference's geometry-step, activation, colour-assignment and deletion kernels (B/kernel_opt_geometry.cu,
    B/kernel_surfel_activation.cu, B/kernel_assign_colors.cu, B/kernel_delete_surfels.cu, compiled for the host) on a COPY of the state an oracle.binding.OracleBA holds: same keyframe images, poses, activations, cameras,
    cfactor image; `surfel_data` (17 rows) and `active` are this object's own arrays."""

    def __init__(self, orc, quantize_texture_weights=False):
        from oracle import binding as ob
        self.L = lib()
        for name in ("ref_flag_pairs_outside_int_range", "ref_update_surfel_activation", "ref_optimize_geometry_iteration", "ref_assign_colors"):
            getattr(self.L, name).restype = None
        self.L.ref_delete_surfels_and_update_radii.restype = C.c_uint32
        self.orc = orc
        self.surfel_data = np.ascontiguousarray(orc.surfel_data.copy())
        self.active = np.ascontiguousarray(orc.active.copy())
        self.keep = []
        self.poses = [ob.SE3.from_array(kf.global_T_frame.to_array()) for kf in orc.keyframes]   # global_T_frame of every keyframe
        K = len(orc.keyframes)
        self.kfs = (RefBaKeyframe * K)()
        for k in range(K):
            arrs, kf = orc.kf_arrays(k), orc.keyframes[k]
            self.kfs[k].depth, self.kfs[k].normals = ob._ptr(arrs["depth"], C.c_uint16), ob._ptr(arrs["normals"], C.c_uint16)
            self.kfs[k].rgba = ob._ptr(arrs["color"], C.c_uint8)
            self.kfs[k].radius = ob._ptr(arrs["radius"], C.c_uint16)
            self.kfs[k].frame_T_global[:] = list(kf.frame_T_global)
            ob.lib().orc_se3_matrix3x4(C.byref(kf.global_T_frame), self.kfs[k].global_T_frame)     # the oracle's own 3x4 of the pose
            self.kfs[k].global_R_frame[:] = list(kf.global_R_frame)
            self.kfs[k].activation = int(kf.activation)
        sc = RefBaScene()
        for name, cam in (("depth_cam", orc.depth_cam), ("color_cam", orc.color_cam)):
            getattr(sc, name)[:] = [cam.fx, cam.fy, cam.cx, cam.cy]
        sc.width, sc.height = orc.depth_cam.width, orc.depth_cam.height
        sc.color_width, sc.color_height = orc.color_cam.width, orc.color_cam.height
        sc.a, sc.raw_to_float_depth, sc.baseline_fx, sc.cell = orc.dp.a, orc.dp.raw_to_float_depth, orc.dp.baseline_fx, orc.dp.cell
        self.cfactor = np.ascontiguousarray(orc.cfactor.copy())          # own copy: the intrinsics step updates it in place
        sc.cfactor, sc.cf_width, sc.cf_height = ob._ptr(self.cfactor, C.c_float), orc.cf_w, orc.cf_h
        sc.surfel_rows, sc.capacity, sc.surfels_size = ob._ptr(self.surfel_data, C.c_float), self.surfel_data.shape[1], orc.surfels_size
        sc.active = ob._ptr(self.active, C.c_uint8)
        sc.quantize_texture_weights = int(quantize_texture_weights)
        sc.num_keyframes, sc.keyframes = K, self.kfs
        self.sc = sc

    def pairs_outside_int_range(self):
        flags = np.zeros(self.surfel_data.shape[1], np.uint8)
        self.L.ref_flag_pairs_outside_int_range(C.byref(self.sc), flags.ctypes.data_as(C.c_void_p))
        return flags[:self.sc.surfels_size].astype(bool)

    def update_surfel_activation(self):
        self.L.ref_update_surfel_activation(C.byref(self.sc))

    def optimize_geometry_iteration(self, use_depth=True, use_desc=True):
        self.L.ref_optimize_geometry_iteration(C.byref(self.sc), int(use_depth), int(use_desc))

    def assign_colors(self):
        self.L.ref_assign_colors(C.byref(self.sc))

    def delete_surfels_and_update_radii(self, min_observation_count):
        return int(self.L.ref_delete_surfels_and_update_radii(C.byref(self.sc), int(min_observation_count)))

    def determine_supporting_surfels(self, keyframe_index, merge=False, merge_dist_factor=None):
        """The three supporting-surfel planes of keyframe `keyframe_index`, restricted to the sparse-cell grid (like
        OracleBA.determine_supporting_surfels), and the number of surfels a merging call marked as deleted."""
        W, H = self.sc.width, self.sc.height
        planes = np.zeros((3, H, W), np.uint32)
        self.L.ref_determine_supporting_surfels.restype = C.c_uint32
        factor = self.orc.merge_factor if merge_dist_factor is None else merge_dist_factor
        deleted = self.L.ref_determine_supporting_surfels(C.byref(self.sc), int(keyframe_index), int(merge), C.c_float(factor),
                                                          planes.ctypes.data_as(C.c_void_p))
        return planes[:, :self.orc.cf_h, :self.orc.cf_w].copy(), int(deleted)

    def create_surfels_for_keyframe(self, keyframe_index, filter_new_surfels=False, covis=None):
        """DirectBA::CreateSurfelsForKeyframe by the reference's kernels; new surfels are appended to this object's surfel_data
        behind surfels_size (which is advanced).  The relative poses of the co-visible keyframes are formed by the oracle's SE(3)
        routines -- the host-side product of B/direct_ba.cc:359-365 -- so that both sides count observations from the same matrices."""
        from oracle import binding as ob
        orc, OL = self.orc, ob.lib()
        if covis is None:
            covis = [j for j in range(len(orc.keyframes)) if j != keyframe_index]
        rel = (C.c_float * (12 * max(1, len(covis))))()
        for c, j in enumerate(covis):
            inv, prod = ob.SE3(), ob.SE3()
            OL.orc_se3_inverse(C.byref(orc.keyframes[j].global_T_frame), C.byref(inv))
            OL.orc_se3_mul(C.byref(inv), C.byref(orc.keyframes[keyframe_index].global_T_frame), C.byref(prod))
            OL.orc_se3_matrix3x4(C.byref(prod), C.cast(C.byref(rel, 48 * c), C.POINTER(C.c_float)))
        self.L.ref_create_surfels_for_keyframe.restype = C.c_uint32
        created = int(self.L.ref_create_surfels_for_keyframe(C.byref(self.sc), int(keyframe_index), int(filter_new_surfels), int(orc.min_observation_count),
                                                            len(covis), (C.c_int * max(1, len(covis)))(*covis), rel))
        self.sc.surfels_size += created
        return created

    def accumulate_pose_coeffs(self, keyframe_index, frame_T_global=None, use_depth=True, use_desc=True):
        """H (21, row-major upper triangle) and b (6) of the pose normal equations by the reference's kernel
        (B/kernel_opt_pose.cu + B/gauss_newton.cuh) over this object's surfels, at the keyframe's own pose or at `frame_T_global`
        (12 floats).  None if a surfel projects beyond the int range there."""
        F = (C.c_float * 12)(*(list(self.kfs[keyframe_index].frame_T_global) if frame_T_global is None else [float(v) for v in frame_T_global]))
        H, b = (C.c_float * 21)(), (C.c_float * 6)()
        self.L.ref_accumulate_pose_estimation_coeffs.restype = C.c_int
        rc = self.L.ref_accumulate_pose_estimation_coeffs(C.byref(self.sc), int(keyframe_index), F, int(use_depth), int(use_desc), H, b)
        return None if rc != 0 else (np.array(list(H), np.float32), np.array(list(b), np.float32))

    def pcg_assemble(self, optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=False, optimize_color_intrinsics=False,
                     gauge_keyframe=0):
        """r and M of the PCG scheme by the reference's PCGInit kernel, once per keyframe (B/direct_ba_pcg.cc:276-365), in the layout
        of OracleBA.pcg_assemble.  None if a surfel projects beyond the int range in some keyframe."""
        orc = self.orc
        cap = 6 * len(orc.keyframes) + 3 * orc.surfels_size + 5 + orc.cf_w * orc.cf_h + 4
        r, M = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        self.L.ref_pcg_assemble.restype = C.c_uint32
        U = int(self.L.ref_pcg_assemble(C.byref(self.sc), int(optimize_poses), int(optimize_geometry), int(orc.use_depth), int(orc.use_desc),
                                        int(optimize_depth_intrinsics), int(optimize_color_intrinsics), int(gauge_keyframe),
                                        r.ctypes.data_as(C.c_void_p), M.ctypes.data_as(C.c_void_p), C.c_uint32(cap)))
        return None if U == 0 else (r[:U], M[:U])

    def intrinsics_accumulators(self, optimize_depth=True, optimize_color=True):
        """(glob[34], cells[S, 8]) of the intrinsics step by the reference's accumulation kernel, once per keyframe
        (B/kernel_opt_intrinsics.cc:39-104), in the layout of OracleBA.intrinsics_accumulators (binary32 here)."""
        S = self.orc.cf_w * self.orc.cf_h
        glob, cells = np.zeros(34, np.float32), np.zeros((S, 8), np.float32)
        self.L.ref_intrinsics_accumulate.restype = C.c_int
        rc = self.L.ref_intrinsics_accumulate(C.byref(self.sc), int(optimize_depth), int(optimize_color), glob.ctypes.data_as(C.c_void_p),
                                              cells.ctypes.data_as(C.c_void_p))
        return None if rc != 0 else (glob, cells)

    def set_pose(self, keyframe_index, global_T_frame):
        """Gives keyframe `keyframe_index` the pose `global_T_frame` (7 numbers, Sophus layout, or an oracle SE3): the three derived
        matrices the kernels take -- frame_T_global, global_T_frame (3 x 4) and the rotation -- are formed by the oracle's SE(3)
        routines, as OracleBA.set_pose forms them for the oracle."""
        from oracle import binding as ob
        OL = ob.lib()
        T = global_T_frame if isinstance(global_T_frame, ob.SE3) else ob.SE3.from_array(global_T_frame)
        inverse = ob.SE3()
        OL.orc_se3_inverse(C.byref(T), C.byref(inverse))
        kf = self.kfs[keyframe_index]
        OL.orc_se3_matrix3x4(C.byref(inverse), kf.frame_T_global)
        OL.orc_se3_matrix3x4(C.byref(T), kf.global_T_frame)
        OL.orc_se3_rotation(C.byref(T), kf.global_R_frame)
        self.poses[keyframe_index] = ob.SE3.from_array(T.to_array())

    def estimate_frame_pose(self, keyframe_index, init, use_depth=True, use_desc=True, max_iterations=30):
        """DirectBA::EstimateFramePose (B/direct_ba_alternating.cc:126-244) with the reference's accumulation kernel: Gauss-Newton on
        the pose of keyframe `keyframe_index`'s images from `init` (global_T_frame, 7 numbers): H x = b solved in binary64 (the
        reference: Eigen LDLT on H.cast<double>()), T <- T * exp(-x) in binary32 by the oracle's SE(3) routines, until the step
        passes the convergence test of B/convergence_analysis.h:43-51 or max_iterations.  Returns (pose as 7 numbers, steps)."""
        from oracle import binding as ob
        OL = ob.lib()
        T = ob.SE3.from_array(init)
        for step in range(max_iterations):
            inverse, F = ob.SE3(), (C.c_float * 12)()
            OL.orc_se3_inverse(C.byref(T), C.byref(inverse))
            OL.orc_se3_matrix3x4(C.byref(inverse), F)
            H21, b = self.accumulate_pose_coeffs(keyframe_index, list(F), use_depth, use_desc)
            H = np.zeros((6, 6))
            H[np.triu_indices(6)] = H21
            H = H + np.triu(H, 1).T
            x = np.linalg.solve(H, b.astype(np.float64)).astype(np.float32)
            update, nxt = ob.SE3(), ob.SE3()
            OL.orc_se3_exp((C.c_float * 6)(*[-float(v) for v in x]), C.byref(update))
            OL.orc_se3_mul(C.byref(T), C.byref(update), C.byref(nxt))
            T = nxt
            if float(np.sum(x[:3] ** 2) + np.sum((10.0 * x[3:]) ** 2)) < 1e-6:
                return T.to_array(), step + 1
        return T.to_array(), max_iterations

    def pcg_outer_iteration(self, gauge_keyframe=0, max_inner_iterations=30, use_depth=True, use_desc=True):
        """One outer iteration of the PCG scheme over poses and geometry by the reference's kernels (ref_pcg_outer_iteration); the
        pose update T <- T * exp(delta) -- host code in the reference, B/direct_ba_pcg.cc:566-583 -- is applied here with the oracle's
        SE(3) routines.  `poses` (list of 7-vectors, global_T_frame) must be what set_pose last installed.  Returns the inner steps."""
        from oracle import binding as ob
        OL = ob.lib()
        K = self.sc.num_keyframes
        delta = (C.c_float * max(1, 6 * (K - 1)))()
        self.L.ref_pcg_outer_iteration.restype = C.c_int
        steps = int(self.L.ref_pcg_outer_iteration(C.byref(self.sc), int(use_depth), int(use_desc), int(gauge_keyframe), int(max_inner_iterations), delta))
        if steps < 0:
            return None
        for k in range(K):
            if k == gauge_keyframe:
                continue
            u = 6 * (k if k < gauge_keyframe else k - 1)
            update, nxt = ob.SE3(), ob.SE3()
            OL.orc_se3_exp((C.c_float * 6)(*[delta[u + c] for c in range(6)]), C.byref(update))
            OL.orc_se3_mul(C.byref(self.poses[k]), C.byref(update), C.byref(nxt))
            self.set_pose(k, nxt)
        return steps

    def optimize_intrinsics(self, optimize_depth=True, optimize_color=True):
        """The intrinsics step of the alternating scheme by the reference's kernels (ref_optimize_intrinsics).  Updates this object's
        cfactor image in place and returns (depth camera [fx, fy, cx, cy], colour camera, a), or None."""
        dc, cc, a = (C.c_float * 4)(), (C.c_float * 4)(), C.c_float()
        self.L.ref_optimize_intrinsics.restype = C.c_int
        rc = self.L.ref_optimize_intrinsics(C.byref(self.sc), int(optimize_depth), int(optimize_color), dc, cc, C.byref(a))
        return None if rc != 0 else (np.array(list(dc), np.float32), np.array(list(cc), np.float32), float(a.value))


# ---- the reference's preprocessing kernels and its compaction (oracle/ref_shim/ref_preprocess.cc) ---------------------------------
def bilateral_filter_and_depth_cutoff(depth_u16, sigma_xy, sigma_value, radius_factor, max_depth, raw_to_float_depth):
    """BilateralFilteringAndDepthCutoffCUDA (B/cuda_depth_processing.cu:42-128) on a dense u16 image."""
    d = np.ascontiguousarray(depth_u16, np.uint16)
    out = np.zeros_like(d)
    L = lib()
    L.ref_bilateral_filter_and_depth_cutoff.restype = None
    L.ref_bilateral_filter_and_depth_cutoff.argtypes = [C.c_float, C.c_float, C.c_float, C.c_uint16, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.ref_bilateral_filter_and_depth_cutoff(sigma_xy, sigma_value, radius_factor, int(max_depth), raw_to_float_depth, d.ctypes.data, d.shape[1], d.shape[0],
                                            out.ctypes.data)
    return out


def compute_brightness(rgb_u8):
    """ComputeBrightnessCUDA (B/cuda_image_processing.cu:165-193): H x W x 3 -> H x W x 4 with the luma in .w."""
    rgb = np.ascontiguousarray(rgb_u8, np.uint8)
    out = np.zeros(rgb.shape[:2] + (4,), np.uint8)
    L = lib()
    L.ref_compute_brightness.restype = None
    L.ref_compute_brightness.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.ref_compute_brightness(rgb.ctypes.data, rgb.shape[1], rgb.shape[0], out.ctypes.data)
    return out


def keyframe_depth_preprocessing(depth_u16, camera, a, raw_to_float_depth, baseline_fx, cell, cfactor):
    """The depth half of the Keyframe constructor (B/keyframe.cc:111-144) with the reference's own kernels.  camera: fx, fy, cx, cy
    (pixel-corner convention).  Returns dict(depth, normals, radius, depth_after_normals, min_depth, max_depth)."""
    d = np.ascontiguousarray(depth_u16, np.uint16)
    H, W = d.shape
    cf = np.ascontiguousarray(cfactor, np.float32)
    out = {name: np.zeros((H, W), np.uint16) for name in ("depth", "normals", "radius", "depth_after_normals")}
    lo, hi = C.c_float(), C.c_float()
    L = lib()
    L.ref_keyframe_depth_preprocessing.restype = None
    L.ref_keyframe_depth_preprocessing.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    cam = (C.c_float * 4)(*[float(v) for v in camera])
    L.ref_keyframe_depth_preprocessing(cam, a, raw_to_float_depth, baseline_fx, int(cell), cf.ctypes.data, cf.shape[1], cf.shape[0], d.ctypes.data, W, H,
                                       out["depth"].ctypes.data, out["normals"].ctypes.data, out["radius"].ctypes.data,
                                       out["depth_after_normals"].ctypes.data, C.byref(lo), C.byref(hi))
    out["min_depth"], out["max_depth"] = lo.value, hi.value
    return out


def compact_surfels(surfel_data, surfels_size, surfel_count, active=None):
    """CompactSurfelsCUDA (B/kernel_compact_surfels.cu:159-279) in place on a 17 x capacity float array (and the activity bytes).
    Returns the new surfels_size."""
    assert surfel_data.dtype == np.float32 and surfel_data.flags.c_contiguous and surfel_data.shape[0] == 17
    L = lib()
    L.ref_compact_surfels.restype = C.c_uint32
    L.ref_compact_surfels.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    return int(L.ref_compact_surfels(surfel_data.ctypes.data, surfel_data.shape[1], int(surfels_size), int(surfel_count),
                                     None if active is None else active.ctypes.data))


def float_to_half_bits(values):
    """The stand-in's __float2half_rn (oracle/ref_shim/cuda_runtime.h) on an array of binary32 values."""
    L = lib()
    L.ref_float_to_half_bits.restype = C.c_uint16
    L.ref_float_to_half_bits.argtypes = [C.c_float]
    return np.array([L.ref_float_to_half_bits(float(v)) for v in np.asarray(values, np.float32)], np.uint16)
