"""Thin Python conveniences over the C ABI (badslam_amd.capi): pitched device buffers, keyframe
image sets and a context object.  Used by the kernel-level parity tests and by tooling; the
DirectBA-level API lives in the C++ host library (badslam_amd/host) and its binding
(badslam_amd.directba).  No arithmetic happens here and nothing falls back to the CPU.
"""
import contextlib
import ctypes as C

import numpy as np

from . import capi


class DeviceBuffer2D:
    """libvis CUDABuffer<T> semantics: (height, width) pitched device allocation."""

    def __init__(self, ctx, height, width, dtype, channels=1):
        self.ctx, self.height, self.width = ctx, int(height), int(width)
        self.dtype, self.channels = np.dtype(dtype), int(channels)
        self.elem_bytes = self.dtype.itemsize * self.channels
        ptr, pitch = C.c_void_p(), C.c_size_t()
        capi.check(ctx.lib.bahip_malloc_pitch(C.byref(ptr), C.byref(pitch), self.width * self.elem_bytes, self.height))
        self.ptr, self.pitch = ptr.value, pitch.value

    def upload(self, array):
        a = np.ascontiguousarray(array, dtype=self.dtype).reshape(self.height, self.width * self.channels)
        capi.check(self.ctx.lib.bahip_memcpy_2d(self.ctx.handle, self.ptr, self.pitch, a.ctypes.data, a.strides[0],
                                                self.width * self.elem_bytes, self.height, 1))
        return self

    def download(self):
        shape = (self.height, self.width) if self.channels == 1 else (self.height, self.width, self.channels)
        out = np.empty(shape, dtype=self.dtype)
        capi.check(self.ctx.lib.bahip_memcpy_2d(self.ctx.handle, out.ctypes.data, self.width * self.elem_bytes, self.ptr,
                                                self.pitch, self.width * self.elem_bytes, self.height, 2))
        return out

    def clear(self, byte_value=0):
        capi.check(self.ctx.lib.bahip_memset_2d(self.ctx.handle, self.ptr, self.pitch, byte_value,
                                                self.width * self.elem_bytes, self.height))
        return self

    def free(self):
        if self.ptr:
            self.ctx.lib.bahip_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    def __init__(self, stream=None):
        self.lib = capi.load()
        h = C.c_void_p()
        capi.check(self.lib.bahip_context_create(C.byref(h), stream))
        self.handle = h

    def synchronize(self):
        capi.check(self.lib.bahip_context_synchronize(self.handle))

    def set_arithmetic(self, arithmetic):
        """"exact" (default: the oracle's bits) or "fast" (hardware reciprocal / square root / exp, contraction): bahip_context_set_arithmetic."""
        mode = {"exact": capi.ARITHMETIC_EXACT, "fast": capi.ARITHMETIC_FAST}.get(arithmetic, arithmetic)
        capi.check(self.lib.bahip_context_set_arithmetic(self.handle, int(mode)))

    @property
    def arithmetic(self):
        return "fast" if self.lib.bahip_context_get_arithmetic(self.handle) == capi.ARITHMETIC_FAST else "exact"

    def close(self):
        if self.handle:
            self.lib.bahip_context_destroy(self.handle)
            self.handle = None


def bilateral_filtering_and_depth_cutoff(ctx, depth_u16, sigma_xy, sigma_value, radius_factor, max_depth, raw_to_float_depth):
    """BadSlam::PreprocessFrame's depth filter on a host image (upload, filter on the GPU, download)."""
    d = np.ascontiguousarray(depth_u16, np.uint16)
    src = DeviceBuffer2D(ctx, d.shape[0], d.shape[1], np.uint16)
    dst = DeviceBuffer2D(ctx, d.shape[0], d.shape[1], np.uint16)
    src.upload(d)
    capi.check(ctx.lib.bahip_bilateral_filtering_and_depth_cutoff(ctx.handle, sigma_xy, sigma_value, radius_factor, int(max_depth),
                                                                  raw_to_float_depth, src.ptr, src.pitch, dst.ptr, dst.pitch,
                                                                  d.shape[1], d.shape[0]))
    return dst.download()


def make_camera(params, width, height):
    p = np.asarray(params, dtype=np.float32)
    return capi.Camera(float(p[0]), float(p[1]), float(p[2]), float(p[3]), int(width), int(height))


class Scene:
    """Intrinsics + cfactor image + surfel buffer + keyframes bound to one context."""

    def __init__(self, ctx, max_surfel_count, raw_to_float_depth, baseline_fx, cell, color_cam, depth_cam):
        self.ctx, self.lib = ctx, ctx.lib
        self.color_cam, self.depth_cam = color_cam, depth_cam
        W, H = depth_cam.width, depth_cam.height
        self.cf_w, self.cf_h = (W - 1) // cell + 1, (H - 1) // cell + 1
        self.cfactor = DeviceBuffer2D(ctx, self.cf_h, self.cf_w, np.float32).clear(0)
        self.dp = capi.DepthParams(0.0, raw_to_float_depth, baseline_fx, cell, self.cfactor.ptr, self.cfactor.pitch,
                                   self.cf_w, self.cf_h)
        self.surfel_buf = DeviceBuffer2D(ctx, capi.SURFEL_ATTRIBUTE_COUNT, max_surfel_count, np.float32).clear(0)
        self.active_buf = DeviceBuffer2D(ctx, 1, max_surfel_count, np.uint8).clear(0)
        self.supporting = [DeviceBuffer2D(ctx, H, W, np.uint32) for _ in range(capi.MERGE_BUFFER_COUNT)]
        self.capacity = int(max_surfel_count)
        self.surfels_size = 0
        self.surfel_count = 0
        self.keyframes = []   # dicts: depth, normals, radius, color (DeviceBuffer2D), pose (7,), activation, min/max depth
        self.set_intrinsics()

    def set_intrinsics(self):
        capi.check(self.lib.bahip_set_intrinsics(self.ctx.handle, C.byref(self.color_cam), C.byref(self.depth_cam), C.byref(self.dp)))

    def surfels_struct(self, size=None):
        return capi.Surfels(self.surfel_buf.ptr, self.surfel_buf.pitch, self.active_buf.ptr,
                            self.surfels_size if size is None else size, self.capacity)

    # Keyframe ctor #2 (B/keyframe.cc:81-158)
    def add_keyframe(self, depth_u16, rgb_u8, global_T_frame):
        ctx, lib, h = self.ctx, self.lib, self.ctx.handle
        W, H = self.depth_cam.width, self.depth_cam.height
        cw, ch = self.color_cam.width, self.color_cam.height
        kf = dict(depth=DeviceBuffer2D(ctx, H, W, np.uint16), normals=DeviceBuffer2D(ctx, H, W, np.uint16),
                  radius=DeviceBuffer2D(ctx, H, W, np.uint16).clear(0), color=DeviceBuffer2D(ctx, ch, cw, np.uint8, 4),
                  pose=np.asarray(global_T_frame, dtype=np.float32).copy(), activation=capi.KF_ACTIVE)
        rgb = DeviceBuffer2D(ctx, ch, cw, np.uint8, 3).upload(rgb_u8)
        capi.check(lib.bahip_compute_brightness(h, rgb.ptr, rgb.pitch, kf["color"].ptr, kf["color"].pitch, cw, ch))
        raw = DeviceBuffer2D(ctx, H, W, np.uint16).upload(depth_u16)
        tmp = DeviceBuffer2D(ctx, H, W, np.uint16)
        capi.check(lib.bahip_compute_normals(h, C.byref(self.depth_cam), C.byref(self.dp), raw.ptr, raw.pitch, tmp.ptr, tmp.pitch,
                                             kf["normals"].ptr, kf["normals"].pitch))
        capi.check(lib.bahip_compute_point_radii_and_remove_isolated_pixels(
            h, C.byref(self.depth_cam), self.dp.raw_to_float_depth, tmp.ptr, tmp.pitch, kf["radius"].ptr, kf["radius"].pitch,
            kf["depth"].ptr, kf["depth"].pitch))
        mn, mx = C.c_float(), C.c_float()
        capi.check(lib.bahip_compute_min_max_depth(h, tmp.ptr, tmp.pitch, W, H, self.dp.raw_to_float_depth, C.byref(mn), C.byref(mx)))
        kf["min_depth"], kf["max_depth"] = mn.value, mx.value
        ctx.synchronize()
        for b in (rgb, raw, tmp):
            b.free()
        self.keyframes.append(kf)
        return len(self.keyframes) - 1

    def frame_struct(self, i):
        kf = self.keyframes[i]
        return capi.Frame(kf["depth"].ptr, kf["depth"].pitch, kf["normals"].ptr, kf["normals"].pitch,
                          kf["radius"].ptr, kf["radius"].pitch, kf["color"].ptr, kf["color"].pitch)

    def set_sum_classes(self, classes):
        """4 (default) or 8 interleaved keyframe classes in the per-surfel sums (bahip_context_set_sum_classes)."""
        capi.check(self.lib.bahip_context_set_sum_classes(self.ctx.handle, int(classes)))

    def set_keyframe_sharding(self, rank, world):
        """Keyframe k lives on rank k % world (bahip_context_set_keyframe_sharding); bind_keyframes then hands over the
        images of this rank's keyframes only (null pointers for the others: the backend must not look at them)."""
        capi.check(self.lib.bahip_context_set_keyframe_sharding(self.ctx.handle, int(rank), int(world)))
        self.kf_shard = (int(rank), int(world))

    def bind_keyframes(self):
        arr = (capi.Keyframe * max(1, len(self.keyframes)))()
        rank, world = getattr(self, "kf_shard", (0, 1))
        for i, kf in enumerate(self.keyframes):
            arr[i].frame = self.frame_struct(i) if i % world == rank else capi.Frame()
            for c in range(7):
                arr[i].global_T_frame[c] = float(kf["pose"][c])
            arr[i].activation = int(kf["activation"])
        capi.check(self.lib.bahip_set_keyframes(self.ctx.handle, arr, len(self.keyframes)))

    def _supporting_ptrs(self):
        return (C.c_void_p * capi.MERGE_BUFFER_COUNT)(*[b.ptr for b in self.supporting])

    def create_surfels_for_keyframe(self, i, filter_new_surfels=False, min_observation_count=2, covis=None):
        self.bind_keyframes()
        if covis is None:
            covis = [j for j in range(len(self.keyframes)) if j != i]
        cv = (C.c_int * max(1, len(covis)))(*covis)
        n = C.c_uint32()
        s = self.surfels_struct()
        capi.check(self.lib.bahip_create_surfels_for_keyframe(self.ctx.handle, i, int(filter_new_surfels), int(min_observation_count),
                                                              cv, len(covis), C.byref(s), self._supporting_ptrs(),
                                                              self.supporting[0].pitch, C.byref(n)))
        self.surfels_size += n.value
        self.surfel_count += n.value
        return n.value

    def create_surfels_for_keyframes(self, plan, filter_new_surfels=False, min_observation_count=2):
        """bahip_create_surfels_for_keyframes; plan = [(keyframe index, co-visible keyframe indices or None for all others), ...]."""
        self.bind_keyframes()
        ids, offsets, covis = [], [0], []
        for i, cv in plan:
            ids.append(i)
            covis += [j for j in range(len(self.keyframes)) if j != i] if cv is None else list(cv)
            offsets.append(len(covis))
        n = C.c_uint32()
        s = self.surfels_struct()
        capi.check(self.lib.bahip_create_surfels_for_keyframes(self.ctx.handle, (C.c_int * len(ids))(*ids), len(ids), int(filter_new_surfels),
                                                               int(min_observation_count), (C.c_int * len(offsets))(*offsets),
                                                               (C.c_int * max(1, len(covis)))(*covis), C.byref(s), self._supporting_ptrs(),
                                                               self.supporting[0].pitch, C.byref(n)))
        self.surfels_size += n.value
        self.surfel_count += n.value
        return n.value

    @contextlib.contextmanager
    def lifecycle_batch(self, keyframes=None, frames=None):
        """bahip_lifecycle_batch_begin / _end around the creations or merges of a batch of keyframes; keyframes (bound indices) or
        frames (frame_T_global 3x4 each): the batch knows its frames and keeps a list of visible tiles for each."""
        s = self.surfels_struct()
        capi.check(self.lib.bahip_lifecycle_batch_begin(self.ctx.handle, C.byref(s)))
        if keyframes is not None:
            self.bind_keyframes()
            capi.check(self.lib.bahip_lifecycle_batch_set_keyframes(self.ctx.handle, (C.c_int * max(1, len(keyframes)))(*keyframes), len(keyframes)))
        if frames is not None:
            flat = [float(v) for F in frames for v in F]
            capi.check(self.lib.bahip_lifecycle_batch_set_frames(self.ctx.handle, (C.c_float * max(1, len(flat)))(*flat), len(frames)))
        try:
            yield
        finally:
            capi.check(self.lib.bahip_lifecycle_batch_end(self.ctx.handle))

    def determine_supporting_surfels(self, i, frame_T_global, merge=False, merge_dist_factor=0.8):
        """DetermineSupportingSurfels[AndMergeSurfels]CUDA for keyframe i; returns (the three supporting planes restricted to
        the sparse-cell grid, number of surfels deleted by merging)."""
        F = (C.c_float * 12)(*[float(v) for v in frame_T_global])
        merged = C.c_uint32()
        fr, s = self.frame_struct(i), self.surfels_struct()
        capi.check(self.lib.bahip_determine_supporting_surfels(self.ctx.handle, int(merge), float(merge_dist_factor), C.byref(fr), F,
                                                               C.byref(s), self._supporting_ptrs(), self.supporting[0].pitch,
                                                               C.byref(merged)))
        self.ctx.synchronize()
        self.surfel_count -= merged.value
        planes = np.stack([b.download()[:self.cf_h, :self.cf_w] for b in self.supporting])
        return planes, merged.value

    def merge_surfels_for_keyframes(self, keyframes, frames_T_global, merge_dist_factor=0.8):
        """bahip_merge_surfels_for_keyframes: the merges of a batch of keyframes, pipelined (two dependent launches per keyframe);
        returns (the three supporting planes afterwards -- left empty --, surfels merged away)."""
        n = len(keyframes)
        structs = (capi.Frame * n)(*[self.frame_struct(k) for k in keyframes])
        F = (C.c_float * (12 * n))(*[float(v) for T in frames_T_global for v in T])
        merged = C.c_uint32()
        s = self.surfels_struct()
        capi.check(self.lib.bahip_merge_surfels_for_keyframes(self.ctx.handle, float(merge_dist_factor), structs, F, n, C.byref(s),
                                                              self._supporting_ptrs(), self.supporting[0].pitch, C.byref(merged)))
        self.ctx.synchronize()
        self.surfel_count -= merged.value
        planes = np.stack([b.download()[:self.cf_h, :self.cf_w] for b in self.supporting])
        return planes, merged.value

    def delete_surfels_and_update_radii(self, min_observation_count):
        deleted = C.c_uint32()
        s = self.surfels_struct()
        capi.check(self.lib.bahip_delete_surfels_and_update_radii(self.ctx.handle, int(min_observation_count), C.byref(s), C.byref(deleted)))
        self.surfel_count -= deleted.value
        return deleted.value

    def compact_surfels(self, with_active=True):
        s = self.surfels_struct()
        if not with_active:
            s.active = None
        capi.check(self.lib.bahip_compact_surfels(self.ctx.handle, self.surfel_count, C.byref(s)))
        self.surfels_size = self.surfel_count

    def download_surfels(self):
        self.ctx.synchronize()
        return self.surfel_buf.download()[:, :self.surfels_size]

    def upload_surfels(self, data, active=None):
        full = np.zeros((capi.SURFEL_ATTRIBUTE_COUNT, self.capacity), np.float32)
        n = data.shape[1]
        full[:data.shape[0], :n] = data
        self.surfel_buf.upload(full)
        self.surfels_size = self.surfel_count = n
        if active is not None:
            a = np.zeros((1, self.capacity), np.uint8)
            a[0, :n] = active
            self.active_buf.upload(a)

    def accumulate_pose_coeffs(self, i, use_depth, use_desc, frame_T_global):
        F = (C.c_float * 12)(*[float(v) for v in frame_T_global])
        H, b = (C.c_float * 21)(), (C.c_float * 6)()
        fr, s = self.frame_struct(i), self.surfels_struct()
        capi.check(self.lib.bahip_accumulate_pose_estimation_coeffs(self.ctx.handle, int(use_depth), int(use_desc), C.byref(fr), F,
                                                                    C.byref(s), H, b))
        return np.array(list(H)), np.array(list(b))

    def estimate_frame_pose(self, i, use_depth, use_desc, init_pose):
        init = (C.c_float * 7)(*[float(v) for v in init_pose])
        out = (C.c_float * 7)()
        its, conv = C.c_int(), C.c_int()
        fr, s = self.frame_struct(i), self.surfels_struct()
        capi.check(self.lib.bahip_estimate_frame_pose(self.ctx.handle, int(use_depth), int(use_desc), C.byref(fr), init, C.byref(s),
                                                      out, C.byref(its), C.byref(conv)))
        return np.array(list(out), dtype=np.float64), its.value, bool(conv.value)

    def update_surfel_activation(self):
        s = self.surfels_struct()
        capi.check(self.lib.bahip_update_surfel_activation(self.ctx.handle, C.byref(s), self.surfels_size))

    def assign_colors(self):
        s = self.surfels_struct()
        capi.check(self.lib.bahip_assign_colors(self.ctx.handle, C.byref(s)))

    def optimize_geometry_iteration(self, use_depth, use_desc):
        s = self.surfels_struct()
        capi.check(self.lib.bahip_optimize_geometry_iteration(self.ctx.handle, int(use_depth), int(use_desc), C.byref(s)))

    def update_activation_and_optimize_geometry(self, use_depth, use_desc, activation_surfels_size=None):
        s = self.surfels_struct()
        n = self.surfels_size if activation_surfels_size is None else int(activation_surfels_size)
        capi.check(self.lib.bahip_update_activation_and_optimize_geometry(self.ctx.handle, int(use_depth), int(use_desc), C.byref(s), n))

    def estimate_keyframe_poses(self, use_depth, use_desc):
        K = len(self.keyframes)
        poses = (C.c_float * (7 * K))()
        its, conv = (C.c_int * K)(), (C.c_int * K)()
        rounds = C.c_int()
        s = self.surfels_struct()
        capi.check(self.lib.bahip_estimate_keyframe_poses(self.ctx.handle, int(use_depth), int(use_desc), C.byref(s), poses, its, conv,
                                                          C.byref(rounds)))
        return (np.array(list(poses), dtype=np.float64).reshape(K, 7), np.array(list(its)), np.array(list(conv)), rounds.value)

    def update_surfel_normals(self):
        s = self.surfels_struct()
        capi.check(self.lib.bahip_update_surfel_normals(self.ctx.handle, C.byref(s)))

    def sort_surfels_spatially(self, grid_cell_size=0.02):
        s = self.surfels_struct()
        capi.check(self.lib.bahip_sort_surfels_spatially(self.ctx.handle, C.byref(s), float(grid_cell_size)))

    def optimize_intrinsics(self, optimize_depth, optimize_color, apply=True):
        """OptimizeIntrinsicsCUDA; with apply=True the new cameras / a are adopted like
        B/direct_ba_alternating.cc:609-619 does."""
        cc, dc, a = capi.Camera(), capi.Camera(), C.c_float()
        s = self.surfels_struct()
        capi.check(self.lib.bahip_optimize_intrinsics(self.ctx.handle, int(optimize_depth), int(optimize_color), C.byref(s),
                                                      C.byref(cc), C.byref(dc), C.byref(a)))
        if apply and self.surfels_size > 0:
            if optimize_color:
                self.color_cam = cc
            if optimize_depth:
                self.depth_cam = dc
                self.dp.a = a.value
            self.set_intrinsics()
        return cc, dc, a.value

    def pcg_iteration(self, optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=False,
                      optimize_color_intrinsics=False, use_depth=True, use_desc=True, max_inner_iterations=30, gauge_keyframe=0):
        """One outer iteration of the PCG scheme on the bound keyframes; adopts new poses / intrinsics."""
        opt = capi.PCGOptions(int(optimize_poses), int(optimize_geometry), int(optimize_depth_intrinsics),
                              int(optimize_color_intrinsics), int(use_depth), int(use_desc), int(max_inner_iterations),
                              int(gauge_keyframe))
        cc, dc, a = capi.Camera(), capi.Camera(), C.c_float()
        steps, conv = C.c_int(), C.c_int()
        s = self.surfels_struct()
        capi.check(self.lib.bahip_pcg_iteration(self.ctx.handle, C.byref(opt), C.byref(s), C.byref(cc), C.byref(dc), C.byref(a),
                                                C.byref(steps), C.byref(conv)))
        K = len(self.keyframes)
        if optimize_poses and K:
            poses = (C.c_float * (7 * K))()
            capi.check(self.lib.bahip_get_keyframe_poses(self.ctx.handle, poses, K))
            arr = np.array(list(poses), dtype=np.float32).reshape(K, 7)
            for k, kf in enumerate(self.keyframes):
                kf["pose"] = arr[k].copy()
        if optimize_color_intrinsics:
            self.color_cam = cc
        if optimize_depth_intrinsics:
            self.depth_cam = dc
            self.dp.a = a.value
        if optimize_color_intrinsics or optimize_depth_intrinsics:
            self.set_intrinsics()
        return steps.value, conv.value

    def read_pcg_vector(self, which, count, offset=0):
        out = np.zeros(count, np.float32)
        capi.check(self.lib.bahip_debug_read_pcg_vector(self.ctx.handle, int(which), int(offset), int(count),
                                                        out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def exact_sum(self, values, mode=0):
        """Exactly rounded binary64 sum of binary32 values through the device's exact accumulators (test hook)."""
        v = np.ascontiguousarray(values, np.float32)
        out = C.c_double()
        capi.check(self.lib.bahip_debug_exact_sum(self.ctx.handle, v.ctypes.data_as(C.POINTER(C.c_float)), v.size, int(mode), C.byref(out)))
        return float(out.value)

    def count_pairs(self):
        """(wave-keyframe candidates, wave-keyframe hits, associated pairs, in-image pairs) of one sweep."""
        out = (C.c_uint64 * 4)()
        s = self.surfels_struct()
        capi.check(self.lib.bahip_debug_count_pairs(self.ctx.handle, C.byref(s), out))
        return [int(v) for v in out]

    def evaluate_pairs(self, i, surfel_indices, frame_T_global):
        idx = np.ascontiguousarray(surfel_indices, dtype=np.uint32)
        out = np.zeros((len(idx), 40), np.float32)
        F = (C.c_float * 12)(*[float(v) for v in frame_T_global])
        fr, s = self.frame_struct(i), self.surfels_struct()
        capi.check(self.lib.bahip_debug_evaluate_pairs(self.ctx.handle, C.byref(fr), F, C.byref(s),
                                                       idx.ctypes.data_as(C.POINTER(C.c_uint32)), len(idx),
                                                       out.ctypes.data_as(C.POINTER(C.c_float))))
        return out
