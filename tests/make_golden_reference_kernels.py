#!/usr/bin/env python3
"""Generates tests/golden/reference_kernels.npz: inputs and OUTPUTS OF THE REFERENCE'S OWN KERNELS for a small scene, stage by
stage.  The kernels are the reference's .cu files compiled for the host (oracle/_ref/libbadslam_ref.so: oracle/Makefile reads the
sources where they lie under /root/reference, oracle/ref_shim/ supplies the stand-in CUDA runtime and launcher), so this script
runs only where /root/reference exists -- the build container.  The committed file travels: tests/test_cpu_golden_reference.py
holds the oracle against it and tests/test_gpu_golden_reference.py the HIP path, on machines that have neither the reference nor
the library built from it.

Everything downstream of an input is produced by the reference's kernels from the reference's own earlier outputs (its filtered
depth feeds its keyframe preprocessing, its keyframe images feed its surfel creation, ...); the oracle is used only as a container
for arrays and for the SE(3) host arithmetic (pose -> 3 x 4 matrices), which the file stores as inputs.
Two runs of this script give the same file except for the stages in which the reference adds with binary32 atomics in arrival
order (blocks run on OpenMP threads here as they run concurrently on a GPU): the pose normal equations, the dense entries of the PCG
system and the intrinsics step differ between runs by ~1e-7 of their largest entry -- the reference's own run-to-run noise, a
hundred times below the tolerances of tests/golden_reference.py.  Everything else is reproduced bit for bit.
(The script lives under tests/ because it uses the oracle's bindings as a container: nothing outside tests/, smoke() and bench.py's
cpu_baseline leg may import oracle/.)
usage: python tests/make_golden_reference_kernels.py [out.npz]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from badslam_amd import synthetic   # noqa: E402
from oracle import binding as ob    # noqa: E402
from oracle import ref_binding as rb   # noqa: E402
from tests import golden_reference as gr   # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "reference_kernels.npz")
W, H, K, CELL = gr.WIDTH, gr.HEIGHT, gr.KEYFRAMES, gr.CELL
scene = synthetic.make_scene(K, W, H, seed=51, cell=CELL, translation_range=0.5, rotation_range=0.2)
rng = np.random.Generator(np.random.PCG64(52))
s = scene.raw_to_float_depth
fix = dict(camera=np.asarray(scene.camera, np.float32), raw_to_float_depth=np.float32(s), baseline_fx=np.float32(scene.baseline_fx),
           poses=np.asarray(scene.poses_gt, np.float32), rgb=np.stack(scene.rgb).astype(np.uint8))

# ---- inputs: noisy raw depth with holes --------------------------------------------------------------------------------------
raw = np.stack([(d.astype(np.float64) + rng.normal(0, 6, d.shape)).clip(0, 65000).astype(np.uint16) for d in scene.depth])
raw[rng.random(raw.shape) < 0.01] = 0
fix["raw"] = raw

# ---- stage 0: BilateralFilteringAndDepthCutoffCUDA ------------------------------------------------------------------------------
fix["filtered"] = np.stack([rb.bilateral_filter_and_depth_cutoff(raw[k], *gr.BILATERAL, int(gr.MAX_DEPTH_M / s), s) for k in range(K)])

# ---- stage 1: the Keyframe constructor's kernels ---------------------------------------------------------------------------------
cam = ob.make_camera(scene.camera, W, H)
cfactor = np.zeros(((H - 1) // CELL + 1, (W - 1) // CELL + 1), np.float32)
pre = [rb.keyframe_depth_preprocessing(fix["filtered"][k], [cam.fx, cam.fy, cam.cx, cam.cy], 0.0, s, scene.baseline_fx, CELL, cfactor) for k in range(K)]
for name in ("depth", "normals", "radius"):
    fix[name] = np.stack([p[name] for p in pre])
fix["min_max_depth"] = np.array([[p["min_depth"], p["max_depth"]] for p in pre], np.float32)
fix["luma"] = np.stack([rb.compute_brightness(fix["rgb"][k])[..., 3] for k in range(K)])

# the container: an oracle scene holding the REFERENCE's keyframe images
orc = gr.oracle_with_reference_images(fix, capacity=gr.CAPACITY)
fix["frame_T_global"] = np.array([list(kf.frame_T_global) for kf in orc.keyframes], np.float32)
fix["global_R_frame"] = np.array([list(kf.global_R_frame) for kf in orc.keyframes], np.float32)

# ---- stage 2: CreateSurfelsForKeyframe, keyframe after keyframe, from an empty cloud ---------------------------------------------
ref = rb.ReferenceKernels(orc)
ref.sc.surfels_size = 0
fix["created_counts"] = np.array([ref.create_surfels_for_keyframe(k, filter_new_surfels=False) for k in range(K)], np.uint32)
N = int(ref.sc.surfels_size)
assert N == int(fix["created_counts"].sum()) and N > 3000
fix["created_rows"] = ref.surfel_data[:8, :N].copy()

# ---- stage 3: activation + geometry step from a perturbed cloud; keyframe 1 co-visible, keyframe 2 inactive -----------------------
state = gr.perturbed_state(fix["created_rows"])
orc.surfel_data[:8, :N] = state
orc.surfels.surfels_size = orc.surfels.surfel_count = N
for k, activation in enumerate(gr.ACTIVATIONS):
    orc.keyframes[k].activation = activation
ref = rb.ReferenceKernels(orc)
assert not ref.pairs_outside_int_range().any()
ref.update_surfel_activation()
fix["active_flags"] = ref.active[:N].copy()
ref.optimize_geometry_iteration(True, True)
assert np.array_equal(ref.surfel_data[4:6, :N].view(np.uint32), state[4:6].view(np.uint32))   # radius and colour rows: untouched
fix["geometry_rows"] = ref.surfel_data[gr.GEOMETRY_ROWS, :N].copy()

# ---- stage 4: pose normal equations of keyframe 0 at a pose 5 mm / 1 mrad off -----------------------------------------------------
off = synthetic.perturb_pose(np.random.Generator(np.random.PCG64(53)), scene.poses_gt[0])
T, inverse, F = ob.SE3.from_array(off), ob.SE3(), (C.c_float * 12)()
ob.lib().orc_se3_inverse(C.byref(T), C.byref(inverse))
ob.lib().orc_se3_matrix3x4(C.byref(inverse), F)
fix["pose_frame_T_global"] = np.array(list(F), np.float32)
ref = rb.ReferenceKernels(orc)                         # the perturbed cloud again
for name, flags in (("both", (True, True)), ("depth", (True, False)), ("desc", (False, True))):
    Hm, b = ref.accumulate_pose_coeffs(0, list(F), *flags)
    fix["pose_H_" + name], fix["pose_b_" + name] = Hm, b

# ---- stage 5: deletion + radius update (every keyframe counts), then compaction ---------------------------------------------------
orc.surfel_data[:8, :N] = gr.state_for_deletion(state)
ref = rb.ReferenceKernels(orc)
deleted = ref.delete_surfels_and_update_radii(gr.MIN_OBSERVATIONS)
mask = ref.surfel_data[0, :N].view(np.uint32) == 0x7fffffff
assert deleted == int(mask.sum()) and 100 < deleted < N // 2
fix["deleted_mask"] = mask
fix["radius_row"] = ref.surfel_data[4, :N].copy()
rows, active = ref.surfel_data.copy(), (np.arange(ref.surfel_data.shape[1]) % 3 == 0).astype(np.uint8)
assert rb.compact_surfels(rows, N, N - deleted, active) == N - deleted
fix["compacted_digest"] = gr.digest(rows[:8, :N - deleted], active[:N - deleted])   # pure data movement: a digest of the result is enough

# ---- stage 6: creation with the observation filter (co-visible keyframes must confirm a new surfel) -------------------------------
for k in range(K):
    orc.keyframes[k].activation = ob.KF_ACTIVE
orc.surfel_data[:] = 0
orc.surfels.surfels_size = orc.surfels.surfel_count = 0
ref = rb.ReferenceKernels(orc)
fix["filtered_created_counts"] = np.array([ref.create_surfels_for_keyframe(k, filter_new_surfels=True) for k in range(K)], np.uint32)
M = int(ref.sc.surfels_size)
assert 500 < M < N
fix["filtered_created_rows"] = ref.surfel_data[:8, :M].copy()

# ---- stage 7: colour assignment on the perturbed cloud ----------------------------------------------------------------------------
orc.surfel_data[:8, :N] = state
orc.surfels.surfels_size = orc.surfels.surfel_count = N
ref = rb.ReferenceKernels(orc)
ref.assign_colors()
fix["assigned_colours"] = ref.surfel_data[5, :N].view(np.uint32).copy()

# ---- stage 8: supporting surfels of keyframe 1 over the created cloud, without and with merging -------------------------------------
orc.surfel_data[:8, :N] = fix["created_rows"]
ref = rb.ReferenceKernels(orc)
fix["supporting_planes"], none = ref.determine_supporting_surfels(1, merge=False)
assert none == 0
ref = rb.ReferenceKernels(orc)
fix["merge_planes"], merged = ref.determine_supporting_surfels(1, merge=True)
fix["merged_mask"] = ref.surfel_data[0, :N].view(np.uint32) == 0x7fffffff
assert merged == int(fix["merged_mask"].sum()) and merged > 100

# ---- stage 9: the PCG system (r, M) over poses, surfels and both sets of intrinsics at perturbed poses, gauge keyframe 1 ------------
orc.surfel_data[:8, :N] = state
pose_rng = np.random.Generator(np.random.PCG64(54))
fix["pcg_poses"] = np.asarray([synthetic.perturb_pose(pose_rng, T, 0.002, 0.0005) for T in scene.poses_gt], np.float32)
for k in range(K):
    orc.set_pose(k, fix["pcg_poses"][k])
fix["pcg_frame_T_global"] = np.array([list(kf.frame_T_global) for kf in orc.keyframes], np.float32)
orc.use_depth = orc.use_desc = 1
fix["pcg_r"], fix["pcg_M"] = rb.ReferenceKernels(orc).pcg_assemble(True, True, True, True, gauge_keyframe=gr.GAUGE_KEYFRAME)
for k in range(K):
    orc.set_pose(k, fix["poses"][k])

# ---- stage 10: the intrinsics step of the alternating scheme from a miscalibrated state ---------------------------------------------
gr.miscalibrate(orc)
ref = rb.ReferenceKernels(orc)
depth_camera, colour_camera, a = ref.optimize_intrinsics(True, True)
fix["intrinsics_depth_camera"], fix["intrinsics_colour_camera"], fix["intrinsics_a"] = depth_camera, colour_camera, np.float32(a)
fix["intrinsics_cfactor"] = ref.cfactor.copy()

# ---- stage 11: two alternating BA iterations end to end (activation, geometry step, Gauss-Newton pose estimation of every keyframe) ---
orc2 = gr.oracle_with_reference_images(fix, capacity=gr.CAPACITY)      # calibrated cameras again, every keyframe active
orc2.surfel_data[:8, :N] = state
orc2.surfels.surfels_size = orc2.surfels.surfel_count = N
for k in range(K):
    orc2.set_pose(k, fix["pcg_poses"][k])
ref = rb.ReferenceKernels(orc2)
poses = [np.asarray(T, np.float64) for T in fix["pcg_poses"]]
steps = 0
for iteration in range(gr.ALTERNATING_ITERATIONS):
    ref.update_surfel_activation()
    ref.optimize_geometry_iteration(True, True)
    for k in range(K):
        poses[k], n = ref.estimate_frame_pose(k, poses[k])
        steps += n
    for k in range(K):                                                   # poses take effect after the phase
        ref.set_pose(k, poses[k])
fix["alternating_poses"] = np.asarray(poses, np.float64)
fix["alternating_positions"] = ref.surfel_data[:3, :N].copy()
fix["alternating_gn_steps"] = np.int32(steps)

np.savez_compressed(out_path, **fix)
print(out_path, os.path.getsize(out_path), "bytes;", N, "surfels created", fix["created_counts"].tolist(), "; active", int(fix["active_flags"].sum()),
      "; deleted", deleted)
