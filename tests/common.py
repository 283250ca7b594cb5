"""Shared builders for the parity tests: the same synthetic scene loaded into the CPU oracle
(oracle/binding.py) and into the HIP backend (badslam_amd.lowlevel over the C ABI)."""
import numpy as np

from badslam_amd import se3, synthetic
from oracle import binding as ob


def build_oracle(scene, max_surfels, use_depth=True, use_desc=True, poses=None, create_from=None, filter_new=False,
                 min_observation_count=2):
    cam = ob.make_camera(scene.camera, scene.width, scene.height)
    cam2 = ob.make_camera(scene.camera, scene.width, scene.height)   # distinct objects: tests perturb them separately
    ba = ob.OracleBA(max_surfels, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, cam, cam2,
                     use_depth_residuals=use_depth, use_descriptor_residuals=use_desc,
                     min_observation_count=min_observation_count)
    poses = scene.poses_gt if poses is None else poses
    for k in range(len(scene.depth)):
        ba.add_keyframe(scene.depth[k], scene.rgb[k], poses[k])
    for k in (range(len(scene.depth)) if create_from is None else create_from):
        ba.create_surfels_for_keyframe(k, filter_new_surfels=filter_new)
    return ba


def build_gpu(scene, max_surfels, poses=None, create_from=None, filter_new=False, min_observation_count=2, ctx=None):
    from badslam_amd import lowlevel as ll
    ctx = ctx or ll.Context()
    cam = ll.make_camera(scene.camera, scene.width, scene.height)
    cam2 = ll.make_camera(scene.camera, scene.width, scene.height)
    g = ll.Scene(ctx, max_surfels, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, cam, cam2)
    poses = scene.poses_gt if poses is None else poses
    for k in range(len(scene.depth)):
        g.add_keyframe(scene.depth[k], scene.rgb[k], poses[k])
    for k in (range(len(scene.depth)) if create_from is None else create_from):
        g.create_surfels_for_keyframe(k, filter_new_surfels=filter_new, min_observation_count=min_observation_count)
    return g


def oracle_surfels(ba):
    n = ba.surfels_size
    return ba.surfel_data[:, :n].copy(), ba.active[:n].copy()


def pose_error(a, b):
    """log(a^-1 b) as a 6-vector (float64)."""
    return se3.log(se3.mul(se3.inverse(np.asarray(a, np.float64)), np.asarray(b, np.float64)))


def small_scene(num_keyframes=4, width=320, height=240, seed=1, cell=2, **kw):
    return synthetic.make_scene(num_keyframes, width, height, seed=seed, cell=cell,
                                translation_range=kw.pop("translation_range", 1.0),
                                rotation_range=kw.pop("rotation_range", 0.4), **kw)
