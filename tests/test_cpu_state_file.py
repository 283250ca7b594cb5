"""The on-disk layout of the backend state is pinned by a committed file: tests/golden/state_v101.bin was
written by `ba_tum --save_state` on an MI355X (scripts/make_golden_state.py: 3 frames of 160x120, one BA call) and is parsed
here by an independent reader (tests/state_file.py) -- a change of field order, width or version shows up without a GPU.
The GPU-side round trip (save -> load -> save byte-identical, resume) is tests/test_gpu_tum_pipeline.py."""
import os

import numpy as np
import pytest

from badslam_amd import se3
from tests import common, state_file

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def state():
    return state_file.read_state(os.path.join(GOLDEN, "state_v101.bin"))


def test_every_byte_is_accounted_for(state):
    assert state["version"] == 101
    assert state["bytes_consumed"] == state["file_size"]


def test_calibration_and_parameters(state):
    scene = common.small_scene(num_keyframes=3, width=160, height=120, seed=11)     # what the generating script used
    for cam in (state["color_camera"], state["depth_camera"]):
        assert (cam["type_int"], cam["width"], cam["height"]) == (1, 160, 120)      # PinholeCamera4f
        assert np.allclose(cam["parameters"], scene.camera, rtol=1e-6)
    assert state["pyramid_level_for_color"] == 0
    assert state["sparse_surfel_cell_size"] == 4
    assert state["cfactor"].shape == ((120 - 1) // 4 + 1, (160 - 1) // 4 + 1) and not state["cfactor"].any()
    assert state["a"] == 0 and state["baseline_fx"] == 40
    assert np.isclose(state["raw_to_float_depth"], 1.0 / 5000)
    assert (state["use_depth_residuals"], state["use_descriptor_residuals"]) == (True, True)
    assert (state["min_observation_count_while_bootstrapping_1"], state["min_observation_count_while_bootstrapping_2"],
            state["min_observation_count"]) == (1, 2, 2)
    assert np.isclose(state["surfel_merge_dist_factor"], 0.8)
    assert state["ba_iteration_count"] == 1 and state["last_ba_iteration_count"] == -1   # one call that increased the count itself


def test_keyframe_table_and_poses(state):
    assert [k["id"] for k in state["keyframes"]] == [0, 1, 2]
    assert [k["frame_index"] for k in state["keyframes"]] == [0, 1, 2]
    assert all(k["activation"] in (0, 1, 2) for k in state["keyframes"])
    poses = state["frame_poses"].astype(np.float64)
    assert np.allclose(np.linalg.norm(poses[:, :4], axis=1), 1, atol=1e-5)           # unit quaternions, Sophus order
    # the trajectory ba_tum wrote for the same run (TUM order, relative to the first frame) is these poses
    written = []
    for line in open(os.path.join(GOLDEN, "state_v101.poses.txt")):
        if line.startswith("#") or not line.strip():
            continue
        v = [float(x) for x in line.split()[1:]]
        written.append(np.array([v[3], v[4], v[5], v[6], v[0], v[1], v[2]]))
    inv0 = se3.inverse(poses[0])
    for pose, w in zip(poses, written):
        rel = se3.mul(inv0, pose)
        if np.dot(rel[:4], w[:4]) < 0:
            rel[:4] = -rel[:4]
        assert np.allclose(rel, w, atol=2e-6)
    # and BA has put them back close to the ground truth (5 mm / 1 mrad perturbation before)
    scene = common.small_scene(num_keyframes=3, width=160, height=120, seed=11)
    gt0 = se3.inverse(np.asarray(scene.poses_gt[0], np.float64))
    for pose, gt in list(zip(poses, scene.poses_gt))[1:]:
        err = common.pose_error(se3.mul(gt0, np.asarray(gt, np.float64)), se3.mul(inv0, pose))
        assert np.linalg.norm(err[:3]) < 3e-3


def test_surfel_rows(state):
    n = state["surfels_size"]
    assert state["surfel_count"] == n == 1900 and state["surfels"].shape == (8, n)
    xyz, radius_sq = state["surfels"][:3], state["surfels"][4]
    assert np.isfinite(xyz).all() and (radius_sq > 0).all() and (radius_sq < 0.05).all()
    packed = state["surfels"][3].view(np.uint32)                  # 10-bit signed x, y, z (B/util_nvcc_only.cuh:51-76)
    comps = np.stack([(packed >> s) & 0x3ff for s in (0, 10, 20)]).astype(np.int32)
    comps = np.where(comps >= 512, comps - 1024, comps) / 511.0
    assert np.abs(np.linalg.norm(comps, axis=0) - 1).max() < 0.01
    assert np.isfinite(state["surfels"][6:8]).all()               # descriptors


def test_reader_rejects_what_load_state_rejects(tmp_path):
    blob = open(os.path.join(GOLDEN, "state_v101.bin"), "rb").read()
    for name, data in (("short", blob[:len(blob) // 2]), ("v1", blob[:7] + bytes([1]) + blob[8:]), ("id", b"BADSLAX" + blob[7:])):
        path = tmp_path / name
        path.write_bytes(data)
        with pytest.raises(ValueError):
            state_file.read_state(str(path))
