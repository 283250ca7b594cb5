"""Prints how close the HIP path is to the oracle on a small scene (used to set test thresholds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import common

sc = common.small_scene(num_keyframes=4, seed=3)
ba = common.build_oracle(sc, 400000)
g = common.build_gpu(sc, 400000)
for k in range(4):
    a = ba.kf_arrays(k); kf = g.keyframes[k]
    print(k, "depth eq", np.array_equal(kf["depth"].download(), a["depth"]),
          "normals diff", np.count_nonzero(kf["normals"].download() != a["normals"]),
          "radius diff", np.count_nonzero(kf["radius"].download() != a["radius"]),
          "color eq", np.array_equal(kf["color"].download(), a["color"]))
ref, _ = common.oracle_surfels(ba)
got = g.download_surfels()
print("surfels", got.shape, ref.shape)
for r in range(8):
    print(" row", r, "bit-diff", np.count_nonzero(got[r].view(np.uint32) != ref[r].view(np.uint32)))

data, active = common.oracle_surfels(ba)
rng = np.random.Generator(np.random.PCG64(5))
data[2] += rng.uniform(0, 0.004, data.shape[1]).astype(np.float32)
data[6] += 3.0
ba.surfel_data[:, :data.shape[1]] = data
g.upload_surfels(data, active * 0)
g.bind_keyframes()
g.update_surfel_activation(); ba.update_surfel_activation()
print("activation diff", np.count_nonzero(g.active_buf.download()[0, :data.shape[1]] != ba.active[:data.shape[1]]))
for (ud, us) in [(True, True), (True, False), (False, True)]:
    ba.surfel_data[:, :data.shape[1]] = data
    g.upload_surfels(data, ba.active[:data.shape[1]])
    ba.use_depth, ba.use_desc = int(ud), int(us)
    g.optimize_geometry_iteration(ud, us); ba.optimize_geometry_iteration()
    got = g.download_surfels(); ref = ba.surfel_data[:, :data.shape[1]]
    print("geometry", ud, us, "bit-diff rows", [int(np.count_nonzero(got[r].view(np.uint32) != ref[r].view(np.uint32))) for r in range(8)],
          "max dpos", np.abs(got[:3] - ref[:3]).max())

rng = np.random.Generator(np.random.PCG64(11))
ba = common.build_oracle(sc, 400000)
data, active = common.oracle_surfels(ba)
g.upload_surfels(data, active)
pert = [common.synthetic.perturb_pose(rng, T) for T in sc.poses_gt]
for k, T in enumerate(pert):
    ba.set_pose(k, T); g.keyframes[k]["pose"] = np.asarray(T, np.float32)
g.bind_keyframes()
ba.use_depth, ba.use_desc = 1, 1
poses, its, conv, rounds = g.estimate_keyframe_poses(True, True)
for k in range(4):
    est, its_ref, conv_ref = ba.estimate_frame_pose(k, pert[k])
    print("pose", k, "its", its[k], its_ref, "err vs oracle", np.abs(common.pose_error(est.to_array(), poses[k])).max(),
          "err vs gt", np.abs(common.pose_error(sc.poses_gt[k], poses[k])).max())
