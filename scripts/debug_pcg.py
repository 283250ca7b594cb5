import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import common
from tests.test_gpu_intrinsics_pcg_vs_oracle import _perturbed_pair, _cam_tuple

np.set_printoptions(linewidth=200, precision=6, suppress=True)
scene = common.small_scene(num_keyframes=5, seed=21)
for (poses_on, di, ci, off) in [(True, True, False, (0.3, -0.2, 0.5, -0.4)), (True, False, True, (0, 0, 0, 0)), (False, True, False, (0.3, -0.2, 0.5, -0.4))]:
    rng = np.random.Generator(np.random.PCG64(33))
    ba, g = _perturbed_pair(scene, depth_cam_offset=off)
    data, _ = common.oracle_surfels(ba)
    data[2] += rng.uniform(0, 0.003, data.shape[1]).astype(np.float32)
    ba.surfel_data[:, :data.shape[1]] = data
    g.upload_surfels(data, np.ones(data.shape[1], np.uint8))
    perturbed = [common.synthetic.perturb_pose(rng, T, 0.003, 0.0005) for T in scene.poses_gt]
    for k, T in enumerate(perturbed):
        ba.set_pose(k, T)
        g.keyframes[k]["pose"] = np.asarray(T, np.float32)
    g.bind_keyframes()
    ba.use_depth, ba.use_desc = 1, 1
    ba.last_ba_iteration_count = ba.ba_iteration_count
    stats = ba.bundle_adjustment(optimize_depth_intrinsics=di, optimize_color_intrinsics=ci, optimize_poses=poses_on,
                                 optimize_geometry=True, min_iterations=1, max_iterations=1, use_pcg=True,
                                 increase_ba_iteration_count=False, pcg_gauge_keyframe=0)
    g.active_buf.upload(np.ones((1, g.capacity), np.uint8))
    g.update_surfel_normals()
    steps, conv = g.pcg_iteration(optimize_poses=poses_on, optimize_geometry=True, optimize_depth_intrinsics=di,
                                  optimize_color_intrinsics=ci, gauge_keyframe=0)
    got = g.download_surfels(); ref = ba.surfel_data[:, :got.shape[1]]
    dpos = np.abs(got[:3] - ref[:3]).max(axis=0)
    print(f"poses={poses_on} di={di} ci={ci}: steps gpu {steps} oracle {stats.pcg_inner_steps_total}; dpos quantiles", np.quantile(dpos, [0.5, 0.99, 1.0]))
    print("  depth cam gpu", _cam_tuple(g.depth_cam), "oracle", _cam_tuple(ba.depth_cam), "a", g.dp.a, ba.dp.a)
    print("  color cam gpu", _cam_tuple(g.color_cam), "oracle", _cam_tuple(ba.color_cam))
    cf_g = g.cfactor.download(); cf_r = ba.cfactor
    print("  cfactor max |gpu|", np.abs(cf_g).max(), "|oracle|", np.abs(cf_r).max(), "max diff", np.abs(cf_g - cf_r).max())
    if poses_on:
        print("  pose err", [float(np.abs(common.pose_error(ba.pose(k), g.keyframes[k]["pose"])).max()) for k in range(5)])
