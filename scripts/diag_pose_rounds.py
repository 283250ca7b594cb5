"""Which keyframes of the bench scene need many GN rounds, and how well are they constrained?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench

args = bench.parse_args()
ctx, g, data, poses_gt = bench.build_scene(args, lambda m: print(m, file=sys.stderr))
g.upload_surfels(data, np.zeros(data.shape[1], np.uint8))
stats = dict(rounds=[], gn_steps=[])
for it in range(3):
    from badslam_amd import capi
    for kf in g.keyframes:
        kf["activation"] = capi.KF_ACTIVE
    g.bind_keyframes()
    g.update_surfel_activation()
    g.optimize_geometry_iteration(True, True)
    poses, its, conv, rounds = g.estimate_keyframe_poses(True, True)
    for k, kf in enumerate(g.keyframes):
        kf["pose"] = poses[k].astype(np.float32)
    bad = np.where(its > 4)[0]
    print("iteration", it, "rounds", rounds, "its histogram", np.bincount(its)[:8], "bad kfs", bad.tolist(), "conv", conv[bad].tolist())
    sample = np.arange(0, data.shape[1], 50, dtype=np.uint32)
    for k in bad[:6]:
        from badslam_amd import se3
        inv = se3.inverse(poses[k])
        F = se3.matrix(inv)[:3, :].astype(np.float32).ravel()
        out = g.evaluate_pairs(k, sample, F)
        valid_px = int(((g.keyframes[k]["depth"].download() & 0x8000) == 0).sum())
        print("   kf", k, "its", its[k], "associated (x50)", int(out[:, 0].sum()) * 50, "valid depth px", valid_px,
              "pose err vs gt", np.abs(se3.log(se3.mul(se3.inverse(poses_gt[k]), poses[k].astype(np.float64)))).max())
