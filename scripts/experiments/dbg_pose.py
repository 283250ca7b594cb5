import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from badslam_amd import capi, synthetic
from tests import common
K = int(sys.argv[1]) if len(sys.argv) > 1 else 80
scene = synthetic.make_scene(K, 160, 120, seed=21, cell=2, translation_range=5.0, rotation_range=0.9)
g = common.build_gpu(scene, 900000)
data = g.download_surfels()
active = np.ones(data.shape[1], np.uint8)
rng = np.random.Generator(np.random.PCG64(41))
perturbed = [synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
res = {}
for form, parts in ((1,1),(2,0)):
    capi.check(g.ctx.lib.bahip_debug_set_pose_form(form)); capi.check(g.ctx.lib.bahip_debug_set_launch_shapes(0, parts))
    g.upload_surfels(data, active)
    for k, T in enumerate(perturbed):
        g.keyframes[k]["pose"] = np.asarray(T, np.float32)
    g.bind_keyframes()
    # H, b of single work items through the same kernels
    Hs = []
    for k in range(min(K, 6)):
        F = None
        H, b = g.accumulate_pose_coeffs(k, True, True, None) if False else (None, None)
    poses, its, conv, rounds = g.estimate_keyframe_poses(True, True)
    res[form] = (poses.copy(), its.copy())
    print("form", form, "rounds", rounds, "its", its[:16], "conv", int(conv.sum()), flush=True)
same = [k for k in range(K) if np.array_equal(res[1][0][k].astype(np.float32), res[2][0][k].astype(np.float32))]
print("keyframes with identical final pose:", len(same), "of", K, "; differing:", [k for k in range(K) if k not in same][:40])
