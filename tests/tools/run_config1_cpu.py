"""BASELINE.json configs[0]: 10 synthetic 320x240 keyframes, ~50 k surfels, 5 BA iterations on the CPU oracle (plumbing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import common
from oracle import binding as ob
scene = common.small_scene(num_keyframes=10, width=320, height=240, seed=0, cell=2)
ba = common.build_oracle(scene, 400000)
rng = np.random.Generator(np.random.PCG64(1))
n = ba.surfels_size
data, _ = common.oracle_surfels(ba)
data[2] += rng.uniform(0, 0.005, n).astype(np.float32)
ba.surfel_data[:, :n] = data
pert = [common.synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
for k, T in enumerate(pert):
    ba.set_pose(k, T)
c0, nres = ba.evaluate_cost()
t0 = time.time()
ba.bundle_adjustment(optimize_poses=True, optimize_geometry=True, do_surfel_updates=False, min_iterations=5, max_iterations=5)
dt = time.time() - t0
c1, _ = ba.evaluate_cost()
err0 = np.sqrt(np.mean([np.sum(common.pose_error(g, p)[:3] ** 2) for g, p in zip(scene.poses_gt, pert)]))
err1 = np.sqrt(np.mean([np.sum(common.pose_error(g, ba.pose(k))[:3] ** 2) for k, g in enumerate(scene.poses_gt)]))
print(f"surfels {n} residuals {nres} threads {ob.lib().orc_num_threads()} 5 iterations {dt:.2f} s ({5 / dt:.2f} it/s) cost {c0:.1f} -> {c1:.1f} pose rmse {err0:.2e} -> {err1:.2e} m")
