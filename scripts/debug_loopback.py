import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import test_gpu_sharded_loopback as T

log = {}
def cam(c): return (c.fx, c.fy, c.cx, c.cy)
def step(g):
    g.bind_keyframes()
    g.update_surfel_normals()
    steps, _ = g.pcg_iteration(optimize_poses=True, optimize_geometry=True, optimize_depth_intrinsics=True,
                               optimize_color_intrinsics=True, max_inner_iterations=10)
    a = (cam(g.color_cam), cam(g.depth_cam), g.dp.a, [kf["pose"].copy() for kf in g.keyframes])
    g.bind_keyframes()
    g.optimize_intrinsics(True, True)
    b = (cam(g.color_cam), cam(g.depth_cam), g.dp.a)
    log.setdefault(id(g), []).append((steps, a, b))
    return steps

ref, results, loop, N = T._run_sharded_and_unsharded(step, seed=10)
l0, l1 = log[id(results[0]["scene"])], log[id(results[1]["scene"])]
for it, (x, y) in enumerate(zip(l0, l1)):
    print("iter", it, "steps", x[0], y[0])
    print("  after pcg: color", np.subtract(x[1][0], y[1][0]), "depth", np.subtract(x[1][1], y[1][1]), "a", x[1][2] - y[1][2],
          "pose maxdiff", max(np.abs(p - q).max() for p, q in zip(x[1][3], y[1][3])))
    print("  after intr: color", np.subtract(x[2][0], y[2][0]), "depth", np.subtract(x[2][1], y[2][1]), "a", x[2][2] - y[2][2])
