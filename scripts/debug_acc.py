import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import common
sc = common.small_scene(num_keyframes=4, seed=3)
ba = common.build_oracle(sc, 400000)
g = common.build_gpu(sc, 400000, create_from=[])
data, active = common.oracle_surfels(ba)
g.upload_surfels(data, active)
np.set_printoptions(linewidth=220, precision=6)
k = 0
F = np.array(list(ba.keyframes[k].frame_T_global), np.float32)
N = data.shape[1]
out = g.evaluate_pairs(k, np.arange(N, dtype=np.uint32), F).astype(np.float64)


def pair_sum(n):
    o = out[:n]
    m = (o[:, 0] == 1) & (o[:, 3] == 1)
    H = np.zeros((6, 6))
    for t in range(2):
        J = o[m][:, 18 + 6 * t:24 + 6 * t]
        w = o[m][:, 16 + t]
        H += (J * w[:, None]).T @ J
    return np.diag(H), int(m.sum())


for n in [64, 128, 256, 1024, 8192, N]:
    g.surfels_size = n
    H, b = g.accumulate_pose_coeffs(k, 0, 1, F)
    ref, cnt = pair_sum(n)
    print(n, cnt, "gpu", H[[0, 6, 11]], "pairsum", ref[:3], "ratio", H[0] / max(ref[0], 1e-30))
