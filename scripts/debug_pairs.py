import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import common
sc = common.small_scene(num_keyframes=4, seed=3)
ba = common.build_oracle(sc, 400000)
g = common.build_gpu(sc, 400000, create_from=[])
data, active = common.oracle_surfels(ba)
g.upload_surfels(data, active)
np.set_printoptions(linewidth=220, precision=7, suppress=True)
k = 0
F = np.array(list(ba.keyframes[k].frame_T_global), np.float32)
N = data.shape[1]
idx = np.arange(N, dtype=np.uint32)
out = g.evaluate_pairs(k, idx, F)
ref = np.zeros_like(out)
for i in range(N):
    ok, e = ba.evaluate_pair(k, i)
    if not ok: continue
    ref[i, 0] = 1; ref[i, 1] = e.px; ref[i, 2] = e.py; ref[i, 3] = e.color_valid; ref[i, 4] = e.calibrated_depth
    ref[i, 5] = e.depth_residual; ref[i, 6] = e.depth_weight; ref[i, 7] = e.depth_inv_stddev
    ref[i, 8:14] = list(e.depth_jac_pose); ref[i, 14:16] = list(e.desc_residual); ref[i, 16:18] = list(e.desc_weight)
    ref[i, 18:24] = list(e.desc_jac_pose[0]); ref[i, 24:30] = list(e.desc_jac_pose[1]); ref[i, 30:34] = list(e.grad)
print("assoc mismatch", np.count_nonzero(out[:, 0] != ref[:, 0]), "of", N, "associated", int(ref[:, 0].sum()))
both = (out[:, 0] == 1) & (ref[:, 0] == 1)
names = {1: "px", 2: "py", 4: "depth", 5: "r_depth", 7: "inv_std", 8: "Jd0", 11: "Jd3", 14: "r_desc1", 15: "r_desc2", 18: "J1_0", 21: "J1_3", 30: "gx1", 31: "gy1", 32: "gx2", 33: "gy2"}
for c, n in names.items():
    d = np.abs(out[both, c] - ref[both, c])
    print(f"{n:8s} exact {np.count_nonzero(d == 0)}/{both.sum()} maxabs {d.max():.3e} ref max {np.abs(ref[both, c]).max():.3e}")
bad = np.where(both & (np.abs(out[:, 30] - ref[:, 30]) > 1e-3))[0][:5]
for i in bad:
    print("surfel", i, "gpu grads", out[i, 30:34], "ref", ref[i, 30:34], "pxx,pxy gpu", out[i, 34:36], "px py", ref[i, 1:3])
