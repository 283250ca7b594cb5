import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import common
sc = common.small_scene(num_keyframes=4, seed=3)
ba = common.build_oracle(sc, 400000)
g = common.build_gpu(sc, 400000, create_from=[])
data, active = common.oracle_surfels(ba)
g.upload_surfels(data, active)
np.set_printoptions(linewidth=200, precision=5)
for ud, us in [(1, 0), (0, 1), (1, 1)]:
    ba.use_depth, ba.use_desc = ud, us
    for k in range(2):
        F = np.array(list(ba.keyframes[k].frame_T_global), np.float32)
        Hd, bd, n, _ = ba.accumulate_pose_coeffs(k, accumulate_double=True)
        Hf, bf, _, _ = ba.accumulate_pose_coeffs(k, accumulate_double=False)
        H, b = g.accumulate_pose_coeffs(k, ud, us, F)
        print(f"depth={ud} desc={us} kf={k} n={n}")
        print("  H relerr gpu-vs-double max", np.abs(H - Hd).max() / np.abs(Hd).max(), " oracle float-vs-double", np.abs(Hf - Hd).max() / np.abs(Hd).max())
        print("  b gpu", b); print("  b dbl", bd); print("  b flt", bf)
        print("  H diag gpu", H[[0, 6, 11, 15, 18, 20]]); print("  H diag dbl", Hd[[0, 6, 11, 15, 18, 20]])
