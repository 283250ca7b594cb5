"""Work census of one sweep over the bench scene (built through the low-level C-ABI wrapper)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import _bench_lowlevel as bench
args = bench.parse_args()
ctx, g, data, poses_gt = bench.build_scene(args, lambda m: print(m, file=sys.stderr))
g.upload_surfels(data, np.zeros(data.shape[1], np.uint8))
g.bind_keyframes()
c = g.count_pairs()
N, K = data.shape[1], len(g.keyframes)
print("N", N, "K", K, "wave-kf candidates", c[0], f"({c[0] / (N / 64 * K):.3%} of all wave-kf, {c[0] / (N / 64):.1f} per wave)", "wave-kf hits", c[1],
      f"({c[1] / (N / 64):.1f} per wave)", "associated pairs", c[2], f"({c[2] / N:.2f} per surfel)", "in-image pairs", c[3],
      f"({c[3] / N:.2f} per surfel)", "avg lanes per hit wave", c[2] / max(1, c[1]))
