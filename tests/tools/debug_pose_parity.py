#!/usr/bin/env python3
"""Diagnostic (GPU box): where do the defined pose sums of the HIP path and of the oracle differ?  Scene (a) of
tests/test_gpu_scale_parity.py; per keyframe H, b at the perturbed pose; for a differing keyframe the first differing
64-surfel tile by bisection over the surfel prefix, then per-pair evaluation of that tile on both sides."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from badslam_amd import synthetic   # noqa: E402
from tests import common            # noqa: E402

scene = synthetic.make_scene(200, 160, 120, seed=21, cell=2, translation_range=5.0, rotation_range=0.9)
orc = common.build_oracle(scene, 900000)
g = common.build_gpu(scene, 900000, create_from=[])
data, active = common.oracle_surfels(orc)
g.upload_surfels(data, active)
N = data.shape[1]
rng = np.random.Generator(np.random.PCG64(41))
perturbed = [synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
orc.use_depth, orc.use_desc = 1, 1
bad = []
for k, T in enumerate(perturbed):
    orc.set_pose(k, T)
    F = np.array(list(orc.keyframes[k].frame_T_global), np.float32)
    H, b = g.accumulate_pose_coeffs(k, True, True, F)
    Hd, bd, n, _ = orc.accumulate_pose_coeffs(k, accumulate_double=False)
    if not (np.array_equal(np.float32(H), np.float32(Hd)) and np.array_equal(np.float32(b), np.float32(bd))):
        bad.append(k)
        print(f"keyframe {k}: n={n} max|dH|={np.abs(H - Hd).max():.3e} of {np.abs(Hd).max():.3e}, max|db|={np.abs(b - bd).max():.3e}")
print(f"{len(bad)} of {len(perturbed)} keyframes differ at the first linearisation point: {bad[:20]}")


def sums(k, F, n):
    g.surfels_size = n
    orc.surfels.surfels_size = n
    H, b = g.accumulate_pose_coeffs(k, True, True, F)
    Hd, bd, _, _ = orc.accumulate_pose_coeffs(k, accumulate_double=False)
    return np.float32(np.concatenate([H, b])), np.float32(np.concatenate([Hd, bd]))


for k in bad[:3]:
    F = np.array(list(orc.keyframes[k].frame_T_global), np.float32)
    lo, hi = 0, (N + 63) // 64          # tiles: prefix of `lo` tiles agrees, prefix of `hi` tiles differs
    while hi - lo > 1:
        mid = (lo + hi) // 2
        a, b_ = sums(k, F, min(N, 64 * mid))
        if np.array_equal(a, b_):
            lo = mid
        else:
            hi = mid
    g.surfels_size = N
    orc.surfels.surfels_size = N
    tile = lo
    idx = np.arange(64 * tile, min(N, 64 * tile + 64), dtype=np.uint32)
    got = g.evaluate_pairs(k, idx, F)
    ref = orc.evaluate_pairs(k, idx, F)
    refi = ref.view(np.int32)
    print(f"keyframe {k}: first differing tile {tile} (surfels {idx[0]}..{idx[-1]}); hip associated {int((got[:, 0] == 1).sum())}, oracle {int((refi[:, 0] == 1).sum())}")
    for lane in range(len(idx)):
        ga, ra = got[lane, 0] == 1, refi[lane, 0] == 1
        if ga != ra:
            print(f"  lane {lane}: association differs hip={ga} oracle={ra}")
        elif ga:
            for name, cols, (o, n) in (("depth_res", [5], (5, 1)), ("depth_w", [6], (6, 1)), ("inv_std", [7], (7, 1)), ("Jd", list(range(8, 14)), (8, 6)),
                                       ("desc_res", [14, 15], (15, 2)), ("desc_w", [16, 17], (17, 2)), ("Jdesc", list(range(18, 30)), (19, 12))):
                x, y = got[lane, cols], ref[lane, o:o + n].view(np.float32)
                if not np.array_equal(x, y):
                    print(f"  lane {lane} {name}: hip {x} oracle {y}")
    # the tile alone
    a, b_ = None, None
