#!/usr/bin/env python3
"""Generates tests/golden/e2e_vga.npz: OUTPUTS of the reference's own kernels (oracle/_ref: its .cu files compiled for the host) for
the whole chain from raw input at VGA -- 20 keyframes 640x480, bilateral filter, keyframe preprocessing, filtered creation, three
alternating iterations with do_surfel_updates = true, end tasks -- once with exact bilinear weights and once with the texture
unit's 8-bit weights (tests/e2e_vga.py describes the chain and the driver).  Runs only where /root/reference exists (the build
container); the committed file travels.  Also replays the chain with the oracle and prints how far it is from the reference.
usage: python tests/make_golden_e2e_vga.py [out.npz]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import e2e_vga as e2e   # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else e2e.PATH
t0 = time.time()
scene, raw, rgb, start = e2e.scene_and_raw_input()
fix = dict(input_digest=e2e.input_digest(raw, rgb, start), start_poses=start, camera=np.asarray(scene.camera, np.float64))
print(f"scene rendered in {time.time() - t0:.0f}s; raw depth valid {np.count_nonzero(raw) / raw.size:.3f}", flush=True)
for prefix, quantize in (("", False), ("quantized_", True)):
    t1 = time.time()
    print(f"reference kernels, {'8-bit texture weights' if quantize else 'exact bilinear weights'}:", flush=True)
    out = e2e.run_reference(scene, raw, rgb, start, quantize)
    for name, value in out.items():
        fix[prefix + name] = value
    print(f"  counts per stage {out['counts'].tolist()}, final {int(out['final_surfels'])}; {time.time() - t1:.0f}s", flush=True)
moved = np.linalg.norm(fix["poses"][:, 4:] - start[:, 4:], axis=1)
delta = np.linalg.norm(fix["poses"][:, 4:] - fix["quantized_poses"][:, 4:], axis=1)
print(f"poses moved by {moved.mean():.2e} m on average; exact vs 8-bit weights: RMSE {np.sqrt(np.mean(delta ** 2)):.2e} m, max {delta.max():.2e} m; "
      f"surfels {int(fix['final_surfels'])} vs {int(fix['quantized_final_surfels'])}")
fix["exact_vs_quantized_pose_rmse"] = np.float64(np.sqrt(np.mean(delta ** 2)))
# for the record: the same chain with the reference's OWN append order (row-major): how many survivors the order decides
rowmajor = e2e.run_reference(scene, raw, rgb, start, False, log=lambda *a: None, creation_order="row-major")
from scipy.spatial import cKDTree   # noqa: E402
dist, _ = cKDTree(rowmajor["sampled_rows"][:3].T.astype(np.float64)).query(fix["sampled_rows"][:3].T.astype(np.float64))
print(f"reference kernels, row-major vs tile-major creation order: final {int(rowmajor['final_surfels'])} vs {int(fix['final_surfels'])} surfels, "
      f"pose delta max {np.abs(rowmajor['poses'] - fix['poses']).max():.1e}")
fix["rowmajor_final_surfels"], fix["rowmajor_poses"] = rowmajor["final_surfels"], rowmajor["poses"]
np.savez_compressed(out_path, **fix)
print(out_path, os.path.getsize(out_path), "bytes")
t2 = time.time()
orc = e2e.run_oracle(scene, raw, rgb, start)
print(f"oracle replay in {time.time() - t2:.0f}s")
for prefix in ("", "quantized_"):
    e2e.compare(orc["poses"], orc["rows"], fix, prefix)
