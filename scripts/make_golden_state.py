#!/usr/bin/env python3
"""Generates tests/golden/state_v101.bin: the binary backend state (badslam_amd/host/rgbd_io.h: SaveState) of a tiny
synthetic TUM-format sequence after one BA call, together with the poses ba_tum wrote for the same run.  Needs the GPU
(run through gpurun); the CPU test tests/test_cpu_state_file.py parses the committed file with tests/state_file.py.
usage: python scripts/make_golden_state.py <out_dir>"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import common, tum_writer   # noqa: E402

out_dir = sys.argv[1]
os.makedirs(out_dir, exist_ok=True)
scene = common.small_scene(num_keyframes=3, width=160, height=120, seed=11)
rng = np.random.Generator(np.random.PCG64(12))
initial = [common.synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
with tempfile.TemporaryDirectory() as tmp:
    tum_writer.write_dataset(tmp, scene, {"initial.txt": initial})
    state = os.path.join(out_dir, "state_v101.bin")
    cmd = [os.path.join(ROOT, "badslam_amd", "lib", "ba_tum"), tmp, "initial.txt", os.path.join(tmp, "out"), "--cell", "4",
           "--iterations", "1", "--max_depth", "8", "--save_state", state]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    print(proc.stdout[-1500:], proc.stderr[-1500:])
    assert proc.returncode == 0
    with open(os.path.join(tmp, "out.poses.txt")) as f, open(os.path.join(out_dir, "state_v101.poses.txt"), "w") as g:
        g.write(f.read())
print("wrote", state, os.path.getsize(state), "bytes")
