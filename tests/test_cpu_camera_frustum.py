"""vis::CameraFrustum::Intersects (the co-visibility test behind DirectBA::AddKeyframe, libvis/src/libvis/camera_frustum.h)
against brute force on random frustum pairs.  The reference has no test for it; a false negative would silently drop a
keyframe from another's co-visibility list (no surfel filtering against it, no covisible activation)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frustum_intersection_against_brute_force(tmp_path):
    exe = str(tmp_path / "frustum_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "badslam_amd", "host"), "-I",
                    os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "frustum_check.cc")], check=True, timeout=300)
    out = subprocess.run([exe, "1500"], capture_output=True, text=True, timeout=600, check=True).stdout
    t = np.array([[int(v) for v in line.split()] for line in out.strip().splitlines()])
    sat, sat_reverse, brute = t[:, 0], t[:, 1], t[:, 2]
    assert len(t) == 1500
    assert np.array_equal(sat, sat_reverse)                          # symmetric
    assert 0.1 < brute.mean() < 0.9                                  # the sample has both kinds
    assert not np.any((brute == 1) & (sat == 0))                     # never misses an intersection a sample point proves
    # the other direction cannot be strict (a thin overlap may contain no sample point), but must be rare
    assert np.sum((sat == 1) & (brute == 0)) <= 0.03 * len(t)
