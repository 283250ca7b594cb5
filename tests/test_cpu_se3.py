"""The SE(3) functions the pose path uses on host and device (badslam_amd/csrc/se3_device.h, compiled here for the host)
against the oracle's restatement of Sophus (oracle_core.c): exp, log, product, inverse, 3x4 matrix on random tangents,
including small rotation angles where both switch to the series expansion."""
import os
import struct
import subprocess

import numpy as np

from oracle import binding as ob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _f32(words):
    return np.array([struct.unpack("<f", struct.pack("<I", int(w, 16)))[0] for w in words], np.float32)


def test_host_se3_functions_match_the_oracle(tmp_path):
    exe = str(tmp_path / "se3_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "badslam_amd", "host"), "-I",
                    os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tests", "cpp", "se3_check.cc")], check=True, timeout=300)
    rng = np.random.default_rng(11)
    tangents = rng.standard_normal((400, 12)).astype(np.float32)
    tangents[:100, 3:6] *= 1e-4     # near the identity rotation: series branch of exp / log
    tangents[:100, 9:12] *= 1e-4
    tangents[100:200] *= 0.05       # the size of a Gauss-Newton update
    text = "\n".join(" ".join(repr(float(v)) for v in row) for row in tangents) + "\n"
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=300, check=True).stdout.strip().splitlines()
    assert len(out) == len(tangents)
    worst = 0.0
    exact = 0
    for row, line in zip(tangents, out):
        got = _f32(line.split())
        Ta, Tb = ob.se3_exp(row[:6]), ob.se3_exp(row[6:])
        prod = ob.se3_mul(Ta, Tb)
        ref = np.concatenate([Ta.to_array(), prod.to_array(), ob.se3_inverse(Ta).to_array(), np.asarray(ob.se3_log(prod), np.float32),
                              np.asarray(ob.se3_matrix3x4(Ta), np.float32).ravel()]).astype(np.float32)
        assert got.shape == ref.shape == (39,)
        worst = max(worst, float(np.abs(got - ref).max()))
        exact += int(np.array_equal(got, ref))
        # group properties on the product itself: T * T^-1 = identity, exp(log(T)) = T
        inv = ob.se3_inverse(prod)
        ident = ob.se3_mul(prod, inv).to_array()
        assert np.abs(ident - np.array([0, 0, 0, 1, 0, 0, 0], np.float32)).max() < 5e-6
    print("bit-identical rows:", exact, "of", len(out), "worst abs difference", worst)
    assert worst < 2e-6
