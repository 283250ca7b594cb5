"""Runs the C++ restatement of the reference's closed-loop BA tests (badslam_amd/host/test_directba.cc)
against the C++ host classes vis::DirectBA / Keyframe / CUDABuffer on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "badslam_amd", "lib", "test_directba")


@pytest.mark.parametrize("name", ["PoseOptimizationWithGeometricResidual", "PoseOptimizationColorOnlyCues",
                                  "AlternatingGeometryOptimizationWithGeometricResidual",
                                  "PCGGeometryOptimizationWithGeometricResidual",
                                  "AlternatingGeometryOptimizationWithPhotometricResidual",
                                  "PCGGeometryOptimizationWithPhotometricResidual",
                                  "AlternatingIntrinsicsOptimizationWithPhotometricResidual",
                                  "PCGIntrinsicsOptimizationWithPhotometricResidual",
                                  "AlternatingDepthDeformationOptimizationWithGeometricResidual",
                                  "PCGDepthDeformationOptimizationWithGeometricResidual",
                                  "AlternatingIntrinsicsOptimizationWithGeometricResidual",
                                  "PCGIntrinsicsOptimizationWithGeometricResidual"])
def test_reference_closed_loop(name):
    assert os.path.exists(BIN), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    proc = subprocess.run([BIN, name], capture_output=True, text=True, timeout=600)
    print(proc.stdout)
    print(proc.stderr)
    assert proc.returncode == 0, proc.stdout[-2000:]
