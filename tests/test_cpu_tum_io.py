"""CPU checks of the data-format code on the host side (badslam_amd/host/rgbd_io.cc): PNG decoding (all five filters, split
IDAT, 8-bit RGB and 16-bit grey), the TUM-format reader (association file, calibration convention, trajectory
parsing) - through `ba_tum --check-dataset`, which needs no GPU."""
import os
import subprocess

import numpy as np

from tests import common, tum_writer

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "badslam_amd", "lib", "ba_tum")


def test_tum_reader_and_png_decoder(tmp_path):
    assert os.path.exists(BIN), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    scene = common.small_scene(num_keyframes=5, width=160, height=120, seed=2)
    stamps = tum_writer.write_dataset(str(tmp_path), scene, {"groundtruth.txt": scene.poses_gt})
    out = subprocess.run([BIN, "--check-dataset", str(tmp_path), "groundtruth.txt"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    head = lines[0].split()
    assert int(head[1]) == 5 and int(head[3]) == 160 and int(head[5]) == 120
    cam = np.array([float(v) for v in head[7:11]])
    assert np.allclose(cam, scene.camera, rtol=1e-6)          # file stores cx, cy - 0.5; the reader adds it back
    for k, line in enumerate(lines[1:]):
        t = line.split()
        assert t[2] == stamps[k] and t[3] == stamps[k]
        depth = scene.depth[k].astype(np.uint64)
        depth[depth == 65535] = 0
        assert int(t[5]) == int(depth.sum())                   # every 16-bit sample decoded exactly
        rgb = scene.rgb[k].astype(np.uint64)
        assert int(t[7]) == int((rgb[..., 0] + 2 * rgb[..., 1] + 3 * rgb[..., 2]).sum())   # channel order preserved
        pose = np.array([float(v) for v in t[9:16]])
        gt = np.asarray(scene.poses_gt[k], np.float64)
        if pose[3] * gt[3] < 0:
            pose[:4] = -pose[:4]
        assert np.abs(pose - gt).max() < 1e-6
