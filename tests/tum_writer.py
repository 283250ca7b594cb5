"""Writes a synthetic scene as a TUM-RGB-D-format dataset directory (what the reference's reader consumes,
libvis/src/libvis/rgbd_video_io_tum_dataset.h:120-240): rgb/*.png (8-bit RGB), depth/*.png (16-bit grey, 0 = no
measurement), associated.txt, calibration.txt (fx fy cx cy, pixel-centre convention) and trajectory files with
"timestamp tx ty tz qx qy qz qw" lines.  PNG encoding with zlib only (no imaging library in this environment)."""
import os
import struct
import zlib

import numpy as np


def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)


def write_png(path, array, filter_type=0):
    """array: (H, W, 3) uint8 -> colour type 2; (H, W) uint16 -> 16-bit grey; (H, W) uint8 -> 8-bit grey.
    filter_type 0 (none), 1 (sub), 2 (up), 3 (average) or 4 (Paeth), applied to every row (to exercise the decoder)."""
    a = np.ascontiguousarray(array)
    h, w = a.shape[:2]
    if a.dtype == np.uint16:
        rows = a.astype(">u2").view(np.uint8).reshape(h, w * 2)
        bit_depth, color_type, bpp = 16, 0, 2
    elif a.ndim == 3:
        rows = a.reshape(h, w * 3)
        bit_depth, color_type, bpp = 8, 2, 3
    else:
        rows = a.reshape(h, w)
        bit_depth, color_type, bpp = 8, 0, 1
    rows = rows.astype(np.uint8)
    if filter_type == 1:
        f = rows.copy()
        f[:, bpp:] = rows[:, bpp:] - rows[:, :-bpp]
    elif filter_type == 2:
        f = rows.copy()
        f[1:] = rows[1:] - rows[:-1]
    elif filter_type in (3, 4):
        x = rows.astype(np.int32)
        left = np.zeros_like(x); left[:, bpp:] = x[:, :-bpp]
        up = np.zeros_like(x); up[1:] = x[:-1]
        upleft = np.zeros_like(x); upleft[1:, bpp:] = x[:-1, :-bpp]
        if filter_type == 3:
            pred = (left + up) // 2
        else:
            p = left + up - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, upleft))
        f = ((x - pred) & 0xff).astype(np.uint8)
    else:
        f = rows
    raw = np.concatenate([np.full((h, 1), filter_type, np.uint8), f], axis=1).tobytes()
    with open(path, "wb") as out:
        out.write(b"\x89PNG\r\n\x1a\n")
        out.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, 0)))
        comp = zlib.compress(raw, 6)
        half = len(comp) // 2            # two IDAT chunks: the decoder must concatenate them
        out.write(_chunk(b"IDAT", comp[:half]))
        out.write(_chunk(b"IDAT", comp[half:]))
        out.write(_chunk(b"IEND", b""))


def write_trajectory(path, timestamps, poses):
    """poses: 7-vectors in the Sophus order (qx qy qz qw tx ty tz)."""
    with open(path, "w") as f:
        f.write("# timestamp tx ty tz qx qy qz qw\n")
        for t, p in zip(timestamps, poses):
            p = np.asarray(p, np.float64)
            f.write("%s %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n" % (t, p[4], p[5], p[6], p[0], p[1], p[2], p[3]))


def write_dataset(folder, scene, trajectories, t0=1305031102.175304, dt=1.0 / 30):
    """scene: badslam_amd.synthetic scene (depth u16 with 65535 = invalid, rgb u8, camera in the pixel-corner
    convention).  trajectories: {filename: list of poses}.  Returns the timestamp strings."""
    os.makedirs(os.path.join(folder, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(folder, "depth"), exist_ok=True)
    stamps = ["%.6f" % (t0 + k * dt) for k in range(len(scene.depth))]
    with open(os.path.join(folder, "associated.txt"), "w") as assoc:
        assoc.write("# rgb_timestamp rgb_file depth_timestamp depth_file\n")
        for k, ts in enumerate(stamps):
            depth = scene.depth[k].copy()
            depth[depth == 65535] = 0
            write_png(os.path.join(folder, "rgb", ts + ".png"), scene.rgb[k], filter_type=k % 5)
            write_png(os.path.join(folder, "depth", ts + ".png"), depth, filter_type=(k + 3) % 5)
            assoc.write("%s rgb/%s.png %s depth/%s.png\n" % (ts, ts, ts, ts))
    fx, fy, cx, cy = [float(v) for v in scene.camera]
    with open(os.path.join(folder, "calibration.txt"), "w") as f:
        f.write("%.9g %.9g %.9g %.9g\n" % (fx, fy, cx - 0.5, cy - 0.5))
    for name, poses in trajectories.items():
        write_trajectory(os.path.join(folder, name), stamps, poses)
    return stamps
