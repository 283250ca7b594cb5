"""TUM-format input end to end on the GPU: a synthetic sequence written as PNGs + associated.txt +
calibration.txt + an initial trajectory -> `ba_tum` (reader, PNG decoder, PreprocessFrame chain incl. the bilateral
filter, Keyframe from buffers, BundleAdjustment with surfel updates, SavePoses / SaveCalibration / PLY export) ->
the written trajectory is compared with the ground truth."""
import os
import struct
import subprocess

import numpy as np
import pytest

from badslam_amd import se3
from tests import common, tum_writer

pytestmark = pytest.mark.gpu

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "badslam_amd", "lib", "ba_tum")


def _read_trajectory(path):
    out = []
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        t = line.split()
        v = [float(x) for x in t[1:]]
        out.append((t[0], np.array([v[3], v[4], v[5], v[6], v[0], v[1], v[2]])))   # -> Sophus order
    return out


def _relative(poses):
    inv0 = se3.inverse(np.asarray(poses[0], np.float64))
    return [se3.mul(inv0, np.asarray(p, np.float64)) for p in poses]


@pytest.mark.parametrize("use_pcg", [False, True])
def test_ba_on_a_tum_format_sequence(tmp_path, use_pcg):
    assert os.path.exists(BIN), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    scene = common.small_scene(num_keyframes=6, width=320, height=240, seed=4)
    rng = np.random.Generator(np.random.PCG64(6))
    initial = [common.synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    stamps = tum_writer.write_dataset(str(tmp_path), scene, {"groundtruth.txt": scene.poses_gt, "initial.txt": initial})
    out = str(tmp_path / "result")
    cmd = [BIN, str(tmp_path), "initial.txt", out, "--cell", "2", "--iterations", "6", "--max_depth", "8"] + (["--pcg"] if use_pcg else [])
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    print(proc.stdout[-3000:])
    print(proc.stderr[-3000:])
    assert proc.returncode == 0

    result = _read_trajectory(out + ".poses.txt")
    assert [t for t, _ in result] == stamps
    est = [p for _, p in result]                      # already relative to the first keyframe (B/io.cc:542,551)
    gt_rel, init_rel = _relative(scene.poses_gt), _relative(initial)
    err_after = np.array([np.linalg.norm(common.pose_error(g, e)[:3]) for g, e in zip(gt_rel[1:], est[1:])])
    err_before = np.array([np.linalg.norm(common.pose_error(g, e)[:3]) for g, e in zip(gt_rel[1:], init_rel[1:])])
    print("relative translation error before", err_before, "after", err_after)
    assert np.allclose(est[0][[3, 4, 5, 6]], [1, 0, 0, 0], atol=1e-6)          # first pose is the identity by construction
    assert err_before.mean() > 3e-3                                              # the 5 mm / 1 mrad perturbation
    assert err_after.mean() < 0.25 * err_before.mean()
    assert err_after.max() < 2.5e-3

    # calibration files: cx, cy are written in the pixel-centre convention (B/io.cc:583-586)
    fx, fy, cx, cy = [float(v) for v in open(out + ".depth_intrinsics.txt").read().split()]
    assert np.allclose([fx, fy, cx + 0.5, cy + 0.5], scene.camera, rtol=1e-5)
    dims = open(out + ".deformation.txt").readline().split()
    assert [int(d) for d in dims] == [(320 - 1) // 2 + 1, (240 - 1) // 2 + 1]

    # binary PLY: header + 27 bytes per vertex, points near the planes of the scene (z about 2.5 m in front of the rig)
    blob = open(out + ".ply", "rb").read()
    header_end = blob.index(b"end_header\n") + len(b"end_header\n")
    header = blob[:header_end].decode()
    n = int([l for l in header.splitlines() if l.startswith("element vertex")][0].split()[-1])
    assert n > 20000 and len(blob) == header_end + 27 * n
    first = struct.unpack_from("<3f3B3f", blob, header_end)
    assert np.isfinite(first[:3]).all() and abs(np.linalg.norm(first[6:9]) - 1) < 1e-3


def _run(cmd, timeout=600):
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    print(proc.stdout[-3000:])
    print(proc.stderr[-3000:])
    assert proc.returncode == 0
    return proc.stdout


def _translation_errors(scene, est):
    gt_rel = _relative(scene.poses_gt)
    return np.array([np.linalg.norm(common.pose_error(g, e)[:3]) for g, e in zip(gt_rel, est)])


@pytest.mark.parametrize("mode", ["--incremental", "--parallel_ba"])
def test_keyframes_fed_one_at_a_time_through_the_scheduler(tmp_path, mode):
    """BAScheduler (B/bad_slam.cc AddKeyframe / RunBundleAdjustment / StartParallelIterations /
    BAThreadMain).  Every keyframe arrives with its pose relative to the previous keyframe; with --parallel_ba the
    iterations run on the BA thread while the main thread prepares the next keyframe (mutex protocol of the reference),
    and a queued keyframe's absolute pose is formed from the previous keyframe's pose *after* BA."""
    scene = common.small_scene(num_keyframes=6, width=320, height=240, seed=4)
    rng = np.random.Generator(np.random.PCG64(6))
    initial = [common.synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    stamps = tum_writer.write_dataset(str(tmp_path), scene, {"initial.txt": initial})
    out = str(tmp_path / "result")
    log = _run([BIN, str(tmp_path), "initial.txt", out, "--cell", "2", "--iterations", "10", "--max_depth", "8", mode])
    assert "6 keyframes through the scheduler" in log
    if mode == "--parallel_ba":
        done = int(log.split("keyframes through the scheduler (BA thread), ")[1].split(" parallel")[0])
        assert 5 <= done <= 50                       # at most max_num_ba_iterations_per_keyframe queued per keyframe
    result = _read_trajectory(out + ".poses.txt")
    assert [t for t, _ in result] == stamps
    err_after = _translation_errors(scene, [p for _, p in result])
    err_before = _translation_errors(scene, _relative(initial))
    print("before", err_before, "after", err_after)
    assert err_before[1:].mean() > 3e-3
    # (how many iterations the BA thread gets in before the last keyframe arrives depends on the timing of the two
    # threads, hence the margin)
    assert err_after[1:].mean() < 0.4 * err_before[1:].mean()
    assert err_after.max() < 4e-3


def test_state_file_round_trip_and_resume(tmp_path):
    """SURVEY 8f row 3 (binary state, B/io.cc:38-535): run A does 2 BA calls and saves; run B loads the state (keyframes are
    rebuilt from the video frames, surfels / poses / counters come from the file), saves again and continues.  The second
    state file must equal the first one bit for bit, and the resumed run must end where an uninterrupted run ends."""
    scene = common.small_scene(num_keyframes=5, width=320, height=240, seed=7)
    rng = np.random.Generator(np.random.PCG64(8))
    initial = [common.synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    tum_writer.write_dataset(str(tmp_path), scene, {"initial.txt": initial})
    base = [BIN, str(tmp_path), "initial.txt"]
    opts = ["--cell", "2", "--max_depth", "8"]
    state_a, state_b = str(tmp_path / "a.state"), str(tmp_path / "b.state")
    _run(base + [str(tmp_path / "a")] + opts + ["--iterations", "2", "--save_state", state_a])
    log = _run(base + [str(tmp_path / "b")] + opts + ["--iterations", "0", "--load_state", state_a, "--save_state", state_b])
    assert "loaded" in log and "5 keyframes" in log
    blob_a, blob_b = open(state_a, "rb").read(), open(state_b, "rb").read()
    assert blob_a[:7] == b"BADSLAM" and blob_a[7] == 101
    assert len(blob_a) > 8 * 4 * 20000                      # eight rows of at least 20 k surfels
    assert blob_a == blob_b
    # the poses written after loading are the poses written before saving
    pa, pb = _read_trajectory(str(tmp_path / "a") + ".poses.txt"), _read_trajectory(str(tmp_path / "b") + ".poses.txt")
    assert np.allclose(np.array([p for _, p in pa]), np.array([p for _, p in pb]), atol=1e-6)

    # resume: 2 + 3 calls end at the same accuracy as 5 calls in one go (not bit-equal: the pose sums use float atomics)
    _run(base + [str(tmp_path / "c")] + opts + ["--iterations", "3", "--load_state", state_a])
    _run(base + [str(tmp_path / "d")] + opts + ["--iterations", "5"])
    pc, pd = _read_trajectory(str(tmp_path / "c") + ".poses.txt"), _read_trajectory(str(tmp_path / "d") + ".poses.txt")
    err_c, err_d = _translation_errors(scene, [p for _, p in pc]), _translation_errors(scene, [p for _, p in pd])
    print("resumed", err_c, "uninterrupted", err_d)
    assert err_c.max() < 2.5e-3 and err_d.max() < 2.5e-3
    assert np.abs(np.array([p for _, p in pc]) - np.array([p for _, p in pd])).max() < 1e-3

    # a truncated file and a file of the reference's own version are refused without touching anything; so are files whose
    # calibration cannot belong to this run: another image size, another surfel cell size, a depth scale of zero
    import struct
    open(str(tmp_path / "short.state"), "wb").write(blob_a[:len(blob_a) // 2])
    open(str(tmp_path / "v1.state"), "wb").write(blob_a[:7] + bytes([1]) + blob_a[8:])
    frames = struct.unpack_from("<I", blob_a, 8)[0]
    color_camera = 12 + 28 * frames                       # type, width, height, parameter count, 4 floats, pyramid level
    depth_camera = color_camera + 36
    cfactor = depth_camera + 32                           # width, height, stride, rows
    cf_h, cf_stride = struct.unpack_from("<ii", blob_a, cfactor + 4)
    depth_params = cfactor + 12 + cf_h * cf_stride        # a, raw_to_float_depth, baseline_fx, cell
    assert struct.unpack_from("<i", blob_a, color_camera + 4)[0] == 320 and struct.unpack_from("<i", blob_a, depth_params + 12)[0] == 2

    def patched(name, offset, fmt, value):
        blob = bytearray(blob_a)
        struct.pack_into(fmt, blob, offset, value)
        open(str(tmp_path / name), "wb").write(bytes(blob))
        return name

    hostile = ["short.state", "v1.state",
               patched("width.state", color_camera + 4, "<i", 321),
               patched("cell.state", depth_params + 12, "<i", 4),
               patched("cell0.state", depth_params + 12, "<i", 0),
               patched("scale.state", depth_params + 4, "<f", 0.0),
               patched("focal.state", depth_camera + 16, "<f", float("nan"))]
    for bad in hostile:
        proc = subprocess.run(base + [str(tmp_path / "e")] + opts + ["--iterations", "0", "--load_state", str(tmp_path / bad)],
                              capture_output=True, text=True, timeout=600)
        assert proc.returncode == 1 and "cannot load state" in proc.stderr


def test_non_keyframes_follow_their_keyframes(tmp_path):
    """Every second frame is a keyframe; the initial trajectory has a smooth drift.  BA corrects the keyframes and the
    trajectory deformation (B/trajectory_deformation.cc:45-130) carries the correction to the frames in between."""
    scene = common.small_scene(num_keyframes=7, width=320, height=240, seed=5)
    drift = np.array([0.004, -0.003, 0.002, 0.0008, -0.0006, 0.0005])
    initial = [se3.mul(np.asarray(T, np.float64), se3.exp(k * drift)) for k, T in enumerate(scene.poses_gt)]
    stamps = tum_writer.write_dataset(str(tmp_path), scene, {"initial.txt": initial})
    out = str(tmp_path / "result")
    proc = subprocess.run([BIN, str(tmp_path), "initial.txt", out, "--cell", "2", "--iterations", "6", "--max_depth", "8", "--interval", "2"],
                          capture_output=True, text=True, timeout=600)
    print(proc.stdout[-2000:])
    print(proc.stderr[-2000:])
    assert proc.returncode == 0
    result = _read_trajectory(out + ".poses.txt")
    assert [t for t, _ in result] == stamps                      # all 7 frames are written, 4 of them keyframes
    est = [p for _, p in result]
    gt_rel, init_rel = _relative(scene.poses_gt), _relative(initial)
    err_after = np.array([np.linalg.norm(common.pose_error(g, e)[:3]) for g, e in zip(gt_rel, est)])
    err_before = np.array([np.linalg.norm(common.pose_error(g, e)[:3]) for g, e in zip(gt_rel, init_rel)])
    print("before", [round(float(v), 5) for v in err_before], "after", [round(float(v), 5) for v in err_after])
    keyframes, others = [2, 4, 6], [1, 3, 5]
    assert err_after[keyframes].max() < 1e-3
    # the in-between frames inherit the interpolated correction: most of their drift is gone too (the synthetic frames are
    # far apart in pose, so interpolating the correction by frame index is only approximately right)
    assert (err_after[others] < 0.5 * err_before[others]).all()


def test_e2e_vga_golden_through_a_tum_format_directory(tmp_path):
    """"outputs match the reference on identical TUM-format input".  The raw input of tests/golden/e2e_vga.npz
    (20 noisy 640x480 depth frames with holes + RGB, perturbed start poses) is written as a TUM RGB-D directory -- 16-bit depth
    PNGs, RGB PNGs, associated.txt, calibration.txt, a trajectory file -- and goes through `ba_tum`: the dataset reader, the PNG
    decoder, the PreprocessFrame chain (bilateral filter + depth cut-off), the Keyframe constructor and
    DirectBA::BundleAdjustment(do_surfel_updates, three iterations, end tasks) with the chain's parameters.  The trajectory it writes
    is held against the REFERENCE's own kernels on the same raw input (the golden's poses, relative to the first keyframe as
    SavePoses writes them): pose RMSE <= 1e-5 m (BASELINE.json), and the PLY holds the golden's number of surfels (<= 0.1 % apart).
    The same with --row_major_creation against the reference run in its own append order."""
    import types
    from tests import e2e_vga as e2e
    assert os.path.exists(BIN)
    scene, raw, rgb, start = e2e.scene_and_raw_input()
    with np.load(e2e.PATH) as f:
        golden = {name: f[name] for name in f.files}
    assert np.array_equal(e2e.input_digest(raw, rgb, start), golden["input_digest"])
    depth = [np.where(r == 0, 65535, r).astype(np.uint16) for r in raw]        # the writer's convention: 65535 = no measurement -> PNG 0
    as_scene = types.SimpleNamespace(depth=depth, rgb=[c for c in rgb], camera=scene.camera)
    stamps = tum_writer.write_dataset(str(tmp_path), as_scene, {"initial.txt": [np.asarray(T, np.float64) for T in start]})
    s = scene.raw_to_float_depth

    def run(extra, tag):
        out = str(tmp_path / tag)
        cmd = [BIN, str(tmp_path), "initial.txt", out, "--cell", str(e2e.CELL), "--iterations", "1", "--ba_call_iterations", str(e2e.ITERATIONS),
               "--max_depth", repr(e2e.MAX_DEPTH_M), "--raw_to_float_depth", repr(float(s)), "--baseline_fx", repr(float(scene.baseline_fx)),
               "--spatial_sort_cell", "0", "--bilateral_sigma_xy", repr(e2e.BILATERAL[0]), "--bilateral_sigma_inv_depth", repr(e2e.BILATERAL[1]),
               "--bilateral_radius_factor", repr(e2e.BILATERAL[2])] + extra
        log = _run(cmd, timeout=900)
        assert "BA call 1: %d iteration(s)" % e2e.ITERATIONS in log
        result = _read_trajectory(out + ".poses.txt")
        assert [t for t, _ in result] == stamps
        blob = open(out + ".ply", "rb").read()
        header = blob[:blob.index(b"end_header\n")].decode()
        n = int([l for l in header.splitlines() if l.startswith("element vertex")][0].split()[-1])
        return np.asarray([p for _, p in result], np.float64), n

    def relative_rmse(est_rel, reference_absolute):
        ref_rel = np.asarray(_relative([np.asarray(p, np.float64) for p in reference_absolute]))
        d = est_rel[:, 4:] - ref_rel[:, 4:]
        return float(np.sqrt(np.mean(np.sum(d * d, axis=1)))), float(np.max(np.linalg.norm(d, axis=1)))

    est, n = run([], "tile_major")
    rmse, worst = relative_rmse(est, golden["poses"])
    print(f"ba_tum on the TUM copy of the e2e input vs the reference's kernels: pose RMSE {rmse:.2e} m (max {worst:.2e} m), {n} vs {int(golden['final_surfels'])} surfels")
    assert rmse <= 1e-5 and worst <= 1e-5 and rmse <= 3e-6, (rmse, worst)
    assert abs(n - int(golden["final_surfels"])) <= 1e-3 * int(golden["final_surfels"])
    est_rm, n_rm = run(["--row_major_creation"], "row_major")
    rmse_rm, worst_rm = relative_rmse(est_rm, golden["rowmajor_poses"])
    print(f"... with the reference's append order vs the unmodified reference run: pose RMSE {rmse_rm:.2e} m (max {worst_rm:.2e} m), {n_rm} vs {int(golden['rowmajor_final_surfels'])} surfels")
    assert rmse_rm <= 1e-5 and worst_rm <= 1e-5 and rmse_rm <= 3e-6, (rmse_rm, worst_rm)
    assert abs(n_rm - int(golden["rowmajor_final_surfels"])) <= 1e-3 * int(golden["rowmajor_final_surfels"])
