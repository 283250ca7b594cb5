"""End-to-end parity from RAW input at VGA: what the generator (tests/make_golden_e2e_vga.py: the REFERENCE's
own kernels on the host), the CPU replay (tests/test_cpu_e2e_vga.py: the oracle) and the GPU test (tests/test_gpu_e2e_vga.py: the
HIP path through vis::DirectBA) have to agree on -- the scene, the raw input (noisy u16 depth with holes + RGB, regenerated from
seeds on every machine and pinned by a digest in the golden file), the parameters of the chain, and the reference-side driver.

The chain, each side running ALL of it by its own code from the same raw arrays (B/ = applications/badslam/src/badslam/):
  BilateralFilteringAndDepthCutoffCUDA (B/bad_slam.cc:700-716, B/cuda_depth_processing.cu:42-128)
  -> Keyframe constructor: normals, radii, isolated-pixel removal, luma (B/keyframe.cc:96-144)
  -> DirectBA::BundleAdjustment(do_surfel_updates = true, three alternating iterations, increase_ba_iteration_count = true):
     filtered surfel creation for every keyframe (B/direct_ba.cc:340-405), activation, geometry step, merging + compaction,
     Gauss-Newton pose estimation of every keyframe (B/direct_ba_alternating.cc:345-718), and the end tasks: merging, deletion +
     radius update, compaction (B/direct_ba.cc:566-653).
Only OUTPUTS go into the golden file: keyframe poses, surfel counts per stage, 10^4 sampled surfels, a digest of the activity
flags -- once with exact bilinear weights and once with the texture unit's 8-bit weights (oracle/ref_shim/cuda_runtime.h)."""
import hashlib
import os

import numpy as np

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_vga.npz")
WIDTH, HEIGHT, KEYFRAMES, CELL = 640, 480, 20, 4
SCENE_SEED, NOISE_SEED, POSE_SEED = 61, 62, 63
BILATERAL = (3.0, 0.05, 2.5)            # sigma_xy, sigma_value (inverse depth), radius_factor: B/bad_slam_config.h defaults
MAX_DEPTH_M = 3.0                       # B/bad_slam_config.h: max_depth
MIN_OBSERVATIONS = 2
MERGE_FACTOR = 0.8
ITERATIONS = 3
SAMPLES = 10000
CAPACITY = 600000


def scene_and_raw_input():
    """The synthetic scene (20 VGA keyframes, all mutually co-visible), its raw input -- depth with sensor-like noise (sigma 6 raw
    units = 1.2 mm) and 1 % dropped pixels, RGB -- and the initial poses (5 mm / 1 mrad off the truth)."""
    from badslam_amd import synthetic
    scene = synthetic.make_scene(KEYFRAMES, WIDTH, HEIGHT, seed=SCENE_SEED, cell=CELL, translation_range=0.8, rotation_range=0.25)
    rng = np.random.Generator(np.random.PCG64(NOISE_SEED))
    raw = np.stack([(d.astype(np.float64) + rng.normal(0, 6, d.shape)).clip(0, 65000).astype(np.uint16) for d in scene.depth])
    raw[rng.random(raw.shape) < 0.01] = 0
    rgb = np.stack(scene.rgb).astype(np.uint8)
    prng = np.random.Generator(np.random.PCG64(POSE_SEED))
    start = np.asarray([synthetic.perturb_pose(prng, T) for T in scene.poses_gt], np.float64)
    return scene, raw, rgb, start


def input_digest(raw, rgb, start):
    h = hashlib.sha256(np.ascontiguousarray(raw).tobytes())
    h.update(np.ascontiguousarray(rgb).tobytes())
    h.update(np.ascontiguousarray(start, np.float32).tobytes())
    return np.frombuffer(h.digest(), np.uint8).copy()


def sample_indices(n):
    return (np.arange(SAMPLES, dtype=np.int64) * n) // SAMPLES


def flags_digest(active):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(active, np.uint8).tobytes()).digest(), np.uint8).copy()


def all_pairs_covisibility(K):
    return [[j for j in range(K) if j != k] for k in range(K)]


def summarise(poses, surfel_rows, n, counts, active_after_last_activation):
    idx = sample_indices(n)
    return dict(poses=np.asarray(poses, np.float64), counts=np.asarray(counts, np.int64), final_surfels=np.int64(n),
                sampled_rows=np.ascontiguousarray(surfel_rows[:8, idx]), active_digest=flags_digest(active_after_last_activation),
                active_count=np.int64(int(np.asarray(active_after_last_activation).sum())))


# ---- the reference side: its kernels on the host, driven in the order of B/direct_ba_alternating.cc --------------------------------
def reference_preprocess(scene, raw, rgb):
    """Bilateral filter, Keyframe-constructor kernels and luma by the reference's own code; returns the arrays per keyframe."""
    from oracle import ref_binding as rb
    s = scene.raw_to_float_depth
    cam = scene.camera
    cfactor = np.zeros(((HEIGHT - 1) // CELL + 1, (WIDTH - 1) // CELL + 1), np.float32)
    out = []
    for k in range(raw.shape[0]):
        filtered = rb.bilateral_filter_and_depth_cutoff(raw[k], *BILATERAL, int(MAX_DEPTH_M / s), s)
        pre = rb.keyframe_depth_preprocessing(filtered, list(cam), 0.0, s, scene.baseline_fx, CELL, cfactor)
        rgba = rb.compute_brightness(rgb[k])
        out.append(dict(depth=pre["depth"], normals=pre["normals"], radius=pre["radius"], rgba=rgba, min_depth=pre["min_depth"], max_depth=pre["max_depth"]))
    return out


def container(scene, images, start, capacity=CAPACITY):
    """An oracle scene used as a container for preprocessed keyframe images at the starting poses, empty cloud."""
    from oracle import binding as ob
    cam, cam2 = ob.make_camera(scene.camera, WIDTH, HEIGHT), ob.make_camera(scene.camera, WIDTH, HEIGHT)
    ba = ob.OracleBA(capacity, scene.raw_to_float_depth, scene.baseline_fx, CELL, cam, cam2, min_observation_count=MIN_OBSERVATIONS)
    ba.merge_factor = MERGE_FACTOR
    for k, im in enumerate(images):
        ba.add_preprocessed_keyframe(im["depth"], im["normals"], im["radius"], im["rgba"], start[k], float(im["min_depth"]), float(im["max_depth"]))
    ba.covis = all_pairs_covisibility(len(images))
    return ba


def tile_major_permutation(rows, frame_T_global, camera):
    """The order in which THIS backend appends the surfels a keyframe creates: 8 x 8-cell tiles of the creating keyframe, row-major
    inside a tile (DESIGN.md section 2 "surfel order"; kernels_lifecycle.hip: tile_seq) -- the reference appends them row-major over
    the whole image (B/kernel_create_surfels.cu:300-369).  A surfel's index is its identity, and which of two mergeable surfels
    survives DetermineSupportingSurfelsAndMergeSurfels is "the one inserted first" = the lower index (B/kernel_supporting_surfels.cu:
    60-86), so the order within a keyframe's block decides ~1 % of the survivors on this scene (measured by the generator).  The
    golden run therefore gives the reference's kernels the cloud in this backend's order: same surfels, permuted within each
    keyframe's block.  The creating pixel of a new surfel is where it projects in its keyframe (a pixel centre)."""
    F = np.asarray(frame_T_global, np.float64).reshape(3, 4)
    p = F[:, :3] @ rows[:3].astype(np.float64) + F[:, 3:4]
    fx, fy, cx, cy = [float(v) for v in camera]
    x = np.floor(fx * p[0] / p[2] + cx).astype(np.int64)
    y = np.floor(fy * p[1] / p[2] + cy).astype(np.int64)
    assert (x >= 0).all() and (x < WIDTH).all() and (y >= 0).all() and (y < HEIGHT).all()
    tp = 8 * CELL
    tpr = (WIDTH + tp - 1) // tp
    key = (((y // tp) * tpr + x // tp) * tp + y % tp) * tp + x % tp
    assert len(np.unique(key)) == len(key)                    # one surfel per pixel at most
    return np.argsort(key, kind="stable")


def run_reference(scene, raw, rgb, start, quantize_texture_weights, log=print, creation_order="tile-major"):
    """The whole chain by the reference's kernels.  Returns summarise(...)."""
    from oracle import ref_binding as rb
    images = reference_preprocess(scene, raw, rgb)
    orc = container(scene, images, start)
    K = len(images)
    ref = rb.ReferenceKernels(orc, quantize_texture_weights=quantize_texture_weights)
    ref.sc.surfels_size = 0
    covis = all_pairs_covisibility(K)
    poses = [np.asarray(T, np.float64) for T in start]
    counts = []
    surfel_count = 0
    active_snapshot = None
    for iteration in range(ITERATIONS):
        old_size = int(ref.sc.surfels_size)
        new_keyframes = list(range(K)) if iteration == 0 else []          # every keyframe is kActive and new to this BA call
        for k in new_keyframes:                                           # B/direct_ba_alternating.cc:373-418
            first = int(ref.sc.surfels_size)
            created = ref.create_surfels_for_keyframe(k, filter_new_surfels=True, covis=covis[k])
            surfel_count += created
            if creation_order == "tile-major" and created > 1:
                block = ref.surfel_data[:, first:first + created]
                block[:] = block[:, tile_major_permutation(block, list(ref.kfs[k].frame_T_global), scene.camera)]
        size = int(ref.sc.surfels_size)
        counts.append(size)
        # activation: old surfels by the kernel, new ones are active (B/direct_ba_alternating.cc:441-466)
        if old_size > 0:
            keep_size = ref.sc.surfels_size
            ref.sc.surfels_size = old_size
            ref.update_surfel_activation()
            ref.sc.surfels_size = keep_size
        ref.active[old_size:size] = 1
        active_snapshot = ref.active[:size].copy()
        assert not ref.pairs_outside_int_range().any()
        ref.optimize_geometry_iteration(True, True)
        if new_keyframes:                                                 # merging + compaction (:489-520)
            for k in new_keyframes:
                _, merged = ref.determine_supporting_surfels(k, merge=True, merge_dist_factor=MERGE_FACTOR)
                surfel_count -= merged
            new_size = rb.compact_surfels(ref.surfel_data, size, surfel_count, ref.active)
            assert new_size == surfel_count
            ref.sc.surfels_size = surfel_count
        counts.append(int(ref.sc.surfels_size))
        steps = 0
        for k in range(K):                                                # B/direct_ba_alternating.cc:547-575
            poses[k], n = ref.estimate_frame_pose(k, poses[k])
            steps += n
        for k in range(K):
            ref.set_pose(k, poses[k])
        log(f"  iteration {iteration}: {int(ref.sc.surfels_size)} surfels, {steps} Gauss-Newton steps")
    # end tasks (B/direct_ba.cc:566-653): merging for the keyframes active in this BA call, deletion + radii, compaction
    size = int(ref.sc.surfels_size)
    for k in range(K):
        _, merged = ref.determine_supporting_surfels(k, merge=True, merge_dist_factor=MERGE_FACTOR)
        surfel_count -= merged
    surfel_count -= ref.delete_surfels_and_update_radii(MIN_OBSERVATIONS)
    new_size = rb.compact_surfels(ref.surfel_data, size, surfel_count, None)
    assert new_size == surfel_count
    counts.append(int(surfel_count))
    return summarise(poses, ref.surfel_data, surfel_count, counts, active_snapshot)


# ---- the oracle side (CPU replay): the same chain by the oracle's restatement ------------------------------------------------------
def run_oracle(scene, raw, rgb, start):
    from oracle import binding as ob
    s = scene.raw_to_float_depth
    cam, cam2 = ob.make_camera(scene.camera, WIDTH, HEIGHT), ob.make_camera(scene.camera, WIDTH, HEIGHT)
    ba = ob.OracleBA(CAPACITY, s, scene.baseline_fx, CELL, cam, cam2, min_observation_count=MIN_OBSERVATIONS)
    ba.merge_factor = MERGE_FACTOR
    for k in range(raw.shape[0]):
        filtered = ob.bilateral_filter_and_depth_cutoff(raw[k], *BILATERAL, int(MAX_DEPTH_M / s), s)
        ba.add_keyframe(filtered, rgb[k], start[k])
    ba.covis = all_pairs_covisibility(raw.shape[0])
    ba.spatial_sort_cell = 0.0          # index-wise comparison with the reference's order
    stats = ba.bundle_adjustment(do_surfel_updates=True, optimize_poses=True, optimize_geometry=True, min_iterations=ITERATIONS,
                                 max_iterations=ITERATIONS, increase_ba_iteration_count=True)
    assert stats.iterations_done == ITERATIONS
    n = ba.surfels_size
    return dict(poses=np.asarray([ba.pose(k) for k in range(raw.shape[0])], np.float64), final_surfels=n,
                rows=ba.surfel_data[:8, :n].copy())


# ---- comparison: what "matches the reference" means here ---------------------------------------------------------------------------
def compare(got_poses, got_rows, golden, prefix, log=print):
    """Pose RMSE / maxima against the golden poses; surfel counts; the golden's sampled surfels against the nearest surfel of the
    result (index-free: a single different merge decision shifts every later index through compaction)."""
    from scipy.spatial import cKDTree
    ref_poses = golden[prefix + "poses"]
    rmse = float(np.sqrt(np.mean(np.sum((np.asarray(got_poses)[:, 4:] - ref_poses[:, 4:]) ** 2, axis=1))))
    dt = float(np.max(np.linalg.norm(np.asarray(got_poses)[:, 4:] - ref_poses[:, 4:], axis=1)))
    q = np.asarray(got_poses)[:, :4] * np.sign(np.sum(np.asarray(got_poses)[:, :4] * ref_poses[:, :4], axis=1, keepdims=True))
    dq = float(np.max(np.linalg.norm(q - ref_poses[:, :4], axis=1)))
    n_ref, n_got = int(golden[prefix + "final_surfels"]), got_rows.shape[1]
    sampled = golden[prefix + "sampled_rows"]
    tree = cKDTree(got_rows[:3].T.astype(np.float64))
    dist, _ = tree.query(sampled[:3].T.astype(np.float64))
    far = int(np.count_nonzero(dist > 1e-5))
    log(f"{prefix or 'exact '}: pose RMSE {rmse:.2e} m (max {dt:.2e} m, quaternion {dq:.1e}); surfels {n_got} vs {n_ref} ({abs(n_got - n_ref)} apart); "
        f"sampled surfels beyond 1e-5 m: {far} of {len(dist)} (median {np.median(dist):.1e}, p99 {np.percentile(dist, 99):.1e}, p99.9 {np.percentile(dist, 99.9):.1e}, "
        f"max {dist.max():.1e} m)")
    return dict(rmse=rmse, max_translation=dt, max_quaternion=dq, count_difference=abs(n_got - n_ref), n_ref=n_ref, far=far, samples=len(dist),
                p999=float(np.percentile(dist, 99.9)), max_distance=float(dist.max()))


def check(result):
    """What the CPU replay (oracle) and the GPU test (HIP path) assert against either reference variant.  BASELINE's bar is the
    pose RMSE; the rest is what was measured: the same number of surfels survives the whole lifecycle, and
    99 % of the reference's surfels have a surfel of the result within 1e-5 m -- the rest sit within 1e-4 m: their keyframes'
    normal images differ from the reference's on 0.07 % of the pixels (a +-1 raw unit rounding of the bilateral filter on ~20
    pixels per frame, the two sides' exponentials being one ulp apart), which flips an association test now and then."""
    assert result["rmse"] <= 1e-5 and result["max_translation"] <= 1e-5, result          # BASELINE.json: pose RMSE within 1e-5 m of reference
    assert result["rmse"] <= 3e-6, result                                                  # ... and what is actually reached
    assert result["count_difference"] <= 1e-3 * result["n_ref"], result                    # surfel flips <= 0.1 %
    assert result["far"] <= 0.015 * result["samples"], result
    assert result["p999"] <= 1e-4 and result["max_distance"] <= 2e-3, result
