"""The alternating loop driven by the device (bahip_alternating_iterations, used by vis::DirectBA::BundleAdjustment whenever
poses + geometry are optimised over a fixed surfel set): every iteration is queued at once, the last solve launch of each pose
phase evaluates the stopping rule of B/direct_ba_alternating.cc:693-701 on the device, and the host waits once.  Against the loop
driven through the stage functions (one host wait per Gauss-Newton round, bahip_debug_set_device_loop(0)): the same iteration
counts, convergence flags, statistics, poses, activations and surfels, bit for bit -- with a fixed number of iterations, with a
loop that ends by convergence, and when a pose phase needs more rounds than were queued for it (the host finishes that phase)."""
import numpy as np
import pytest

from badslam_amd import capi, synthetic
from tests import common

pytestmark = pytest.mark.gpu


def _run(scene, perturbed, surfels, device_loop, rounds_ahead, fused_begin=True, **ba_args):
    from badslam_amd.directba import DirectBA
    lib = capi.load()
    capi.check(lib.bahip_debug_set_device_loop(int(device_loop)))
    capi.check(lib.bahip_debug_set_fused_iteration_begin(int(fused_begin)))
    capi.check(lib.bahip_debug_set_pose_rounds_ahead(rounds_ahead))
    try:
        ba = DirectBA(600000, scene.raw_to_float_depth, scene.baseline_fx, scene.cell, scene.width, scene.height, scene.camera, scene.camera)
        for k in range(len(scene.depth)):
            ba.AddKeyframe(scene.depth[k], scene.rgb[k], scene.poses_gt[k])
        if surfels is None:
            for k in range(len(scene.depth)):
                ba.CreateSurfelsForKeyframe(k, filter_new_surfels=False)
            surfels = ba.download_surfels(8)
            rng = np.random.Generator(np.random.PCG64(5))
            surfels[2] += rng.uniform(0, 0.004, surfels.shape[1]).astype(np.float32)
        ba.upload_surfels(surfels)
        for k, T in enumerate(perturbed):
            ba.set_keyframe_pose(k, T)
        ba.set_ba_iteration_counts(1, 1)      # no end tasks: the surfel set and its order stay
        import ctypes as C
        h0, d0, h1, d1 = C.c_longlong(), C.c_longlong(), C.c_longlong(), C.c_longlong()
        capi.check(lib.bahip_debug_alternating_loop_calls(C.byref(h0), C.byref(d0)))
        done, conv = ba.BundleAdjustment(do_surfel_updates=False, optimize_poses=True, optimize_geometry=True,
                                         increase_ba_iteration_count=False, **ba_args)
        capi.check(lib.bahip_debug_alternating_loop_calls(C.byref(h1), C.byref(d1)))
        # the call went where it was sent (round 5: the switch's initialiser was miscompiled and EVERY call was declined; results are
        # the same either way, so only this counter can tell)
        assert (h1.value - h0.value, d1.value - d0.value) == ((1, 0) if device_loop else (0, 1))
        K = len(perturbed)
        return dict(done=done, conv=conv, stats=ba.last_stats(), poses=np.asarray([ba.keyframe_pose(k) for k in range(K)], np.float32),
                    activation=[ba.keyframe_activation(k) for k in range(K)], surfels=ba.download_surfels(8)), surfels
    finally:
        capi.check(lib.bahip_debug_set_device_loop(1))
        capi.check(lib.bahip_debug_set_fused_iteration_begin(0))
        capi.check(lib.bahip_debug_set_pose_rounds_ahead(0))


@pytest.mark.parametrize("fused_begin", [True, False], ids=["phase end opens the next iteration", "iteration_begin launch"])
@pytest.mark.parametrize("case", ["fixed iterations", "until converged", "phase outruns the queue", "single keyframe"])
def test_device_driven_loop_is_the_host_driven_loop(case, fused_begin):
    scene = common.small_scene(num_keyframes=1 if case == "single keyframe" else 6, seed=33)
    rng = np.random.Generator(np.random.PCG64(4))
    perturbed = [synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    args = dict(min_iterations=4, max_iterations=4) if case in ("fixed iterations", "phase outruns the queue") else dict(min_iterations=1, max_iterations=40)
    # (six keyframes, window 0 .. 5: the fixed set of B/direct_ba_alternating.cc:338-361, re-applied by every iteration's top; one
    # keyframe, window 0 .. 0: no fixed set, the top of an iteration is the co-visible propagation)
    ahead = 1 if case == "phase outruns the queue" else 0
    ref, surfels = _run(scene, perturbed, None, device_loop=False, rounds_ahead=1, **args)
    got, _ = _run(scene, perturbed, surfels, device_loop=True, rounds_ahead=ahead, fused_begin=fused_begin, **args)
    assert got["done"] == ref["done"] and got["conv"] == ref["conv"], (got["done"], ref["done"], got["conv"], ref["conv"])
    if case in ("until converged", "single keyframe"):
        assert ref["conv"] and 1 <= ref["done"] < 40
    if case == "phase outruns the queue":
        assert ref["stats"]["pose_rounds"] > ref["done"]                     # phases of more than one round: the hand-over was exercised
    assert got["stats"]["pose_rounds"] == ref["stats"]["pose_rounds"] and got["stats"]["pose_steps"] == ref["stats"]["pose_steps"], (got["stats"], ref["stats"])
    assert got["activation"] == ref["activation"]
    assert np.array_equal(got["poses"], ref["poses"])
    assert np.array_equal(got["surfels"].view(np.uint32), ref["surfels"].view(np.uint32))
    if case != "single keyframe":          # (one keyframe against its own, slightly displaced surfels: microns)
        assert np.abs(got["poses"] - np.asarray(perturbed, np.float32)).max() > 1e-4     # the loop did move the poses


def test_the_default_configuration_takes_the_device_driven_loop():
    """In a fresh process, with nothing set: vis::DirectBA::BundleAdjustment over poses + geometry is handled by
    bahip_alternating_iterations (the tests above switch the loop on explicitly and restore `1` afterwards, which hid a default of 0 --
    the initialiser of the switch had been given another lambda's body by the compiler -- for half a round)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("BAHIP_")}
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--keyframes", "12", "--surfels", "100000",
                           "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert proc.returncode == 0, proc.stderr[-3000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.strip()][-1])
    assert out["loop"]["timed_calls_driven_by_the_device"] == 1 and out["loop"]["timed_calls_driven_by_the_host"] == 0, out["loop"]


def test_the_hybrid_shape_of_the_geometry_step_is_the_other_shapes():
    """kernels_surfel.hip: geometry_hybrid_kernel -- on a cloud of few tiles the heavy tiles of the sweeps' run order take four
    wavefronts (a keyframe class each), every other tile one, in one launch.  The per-surfel sums are defined per class, so the same
    BundleAdjustment call must end with the same bits whether every tile takes one wavefront, every tile four, or the hybrid shape
    is used -- which this small scene takes by default from the second iteration on (the run order comes out of the first pose
    phase's census); the route is asserted."""
    import ctypes as C
    lib = capi.load()
    scene = common.small_scene(num_keyframes=6, seed=33)
    rng = np.random.Generator(np.random.PCG64(4))
    perturbed = [synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    runs, surfels = {}, None
    try:
        for shape in (1, 4, 5, 0):
            capi.check(lib.bahip_debug_set_launch_shapes(shape, 0))
            before = C.c_longlong()
            capi.check(lib.bahip_debug_geometry_hybrid_launches(C.byref(before)))
            runs[shape], surfels = _run(scene, perturbed, surfels, device_loop=True, rounds_ahead=0, fused_begin=False, min_iterations=5, max_iterations=5)
            after = C.c_longlong()
            capi.check(lib.bahip_debug_geometry_hybrid_launches(C.byref(after)))
            hybrid = after.value - before.value
            assert (hybrid >= 3) if shape in (5, 0) else (hybrid == 0), (shape, hybrid)
    finally:
        capi.check(lib.bahip_debug_set_launch_shapes(0, 0))
    for shape in (4, 5, 0):
        assert runs[shape]["done"] == runs[1]["done"] == 5
        assert np.array_equal(runs[shape]["poses"].view(np.uint32), runs[1]["poses"].view(np.uint32)), shape
        assert np.array_equal(runs[shape]["surfels"].view(np.uint32), runs[1]["surfels"].view(np.uint32)), shape
