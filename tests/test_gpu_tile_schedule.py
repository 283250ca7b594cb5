"""Heavy work first (badslam_amd/csrc/wave_cull.h: scheduled_tile): the sweeps take their surfel tiles by a schedule the pose
phase builds from its per-tile candidate counts.  It is a scheduling hint -- every tile must still be processed exactly once and
no result may depend on it."""
import ctypes as C

import numpy as np
import pytest

from tests import common

pytestmark = pytest.mark.gpu

HEAVY_SLOTS, PERM = 1024, 8 + 1024


def _schedule(g):
    lib, h = g.ctx.lib, g.ctx.handle
    padded = C.c_uint32()
    from badslam_amd import capi
    capi.check(lib.bahip_debug_read_tile_schedule(h, C.byref(padded), None, 0))
    if padded.value == 0:
        return 0, None
    words = (C.c_uint32 * (PERM + 2 * padded.value))()
    capi.check(lib.bahip_debug_read_tile_schedule(h, C.byref(padded), words, len(words)))
    return padded.value, np.frombuffer(words, np.uint32).copy()


def _iteration(g, poses):
    from badslam_amd import capi
    for k, kf in enumerate(g.keyframes):
        kf["activation"] = capi.KF_ACTIVE
        kf["pose"] = np.asarray(poses[k], np.float32)
    g.bind_keyframes()
    g.update_activation_and_optimize_geometry(True, True)
    out, its, conv, rounds = g.estimate_keyframe_poses(True, True)
    return out


@pytest.mark.parametrize("surfels_per_side", [1, 3])
def test_schedule_is_a_permutation_and_changes_no_bit(surfels_per_side):
    """Two alternating iterations with the schedule switched off and on (the second iteration's sweeps run scheduled): same
    surfels, flags and poses, bit for bit; and the schedule itself puts every tile at exactly one regular position, lists only
    flagged tiles as heavy, each once, and flags nothing else.  The larger scene (three copies of the cloud side by side in the
    buffer: > 8192 tiles would take the 128-tile runs; here it just makes more runs) has tiles of very different cost."""
    import torch
    from badslam_amd import capi
    torch.cuda.set_device(0)
    scene = common.small_scene(num_keyframes=6, seed=31)
    rng = np.random.Generator(np.random.PCG64(5))
    start = [common.synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    lib = capi.load()
    results = {}
    for enabled in (0, 1):
        capi.check(lib.bahip_debug_set_tile_order(enabled))
        try:
            g = common.build_gpu(scene, 900000)
            data = g.download_surfels()
            data = np.concatenate([data] * surfels_per_side, axis=1)           # the same surfels again: heavy and light stretches
            N = data.shape[1]
            g.upload_surfels(data, np.ones(N, np.uint8))
            poses = start
            for _ in range(3):
                poses = _iteration(g, poses)
            padded, words = _schedule(g)
            results[enabled] = dict(surfels=g.download_surfels(), active=g.active_buf.download().ravel()[:N].copy(), poses=np.asarray(poses),
                                    padded=padded, words=words, N=N)
        finally:
            capi.check(lib.bahip_debug_set_tile_order(1))
    off, on = results[0], results[1]
    assert off["padded"] == 0 and on["padded"] >= (on["N"] + 63) // 64
    assert np.array_equal(off["surfels"][:8].view(np.uint32), on["surfels"][:8].view(np.uint32))
    assert np.array_equal(off["active"], on["active"])
    assert np.array_equal(off["poses"], on["poses"])
    padded, w = on["padded"], on["words"]
    heavy_count = int(w[0])
    heavy = w[8:8 + heavy_count]
    perm, flags = w[PERM:PERM + padded], w[PERM + padded:PERM + 2 * padded]
    assert heavy_count <= HEAVY_SLOTS
    assert np.array_equal(np.sort(perm), np.arange(padded, dtype=np.uint32))           # every tile at exactly one regular position
    assert len(set(heavy.tolist())) == heavy_count                                       # no tile twice in the heavy list
    assert np.array_equal(np.sort(heavy), np.nonzero(flags)[0].astype(np.uint32))       # flagged <=> listed


def test_run_order_is_rebuilt_after_the_surfels_were_rearranged():
    """Round 5: the order describes an ARRANGEMENT of the surfels.  The Morton sort, a compaction that moves something and the hint
    for outside writers (bahip_context_surfels_rearranged) drop it; the next pose phase takes the census again and the order is back --
    a permutation of the tiles as before.  (It used to survive a reorder for up to 32 pose phases: the sweeps then ran 10-15 % slower,
    results unchanged, and which bench figure suffered was a lottery of the phase counter.)"""
    import torch
    from badslam_amd import capi
    torch.cuda.set_device(0)
    scene = common.small_scene(num_keyframes=6, seed=31)
    rng = np.random.Generator(np.random.PCG64(5))
    poses = [common.synthetic.perturb_pose(rng, T) for T in scene.poses_gt]
    g = common.build_gpu(scene, 900000)
    lib, h = g.ctx.lib, g.ctx.handle
    for _ in range(2):
        poses = _iteration(g, poses)
    padded, _ = _schedule(g)
    assert padded > 0
    for rearrange in ("sort", "hint", "sort"):
        if rearrange == "sort":
            g.sort_surfels_spatially(0.02)
        else:
            capi.check(lib.bahip_context_surfels_rearranged(h))
        assert _schedule(g)[0] == 0                      # dropped: the sweeps of the next iteration run in buffer order ...
        poses = _iteration(g, poses)
        padded_again, words = _schedule(g)               # ... and its pose phase has rebuilt it
        assert padded_again == padded
        assert np.array_equal(np.sort(words[PERM:PERM + padded]), np.arange(padded, dtype=np.uint32))
