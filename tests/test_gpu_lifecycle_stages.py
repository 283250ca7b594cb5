"""Stage-level parity of the surfel lifecycle ("next round" item 2): every C-ABI entry point of
B/kernel_supporting_surfels.cu:45, B/kernel_create_surfels.cu:213-356, B/kernel_delete_surfels.cu:84-133 and
B/kernel_compact_surfels.cu:101-279 against the oracle's restatement of the same function, bit for bit -- not only the
survivor counts of a whole BundleAdjustment call."""
import ctypes as C

import numpy as np
import pytest

from badslam_amd import capi, synthetic
from tests import common

pytestmark = pytest.mark.gpu

NAN_BITS = 0x7fffffff     # deleted-surfel marker, B/kernel_delete_surfels.cu:145


@pytest.fixture()
def world():
    """Four keyframes; surfels created (unfiltered) from the first three on both sides."""
    scene = common.small_scene(num_keyframes=4, seed=23)
    orc = common.build_oracle(scene, 500000, create_from=[0, 1, 2])
    g = common.build_gpu(scene, 500000, create_from=[0, 1, 2])
    ref, _ = common.oracle_surfels(orc)
    assert np.array_equal(g.download_surfels()[:8].view(np.uint32), ref[:8].view(np.uint32))
    return scene, orc, g


def _sync(orc, g, data, count=None):
    n = data.shape[1]
    orc.surfel_data[:, :n] = data
    orc.surfels.surfels_size = n
    orc.surfels.surfel_count = n if count is None else count
    g.upload_surfels(data, np.zeros(n, np.uint8))
    g.surfel_count = n if count is None else count


def _rows(a):
    return np.ascontiguousarray(a[:8]).view(np.uint32)


def test_supporting_surfels_lists_bit_exact(world):
    scene, orc, g = world
    for k in (3, 0):        # a keyframe that created no surfels yet, and one that did
        F = np.array(list(orc.keyframes[k].frame_T_global), np.float32)
        planes, merged = g.determine_supporting_surfels(k, F, merge=False)
        ref = orc.determine_supporting_surfels(k, merge=False)
        assert merged == 0
        assert np.array_equal(planes, ref), k
        filled = (ref != 0xffffffff).sum(axis=(1, 2))
        assert filled[0] > 5000 and filled[1] > 100, filled           # second slots are in use: cells seen by several surfels


@pytest.mark.parametrize("batch", [False, True, "frames", "pipelined", "pipelined without lists", "cells"],
                         ids=["per keyframe", "lifecycle batch", "lifecycle batch that knows its frames", "pipelined batch", "pipelined batch without tile lists",
                              "batch by cell lists"])
def test_merge_bit_exact(world, batch, request):
    """batch: inside bahip_lifecycle_batch_begin / _end the sweeps skip the tiles a keyframe cannot see (here: a copy of the cloud
    50 m away, in the middle of the buffer) -- the oracle knows no such bracket and must see the same buffer.
    pipelined: bahip_merge_surfels_for_keyframes, the whole batch in one call with keyframe j's apply sweep beside keyframe j + 1's
    insert sweep (two sets of planes; a surfel keyframe j merges away must not enter keyframe j + 1's planes although its NaN is being
    written in the same launch) -- against the oracle's keyframe-by-keyframe merges; the keyframes overlap, so the order matters."""
    import contextlib
    scene, orc, g = world
    lib = capi.load()
    by_cells = batch == "cells"      # bahip_merge_surfels_for_keyframes by cell lists (merge_cells_kernel): the members of every (frame, cell) up front,
    if by_cells:                     # one launch per frame -- same deletions; "pipelined" keeps round 6's first form under test
        batch = "pipelined"
    else:
        capi.check(lib.bahip_debug_set_merge_cells(0))
        request.addfinalizer(lambda: lib.bahip_debug_set_merge_cells(1))
    cells_before = C.c_longlong()
    capi.check(lib.bahip_debug_merge_cells_batches(C.byref(cells_before)))
    data, _ = common.oracle_surfels(orc)
    if batch:
        far = data.copy()
        far[0] += 50.0
        data = np.concatenate([data[:, :data.shape[1] // 2], far, data[:, data.shape[1] // 2:]], axis=1)
    n = data.shape[1]
    # near-duplicates of a third of the cloud (1 mm away, same normal): candidates for merging, appended after the originals
    rng = np.random.Generator(np.random.PCG64(3))
    pick = np.sort(rng.choice(n, n // 3, replace=False))
    dup = data[:, pick].copy()
    dup[:3] += rng.normal(0, 0.001, (3, len(pick))).astype(np.float32)
    both = np.concatenate([data, dup], axis=1)
    _sync(orc, g, both)
    total_merged = 0
    frames = [np.array(list(orc.keyframes[k].frame_T_global), np.float32) for k in (0, 1, 3)] if batch in ("frames", "pipelined") else None
    if batch in ("pipelined", "pipelined without lists"):
        order = (0, 1, 3, 2, 0)          # (a keyframe twice: its second visit finds its first visit's deletions)
        Fs = [np.array(list(orc.keyframes[k].frame_T_global), np.float32) for k in order]
        before = int(orc.surfels.surfel_count)
        with (g.lifecycle_batch(frames=Fs) if batch == "pipelined" else contextlib.nullcontext()):
            planes, total_merged = g.merge_surfels_for_keyframes(order, Fs, merge_dist_factor=orc.merge_factor)
        for k in order:
            orc.determine_supporting_surfels(k, merge=True)
        assert total_merged == before - int(orc.surfels.surfel_count)
        cells_after = C.c_longlong()
        capi.check(lib.bahip_debug_merge_cells_batches(C.byref(cells_after)))
        assert cells_after.value - cells_before.value == (1 if by_cells else 0)
        assert np.all(planes == 0xffffffff)
        got = g.surfel_buf.download()[:, :both.shape[1]]
        assert np.array_equal(_rows(got), _rows(orc.surfel_data[:, :both.shape[1]]))
        # ... and a second batch on the same context (the planes were left empty, the decision words are cleared again)
        before = int(orc.surfels.surfel_count)
        _, again = g.merge_surfels_for_keyframes((3, 1), [Fs[2], Fs[1]], merge_dist_factor=orc.merge_factor)
        for k in (3, 1):
            orc.determine_supporting_surfels(k, merge=True)
        assert again == before - int(orc.surfels.surfel_count)
        total_merged += again
        assert np.array_equal(_rows(g.surfel_buf.download()[:, :both.shape[1]]), _rows(orc.surfel_data[:, :both.shape[1]]))
    with (g.lifecycle_batch(frames=frames) if batch in (True, "frames") else contextlib.nullcontext()):
        for k in (0, 1, 3) if batch in (False, True, "frames") else ():
            F = np.array(list(orc.keyframes[k].frame_T_global), np.float32)
            before = int(orc.surfels.surfel_count)
            planes, merged = g.determine_supporting_surfels(k, F, merge=True, merge_dist_factor=orc.merge_factor)
            ref = orc.determine_supporting_surfels(k, merge=True)
            assert merged == before - int(orc.surfels.surfel_count), k
            if batch == "frames":
                # a batch that knows its frames owns the supporting planes: every merge call leaves them empty for the next keyframe
                # (merge_apply_kernel), the lists are not an output there (include/badslam_hip.h: bahip_lifecycle_batch_set_frames)
                assert np.all(planes == 0xffffffff), k
            else:
                assert np.array_equal(planes, ref), k
            got = g.surfel_buf.download()[:, :both.shape[1]]
            assert np.array_equal(_rows(got), _rows(orc.surfel_data[:, :both.shape[1]])), k     # the same surfels carry the NaN marker
            total_merged += merged
    assert total_merged > 1000, total_merged
    assert g.surfel_count == int(orc.surfels.surfel_count)
    # compaction after merging (what the BA loop does next, B/direct_ba_alternating.cc:505-520), with the active flags
    act = (np.arange(both.shape[1]) % 3 == 0).astype(np.uint8)
    orc.active[:both.shape[1]] = act
    a = np.zeros((1, g.capacity), np.uint8); a[0, :both.shape[1]] = act
    g.active_buf.upload(a)
    g.compact_surfels(with_active=True)
    orc.compact_surfels()
    m = orc.surfels_size
    assert g.surfels_size == m == both.shape[1] - total_merged
    assert np.array_equal(_rows(g.download_surfels()), _rows(orc.surfel_data[:, :m]))
    assert np.array_equal(g.active_buf.download()[0, :m], orc.active[:m])
    assert not np.any(_rows(orc.surfel_data[:, :m])[0] == NAN_BITS)


@pytest.mark.parametrize("batch", [False, True, "frames"], ids=["per keyframe", "lifecycle batch", "lifecycle batch that knows its frames"])
@pytest.mark.parametrize("min_obs", [1, 2, 3])
def test_filtered_creation_bit_exact(min_obs, batch):
    """CreateSurfelsForKeyframe with filter_new_surfels: observation / free-space-violation counting over the co-visible
    keyframes (B/kernel_create_surfels.cu:213-356), for complete and partial co-visibility lists."""
    scene = common.small_scene(num_keyframes=5, seed=29)
    rng = np.random.Generator(np.random.PCG64(7))
    # slightly wrong poses for two keyframes: some new surfels then violate free space in the others and are filtered
    poses = [T if k in (0, 2, 4) else synthetic.perturb_pose(rng, T, 0.03, 0.01) for k, T in enumerate(scene.poses_gt)]
    orc = common.build_oracle(scene, 600000, poses=poses, create_from=[], min_observation_count=min_obs)
    g = common.build_gpu(scene, 600000, poses=poses, create_from=[])
    plan = [(0, [1, 2, 3, 4]), (1, [0, 2]), (2, [4]), (3, [0, 1, 2, 4]), (4, [])]
    created = []
    import contextlib
    for k, covis in plan:
        # (a batch opened when the cloud already holds surfels: the bounded tiles, and behind them what this batch appends)
        with (g.lifecycle_batch(keyframes=[k] if batch == "frames" else None) if batch and k >= 2 else contextlib.nullcontext()):
            n_ref = orc.create_surfels_for_keyframe(k, filter_new_surfels=True, covis=covis)
            n_got = g.create_surfels_for_keyframe(k, filter_new_surfels=True, min_observation_count=min_obs, covis=covis)
        assert n_got == n_ref, (k, n_got, n_ref)
        created.append(n_ref)
        assert np.array_equal(_rows(g.download_surfels()), _rows(orc.surfel_data[:, :orc.surfels_size])), k
    unfiltered = common.build_oracle(scene, 600000, poses=poses, create_from=[0]).surfels_size
    if min_obs >= 2:
        assert 0 < created[0] < unfiltered            # the filter removed something and kept something
        assert created[4] == 0                        # no co-visible keyframe: a single observation is not enough
    else:
        assert sum(created) > 10000


@pytest.mark.parametrize("min_obs,append_groups", [(1, 0), (2, 0), (2, 3), (1, 1)])
def test_creation_batch_is_the_sequence_of_creations(min_obs, append_groups, request):
    """bahip_create_surfels_for_keyframes -- the creations of a batch of keyframes with the cloud's size on the device in between --
    against the oracle's creations one by one: the same surfels in the same places, also when the batch starts on an empty cloud,
    when a keyframe of it has no co-visible keyframe, and inside a lifecycle batch (tile bounds).  append_groups: the scan + append
    launch with its grid clamped as on a device that holds only three / one of its workgroups at once (its grid handshake needs every
    workgroup resident: ADVICE r5; bahip_debug_set_append_groups)."""
    capi.check(capi.load().bahip_debug_set_append_groups(append_groups))
    request.addfinalizer(lambda: capi.load().bahip_debug_set_append_groups(0))
    scene = common.small_scene(num_keyframes=5, seed=29)
    rng = np.random.Generator(np.random.PCG64(7))
    poses = [T if k in (0, 2, 4) else synthetic.perturb_pose(rng, T, 0.03, 0.01) for k, T in enumerate(scene.poses_gt)]
    orc = common.build_oracle(scene, 600000, poses=poses, create_from=[], min_observation_count=min_obs)
    g = common.build_gpu(scene, 600000, poses=poses, create_from=[])
    first, second = [(0, [1, 2, 3, 4]), (1, [0, 2])], [(2, [4]), (3, [0, 1, 2, 4]), (4, [])]
    for plan in (first, second):
        n_ref = sum(orc.create_surfels_for_keyframe(k, filter_new_surfels=True, covis=covis) for k, covis in plan)
        with g.lifecycle_batch(keyframes=[k for k, _ in plan] if min_obs == 2 else None):     # (with and without per-frame tile lists)
            n_got = g.create_surfels_for_keyframes(plan, filter_new_surfels=True, min_observation_count=min_obs)
        assert n_got == n_ref > 0, (n_got, n_ref)
        assert g.surfels_size == orc.surfels_size
        assert np.array_equal(_rows(g.download_surfels()), _rows(orc.surfel_data[:, :orc.surfels_size]))


def _chain_batches():
    n = C.c_longlong()
    capi.check(capi.load().bahip_debug_creation_chain_batches(C.byref(n)))
    return int(n.value)


@pytest.mark.parametrize("min_obs,append_groups,chain", [(2, 0, True), (1, 0, True), (2, 3, True), (2, 0, False)])
def test_creation_chain_is_the_sequence_of_creations(min_obs, append_groups, chain, request):
    """The creation batch as a chain of ONE launch per keyframe (kernels_lifecycle.hip: create_chain_kernel -- occupancy of the cloud at
    the batch's begin, candidates and filter for all keyframes up front; then per keyframe the append, the push of what it appended
    into the next keyframe's occupancy and the pull of what the batch appended before) against the oracle's creations one by one: the
    same surfels at the same indices.  Seven keyframes: one creates alone, six form the batch (pushes, pulls over a growing tail, the
    last keyframe on the old path); the route is asserted."""
    lib = capi.load()
    capi.check(lib.bahip_debug_set_append_groups(append_groups))
    capi.check(lib.bahip_debug_set_creation_chain(1 if chain else 0))
    request.addfinalizer(lambda: (lib.bahip_debug_set_append_groups(0), lib.bahip_debug_set_creation_chain(1)))
    scene = common.small_scene(num_keyframes=7, seed=31)
    rng = np.random.Generator(np.random.PCG64(11))
    poses = [T if k in (0, 2, 4, 6) else synthetic.perturb_pose(rng, T, 0.03, 0.01) for k, T in enumerate(scene.poses_gt)]
    orc = common.build_oracle(scene, 900000, poses=poses, create_from=[], min_observation_count=min_obs)
    g = common.build_gpu(scene, 900000, poses=poses, create_from=[])
    assert orc.create_surfels_for_keyframe(3, filter_new_surfels=True, covis=[0, 1, 2, 4, 5, 6]) == \
        g.create_surfels_for_keyframe(3, filter_new_surfels=True, min_observation_count=min_obs, covis=[0, 1, 2, 4, 5, 6]) > 0
    plan = [(0, [1, 2, 3, 4, 5, 6]), (1, [0, 2]), (2, [4, 6]), (4, [0, 1, 2, 3, 5, 6]), (5, []), (6, [0, 5])]
    n_ref = [orc.create_surfels_for_keyframe(k, filter_new_surfels=True, covis=covis) for k, covis in plan]
    before = _chain_batches()
    with g.lifecycle_batch(keyframes=[k for k, _ in plan]):
        n_got = g.create_surfels_for_keyframes(plan, filter_new_surfels=True, min_observation_count=min_obs)
    assert _chain_batches() - before == (1 if chain else 0)
    assert n_got == sum(n_ref) and sum(1 for n in n_ref if n > 0) >= (4 if min_obs == 1 else 2), (n_got, n_ref)
    assert g.surfels_size == orc.surfels_size
    assert np.array_equal(_rows(g.download_surfels()), _rows(orc.surfel_data[:, :orc.surfels_size]))


def test_creation_chain_respects_the_capacity():
    """A keyframe in the middle of the chain that does not fit creates nothing and raises the flag; the keyframes behind it see the
    cloud without it (B/kernel_create_surfels.cc:162-165, per keyframe)."""
    scene = common.small_scene(num_keyframes=5, seed=29)
    lib = capi.load()
    runs = []
    for chain in (0, 1):
        capi.check(lib.bahip_debug_set_creation_chain(chain))
        try:
            probe = common.build_gpu(scene, 600000, create_from=[])
            first = probe.create_surfels_for_keyframe(0, filter_new_surfels=False)
            second = probe.create_surfels_for_keyframe(1, filter_new_surfels=False)
            third = probe.create_surfels_for_keyframe(2, filter_new_surfels=False)
            assert min(first, second, third) > 0
            g = common.build_gpu(scene, first + second + third // 2, create_from=[])
            assert g.create_surfels_for_keyframe(0, filter_new_surfels=False) == first
            before = _chain_batches()
            with g.lifecycle_batch(keyframes=[1, 2, 3, 4]):
                created = g.create_surfels_for_keyframes([(1, None), (2, None), (3, None), (4, None)], filter_new_surfels=False)
            assert _chain_batches() - before == chain
            assert g.ctx.lib.bahip_context_take_capacity_exceeded(g.ctx.handle) == 1
            runs.append((created, _rows(g.download_surfels()).copy()))
        finally:
            capi.check(lib.bahip_debug_set_creation_chain(1))
    assert runs[0][0] == runs[1][0] >= second and np.array_equal(runs[0][1], runs[1][1])


def test_creation_batch_respects_the_capacity():
    """A keyframe of the batch that does not fit creates nothing and raises the flag; the ones before it did create
    (B/kernel_create_surfels.cc:162-165, per keyframe)."""
    scene = common.small_scene(num_keyframes=3, seed=29)
    full = common.build_gpu(scene, 600000, create_from=[])
    counts = [full.create_surfels_for_keyframe(k, filter_new_surfels=False) for k in range(3)]
    assert min(counts[:2]) > 0
    g = common.build_gpu(scene, counts[0] + counts[1] // 2, create_from=[])
    created = g.create_surfels_for_keyframes([(0, None), (1, None)], filter_new_surfels=False)
    assert created == counts[0]
    assert g.ctx.lib.bahip_context_take_capacity_exceeded(g.ctx.handle) == 1
    assert np.array_equal(_rows(g.download_surfels()), _rows(full.download_surfels()[:, :counts[0]]))


def test_delete_and_update_radii_then_compact_bit_exact(world):
    scene, orc, g = world
    data, _ = common.oracle_surfels(orc)
    n = data.shape[1]
    rng = np.random.Generator(np.random.PCG64(5))
    data = data.copy()
    far = rng.choice(n, n // 10, replace=False)
    data[2, far] += 0.6                                   # behind the surface: unobserved or free-space violating -> deleted
    near = rng.choice(n, n // 10, replace=False)
    data[2, near] -= 0.5                                  # in front of the surface: free-space violations in other keyframes
    data[4] *= 4.0                                        # inflated radii: the update takes the smallest observed radius
    for min_obs in (1, 2):
        _sync(orc, g, data)
        g.bind_keyframes()
        deleted = g.delete_surfels_and_update_radii(min_obs)
        ref_deleted = orc.delete_surfels_and_update_radii(min_obs)
        assert deleted == ref_deleted, (min_obs, deleted, ref_deleted)
        assert n // 20 < deleted < n // 2, deleted
        got = g.surfel_buf.download()[:, :n]
        assert np.array_equal(_rows(got), _rows(orc.surfel_data[:, :n])), min_obs
        assert np.count_nonzero(got[4] != data[4]) > n // 2                   # radii were updated
        g.compact_surfels(with_active=False)                                  # B/direct_ba.cc:619: no active-flag buffer here
        active = orc.surfels.active
        orc.surfels.active = None
        orc.compact_surfels()
        orc.surfels.active = active
        m = orc.surfels_size
        assert g.surfels_size == m == n - deleted
        assert np.array_equal(_rows(g.download_surfels()), _rows(orc.surfel_data[:, :m]))


@pytest.mark.parametrize("seed,width,height,cell,keyframes", [(41, 320, 240, 2, 9), (42, 200, 152, 3, 8), (43, 320, 240, 1, 6), (44, 168, 120, 4, 10)])
def test_lifecycle_batches_on_other_scenes(seed, width, height, cell, keyframes):
    """The round-6 forms of both lifecycle batches -- creation as a chain, merging by cell lists -- on scenes of other sizes, sparse-cell
    sizes (also ones that do not divide the image) and batch lengths, through three rounds of [creation batch, merge batch] with
    perturbed poses (surfels appear, overlap and merge), against the oracle's keyframe-by-keyframe calls: the same surfels at the same
    indices, the same deletions, after every batch."""
    lib = capi.load()
    scene = common.small_scene(num_keyframes=keyframes, width=width, height=height, seed=seed, cell=cell)
    rng = np.random.Generator(np.random.PCG64(seed))
    poses = [T if k % 3 == 0 else synthetic.perturb_pose(rng, T, 0.02, 0.008) for k, T in enumerate(scene.poses_gt)]
    capacity = 2 * keyframes * (width // cell + 1) * (height // cell + 1)
    orc = common.build_oracle(scene, capacity, poses=poses, create_from=[], min_observation_count=2)
    g = common.build_gpu(scene, capacity, poses=poses, create_from=[])
    everyone = list(range(keyframes))
    assert orc.create_surfels_for_keyframe(0, filter_new_surfels=False) == g.create_surfels_for_keyframe(0, filter_new_surfels=False) > 0
    chains, cells = _chain_batches(), C.c_longlong()
    capi.check(lib.bahip_debug_merge_cells_batches(C.byref(cells)))
    for round_ in range(3):
        order = everyone[round_:] + everyone[:round_]
        plan = [(k, [c for c in everyone if c != k and (c + k + round_) % 4 != 0]) for k in order]
        n_ref = sum(orc.create_surfels_for_keyframe(k, filter_new_surfels=True, covis=covis) for k, covis in plan)
        with g.lifecycle_batch(keyframes=order):
            n_got = g.create_surfels_for_keyframes(plan, filter_new_surfels=True, min_observation_count=2)
        assert n_got == n_ref and g.surfels_size == orc.surfels_size, (round_, n_got, n_ref)
        assert np.array_equal(_rows(g.download_surfels()), _rows(orc.surfel_data[:, :orc.surfels_size])), round_
        Fs = [np.array(list(orc.keyframes[k].frame_T_global), np.float32) for k in order]
        before = int(orc.surfels.surfel_count)
        with g.lifecycle_batch(frames=Fs):
            _, merged = g.merge_surfels_for_keyframes(order, Fs, merge_dist_factor=orc.merge_factor)
        for k in order:
            orc.determine_supporting_surfels(k, merge=True)
        assert merged == before - int(orc.surfels.surfel_count), round_
        assert np.array_equal(_rows(g.surfel_buf.download()[:, :orc.surfels_size]), _rows(orc.surfel_data[:, :orc.surfels_size])), round_
        g.compact_surfels(with_active=True)
        orc.compact_surfels()
        assert g.surfels_size == orc.surfels_size
        assert np.array_equal(_rows(g.download_surfels()), _rows(orc.surfel_data[:, :orc.surfels_size])), round_
    after = C.c_longlong()
    capi.check(lib.bahip_debug_merge_cells_batches(C.byref(after)))
    assert _chain_batches() - chains == 3 and after.value - cells.value == 3


def test_merge_by_cell_lists_with_crowded_cells(world):
    """Cells with more members than merge_pairs_kernel requests at once (eight): a few hundred surfels of the cloud are repeated fourteen
    times each, a fraction of a millimetre apart and with slightly different normals and radii, so that their cells hold 15 and more
    associated surfels of which some merge and some do not -- the batch by cell lists against the oracle's keyframe-by-keyframe merges,
    twice over the keyframes (the second round finds the first round's deletions)."""
    scene, orc, g = world
    lib = capi.load()
    data, _ = common.oracle_surfels(orc)
    n = data.shape[1]
    rng = np.random.Generator(np.random.PCG64(17))
    pick = np.sort(rng.choice(n, 300, replace=False))
    crowd = np.repeat(data[:, pick], 14, axis=1).copy()
    crowd[:3] += rng.normal(0, 0.0004, (3, crowd.shape[1])).astype(np.float32)
    crowd[4] *= rng.uniform(0.2, 3.0, crowd.shape[1]).astype(np.float32)     # radius squared: the merge distance scales with the smaller one
    both = np.concatenate([data, crowd], axis=1)
    _sync(orc, g, both)
    order = (0, 1, 2, 3, 1, 0)
    Fs = [np.array(list(orc.keyframes[k].frame_T_global), np.float32) for k in order]
    before_cells = C.c_longlong()
    capi.check(lib.bahip_debug_merge_cells_batches(C.byref(before_cells)))
    before = int(orc.surfels.surfel_count)
    with g.lifecycle_batch(frames=Fs):
        _, merged = g.merge_surfels_for_keyframes(order, Fs, merge_dist_factor=orc.merge_factor)
    for k in order:
        orc.determine_supporting_surfels(k, merge=True)
    after_cells = C.c_longlong()
    capi.check(lib.bahip_debug_merge_cells_batches(C.byref(after_cells)))
    assert after_cells.value - before_cells.value == 1
    assert merged == before - int(orc.surfels.surfel_count) and merged > 1000, merged
    assert np.array_equal(_rows(g.surfel_buf.download()[:, :both.shape[1]]), _rows(orc.surfel_data[:, :both.shape[1]]))
