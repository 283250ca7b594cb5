"""world_size-2 test of the multi-GPU decomposition on CPU (gloo): surfels are sharded over the
ranks, keyframes replicated; each rank builds the pose normal equations of every keyframe from its
shard as 48.16 fixed-point integers (the backend's definition of the sum), the K x 28 int64 block is
all-reduced (SUM) and every rank solves the same 6x6 systems.  The per-shard partial sums come from the
oracle here (no GPU in this container); the partition rule (badslam_amd.multigpu.shard_chunks), the
integer reduction and the EXACT equality with the unsharded result are what is under test."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from badslam_amd import multigpu
    from tests import common
    scene = common.small_scene(num_keyframes=3, seed=4, width=160, height=120)
    ba = common.build_oracle(scene, 60000)
    N = ba.surfels_size
    data = ba.surfel_data[:, :N].copy()
    mine = multigpu.shard_chunks(N, rank, world, chunk=1024)   # the partition bench.py uses (there: chunks of 4096)
    # this rank's shard becomes the whole surfel buffer of its (replicated-keyframe) scene
    ba.surfel_data[:, :mine.size] = data[:, mine]
    ba.surfels.surfels_size = mine.size
    K = len(ba.keyframes)
    Hb = np.zeros((K, 28, 2), np.int64)          # the backend's buffer: [work item][coefficient][limb]
    for k in range(K):
        Hb[k, :27] = ba.accumulate_pose_coeffs_fixed(k)
    t = torch.from_numpy(Hb)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)       # what ncclAllReduce(ncclInt64) / the hook does on the device buffer
    np.save(os.path.join(out_dir, f"hb_{rank}.npy"), t.numpy())
    np.save(os.path.join(out_dir, f"owned_{rank}.npy"), mine)
    dist.destroy_process_group()


def test_surfel_sharded_pose_normal_equations_match_unsharded(tmp_path):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from tests import common
    scene = common.small_scene(num_keyframes=3, seed=4, width=160, height=120)
    ba = common.build_oracle(scene, 60000)
    ref = np.zeros((3, 28, 2), np.int64)
    for k in range(3):
        ref[k, :27] = ba.accumulate_pose_coeffs_fixed(k)
    owned = np.concatenate([np.load(tmp_path / "owned_0.npy"), np.load(tmp_path / "owned_1.npy")])
    assert np.array_equal(np.sort(owned), np.arange(ba.surfels_size))                # complete, disjoint
    hb0, hb1 = np.load(tmp_path / "hb_0.npy"), np.load(tmp_path / "hb_1.npy")
    assert np.array_equal(hb0, hb1)                                                  # every rank holds the same sums
    # shards are whole 64-surfel tiles of the unsharded cloud (chunks of 1024) and integer addition is associative: the
    # sharded sums ARE the unsharded sums, bit for bit -- every rank takes exactly the single-GPU Gauss-Newton step
    assert np.array_equal(hb0, ref)
    assert np.abs(ref[:, :21, 1]).max() > 2 ** 14 and np.abs(ref[:, :21, 0]).max() > 2 ** 32    # real magnitudes in both limbs
    for k in range(3):   # and the binary32 H, b agree with a plain binary64 running sum to binary32 precision
        H, b, _, _ = ba.accumulate_pose_coeffs(k, accumulate_double=True)
        Hf = ba.pose_limbs_value(ref[k, :21]).astype(np.float32)
        assert np.allclose(Hf, H, rtol=0, atol=2e-7 * np.abs(H).max())
        H32, _, _, _ = ba.accumulate_pose_coeffs(k, accumulate_double=False)
        assert np.array_equal(Hf.view(np.uint32), np.asarray(H32, np.float32).view(np.uint32))


def test_shard_chunks_partition():
    from badslam_amd import multigpu
    for total in (0, 1, 4095, 4096, 4097, 100000, 3000000):
        for world in (1, 2, 3, 8):
            parts = [multigpu.shard_chunks(total, r, world) for r in range(world)]
            allidx = np.concatenate(parts)
            assert allidx.size == total and np.array_equal(np.sort(allidx), np.arange(total))    # complete and disjoint
            assert all(np.all(np.diff(p) > 0) for p in parts if p.size > 1)                         # ascending
            if total >= 4096 * world * 8:
                sizes = [p.size for p in parts]
                assert max(sizes) - min(sizes) <= 4096


def test_shard_range_partitions():
    from badslam_amd import multigpu
    for total in (0, 1, 7, 1000, 3000001):
        for world in (1, 2, 3, 8):
            edges = [multigpu.shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


# ---- sharded PCG: the exchange protocol of capi_solvers.hip (bahip_pcg_iteration) on a model problem ---------------------------------
def _pcg_model_problem(seed=3, head=13, surfels=600, rows_per_surfel=5):
    """Residuals that each touch the dense head (poses / intrinsics) and ONE surfel unknown -- the arrowhead structure of
    the BA normal equations.  Returns (J_head [R, head], J_surfel [R], surfel index [R], residual [R]), binary32."""
    rng = np.random.default_rng(seed)
    R = surfels * rows_per_surfel
    idx = np.repeat(np.arange(surfels), rows_per_surfel)
    f = np.float32
    return rng.standard_normal((R, head)).astype(f), (rng.standard_normal(R) + 2.0).astype(f), idx, rng.standard_normal(R).astype(f)


def _pcg_sharded(Jh, Js, idx, res, owned, allreduce_limbs, steps=12, eps=np.float32(1e-8)):
    """One rank's view, in the backend's arithmetic: binary32 vectors; the head of every vector is replicated, the surfel
    block local; every sum over residual rows into a head entry and every dot product is an EXACT sum of binary32 terms whose
    int64 limbs are summed over the ranks (oracle_exact.c) -- the head's own share of a dot product is identical on every rank
    and is not exchanged.  Mirrors init -> init2 -> (step1, step2, step3)* with the two exchanges per inner step."""
    from oracle import binding as ob
    f = np.float32
    H, S = Jh.shape[1], int(idx.max()) + 1
    rows = np.isin(idx, owned)
    Jh, Js, idx, res = Jh[rows], Js[rows], idx[rows], res[rows]
    local = np.zeros(S, bool); local[owned] = True

    def head_sums(terms):                                       # terms [rows, H] -> H exact sums, exchanged
        limbs = np.stack([ob.exact_limbs(terms[:, c]) for c in range(terms.shape[1])])
        limbs = allreduce_limbs(limbs)
        return np.array([ob.exact_resolve(l) for l in limbs]).astype(f)

    def surfel_sums(terms):                                     # per-surfel chains in row order (local, no exchange)
        out = np.zeros(S, f)
        for r in range(len(terms)):
            out[idx[r]] = f(out[idx[r]] + terms[r])
        return out

    def dot(ah, as_, bh, bs, extra_local_terms=()):
        mine = ob.exact_limbs((as_[local] * bs[local]).astype(f))
        for t in extra_local_terms:
            mine = ob.exact_limbs(t, mine)
        total = allreduce_limbs(mine[None])[0] + ob.exact_limbs((ah * bh).astype(f))   # head share: not exchanged
        return f(ob.exact_resolve(total))

    rh, Mh = np.split(head_sums(np.concatenate([-(Jh * res[:, None]), Jh * Jh], axis=1).astype(f)), 2)   # exchange: head of r and M
    rs, Ms = surfel_sums((-(Js * res)).astype(f)), surfel_sums((Js * Js).astype(f))
    ph, ps = (rh / (Mh + eps)).astype(f), np.where(local, rs / (Ms + eps), 0).astype(f)
    dh, ds = np.zeros(H, f), np.zeros(S, f)
    alpha_n = dot(rh, rs, ph, ps)
    for _ in range(steps):
        jp = ((Jh * ph[None, :]).sum(1, dtype=f) + Js * ps[idx]).astype(f)
        gs = surfel_sums((Js * jp).astype(f))
        gh = head_sums((Jh * jp[:, None]).astype(f))                                                      # exchange 1: head of g ...
        alpha_d = dot((eps * ph).astype(f), (eps * ps).astype(f), ph, ps, extra_local_terms=[(jp * jp).astype(f)])   # ... and alpha_d
        alpha = f(alpha_n / alpha_d) if alpha_d >= 1e-35 else f(0)
        dh, ds = (dh + alpha * ph).astype(f), (ds + alpha * ps).astype(f)
        rh, rs = (rh - alpha * (gh + eps * ph)).astype(f), (rs - alpha * (gs + eps * ps)).astype(f)
        zh, zs = (rh / (Mh + eps)).astype(f), np.where(local, rs / (Ms + eps), 0).astype(f)
        beta_n = dot(zh, zs, rh, rs)                                                                      # exchange 2: beta_n
        beta = f(beta_n / alpha_n) if alpha_n >= 1e-35 else f(0)
        ph, ps = (zh + beta * ph).astype(f), (zs + beta * ps).astype(f)
        alpha_n = beta_n
    return dh, ds


def _pcg_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from badslam_amd import multigpu
    Jh, Js, idx, res = _pcg_model_problem()
    owned = multigpu.shard_chunks(int(idx.max()) + 1, rank, world, chunk=64)

    def allreduce_limbs(limbs):
        t = torch.from_numpy(np.ascontiguousarray(limbs, np.int64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)      # BAHIP_SUM_I64
        return t.numpy()

    dh, ds = _pcg_sharded(Jh, Js, idx, res, owned, allreduce_limbs)
    np.save(os.path.join(out_dir, f"pcg_head_{rank}.npy"), dh)
    np.save(os.path.join(out_dir, f"pcg_surfels_{rank}.npy"), ds)
    np.save(os.path.join(out_dir, f"pcg_owned_{rank}.npy"), owned)
    dist.destroy_process_group()


def test_sharded_pcg_exchange_protocol(tmp_path):
    """Dense head replicated, its sums exchanged as the int64 limbs of exact accumulators; surfel block local; dot products =
    exchanged local limbs + the head's own limbs: two ranks following the protocol produce, bit for bit, what one rank
    produces -- and that is a solution of the normal equations to binary32 accuracy."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_pcg_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    Jh, Js, idx, res = _pcg_model_problem()
    S = int(idx.max()) + 1
    J = np.zeros((len(res), Jh.shape[1] + S))
    J[:, :Jh.shape[1]] = Jh
    J[np.arange(len(res)), Jh.shape[1] + idx] = Js
    solution = np.linalg.lstsq(J, -res.astype(np.float64), rcond=None)[0]
    single_h, single_s = _pcg_sharded(Jh, Js, idx, res, np.arange(S), lambda limbs: limbs)     # the same code on one rank
    assert np.abs(np.concatenate([single_h, single_s]) - solution).max() < 2e-4
    h0, h1 = np.load(tmp_path / "pcg_head_0.npy"), np.load(tmp_path / "pcg_head_1.npy")
    assert np.array_equal(h0.view(np.uint32), h1.view(np.uint32)) and np.array_equal(h0.view(np.uint32), single_h.view(np.uint32))
    surfels = np.zeros(S, np.float32)
    for r in range(world):
        owned = np.load(tmp_path / f"pcg_owned_{r}.npy")
        surfels[owned] = np.load(tmp_path / f"pcg_surfels_{r}.npy")[owned]
    assert np.array_equal(surfels.view(np.uint32), single_s.view(np.uint32))


# ---- the intrinsics step's exchange: binary64 accumulators (BAHIP_SUM_F64) ---------------------------------------------------
def _intrinsics_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from badslam_amd import multigpu
    from tests import common
    scene = common.small_scene(num_keyframes=3, seed=6, width=160, height=120)
    ba = common.build_oracle(scene, 60000)
    ba.depth_cam.fx += 0.5; ba.depth_cam.cy -= 0.7; ba.color_cam.fy += 0.3      # something to correct
    N = ba.surfels_size
    data = ba.surfel_data[:, :N].copy()
    mine = multigpu.shard_chunks(N, rank, world, chunk=1024)
    ba.surfel_data[:, :mine.size] = data[:, mine]
    ba.surfels.surfels_size = mine.size
    glob, cells = ba.intrinsics_accumulators()
    t = torch.from_numpy(np.concatenate([glob, cells.ravel()]))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)       # ncclAllReduce(ncclDouble) / the hook with BAHIP_SUM_F64
    np.save(os.path.join(out_dir, f"intr_{rank}.npy"), t.numpy())
    dist.destroy_process_group()


def test_surfel_sharded_intrinsics_accumulators_round_to_the_unsharded_ones(tmp_path):
    """The accumulators of the intrinsics step are binary64 sums of binary32 terms (per-tile totals, per-pair cell terms).
    Summed shard by shard and exchanged, they differ from the unsharded sums in the last bits of the binary64 value at most,
    so their binary32 roundings -- what the Schur complement works on -- are those of the unsharded run."""
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_intrinsics_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from tests import common
    scene = common.small_scene(num_keyframes=3, seed=6, width=160, height=120)
    ba = common.build_oracle(scene, 60000)
    ba.depth_cam.fx += 0.5; ba.depth_cam.cy -= 0.7; ba.color_cam.fy += 0.3
    glob, cells = ba.intrinsics_accumulators()
    ref = np.concatenate([glob, cells.ravel()])
    got0, got1 = np.load(tmp_path / "intr_0.npy"), np.load(tmp_path / "intr_1.npy")
    assert np.array_equal(got0, got1)
    # (entries of the `a` row are zero while cfactor is, and the colour sums may be empty: zero residuals are skipped)
    assert np.count_nonzero(ref) > 1000 and np.count_nonzero(ref[:20]) >= 12
    assert np.array_equal(got0.astype(np.float32).view(np.uint32), ref.astype(np.float32).view(np.uint32))
    # and the binary64 values themselves agree to a few units in the last place
    assert np.abs(got0 - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.all(np.abs(got0 - ref) <= 8 * np.spacing(np.abs(ref)) + 0)
    assert np.array_equal(got0.reshape(-1)[34:].reshape(-1, 8)[:, 7], ref[34:].reshape(-1, 8)[:, 7])   # observation counts: exact


# ---- keyframe sharding (bahip_context_set_keyframe_sharding) -----------------------------------------------------------------
ACCUM0 = 8      # first accumulator row of the surfel buffer (B/kernels.cuh:69-93); the oracle parks its sums there


def _keyframe_shard_worker(rank, world, port, out_dir, classes=4, num_keyframes=6):
    """Rank `rank` holds all surfels and the images of the keyframes k with k % world == rank (whole keyframe classes k % classes
    of the per-surfel sums: classes = 4, or 8 for eight ranks).  Normals pass: the
    partial sums of each keyframe class it owns (the oracle run with only that class's keyframes not inactive leaves
    ((0 + p_c) ...) = p_c in the accumulator rows); pose phase: the fixed-point normal equations of its keyframes, zero rows
    for the others.  Both travel as int64 sums -- the class partials as BIT PATTERNS, two binary32 values per word."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as ob
    from tests import common
    scene = common.small_scene(num_keyframes=num_keyframes, seed=8, width=160, height=120)
    ba = common.build_oracle(scene, 60000)
    ob.lib().orc_set_sum_classes(classes)
    N, K = ba.surfels_size, len(ba.keyframes)
    ba.active[:N] = 1
    stride = (N + 63) & ~63
    partials = np.zeros((classes, 4, stride), np.float32)       # [class][x, y, z, count][surfel], zero where another rank computes
    normals_before = ba.surfel_data[3, :N].copy()
    for cls in range(classes):
        if cls % world != rank:
            continue
        for k in range(K):
            ba.keyframes[k].activation = ob.KF_ACTIVE if k % classes == cls else ob.KF_INACTIVE
        ba.surfel_data[3, :N] = normals_before                  # every class is evaluated at the same (old) normals
        ba.update_surfel_normals()
        partials[cls, :, :N] = ba.surfel_data[ACCUM0:ACCUM0 + 4, :N]
    ba.surfel_data[3, :N] = normals_before
    words = torch.from_numpy(partials.reshape(-1).view(np.int64))
    dist.all_reduce(words, op=dist.ReduceOp.SUM)                # BAHIP_SUM_I64 over classes * 4 * stride / 2 words
    np.save(os.path.join(out_dir, f"partials_{rank}.npy"), words.numpy().view(np.float32).reshape(classes, 4, stride)[:, :, :N])

    Hb = np.zeros((K, 28, 2), np.int64)
    for k in range(K):
        if k % world == rank:
            Hb[k, :27] = ba.accumulate_pose_coeffs_fixed(k)
    t = torch.from_numpy(Hb)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    np.save(os.path.join(out_dir, f"kf_hb_{rank}.npy"), t.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,classes,num_keyframes", [(2, 4, 6), (8, 8, 11)])
def test_keyframe_sharded_class_partials_and_pose_equations_match_unsharded(tmp_path, world, classes, num_keyframes):
    """(8, 8, 11): BASELINE configs[3]'s rank count -- eight ranks, one keyframe class of the 8-class definition each."""
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_keyframe_shard_worker, args=(world, port, str(tmp_path), classes, num_keyframes), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import binding as ob
    from tests import common
    scene = common.small_scene(num_keyframes=num_keyframes, seed=8, width=160, height=120)
    ba = common.build_oracle(scene, 60000)
    ob.lib().orc_set_sum_classes(classes)
    N, K = ba.surfels_size, len(ba.keyframes)
    ba.active[:N] = 1
    want = np.zeros((K, 28, 2), np.int64)                        # (before the normals pass below changes the normals)
    for k in range(K):
        want[k, :27] = ba.accumulate_pose_coeffs_fixed(k)
    try:
        ba.update_surfel_normals()                               # unsharded: ((p0 + p1) + p2) + p3 (+ ... + p7) over all keyframes
    finally:
        ob.lib().orc_set_sum_classes(4)
    ref = ba.surfel_data[ACCUM0:ACCUM0 + 4, :N].copy()
    parts = [np.load(tmp_path / f"partials_{r}.npy") for r in range(world)]
    p0 = parts[0]
    assert all(np.array_equal(p0.view(np.uint32), p.view(np.uint32)) for p in parts)  # every rank holds every class's partials
    assert all(np.count_nonzero(p0[c, 3]) > N // 10 for c in range(classes))          # every class saw surfels
    total = p0[0] + p0[1]                                                             # the defined combination, binary32, ascending
    for c in range(2, classes):
        total = total + p0[c]
    assert total.dtype == np.float32
    assert np.array_equal(total.view(np.uint32), ref.view(np.uint32))                 # the unsharded sums, bit for bit

    hs = [np.load(tmp_path / f"kf_hb_{r}.npy") for r in range(world)]
    assert all(np.array_equal(hs[0], h) for h in hs) and np.array_equal(hs[0], want)  # the "all-reduce of pose Hessians"
