"""world_size-2 test of the multi-GPU decomposition on CPU (gloo): surfels are sharded over the
ranks, keyframes replicated; each rank builds the pose normal equations of every keyframe from its
shard, the K x 28 block is all-reduced (SUM) and every rank solves the same 6x6 systems.  The
per-shard partial sums come from the oracle here (no GPU in this container); the partition rule
(badslam_amd.multigpu.shard_chunks), the reduction and the equality with the unsharded result are
what is under test."""
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
    Hb = np.zeros((K, 28), np.float32)
    for k in range(K):
        H, b, _, _ = ba.accumulate_pose_coeffs(k, accumulate_double=True)
        Hb[k, :21], Hb[k, 21:27] = H, b
    t = torch.from_numpy(Hb)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)       # what the hook does on the device buffer over RCCL
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
    ref = np.zeros((3, 28), np.float32)
    for k in range(3):
        H, b, _, _ = ba.accumulate_pose_coeffs(k, accumulate_double=True)
        ref[k, :21], ref[k, 21:27] = H, b
    owned = np.concatenate([np.load(tmp_path / "owned_0.npy"), np.load(tmp_path / "owned_1.npy")])
    assert np.array_equal(np.sort(owned), np.arange(ba.surfels_size))                # complete, disjoint
    hb0, hb1 = np.load(tmp_path / "hb_0.npy"), np.load(tmp_path / "hb_1.npy")
    assert np.array_equal(hb0, hb1)                                                  # every rank holds the same sums
    assert np.allclose(hb0, ref, rtol=0, atol=3e-6 * np.abs(ref).max())
    for k in range(3):   # and therefore takes the same Gauss-Newton step
        M = np.zeros((6, 6)); M[np.triu_indices(6)] = ref[k, :21]; M = M + np.triu(M, 1).T
        M2 = np.zeros((6, 6)); M2[np.triu_indices(6)] = hb0[k, :21]; M2 = M2 + np.triu(M2, 1).T
        assert np.abs(np.linalg.solve(M, ref[k, 21:27]) - np.linalg.solve(M2, hb0[k, 21:27])).max() < 1e-7


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
