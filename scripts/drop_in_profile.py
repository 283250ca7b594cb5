#!/usr/bin/env python3
"""The drop-in call path alone, for a kernel trace: the bench scene, one absorbing call, then three calls of
vis::DirectBA::BundleAdjustment(do_surfel_updates = true, 10 iterations) -- run under `rocprofv3 --kernel-trace --stats`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    sys.argv = [sys.argv[0]]
    args = bench.parse_args()
    import torch
    torch.cuda.set_device(0)
    ba, data, poses_gt = bench.build_scene(args, lambda m: print("[drop_in]", m, file=sys.stderr))
    creation_order = bench.build_scene.creation_order
    ba.upload_surfels(creation_order)
    K = args.keyframes

    def call():
        done, _ = ba.BundleAdjustment(do_surfel_updates=True, optimize_poses=True, optimize_geometry=True, min_iterations=10, max_iterations=10,
                                      active_keyframe_window_start=0, active_keyframe_window_end=K - 1, increase_ba_iteration_count=True)
        return done

    call()
    ctx = ba.backend_context()
    ctx.synchronize()
    print("[drop_in] MARK timed calls begin", file=sys.stderr)
    t = time.perf_counter()
    n = sum(call() for _ in range(3))
    ctx.synchronize()
    dt = time.perf_counter() - t
    print(f"[drop_in] {1e3 * dt / 3:.2f} ms per call, {n / dt:.1f} it/s, {ba.surfels_size()} surfels", file=sys.stderr)


if __name__ == "__main__":
    main()
