"""Multi-GPU glue: one process per GPU, torch.distributed ("nccl" == RCCL on ROCm) over xGMI.

Sharding: surfels are split over the ranks, keyframe images / poses / intrinsics are
replicated.  Activation and the geometry step are then purely local; the only exchange on the
alternating path is the sum of the per-keyframe pose normal equations, K x 28 floats per
Gauss-Newton round ("RCCL all-reduce of pose Hessians").  The C ABI calls back into
`bahip_allreduce_fn` with the device buffer to be summed; this module implements that hook with
torch.distributed.all_reduce on a zero-copy view of the buffer.
"""
import ctypes as C

from . import capi


class _DevicePtrView:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def shard_range(total, rank, world):
    """Contiguous surfel slice owned by `rank` (identical partition rule on every rank)."""
    return (total * rank) // world, (total * (rank + 1)) // world


def shard_chunks(total, rank, world, chunk=4096):
    """Chunk-cyclic surfel partition: rank r owns chunks r, r + world, r + 2 world, ... of `chunk` consecutive
    surfels.  Consecutive surfels are spatial neighbours (tile-major creation order), and the work per surfel varies
    over the scene with the number of keyframes that see it; dealing chunks round-robin keeps every rank's share
    of each region - and so its kernel time - the same, while a chunk (64 wavefronts) keeps its locality.
    Returns the index array (ascending) of the surfels owned by `rank`."""
    import numpy as np
    starts = np.arange(rank * chunk, total, world * chunk, dtype=np.int64)
    if starts.size == 0:
        return np.zeros(0, np.int64)
    idx = (starts[:, None] + np.arange(chunk, dtype=np.int64)[None, :]).ravel()
    return idx[idx < total]


def make_allreduce_callback(all_reduce_tensor):
    """Wraps `all_reduce_tensor(torch_tensor)` as a bahip_allreduce_fn."""
    import torch

    views = {}   # (pointer, count) -> aliasing tensor; the backend reuses a handful of buffers, wrapping costs ~25 us

    def _hook(device_ptr, count, _user):
        try:
            t = views.get((device_ptr, count))
            if t is None:
                if len(views) > 64:
                    views.clear()
                t = views[(device_ptr, count)] = torch.as_tensor(_DevicePtrView(device_ptr, count), device="cuda")
            all_reduce_tensor(t)
            return 0
        except Exception as e:  # pragma: no cover - surfaced through bahip_last_error
            print(f"[badslam_amd.multigpu] all-reduce hook failed: {e}", flush=True)
            return 1

    return capi.ALLREDUCE_FN(_hook)


def install_allreduce(ctx, dist):
    """Installs an RCCL all-reduce (SUM) as the context's reduction hook.  Returns the callback
    object, which the caller must keep alive for the lifetime of the context."""
    cb = make_allreduce_callback(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
    capi.check(ctx.lib.bahip_context_set_allreduce(ctx.handle, cb, None))
    return cb
