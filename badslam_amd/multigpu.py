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


def make_allreduce_callback(all_reduce_tensor):
    """Wraps `all_reduce_tensor(torch_tensor)` as a bahip_allreduce_fn."""
    import torch

    def _hook(device_ptr, count, _user):
        try:
            t = torch.as_tensor(_DevicePtrView(device_ptr, count), device="cuda")
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
