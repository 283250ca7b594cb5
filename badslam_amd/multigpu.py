"""Multi-GPU glue: one process per GPU, torch.distributed ("nccl" == RCCL on ROCm) over xGMI.

Sharding: surfels are split over the ranks, keyframe images / poses / intrinsics are
replicated.  Activation and the geometry step are then purely local; the only exchange on the
alternating path is the sum of the per-keyframe pose normal equations, K x 28 fixed-point int64 per
Gauss-Newton round ("RCCL all-reduce of pose Hessians").

Two transports (include/badslam_hip.h):
  init_rccl(ctx, dist)          the native path: the backend owns an RCCL communicator and issues ncclAllReduce on its
                                own stream; torch.distributed only carries the 128-byte communicator id to the ranks.
  install_allreduce(ctx, dist)  the hook path: the C ABI calls back with (buffer, count, dtype, stream) and this module
                                runs torch.distributed.all_reduce on a zero-copy view of the buffer, on that stream.
"""
import ctypes as C

from . import capi


class _DevicePtrView:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr, count, dtype=0):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": {capi.SUM_I64: "<i8", capi.SUM_F64: "<f8"}.get(dtype, "<f4"),
                                         "data": (int(ptr), False), "version": 2}


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

    views = {}     # (pointer, count, dtype) -> aliasing tensor; the backend reuses a handful of buffers, wrapping costs ~25 us
    streams = {}   # hipStream_t -> torch.cuda.ExternalStream

    def _hook(device_ptr, count, dtype, hip_stream, _user):
        try:
            t = views.get((device_ptr, count, dtype))
            if t is None:
                if len(views) > 64:
                    views.clear()
                t = views[(device_ptr, count, dtype)] = torch.as_tensor(_DevicePtrView(device_ptr, count, dtype), device="cuda")
            # the reduction is queued on the stream the backend works on (NULL = the legacy default stream, which is what
            # torch's default stream is): ordered after the kernels that produced the buffer and before those that read it
            if hip_stream:
                st = streams.get(hip_stream)
                if st is None:
                    st = streams[hip_stream] = torch.cuda.ExternalStream(hip_stream)
                with torch.cuda.stream(st):
                    all_reduce_tensor(t)
            else:
                with torch.cuda.stream(torch.cuda.default_stream()):
                    all_reduce_tensor(t)
            return 0
        except Exception as e:  # pragma: no cover - surfaced through bahip_last_error
            print(f"[badslam_amd.multigpu] all-reduce hook failed: {e}", flush=True)
            return 1

    return capi.ALLREDUCE_FN(_hook)


def init_rccl(ctx, dist=None, rank=0, world=1):
    """Native path: gives the backend context its own RCCL communicator (ncclAllReduce on the context's stream, no host
    round trip).  Rank 0 creates the communicator id, torch.distributed (any backend) broadcasts its 128 bytes."""
    import torch
    if dist is not None:
        rank, world = dist.get_rank(), dist.get_world_size()
    uid = C.create_string_buffer(capi.RCCL_UNIQUE_ID_BYTES)
    if rank == 0:
        capi.check(ctx.lib.bahip_rccl_get_unique_id(uid))
    if dist is not None and world > 1:
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor(list(uid.raw), dtype=torch.uint8, device="cuda" if on_gpu else "cpu")
        dist.broadcast(t, src=0)
        uid = C.create_string_buffer(bytes(t.cpu().tolist()), capi.RCCL_UNIQUE_ID_BYTES)
    capi.check(ctx.lib.bahip_context_init_rccl(ctx.handle, uid, int(rank), int(world)))


def install_allreduce(ctx, dist):
    """Installs a torch.distributed all-reduce (SUM) as the context's reduction hook.  Returns the callback
    object, which the caller must keep alive for the lifetime of the context."""
    cb = make_allreduce_callback(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
    capi.check(ctx.lib.bahip_context_set_allreduce(ctx.handle, cb, None))
    return cb
