// cub/cub.cuh -- HOST stand-in (TEST INFRASTRUCTURE, see oracle/oracle.h).  Of the reference's kernels that are compiled for the
// host (oracle/ref_shim/ref_kernels.cc) only MarkDeletedSurfelsCUDAKernel (B/kernel_delete_surfels.cu:137-172) uses CUB: a
// BlockReduce that counts the surfels a block deleted.  The stand-in launcher runs the threads of a block one after the other, so
// a block-wide sum cannot be formed inside the kernel: Sum(v) returns the calling thread's own value (the kernel's counter then
// receives thread 0's share only and is NOT used; ref_delete_surfels_and_update_radii counts the deletion marks itself).
// The kernels with real block collectives (pose accumulation, PCG: B/gauss_newton.cuh) are not compiled for the host.
#pragma once

namespace cub {
enum BlockReduceAlgorithm { BLOCK_REDUCE_RAKING_COMMUTATIVE_ONLY, BLOCK_REDUCE_RAKING, BLOCK_REDUCE_WARP_REDUCTIONS };
template <typename T, int kBlockWidth, BlockReduceAlgorithm kAlgorithm = BLOCK_REDUCE_WARP_REDUCTIONS, int kBlockHeight = 1, int kBlockDepth = 1>
struct BlockReduce {
  struct TempStorage {};
  explicit BlockReduce(TempStorage&) {}
  T Sum(T value) { return value; }
};
}  // namespace cub
