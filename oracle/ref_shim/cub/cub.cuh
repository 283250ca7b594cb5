// cub/cub.cuh -- HOST stand-in (TEST INFRASTRUCTURE, see oracle/oracle.h).  Of the reference's kernels that are compiled for the
// host (oracle/ref_shim/ref_kernels.cc) only MarkDeletedSurfelsCUDAKernel (B/kernel_delete_surfels.cu:137-172) uses CUB: a
// BlockReduce that counts the surfels a block deleted (and the same in B/kernel_supporting_surfels.cu), and
// CreateSurfelsForKeyframeCUDA_CountNewSurfels (B/kernel_create_surfels.cu:432-475) a device-wide inclusive sum over the new-surfel
// flags read through a converting iterator -- here a loop.  The stand-in launcher runs the threads of a block one after the other, so
// a block-wide sum cannot be formed inside the kernel: Sum(v) returns the calling thread's own value (the kernel's counter then
// receives thread 0's share only and is NOT used; ref_delete_surfels_and_update_radii counts the deletion marks itself).
// The kernels with real block collectives (pose accumulation, PCG: B/gauss_newton.cuh) are not compiled for the host.
#pragma once
#include <cstddef>

namespace cub {
enum BlockReduceAlgorithm { BLOCK_REDUCE_RAKING_COMMUTATIVE_ONLY, BLOCK_REDUCE_RAKING, BLOCK_REDUCE_WARP_REDUCTIONS };
template <typename T, int kBlockWidth, BlockReduceAlgorithm kAlgorithm = BLOCK_REDUCE_WARP_REDUCTIONS, int kBlockHeight = 1, int kBlockDepth = 1>
struct BlockReduce {
  struct TempStorage {};
  explicit BlockReduce(TempStorage&) {}
  T Sum(T value) { return value; }
};

template <typename ValueType, typename ConversionOp, typename InputIterator>
struct TransformInputIterator {
  InputIterator input;
  ConversionOp op;
  TransformInputIterator(InputIterator input_, ConversionOp op_) : input(input_), op(op_) {}
  ValueType operator[](size_t i) const { return op(input[i]); }
};

struct DeviceScan {
  // CUB's two-phase protocol: a first call with temp_storage == NULL only reports the scratch size
  template <typename InputIterator, typename OutputIterator>
  static int InclusiveSum(void* temp_storage, size_t& temp_storage_bytes, InputIterator in, OutputIterator out, int num_items, void* /*stream*/ = nullptr) {
    if (temp_storage == nullptr) { temp_storage_bytes = 1; return 0; }
    auto running = in[0];
    running = 0;
    for (int i = 0; i < num_items; ++i) { running += in[i]; out[i] = running; }
    return 0;
  }
};
}  // namespace cub
