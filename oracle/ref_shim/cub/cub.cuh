// cub/cub.cuh -- HOST stand-in (TEST INFRASTRUCTURE, see oracle/oracle.h).  Of the reference's kernels that are compiled for the
// host (oracle/ref_shim/ref_kernels.cc) three use CUB.  BlockReduce(...).Sum -- B/gauss_newton.cuh:46-93 (the 27 sums per residual
// of the pose normal equations; thread 0 adds each block total to H / b) and the deletion counters of
// B/kernel_delete_surfels.cu:137-172 / B/kernel_supporting_surfels.cu:87-94: the stand-in launcher runs the threads of a block one
// after the other, the k-th Sum call of every thread adds its value to slot k of the block (binary64) and returns the slot's
// running total -- the block's total for the thread that runs last, which the launcher makes thread 0 where the result matters
// (ref_thread0_last; the deletion counters are not used: the callers recount the marks).  And
// CreateSurfelsForKeyframeCUDA_CountNewSurfels (B/kernel_create_surfels.cu:432-475): a device-wide inclusive sum over the new-surfel
// flags read through a converting iterator -- here a loop.
#pragma once
#include <cstddef>

#include <ref_block.h>

namespace cub {
enum BlockReduceAlgorithm { BLOCK_REDUCE_RAKING_COMMUTATIVE_ONLY, BLOCK_REDUCE_RAKING, BLOCK_REDUCE_WARP_REDUCTIONS };
template <typename T, int kBlockWidth, BlockReduceAlgorithm kAlgorithm = BLOCK_REDUCE_WARP_REDUCTIONS, int kBlockHeight = 1, int kBlockDepth = 1>
struct BlockReduce {
  struct TempStorage {};
  explicit BlockReduce(TempStorage&) {}
  T Sum(T value) {
    RefBlockState& state = ref_block;
    const int call = state.sum_call++;
    if (call >= (int)state.sums.size()) state.sums.resize(call + 1, 0.0);
    state.sums[call] += (double)value;
    return (T)state.sums[call];
  }
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
