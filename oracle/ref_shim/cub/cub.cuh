// cub/cub.cuh -- HOST stand-in (TEST INFRASTRUCTURE, see oracle/oracle.h).  Of the reference's kernels that are compiled for the
// host (oracle/ref_shim/ref_kernels.cc) three use CUB (and two of ref_preprocess.cc: the block minimum / maximum of the depth range and the scans of the compaction).  BlockReduce(...).Sum -- B/gauss_newton.cuh:46-93 (the 27 sums per residual
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
  // Reduce(value, cub::Min() / cub::Max()) -- B/cuda_depth_processing.cu:406-418 (a 32 x 32 block): the slot holds the running
  // result over the threads that have run; the thread that runs last gets the block's result (thread 0 under ref_thread0_last --
  // the only thread whose return value CUB defines).  The kernel lets every thread with threadIdx.x == 0 use its value; the
  // others get the result over a subset of the block's real values here, which an atomicMin / atomicMax absorbs.
  template <typename Op>
  T Reduce(T value, Op op) {
    RefBlockState& state = ref_block;
    const int call = state.sum_call++;
    if (call >= (int)state.sums.size()) state.sums.resize(call + 1, 0.0);
    if (call >= (int)state.slot_used.size()) state.slot_used.resize(call + 1, 0);
    state.sums[call] = state.slot_used[call] ? (double)op((T)state.sums[call], value) : (double)value;
    state.slot_used[call] = 1;
    return (T)state.sums[call];
  }
};
struct Min { template <typename T> T operator()(const T& a, const T& b) const { return b < a ? b : a; } };
struct Max { template <typename T> T operator()(const T& a, const T& b) const { return a < b ? b : a; } };

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
  // B/kernel_compact_surfels.cu:200-246: plain pointers and the file's own reversing iterators (operator[] only)
  template <typename InputIterator, typename OutputIterator>
  static int ExclusiveSum(void* temp_storage, size_t& temp_storage_bytes, InputIterator in, OutputIterator out, int num_items, void* /*stream*/ = nullptr) {
    if (temp_storage == nullptr) { temp_storage_bytes = 1; return 0; }
    unsigned int running = 0;
    for (int i = 0; i < num_items; ++i) { const unsigned int value = in[i]; out[i] = running; running += value; }
    return 0;
  }
};
}  // namespace cub
