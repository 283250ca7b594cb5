// libvis/cuda/cuda_auto_tuner.h -- HOST stand-in for the reference's kernel launcher (TEST INFRASTRUCTURE, see oracle/oracle.h).
// The reference launches every kernel through CUDA_AUTO_TUNE_1D / _1D_TEMPLATED / _2D (L/cuda/cuda_auto_tuner.h:288-610:
// kernel<<<ceil(domain / block), block, shared, stream>>>(arguments)).  Here the macros run the grid on the host
// (ref_run_grid): blocks are dealt to OpenMP threads, the threads of a block run one after the other, and threadIdx / blockIdx /
// blockDim / gridDim are thread-local variables the kernel bodies read.  What a block's threads do TOGETHER is modelled just
// far enough for the kernels oracle/ref_shim/ref_kernels.cc and ref_preprocess.cc compile:
//   * __syncthreads_or (B/surfel_projection_nvcc_only.cuh:394,409: "leave together if nobody in the block is visible"): the
//     block is run in passes.  In a pass every thread runs from the start; at the first vote that is not resolved yet it
//     deposits its predicate and stops (RefVotePending); after the pass the vote's result is the OR of the deposits, and the
//     next pass replays it.  Everything before a vote is free of side effects in these kernels, so re-running it is harmless.
//   * cub::BlockReduce(...).Sum (B/gauss_newton.cuh:46-93: 27 sums per residual, thread 0 adds each to H / b): the k-th Sum call
//     of every thread of a block goes to slot k (binary64), and with ref_thread0_last set thread 0 runs after the others, so the
//     value it gets back -- the only one the kernels use -- is the block's total.
//   * __syncthreads() is nothing by default -- in the kernels above it only separates reuses of the reduction scratch.  One
//     kernel (B/kernel_opt_intrinsics.cu:374-424) hands values from thread to thread through __shared__ memory across it; for that
//     launch ref_barrier_passes makes the barrier end a pass exactly like a vote (nothing but __shared__ stores precede it).
//     __shared__ variables are static thread_local here: one copy per OpenMP thread, i.e. per block being run.
//   * cub::BlockReduce(...).Reduce(value, cub::Min() / cub::Max()) (B/cuda_depth_processing.cu:406-418) keeps a running result per slot
//     the same way; thread 0, run last, gets the result over the block.
//   * kernels whose result depends on which thread wins an atomicCAS run with ref_launch_sequential: blocks and threads in
//     ascending order on the calling thread, so the lowest index wins -- the rule oracle and HIP kernels define.
#pragma once

#include <cuda_runtime.h>

#include <ref_block.h>

// selects the instantiation of a bool-templated kernel at run time (the reference's L/cuda/cuda_util.h:53-75 has the same helpers;
// that header cannot be included here: it pulls in the logging library)
#define COMPILE_OPTION(option, ...)                                                                      \
  do {                                                                                                   \
    if (option) { constexpr bool _##option = true; (void)_##option; __VA_ARGS__; }                     \
    else { constexpr bool _##option = false; (void)_##option; __VA_ARGS__; }                           \
  } while (false)
#define COMPILE_OPTION_2(a, b, ...) do { COMPILE_OPTION(a, COMPILE_OPTION(b, __VA_ARGS__);); } while (false)
#define COMPILE_OPTION_3(a, b, c, ...) do { COMPILE_OPTION(a, COMPILE_OPTION(b, COMPILE_OPTION(c, __VA_ARGS__););); } while (false)
#define COMPILE_OPTION_4(a, b, c, d, ...) do { COMPILE_OPTION(a, COMPILE_OPTION(b, COMPILE_OPTION(c, COMPILE_OPTION(d, __VA_ARGS__);););); } while (false)

#define CHECK_CUDA_NO_ERROR() do {} while (false)   // B/cuda_util.cuh:40-45: CUDA_CHECK() expands to this

// Runs `body` (one kernel thread) over a grid of blocks_x x blocks_y blocks of width x height threads.  Within a block the threads
// run row by row (threadIdx.y outer, threadIdx.x inner): a sparse cell lies inside one 32 x 32 block, so under
// ref_launch_sequential its first pixel in row-major order is the one that wins the cell's atomicCAS.
template <typename Body>
void ref_run_grid(long long blocks_x, long long blocks_y, unsigned int width, unsigned int height, Body body) {
  const bool sequential = ref_launch_sequential, thread0_last = ref_thread0_last;
#pragma omp parallel for if (!sequential) schedule(dynamic, 4)
  for (long long block = 0; block < blocks_x * blocks_y; ++block) {
    blockDim = RefDim3{width, height, 1};
    gridDim = RefDim3{(unsigned int)blocks_x, (unsigned int)blocks_y, 1};
    blockIdx = RefDim3{(unsigned int)(block % blocks_x), (unsigned int)(block / blocks_x), 0};
    RefBlockState& state = ref_block;
    state.votes.clear();
    const unsigned int threads = width * height;
    for (;;) {   // one pass per vote the block meets, plus the pass that completes
      state.pending = false;
      state.pending_or = 0;
      state.sums.assign(state.sums.size(), 0.0);
      state.slot_used.assign(state.slot_used.size(), 0);
      for (unsigned int step = 0; step < threads; ++step) {
        const unsigned int t = thread0_last ? (step + 1 == threads ? 0u : step + 1u) : step;
        threadIdx = RefDim3{t % width, t / width, 0};
        state.vote_call = 0;
        state.sum_call = 0;
        try { body(); } catch (const RefVotePending&) {}
      }
      if (!state.pending) break;
      state.votes.push_back(state.pending_or);
    }
  }
}

#define CUDA_AUTO_TUNE_1D(kernel_name, default_block_width, domain_width, shared_memory_size, stream, ...)                               \
  ref_run_grid(((long long)(domain_width) + (default_block_width) - 1) / (default_block_width), 1, (default_block_width), 1,              \
               [&]() { kernel_name(__VA_ARGS__); })

// L/cuda/cuda_auto_tuner.h:517-610: the kernel's template arguments may name the block size (`block_width`, `block_height`)
#define TEMPLATE_ARGUMENTS(...) __VA_ARGS__
#define CUDA_AUTO_TUNE_1D_TEMPLATED(kernel_name, default_block_width, domain_width, shared_memory_size, stream, template_parameters, ...) \
  do {                                                                                                                                   \
    constexpr int block_width = (default_block_width);                                                                                  \
    (void)block_width;                                                                                                                  \
    ref_run_grid(((long long)(domain_width) + block_width - 1) / block_width, 1, block_width, 1,                                          \
                 [&]() { kernel_name<template_parameters>(__VA_ARGS__); });                                                              \
  } while (false)

#define CUDA_AUTO_TUNE_2D(kernel_name, default_block_width, default_block_height, domain_width, domain_height, shared_memory_size, stream, ...) \
  ref_run_grid(((long long)(domain_width) + (default_block_width) - 1) / (default_block_width),                                          \
               ((long long)(domain_height) + (default_block_height) - 1) / (default_block_height), (default_block_width),                 \
               (default_block_height), [&]() { kernel_name(__VA_ARGS__); })

#define CUDA_AUTO_TUNE_2D_TEMPLATED(kernel_name, default_block_width, default_block_height, domain_width, domain_height, shared_memory_size, stream, \
                                    template_parameters, ...)                                                                            \
  do {                                                                                                                                   \
    constexpr int block_width = (default_block_width), block_height = (default_block_height);                                           \
    (void)block_width; (void)block_height;                                                                                              \
    ref_run_grid(((long long)(domain_width) + block_width - 1) / block_width, ((long long)(domain_height) + block_height - 1) / block_height, \
                 block_width, block_height, [&]() { kernel_name<template_parameters>(__VA_ARGS__); });                                   \
  } while (false)

// L/cuda/cuda_auto_tuner.h:309-345: blocks overlap by the border, the grid covers the domain with (block - border) per block
#define CUDA_AUTO_TUNE_2D_BORDER_TEMPLATED(kernel_name, default_block_width, default_block_height, border_width, border_height, domain_width,    \
                                           domain_height, shared_memory_size, stream, template_parameters, ...)                              \
  do {                                                                                                                                   \
    constexpr int block_width = (default_block_width), block_height = (default_block_height);                                           \
    ref_run_grid(((long long)(domain_width) + (block_width - (border_width)) - 1) / (block_width - (border_width)),                     \
                 ((long long)(domain_height) + (block_height - (border_height)) - 1) / (block_height - (border_height)), block_width,   \
                 block_height, [&]() { kernel_name<template_parameters>(__VA_ARGS__); });                                                \
  } while (false)
