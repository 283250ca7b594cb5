// libvis/cuda/cuda_auto_tuner.h -- HOST stand-in for the reference's kernel launcher (TEST INFRASTRUCTURE, see oracle/oracle.h).
// The reference launches every kernel through CUDA_AUTO_TUNE_1D(kernel, block width, domain width, shared memory, stream,
// arguments...) (L/cuda/cuda_auto_tuner.h:447-507: kernel<<<ceil(domain / width), width, shared, stream>>>(arguments)).  Here
// the macro runs the grid on the host: blocks are dealt to OpenMP threads, the threads of a block run one after the other, and
// threadIdx / blockIdx / blockDim / gridDim are thread-local variables the kernel bodies read.  Valid for kernels without block
// collectives and without inter-thread communication -- which is what oracle/ref_shim/ref_kernels.cc compiles: every thread of
// B/kernel_opt_geometry.cu and B/kernel_surfel_activation.cu owns one surfel (__syncthreads_or inside
// SurfelProjectsToAssociatedPixel only lets a block leave early together; for one thread it is the identity, cuda_runtime.h).
#pragma once

#include <cuda_runtime.h>

struct RefDim3 { unsigned int x, y, z; };
extern thread_local RefDim3 threadIdx, blockIdx, blockDim, gridDim;
// true: blocks run one after the other, in ascending order, on the calling thread -- for kernels whose result depends on which
// thread wins an atomic (B/kernel_supporting_surfels.cu): the lowest surfel index then wins, which is the rule the oracle and
// the HIP kernels define ("FIX: lowest-index winner")
extern bool ref_launch_sequential;

// selects the instantiation of a bool-templated kernel at run time (the reference's L/cuda/cuda_util.h:53-64 has the same helper;
// that header cannot be included here: it pulls in the logging library)
#define COMPILE_OPTION(option, ...)                                                                      \
  do {                                                                                                   \
    if (option) { constexpr bool _##option = true; (void)_##option; __VA_ARGS__; }                     \
    else { constexpr bool _##option = false; (void)_##option; __VA_ARGS__; }                           \
  } while (false)

#define CHECK_CUDA_NO_ERROR() do {} while (false)   // B/cuda_util.cuh:40-45: CUDA_CHECK() expands to this

#define CUDA_AUTO_TUNE_1D(kernel_name, default_block_width, domain_width, shared_memory_size, stream, ...)                              \
  do {                                                                                                                                   \
    const long long ref_domain = (long long)(domain_width);                                                                             \
    const unsigned int ref_width = (unsigned int)(default_block_width);                                                                 \
    const long long ref_blocks = (ref_domain + ref_width - 1) / ref_width;                                                              \
    _Pragma("omp parallel for if(!ref_launch_sequential) schedule(dynamic, 4)")                                                                                    \
    for (long long ref_block = 0; ref_block < ref_blocks; ++ref_block) {                                                                \
      blockDim = RefDim3{ref_width, 1, 1};                                                                                              \
      gridDim = RefDim3{(unsigned int)ref_blocks, 1, 1};                                                                                \
      blockIdx = RefDim3{(unsigned int)ref_block, 0, 0};                                                                                \
      for (unsigned int ref_thread = 0; ref_thread < ref_width; ++ref_thread) {                                                         \
        threadIdx = RefDim3{ref_thread, 0, 0};                                                                                          \
        kernel_name(__VA_ARGS__);                                                                                                       \
      }                                                                                                                                  \
    }                                                                                                                                    \
  } while (false)

// L/cuda/cuda_auto_tuner.h:517-610: the kernel's template arguments may name the block width (`block_width`)
#define TEMPLATE_ARGUMENTS(...) __VA_ARGS__
#define CUDA_AUTO_TUNE_1D_TEMPLATED(kernel_name, default_block_width, domain_width, shared_memory_size, stream, template_parameters, ...)   \
  do {                                                                                                                                   \
    constexpr int block_width = (default_block_width);                                                                                  \
    (void)block_width;                                                                                                                  \
    const long long ref_domain = (long long)(domain_width);                                                                             \
    const unsigned int ref_width = (unsigned int)(default_block_width);                                                                 \
    const long long ref_blocks = (ref_domain + ref_width - 1) / ref_width;                                                              \
    _Pragma("omp parallel for if(!ref_launch_sequential) schedule(dynamic, 4)")                                                                                    \
    for (long long ref_block = 0; ref_block < ref_blocks; ++ref_block) {                                                                \
      blockDim = RefDim3{ref_width, 1, 1};                                                                                              \
      gridDim = RefDim3{(unsigned int)ref_blocks, 1, 1};                                                                                \
      blockIdx = RefDim3{(unsigned int)ref_block, 0, 0};                                                                                \
      for (unsigned int ref_thread = 0; ref_thread < ref_width; ++ref_thread) {                                                         \
        threadIdx = RefDim3{ref_thread, 0, 0};                                                                                          \
        kernel_name<template_parameters>(__VA_ARGS__);                                                                                  \
      }                                                                                                                                  \
    }                                                                                                                                    \
  } while (false)

// L/cuda/cuda_auto_tuner.h:288-294: a 2-D grid of default_block_width x default_block_height blocks over domain_width x
// domain_height.  Within a block the threads run row by row (threadIdx.y outer, threadIdx.x inner): a sparse cell lies inside one
// block, so under ref_launch_sequential its first pixel in row-major order is the one that wins the cell's atomicCAS.
#define CUDA_AUTO_TUNE_2D(kernel_name, default_block_width, default_block_height, domain_width, domain_height, shared_memory_size, stream, ...) \
  do {                                                                                                                                   \
    const unsigned int ref_bw = (unsigned int)(default_block_width), ref_bh = (unsigned int)(default_block_height);                     \
    const long long ref_bx = ((long long)(domain_width) + ref_bw - 1) / ref_bw, ref_by = ((long long)(domain_height) + ref_bh - 1) / ref_bh; \
    _Pragma("omp parallel for if(!ref_launch_sequential) schedule(dynamic, 1)")                                                         \
    for (long long ref_block = 0; ref_block < ref_bx * ref_by; ++ref_block) {                                                           \
      blockDim = RefDim3{ref_bw, ref_bh, 1};                                                                                            \
      gridDim = RefDim3{(unsigned int)ref_bx, (unsigned int)ref_by, 1};                                                                 \
      blockIdx = RefDim3{(unsigned int)(ref_block % ref_bx), (unsigned int)(ref_block / ref_bx), 0};                                    \
      for (unsigned int ref_ty = 0; ref_ty < ref_bh; ++ref_ty)                                                                          \
        for (unsigned int ref_tx = 0; ref_tx < ref_bw; ++ref_tx) {                                                                      \
          threadIdx = RefDim3{ref_tx, ref_ty, 0};                                                                                       \
          kernel_name(__VA_ARGS__);                                                                                                     \
        }                                                                                                                                \
    }                                                                                                                                    \
  } while (false)
