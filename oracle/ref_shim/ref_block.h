// ref_block.h -- what the host stand-ins for a CUDA thread block share (TEST INFRASTRUCTURE, see oracle/oracle.h): the launcher
// (libvis/cuda/cuda_auto_tuner.h), __syncthreads_or (cuda_runtime.h) and cub::BlockReduce (cub/cub.cuh).
#pragma once

#include <vector>

struct RefDim3 { unsigned int x, y, z; };
extern thread_local RefDim3 threadIdx, blockIdx, blockDim, gridDim;
extern bool ref_launch_sequential;   // blocks one after the other, in ascending order, on the calling thread
extern bool ref_thread0_last;        // within a block: threads 1 .. n - 1, then thread 0 (kernels that use a BlockReduce result)
extern bool ref_barrier_passes;      // __syncthreads() ends a pass like a block vote (kernels that hand values from thread to thread
                                     // through __shared__ memory and have no other side effect before the barrier)

// per OpenMP thread: the collective state of the block it is running (see cuda_auto_tuner.h)
struct RefBlockState {
  std::vector<int> votes;            // results of the __syncthreads_or calls resolved so far
  int vote_call = 0;                 // how many of them the running thread has passed
  int pending_or = 0;                // predicates deposited at the first unresolved vote in this pass
  bool pending = false;
  std::vector<double> sums;          // BlockReduce slots, one per Sum call in program order
  int sum_call = 0;
  std::vector<char> slot_used;       // BlockReduce::Reduce: has a thread of this pass put a value in the slot yet
};
extern thread_local RefBlockState ref_block;
struct RefVotePending {};
int ref_syncthreads_or(int predicate);   // ref_kernels.cc
void ref_syncthreads();
