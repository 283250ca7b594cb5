// launch_gap.hip -- what separates two dependent launches on one stream on gfx950, by the shape of the launched kernel
// (round 6: the device-driven BA loop shows 5.8 us before its small kernels and 10.4 us before the pose sweep / solve, the lifecycle
// chain 0.1 us between its kernels: profiles/r6_shard_experiments.txt).  Every variant is launched 200 times back to back; run under
// `rocprofv3 --kernel-trace` and feed the trace to scripts/microbench/launch_gap.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct BigArgs { float v[96]; };   // ~384 bytes of kernel arguments, like Intrinsics + KfEntry

__global__ void tiny_kernel(int* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1; }
__global__ void tiny_many_kernel(int* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1; }
__global__ void bigargs_kernel(BigArgs a, BigArgs b, int* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += (int)(a.v[3] + b.v[5]); }
__global__ void __launch_bounds__(1024) wg1024_kernel(int* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1; }
extern __shared__ char dyn_lds[];
__global__ void __launch_bounds__(1024) lds_kernel(int* out) { if (threadIdx.x == 0) dyn_lds[0] = 1; __syncthreads(); if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += dyn_lds[0]; }
__global__ void scratch_kernel(int* out, int n) {   // a private array indexed dynamically: scratch memory
  volatile int a[64];
  for (int i = 0; i < 64; ++i) a[i] = i * n;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += a[n & 63];
}
__global__ void writer_kernel(float* buf, size_t n) {   // leaves n * 4 bytes of dirty lines behind
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = (float)i;
}
__global__ void after_writer_kernel(int* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1; }
__global__ void hostmem_kernel(int* out, volatile int* mapped) { if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] += 1; mapped[0] = out[0]; } }

int main() {
  int* out; CHECK(hipMalloc(&out, 64)); CHECK(hipMemset(out, 0, 64));
  float* buf; const size_t n = 16u << 20; CHECK(hipMalloc(&buf, n * 4));
  int* mapped; CHECK(hipHostMalloc(&mapped, 64, hipHostMallocMapped | hipHostMallocCoherent));
  hipStream_t st; CHECK(hipStreamCreate(&st));
  CHECK(hipFuncSetAttribute((const void*)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  BigArgs a = {}, b = {};
  const int N = 200;
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, st, out);
  CHECK(hipStreamSynchronize(st));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(bigargs_kernel, dim3(256), dim3(256), 0, st, a, b, out);
  CHECK(hipStreamSynchronize(st));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(wg1024_kernel, dim3(256), dim3(1024), 0, st, out);
  CHECK(hipStreamSynchronize(st));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(1024), 96 * 1024, st, out);
  CHECK(hipStreamSynchronize(st));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(scratch_kernel, dim3(256), dim3(256), 0, st, out, i);
  CHECK(hipStreamSynchronize(st));
  for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(writer_kernel, dim3(4096), dim3(256), 0, st, buf, n); hipLaunchKernelGGL(after_writer_kernel, dim3(256), dim3(256), 0, st, out); }
  CHECK(hipStreamSynchronize(st));
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(hostmem_kernel, dim3(256), dim3(256), 0, st, out, (volatile int*)mapped);
  CHECK(hipStreamSynchronize(st));
  // a grid of many workgroups that all return at once (the in-vain launches of the loop have 256 x 1024 threads)
  for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny_many_kernel, dim3(23440), dim3(256), 0, st, out);
  CHECK(hipStreamSynchronize(st));
  printf("done %d\n", N);
  return 0;
}
