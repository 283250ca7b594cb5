// VALU issue-rate microbenchmark (gfx950): how many cycles does one wave64 binary32 VALU instruction occupy a SIMD?
// Independent fma chains (no memory, no dependencies closer than 8 instructions), W waves per SIMD; prints wave-instructions
// per SIMD per cycle for v_fma_f32 and v_pk_fma_f32.  The roofline "VALU issue fraction" in bench.py / DESIGN.md rests on it.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int kIters>
__global__ void __launch_bounds__(64) fma_kernel(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
// one dependent chain: every instruction needs the previous result
template <int kIters>
__global__ void __launch_bounds__(64) dep_fma_kernel(float* out, float a, float b) {
  float x0 = threadIdx.x;
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                   : "+v"(x0) : "v"(a), "v"(b));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = x0;
}
template <int kIters>
__global__ void __launch_bounds__(64) pk_fma_kernel(float* out, float a, float b) {
  float2v x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  float2v va = {a, a}, vb = {b, b};
#pragma unroll 1
  for (int i = 0; i < kIters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(va), "v"(vb));
    }
  }
  const float2v s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const double clock_hz = 1e3 * p.clockRate;
  constexpr int kIters = 4096;
  float* out;
  hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%s: %d CUs, %.0f MHz\n", p.gcnArchName, cus, clock_hz / 1e6);
  for (int packed = 0; packed < 3; ++packed)
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd = (waves_per_simd < 4 ? waves_per_simd + 1 : waves_per_simd * 2)) {
      const int blocks = cus * 4 * waves_per_simd;
      float ms = 0, best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (packed == 2) hipLaunchKernelGGL(dep_fma_kernel<kIters>, dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f);
        else if (packed) hipLaunchKernelGGL(pk_fma_kernel<kIters>, dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(fma_kernel<kIters>, dim3(blocks), dim3(64), 0, 0, out, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double instr_per_simd = (double)kIters * 64 * waves_per_simd;
      const double cycles = best * 1e-3 * clock_hz;
      printf("%-13s %d waves/SIMD: %.3f ms, %.2f cycles per wave64 instruction per SIMD, %.1f TFLOP/s\n", packed == 2 ? "dependent fma" : packed ? "v_pk_fma_f32" : "v_fma_f32",
             waves_per_simd, best, cycles / instr_per_simd, instr_per_simd * cus * 4 * 64 * (packed == 1 ? 4 : 2) / (best * 1e-3) / 1e12);
    }
  return 0;
}
