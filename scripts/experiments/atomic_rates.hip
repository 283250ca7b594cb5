// atomic_rates.hip -- experiment, not part of the product: how fast does gfx950 add 8-value records into a table of
// ~77 k records (the sparse-cell accumulators of the intrinsics step) with the atomic forms available?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_rates scripts/experiments/atomic_rates.hip && /tmp/atomic_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15; }

// mode: 0 f64 shared | 1 f64 per-XCC | 2 f32 shared | 3 u64 shared | 4 f64 workgroup scope per-XCC | 5 f64 sc1 shared | 6 f32 per-XCC
// | 7 u64 per-XCC | 8 f64 one lane per record (8 instructions, 64 lines each)
template <int kMode>
__global__ void __launch_bounds__(256) add_records(double* table, uint32_t records, int rounds, int coherent, size_t copy_stride) {
  const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  constexpr bool kPrivate = kMode == 1 || kMode == 4 || kMode == 6 || kMode == 7;
  double* base = table + (kPrivate ? copy_stride * xcc_id() : 0);
  for (int it = 0; it < rounds; ++it) {
    const uint32_t seed = hash32(wave * 7919u + it);
    uint32_t rec, val;
    if (kMode == 8) { rec = coherent ? (seed + 3 * lane) % records : hash32(seed + lane) % records; val = 0; }
    else { const uint32_t g = lane >> 3; rec = coherent ? (seed + 3 * g) % records : hash32(seed + g) % records; val = lane & 7; }
    if (kMode == 8) {
#pragma unroll
      for (int c = 0; c < 8; ++c) unsafeAtomicAdd(base + (size_t)rec * 8 + c, 1.0);
    } else if (kMode == 0 || kMode == 1) {
      unsafeAtomicAdd(base + (size_t)rec * 8 + val, 1.0);
    } else if (kMode == 2 || kMode == 6) {
      unsafeAtomicAdd(reinterpret_cast<float*>(base) + (size_t)rec * 8 + val, 1.0f);
    } else if (kMode == 3 || kMode == 7) {
      atomicAdd(reinterpret_cast<unsigned long long*>(base) + (size_t)rec * 8 + val, 1ull);
    } else if (kMode == 4) {
      __hip_atomic_fetch_add(base + (size_t)rec * 8 + val, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (kMode == 5) {
      double* p = base + (size_t)rec * 8 + val; const double one = 1.0;
      asm volatile("global_atomic_add_f64 %0, %1, off sc1" ::"v"(p), "v"(one) : "memory");
    }
  }
}

template <int kMode> float run(double* table, uint32_t records, int blocks, int rounds, int coherent, size_t copy_stride, size_t bytes, double* total_out) {
  TRY(hipMemset(table, 0, bytes));
  hipEvent_t a, b; TRY(hipEventCreate(&a)); TRY(hipEventCreate(&b));
  hipLaunchKernelGGL(add_records<kMode>, dim3(blocks), dim3(256), 0, 0, table, records, 1, coherent, copy_stride);   // warm
  TRY(hipDeviceSynchronize());
  TRY(hipMemset(table, 0, bytes));
  TRY(hipEventRecord(a));
  hipLaunchKernelGGL(add_records<kMode>, dim3(blocks), dim3(256), 0, 0, table, records, rounds, coherent, copy_stride);
  TRY(hipEventRecord(b)); TRY(hipEventSynchronize(b));
  float ms = 0; TRY(hipEventElapsedTime(&ms, a, b));
  std::vector<char> host(bytes);
  TRY(hipMemcpy(host.data(), table, bytes, hipMemcpyDeviceToHost));
  double total = 0;
  const size_t n = bytes / 8;
  if (kMode == 2 || kMode == 6) { const float* f = reinterpret_cast<const float*>(host.data()); for (size_t i = 0; i < 2 * n; ++i) total += f[i]; }
  else if (kMode == 3 || kMode == 7) { const unsigned long long* u = reinterpret_cast<const unsigned long long*>(host.data()); for (size_t i = 0; i < n; ++i) total += (double)u[i]; }
  else { const double* d = reinterpret_cast<const double*>(host.data()); for (size_t i = 0; i < n; ++i) total += d[i]; }
  *total_out = total;
  return ms;
}

int main() {
  const uint32_t records = 76800;
  const size_t copy_stride = (size_t)records * 8 + 1024;     // doubles per private copy
  const size_t bytes = copy_stride * 16 * 8;
  double* table; TRY(hipMalloc(&table, bytes));
  const int blocks = 2048, rounds = 800;                      // 8192 waves x 800 rounds x 8 records = 52.4 M records
  const double records_added = (double)blocks * 4 * rounds * 8;
  const char* names[9] = {"f64 shared", "f64 per-XCC", "f32 shared", "u64 shared", "f64 wg-scope per-XCC", "f64 sc1 shared", "f32 per-XCC", "u64 per-XCC",
                          "f64 lane-per-record"};
  for (int coherent = 0; coherent < 2; ++coherent) {
    for (int mode = 0; mode < 9; ++mode) {
      double total = 0; float ms = 0;
      switch (mode) {
        case 0: ms = run<0>(table, records, blocks, rounds, coherent, copy_stride, bytes, &total); break;
        case 1: ms = run<1>(table, records, blocks, rounds, coherent, copy_stride, bytes, &total); break;
        case 2: ms = run<2>(table, records, blocks, rounds, coherent, copy_stride, bytes, &total); break;
        case 3: ms = run<3>(table, records, blocks, rounds, coherent, copy_stride, bytes, &total); break;
        case 4: ms = run<4>(table, records, blocks, rounds, coherent, copy_stride, bytes, &total); break;
        case 5: ms = run<5>(table, records, blocks, rounds, coherent, copy_stride, bytes, &total); break;
        case 6: ms = run<6>(table, records, blocks, rounds, coherent, copy_stride, bytes, &total); break;
        case 7: ms = run<7>(table, records, blocks, rounds, coherent, copy_stride, bytes, &total); break;
        case 8: ms = run<8>(table, records, blocks, rounds / 8, coherent, copy_stride, bytes, &total); break;
      }
      const double added = mode == 8 ? (double)blocks * 4 * (rounds / 8) * 64 : records_added;
      printf("%-9s %-22s %8.3f ms  %7.2f G records/s  lost updates: %.0f\n", coherent ? "coherent" : "random", names[mode], ms, added / ms * 1e-6,
             added * 8 - total);
    }
  }
  return 0;
}
