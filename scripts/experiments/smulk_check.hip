// scripts/experiments/smulk_check.hip -- does S_MULK_I32 (the in-place scalar multiply by a 16-bit immediate) give x * 448 on this GPU?
// Round-4 bisect of the round-3 LDS anomaly: the failing build multiplies the loop's item register in place with s_mulk_i32 s84, 0x1c0,
// the passing one uses s_mul_i32 s0, s84, 0x1c0.  build: hipcc --offload-arch=gfx950 -O2 smulk_check.hip -o smulk_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void check(int* out_k, int* out_m, int n) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= n) return;
  int x = __builtin_amdgcn_readfirstlane(wave);
  int a = x, b = x;
  asm volatile("s_mulk_i32 %0, 0x1c0" : "+s"(a));
  asm volatile("s_mul_i32 %0, %1, 0x1c0" : "=s"(b) : "s"(x));
  if ((threadIdx.x & 63) == 0) { out_k[wave] = a; out_m[wave] = b; }
}
int main() {
  const int n = 1 << 16;
  int *dk, *dm;
  hipMalloc(&dk, n * sizeof(int)); hipMalloc(&dm, n * sizeof(int));
  hipLaunchKernelGGL(check, dim3(n * 64 / 256), dim3(256), 0, 0, dk, dm, n);
  std::vector<int> k(n), m(n);
  hipMemcpy(k.data(), dk, n * sizeof(int), hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * sizeof(int), hipMemcpyDeviceToHost);
  int bad_k = 0, bad_m = 0;
  for (int x = 0; x < n; ++x) {
    if (k[x] != x * 448) { if (bad_k++ < 10) printf("s_mulk_i32: %d * 448 -> %d (expected %d)\n", x, k[x], x * 448); }
    if (m[x] != x * 448) { if (bad_m++ < 10) printf("s_mul_i32:  %d * 448 -> %d (expected %d)\n", x, m[x], x * 448); }
  }
  printf("s_mulk_i32 wrong for %d of %d inputs; s_mul_i32 wrong for %d\n", bad_k, n, bad_m);
  return 0;
}
