// cuda_runtime.h -- a HOST stand-in for the CUDA runtime header, just large enough to compile the reference's device-math
// headers (applications/badslam/src/badslam/*.cuh) with g++.  TEST INFRASTRUCTURE (see oracle/oracle.h): it exists so that
// oracle/_ref/libbadslam_ref.so can be built from the reference's own sources, read where they lie under /root/reference at
// build time -- nothing of the reference is copied into this repository.
//
// What is modelled:
//   * the vector PODs and make_* constructors the headers use;
//   * __device__ / __host__ / __forceinline__ as nothing / inline;  __CUDA_ARCH__ defined, so Norm() takes its sqrtf branch
//     (B/cuda_util.cuh:78-85);
//   * tex2D<float4> on a uchar4 image: the reference's colour texture is created with clamp addressing, linear filtering,
//     cudaReadModeNormalizedFloat, unnormalised coordinates (B/keyframe.cc:67-73).  CUDA's linear filter (programming guide,
//     "Linear Filtering"): xB = x - 0.5, i = floor(xB), alpha = frac(xB) kept in 9-bit fixed point with 8 fractional bits;
//     result = (1-a)(1-b) T[i,j] + a (1-b) T[i+1,j] + (1-a) b T[i,j+1] + a b T[i+1,j+1].  `quantize_weights` selects the
//     hardware's 8-bit weights (round to nearest) or exact binary32 weights (what oracle and kernels use);
//   * __syncthreads_or(x) == x for the single "thread" that runs in ref_entry.cc; under REF_BLOCK_COLLECTIVES (ref_kernels.cc) the
//     block vote of the stand-in launcher.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#ifndef __CUDA_ARCH__
#define __CUDA_ARCH__ 600
#endif

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { unsigned int x, y; };
struct uchar4 { unsigned char x, y, z, w; };
struct uchar3 { unsigned char x, y, z; };
inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
inline float3 make_float3(float x, float y, float z) { float3 r = {x, y, z}; return r; }
inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
inline int2 make_int2(int x, int y) { int2 r = {x, y}; return r; }
inline uint2 make_uint2(unsigned int x, unsigned int y) { uint2 r = {x, y}; return r; }
inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { uchar4 r = {x, y, z, w}; return r; }

inline bool isnan(float x) { return x != x; }   // CUDA's global-namespace overload (B/kernel_opt_intrinsics.cu:400)
// the headers call ::min / ::max (CUDA's global overloads)
inline float max(float a, float b) { return a > b ? a : b; }
inline float min(float a, float b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }

typedef void* cudaStream_t;
typedef unsigned long long cudaTextureObject_t;   // here: the address of a RefTexture

struct RefTexture {
  const uchar4* texels;   // pitch-linear RGBA, w = luma (B/cuda_image_processing.cu:165-193)
  int width, height;
  size_t pitch_bytes;
  int quantize_weights;   // 1: 8 fractional bits like the texture unit; 0: exact binary32 weights
};

template <typename T> T tex2D(cudaTextureObject_t tex, float x, float y);

inline float ref_filter_weight(float frac, int quantize) {
  return quantize ? std::floor(frac * 256.f + 0.5f) * (1.f / 256.f) : frac;
}
template <> inline float4 tex2D<float4>(cudaTextureObject_t handle, float x, float y) {
  const RefTexture& t = *reinterpret_cast<const RefTexture*>(handle);
  const float xb = x - 0.5f, yb = y - 0.5f;
  const float fx = std::floor(xb), fy = std::floor(yb);
  const float a = ref_filter_weight(xb - fx, t.quantize_weights), b = ref_filter_weight(yb - fy, t.quantize_weights);
  const int i = (int)fx, j = (int)fy;
  auto at = [&](int u, int v) {
    u = u < 0 ? 0 : (u >= t.width ? t.width - 1 : u);     // cudaAddressModeClamp
    v = v < 0 ? 0 : (v >= t.height ? t.height - 1 : v);
    const uchar4 c = *reinterpret_cast<const uchar4*>(reinterpret_cast<const char*>(t.texels) + (size_t)v * t.pitch_bytes + (size_t)u * 4);
    return make_float4(c.x * (1.f / 255.f), c.y * (1.f / 255.f), c.z * (1.f / 255.f), c.w * (1.f / 255.f));   // cudaReadModeNormalizedFloat
  };
  const float4 t00 = at(i, j), t10 = at(i + 1, j), t01 = at(i, j + 1), t11 = at(i + 1, j + 1);
  const float w00 = (1 - a) * (1 - b), w10 = a * (1 - b), w01 = (1 - a) * b, w11 = a * b;
  return make_float4(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x, w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
                     w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z, w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w);
}
// single-channel float textures (the *WithFloatTexture variants of the headers; not exercised here, but they must compile)
template <> inline float tex2D<float>(cudaTextureObject_t handle, float x, float y) { return tex2D<float4>(handle, x, y).w; }

#ifdef REF_BLOCK_COLLECTIVES   // ref_kernels.cc: whole kernels under the stand-in launcher (libvis/cuda/cuda_auto_tuner.h)
int ref_syncthreads_or(int predicate);
inline int __syncthreads_or(int predicate) { return ref_syncthreads_or(predicate); }
void ref_syncthreads();
inline void __syncthreads() { ref_syncthreads(); }
#else                          // ref_entry.cc: single functions, one "thread"
inline int __syncthreads_or(int predicate) { return predicate; }
#endif

// host stand-ins for the three runtime calls of CreateSurfelsForKeyframeCUDA_CountNewSurfels (B/kernel_create_surfels.cu:432-475)
enum cudaMemcpyKind { cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3 };
inline int cudaMalloc(void** ptr, size_t bytes) { *ptr = std::malloc(bytes ? bytes : 1); return 0; }
inline int cudaMemcpyAsync(void* dst, const void* src, size_t bytes, cudaMemcpyKind, cudaStream_t) { std::memcpy(dst, src, bytes); return 0; }
inline int cudaStreamSynchronize(cudaStream_t) { return 0; }
inline int cudaMemsetAsync(void* dst, int value, size_t bytes, cudaStream_t) { std::memset(dst, value, bytes); return 0; }

// what the whole kernels of ref_kernels.cc need on top of the device-math headers (one "thread" at a time, several OpenMP threads)
#define __shared__ static thread_local
inline unsigned int atomicAdd(unsigned int* address, unsigned int value) { return __atomic_fetch_add(address, value, __ATOMIC_RELAXED); }
inline float atomicAdd(float* address, float value) {   // binary32 add, atomically (blocks run on several OpenMP threads)
  unsigned int* word = reinterpret_cast<unsigned int*>(address);
  unsigned int seen = __atomic_load_n(word, __ATOMIC_RELAXED), wanted;
  float before;
  do {
    std::memcpy(&before, &seen, sizeof(before));
    const float after = before + value;
    std::memcpy(&wanted, &after, sizeof(wanted));
  } while (!__atomic_compare_exchange_n(word, &seen, wanted, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return before;
}
inline unsigned int atomicCAS(unsigned int* address, unsigned int compare, unsigned int value) {   // returns the old word
  __atomic_compare_exchange_n(address, &compare, value, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return compare;
}
inline unsigned long long atomicCAS(unsigned long long* address, unsigned long long compare, unsigned long long value) {
  __atomic_compare_exchange_n(address, &compare, value, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return compare;
}
inline long long __double_as_longlong(double d) { long long i; std::memcpy(&i, &d, sizeof(i)); return i; }
inline double __longlong_as_double(long long i) { double d; std::memcpy(&d, &i, sizeof(d)); return d; }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, sizeof(i)); return i; }
struct __half { unsigned short bits; };
inline __half __ushort_as_half(unsigned short bits) { __half h = {bits}; return h; }
inline float __half2float(__half h) {   // binary16 -> binary32, exact
  const uint32_t sign = (uint32_t)(h.bits & 0x8000u) << 16, exponent = (h.bits >> 10) & 0x1fu, mantissa = h.bits & 0x3ffu;
  uint32_t out;
  if (exponent == 0) {
    if (mantissa == 0) { out = sign; }
    else { int e = -1; uint32_t m = mantissa; do { ++e; m <<= 1; } while (!(m & 0x400u)); out = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13); }
  } else if (exponent == 31) { out = sign | 0x7f800000u | (mantissa << 13); }
  else { out = sign | ((exponent + 127 - 15) << 23) | (mantissa << 13); }
  float f; std::memcpy(&f, &out, sizeof(f)); return f;
}
inline int __all(int predicate) { return predicate; }
// keyframe preprocessing (ref_preprocess.cc: B/cuda_depth_processing.cu, B/cuda_image_processing.cu)
inline int atomicMin(int* address, int value) {   // returns the old word
  int seen = __atomic_load_n(address, __ATOMIC_RELAXED);
  while (value < seen && !__atomic_compare_exchange_n(address, &seen, value, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return seen;
}
inline int atomicMax(int* address, int value) {
  int seen = __atomic_load_n(address, __ATOMIC_RELAXED);
  while (value > seen && !__atomic_compare_exchange_n(address, &seen, value, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return seen;
}
inline __half __float2half_rn(float f) {   // binary32 -> binary16, round to nearest even, overflow to infinity, subnormals kept
  uint32_t in; std::memcpy(&in, &f, sizeof(in));
  const uint32_t sign = (in >> 16) & 0x8000u, magnitude = in & 0x7fffffffu;
  uint16_t out;
  if (magnitude >= 0x7f800000u) out = (uint16_t)(magnitude > 0x7f800000u ? 0x7fffu : 0x7c00u);          // NaN (CUDA's canonical 0x7fff) / infinity
  else if (magnitude >= 0x477ff000u) out = 0x7c00u;                                                      // rounds to >= 2^16: infinity
  else if (magnitude < 0x33000001u) out = 0;                                                             // <= 2^-25: rounds to zero
  else {
    const int exponent = (int)(magnitude >> 23) - 127;                                                   // unbiased
    uint32_t mantissa = (magnitude & 0x7fffffu) | 0x800000u;                                             // 24 bits, leading one explicit
    const int shift = exponent >= -14 ? 13 : 13 + (-14 - exponent);                                      // bits dropped (more for a subnormal result)
    const uint32_t kept = mantissa >> shift, rest = mantissa & ((1u << shift) - 1u), half = 1u << (shift - 1);
    uint32_t rounded = kept + ((rest > half || (rest == half && (kept & 1u))) ? 1u : 0u);
    // normal: rounded has its leading one at bit 10, adding (exponent + 14) << 10 on top of it gives biased exponent + mantissa, and a
    // carry out of the mantissa moves into the exponent by itself; subnormal: the exponent field is what the carry makes it
    out = (uint16_t)(exponent >= -14 ? rounded + ((uint32_t)(exponent + 14) << 10) : rounded);
  }
  __half h = {(unsigned short)(sign | out)};
  return h;
}
inline unsigned short __half_as_ushort(__half h) { return h.bits; }

