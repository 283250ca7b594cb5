// math_constants.h -- HOST stand-in for CUDA's header (TEST INFRASTRUCTURE, see oracle/oracle.h): the constants
// B/kernel_delete_surfels.cu, B/cuda_depth_processing.cu and B/cuda_image_processing.cu use.  CUDART_NAN_F is the quiet NaN 0x7fffffff there (the kernel tests for exactly that pattern).
#pragma once
#include <cstdint>
#include <cstring>
inline float ref_float_from_bits(uint32_t bits) { float f; std::memcpy(&f, &bits, sizeof(f)); return f; }
#define CUDART_INF_F ref_float_from_bits(0x7f800000u)
#define CUDART_NAN_F ref_float_from_bits(0x7fffffffu)
#define CUDART_SQRT_TWO_F 1.414213562f
