// Issue cost of the instruction classes the BA sweeps are made of (gfx950), and the shader clock they run at.
// For every class: 8 independent chains per wavefront, W wavefronts per SIMD, every SIMD of the chip busy; reports shader
// cycles (s_memtime) per wave64 instruction per SIMD and the effective clock (s_memtime ticks / s_memrealtime at 100 MHz).
// A VALU instruction that "costs" more than v_fma_f32 here is worth more when it is removed from a sweep.
//   hipcc --offload-arch=gfx950 -O3 -o inst_cost inst_cost.hip && ./inst_cost [waves_per_simd=4]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(S) S S S S S S S S
constexpr int kIters = 1024;

struct Stamp { unsigned long long c0, c1, r0, r1; };

#define KERNEL(NAME, BODY, ...)                                                                                        \
  __global__ void __launch_bounds__(64) NAME(float* out, Stamp* stamps, float a, float b, int ia) {                   \
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();                                           \
    _Pragma("unroll 1") for (int i = 0; i < kIters; ++i) {                                                              \
      asm volatile(REP8(BODY) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)         \
                   : "v"(a), "v"(b), "s"(a), "v"(ia) : __VA_ARGS__);                                                    \
    }                                                                                                                   \
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();                                           \
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamp{c0, c1, r0, r1};                                                   \
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;                                         \
  }

// every BODY is 8 instructions, one per chain
#define EACH(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define B_FMA "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define B_FMAC "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
#define B_FMA_S "v_fma_f32 %0, %0, %10, %9\n v_fma_f32 %1, %1, %10, %9\n v_fma_f32 %2, %2, %10, %9\n v_fma_f32 %3, %3, %10, %9\n v_fma_f32 %4, %4, %10, %9\n v_fma_f32 %5, %5, %10, %9\n v_fma_f32 %6, %6, %10, %9\n v_fma_f32 %7, %7, %10, %9\n"
#define B_FMAAK "v_fmaak_f32 %0, %0, %8, 0x3c088889\n v_fmaak_f32 %1, %1, %8, 0x3c088889\n v_fmaak_f32 %2, %2, %8, 0x3c088889\n v_fmaak_f32 %3, %3, %8, 0x3c088889\n v_fmaak_f32 %4, %4, %8, 0x3c088889\n v_fmaak_f32 %5, %5, %8, 0x3c088889\n v_fmaak_f32 %6, %6, %8, 0x3c088889\n v_fmaak_f32 %7, %7, %8, 0x3c088889\n"
#define B_MUL "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
#define B_MOV "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
#define B_MOV0 "v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0\n v_mov_b32 %4, 0\n v_mov_b32 %5, 0\n v_mov_b32 %6, 0\n v_mov_b32 %7, 0\n"
#define B_CNDMASK "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
#define B_CMP "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n"
#define B_CMP_SGPR "v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[22:23], %1, %8\n v_cmp_lt_f32 s[24:25], %2, %8\n v_cmp_lt_f32 s[26:27], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[22:23], %5, %8\n v_cmp_lt_f32 s[24:25], %6, %8\n v_cmp_lt_f32 s[26:27], %7, %8\n"
#define B_CMP_CND "v_cmp_lt_f32 vcc, %0, %8\n s_nop 1\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n s_nop 1\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_f32 vcc, %4, %8\n s_nop 1\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n s_nop 1\n v_cndmask_b32 %7, %7, %8, vcc\n"
#define B_CVT_FI "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n"
#define B_CVT_UB "v_cvt_f32_ubyte1 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte1 %2, %2\n v_cvt_f32_ubyte1 %3, %3\n v_cvt_f32_ubyte1 %4, %4\n v_cvt_f32_ubyte1 %5, %5\n v_cvt_f32_ubyte1 %6, %6\n v_cvt_f32_ubyte1 %7, %7\n"
#define B_RCP "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
#define B_SQRT "v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
#define B_FLOOR "v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_floor_f32 %2, %2\n v_floor_f32 %3, %3\n v_floor_f32 %4, %4\n v_floor_f32 %5, %5\n v_floor_f32 %6, %6\n v_floor_f32 %7, %7\n"
#define B_MED3 "v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
#define B_MUL24 "v_mul_u32_u24 %0, %0, %11\n v_mul_u32_u24 %1, %1, %11\n v_mul_u32_u24 %2, %2, %11\n v_mul_u32_u24 %3, %3, %11\n v_mul_u32_u24 %4, %4, %11\n v_mul_u32_u24 %5, %5, %11\n v_mul_u32_u24 %6, %6, %11\n v_mul_u32_u24 %7, %7, %11\n"
#define B_MULLO "v_mul_lo_u32 %0, %0, %11\n v_mul_lo_u32 %1, %1, %11\n v_mul_lo_u32 %2, %2, %11\n v_mul_lo_u32 %3, %3, %11\n v_mul_lo_u32 %4, %4, %11\n v_mul_lo_u32 %5, %5, %11\n v_mul_lo_u32 %6, %6, %11\n v_mul_lo_u32 %7, %7, %11\n"
#define B_ADDLSHL "v_add_lshl_u32 %0, %0, %11, 5\n v_add_lshl_u32 %1, %1, %11, 5\n v_add_lshl_u32 %2, %2, %11, 5\n v_add_lshl_u32 %3, %3, %11, 5\n v_add_lshl_u32 %4, %4, %11, 5\n v_add_lshl_u32 %5, %5, %11, 5\n v_add_lshl_u32 %6, %6, %11, 5\n v_add_lshl_u32 %7, %7, %11, 5\n"
#define B_ADD_U32 "v_add_u32 %0, %0, %11\n v_add_u32 %1, %1, %11\n v_add_u32 %2, %2, %11\n v_add_u32 %3, %3, %11\n v_add_u32 %4, %4, %11\n v_add_u32 %5, %5, %11\n v_add_u32 %6, %6, %11\n v_add_u32 %7, %7, %11\n"
#define B_DPP_ADD "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define B_PERMSWAP "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
#define B_PERM16 "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
#define B_PKFMA "v_pk_fma_f32 %0, %0, %0, %0\n" /* placeholder, unused */
#define B_FMA_SALU "v_fma_f32 %0, %0, %8, %9\n s_and_b64 s[20:21], s[22:23], s[24:25]\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s20, s21, s22\n v_fma_f32 %2, %2, %8, %9\n s_and_b64 s[20:21], s[22:23], s[24:25]\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s20, s21, s22\n v_fma_f32 %4, %4, %8, %9\n s_and_b64 s[20:21], s[22:23], s[24:25]\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 s20, s21, s22\n v_fma_f32 %6, %6, %8, %9\n s_and_b64 s[20:21], s[22:23], s[24:25]\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 s20, s21, s22\n"
#define B_FMA_NOP "v_fma_f32 %0, %0, %8, %9\n s_nop 0\n v_fma_f32 %1, %1, %8, %9\n s_nop 0\n v_fma_f32 %2, %2, %8, %9\n s_nop 0\n v_fma_f32 %3, %3, %8, %9\n s_nop 0\n v_fma_f32 %4, %4, %8, %9\n s_nop 0\n v_fma_f32 %5, %5, %8, %9\n s_nop 0\n v_fma_f32 %6, %6, %8, %9\n s_nop 0\n v_fma_f32 %7, %7, %8, %9\n s_nop 0\n"
#define B_FMA_SAVEEXEC "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n s_and_saveexec_b64 s[20:21], vcc\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n s_or_b64 exec, exec, s[20:21]\n"
#define B_FMA_DEP "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
#define B_FMA_DEP2 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n"
#define B_LSHL64 "v_lshlrev_b64 %0, 3, %0\n"

#define B_FMAC_S "v_fmac_f32 %0, %10, %9\n v_fmac_f32 %1, %10, %9\n v_fmac_f32 %2, %10, %9\n v_fmac_f32 %3, %10, %9\n v_fmac_f32 %4, %10, %9\n v_fmac_f32 %5, %10, %9\n v_fmac_f32 %6, %10, %9\n v_fmac_f32 %7, %10, %9\n "
#define B_MUL_S "v_mul_f32 %0, %10, %0\n v_mul_f32 %1, %10, %1\n v_mul_f32 %2, %10, %2\n v_mul_f32 %3, %10, %3\n v_mul_f32 %4, %10, %4\n v_mul_f32 %5, %10, %5\n v_mul_f32 %6, %10, %6\n v_mul_f32 %7, %10, %7\n "
#define B_CND_SGPR "v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n "
#define B_MAX_I32 "v_max_i32 %0, %0, %11\n v_max_i32 %1, %1, %11\n v_max_i32 %2, %2, %11\n v_max_i32 %3, %3, %11\n v_max_i32 %4, %4, %11\n v_max_i32 %5, %5, %11\n v_max_i32 %6, %6, %11\n v_max_i32 %7, %7, %11\n "
#define B_SUB_F32 "v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n "
#define B_AND_B32 "v_and_b32 %0, %0, %11\n v_and_b32 %1, %1, %11\n v_and_b32 %2, %2, %11\n v_and_b32 %3, %3, %11\n v_and_b32 %4, %4, %11\n v_and_b32 %5, %5, %11\n v_and_b32 %6, %6, %11\n v_and_b32 %7, %7, %11\n "
#define B_LSHR "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7\n "
#define B_OR3 "v_or3_b32 %0, %0, %11, %11\n v_or3_b32 %1, %1, %11, %11\n v_or3_b32 %2, %2, %11, %11\n v_or3_b32 %3, %3, %11, %11\n v_or3_b32 %4, %4, %11, %11\n v_or3_b32 %5, %5, %11, %11\n v_or3_b32 %6, %6, %11, %11\n v_or3_b32 %7, %7, %11, %11\n "
#define B_BFE "v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n v_bfe_u32 %4, %4, 3, 8\n v_bfe_u32 %5, %5, 3, 8\n v_bfe_u32 %6, %6, 3, 8\n v_bfe_u32 %7, %7, 3, 8\n "
#define B_MIN3 "v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n "
#define B_FRACT "v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %2, %2\n v_fract_f32 %3, %3\n v_fract_f32 %4, %4\n v_fract_f32 %5, %5\n v_fract_f32 %6, %6\n v_fract_f32 %7, %7\n "
#define B_TRUNC "v_trunc_f32 %0, %0\n v_trunc_f32 %1, %1\n v_trunc_f32 %2, %2\n v_trunc_f32 %3, %3\n v_trunc_f32 %4, %4\n v_trunc_f32 %5, %5\n v_trunc_f32 %6, %6\n v_trunc_f32 %7, %7\n "
#define B_DIVFIX "v_div_fixup_f32 %0, %0, %8, 1.0\n v_div_fixup_f32 %1, %1, %8, 1.0\n v_div_fixup_f32 %2, %2, %8, 1.0\n v_div_fixup_f32 %3, %3, %8, 1.0\n v_div_fixup_f32 %4, %4, %8, 1.0\n v_div_fixup_f32 %5, %5, %8, 1.0\n v_div_fixup_f32 %6, %6, %8, 1.0\n v_div_fixup_f32 %7, %7, %8, 1.0\n "
#define B_CVT_U32 "v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n v_cvt_u32_f32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_cvt_u32_f32 %6, %6\n v_cvt_u32_f32 %7, %7\n "
#define B_CVT_F32_I32 "v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7\n "
#define B_MOV_DPP "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n "
#define B_ADD_DPP_ROR "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n "
#define B_LDEXP "v_ldexp_f32 %0, %0, %11\n v_ldexp_f32 %1, %1, %11\n v_ldexp_f32 %2, %2, %11\n v_ldexp_f32 %3, %3, %11\n v_ldexp_f32 %4, %4, %11\n v_ldexp_f32 %5, %5, %11\n v_ldexp_f32 %6, %6, %11\n v_ldexp_f32 %7, %7, %11\n "
#define B_MAD_U24 "v_mad_u32_u24 %0, %0, %11, %11\n v_mad_u32_u24 %1, %1, %11, %11\n v_mad_u32_u24 %2, %2, %11, %11\n v_mad_u32_u24 %3, %3, %11, %11\n v_mad_u32_u24 %4, %4, %11, %11\n v_mad_u32_u24 %5, %5, %11, %11\n v_mad_u32_u24 %6, %6, %11, %11\n v_mad_u32_u24 %7, %7, %11, %11\n "
#define B_ABS_MUL "v_mul_f32 %0, |%0|, %8\n v_mul_f32 %1, |%1|, %8\n v_mul_f32 %2, |%2|, %8\n v_mul_f32 %3, |%3|, %8\n v_mul_f32 %4, |%4|, %8\n v_mul_f32 %5, |%5|, %8\n v_mul_f32 %6, |%6|, %8\n v_mul_f32 %7, |%7|, %8\n "
#define B_FMA_NEG "v_fma_f32 %0, -%0, %8, 1.0\n v_fma_f32 %1, -%1, %8, 1.0\n v_fma_f32 %2, -%2, %8, 1.0\n v_fma_f32 %3, -%3, %8, 1.0\n v_fma_f32 %4, -%4, %8, 1.0\n v_fma_f32 %5, -%5, %8, 1.0\n v_fma_f32 %6, -%6, %8, 1.0\n v_fma_f32 %7, -%7, %8, 1.0\n "
#define CLOB "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"
KERNEL(k_fma, B_FMA, CLOB)
KERNEL(k_fmac, B_FMAC, CLOB)
KERNEL(k_fma_sgpr, B_FMA_S, CLOB)
KERNEL(k_fmaak, B_FMAAK, CLOB)
KERNEL(k_mul, B_MUL, CLOB)
KERNEL(k_mov, B_MOV, CLOB)
KERNEL(k_mov0, B_MOV0, CLOB)
KERNEL(k_cndmask, B_CNDMASK, CLOB)
KERNEL(k_cmp_vcc, B_CMP, CLOB)
KERNEL(k_cmp_sgpr, B_CMP_SGPR, CLOB)
KERNEL(k_cmp_cnd, B_CMP_CND, CLOB)
KERNEL(k_cvt_i32_f32, B_CVT_FI, CLOB)
KERNEL(k_cvt_ubyte, B_CVT_UB, CLOB)
KERNEL(k_rcp, B_RCP, CLOB)
KERNEL(k_sqrt, B_SQRT, CLOB)
KERNEL(k_floor, B_FLOOR, CLOB)
KERNEL(k_med3, B_MED3, CLOB)
KERNEL(k_mul_u24, B_MUL24, CLOB)
KERNEL(k_mul_lo, B_MULLO, CLOB)
KERNEL(k_add_lshl, B_ADDLSHL, CLOB)
KERNEL(k_add_u32, B_ADD_U32, CLOB)
KERNEL(k_dpp_add, B_DPP_ADD, CLOB)
KERNEL(k_permlane32_swap, B_PERMSWAP, CLOB)
KERNEL(k_permlane16_swap, B_PERM16, CLOB)
KERNEL(k_fma_plus_salu, B_FMA_SALU, CLOB)
KERNEL(k_fma_plus_nop, B_FMA_NOP, CLOB)
KERNEL(k_fma_saveexec, B_FMA_SAVEEXEC, CLOB)
KERNEL(k_fma_dep1, B_FMA_DEP, CLOB)
KERNEL(k_fma_dep2, B_FMA_DEP2, CLOB)

KERNEL(k_b_fmac_s, B_FMAC_S, CLOB)
KERNEL(k_b_mul_s, B_MUL_S, CLOB)
KERNEL(k_b_cnd_sgpr, B_CND_SGPR, CLOB)
KERNEL(k_b_max_i32, B_MAX_I32, CLOB)
KERNEL(k_b_sub_f32, B_SUB_F32, CLOB)
KERNEL(k_b_and_b32, B_AND_B32, CLOB)
KERNEL(k_b_lshr, B_LSHR, CLOB)
KERNEL(k_b_or3, B_OR3, CLOB)
KERNEL(k_b_bfe, B_BFE, CLOB)
KERNEL(k_b_min3, B_MIN3, CLOB)
KERNEL(k_b_fract, B_FRACT, CLOB)
KERNEL(k_b_trunc, B_TRUNC, CLOB)
KERNEL(k_b_divfix, B_DIVFIX, CLOB)
KERNEL(k_b_cvt_u32, B_CVT_U32, CLOB)
KERNEL(k_b_cvt_f32_i32, B_CVT_F32_I32, CLOB)
KERNEL(k_b_mov_dpp, B_MOV_DPP, CLOB)
KERNEL(k_b_add_dpp_ror, B_ADD_DPP_ROR, CLOB)
KERNEL(k_b_ldexp, B_LDEXP, CLOB)
KERNEL(k_b_mad_u24, B_MAD_U24, CLOB)
KERNEL(k_b_abs_mul, B_ABS_MUL, CLOB)
KERNEL(k_b_fma_neg, B_FMA_NEG, CLOB)
typedef void (*Kern)(float*, Stamp*, float, float, int);
struct Entry { const char* name; Kern k; int valu_per_body; const char* note; };

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  std::vector<int> wave_counts = {4};
  if (argc > 1) { wave_counts.clear(); for (int a = 1; a < argc; ++a) wave_counts.push_back(atoi(argv[a])); }
  printf("%s: %d CUs, nominal %.0f MHz\n", p.gcnArchName, cus, p.clockRate / 1e3);
  float* out; Stamp* stamps;
  hipMalloc(&out, sizeof(float) * 64 * cus * 4 * 8);
  hipMalloc(&stamps, sizeof(Stamp) * cus * 4 * 8);
  std::vector<Stamp> host(cus * 4 * 8);
  const Entry entries[] = {
      {"v_fma_f32 (vgpr operands)", k_fma, 8, ""}, {"v_fmac_f32 e32", k_fmac, 8, ""}, {"v_fma_f32 with an SGPR operand", k_fma_sgpr, 8, ""},
      {"v_fmaak_f32 (32-bit literal)", k_fmaak, 8, ""}, {"v_mul_f32 e32", k_mul, 8, ""}, {"v_mov_b32 vgpr", k_mov, 8, ""}, {"v_mov_b32 0", k_mov0, 8, ""},
      {"v_cndmask_b32 (vcc)", k_cndmask, 8, ""}, {"v_cmp_lt_f32 -> vcc", k_cmp_vcc, 8, ""}, {"v_cmp_lt_f32 -> sgpr pair", k_cmp_sgpr, 8, ""},
      {"v_cmp + s_nop 1 + v_cndmask (per pair)", k_cmp_cnd, 8, "4 cmp + 4 cndmask + 4 s_nop"},
      {"v_cvt_i32_f32", k_cvt_i32_f32, 8, ""}, {"v_cvt_f32_ubyte1", k_cvt_ubyte, 8, ""}, {"v_rcp_f32", k_rcp, 8, ""}, {"v_sqrt_f32", k_sqrt, 8, ""},
      {"v_floor_f32", k_floor, 8, ""}, {"v_med3_f32", k_med3, 8, ""}, {"v_mul_u32_u24", k_mul_u24, 8, ""}, {"v_mul_lo_u32", k_mul_lo, 8, ""},
      {"v_add_lshl_u32", k_add_lshl, 8, ""}, {"v_add_u32", k_add_u32, 8, ""}, {"v_add_f32_dpp quad_perm", k_dpp_add, 8, ""},
      {"v_permlane32_swap_b32", k_permlane32_swap, 8, ""}, {"v_permlane16_swap_b32", k_permlane16_swap, 8, ""},
      {"v_fmac_f32 e32 with an SGPR src0", k_b_fmac_s, 8, ""},
      {"v_mul_f32 e32 with an SGPR src0", k_b_mul_s, 8, ""},
      {"v_cndmask_b32 (sgpr-pair mask)", k_b_cnd_sgpr, 8, ""},
      {"v_max_i32", k_b_max_i32, 8, ""},
      {"v_sub_f32", k_b_sub_f32, 8, ""},
      {"v_and_b32", k_b_and_b32, 8, ""},
      {"v_lshrrev_b32", k_b_lshr, 8, ""},
      {"v_or3_b32", k_b_or3, 8, ""},
      {"v_bfe_u32", k_b_bfe, 8, ""},
      {"v_min3_f32", k_b_min3, 8, ""},
      {"v_fract_f32", k_b_fract, 8, ""},
      {"v_trunc_f32", k_b_trunc, 8, ""},
      {"v_div_fixup_f32", k_b_divfix, 8, ""},
      {"v_cvt_u32_f32", k_b_cvt_u32, 8, ""},
      {"v_cvt_f32_i32", k_b_cvt_f32_i32, 8, ""},
      {"v_mov_b32_dpp quad_perm", k_b_mov_dpp, 8, ""},
      {"v_add_f32_dpp row_ror:8 bank_mask:0xc", k_b_add_dpp_ror, 8, ""},
      {"v_ldexp_f32", k_b_ldexp, 8, ""},
      {"v_mad_u32_u24", k_b_mad_u24, 8, ""},
      {"v_mul_f32 with |src| (VOP3)", k_b_abs_mul, 8, ""},
      {"v_fma_f32 -a, b, 1.0", k_b_fma_neg, 8, ""},
      {"v_fma + one SALU each", k_fma_plus_salu, 8, "8 fma + 8 salu"}, {"v_fma + s_nop 0 each", k_fma_plus_nop, 8, "8 fma + 8 s_nop"},
      {"8 v_fma + saveexec/restore", k_fma_saveexec, 8, "8 fma + 2 salu on exec"},
      {"v_fma one dependent chain", k_fma_dep1, 8, ""}, {"v_fma two dependent chains", k_fma_dep2, 8, ""},
  };
  for (int waves : wave_counts) {
    const int blocks = cus * 4 * waves;
    printf("---- %d wavefronts per SIMD (%d workgroups of one wavefront) ----\n", waves, blocks);
    for (const Entry& e : entries) {
      double best_cyc = 1e30, best_clock = 0, best_ms = 0;
      for (int rep = 0; rep < 4; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, out, stamps, 1.0001f, 0.5f, 3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(host.data(), stamps, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
        // per-wave cycles of the loop, averaged; waves on a SIMD overlap, so cycles per instruction per SIMD = wave cycles / (instructions per wave x waves)
        double cyc = 0, clock = 0;
        for (int b = 0; b < blocks; ++b) { cyc += double(host[b].c1 - host[b].c0); clock += double(host[b].c1 - host[b].c0) / (double(host[b].r1 - host[b].r0) / 100.0); }
        cyc /= blocks; clock /= blocks;
        const double per_inst = cyc / (double(kIters) * 64 * waves);
        if (per_inst < best_cyc) { best_cyc = per_inst; best_clock = clock; best_ms = ms; }
      }
      printf("%-42s %6.2f shader cycles per wave64 VALU instruction per SIMD   clock %5.0f MHz   %.3f ms  %s\n", e.name, best_cyc, best_clock, best_ms, e.note);
    }
  }
  return 0;
}
