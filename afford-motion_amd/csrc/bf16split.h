// f32 arithmetic on the bf16 matrix pipe (shared by gemm_split.hip and attention.hip): every f32 operand is split EXACTLY into
// three bf16 terms (x = x1 + x2 + x3: 8 + 8 + 8 significant bits, round-to-nearest residual chain, every subtraction exact), the
// nine cross products x_i * w_j are exact in f32 inside v_mfma_f32_32x32x16_bf16 and accumulate in f32.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));     // first-class values (HIP's uint4 / uint2 struct copies can pin arrays in scratch)
typedef float f32x2 __attribute__((ext_vector_type(2)));

// v_cvt_pk_bf16_f32 (round to nearest even); a builtin conversion, not inline asm, so the machine scheduler can place it
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// x - bf16 half of a packed pair.  Round 4 tried ONE instruction for it - v_dot2c_f32_bf16 (acc + a.lo * b.lo + a.hi * b.hi with b = {-1, 0} /
// {0, -1}: 7 instead of 11 VALU instructions per pair of floats, and the split is what the K loops are bound by) - and measured it in one call:
// NOT exact (tests/test_gpu_ops.py::test_linear_shapes fails: the dot unit does not deliver the residual bit for bit) and SLOWER (B = 32:
// 460 against 468 steps/s, B = 1: 472 against 427 us per step - the instruction is not full rate).  Kept behind AFM_SPLIT_DOT2 for the record;
// the shipped form is the shift / mask + v_sub pair, whose exactness tests/test_gpu_ops.py::test_split_reconstructs_f32_exactly pins.
#ifndef AFM_SPLIT_DOT2
#define AFM_SPLIT_DOT2 0
#endif
__device__ __forceinline__ float sub_bf16_lo(float x, uint32_t p) {
#if AFM_SPLIT_DOT2
    const bf16x2 m = {(__bf16)-1.0f, (__bf16)0.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), m, x, false);
#else
    return x - __uint_as_float(p << 16);
#endif
}
__device__ __forceinline__ float sub_bf16_hi(float x, uint32_t p) {
#if AFM_SPLIT_DOT2
    const bf16x2 m = {(__bf16)0.0f, (__bf16)-1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, p), m, x, false);
#else
    return x - __uint_as_float(p & 0xffff0000u);
#endif
}

// two f32 -> three packed bf16 pairs; every residual subtraction is exact (|x - bf16(x)| <= half a bf16 ulp of x)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = cvt_pk_bf16(x0, x1);
    float r0 = sub_bf16_lo(x0, p1), r1 = sub_bf16_hi(x1, p1);
    p2 = cvt_pk_bf16(r0, r1);
    r0 = sub_bf16_lo(r0, p2);
    r1 = sub_bf16_hi(r1, p2);
    p3 = cvt_pk_bf16(r0, r1);
}

// D[i][j] += sum_k A[i][k] B[k][j], k = 0..15: lane l supplies A[i = l & 31][k = 8 (l >> 5) .. + 7] and B[k = 8 (l >> 5) .. + 7][j = l & 31]
// (eight consecutive bf16 each); lane l, reg r of the result holds D[i = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][j = l & 31].
__device__ __forceinline__ f32x16 mfma_bf16(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// product order of the nine cross terms (index 0 = leading term): small terms first inside every K16 step
__device__ constexpr int AFM_PA[9] = {2, 2, 1, 1, 2, 0, 1, 0, 0};
__device__ constexpr int AFM_PB[9] = {2, 1, 2, 1, 0, 2, 0, 1, 0};
