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

// two f32 -> three packed bf16 pairs; every residual subtraction is exact (|x - bf16(x)| <= half a bf16 ulp of x)
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = cvt_pk_bf16(x0, x1);
    float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(r0, r1);
    r0 -= __uint_as_float(p2 << 16);
    r1 -= __uint_as_float(p2 & 0xffff0000u);
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
