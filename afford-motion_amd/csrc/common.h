// Shared device helpers for the gfx950 kernels (wave64, f32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "afm_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define AFM_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

// D[i][j] += sum_k A[i][k] * B[k][j] on v_mfma_f32_32x32x2_f32 (exact f32, 64 cycles/SIMD).
// lane l supplies a = A[i = l&31][k = l>>5], b = B[k = l>>5][j = l&31];
// lane l, reg r of the result holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// The value held by lane (l ^ 32), as one VALU instruction: v_permlane32_swap (gfx950) exchanges the upper half of one register with
// the lower half of another, so swapping two copies of v leaves {v_lo, v_lo} and {v_hi, v_hi}; the lane picks the half it does not own.
// (__shfl_xor(v, 32) is a ds_bpermute: an LDS-crossbar round trip on the critical path of the softmax.)
__device__ __forceinline__ float xor32(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

__device__ __forceinline__ float xor16(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes, result in every lane, entirely on the VALU: four DPP steps inside each row of 16 lanes (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror), then v_permlane16_swap / v_permlane32_swap across rows.  No ds_bpermute: the
// LayerNorm-style kernels (two to four of these per row on the critical path) were bound by the LDS crossbar round trips.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    v += xor16(v);
    v += xor32(v);
    return v;
}

// Reduce NV per-lane values over the 64 lanes with a halving butterfly: NV/2 + NV/4 + ... + 1 exchanges, then
// log2(64/NV) plain steps (17 cross-lane ops for NV = 16 instead of 96).  On return v[0] of lane l holds the wave
// total of value index  owner(l) = sum_b bit(l, 5-b) * (NV >> (b+1))  (NV = 16: bits 5,4,3,2 -> 8,4,2,1).
// value of lane (l ^ M) on the VALU (DPP inside a row, v_permlane*_swap across rows)
template <int M>
__device__ __forceinline__ float lane_xor(float v) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "M");
    if (M == 1) return dpp_f32<0xB1>(v);
    if (M == 2) return dpp_f32<0x4E>(v);
    if (M == 4) { const float a = dpp_f32<0x124>(v), b = dpp_f32<0x12C>(v); return (threadIdx.x & 4) ? a : b; }   // row_ror:4 / row_ror:12
    if (M == 8) return dpp_f32<0x128>(v);                                                                              // row_ror:8
    if (M == 16) return xor16(v);
    return xor32(v);
}
template <int N, int MASK>
__device__ __forceinline__ void halve_step(float* v, int lane) {
    const bool up = (lane & MASK) != 0;
#pragma unroll
    for (int q = 0; q < N / 2; ++q) {
        const float keep = up ? v[q + N / 2] : v[q];
        const float send = up ? v[q] : v[q + N / 2];
        v[q] = keep + lane_xor<MASK>(send);
    }
}
template <int NV>
__device__ __forceinline__ float wave_reduce_multi(float (&v)[NV], int lane) {
    static_assert(NV == 16 || NV == 8 || NV == 4, "NV");
    if (NV == 16) { halve_step<16, 32>(v, lane); halve_step<8, 16>(v, lane); halve_step<4, 8>(v, lane); halve_step<2, 4>(v, lane);
                    v[0] += lane_xor<2>(v[0]); v[0] += lane_xor<1>(v[0]); }
    if (NV == 8) { halve_step<8, 32>(v, lane); halve_step<4, 16>(v, lane); halve_step<2, 8>(v, lane);
                   v[0] += lane_xor<4>(v[0]); v[0] += lane_xor<2>(v[0]); v[0] += lane_xor<1>(v[0]); }
    if (NV == 4) { halve_step<4, 32>(v, lane); halve_step<2, 16>(v, lane);
                   v[0] += lane_xor<8>(v[0]); v[0] += lane_xor<4>(v[0]); v[0] += lane_xor<2>(v[0]); v[0] += lane_xor<1>(v[0]); }
    return v[0];
}
// lane that owns value index q after wave_reduce_multi<NV> (its low bits are free: use the lowest such lane)
template <int NV>
__device__ __forceinline__ constexpr int multi_owner_lane(int q) {
    return NV == 16 ? ((q >> 3) & 1) * 32 + ((q >> 2) & 1) * 16 + ((q >> 1) & 1) * 8 + (q & 1) * 4
         : NV == 8  ? ((q >> 2) & 1) * 32 + ((q >> 1) & 1) * 16 + (q & 1) * 8
                    : ((q >> 1) & 1) * 32 + (q & 1) * 16;
}
template <int NV>
__device__ __forceinline__ int multi_owned_index(int lane) {
    return NV == 16 ? ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1)
         : NV == 8  ? ((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1)
                    : ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1);
}
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {      // src_lane must be wave-uniform
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}

// Wave-wide maximum of a 64-bit key without touching the LDS crossbar: four DPP steps inside each row of 16 lanes
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror - every lane of a row ends with the row maximum), then
// the four row results are read into SGPRs (v_readlane) and combined on the scalar unit.  Returns a wave-uniform value.
// (__shfl_xor on 64-bit values is 2 x ds_bpermute per step, ~100+ cycles of dependent latency each.)
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v, const int ctrl_tag) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    switch (ctrl_tag) {
        case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, true); break;
        case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xF, 0xF, true); break;
        case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xF, 0xF, true); break;
        default: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xF, 0xF, true); break;
    }
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long row_max_u64(unsigned long long v) {
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        const unsigned long long o = dpp_u64(v, st);
        v = o > v ? o : v;
    }
    return v;
}
// 32-bit maximum / minimum over the 64 lanes, wave-uniform result: four DPP steps inside the rows, the four row results through SGPRs
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, dpp_u32<0xB1>(v)); v = max(v, dpp_u32<0x4E>(v)); v = max(v, dpp_u32<0x141>(v)); v = max(v, dpp_u32<0x140>(v));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return ~wave_max_u32(~v); }

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    v = row_max_u64(v);
    unsigned long long r = 0ull;
#pragma unroll
    for (int row = 0; row < 4; ++row) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, row * 16);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), row * 16);
        const unsigned long long k = ((unsigned long long)hi << 32) | lo;
        r = k > r ? k : r;
    }
    return r;
}

// LayerNorm of one row by one wave (dim % 4 == 0, dim <= 256 * MAXV): the row lives in registers, lane l holds the float4 at columns
// (64 i + l) * 4.  Shared by layernorm_kernel and the fused "last arriver" LayerNorm of the GEMM epilogues: no fp contraction, so that
// both contexts compile to the same operations (the fused and the two-launch forms are bit-identical).
template <int MAXV>
__device__ __forceinline__ void layernorm_row(const float* __restrict__ xp, const float* __restrict__ gamma, const float* __restrict__ beta,
                                              float* __restrict__ yp, int dim, float eps, int lane) {
#pragma clang fp contract(off)
    float4 v[MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = (c < dim) ? *reinterpret_cast<const float4*>(xp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(sum) / (float)dim;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < dim) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            sq += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)dim + eps);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < dim) {
            const float4 g = *reinterpret_cast<const float4*>(gamma + c);
            const float4 bb = *reinterpret_cast<const float4*>(beta + c);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + bb.x;
            o.y = (v[i].y - mean) * rstd * g.y + bb.y;
            o.z = (v[i].z - mean) * rstd * g.z + bb.z;
            o.w = (v[i].w - mean) * rstd * g.w + bb.w;
            *reinterpret_cast<float4*>(yp + c) = o;
        }
    }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// The same function for the fused CDM decoder, which evaluates 256 of these per point (dec_point_kernel is bound by their VALU issue slots):
// GELU(x) = max(x, 0) - |x| q / 2 with q = erfc(|x| / sqrt 2) (both signs of x; for x < 0 the small factor 1 + erf is q itself: no cancellation),
// and q / 2 = 2^P(a), a = min(|x|, 5.5), P the degree-9 least-squares fit of log2(erfc(a / sqrt 2) / 2) at 6000 Chebyshev nodes of [0, 5.5]
// (beyond 5.5, |x| q / 2 < 1e-7).  ONE transcendental (v_exp_f32, quarter rate) and 12 full-rate instructions: 16 issue slots per element against
// 19 for round 3's Abramowitz & Stegun 7.1.26 form (reciprocal + exponential), and more accurate: max |error| against float64 over all f32 inputs
// in [-9, 9] 2.4e-7 (A&S form: 3.3e-7; erff-based gelu_erf: 1.2e-7), evaluated with the f32 operations below (round 4, fit and error
// scan: numpy / scipy in the build container).
__device__ __forceinline__ float gelu_erf_fast(float x) {
    const float a = fminf(fabsf(x), 5.5f);
    float p = __builtin_fmaf(7.329674645e-08f, a, -1.913058668e-06f);
    p = __builtin_fmaf(p, a, 1.896382855e-05f);
    p = __builtin_fmaf(p, a, -6.020677392e-05f);
    p = __builtin_fmaf(p, a, -5.156729021e-04f);
    p = __builtin_fmaf(p, a, 7.680844516e-03f);
    p = __builtin_fmaf(p, a, -5.303888768e-02f);
    p = __builtin_fmaf(p, a, -4.589743018e-01f);
    p = __builtin_fmaf(p, a, -1.151143670e+00f);
    p = __builtin_fmaf(p, a, -9.999989867e-01f);
    return __builtin_fmaf(-fabsf(x), __builtin_amdgcn_exp2f(p), fmaxf(x, 0.f));
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case AFM_ACT_GELU: return gelu_erf(v);
        case AFM_ACT_RELU: return fmaxf(v, 0.0f);
        case AFM_ACT_SILU: return silu(v);
        default: return v;
    }
}

// derivative of AFM_ACT_* at the pre-activation z (backward of the fused activations)
__device__ __forceinline__ float act_grad(float z, int act) {
    switch (act) {
        case AFM_ACT_GELU: return 0.5f * (1.0f + erff(z * 0.70710678118654752440f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
        case AFM_ACT_RELU: return z > 0.0f ? 1.0f : 0.0f;
        case AFM_ACT_SILU: { const float s = 1.0f / (1.0f + __expf(-z)); return s * (1.0f + z * (1.0f - s)); }
        default: return 1.0f;
    }
}

// ---- dropout keep-mask: a counter hash of (seed, mask id, row, column), one 32-bit draw per element so that
// the forward and the backward kernels regenerate the same mask whatever their register layout is (two rounds of
// the murmur3 finaliser; keep when the draw >= p * 2^32).  Returns 0 or 1/(1-p).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
struct DropKey {
    uint32_t k0, k1, thresh; float inv_keep;
    __device__ __forceinline__ DropKey(float p, uint64_t seed, uint32_t id) {
        k0 = (uint32_t)seed ^ (id * 0x9E3779B1u);
        k1 = (uint32_t)(seed >> 32) + id * 0x7FEB352Du + 0x632BE5ABu;
        const float t = p * 4294967296.0f;
        thresh = t >= 4294967040.0f ? 0xFFFFFFFFu : (uint32_t)t;
        inv_keep = 1.0f / (1.0f - p);
    }
    // element (row, col) of the logical matrix the mask belongs to (attention: row = (b*H+h)*T + query, col = key)
    __device__ __forceinline__ float operator()(uint32_t row, uint32_t col) const {
        const uint32_t h = mix32(mix32(col ^ k0) + row * 0x9E3779B1u + k1);
        return h >= thresh ? inv_keep : 0.0f;
    }
};

// ---- Philox4x32-10 counter-based generator (Salmon et al. 2011), Box-Muller normals.
// counter = (element/4 low, element/4 high, step, 0), key = seed ^ hash(global sample index).
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// 4 standard normals for (seed, sample, step, quad index q = element / 4)
__device__ __forceinline__ void philox_normal4(uint64_t seed, int64_t sample, int32_t step, uint64_t q, float (&z)[4]) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)step, (uint32_t)(sample >> 32)};
    uint32_t k0 = (uint32_t)seed ^ ((uint32_t)sample * 0x9E3779B1u), k1 = (uint32_t)(seed >> 32) ^ 0x85EBCA6Bu;
    philox4x32_10(c, k0, k1);
    const float S = 2.3283064365386963e-10f;          // 2^-32
    float u0 = ((float)c[0] + 0.5f) * S, u1 = (float)c[1] * S;
    float u2 = ((float)c[2] + 0.5f) * S, u3 = (float)c[3] * S;
    u0 = fminf(u0, 0.99999994f); u2 = fminf(u2, 0.99999994f);
    float r0 = sqrtf(-2.0f * __logf(u0)), r1 = sqrtf(-2.0f * __logf(u2));
    float s0, c0, s1, c1;
    __sincosf(6.283185307179586f * u1, &s0, &c0);
    __sincosf(6.283185307179586f * u3, &s1, &c1);
    z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}
