// Thin nn.Linear: one side of the weight matrix is at most 16 wide and the rows number in the millions - the per-neighbour linears of
// the point transformer (pointtransformer.py:24-29: linear_p 3 -> 3 -> C, linear_w C -> C/8 -> C/8 on [n * nsample] rows) and their
// input gradients.  A 64 x 64 MFMA tile spends 61 of its 64 columns on padding there and, with lda = 3, loads a row as three scalar
// dwords: 370 us for the 50 MB of a (3, 3) layer over 2 M rows (profiles/r04_train_full_calls_before.txt).  These layers are streams:
// the rows of A and C are contiguous (lda == K, ldc == N), so a block of rows is ONE flat array that is loaded and stored with
// full 16-byte lanes through LDS, and the arithmetic is a short f32 FMA chain per output, k ascending (exact f32 products, no MFMA).
//   K <= 16 (linear_thin_k): x rows and W in LDS; a thread produces four consecutive outputs of the flat C block.
//   N <= 16, K % 4 == 0, K <= 64 (linear_thin_n): a thread owns a row, its accumulators live in registers, W comes through the scalar
//   cache (uniform addresses), x as 16-byte LDS reads from rows padded by four floats (conflict-free).
// The choice is a function of (N, K, strides, epilogue fields) - never of M and never of pointer alignment (unaligned views run the same
// chain through scalar accesses) -, so a batch and its shards run the same arithmetic.  Epilogue: scale, bias, act (the forms these layers use); anything else takes the MFMA kernels.
#include "common.h"
#include "profile.h"

namespace {

constexpr int THIN_MAX = 16;          // the small side
constexpr int THIN_N_KMAX = 64;       // the long side of the N-thin form (LDS: 256 rows x (K + 4) floats)
constexpr int THIN_K_NMAX = 256;      // the long side of the K-thin form

__device__ __forceinline__ void stage_flat(const float* __restrict__ src, float* dst, int count, int tid, int nthreads) {
    for (int i = tid * 4; i < count; i += nthreads * 4) {
        if (i + 3 < count) {
            *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
        } else {
            for (int j = i; j < count; ++j) dst[j] = src[j];
        }
    }
}

// K <= 16.  R rows per block (R % 4 == 0, R * N <= 4096): the block's outputs are R * N consecutive floats of C.
template <bool N4>
__global__ __launch_bounds__(256) void linear_thin_k_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                            const float* __restrict__ scale, float* __restrict__ C, int64_t M, int N, int K, int R,
                                                            int64_t nblk, int act) {
    extern __shared__ __align__(16) float sm[];
    float* Xs = sm;                                   // [R][K] flat
    float* Ws = Xs + ((R * K + 3) & ~3);              // [N][K]
    float* Bs = Ws + ((N * K + 3) & ~3);              // bias [N], scale [N]
    const int tid = threadIdx.x;
    for (int i = tid; i < N * K; i += 256) Ws[i] = W[i];
    for (int i = tid; i < N; i += 256) { Bs[i] = bias ? bias[i] : 0.0f; Bs[N + i] = scale ? scale[i] : 1.0f; }
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t m0 = blk * R;
        const int vr = (int)(M - m0 < R ? M - m0 : R);
        __syncthreads();                              // the previous block's reads of Xs are done (and W / bias are staged)
        stage_flat(A + m0 * K, Xs, vr * K, tid, 256);
        __syncthreads();
        const int outc = vr * N;
        float* dst = C + m0 * N;
        for (int e0 = tid * 4; e0 < outc; e0 += 1024) {
            float v[4];
            int row = e0 / N, n = e0 - row * N;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.0f;
                if (e0 + c < outc) {
                    const float* x = Xs + row * K;
                    const float* w = Ws + n * K;
                    for (int k = 0; k < K; ++k) acc = __builtin_fmaf(x[k], w[k], acc);
                    if (scale) acc *= Bs[N + n];
                    if (bias) acc += Bs[n];
                    acc = apply_act(acc, act);
                }
                v[c] = acc;
                if (N4) { ++n; } else if (++n == N) { n = 0; ++row; }
            }
            if (e0 + 3 < outc) *reinterpret_cast<float4*>(dst + e0) = make_float4(v[0], v[1], v[2], v[3]);
            else for (int c = 0; e0 + c < outc; ++c) dst[e0 + c] = v[c];
        }
    }
}

// K <= KT <= 16 and N a power of two in 4 .. 256: a thread keeps its column quad for the whole launch (256 is a multiple of N / 4), so its
// 4 x K weights, bias and scale live in registers and a row costs it K LDS reads - the generic kernel above re-reads the weights for
// every output (N = 128, K = 16 over 262144 rows: 909 us there, LDS-bound).  Same FMA chain per output, k ascending: the same bits.
template <int KT>
__global__ __launch_bounds__(256) void linear_thin_kreg_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                               const float* __restrict__ scale, float* __restrict__ C, int64_t M, int N, int K, int R,
                                                               int64_t nblk, int act) {
    extern __shared__ __align__(16) float sm[];
    const int tid = threadIdx.x, nq = N >> 2, cq = tid % nq, r0 = tid / nq, rstep = 256 / nq;
    const int KP = (K & 7) ? K : K + 4;
    float w[4][KT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < KT; ++k) w[c][k] = k < K ? W[(cq * 4 + c) * K + k] : 0.0f;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), s4 = make_float4(1.f, 1.f, 1.f, 1.f);
    if (bias) b4 = *reinterpret_cast<const float4*>(bias + cq * 4);
    if (scale) s4 = *reinterpret_cast<const float4*>(scale + cq * 4);
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t m0 = blk * R;
        const int vr = (int)(M - m0 < R ? M - m0 : R);
        __syncthreads();
        if (KP == K) {
            stage_flat(A + m0 * K, sm, vr * K, tid, 256);
        } else {                                       // K % 8 == 0: rows padded by four floats (rows of 32 floats would all start in bank 0)
            const float* src = A + m0 * K;
            const int kq = K >> 2;
            for (int i4 = tid; i4 < vr * kq; i4 += 256) {
                const int row = i4 / kq, cq4 = i4 - row * kq;
                *reinterpret_cast<float4*>(sm + row * KP + cq4 * 4) = *reinterpret_cast<const float4*>(src + (int64_t)i4 * 4);
            }
        }
        __syncthreads();
        float* dst = C + m0 * N + cq * 4;
        for (int row = r0; row < vr; row += rstep) {
            float x[KT];
#pragma unroll
            for (int k = 0; k < KT; ++k) x[k] = k < K ? sm[row * KP + k] : 0.0f;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    if (k < K) acc = __builtin_fmaf(x[k], w[c][k], acc);          // uniform
                v[c] = acc;
            }
            if (scale) { v[0] *= s4.x; v[1] *= s4.y; v[2] *= s4.z; v[3] *= s4.w; }
            if (bias) { v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
            *reinterpret_cast<float4*>(dst + (int64_t)row * N) = make_float4(apply_act(v[0], act), apply_act(v[1], act), apply_act(v[2], act), apply_act(v[3], act));
        }
    }
}

// N <= NT <= 16, K % 4 == 0, K <= 64.  256 rows per block, one per thread.
template <int NT>
__global__ __launch_bounds__(256) void linear_thin_n_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                            const float* __restrict__ scale, float* __restrict__ C, int64_t M, int N, int K,
                                                            int64_t nblk, int act) {
    extern __shared__ __align__(16) float sm[];
    const int KP = K + 4;
    float* Xs = sm;                                   // [256][KP]
    float* Os = Xs + 256 * KP;                        // [256][N] flat
    const int tid = threadIdx.x, kq = K >> 2;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int64_t m0 = blk * 256;
        const int vr = (int)(M - m0 < 256 ? M - m0 : 256);
        __syncthreads();                              // the previous block's reads of Xs / Os are done
        const float* src = A + m0 * K;
        for (int i4 = tid; i4 < vr * kq; i4 += 256) {
            const int row = i4 / kq, cq = i4 - row * kq;
            *reinterpret_cast<float4*>(Xs + row * KP + cq * 4) = *reinterpret_cast<const float4*>(src + (int64_t)i4 * 4);
        }
        __syncthreads();
        if (tid < vr) {
            float acc[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[n] = 0.0f;
            const float* x = Xs + tid * KP;
            for (int q = 0; q < kq; ++q) {
                const float4 xv = *reinterpret_cast<const float4*>(x + q * 4);
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (n < N) {                       // uniform
                        const float* w = W + n * K + q * 4;      // uniform address: scalar loads
                        acc[n] = __builtin_fmaf(xv.x, w[0], acc[n]);
                        acc[n] = __builtin_fmaf(xv.y, w[1], acc[n]);
                        acc[n] = __builtin_fmaf(xv.z, w[2], acc[n]);
                        acc[n] = __builtin_fmaf(xv.w, w[3], acc[n]);
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (n < N) {
                    float v = acc[n];
                    if (scale) v *= scale[n];
                    if (bias) v += bias[n];
                    Os[tid * N + n] = apply_act(v, act);
                }
            }
        }
        __syncthreads();
        const int outc = vr * N;
        float* dst = C + m0 * N;
        for (int i = tid * 4; i < outc; i += 1024) {
            if (i + 3 < outc) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(Os + i);
            else for (int j = i; j < outc; ++j) dst[j] = Os[j];
        }
    }
}

// The same layers through pointers that are NOT 16-byte aligned (an offset view of a row pool): one output per thread, scalar loads, the same
// FMA chain (k ascending from zero, then scale, bias, activation) - slow, and bit-identical to the staged kernels above, so that WHICH
// arithmetic a layer runs is a function of its shape alone and never of where its rows happen to start.
__global__ __launch_bounds__(256) void linear_thin_scalar_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                                                                 const float* __restrict__ scale, float* __restrict__ C, int64_t M, int N, int K, int act) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < M * N; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = e / N;
        const int n = (int)(e - row * N);
        const float* x = A + row * K;
        const float* w = W + (int64_t)n * K;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = __builtin_fmaf(x[k], w[k], acc);
        if (scale) acc *= scale[n];
        if (bias) acc += bias[n];
        C[e] = apply_act(acc, act);
    }
}

bool plain_epilogue(const afm_linear_args& a) {
    return a.C && !a.residual && !a.rowtab && !a.act_post && !a.a_grp && !a.c_grp && !a.ddpm_out && !a.preact && !a.dact_z && !a.dact &&
           !(a.drop_p > 0.0f) && !a.rowdot_w && !a.rowdot_out && !a.ln_out && !a.stat_out && !a.a_stat && !a.res_stat && !a.aux_dst &&
           !a.a_skip && !a.c_skip && !a.ddpm_out2;
}

}  // namespace

// 0 = not a thin layer (the caller goes on to the MFMA kernels), 1 = K-thin (any N, N * K <= 144), 2 = N-thin, 3 = K-thin with the weights in registers
int afm_linear_thin_mode(const afm_linear_args& a) {
    if (a.arith != AFM_ARITH_DEFAULT && a.arith != AFM_ARITH_BF16X9) return 0;      // explicit arithmetic requests keep their kernels
    if (!plain_epilogue(a) || a.lda != a.K || a.ldw != a.K || a.ldc != a.N) return 0;
    // (shape, strides and epilogue fields only - NOT pointer alignment: unaligned rows take linear_thin_scalar_kernel, the same FMA chain)
    const bool n_pow2 = a.N >= 4 && a.N <= THIN_K_NMAX && (a.N & (a.N - 1)) == 0;
    if (a.K <= THIN_MAX && n_pow2) return 3;
    if (a.K <= THIN_MAX && a.N * a.K <= 144) return 1;
    if (a.N <= THIN_MAX && (a.K & 3) == 0 && a.K <= THIN_N_KMAX) return 2;
    // (K = 32 with the weights of a column quad in 128 registers: measured 76 us against the MFMA tile kernel's 52 over 262144 x 32 x 32 - not taken)
    return 0;
}

int afm_linear_thin(const afm_linear_args& a, int mode, hipStream_t s) {
    AfmProf prof(AFM_PROF_MISC, 4.0 * a.M * (a.N + a.K), s);          // work = bytes streamed
    if ((((uintptr_t)a.A | (uintptr_t)a.C | (uintptr_t)a.W | (uintptr_t)a.bias | (uintptr_t)a.scale) & 15) != 0) {      // an offset view: same arithmetic, scalar accesses
        const int64_t n = (int64_t)a.M * a.N;
        const unsigned grid = (unsigned)(((n + 255) / 256) < 8192 ? ((n + 255) / 256) : 8192);
        hipLaunchKernelGGL(linear_thin_scalar_kernel, dim3(grid ? grid : 1), dim3(256), 0, s, a.A, a.W, a.bias, a.scale, a.C, (int64_t)a.M, a.N, a.K, a.act);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    if (mode == 1) {
        int R = 4096 / a.N;
        if (R > 8192 / a.K) R = 8192 / a.K;                             // x rows of a block: at most 32 KB of LDS
        R &= ~3;
        if (R > 1024) R = 1024;
        if (R < 4) R = 4;                                               // N > 1024 never gets here
        const int64_t nblk = ((int64_t)a.M + R - 1) / R;
        const size_t lds = (size_t)(((R * a.K + 3) & ~3) + ((a.N * a.K + 3) & ~3) + 2 * a.N) * sizeof(float);
        const unsigned grid = (unsigned)(nblk < 4096 ? nblk : 4096);
        if (a.N % 4 == 0)
            hipLaunchKernelGGL(linear_thin_k_kernel<true>, dim3(grid), dim3(256), lds, s, a.A, a.W, a.bias, a.scale, a.C, (int64_t)a.M, a.N, a.K, R, nblk, a.act);
        else
            hipLaunchKernelGGL(linear_thin_k_kernel<false>, dim3(grid), dim3(256), lds, s, a.A, a.W, a.bias, a.scale, a.C, (int64_t)a.M, a.N, a.K, R, nblk, a.act);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    if (mode == 3) {
        const int KP = (a.K & 7) ? a.K : a.K + 4;
        int R = 16384 / a.N;
        if (R > 8192 / KP) R = 8192 / KP;
        R &= ~3;
        const int64_t nblk = ((int64_t)a.M + R - 1) / R;
        const size_t lds = (size_t)R * KP * sizeof(float);
        const unsigned grid = (unsigned)(nblk < 4096 ? nblk : 4096);
#define AFM_THIN_KR(KT_) hipLaunchKernelGGL(linear_thin_kreg_kernel<KT_>, dim3(grid), dim3(256), lds, s, a.A, a.W, a.bias, a.scale, a.C, (int64_t)a.M, a.N, a.K, R, nblk, a.act)
        if (a.K <= 4) AFM_THIN_KR(4); else if (a.K <= 8) AFM_THIN_KR(8); else AFM_THIN_KR(16);
#undef AFM_THIN_KR
        AFM_CHECK_LAUNCH();
        return 0;
    }
    const int64_t nblk = ((int64_t)a.M + 255) / 256;
    const size_t lds = (size_t)(256 * (a.K + 4) + 256 * a.N) * sizeof(float);
    const unsigned grid = (unsigned)(nblk < 4096 ? nblk : 4096);
#define AFM_THIN_N(NT_)                                                                                                                              \
    do {                                                                                                                                             \
        static const int attr = (int)hipFuncSetAttribute((const void*)linear_thin_n_kernel<NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); \
        if (attr != 0) return attr;                                                                                                                  \
        hipLaunchKernelGGL(linear_thin_n_kernel<NT_>, dim3(grid), dim3(256), lds, s, a.A, a.W, a.bias, a.scale, a.C, (int64_t)a.M, a.N, a.K, nblk, a.act); \
    } while (0)
    if (a.N <= 4) AFM_THIN_N(4); else if (a.N <= 8) AFM_THIN_N(8); else AFM_THIN_N(16);
#undef AFM_THIN_N
    AFM_CHECK_LAUNCH();
    return 0;
}
