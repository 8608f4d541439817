// Backward-pass kernels of the training path (loss.backward() of training_losses, utils/training.py:140-152):
// weight / bias gradients, LayerNorm backward, masked-MSE backward, transposes for the input-gradient GEMMs, AdamW.
//
// gfx950 design
//   * afm_linear_wgrad: dW[N,K] = dY^T X has BOTH operands reduction-major (the reduction index m is the slow one),
//     so tiles are staged in LDS as [32 m-rows][128 outputs] straight from coalesced float4 row reads and the MFMA
//     operands are read with conflict-free ds_read_b32 (lane = output index) - no transposes anywhere.
//     A workgroup owns a 128 x 128 output tile (4 waves x 2x2 v_mfma_f32_32x32x2_f32 tiles) and a slice of the M
//     reduction; slices are summed in a fixed order by a second kernel (deterministic, no atomics).  The output is
//     tiny (<= 1536 x 512) while M = B*T is ~10^4, so the split is what fills 256 CUs.
//   * LayerNorm backward: one wave per row (row in registers, two wave reductions), per-lane dgamma/dbeta
//     accumulators over a grid-stride row loop, fixed-order two-stage reduction.
#include "common.h"
#include "profile.h"

namespace {

struct RowMap3 {
    int grp, stride, off;
    __device__ __forceinline__ int64_t operator()(int r) const {
        return grp ? (int64_t)(r / grp) * stride + off + (r % grp) : (int64_t)r;
    }
};

// ------------------------------------------------------------------------------------------------ transpose
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        if (r < rows && c < cols) t[ty + 8 * i][tx] = in[(int64_t)r * cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (r < rows && c < cols) out[(int64_t)c * rows + r] = t[tx][ty + 8 * i];
    }
}

// ------------------------------------------------------------------------------------------------ wgrad
constexpr int RB = 16;           // reduction rows per stage (2 operands x 2 buffers x 16 x 132 floats = 33 KB LDS)
constexpr int RPT = RB / 8;      // rows per thread per operand tile
constexpr int WT = 128;          // output tile edge
constexpr int LDT = WT + 4;      // padded LDS row

// one thread stages RPT rows x 4 consecutive outputs of a [RB x 128] tile; loads are unconditional (clamped), the
// out-of-range rows / columns are zeroed when the tile is written to LDS
template <bool VEC>
__device__ __forceinline__ void wg_load(const float* __restrict__ base, int64_t ld, const RowMap3& map, int m0, int m_end, int col, int ncols,
                                        int lr, float4 (&reg)[RPT]) {
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int m = min(m0 + lr + 8 * i, m_end - 1);
        const float* p = base + map(m) * ld;
        if (VEC) {
            reg[i] = *reinterpret_cast<const float4*>(p + min(col, ncols - 4));
        } else {
            reg[i] = make_float4(p[min(col, ncols - 1)], p[min(col + 1, ncols - 1)], p[min(col + 2, ncols - 1)], p[min(col + 3, ncols - 1)]);
        }
    }
}
__device__ __forceinline__ void wg_store(float* __restrict__ lds, int m0, int m_end, int col, int ncols, int lr, int c4, const float4 (&reg)[RPT]) {
    const bool c0 = col < ncols, c1 = col + 1 < ncols, c2 = col + 2 < ncols, c3 = col + 3 < ncols;
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const bool ok = m0 + lr + 8 * i < m_end;
        float4 v = reg[i];
        v.x = (ok && c0) ? v.x : 0.f; v.y = (ok && c1) ? v.y : 0.f; v.z = (ok && c2) ? v.z : 0.f; v.w = (ok && c3) ? v.w : 0.f;
        *reinterpret_cast<float4*>(lds + (lr + 8 * i) * LDT + c4) = v;
    }
}

template <bool VEC>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const afm_linear_wgrad_args p, float* __restrict__ out, int64_t out_ld, int64_t split_stride,
                                                       int ntk, int tiles, int rows_per_split) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][RB * LDT];
    const int split = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int n0 = (tile / ntk) * WT, k0 = (tile % ntk) * WT;
    const int m_begin = split * rows_per_split, m_end = min(p.M, m_begin + rows_per_split);
    float* dst = out + split * split_stride;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r32 = lane & 31, hh = lane >> 5;
    const int lr = tid >> 5, c4 = (tid & 31) * 4;
    const RowMap3 ymap{p.dy_grp, p.dy_stride, p.dy_off}, xmap{p.x_grp, p.x_stride, p.x_off};

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (m_begin < m_end) {
        const int nst = (m_end - m_begin + RB - 1) / RB;
        float4 ra[RPT], rb[RPT];
        wg_load<VEC>(p.dY, p.lddy, ymap, m_begin, m_end, n0 + c4, p.N, lr, ra);
        wg_load<VEC>(p.X, p.ldx, xmap, m_begin, m_end, k0 + c4, p.K, lr, rb);
        wg_store(lds[0][0], m_begin, m_end, n0 + c4, p.N, lr, c4, ra);
        wg_store(lds[0][1], m_begin, m_end, k0 + c4, p.K, lr, c4, rb);
        __syncthreads();
        for (int st = 0; st < nst; ++st) {
            const int buf = st & 1, mn = m_begin + (st + 1) * RB;
            if (st + 1 < nst) {
                wg_load<VEC>(p.dY, p.lddy, ymap, mn, m_end, n0 + c4, p.N, lr, ra);
                wg_load<VEC>(p.X, p.ldx, xmap, mn, m_end, k0 + c4, p.K, lr, rb);
            }
            const float* a_base = lds[buf][0] + hh * LDT + wm * 64 + r32;
            const float* b_base = lds[buf][1] + hh * LDT + wn * 64 + r32;
#pragma unroll
            for (int s = 0; s < RB / 2; ++s) {
                const float a0 = a_base[2 * s * LDT], a1 = a_base[2 * s * LDT + 32];
                const float b0 = b_base[2 * s * LDT], b1 = b_base[2 * s * LDT + 32];
                acc[0][0] = mfma32(a0, b0, acc[0][0]);
                acc[0][1] = mfma32(a0, b1, acc[0][1]);
                acc[1][0] = mfma32(a1, b0, acc[1][0]);
                acc[1][1] = mfma32(a1, b1, acc[1][1]);
            }
            if (st + 1 < nst) {
                wg_store(lds[buf ^ 1][0], mn, m_end, n0 + c4, p.N, lr, c4, ra);
                wg_store(lds[buf ^ 1][1], mn, m_end, k0 + c4, p.K, lr, c4, rb);
            }
            __syncthreads();
        }
    }
    // lane holds output column k0 + wn*64 + tn*32 + r32 and rows mfma_row(r): 128-byte row segments
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int k = k0 + wn * 64 + tn * 32 + r32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + tm * 32 + mfma_row(r, lane);
                if (n < p.N && k < p.K) dst[(int64_t)n * out_ld + k] = acc[tm][tn][r];
            }
        }
}

// dst[r][c] (+)= sum_s src[s * split_stride + r * cols + c]   (fixed summation order)
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ src, int64_t split_stride, int S, float* __restrict__ dst,
                                                            int64_t dst_ld, int rows, int cols, int accumulate) {
    const int64_t n = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;          // four partial rows in flight (one dependent load at a time: 0.7 us per slice)
        int s = 0;
        for (; s + 3 < S; s += 4) {
            a0 += src[s * split_stride + i];
            a1 += src[(s + 1) * split_stride + i];
            a2 += src[(s + 2) * split_stride + i];
            a3 += src[(s + 3) * split_stride + i];
        }
        for (; s < S; ++s) a0 += src[s * split_stride + i];
        const float v = (a0 + a1) + (a2 + a3);
        float* d = dst + (i / cols) * dst_ld + (i % cols);
        *d = accumulate ? *d + v : v;
    }
}

// Few columns, many partial rows: dst[c] (+)= sum_s src[s * stride + c] for c < n_total; columns >= n0 go to dst1[c - n0].
// Block = CW columns x 1024 / CW partial-row lanes (loads of one thread are independent and unrolled), fixed-order LDS tree.  CW = 16: the
// tiny gradients (N * K <= 16: 64 lanes share the partial rows of one column instead of 16).
template <int CW = 64>
__global__ __launch_bounds__(1024) void reduce_cols_kernel(const float* __restrict__ src, int64_t stride, int S, float* __restrict__ dst0,
                                                           float* __restrict__ dst1, int n0, int n_total, int accumulate) {
    constexpr int RL = 1024 / CW;
    __shared__ float red[RL][CW];
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int c = blockIdx.x * CW + tx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < n_total) {
        int s = ty;
        for (; s + 3 * RL < S; s += 4 * RL) {
            a0 += src[(int64_t)s * stride + c];
            a1 += src[(int64_t)(s + RL) * stride + c];
            a2 += src[(int64_t)(s + 2 * RL) * stride + c];
            a3 += src[(int64_t)(s + 3 * RL) * stride + c];
        }
        for (; s < S; s += RL) a0 += src[(int64_t)s * stride + c];
    }
    red[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && c < n_total) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < RL; ++i) v += red[i][tx];
        float* d = c < n0 ? dst0 + c : dst1 + (c - n0);
        *d = accumulate ? *d + v : v;
    }
}

// partial column sums: out[chunk][n] = sum over the chunk's rows of dY[map(m)][n]; block = 64 columns x 4 row lanes
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dY, int64_t ld, RowMap3 map, int M, int N, int rows_per_chunk,
                                                     float* __restrict__ out) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), ty = threadIdx.x >> 6;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    float s = 0.f;
    if (c < N)
        for (int m = m0 + ty; m < m1; m += 4) s += dY[map(m) * ld + c];
    red[ty][threadIdx.x & 63] = s;
    __syncthreads();
    if (ty == 0 && c < N) out[(int64_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Skinny weight gradient (the point-cloud linears: M = n*k up to millions of rows, N or K <= 32): the output is a
// handful of 32x32 MFMA tiles and the kernel is a pure stream over dY and X, so there is no LDS staging at all - each
// lane loads its own MFMA operand (row m = 2s + lane/32, column lane%32: 128-byte row segments) and every wave owns a
// contiguous row range; the bias gradient is accumulated from the same dY loads.  One partial per wave, summed in a
// fixed order afterwards.
template <int TN, int TK>
__global__ __launch_bounds__(256) void wgrad_skinny_kernel(const float* __restrict__ dY, int64_t lddy, const float* __restrict__ X, int64_t ldx,
                                                           int M, int N, int K, int rows_per_wave, float* __restrict__ part_w,
                                                           float* __restrict__ part_b) {
    const int lane = threadIdx.x & 63, r32 = lane & 31, hh = lane >> 5;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m0 = gw * rows_per_wave, m1 = min(M, m0 + rows_per_wave);
    f32x16 acc[TN][TK];
    float bsum[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    int ncol[TN], kcol[TK];
    bool nok[TN], kok[TK];
#pragma unroll
    for (int i = 0; i < TN; ++i) { nok[i] = i * 32 + r32 < N; ncol[i] = min(i * 32 + r32, N - 1); }
#pragma unroll
    for (int j = 0; j < TK; ++j) { kok[j] = j * 32 + r32 < K; kcol[j] = min(j * 32 + r32, K - 1); }
    constexpr int U = (TN + TK <= 3) ? 8 : 4;             // MFMA steps per unrolled iteration (loads issued first)
    for (int m = m0; m < m1; m += 2 * U) {
        float a[U][TN], b[U][TK];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int row = m + 2 * u + hh;
            const bool ok = row < m1;
            const int64_t rc = min(row, m1 - 1);
#pragma unroll
            for (int i = 0; i < TN; ++i) { const float v = dY[rc * lddy + ncol[i]]; a[u][i] = (ok && nok[i]) ? v : 0.f; }
#pragma unroll
            for (int j = 0; j < TK; ++j) { const float v = X[rc * ldx + kcol[j]]; b[u][j] = (ok && kok[j]) ? v : 0.f; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                bsum[i] += a[u][i];
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j] = mfma32(a[u][i], b[u][j], acc[i][j]);
            }
        }
    }
    float* pw = part_w + (int64_t)gw * N * K;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int k = j * 32 + r32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = i * 32 + mfma_row(r, lane);
                if (n < N && k < K) pw[(int64_t)n * K + k] = acc[i][j][r];
            }
        }
        const float t = bsum[i] + xor32(bsum[i]);
        if (part_b && hh == 0 && i * 32 + r32 < N) part_b[(int64_t)gw * N + i * 32 + r32] = t;
    }
}

// The same gradient when dY and X are dense (lddy == N, ldx == K: every point-cloud linear): a block of RB rows of either operand is ONE
// contiguous array, so it is fetched with full 16-byte lanes whatever N and K are (3, 35, 67 ...: wgrad_skinny_kernel above loads a row of
// three floats as a 24-byte wave access and ran at 0.2 - 1 TB/s, profiles/r04_train_full_calls_before.txt), parked in LDS (double
// buffered: the next block's loads fly under this block's MFMAs) and read from there in the MFMA operand layout.  Wave w of the workgroup
// takes the row pairs w, w + 4, ... of every block; one partial per wave, summed afterwards in a fixed order that depends on (M, N, K) only.
template <int TN, int TK>
__global__ __launch_bounds__(256) void wgrad_flat_kernel(const float* __restrict__ dY, const float* __restrict__ X, int M, int N, int K, int RB,
                                                         int blocks_per_wg, float* __restrict__ part, int want_bias) {
    __shared__ __align__(16) float sm[2][4096];           // per buffer: RB * N floats of dY, then RB * K floats of X (each <= 2048)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r32 = lane & 31, hh = lane >> 5;
    const int ny = RB * N, nx = RB * K;                   // multiples of 4 (RB % 8 == 0)
    const int64_t blk0 = (int64_t)blockIdx.x * blocks_per_wg;
    const int64_t nblk_all = ((int64_t)M + RB - 1) / RB;
    const int nblk = (int)(nblk_all - blk0 < blocks_per_wg ? (nblk_all - blk0 > 0 ? nblk_all - blk0 : 0) : blocks_per_wg);
    f32x16 acc[TN][TK];
    float bsum[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }
    float4 ry[2], rx[2];
    auto gload = [&](int b) {
        const int64_t m = (blk0 + b) * RB;
        const int vr = (int)(M - m < RB ? M - m : RB);
        const float* sy = dY + m * N;
        const float* sx = X + m * K;
        const int cy = vr * N, cx = vr * K;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = (tid + u * 256) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i + 3 < cy) v = *reinterpret_cast<const float4*>(sy + i);
            else if (i < cy) { v.x = sy[i]; if (i + 1 < cy) v.y = sy[i + 1]; if (i + 2 < cy) v.z = sy[i + 2]; }
            ry[u] = v;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i + 3 < cx) w = *reinterpret_cast<const float4*>(sx + i);
            else if (i < cx) { w.x = sx[i]; if (i + 1 < cx) w.y = sx[i + 1]; if (i + 2 < cx) w.z = sx[i + 2]; }
            rx[u] = w;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = (tid + u * 256) * 4;
            if (i < ny) *reinterpret_cast<float4*>(&sm[buf][i]) = ry[u];           // rows past M are zeros: they add nothing
            if (i < nx) *reinterpret_cast<float4*>(&sm[buf][ny + i]) = rx[u];
        }
    };
    if (nblk > 0) { gload(0); lstore(0); }
    __syncthreads();
    for (int b = 0; b < nblk; ++b) {
        if (b + 1 < nblk) gload(b + 1);
        const float* Ys = sm[b & 1];
        const float* Xs = Ys + ny;
        for (int sp = wave; sp < RB / 2; sp += 4) {
            const int row = 2 * sp + hh;
            float a[TN], bb[TK];
#pragma unroll
            for (int i = 0; i < TN; ++i) a[i] = (i * 32 + r32 < N) ? Ys[row * N + i * 32 + r32] : 0.f;
#pragma unroll
            for (int j = 0; j < TK; ++j) bb[j] = (j * 32 + r32 < K) ? Xs[row * K + j * 32 + r32] : 0.f;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                bsum[i] += a[i];
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j] = mfma32(a[i], bb[j], acc[i][j]);
            }
        }
        if (b + 1 < nblk) lstore((b + 1) & 1);
        __syncthreads();
    }
    // the four waves' accumulators are summed through LDS, (w0 + w1) + w2) + w3, so that the workgroup writes ONE record:
    // [N * K] weight-gradient partial, then [N] bias-gradient partial (one reduce launch sums both over the workgroups)
    float* red = &sm[0][0];                               // [4 waves][16 registers][64 lanes]; the loop's last barrier freed the buffers
    float* pw = part + (int64_t)blockIdx.x * ((int64_t)N * K + N);
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int j = 0; j < TK; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[i][j][r];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = tid + u * 256, r = idx >> 6, l = idx & 63;
                const float v = ((red[idx] + red[1024 + idx]) + red[2048 + idx]) + red[3072 + idx];
                const int n = i * 32 + mfma_row(r, l), k = j * 32 + (l & 31);
                if (n < N && k < K) pw[(int64_t)n * K + k] = v;
            }
            __syncthreads();
        }
    }
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const float t = bsum[i] + xor32(bsum[i]);
            if (hh == 0) red[wave * 256 + i * 32 + r32] = t;
        }
        __syncthreads();
        if (tid < N) pw[(int64_t)N * K + tid] = ((red[tid] + red[256 + tid]) + red[512 + tid]) + red[768 + tid];
    }
}

struct FlatPlan { bool ok; int tn, tk, rb, blocks_per_wg, nwg; };
FlatPlan flat_plan(int M, int N, int K) {
    FlatPlan p;
    p.tn = (N + 31) / 32; p.tk = (K + 31) / 32;
    const int big = N > K ? N : K;
    p.ok = p.tn * p.tk <= 12 && big <= 256 && M >= 4096;
    p.rb = p.blocks_per_wg = p.nwg = 0;
    if (!p.ok) return p;
    int rb = (2048 / big) & ~7;
    if (rb > 512) rb = 512;
    p.rb = rb;
    const int64_t nblk = ((int64_t)M + rb - 1) / rb;
    int64_t nwg = 1024;                                       // four workgroups per CU
    int64_t pbytes = (int64_t)M * (N + K) / 2;                // the partials (one per workgroup; written and read once more) stay under an eighth of
    if (pbytes < (4ll << 20)) pbytes = 4ll << 20;             // the operand bytes, at least 4 MB
    const int64_t cap = pbytes / ((int64_t)N * K * 4);
    if (nwg > cap) nwg = cap;
    if (nwg < 1) nwg = 1;
    if (nwg > nblk) nwg = nblk;
    p.blocks_per_wg = (int)((nblk + nwg - 1) / nwg);
    p.nwg = (int)((nblk + p.blocks_per_wg - 1) / p.blocks_per_wg);
    return p;
}

struct SkinnyPlan { bool ok; int tn, tk, nwaves, rows_per_wave; };
SkinnyPlan skinny_plan(int M, int N, int K) {
    SkinnyPlan p;
    p.tn = (N + 31) / 32; p.tk = (K + 31) / 32;
    p.ok = (p.tn == 1 && (p.tk == 1 || p.tk == 2 || p.tk == 4 || p.tk == 8)) || (p.tk == 1 && (p.tn == 2 || p.tn == 4 || p.tn == 8)) ||
           (p.tn == 2 && p.tk == 2);
    if (p.tn == 1 && p.tk == 3) { p.tk = 4; p.ok = true; }
    if (p.tk == 1 && p.tn == 3) { p.tn = 4; p.ok = true; }
    if ((p.tn == 1 && p.tk > 4 && p.tk < 8)) { p.tk = 8; p.ok = true; }
    if ((p.tk == 1 && p.tn > 4 && p.tn < 8)) { p.tn = 8; p.ok = true; }
    p.ok = p.ok && M >= 4096;                                  // small M: the tiled kernel's split is enough
    int64_t waves = (M + 127) / 128;                           // >= 128 rows per wave; many short waves keep more loads in flight
    if (waves > 16384) waves = 16384;
    const int64_t cap = (64ll << 20) / ((int64_t)N * K * 4);   // partials stay under 64 MB
    if (waves > cap) waves = cap;
    if (waves < 4) waves = 4;
    waves = (waves + 3) / 4 * 4;
    p.nwaves = (int)waves;
    p.rows_per_wave = (int)(((M + waves - 1) / waves + 15) / 16 * 16);
    return p;
}

struct WgradPlan { int ntn, ntk, tiles, S, rows_per_split, chunks, rows_per_chunk; };
WgradPlan wgrad_plan(int M, int N, int K) {
    WgradPlan w;
    w.ntn = (N + WT - 1) / WT; w.ntk = (K + WT - 1) / WT; w.tiles = w.ntn * w.ntk;
    int S = (768 + w.tiles - 1) / w.tiles;                    // ~3 workgroups per CU in flight
    const int maxS = (M + 4 * RB - 1) / (4 * RB);             // at least 4 stages per slice
    if (S > maxS) S = maxS;
    // the S partial outputs are written and read once more: keep them under half of what the launch reads (a 128 x 128 gradient over 16384
    // rows ran with 768 slices = 50 MB of partials for 17 MB of operands, 93 us; profiles/r04_train_full_calls_after.txt)
    // - for outputs of at most 8 tiles only: the encoder-sized gradients (48 tiles x 16 slices) need every slice to fill the chip (capped: 197 -> 224 us)
    const int64_t capS = ((int64_t)M * (N + K)) / (2ll * N * K);
    if (w.tiles <= 8 && S > capS) S = (int)(capS < 4 ? 4 : capS);
    if (S > maxS) S = maxS;
    if (S > 1024) S = 1024;
    if (S < 1) S = 1;
    w.rows_per_split = (((M + S - 1) / S) + RB - 1) / RB * RB;
    if (w.rows_per_split < RB) w.rows_per_split = RB;          // M == 0
    w.S = (M + w.rows_per_split - 1) / w.rows_per_split;
    if (w.S < 1) w.S = 1;
    w.chunks = (M + 63) / 64; if (w.chunks > 256) w.chunks = 256; if (w.chunks < 1) w.chunks = 1;
    w.rows_per_chunk = (M + w.chunks - 1) / w.chunks;
    return w;
}

// ------------------------------------------------------------------------------------------------ LayerNorm backward
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ dy,
                                                            float* __restrict__ dx, float* __restrict__ dx_drop, float* __restrict__ part,
                                                            int64_t rows, int dim, float eps, float drop_p, uint64_t seed, uint32_t id) {
    __shared__ float red[4][2][64 * 4 * MAXV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const DropKey dk(dx_drop ? drop_p : 0.0f, seed, id);
    float4 gam[MAXV], dg[MAXV], db[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = (i * 64 + lane) * 4;
        gam[i] = (c < dim) ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float inv_dim = 1.0f / (float)dim;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float* xp = x + row * dim;
        const float* gp = dy + row * dim;
        float4 v[MAXV], g[MAXV];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * 4;
            const bool ok = c < dim;
            v[i] = ok ? *reinterpret_cast<const float4*>(xp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            g[i] = ok ? *reinterpret_cast<const float4*>(gp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
        const float mean = wave_sum(sum) * inv_dim;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < dim) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_dim + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;                 // xhat
            dg[i].x += g[i].x * v[i].x; dg[i].y += g[i].y * v[i].y; dg[i].z += g[i].z * v[i].z; dg[i].w += g[i].w * v[i].w;
            db[i].x += g[i].x; db[i].y += g[i].y; db[i].z += g[i].z; db[i].w += g[i].w;
            g[i].x *= gam[i].x; g[i].y *= gam[i].y; g[i].z *= gam[i].z; g[i].w *= gam[i].w;   // dy * gamma
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
        }
        s1 = wave_sum(s1) * inv_dim; s2 = wave_sum(s2) * inv_dim;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < dim) {
                float4 o;
                o.x = rstd * (g[i].x - s1 - v[i].x * s2); o.y = rstd * (g[i].y - s1 - v[i].y * s2);
                o.z = rstd * (g[i].z - s1 - v[i].z * s2); o.w = rstd * (g[i].w - s1 - v[i].w * s2);
                *reinterpret_cast<float4*>(dx + row * dim + c) = o;
                if (dx_drop) {
                    const uint32_t ro = (uint32_t)row, co = (uint32_t)c;
                    o.x *= dk(ro, co); o.y *= dk(ro, co + 1); o.z *= dk(ro, co + 2); o.w *= dk(ro, co + 3);
                    *reinterpret_cast<float4*>(dx_drop + row * dim + c) = o;
                }
            }
        }
    }
    // block partial (4 waves, fixed order) -> part[block][2][dim]
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        *reinterpret_cast<float4*>(&red[wave][0][(i * 64 + lane) * 4]) = dg[i];
        *reinterpret_cast<float4*>(&red[wave][1][(i * 64 + lane) * 4]) = db[i];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * dim; e += 256) {
        const int which = e / dim, c = e % dim;
        part[(int64_t)blockIdx.x * 2 * dim + e] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// any dim <= 1024 (not a multiple of 4): lane-strided scalar accesses, per-lane dgamma / dbeta for columns lane + 64 j
__global__ __launch_bounds__(256) void layernorm_bwd_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                    const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ part,
                                                                    int64_t rows, int dim, float eps) {
    constexpr int MAXJ = 16;
    __shared__ float red[4][2][64 * MAXJ];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[MAXJ], db[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) { dg[j] = 0.f; db[j] = 0.f; }
    const float inv_dim = 1.0f / (float)dim;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float* xp = x + row * dim;
        const float* gp = dy + row * dim;
        float sum = 0.f;
        for (int c = lane; c < dim; c += 64) sum += xp[c];
        const float mean = wave_sum(sum) * inv_dim;
        float sq = 0.f;
        for (int c = lane; c < dim; c += 64) { const float d = xp[c] - mean; sq += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(sq) * inv_dim + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int c = lane + 64 * j;
            if (c < dim) {
                const float xh = (xp[c] - mean) * rstd, g = gp[c];
                dg[j] += g * xh; db[j] += g;
                const float gg = g * gamma[c];
                s1 += gg; s2 += gg * xh;
            }
        }
        s1 = wave_sum(s1) * inv_dim; s2 = wave_sum(s2) * inv_dim;
        for (int c = lane; c < dim; c += 64) {
            const float xh = (xp[c] - mean) * rstd;
            dx[row * dim + c] = rstd * (gp[c] * gamma[c] - s1 - xh * s2);
        }
    }
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) { red[wave][0][lane + 64 * j] = dg[j]; red[wave][1][lane + 64 * j] = db[j]; }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * dim; e += 256) {
        const int which = e / dim, c = e % dim;
        part[(int64_t)blockIdx.x * 2 * dim + e] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

int ln_bwd_blocks(int64_t rows) {
    int64_t b = (rows + 15) / 16;               // >= 4 rows per wave
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------------------------------------ loss backward, AdamW
__global__ __launch_bounds__(256) void masked_mse_bwd_kernel(const float* __restrict__ target, const float* __restrict__ pred,
                                                             const uint8_t* __restrict__ mask, const float* __restrict__ dloss,
                                                             float* __restrict__ dpred, int L, int D) {
    __shared__ float cnt_s[4];
    const int b = blockIdx.y;
    float cnt = 0.f;
    for (int l = threadIdx.x; l < L; l += 256) cnt += (mask && mask[(int64_t)b * L + l]) ? 0.f : 1.f;
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) cnt_s[threadIdx.x >> 6] = cnt;
    __syncthreads();
    const float kept = (cnt_s[0] + cnt_s[1]) + (cnt_s[2] + cnt_s[3]);
    const float coef = 2.0f * dloss[b] / (kept * (float)D);
    const int64_t base = (int64_t)b * L * D;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)L * D; i += (int64_t)gridDim.x * 256) {
        const int l = (int)(i / D);
        const bool keep = !(mask && mask[(int64_t)b * L + l]);
        dpred[base + i] = keep ? coef * (pred[base + i] - target[base + i]) : 0.f;
    }
}

// out = (x + rowtab[row % period]) * act'(z) * keep(row, col): the small glue passes of the training graph
// (PositionalEncoding add + dropout, modules.py:43-45; activation / dropout backward of a stand-alone linear)
__global__ __launch_bounds__(256) void rowop_kernel(const float* __restrict__ x, const float* __restrict__ rowtab, int period,
                                                    const float* __restrict__ z, int act, float* __restrict__ out, int64_t rows, int cols,
                                                    float drop_p, uint64_t seed, uint32_t id) {
    const DropKey dk(drop_p, seed, id);
    const int64_t n = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        float v = x[i];
        if (rowtab) v += rowtab[(r % period) * cols + c];
        if (z) v *= act_grad(z[i], act);
        if (drop_p > 0.0f) v *= dk((uint32_t)r, (uint32_t)c);
        out[i] = v;
    }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
        p[i] = pi;
    }
}

// one launch for every parameter tensor: blockIdx.y = tensor, blockIdx.x = 16K-element chunk of it
__global__ __launch_bounds__(256) void adamw_multi_kernel(const afm_adamw_tensor* __restrict__ tab, float lr, float b1, float b2, float eps, float wd,
                                                          float bc1, float bc2_sqrt) {
    const afm_adamw_tensor t = tab[blockIdx.y];
    const int64_t lo = (int64_t)blockIdx.x * 16384, hi = min(t.n, lo + 16384);
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const float gi = t.g[i];
        float pi = t.p[i] * (1.0f - lr * wd);
        const float mi = b1 * t.m[i] + (1.0f - b1) * gi;
        const float vi = b2 * t.v[i] + (1.0f - b2) * gi * gi;
        t.m[i] = mi; t.v[i] = vi;
        pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
        t.p[i] = pi;
    }
}

}  // namespace

extern "C" int afm_adamw_multi(const afm_adamw_tensor* d_table, int32_t n_tensors, int64_t max_n, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int32_t step, void* stream) {
    if (n_tensors == 0) return 0;
    if (!d_table || n_tensors < 0 || n_tensors > 65535 || max_n <= 0 || step < 1) return AFM_E_BADARG;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = sqrtf(1.0f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)((max_n + 16383) / 16384), n_tensors), dim3(256), 0, (hipStream_t)stream, d_table, lr, beta1,
                       beta2, eps, weight_decay, bc1, bc2);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_transpose(const float* in, float* out, int32_t rows, int32_t cols, void* stream) {
    if (rows == 0 || cols == 0) return 0;
    if (!in || !out || rows < 0 || cols < 0) return AFM_E_BADARG;
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, (hipStream_t)stream, in, out, rows, cols);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t afm_linear_wgrad_workspace_bytes(int32_t M, int32_t N, int32_t K) {
    if (M < 0 || N <= 0 || K <= 0) return AFM_E_BADARG;
    const WgradPlan w = wgrad_plan(M, N, K);
    const SkinnyPlan sp = skinny_plan(M, N, K);
    const int64_t tiled = (int64_t)w.S * N * K + (int64_t)w.chunks * N;
    const int64_t skinny = sp.ok ? (int64_t)sp.nwaves * ((int64_t)N * K + N) : 0;
    const FlatPlan fp = flat_plan(M, N, K);
    const int64_t flat = fp.ok ? (int64_t)fp.nwg * ((int64_t)N * K + N) : 0;
    int64_t need = tiled > skinny ? tiled : skinny;
    if (flat > need) need = flat;
    return need * (int64_t)sizeof(float);
}

extern "C" int afm_linear_wgrad(const afm_linear_wgrad_args* args, void* stream) {
    if (!args) return AFM_E_BADARG;
    const afm_linear_wgrad_args& a = *args;
    if (a.N <= 0 || a.K <= 0 || a.M < 0 || !a.dW) return AFM_E_BADARG;
    if (a.M > 0 && (!a.dY || !a.X)) return AFM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (a.M == 0) {                                           // empty batch: the gradient is zero
        if (!a.accumulate) {
            hipError_t e = hipMemset2DAsync(a.dW, a.lddw * sizeof(float), 0, (size_t)a.K * sizeof(float), a.N, s);
            if (e != hipSuccess) return (int)e;
            if (a.db) { e = hipMemsetAsync(a.db, 0, (size_t)a.N * sizeof(float), s); if (e != hipSuccess) return (int)e; }
        }
        return 0;
    }
    const FlatPlan fp = flat_plan(a.M, a.N, a.K);
    if (fp.ok && !a.dy_grp && !a.x_grp && a.lddw == a.K && a.lddy == a.N && a.ldx == a.K && ((((uintptr_t)a.dY) | ((uintptr_t)a.X)) & 15) == 0) {
        const int nparts = fp.nwg;
        if (!a.ws || a.ws_bytes < (int64_t)nparts * ((int64_t)a.N * a.K + a.N) * (int64_t)sizeof(float)) return AFM_E_WORKSPACE;
        float* pw = (float*)a.ws;
        {
            AfmProf prof(AFM_PROF_WGRAD_SKINNY, 4.0 * a.M * (a.N + a.K), s);          // work = bytes streamed
#define AFM_FL(TN_, TK_) hipLaunchKernelGGL((wgrad_flat_kernel<TN_, TK_>), dim3(fp.nwg), dim3(256), 0, s, a.dY, a.X, a.M, a.N, a.K, fp.rb, fp.blocks_per_wg, pw, a.db ? 1 : 0)
#define AFM_FLK(TN_, MAXTK_)                                                        \
    case TN_:                                                                       \
        switch (fp.tk) {                                                            \
            case 1: AFM_FL(TN_, 1); break;                                          \
            case 2: if (MAXTK_ >= 2) AFM_FL(TN_, (MAXTK_ >= 2 ? 2 : 1)); break;     \
            case 3: if (MAXTK_ >= 3) AFM_FL(TN_, (MAXTK_ >= 3 ? 3 : 1)); break;     \
            case 4: if (MAXTK_ >= 4) AFM_FL(TN_, (MAXTK_ >= 4 ? 4 : 1)); break;     \
            case 5: if (MAXTK_ >= 5) AFM_FL(TN_, (MAXTK_ >= 5 ? 5 : 1)); break;     \
            case 6: if (MAXTK_ >= 6) AFM_FL(TN_, (MAXTK_ >= 6 ? 6 : 1)); break;     \
            case 7: if (MAXTK_ >= 7) AFM_FL(TN_, (MAXTK_ >= 7 ? 7 : 1)); break;     \
            case 8: if (MAXTK_ >= 8) AFM_FL(TN_, (MAXTK_ >= 8 ? 8 : 1)); break;     \
            default: return AFM_E_UNSUPPORTED;                                      \
        }                                                                           \
        break;
            switch (fp.tn) {                                  // tn * tk <= 12 (flat_plan)
                AFM_FLK(1, 8) AFM_FLK(2, 6) AFM_FLK(3, 4) AFM_FLK(4, 3) AFM_FLK(5, 2) AFM_FLK(6, 2) AFM_FLK(7, 1) AFM_FLK(8, 1)
                default: return AFM_E_UNSUPPORTED;
            }
#undef AFM_FLK
#undef AFM_FL
            AFM_CHECK_LAUNCH();
        }
        AfmProf prof(AFM_PROF_TRAIN_MISC, 0.0, s);
        const int nk = a.N * a.K, ntot = nk + (a.db ? a.N : 0);
        if (ntot <= 16)
            hipLaunchKernelGGL(reduce_cols_kernel<16>, dim3(1), dim3(1024), 0, s, pw, (int64_t)nk + a.N, nparts, a.dW, a.db, nk, ntot, a.accumulate);
        else
            hipLaunchKernelGGL(reduce_cols_kernel<64>, dim3((ntot + 63) / 64), dim3(1024), 0, s, pw, (int64_t)nk + a.N, nparts, a.dW, a.db, nk, ntot, a.accumulate);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    const SkinnyPlan sp = skinny_plan(a.M, a.N, a.K);
    if (sp.ok && !a.dy_grp && !a.x_grp && a.lddw == a.K) {
        if (!a.ws || a.ws_bytes < (int64_t)sp.nwaves * ((int64_t)a.N * a.K + a.N) * (int64_t)sizeof(float)) return AFM_E_WORKSPACE;
        float* pw = (float*)a.ws;
        float* pb = a.db ? pw + (int64_t)sp.nwaves * a.N * a.K : nullptr;
        {
            AfmProf prof(AFM_PROF_WGRAD_SKINNY, 4.0 * a.M * (a.N + a.K), s);          // work = bytes streamed
#define AFM_SK(TN_, TK_) hipLaunchKernelGGL((wgrad_skinny_kernel<TN_, TK_>), dim3(sp.nwaves / 4), dim3(256), 0, s, a.dY, a.lddy, a.X, a.ldx, a.M, a.N, a.K, sp.rows_per_wave, pw, pb)
            if (sp.tn == 1 && sp.tk == 1) AFM_SK(1, 1); else if (sp.tn == 1 && sp.tk == 2) AFM_SK(1, 2); else if (sp.tn == 1 && sp.tk == 4) AFM_SK(1, 4);
            else if (sp.tn == 1 && sp.tk == 8) AFM_SK(1, 8); else if (sp.tn == 2 && sp.tk == 1) AFM_SK(2, 1); else if (sp.tn == 4 && sp.tk == 1) AFM_SK(4, 1);
            else if (sp.tn == 8 && sp.tk == 1) AFM_SK(8, 1); else AFM_SK(2, 2);
#undef AFM_SK
            AFM_CHECK_LAUNCH();
        }
        AfmProf prof(AFM_PROF_TRAIN_MISC, 0.0, s);
        const int nk = a.N * a.K;
        hipLaunchKernelGGL(reduce_cols_kernel<64>, dim3((nk + 63) / 64), dim3(1024), 0, s, pw, (int64_t)nk, sp.nwaves, a.dW, a.dW, nk, nk, a.accumulate);
        if (a.db) hipLaunchKernelGGL(reduce_cols_kernel<64>, dim3((a.N + 63) / 64), dim3(1024), 0, s, pb, (int64_t)a.N, sp.nwaves, a.db, a.db, a.N, a.N, a.accumulate);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    const WgradPlan w = wgrad_plan(a.M, a.N, a.K);
    const int64_t need = ((int64_t)w.S * a.N * a.K + (int64_t)w.chunks * a.N) * (int64_t)sizeof(float);
    if (!a.ws || a.ws_bytes < need) return AFM_E_WORKSPACE;
    float* part = (float*)a.ws;
    float* colpart = part + (int64_t)w.S * a.N * a.K;
    const bool vec = (a.N % 4 == 0) && (a.K % 4 == 0) && (a.lddy % 4 == 0) && (a.ldx % 4 == 0) &&
                     (((uintptr_t)a.dY | (uintptr_t)a.X) & 15) == 0;
    const bool direct = w.S == 1 && !a.accumulate;
    float* out = direct ? a.dW : part;
    const int64_t out_ld = direct ? a.lddw : a.K;
    {
        AfmProf prof(AFM_PROF_WGRAD, 2.0 * a.M * a.N * a.K, s);
        if (vec) hipLaunchKernelGGL(wgrad_kernel<true>, dim3(w.tiles * w.S), dim3(256), 0, s, a, out, out_ld, (int64_t)a.N * a.K, w.ntk, w.tiles, w.rows_per_split);
        else hipLaunchKernelGGL(wgrad_kernel<false>, dim3(w.tiles * w.S), dim3(256), 0, s, a, out, out_ld, (int64_t)a.N * a.K, w.ntk, w.tiles, w.rows_per_split);
        AFM_CHECK_LAUNCH();
    }
    AfmProf prof(AFM_PROF_TRAIN_MISC, 0.0, s);
    if (!direct) {
        const int64_t n = (int64_t)a.N * a.K;
        hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, s, part,
                           (int64_t)a.N * a.K, w.S, a.dW, a.lddw, a.N, a.K, a.accumulate);
        AFM_CHECK_LAUNCH();
    }
    if (a.db) {
        hipLaunchKernelGGL(colsum_kernel, dim3((a.N + 63) / 64, w.chunks), dim3(256), 0, s, a.dY, a.lddy, RowMap3{a.dy_grp, a.dy_stride, a.dy_off},
                           a.M, a.N, w.rows_per_chunk, colpart);
        AFM_CHECK_LAUNCH();
        hipLaunchKernelGGL(reduce_cols_kernel<64>, dim3((a.N + 63) / 64), dim3(1024), 0, s, colpart, (int64_t)a.N, w.chunks, a.db, a.db, a.N, a.N,
                           a.accumulate);
        AFM_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int64_t afm_layernorm_bwd_workspace_bytes(int64_t rows, int32_t dim) {
    if (rows < 0 || dim <= 0) return AFM_E_BADARG;
    return (int64_t)ln_bwd_blocks(rows) * 2 * dim * (int64_t)sizeof(float);
}

extern "C" int afm_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dx_drop, float* dgamma, float* dbeta,
                                 int64_t rows, int32_t dim, float eps, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* ws,
                                 int64_t ws_bytes, void* stream) {
    if (dim <= 0 || dim > 1024) return dim > 1024 ? AFM_E_UNSUPPORTED : AFM_E_BADARG;
    if (!dgamma || !dbeta || rows < 0) return AFM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (rows == 0) {
        hipError_t e = hipMemsetAsync(dgamma, 0, (size_t)dim * sizeof(float), s);
        if (e == hipSuccess) e = hipMemsetAsync(dbeta, 0, (size_t)dim * sizeof(float), s);
        return (int)e;
    }
    if (!x || !gamma || !dy || !dx) return AFM_E_BADARG;
    if (drop_p < 0.0f || drop_p >= 1.0f) return AFM_E_BADARG;
    const int nb = ln_bwd_blocks(rows);
    if (!ws || ws_bytes < (int64_t)nb * 2 * dim * (int64_t)sizeof(float)) return AFM_E_WORKSPACE;
    float* part = (float*)ws;
    const bool generic = (dim & 3) || ((((uintptr_t)x) | ((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)dx_drop) | ((uintptr_t)gamma)) & 15);
    if (generic) {
        if (dx_drop && drop_p > 0.0f) return AFM_E_UNSUPPORTED;
        AfmProf prof(AFM_PROF_LN_BWD, 16.0 * rows * dim, s);
        hipLaunchKernelGGL(layernorm_bwd_generic_kernel, dim3(nb), dim3(256), 0, s, x, gamma, dy, dx, part, rows, dim, eps);
        AFM_CHECK_LAUNCH();
    } else {
    AfmProf prof(AFM_PROF_LN_BWD, 16.0 * rows * dim, s);
#define AFM_LNB(V) hipLaunchKernelGGL(layernorm_bwd_kernel<V>, dim3(nb), dim3(256), 0, s, x, gamma, dy, dx, dx_drop, part, rows, dim, eps, drop_p, drop_seed, drop_id)
    if (dim <= 256) AFM_LNB(1); else if (dim <= 512) AFM_LNB(2); else AFM_LNB(4);
#undef AFM_LNB
    AFM_CHECK_LAUNCH();
    }
    AfmProf prof(AFM_PROF_TRAIN_MISC, 0.0, s);
    hipLaunchKernelGGL(reduce_cols_kernel<64>, dim3((2 * dim + 63) / 64), dim3(1024), 0, s, part, (int64_t)2 * dim, nb, dgamma, dbeta, dim, 2 * dim, 0);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_masked_mse_bwd(const float* target, const float* pred, const uint8_t* frame_mask, const float* dloss, float* dpred,
                                  int32_t B, int32_t L, int32_t D, void* stream) {
    if (B == 0) return 0;
    if (!target || !pred || !dloss || !dpred || B < 0 || L <= 0 || D <= 0) return AFM_E_BADARG;
    int gx = (int)(((int64_t)L * D + 255) / 256); if (gx > 64) gx = 64;
    hipLaunchKernelGGL(masked_mse_bwd_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, target, pred, frame_mask, dloss, dpred, L, D);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_rowop(const float* x, const float* rowtab, int32_t period, const float* z, int32_t act, float* out, int64_t rows,
                         int32_t cols, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* stream) {
    if (rows == 0) return 0;
    if (!x || !out || rows < 0 || cols <= 0 || (rowtab && period <= 0) || drop_p < 0.0f || drop_p >= 1.0f) return AFM_E_BADARG;
    int64_t gx = (rows * cols + 255) / 256; if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(rowop_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, x, rowtab, period, z, act, out, rows, cols, drop_p,
                       drop_seed, drop_id);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int32_t step, void* stream) {
    if (n == 0) return 0;
    if (!p || !g || !m || !v || n < 0 || step < 1) return AFM_E_BADARG;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = sqrtf(1.0f - powf(beta2, (float)step));
    int64_t gx = (n + 255) / 256; if (gx > 4096) gx = 4096;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2);
    AFM_CHECK_LAUNCH();
    return 0;
}
