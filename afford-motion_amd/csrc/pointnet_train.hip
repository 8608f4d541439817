// Train-mode point-cloud operators: BatchNorm on BATCH statistics breaks the single-kernel fusions of pointnet.hip
// (every BatchNorm needs a full pass over its input before anything downstream can be computed), so the training
// forward / backward of TransitionDown and PointTransformerLayer (pointtransformer.py:26-69) is composed from
// bandwidth-bound passes over materialised [n, k, c] tensors - 288 GB of HBM hold them comfortably:
//   column statistics (two-stage, fixed order), per-column affine (+ residual, ReLU), BatchNorm backward,
//   neighbour gather / atomic scatter-add, group max (+ argmax), and the vector-attention glue
//   (w = k_g - q + p_r;  softmax over the k neighbours;  share-planes weighted sum) with their backward passes.
// All of them stream rows with the channel index on consecutive lanes (coalesced), one thread per element group.
#include "common.h"
#include "profile.h"
#include <math.h>

namespace {

// dst[c] (+)= sum_s src[s * stride + c]; block = 64 columns x 16 partial-row lanes, fixed-order tree
__global__ __launch_bounds__(1024) void reduce_cols2_kernel(const float* __restrict__ src, int64_t stride, int S, float* __restrict__ dst, int n) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float a0 = 0.f, a1 = 0.f;
    if (c < n) {
        int s = ty;
        for (; s + 16 < S; s += 32) { a0 += src[(int64_t)s * stride + c]; a1 += src[(int64_t)(s + 16) * stride + c]; }
        for (; s < S; s += 16) a0 += src[(int64_t)s * stride + c];
    }
    red[ty][tx] = a0 + a1;
    __syncthreads();
    if (ty == 0 && c < n) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += red[i][tx];
        dst[c] = v;
    }
}

// Per-column pair sums over a chunk of rows.  MODE 0: (x, x^2).  MODE 1: (g, g * xhat) with g = dy * (y > 0 when y given),
// xhat = (x - mean) * rstd.  Block = CW column lanes x (256 / CW) row lanes; thread keeps NC = ceil(C / CW) column pairs.
template <int MODE, int NC>
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ y,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd, int64_t rows, int C, int CW,
                                                       int64_t rows_per_chunk, float* __restrict__ part) {
    extern __shared__ float red[];                       // [RL][2][C]
    const int tx = threadIdx.x % CW, ty = threadIdx.x / CW, RL = 256 / CW;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float s0[NC], s1[NC], mu[NC], rs[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        s0[j] = 0.f; s1[j] = 0.f;
        const int c = tx + j * CW;
        // MODE 0: sums are taken about the tensor's first row (shifted-data variance: no E[x^2] - m^2 cancellation)
        mu[j] = c < C ? (MODE == 1 ? mean[c] : a[c]) : 0.f;
        rs[j] = (MODE == 1 && c < C) ? rstd[c] : 0.f;
    }
    for (int64_t r = r0 + ty; r < r1; r += RL) {
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            const int c = tx + j * CW;
            if (c < C) {
                const int64_t i = r * C + c;
                if (MODE == 0) {
                    const float v = a[i] - mu[j];
                    s0[j] += v; s1[j] += v * v;
                } else {
                    float g = a[i];
                    if (y) g = y[i] > 0.0f ? g : 0.0f;
                    s0[j] += g; s1[j] += g * ((x[i] - mu[j]) * rs[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int c = tx + j * CW;
        if (c < C) { red[(ty * 2 + 0) * C + c] = s0[j]; red[(ty * 2 + 1) * C + c] = s1[j]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        float v = 0.f;
        for (int t = 0; t < RL; ++t) v += red[t * 2 * C + e];
        part[(int64_t)blockIdx.x * 2 * C + e] = v;
    }
}

// The same sums with 16-byte lanes for C a power of two (every BatchNorm of the point transformer: 32 .. 512 channels): a thread owns four
// consecutive columns and every 256 / (C / 4)-th row of the chunk, and the loads of UNR rows are issued before the first is consumed - the
// scalar kernel above has one 4-byte load per thread in flight and ran at 0.5 - 1.6 TB/s (profiles/r04_train_full_calls_before.txt).
// Rows are added in ascending order per (column, row lane), the row lanes in a fixed LDS order: deterministic, independent of the launch.
template <int MODE>
__global__ __launch_bounds__(256) void colstats4_kernel(const float* __restrict__ a, const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, int64_t rows, int C,
                                                        int64_t rows_per_chunk, float* __restrict__ part) {
    extern __shared__ float red[];                       // [RL][2][C]
    constexpr int UNR = MODE == 0 ? 4 : 2;
    const int QW = C >> 2, tx = threadIdx.x % QW, ty = threadIdx.x / QW, RL = 256 / QW;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    // MODE 0: sums are taken about the tensor's first row (shifted-data variance: no E[x^2] - m^2 cancellation)
    const float4 mu = *reinterpret_cast<const float4*>((MODE == 1 ? mean : a) + tx * 4);
    const float4 rs = MODE == 1 ? *reinterpret_cast<const float4*>(rstd + tx * 4) : s0;
    auto add = [&](const float4& av, const float4& xv, const float4& yv) {
        if (MODE == 0) {
            const float vx = av.x - mu.x, vy = av.y - mu.y, vz = av.z - mu.z, vw = av.w - mu.w;
            s0.x += vx; s0.y += vy; s0.z += vz; s0.w += vw;
            s1.x += vx * vx; s1.y += vy * vy; s1.z += vz * vz; s1.w += vw * vw;
        } else {
            float4 g = av;
            if (y) { g.x = yv.x > 0.0f ? g.x : 0.0f; g.y = yv.y > 0.0f ? g.y : 0.0f; g.z = yv.z > 0.0f ? g.z : 0.0f; g.w = yv.w > 0.0f ? g.w : 0.0f; }
            s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
            s1.x += g.x * ((xv.x - mu.x) * rs.x); s1.y += g.y * ((xv.y - mu.y) * rs.y);
            s1.z += g.z * ((xv.z - mu.z) * rs.z); s1.w += g.w * ((xv.w - mu.w) * rs.w);
        }
    };
    int64_t r = r0 + ty;
    for (; r + (int64_t)(UNR - 1) * RL < r1; r += (int64_t)UNR * RL) {
        float4 av[UNR], xv[UNR], yv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t i = (r + (int64_t)u * RL) * C + tx * 4;
            av[u] = *reinterpret_cast<const float4*>(a + i);
            if (MODE == 1) { xv[u] = *reinterpret_cast<const float4*>(x + i); if (y) yv[u] = *reinterpret_cast<const float4*>(y + i); }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) add(av[u], xv[u], yv[u]);
    }
    for (; r < r1; r += RL) {
        const int64_t i = r * C + tx * 4;
        const float4 av = *reinterpret_cast<const float4*>(a + i);
        float4 xv = av, yv = av;
        if (MODE == 1) { xv = *reinterpret_cast<const float4*>(x + i); if (y) yv = *reinterpret_cast<const float4*>(y + i); }
        add(av, xv, yv);
    }
    *reinterpret_cast<float4*>(&red[(ty * 2 + 0) * C + tx * 4]) = s0;
    *reinterpret_cast<float4*>(&red[(ty * 2 + 1) * C + tx * 4]) = s1;
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        float v = 0.f;
        for (int t = 0; t < RL; ++t) v += red[t * 2 * C + e];
        part[(int64_t)blockIdx.x * 2 * C + e] = v;
    }
}

struct StatPlan { int CW, NC, chunks; int64_t rows_per_chunk; };
StatPlan stat_plan(int64_t rows, int C) {
    StatPlan p;
    p.CW = 4; while (p.CW < C && p.CW < 64) p.CW <<= 1;
    p.NC = (C + p.CW - 1) / p.CW;
    const int RL = 256 / p.CW;
    int64_t ch = (rows + (int64_t)RL * 16 - 1) / ((int64_t)RL * 16);      // >= 16 rows per row lane
    if (ch > 1024) ch = 1024;
    if (ch < 1) ch = 1;
    p.chunks = (int)ch;
    p.rows_per_chunk = (rows + ch - 1) / ch;
    return p;
}

template <int MODE>
int launch_colstats(const float* a, const float* x, const float* y, const float* mean, const float* rstd, int64_t rows, int C, float* stats,
                    void* ws, int64_t ws_bytes, hipStream_t s) {
    if (C <= 0 || C > 512 || rows < 0) return C > 512 ? AFM_E_UNSUPPORTED : AFM_E_BADARG;
    if (rows == 0) { hipError_t e = hipMemsetAsync(stats, 0, 2 * C * sizeof(float), s); return (int)e; }
    const StatPlan p = stat_plan(rows, C);
    if (!ws || ws_bytes < (int64_t)p.chunks * 2 * C * (int64_t)sizeof(float)) return AFM_E_WORKSPACE;
    float* part = (float*)ws;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, s);
    const uintptr_t al = (uintptr_t)a | (uintptr_t)x | (uintptr_t)y | (uintptr_t)mean | (uintptr_t)rstd;
    if (C >= 16 && C <= 512 && (C & (C - 1)) == 0 && (al & 15) == 0) {          // 16-byte lanes (the choice depends on C and alignment only)
        const int RL = 1024 / C;
        int64_t ch = (rows + (int64_t)RL * 32 - 1) / ((int64_t)RL * 32);        // >= 32 rows per row lane
        if (ch > 2048) ch = 2048;
        if (ch > p.chunks) ch = p.chunks;                                       // (the workspace is sized by stat_plan)
        if (ch < 1) ch = 1;
        const int64_t rpc = (rows + ch - 1) / ch;
        hipLaunchKernelGGL((colstats4_kernel<MODE>), dim3((unsigned)ch), dim3(256), (size_t)RL * 2 * C * sizeof(float), s, a, x, y, mean, rstd, rows, C, rpc, part);
        AFM_CHECK_LAUNCH();
        hipLaunchKernelGGL(reduce_cols2_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, s, part, (int64_t)2 * C, (int)ch, stats, 2 * C);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = (size_t)(256 / p.CW) * 2 * C * sizeof(float);
#define AFM_CS(NC) hipLaunchKernelGGL((colstats_kernel<MODE, NC>), dim3(p.chunks), dim3(256), lds, s, a, x, y, mean, rstd, rows, C, p.CW, p.rows_per_chunk, part)
    if (p.NC == 1) AFM_CS(1); else if (p.NC == 2) AFM_CS(2); else if (p.NC <= 4) AFM_CS(4); else AFM_CS(8);
#undef AFM_CS
    AFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(reduce_cols2_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, s, part, (int64_t)2 * C, p.chunks, stats, 2 * C);
    AFM_CHECK_LAUNCH();
    return 0;
}

// stats [world][3C] per rank: sum(x - K), sum((x - K)^2), K (= that rank's first row); equal row counts per rank.
// Per-rank mean / M2 are combined with the parallel-variance formula (as torch's batch_norm_gather_stats_with_counts).
__global__ void bn_finalize_kernel(const float* __restrict__ stats, int world, float rows, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                   float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float msum = 0.f;
    for (int r = 0; r < world; ++r) msum += stats[(int64_t)r * 3 * C + 2 * C + c] + stats[(int64_t)r * 3 * C + c] / rows;
    const float m = msum / (float)world;
    float m2 = 0.f;
    for (int r = 0; r < world; ++r) {
        const float s1 = stats[(int64_t)r * 3 * C + c], s2 = stats[(int64_t)r * 3 * C + C + c], k = stats[(int64_t)r * 3 * C + 2 * C + c];
        const float mr = k + s1 / rows;
        m2 += fmaxf(s2 - s1 * s1 / rows, 0.0f) + rows * (mr - m) * (mr - m);
    }
    const float count = rows * (float)world;
    const float var = m2 / count;
    const float r = 1.0f / sqrtf(var + eps);
    mean[c] = m; rstd[c] = r;
    const float sc = gamma[c] * r;
    scale[c] = sc; shift[c] = beta[c] - m * sc;
    if (running_mean) {                      // nn.BatchNorm1d: running = (1 - momentum) * running + momentum * batch (unbiased variance)
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * m;
        const float unbiased = count > 1.0f ? m2 / (count - 1.0f) : var;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
    }
}

__global__ void copy_row_kernel(const float* __restrict__ src, float* __restrict__ dst, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) dst[c] = src[c];
}

// y = relu?(x * scale[c] + shift[c] + residual)
__global__ __launch_bounds__(256) void colaffine_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ res, int relu, float* __restrict__ y, int64_t n, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float v = x[i] * scale[c] + shift[c];
        if (res) v += res[i];
        y[i] = relu ? fmaxf(v, 0.0f) : v;
    }
}

// dx = gamma * rstd * (g - sum_g / count - xhat * sum_gx / count);  dres = g  (g = dy masked by the ReLU output)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ stats, float inv_count, float* __restrict__ dx,
                                                           float* __restrict__ dres, int64_t n, int C) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        float g = dy[i];
        if (y) g = y[i] > 0.0f ? g : 0.0f;
        const float xh = (x[i] - mean[c]) * rstd[c];
        dx[i] = gamma[c] * rstd[c] * (g - stats[c] * inv_count - xh * stats[C + c] * inv_count);
        if (dres) dres[i] = g;
    }
}

// out[r, 0:3] = xyz[idx[r]] - new_xyz[r / k];  out[r, 3:3+C] = feat[idx[r]]   (queryandgroup, pointops.py:79-100)
__global__ __launch_bounds__(256) void group_points_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz, const float* __restrict__ feat,
                                                           const int32_t* __restrict__ idx, float* __restrict__ out, int64_t rows, int k, int C) {
    const int W = 3 + C;
    const int64_t n = rows * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / W;
        const int c = (int)(i - r * W);
        const int64_t src = idx[r];
        out[i] = c < 3 ? xyz[src * 3 + c] - new_xyz[(r / k) * 3 + c] : feat[src * C + (c - 3)];
    }
}

// dst[idx[r], c] += src[r * ld + off + c]   (backward of a row gather; f32 atomics, as the reference's CUDA grouping backward)
__global__ __launch_bounds__(256) void scatter_add_kernel(const float* __restrict__ src, int64_t ld, int off, const int32_t* __restrict__ idx,
                                                          float* __restrict__ dst, int64_t rows, int C) {
    const int64_t n = rows * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / C;
        const int c = (int)(i - r * C);
        atomicAdd(dst + (int64_t)idx[r] * C + c, src[r * ld + off + c]);
    }
}

// ---- deterministic scatter-add (round 6): the f32 atomics above add in arrival order (last-bit differences run to run).  The index lists of
// this path (kNN neighbour lists, the 3-NN of the interpolation) are fixed for a whole training step and used by several backward operators,
// so their INVERSE is built once - for every destination row the list of entries that point at it, ASCENDING - and every scatter-add becomes a
// segmented sum in that fixed order, one thread per (destination row, channel), no atomics on floats, no zero-fill of the destination.
// Plan = seg_off [n_dst + 1] (exclusive prefix of the per-destination counts) | seg_ent [entries] (the entries of each destination, ascending).
// Built with integer atomics only (their results do not depend on the order): count -> scan -> fill in arrival order -> rank inside the segment.
__global__ __launch_bounds__(256) void plan_count_kernel(const int32_t* __restrict__ idx, int64_t entries, int64_t n_dst, int32_t* __restrict__ cnt) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < entries; e += (int64_t)gridDim.x * 256) {
        const int32_t d = idx[e];
        if (d >= 0 && d < n_dst) atomicAdd(cnt + d, 1);
    }
}
// exclusive prefix sum of cnt [n] -> off [n + 1] in three small launches (a single workgroup walking 262144 counters serially cost 250 us per index
// list - 2 ms of a training step): (a) sums of blocks of 1024 counters, (b) their exclusive scan by one workgroup, (c) the scan inside every block
// plus its block offset.  cnt is left ZERO (the fill pass uses it as its cursors).
constexpr int PLAN_BLK = 1024;                            // counters per block (256 threads x 4)
__device__ __forceinline__ int32_t plan_block_scan(int32_t v, int32_t* lds, int* total) {      // exclusive scan of one value per thread over 256 threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int32_t x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int32_t y = __shfl_up(x, o); if (lane >= o) x += y; }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    int32_t base = 0;
    for (int w = 0; w < wave; ++w) base += lds[w];
    *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + x - v;
}
__global__ __launch_bounds__(256) void plan_blocksum_kernel(const int32_t* __restrict__ cnt, int64_t n, int32_t* __restrict__ bsum) {
    __shared__ int32_t lds[4];
    const int64_t i0 = (int64_t)blockIdx.x * PLAN_BLK + threadIdx.x * 4;
    int32_t s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) s += i0 + u < n ? cnt[i0 + u] : 0;
    int total;
    (void)plan_block_scan(s, lds, &total);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void plan_scan_blocks_kernel(int32_t* __restrict__ bsum, int64_t nblk, int32_t* __restrict__ off_last) {
    __shared__ int32_t lds[4];
    int32_t carry = 0;
    for (int64_t b0 = 0; b0 < nblk; b0 += 256) {              // (nblk <= 256 for every index list of this path: one trip)
        const int64_t b = b0 + threadIdx.x;
        const int32_t v = b < nblk ? bsum[b] : 0;
        int total;
        const int32_t ex = plan_block_scan(v, lds, &total);
        if (b < nblk) bsum[b] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *off_last = carry;                  // off[n]: the number of valid entries
}
__global__ __launch_bounds__(256) void plan_scan_kernel(int32_t* __restrict__ cnt, int64_t n, const int32_t* __restrict__ bsum, int32_t* __restrict__ off) {
    __shared__ int32_t lds[4];
    const int64_t i0 = (int64_t)blockIdx.x * PLAN_BLK + threadIdx.x * 4;
    int32_t c[4], s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { c[u] = i0 + u < n ? cnt[i0 + u] : 0; s += c[u]; }
    int total;
    int32_t run = bsum[blockIdx.x] + plan_block_scan(s, lds, &total);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (i0 + u < n) { off[i0 + u] = run; run += c[u]; cnt[i0 + u] = 0; }
}
__global__ __launch_bounds__(256) void plan_fill_kernel(const int32_t* __restrict__ idx, int64_t entries, int64_t n_dst, const int32_t* __restrict__ off,
                                                        int32_t* __restrict__ cursor, int32_t* __restrict__ tmp) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < entries; e += (int64_t)gridDim.x * 256) {
        const int32_t d = idx[e];
        if (d >= 0 && d < n_dst) tmp[off[d] + atomicAdd(cursor + d, 1)] = (int32_t)e;
    }
}
// entry at position p of its segment -> position (number of smaller entries of the segment): the segment in ascending order, whatever order
// the fill pass wrote it in.  One thread per entry; the threads of a segment read the same short list (kNN lists: ~k entries per destination)
__global__ __launch_bounds__(256) void plan_rank_kernel(const int32_t* __restrict__ idx, int64_t n_dst, const int32_t* __restrict__ off,
                                                        const int32_t* __restrict__ tmp, int32_t* __restrict__ ent) {
    const int64_t total = off[n_dst];
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
        const int32_t e = tmp[p], d = idx[e];
        const int32_t lo = off[d], hi = off[d + 1];
        int32_t rank = 0;
        for (int32_t q = lo; q < hi; ++q) rank += tmp[q] < e ? 1 : 0;
        ent[lo + rank] = e;
    }
}
// dst[d, c] = sum over the entries e of destination d, ascending, of weight(e) * src[(e / row_div) * ld + col_offset + c]
//   weight(e) = 1, or (d2 != NULL: the interpolation's backward) w_e / sum_j w_{row, j}, w = 1 / (sqrt(d2) + 1e-8), row = e / k - the expression of
//   interpolate_bwd_kernel, so that both forms add the same numbers
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ src, int64_t ld, int off_col, int row_div, const float* __restrict__ d2,
                                                          const int32_t* __restrict__ seg_off, const int32_t* __restrict__ seg_ent, float* __restrict__ dst,
                                                          int64_t n_dst, int C) {
    const int64_t n = n_dst * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t d = i / C;
        const int c = (int)(i - d * C);
        float acc = 0.f;
        const int32_t lo = seg_off[d], hi = seg_off[d + 1];
        for (int32_t q = lo; q < hi; ++q) {
            const int64_t e = seg_ent[q], r = e / row_div;
            float v = src[r * ld + off_col + c];
            if (d2) {
                float norm = 0.f, we = 0.f;
                for (int j = 0; j < row_div; ++j) { const float wj = 1.0f / (sqrtf(d2[r * row_div + j]) + 1e-8f); norm += wj; if (r * row_div + j == e) we = wj; }
                v = v * (we / norm);
            }
            acc += v;
        }
        dst[i] = acc;
    }
}

__global__ __launch_bounds__(256) void group_max_kernel(const float* __restrict__ x, float* __restrict__ y, int32_t* __restrict__ arg, int64_t m, int k, int C) {
    const int64_t n = m * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / C;
        const int c = (int)(i - g * C);
        float best = x[(g * k) * C + c];
        int bi = 0;
        for (int j = 1; j < k; ++j) {
            const float v = x[(g * k + j) * C + c];
            if (v > best) { best = v; bi = j; }             // first maximum wins, like torch max_pool1d
        }
        y[i] = best; arg[i] = bi;
    }
}

__global__ __launch_bounds__(256) void group_max_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ arg, float* __restrict__ dx, int64_t m, int k, int C) {
    const int64_t n = m * k * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t r = i / C;
        const int64_t g = r / k;
        const int j = (int)(r - g * k);
        dx[i] = arg[g * C + c] == j ? dy[g * C + c] : 0.0f;
    }
}

// out[g, c] = scale * sum_j x[g, j, c]
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t m, int k, int C, float scale) {
    const int64_t n = m * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / C;
        const int c = (int)(i - g * C);
        float s = 0.f;
        for (int j = 0; j < k; ++j) s += x[(g * k + j) * C + c];
        out[i] = scale * s;
    }
}

// w0[g, j, c] = kg[g, j, c] - q[g, c] + pr[g, j, c]   (pointtransformer.py:34)
__global__ __launch_bounds__(256) void pt_w0_kernel(const float* __restrict__ kg, const float* __restrict__ q, const float* __restrict__ pr,
                                                    float* __restrict__ out, int64_t m, int k, int C) {
    const int64_t n = m * k * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t g = i / ((int64_t)k * C);
        out[i] = kg[i] - q[g * C + c] + pr[i];
    }
}

// sw = softmax over the k neighbours of w2[g, :, j];  out[g, s*Cs + j] = sum_k (vg + pr)[g, k, s*Cs + j] * sw[g, k, j]
// (pointtransformer.py:35-37).  One thread per (g, channel c): recomputes its column's softmax (k <= 16 values).
__global__ __launch_bounds__(256) void pt_aggregate_kernel(const float* __restrict__ vg, const float* __restrict__ pr, const float* __restrict__ w2,
                                                           float* __restrict__ out, float* __restrict__ sw, int64_t m, int k, int C, int Cs) {
    const int64_t n = m * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / C;
        const int c = (int)(i - g * C), j = c % Cs;
        float mx = -INFINITY;
        for (int t = 0; t < k; ++t) mx = fmaxf(mx, w2[(g * k + t) * Cs + j]);
        float den = 0.f;
        for (int t = 0; t < k; ++t) den += __expf(w2[(g * k + t) * Cs + j] - mx);
        const float inv = 1.0f / den;
        float acc = 0.f;
        for (int t = 0; t < k; ++t) {
            const float p = __expf(w2[(g * k + t) * Cs + j] - mx) * inv;
            acc += (vg[(g * k + t) * C + c] + pr[(g * k + t) * C + c]) * p;
            if (c < Cs) sw[(g * k + t) * Cs + j] = p;
        }
        out[i] = acc;
    }
}

// da[g,t,c] = dout[g,c] * sw[g,t,c%Cs]  (gradient of both vg and pr);
// dsw[g,t,j] = sum_s dout[g, s*Cs+j] * a[g,t,s*Cs+j];  dw2 = sw * (dsw - sum_t sw*dsw)
__global__ __launch_bounds__(256) void pt_aggregate_bwd_a_kernel(const float* __restrict__ dout, const float* __restrict__ sw, float* __restrict__ da,
                                                                 int64_t m, int k, int C, int Cs) {
    const int64_t n = m * k * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const int64_t r = i / C;                 // g * k + t
        da[i] = dout[(r / k) * C + c] * sw[r * Cs + c % Cs];
    }
}
__global__ __launch_bounds__(256) void pt_aggregate_bwd_w_kernel(const float* __restrict__ vg, const float* __restrict__ pr, const float* __restrict__ sw,
                                                                 const float* __restrict__ dout, float* __restrict__ dw2, int64_t m, int k, int C, int Cs) {
    const int64_t n = m * Cs;
    const int S = C / Cs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t g = i / Cs;
        const int j = (int)(i - g * Cs);
        float dot = 0.f;
        float dsw[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            dsw[t] = 0.f;
            if (t < k) {
                float d = 0.f;
                for (int s = 0; s < S; ++s) {
                    const int64_t e = (g * k + t) * C + s * Cs + j;
                    d += dout[g * C + s * Cs + j] * (vg[e] + pr[e]);
                }
                dsw[t] = d;
                dot += d * sw[(g * k + t) * Cs + j];
            }
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
            if (t < k) dw2[(g * k + t) * Cs + j] = sw[(g * k + t) * Cs + j] * (dsw[t] - dot);
    }
}

// The forward aggregation staged the same way (see pt_aggregate_bwd_fused_kernel): the softmax of a (point, share-group) column is computed
// once (the thread-per-channel kernel above recomputes it for each of its C / Cs planes), arithmetic and order unchanged.
__global__ __launch_bounds__(256) void pt_aggregate_fused_kernel(const float* __restrict__ vg, const float* __restrict__ pr, const float* __restrict__ w2,
                                                                 float* __restrict__ out, float* __restrict__ sw, int64_t m, int k, int C, int Cs, int G) {
    extern __shared__ __align__(16) float sm[];
    const int kC = k * C, kCs = k * Cs, tid = threadIdx.x;
    float* As = sm;                       // [G][k][C]   vg + pr
    float* Ws = As + G * kC;              // [G][k][Cs]  w2, then the softmax
    for (int64_t g0 = (int64_t)blockIdx.x * G; g0 < m; g0 += (int64_t)gridDim.x * G) {
        const int ng = (int)(m - g0 < G ? m - g0 : G);
        __syncthreads();
        {
            const float4* v4 = reinterpret_cast<const float4*>(vg + g0 * kC);
            const float4* p4 = reinterpret_cast<const float4*>(pr + g0 * kC);
            for (int i = tid; i < ng * kC / 4; i += 256) {
                const float4 a = v4[i], b = p4[i];
                reinterpret_cast<float4*>(As)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            }
            const float4* w4 = reinterpret_cast<const float4*>(w2 + g0 * kCs);
            for (int i = tid; i < ng * kCs / 4; i += 256) reinterpret_cast<float4*>(Ws)[i] = w4[i];
        }
        __syncthreads();
        for (int i = tid; i < ng * Cs; i += 256) {          // one (point, share-group column) per thread
            const int g = i / Cs, j = i - g * Cs;
            float* w = Ws + g * kCs + j;
            float mx = -INFINITY;
            for (int t = 0; t < k; ++t) mx = fmaxf(mx, w[t * Cs]);
            float den = 0.f;
            for (int t = 0; t < k; ++t) den += __expf(w[t * Cs] - mx);
            const float inv = 1.0f / den;
            for (int t = 0; t < k; ++t) w[t * Cs] = __expf(w[t * Cs] - mx) * inv;
        }
        __syncthreads();
        {
            float4* s4 = reinterpret_cast<float4*>(sw + g0 * kCs);
            for (int i = tid; i < ng * kCs / 4; i += 256) s4[i] = reinterpret_cast<const float4*>(Ws)[i];
        }
        for (int i = tid; i < ng * C; i += 256) {
            const int g = i / C, c = i - g * C, j = c % Cs;
            const float* a = As + g * kC + c;
            const float* w = Ws + g * kCs + j;
            float acc = 0.f;
            for (int t = 0; t < k; ++t) acc += a[t * C] * w[t * Cs];
            out[g0 * C + i] = acc;
        }
    }
}

// Both gradients of the aggregation in ONE pass for the shapes of the point transformer (k * C a multiple of 4, C / Cs = share planes): a workgroup
// takes G consecutive points, fetches their (vg + pr) rows - G * k * C contiguous floats of each tensor - with 16-byte lanes into LDS, and writes
// da (contiguous again) and dw2 from there.  The two-kernel form above reads a point's rows with a 16-byte-per-kilobyte pattern from a thread
// per (point, share-group) and ran at ~1.5 TB/s of useful traffic (profiles/r04_train_full_calls_after.txt: 600 us for 0.9 GB).
__global__ __launch_bounds__(256) void pt_aggregate_bwd_fused_kernel(const float* __restrict__ vg, const float* __restrict__ pr, const float* __restrict__ sw,
                                                                     const float* __restrict__ dout, float* __restrict__ da, float* __restrict__ dw2,
                                                                     int64_t m, int k, int C, int Cs, int G) {
    extern __shared__ __align__(16) float sm[];
    const int kC = k * C, kCs = k * Cs, S = C / Cs, tid = threadIdx.x;
    float* As = sm;                       // [G][k][C]   vg + pr
    float* Ds = As + G * kC;              // [G][C]      dout
    float* Ws = Ds + G * C;               // [G][k][Cs]  sw
    float* Es = Ws + G * kCs;             // [G][k][Cs]  dsw
    for (int64_t g0 = (int64_t)blockIdx.x * G; g0 < m; g0 += (int64_t)gridDim.x * G) {
        const int ng = (int)(m - g0 < G ? m - g0 : G);
        __syncthreads();
        {
            const float4* v4 = reinterpret_cast<const float4*>(vg + g0 * kC);
            const float4* p4 = reinterpret_cast<const float4*>(pr + g0 * kC);
            for (int i = tid; i < ng * kC / 4; i += 256) {
                const float4 a = v4[i], b = p4[i];
                reinterpret_cast<float4*>(As)[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            }
            const float4* d4 = reinterpret_cast<const float4*>(dout + g0 * C);
            for (int i = tid; i < ng * C / 4; i += 256) reinterpret_cast<float4*>(Ds)[i] = d4[i];
            const float4* w4 = reinterpret_cast<const float4*>(sw + g0 * kCs);
            for (int i = tid; i < ng * kCs / 4; i += 256) reinterpret_cast<float4*>(Ws)[i] = w4[i];
        }
        __syncthreads();
        // da[g, t, c] = dout[g, c] * sw[g, t, c % Cs]: four consecutive channels per thread (Cs % 4 == 0 or Cs divides 4 is NOT assumed: per element)
        {
            float4* o4 = reinterpret_cast<float4*>(da + g0 * kC);
            for (int i = tid; i < ng * kC / 4; i += 256) {
                const int e = i * 4, g = e / kC, r = e - g * kC, t = r / C, c = r - t * C;
                const float* d = Ds + g * C + c;
                const float* w = Ws + (g * k + t) * Cs;
                o4[i] = make_float4(d[0] * w[c % Cs], d[1] * w[(c + 1) % Cs], d[2] * w[(c + 2) % Cs], d[3] * w[(c + 3) % Cs]);
            }
        }
        // dsw[g, t, j] = sum_s dout[g, s Cs + j] * a[g, t, s Cs + j]   (s ascending, as the two-kernel form)
        for (int i = tid; i < ng * kCs; i += 256) {
            const int g = i / kCs, r = i - g * kCs, t = r / Cs, j = r - t * Cs;
            const float* d = Ds + g * C + j;
            const float* a = As + (g * k + t) * C + j;
            float acc = 0.f;
            for (int s2 = 0; s2 < S; ++s2) acc += d[s2 * Cs] * a[s2 * Cs];
            Es[i] = acc;
        }
        __syncthreads();
        // dw2 = sw * (dsw - sum_t sw * dsw)   (t ascending)
        for (int i = tid; i < ng * kCs; i += 256) {
            const int g = i / kCs, r = i - g * kCs, j = r % Cs;
            float dot = 0.f;
            for (int t = 0; t < k; ++t) dot += Es[(g * k + t) * Cs + j] * Ws[(g * k + t) * Cs + j];
            dw2[g0 * kCs + i] = Ws[i] * (Es[i] - dot);
        }
    }
}

inline unsigned grid_for(int64_t n) { int64_t g = (n + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" int64_t afm_colstats_workspace_bytes(int64_t rows, int32_t C) {
    if (rows < 0 || C <= 0) return AFM_E_BADARG;
    return (int64_t)stat_plan(rows, C).chunks * 2 * C * (int64_t)sizeof(float);
}

extern "C" int afm_colstats(const float* x, int64_t rows, int32_t C, float* stats, void* ws, int64_t ws_bytes, void* stream) {
    if (!stats || rows <= 0 || !x) return AFM_E_BADARG;
    const int rc = launch_colstats<0>(x, nullptr, nullptr, nullptr, nullptr, rows, C, stats, ws, ws_bytes, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(copy_row_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, x, stats + 2 * C, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_bn_finalize(const float* stats, int32_t world, int64_t rows_per_rank, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                               int32_t C, void* stream) {
    if (!stats || !gamma || !beta || !mean || !rstd || !scale || !shift || C <= 0 || rows_per_rank <= 0 || world <= 0) return AFM_E_BADARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return AFM_E_BADARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, stats, world, (float)rows_per_rank, gamma, beta, eps,
                       momentum, running_mean, running_var, mean, rstd, scale, shift, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_colaffine(const float* x, const float* scale, const float* shift, const float* residual, int32_t relu, float* y, int64_t rows,
                             int32_t C, void* stream) {
    if (rows == 0) return 0;
    if (!x || !scale || !shift || !y || rows < 0 || C <= 0) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(colaffine_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, residual, relu, y, rows * C, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_bn_bwd_stats(const float* dy, const float* x, const float* y, const float* mean, const float* rstd, int64_t rows, int32_t C,
                                float* stats, void* ws, int64_t ws_bytes, void* stream) {
    if (!stats || !mean || !rstd || (rows > 0 && (!dy || !x))) return AFM_E_BADARG;
    return launch_colstats<1>(dy, x, y, mean, rstd, rows, C, stats, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int afm_bn_bwd_apply(const float* dy, const float* x, const float* y, const float* mean, const float* rstd, const float* gamma,
                                const float* stats, int64_t count, float* dx, float* dres, int64_t rows, int32_t C, void* stream) {
    if (rows == 0) return 0;
    if (!dy || !x || !mean || !rstd || !gamma || !stats || !dx || rows < 0 || C <= 0 || count <= 0) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, dy, x, y, mean, rstd, gamma, stats,
                       1.0f / (float)count, dx, dres, rows * C, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_group_points(const float* xyz, const float* new_xyz, const float* feat, const int32_t* idx, float* out, int64_t rows, int32_t k,
                                int32_t C, void* stream) {
    if (rows == 0) return 0;
    if (!xyz || !new_xyz || !idx || !out || rows < 0 || k <= 0 || C < 0 || (C > 0 && !feat)) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(group_points_kernel, dim3(grid_for(rows * (3 + C))), dim3(256), 0, (hipStream_t)stream, xyz, new_xyz, feat, idx, out, rows, k, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_scatter_add_rows(const float* src, int64_t ld, int32_t col_offset, const int32_t* idx, float* dst, int64_t rows, int32_t C,
                                    void* stream) {
    if (rows == 0) return 0;
    if (!src || !idx || !dst || rows < 0 || C <= 0 || ld < col_offset + C) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(scatter_add_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, src, ld, col_offset, idx, dst, rows, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t afm_scatter_plan_words(int64_t entries, int64_t n_dst) {
    if (entries < 0 || n_dst < 0 || entries >= (int64_t)1 << 31 || n_dst >= (int64_t)1 << 31) return AFM_E_BADARG;
    return (n_dst + 1) + entries + /* scratch: counters / cursors + arrival-order list + block sums of the scan */ n_dst + entries + (n_dst / PLAN_BLK + 1);
}

extern "C" int afm_scatter_plan(const int32_t* idx, int64_t entries, int64_t n_dst, int32_t* plan, void* stream) {
    if (n_dst == 0 && entries == 0) return 0;                   // empty batch: nothing to invert
    if (entries < 0 || n_dst <= 0 || entries >= (int64_t)1 << 31 || n_dst >= (int64_t)1 << 31 || !plan || (entries > 0 && !idx)) return AFM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    int32_t *off = plan, *ent = plan + (n_dst + 1), *cnt = ent + entries, *tmp = cnt + n_dst;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, s);
    if (hipMemsetAsync(cnt, 0, (size_t)n_dst * 4, s) != hipSuccess) return (int)hipGetLastError();
    if (entries > 0) hipLaunchKernelGGL(plan_count_kernel, dim3(grid_for(entries)), dim3(256), 0, s, idx, entries, n_dst, cnt);
    const int64_t nblk = (n_dst + PLAN_BLK - 1) / PLAN_BLK;
    int32_t* bsum = tmp + entries;
    hipLaunchKernelGGL(plan_blocksum_kernel, dim3((unsigned)nblk), dim3(256), 0, s, cnt, n_dst, bsum);
    hipLaunchKernelGGL(plan_scan_blocks_kernel, dim3(1), dim3(256), 0, s, bsum, nblk, off + n_dst);
    hipLaunchKernelGGL(plan_scan_kernel, dim3((unsigned)nblk), dim3(256), 0, s, cnt, n_dst, bsum, off);
    if (entries > 0) {
        hipLaunchKernelGGL(plan_fill_kernel, dim3(grid_for(entries)), dim3(256), 0, s, idx, entries, n_dst, off, cnt, tmp);
        hipLaunchKernelGGL(plan_rank_kernel, dim3(grid_for(entries)), dim3(256), 0, s, idx, n_dst, off, tmp, ent);
    }
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_segment_sum_rows(const float* src, int64_t ld, int32_t col_offset, int32_t row_div, const float* dist2, const int32_t* plan, float* dst,
                                    int64_t n_dst, int32_t C, void* stream) {
    if (n_dst == 0) return 0;
    if (!src || !plan || !dst || n_dst < 0 || C <= 0 || row_div <= 0 || ld < col_offset + C || (dist2 && row_div > 8)) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(segment_sum_kernel, dim3(grid_for(n_dst * C)), dim3(256), 0, (hipStream_t)stream, src, ld, col_offset, row_div, dist2, plan,
                       plan + (n_dst + 1), dst, n_dst, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_group_max(const float* x, float* y, int32_t* arg, int64_t m, int32_t k, int32_t C, void* stream) {
    if (m == 0) return 0;
    if (!x || !y || !arg || m < 0 || k <= 0 || C <= 0) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(group_max_kernel, dim3(grid_for(m * C)), dim3(256), 0, (hipStream_t)stream, x, y, arg, m, k, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_group_max_bwd(const float* dy, const int32_t* arg, float* dx, int64_t m, int32_t k, int32_t C, void* stream) {
    if (m == 0) return 0;
    if (!dy || !arg || !dx || m < 0 || k <= 0 || C <= 0) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3(grid_for(m * k * C)), dim3(256), 0, (hipStream_t)stream, dy, arg, dx, m, k, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_group_sum(const float* x, float* out, int64_t m, int32_t k, int32_t C, float scale, void* stream) {
    if (m == 0) return 0;
    if (!x || !out || m < 0 || k <= 0 || C <= 0) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(group_sum_kernel, dim3(grid_for(m * C)), dim3(256), 0, (hipStream_t)stream, x, out, m, k, C, scale);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_pt_w0(const float* kg, const float* q, const float* pr, float* out, int64_t m, int32_t k, int32_t C, void* stream) {
    if (m == 0) return 0;
    if (!kg || !q || !pr || !out || m < 0 || k <= 0 || C <= 0) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(pt_w0_kernel, dim3(grid_for(m * k * C)), dim3(256), 0, (hipStream_t)stream, kg, q, pr, out, m, k, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_pt_aggregate(const float* vg, const float* pr, const float* w2, float* out, float* sw, int64_t m, int32_t k, int32_t C,
                                int32_t share_planes, void* stream) {
    if (m == 0) return 0;
    if (!vg || !pr || !w2 || !out || !sw || m < 0 || k <= 0 || k > 16 || C <= 0 || share_planes <= 0 || C % share_planes) return AFM_E_BADARG;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    const int Cs = C / share_planes;
    const uintptr_t al = (uintptr_t)vg | (uintptr_t)pr | (uintptr_t)w2 | (uintptr_t)sw;
    // the fused form's dynamic LDS stays inside the 64 KB every kernel may use without opting in (small share_planes make the weight slab as
    // large as the value slab): G shrinks until it fits, a shape whose single point does not fit takes the two-pass kernels below
    auto lds_of = [&](int g) { return (size_t)g * (k * C + k * Cs) * sizeof(float); };
    int G = k * C <= 8192 ? 8192 / (k * C) : 0;
    if (G > 64) G = 64;
    while (G > 1 && lds_of(G) > 65536) --G;
    if ((C & 3) == 0 && ((k * Cs) & 3) == 0 && (al & 15) == 0 && G >= 1 && lds_of(G) <= 65536) {
        const size_t lds = lds_of(G);
        int64_t nb = (m + G - 1) / G;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(pt_aggregate_fused_kernel, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, vg, pr, w2, out, sw, m, k, C, Cs, G);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(pt_aggregate_kernel, dim3(grid_for(m * C)), dim3(256), 0, (hipStream_t)stream, vg, pr, w2, out, sw, m, k, C, C / share_planes);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_pt_aggregate_bwd(const float* vg, const float* pr, const float* sw, const float* dout, float* da, float* dw2, int64_t m,
                                    int32_t k, int32_t C, int32_t share_planes, void* stream) {
    if (m == 0) return 0;
    if (!vg || !pr || !sw || !dout || !da || !dw2 || m < 0 || k <= 0 || k > 16 || C <= 0 || share_planes <= 0 || C % share_planes) return AFM_E_BADARG;
    const int Cs = C / share_planes;
    AfmProf prof(AFM_PROF_POINT_TRAIN, 0.0, (hipStream_t)stream);
    const uintptr_t al = (uintptr_t)vg | (uintptr_t)pr | (uintptr_t)sw | (uintptr_t)dout | (uintptr_t)da | (uintptr_t)dw2;
    auto lds_of = [&](int g) { return (size_t)g * (k * C + C + 2 * k * Cs) * sizeof(float); };      // (share_planes = 1, k = 16, C = 32: 100 KB at the old G)
    int G = k * C <= 8192 ? 8192 / (k * C) : 0;
    if (G > 64) G = 64;
    while (G > 1 && lds_of(G) > 65536) --G;
    if ((C & 3) == 0 && ((k * Cs) & 3) == 0 && (al & 15) == 0 && G >= 1 && lds_of(G) <= 65536) {          // the fused form (a function of the shape only)
        const size_t lds = lds_of(G);
        int64_t nb = (m + G - 1) / G;
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(pt_aggregate_bwd_fused_kernel, dim3((unsigned)nb), dim3(256), lds, (hipStream_t)stream, vg, pr, sw, dout, da, dw2, m, k, C, Cs, G);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    hipLaunchKernelGGL(pt_aggregate_bwd_a_kernel, dim3(grid_for(m * k * C)), dim3(256), 0, (hipStream_t)stream, dout, sw, da, m, k, C, Cs);
    hipLaunchKernelGGL(pt_aggregate_bwd_w_kernel, dim3(grid_for(m * Cs)), dim3(256), 0, (hipStream_t)stream, vg, pr, sw, dout, dw2, m, k, C, Cs);
    AFM_CHECK_LAUNCH();
    return 0;
}
