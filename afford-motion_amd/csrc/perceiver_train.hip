// Training-path attention of the CDM ContactPerceiver (models/cdm.py:155-188, models/modules.py:301-381 under model.train()):
// the two cross-attentions are extremely rectangular - 2 latent queries over N = 8192 point keys (encoder) and N point
// queries over 2 latent keys (decoder) - so neither is GEMM-shaped work: both are HBM streams over the [B, N, C] point
// tensors with 16 dot products / 16 axpys per point.  As-written formulation (K, V, Q of the points are materialised by the
// MFMA GEMMs of afm_linear), attention-probability dropout by the counter hash of common.h.
//
//   few-query attention (encoder): S[b, h*TQ+q, n] = scale * Q[b,q,h,:] . K[b,n,h,:]  -> row softmax over n -> dropout ->
//                                  O[b,q,h,:] = sum_n Pd[b,hq,n] V[b,n,h,:]
//     xq_dots   : wave per point, lane = C/64 contiguous channels, lane-group (= one head) butterfly for the TQ dot products
//     xq_softmax: workgroup per (b, hq) row of N
//     xq_wsum   : thread per channel, points split over workgroups, fixed-order second stage
//     xq_outer  : Y[b,n,h,:] = sum_q W[b,hq,n] * A[b,q,h,:]   (dV from Pd and dO, dK from dS and Q)
//   few-key attention (decoder): per point and head a softmax over TK = 2 keys - one wave per point, everything in registers;
//     the K / V gradients are per-workgroup partial sums reduced in fixed order.
#include "common.h"
#include "profile.h"
#include <math.h>

namespace {

constexpr int TQ = 2;      // latent tokens (text, time)

// dst[b][c] = sum_s src[b][s][c] (fixed order); blockIdx.y = b
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ src, int64_t src_batch, int64_t stride, int S,
                                                           float* __restrict__ dst, int64_t dst_batch, int n) {
    __shared__ float red[16][64];
    src += blockIdx.y * src_batch; dst += blockIdx.y * dst_batch;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float a0 = 0.f;
    if (c < n) for (int s = ty; s < S; s += 16) a0 += src[(int64_t)s * stride + c];
    red[ty][tx] = a0;
    __syncthreads();
    if (ty == 0 && c < n) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += red[i][tx];
        dst[c] = v;
    }
}

// S[b, h*TQ+q, n] = scale * sum_c Q[b,q,h*dh+c] * X[b,n,h*dh+c];  wave per point, VPL = C/64 channels per lane
template <int VPL>
__global__ __launch_bounds__(256) void xq_dots_kernel(const float* __restrict__ Q, const float* __restrict__ X, float* __restrict__ S, int B, int N, int H,
                                                      float scale) {
    const int C = 64 * VPL, dh = C / H, gl = dh / VPL;            // gl lanes per head
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    float q[TQ][VPL];
#pragma unroll
    for (int t = 0; t < TQ; ++t)
#pragma unroll
        for (int v = 0; v < VPL; ++v) q[t][v] = Q[((int64_t)b * TQ + t) * C + lane * VPL + v] * scale;
    const int h = lane / gl;
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        const float* xp = X + ((int64_t)b * N + n) * C + lane * VPL;
        float x[VPL];
#pragma unroll
        for (int v = 0; v < VPL; v += 4) {
            const float4 a = *reinterpret_cast<const float4*>(xp + v);
            x[v] = a.x; x[v + 1] = a.y; x[v + 2] = a.z; x[v + 3] = a.w;
        }
        float d[TQ];
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < VPL; ++v) s += q[t][v] * x[v];
            s += lane_xor<4>(s); s += lane_xor<2>(s); s += lane_xor<1>(s);          // gl == 8 lanes per head (checked on the host)
            d[t] = s;
        }
        if ((lane % gl) == 0) {
#pragma unroll
            for (int t = 0; t < TQ; ++t) S[((int64_t)b * H * TQ + h * TQ + t) * N + n] = d[t];
        }
    }
}

// in-place row softmax over N; one workgroup per row
__global__ __launch_bounds__(256) void xq_softmax_kernel(float* __restrict__ S, int N) {
    __shared__ float red[4];
    float* row = S + (int64_t)blockIdx.x * N;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < N; i += 256) mx = fmaxf(mx, row[i]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) { const float e = __expf(row[i] - mx); row[i] = e; sum += e; }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = threadIdx.x; i < N; i += 256) row[i] *= inv;
}

// dS = P * (dPd * keep - sum_n P * dPd * keep), in place on dPd; one workgroup per row
__global__ __launch_bounds__(256) void xq_softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, int N, float drop_p, uint64_t seed,
                                                             uint32_t id) {
    __shared__ float red[4];
    const DropKey dk(drop_p, seed, id);
    const float* p = P + (int64_t)blockIdx.x * N;
    float* d = dP + (int64_t)blockIdx.x * N;
    float dot = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) {
        float v = d[i];
        if (drop_p > 0.0f) v *= dk(blockIdx.x, (uint32_t)i);
        d[i] = v;
        dot += p[i] * v;
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    dot = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = threadIdx.x; i < N; i += 256) d[i] = p[i] * (d[i] - dot);
}

// part[b][chunk][q][c] = sum_{n in chunk} W[b, h(c)*TQ+q, n] * keep * X[b,n,c];  thread per channel c
__global__ __launch_bounds__(512) void xq_wsum_kernel(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ part, int N, int C,
                                                      int H, int chunks, float scale, float drop_p, uint64_t seed, uint32_t id) {
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (N + chunks - 1) / chunks, n0 = chunk * per, n1 = min(N, n0 + per);
    const DropKey dk(drop_p, seed, id);
    for (int c = threadIdx.x; c < C; c += 512) {
        const int h = c / (C / H);
        const uint32_t r0 = (uint32_t)((b * H + h) * TQ);
        const float* w0 = W + (int64_t)r0 * N;
        float acc[TQ];
#pragma unroll
        for (int t = 0; t < TQ; ++t) acc[t] = 0.f;
        for (int n = n0; n < n1; ++n) {
            const float x = X[((int64_t)b * N + n) * C + c];
#pragma unroll
            for (int t = 0; t < TQ; ++t) {
                float w = w0[(int64_t)t * N + n];
                if (drop_p > 0.0f) w *= dk(r0 + t, (uint32_t)n);
                acc[t] += w * x;
            }
        }
#pragma unroll
        for (int t = 0; t < TQ; ++t) part[(((int64_t)b * chunks + chunk) * TQ + t) * C + c] = acc[t] * scale;
    }
}

// Y[b,n,c] = scale * sum_q W[b, h(c)*TQ+q, n] * keep * A[b,q,c]
__global__ __launch_bounds__(256) void xq_outer_kernel(const float* __restrict__ W, const float* __restrict__ A, float* __restrict__ Y, int N, int C, int H,
                                                       float scale, float drop_p, uint64_t seed, uint32_t id) {
    const int b = blockIdx.y;
    const DropKey dk(drop_p, seed, id);
    const int64_t total = (int64_t)N * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / C), c = (int)(i - (int64_t)n * C);
        const int h = c / (C / H);
        const uint32_t r0 = (uint32_t)((b * H + h) * TQ);
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < TQ; ++t) {
            float w = W[((int64_t)r0 + t) * N + n];
            if (drop_p > 0.0f) w *= dk(r0 + t, (uint32_t)n);
            acc += w * A[((int64_t)b * TQ + t) * C + c];
        }
        Y[((int64_t)b * N + n) * C + c] = acc * scale;
    }
}

// ---------------------------------------------------------------------------------------------- few-key attention (TK = 2)
// wave per point; lane holds VPL = C/64 channels; gl = dh/VPL lanes per head
template <int VPL, bool BWD>
__global__ __launch_bounds__(256) void xk_attn_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                      float* __restrict__ O, const float* __restrict__ dO, float* __restrict__ dQ,
                                                      float* __restrict__ partKV, int N, int H, float scale, float drop_p, uint64_t seed, uint32_t id) {
    constexpr int TK = 2;
    const int C = 64 * VPL, dh = C / H, gl = dh / VPL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y, h = lane / gl;
    const DropKey dk(drop_p, seed, id);
    float k[TK][VPL], v[TK][VPL];
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
        for (int e = 0; e < VPL; ++e) {
            k[j][e] = K[((int64_t)b * TK + j) * C + lane * VPL + e];
            v[j][e] = V[((int64_t)b * TK + j) * C + lane * VPL + e];
        }
    float dk_acc[TK][VPL], dv_acc[TK][VPL];
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
        for (int e = 0; e < VPL; ++e) { dk_acc[j][e] = 0.f; dv_acc[j][e] = 0.f; }

    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        const int64_t row = ((int64_t)b * N + n) * C + lane * VPL;
        float q[VPL];
#pragma unroll
        for (int e = 0; e < VPL; ++e) q[e] = Q[row + e];
        float s[TK];
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < VPL; ++e) a += q[e] * k[j][e];
            a += lane_xor<4>(a); a += lane_xor<2>(a); a += lane_xor<1>(a);
            s[j] = a * scale;
        }
        const float mx = fmaxf(s[0], s[1]);
        const float e0 = __expf(s[0] - mx), e1 = __expf(s[1] - mx), inv = 1.0f / (e0 + e1);
        float p[TK] = {e0 * inv, e1 * inv};
        float keep[TK] = {1.0f, 1.0f};
        if (drop_p > 0.0f) {
            const uint32_t r = (uint32_t)(((int64_t)b * H + h) * N + n);             // row = (b, head, query point), col = key
            keep[0] = dk(r, 0u); keep[1] = dk(r, 1u);
        }
        if (!BWD) {
#pragma unroll
            for (int e = 0; e < VPL; ++e) O[row + e] = p[0] * keep[0] * v[0][e] + p[1] * keep[1] * v[1][e];
        } else {
            float g[VPL];
#pragma unroll
            for (int e = 0; e < VPL; ++e) g[e] = dO[row + e];
            float dp[TK];
#pragma unroll
            for (int j = 0; j < TK; ++j) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < VPL; ++e) a += g[e] * v[j][e];
                a += lane_xor<4>(a); a += lane_xor<2>(a); a += lane_xor<1>(a);
                dp[j] = a * keep[j];
            }
            const float dot = p[0] * dp[0] + p[1] * dp[1];
            const float ds0 = p[0] * (dp[0] - dot) * scale, ds1 = p[1] * (dp[1] - dot) * scale;
#pragma unroll
            for (int e = 0; e < VPL; ++e) {
                dQ[row + e] = ds0 * k[0][e] + ds1 * k[1][e];
                dk_acc[0][e] += ds0 * q[e]; dk_acc[1][e] += ds1 * q[e];
                dv_acc[0][e] += p[0] * keep[0] * g[e]; dv_acc[1][e] += p[1] * keep[1] * g[e];
            }
        }
    }
    if (BWD) {
        // per-workgroup partial of dK / dV: 4 waves summed through LDS in fixed order -> partKV[b][block][2 (k|v)][TK][C]
        __shared__ float red[4][2 * TK * 64 * VPL];
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int e = 0; e < VPL; ++e) {
                red[wave][(0 * TK + j) * C + lane * VPL + e] = dk_acc[j][e];
                red[wave][(1 * TK + j) * C + lane * VPL + e] = dv_acc[j][e];
            }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * TK * C; i += 256)
            partKV[((int64_t)b * gridDim.x + blockIdx.x) * 2 * TK * C + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    }
}

inline int xq_chunks(int N) { int c = (N + 127) / 128; return c > 64 ? 64 : (c < 1 ? 1 : c); }
inline int xk_blocks(int N) { int g = (N + 63) / 64; return g > 64 ? 64 : (g < 1 ? 1 : g); }

}  // namespace

extern "C" int64_t afm_xq_workspace_bytes(int32_t B, int32_t N, int32_t C) {
    if (B < 0 || N <= 0 || C <= 0) return AFM_E_BADARG;
    return (int64_t)B * xq_chunks(N) * TQ * C * (int64_t)sizeof(float);
}

// P [B, H*2, N] <- softmax_n(scale * Q K^T);  O [B, 2, C] <- sum_n dropout(P) V
extern "C" int afm_xq_attention_fwd(const float* Q, const float* K, const float* V, float* P, float* O, int32_t B, int32_t N, int32_t H, int32_t C,
                                    float drop_p, uint64_t drop_seed, uint32_t drop_id, void* ws, int64_t ws_bytes, void* stream) {
    if (B == 0) return 0;
    if (!Q || !K || !V || !P || !O || B < 0 || N <= 0 || H <= 0 || (C != 512 && C != 256) || C % H || (C / H) / (C / 64) != 8 || drop_p < 0.0f || drop_p >= 1.0f) return AFM_E_BADARG;
    const int chunks = xq_chunks(N);
    if (!ws || ws_bytes < (int64_t)B * chunks * TQ * C * (int64_t)sizeof(float)) return AFM_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)(C / H));
    AfmProf prof(AFM_PROF_CDM_TRAIN, 0.0, s);
    const int gx = (N + 3) / 4 > 2048 ? 2048 : (N + 3) / 4;
    if (C == 512) hipLaunchKernelGGL(xq_dots_kernel<8>, dim3(gx, B), dim3(256), 0, s, Q, K, P, B, N, H, scale);
    else hipLaunchKernelGGL(xq_dots_kernel<4>, dim3(gx, B), dim3(256), 0, s, Q, K, P, B, N, H, scale);
    hipLaunchKernelGGL(xq_softmax_kernel, dim3(B * H * TQ), dim3(256), 0, s, P, N);
    hipLaunchKernelGGL(xq_wsum_kernel, dim3(chunks, B), dim3(512), 0, s, P, V, (float*)ws, N, C, H, chunks, 1.0f, drop_p, drop_seed, drop_id);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((TQ * C + 63) / 64, B), dim3(1024), 0, s, (float*)ws, (int64_t)chunks * TQ * C, (int64_t)TQ * C, chunks, O,
                       (int64_t)TQ * C, TQ * C);
    AFM_CHECK_LAUNCH();
    return 0;
}

// dQ [B,2,C], dK, dV [B,N,C] from dO [B,2,C]; P from the forward; dS scratch [B, H*2, N]
extern "C" int afm_xq_attention_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO, float* dS, float* dQ, float* dK,
                                    float* dV, int32_t B, int32_t N, int32_t H, int32_t C, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* ws,
                                    int64_t ws_bytes, void* stream) {
    if (B == 0) return 0;
    if (!Q || !K || !V || !P || !dO || !dS || !dQ || !dK || !dV || B < 0 || N <= 0 || H <= 0 || (C != 512 && C != 256) || C % H || (C / H) / (C / 64) != 8) return AFM_E_BADARG;
    const int chunks = xq_chunks(N);
    if (!ws || ws_bytes < (int64_t)B * chunks * TQ * C * (int64_t)sizeof(float)) return AFM_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)(C / H));
    AfmProf prof(AFM_PROF_CDM_TRAIN, 0.0, s);
    const int gx = (N + 3) / 4 > 2048 ? 2048 : (N + 3) / 4;
    const int go = (int)(((int64_t)N * C + 255) / 256 > 4096 ? 4096 : ((int64_t)N * C + 255) / 256);
    // dV = dropout(P)^T dO
    hipLaunchKernelGGL(xq_outer_kernel, dim3(go, B), dim3(256), 0, s, P, dO, dV, N, C, H, 1.0f, drop_p, drop_seed, drop_id);
    // dPd = dO . V ; dS = P * (keep * dPd - rowdot)
    if (C == 512) hipLaunchKernelGGL(xq_dots_kernel<8>, dim3(gx, B), dim3(256), 0, s, dO, V, dS, B, N, H, 1.0f);
    else hipLaunchKernelGGL(xq_dots_kernel<4>, dim3(gx, B), dim3(256), 0, s, dO, V, dS, B, N, H, 1.0f);
    hipLaunchKernelGGL(xq_softmax_bwd_kernel, dim3(B * H * TQ), dim3(256), 0, s, P, dS, N, drop_p, drop_seed, drop_id);
    // dK = scale * dS^T Q ; dQ = scale * dS K
    hipLaunchKernelGGL(xq_outer_kernel, dim3(go, B), dim3(256), 0, s, dS, Q, dK, N, C, H, scale, 0.0f, 0, 0);
    hipLaunchKernelGGL(xq_wsum_kernel, dim3(chunks, B), dim3(512), 0, s, dS, K, (float*)ws, N, C, H, chunks, scale, 0.0f, 0, 0);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((TQ * C + 63) / 64, B), dim3(1024), 0, s, (float*)ws, (int64_t)chunks * TQ * C, (int64_t)TQ * C, chunks, dQ,
                       (int64_t)TQ * C, TQ * C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t afm_xk_workspace_bytes(int32_t B, int32_t N, int32_t C) {
    if (B < 0 || N <= 0 || C <= 0) return AFM_E_BADARG;
    return (int64_t)B * xk_blocks(N) * 4 * C * (int64_t)sizeof(float);
}

// O [B,N,C] = dropout(softmax_2(scale * Q K^T)) V with K, V [B,2,C]
extern "C" int afm_xk_attention_fwd(const float* Q, const float* K, const float* V, float* O, int32_t B, int32_t N, int32_t H, int32_t C, float drop_p,
                                    uint64_t drop_seed, uint32_t drop_id, void* stream) {
    if (B == 0) return 0;
    if (!Q || !K || !V || !O || B < 0 || N <= 0 || H <= 0 || (C != 512 && C != 256) || C % H || (C / H) / (C / 64) != 8 || drop_p < 0.0f || drop_p >= 1.0f) return AFM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)(C / H));
    AfmProf prof(AFM_PROF_CDM_TRAIN, 0.0, s);
    const int gx = (N + 3) / 4 > 1024 ? 1024 : (N + 3) / 4;
    if (C == 512) hipLaunchKernelGGL((xk_attn_kernel<8, false>), dim3(gx, B), dim3(256), 0, s, Q, K, V, O, nullptr, nullptr, nullptr, N, H, scale, drop_p, drop_seed, drop_id);
    else hipLaunchKernelGGL((xk_attn_kernel<4, false>), dim3(gx, B), dim3(256), 0, s, Q, K, V, O, nullptr, nullptr, nullptr, N, H, scale, drop_p, drop_seed, drop_id);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_xk_attention_bwd(const float* Q, const float* K, const float* V, const float* dO, float* dQ, float* dK, float* dV, int32_t B, int32_t N,
                                    int32_t H, int32_t C, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* ws, int64_t ws_bytes, void* stream) {
    if (B == 0) return 0;
    if (!Q || !K || !V || !dO || !dQ || !dK || !dV || B < 0 || N <= 0 || H <= 0 || (C != 512 && C != 256) || C % H || (C / H) / (C / 64) != 8) return AFM_E_BADARG;
    const int blocks = xk_blocks(N);
    if (!ws || ws_bytes < (int64_t)B * blocks * 4 * C * (int64_t)sizeof(float)) return AFM_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const float scale = 1.0f / sqrtf((float)(C / H));
    AfmProf prof(AFM_PROF_CDM_TRAIN, 0.0, s);
    float* part = (float*)ws;
    if (C == 512) hipLaunchKernelGGL((xk_attn_kernel<8, true>), dim3(blocks, B), dim3(256), 0, s, Q, K, V, nullptr, dO, dQ, part, N, H, scale, drop_p, drop_seed, drop_id);
    else hipLaunchKernelGGL((xk_attn_kernel<4, true>), dim3(blocks, B), dim3(256), 0, s, Q, K, V, nullptr, dO, dQ, part, N, H, scale, drop_p, drop_seed, drop_id);
    // part[b][block][k|v][2][C] -> dK[b][2][C], dV[b][2][C]
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((2 * C + 63) / 64, B), dim3(1024), 0, s, part, (int64_t)blocks * 4 * C, (int64_t)4 * C, blocks, dK,
                       (int64_t)2 * C, 2 * C);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((2 * C + 63) / 64, B), dim3(1024), 0, s, part + 2 * C, (int64_t)blocks * 4 * C, (int64_t)4 * C, blocks, dV,
                       (int64_t)2 * C, 2 * C);
    AFM_CHECK_LAUNCH();
    return 0;
}
