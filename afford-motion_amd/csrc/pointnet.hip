// Fused point-cloud layers of the SceneMapEncoder (reference models/scene_models/pointtransformer.py):
//
//  afm_transition_down  - "set abstraction": for every sampled point i and its k = 16 nearest source
//      points j:  y[i] = max_j ReLU(BN(W [p_j - p'_i ; x_j]))   (pointtransformer.py:53-69)
//      = a GEMM over gathered rows with a max-pool epilogue.  The (m, k, 3+c) grouped tensor the
//      reference materialises never exists: rows are gathered straight into the LDS A-tile, products
//      run on v_mfma_f32_32x32x2_f32, and the 16 rows of one point sit in one 32x32 MFMA tile so the
//      pool is 8 in-register maxima + one cross-half exchange.
//
//  afm_pt_attention - the whole PointTransformerLayer after its q/k/v projections
//      (pointtransformer.py:26-38): gather k neighbours, position MLP, weight MLP
//      (BN-ReLU-Linear-BN-ReLU-Linear), softmax over the neighbours, grouped weighted sum, optional
//      fused BatchNorm+ReLU of the enclosing block.  One wave per point; the channel-parallel phases
//      keep (v + p_r) in registers, the (neighbour, weight-channel)-parallel phases go through LDS.
#include "common.h"
#include "profile.h"

namespace {

constexpr int BK = 32, LDK = 36, TD_K = 16;

__global__ __launch_bounds__(256, 2) void transition_down_kernel(const float* __restrict__ p, const float* __restrict__ x, int c,
                                                                 const float* __restrict__ n_p, const int* __restrict__ knn,
                                                                 const float* __restrict__ W, int cout,
                                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                                 float* __restrict__ out, int M) {
    constexpr int BM = 128, BN = 64;
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDK];
    constexpr int STAGE = (BM + BN) * LDK;
    const int K = 3 + c;
    const int bm = blockIdx.x, bn = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    const int c4 = tid & 7, r0 = tid >> 3;
    const int64_t rows = (int64_t)M * TD_K;

    // per-thread gather descriptors of its 4 A rows (fixed over the K loop)
    int nb[4];
    float ctr[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t gr = (int64_t)bm * BM + r0 + 32 * i;
        if (gr < rows) {
            nb[i] = knn[gr];
            const int pt = (int)(gr >> 4);
            ctr[i][0] = n_p[pt * 3 + 0]; ctr[i][1] = n_p[pt * 3 + 1]; ctr[i][2] = n_p[pt * 3 + 2];
        } else {
            nb[i] = -1; ctr[i][0] = ctr[i][1] = ctr[i][2] = 0.f;
        }
    }
    auto a_val = [&](int i, int k) -> float {
        if (nb[i] < 0 || k >= K) return 0.f;
        return k < 3 ? p[(int64_t)nb[i] * 3 + k] - ctr[i][k] : x[(int64_t)nb[i] * c + (k - 3)];
    };
    float4 ra[4], rw[2];
    auto load_tiles = [&](int k0) {
        const int k = k0 + c4 * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = make_float4(a_val(i, k), a_val(i, k + 1), a_val(i, k + 2), a_val(i, k + 3));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int o = bn * BN + r0 + 32 * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o < cout) {
                const float* wp = W + (int64_t)o * K + k;
                if (k + 0 < K) v.x = wp[0];
                if (k + 1 < K) v.y = wp[1];
                if (k + 2 < K) v.z = wp[2];
                if (k + 3 < K) v.w = wp[3];
            }
            rw[i] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        float* a = lds + buf * STAGE;
        float* w = a + BM * LDK;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(a + (r0 + 32 * i) * LDK + c4 * 4) = ra[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(w + (r0 + 32 * i) * LDK + c4 * 4) = rw[i];
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int nk = (K + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);
        const float* a_base = lds + cur * STAGE + (wm * 64 + r32) * LDK + hh * 16;
        const float* w_base = lds + cur * STAGE + BM * LDK + (wn * 32 + r32) * LDK + hh * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 a0 = *reinterpret_cast<const float4*>(a_base + j * 4);
            const float4 a1 = *reinterpret_cast<const float4*>(a_base + 32 * LDK + j * 4);
            const float4 b = *reinterpret_cast<const float4*>(w_base + j * 4);
            acc[0] = mfma32(a0.x, b.x, acc[0]); acc[1] = mfma32(a1.x, b.x, acc[1]);
            acc[0] = mfma32(a0.y, b.y, acc[0]); acc[1] = mfma32(a1.y, b.y, acc[1]);
            acc[0] = mfma32(a0.z, b.z, acc[0]); acc[1] = mfma32(a1.z, b.z, acc[1]);
            acc[0] = mfma32(a0.w, b.w, acc[0]); acc[1] = mfma32(a1.w, b.w, acc[1]);
        }
        if (kt + 1 < nk) store_tiles(cur ^ 1);
        __syncthreads();
    }

    // epilogue: BN (scale/shift) + ReLU + max over the 16 neighbour rows of each point.
    // 32-row tile = 2 points; reg r <-> row (r&3) + 8*(r>>2) + 4*hh, so r < 8 is the first point.
    const int col = bn * BN + wn * 32 + r32;
    if (col < cout) {
        const float sc = scale[col], sh = shift[col];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            float mA = 0.f, mB = 0.f;                                    // ReLU output >= 0
#pragma unroll
            for (int r = 0; r < 8; ++r) mA = fmaxf(mA, acc[tm][r] * sc + sh);
#pragma unroll
            for (int r = 8; r < 16; ++r) mB = fmaxf(mB, acc[tm][r] * sc + sh);
            mA = fmaxf(mA, xor32(mA));
            mB = fmaxf(mB, xor32(mB));
            const int pA = (bm * BM + wm * 64 + tm * 32) >> 4;
            if (hh == 0) {
                if (pA < M) out[(int64_t)pA * cout + col] = mA;
                if (pA + 1 < M) out[(int64_t)(pA + 1) * cout + col] = mB;
            }
        }
    }
}

struct PtAttnArgs {
    const float* p; const float* qkv; const int* knn; float* out;
    int n, C, KN, CS;
    const float *lp0_w, *lp0_b, *lp_s, *lp_t, *lp3_w, *lp3_b;
    const float *w0_s, *w0_t, *w2_w, *w2_b, *w3_s, *w3_t, *w5_w, *w5_b;
    const float *out_s, *out_t;
    int relu;
    int w2_lds;      // 1: linear_w.2 weight transposed into LDS (C <= 256); 0: read from global (C = 512, few points)
};

template <int CPL, int KN, int NE>
__global__ __launch_bounds__(256) void pt_attention_kernel(const PtAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = a.C, CS = a.CS, LW = C + 1;
    float* W2T = smem;                               // [C][CS] (only when a.w2_lds)
    float* W5T = W2T + (a.w2_lds ? C * CS : 0);      // [CS][CS]  (W5T[t'][t] = W5[t][t'])
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwv = blockDim.x >> 6;
    float* w1 = W5T + CS * CS + wave * (KN * LW + 2 * KN * CS);   // [KN][C+1]
    float* u = w1 + KN * LW;                         // [KN][CS]
    float* att = u + KN * CS;                        // [KN][CS]

    if (a.w2_lds)
        for (int i = threadIdx.x; i < C * CS; i += blockDim.x) { const int t = i / C, ch = i % C; W2T[ch * CS + t] = a.w2_w[i]; }
    for (int i = threadIdx.x; i < CS * CS; i += blockDim.x) { const int t = i / CS, t2 = i % CS; W5T[t2 * CS + t] = a.w5_w[i]; }

    // position-MLP constants (3x3 + BN fold), identical in every lane
    float l0[9], l0b[3], ls[3], lt[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) l0[i] = a.lp0_w[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { l0b[i] = a.lp0_b[i]; ls[i] = a.lp_s[i]; lt[i] = a.lp_t[i]; }
    // per-lane channel constants
    float p3w[CPL][3], p3b[CPL], s0[CPL], t0[CPL];
#pragma unroll
    for (int s = 0; s < CPL; ++s) {
        const int ch = lane + 64 * s;
        const bool ok = ch < C;
        p3w[s][0] = ok ? a.lp3_w[ch * 3 + 0] : 0.f; p3w[s][1] = ok ? a.lp3_w[ch * 3 + 1] : 0.f; p3w[s][2] = ok ? a.lp3_w[ch * 3 + 2] : 0.f;
        p3b[s] = ok ? a.lp3_b[ch] : 0.f; s0[s] = ok ? a.w0_s[ch] : 0.f; t0[s] = ok ? a.w0_t[ch] : 0.f;
    }
    const int tt = lane % CS, jb = lane / CS, jstep = 64 / CS;
    const float b2 = a.w2_b[tt], s3 = a.w3_s[tt], t3 = a.w3_t[tt], b5 = a.w5_b[tt];
    __syncthreads();

    const int quads = (a.n + nwv - 1) / nwv;
    for (int qd = blockIdx.x; qd < quads; qd += gridDim.x) {
        const int pt = qd * nwv + wave;
        const bool live = pt < a.n;
        const int pi = live ? pt : a.n - 1;
        float vp[KN][CPL];
        {   // ---- phase A (channel-parallel): gather, position MLP, first BN-ReLU of the weight MLP
            const float cx = a.p[(int64_t)pi * 3], cy = a.p[(int64_t)pi * 3 + 1], cz = a.p[(int64_t)pi * 3 + 2];
            float qv[CPL];
#pragma unroll
            for (int s = 0; s < CPL; ++s) { const int ch = lane + 64 * s; qv[s] = ch < C ? a.qkv[(int64_t)pi * 3 * C + ch] : 0.f; }
#pragma unroll
            for (int j = 0; j < KN; ++j) {
                const int nb = a.knn[(int64_t)pi * KN + j];
                const float dx = a.p[(int64_t)nb * 3] - cx, dy = a.p[(int64_t)nb * 3 + 1] - cy, dz = a.p[(int64_t)nb * 3 + 2] - cz;
                float h[3];
#pragma unroll
                for (int o = 0; o < 3; ++o) {
                    const float v = l0[o * 3] * dx + l0[o * 3 + 1] * dy + l0[o * 3 + 2] * dz + l0b[o];
                    h[o] = fmaxf(v * ls[o] + lt[o], 0.f);
                }
                const float* row = a.qkv + (int64_t)nb * 3 * C;
#pragma unroll
                for (int s = 0; s < CPL; ++s) {
                    const int ch = lane + 64 * s;
                    if (ch < C) {
                        const float pr = p3w[s][0] * h[0] + p3w[s][1] * h[1] + p3w[s][2] * h[2] + p3b[s];
                        const float kx = row[C + ch], vx = row[2 * C + ch];
                        vp[j][s] = vx + pr;
                        w1[j * LW + ch] = fmaxf(((kx - qv[s]) + pr) * s0[s] + t0[s], 0.f);
                    } else {
                        vp[j][s] = 0.f;
                    }
                }
            }
        }
        __syncthreads();
        {   // ---- phase B ((neighbour, weight-channel)-parallel): Linear(C -> C/s) + BN + ReLU
            float accu[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) accu[e] = b2;
            for (int ch = 0; ch < C; ++ch) {
                const float wv = a.w2_lds ? W2T[ch * CS + tt] : a.w2_w[(int64_t)tt * C + ch];
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int j = jb + e * jstep;
                    if (j < KN) accu[e] += wv * w1[j * LW + ch];
                }
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int j = jb + e * jstep;
                if (j < KN) u[j * CS + tt] = fmaxf(accu[e] * s3 + t3, 0.f);
            }
        }
        __syncthreads();
        {   // ---- phase C: Linear(C/s -> C/s)
            float z[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) z[e] = b5;
            for (int t2 = 0; t2 < CS; ++t2) {
                const float wv = W5T[t2 * CS + tt];
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int j = jb + e * jstep;
                    if (j < KN) z[e] += wv * u[j * CS + t2];
                }
            }
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int j = jb + e * jstep;
                if (j < KN) att[j * CS + tt] = z[e];
            }
        }
        __syncthreads();
        if (lane < CS) {   // ---- softmax over the KN neighbours, one lane per weight channel
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < KN; ++j) mx = fmaxf(mx, att[j * CS + lane]);
            float e[KN], sum = 0.f;
#pragma unroll
            for (int j = 0; j < KN; ++j) { e[j] = __expf(att[j * CS + lane] - mx); sum += e[j]; }
            const float inv = 1.0f / sum;
#pragma unroll
            for (int j = 0; j < KN; ++j) att[j * CS + lane] = e[j] * inv;
        }
        __syncthreads();
        // ---- phase D (channel-parallel): out[ch] = sum_j (v_j + p_r_j)[ch] * att[j][ch % (C/s)]
#pragma unroll
        for (int s = 0; s < CPL; ++s) {
            const int ch = lane + 64 * s;
            if (ch < C) {
                float o = 0.f;
#pragma unroll
                for (int j = 0; j < KN; ++j) o += vp[j][s] * att[j * CS + (ch % CS)];
                if (a.out_s) o = o * a.out_s[ch] + a.out_t[ch];
                if (a.relu) o = fmaxf(o, 0.f);
                if (live) a.out[(int64_t)pt * C + ch] = o;
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int afm_transition_down(const float* p, const float* x, int32_t c, const float* new_p, const int32_t* knn_idx,
                                   int32_t nsample, const float* weight, int32_t cout, const float* scale, const float* shift,
                                   float* out, int32_t M, void* stream) {
    if (!p || !x || !new_p || !knn_idx || !weight || !scale || !shift || !out || c <= 0 || cout <= 0 || M < 0) return AFM_E_BADARG;
    if (nsample != TD_K) return AFM_E_UNSUPPORTED;          // every strided TransitionDown of the reference uses k = 16
    if (M == 0) return 0;
    const dim3 grid((unsigned)(((int64_t)M * TD_K + 127) / 128), (cout + 63) / 64), block(256);
    AfmProf prof(AFM_PROF_TD, 2.0 * M * TD_K * (3 + c) * cout, (hipStream_t)stream);
    hipLaunchKernelGGL(transition_down_kernel, grid, block, 0, (hipStream_t)stream, p, x, c, new_p, knn_idx, weight, cout, scale,
                       shift, out, M);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_pt_attention(const afm_pt_attention_args* g, void* stream) {
    if (!g || !g->p || !g->qkv || !g->knn_idx || !g->out || g->n < 0) return AFM_E_BADARG;
    const int C = g->channels, KN = g->nsample, S = g->share_planes;
    if (C <= 0 || S <= 0 || C % S) return AFM_E_BADARG;
    const int CS = C / S;
    if (C > 512 || (KN != 8 && KN != 16) || CS > 64 || (64 % CS) != 0) return AFM_E_UNSUPPORTED;
    if (g->n == 0) return 0;
    PtAttnArgs a;
    a.p = g->p; a.qkv = g->qkv; a.knn = g->knn_idx; a.out = g->out; a.n = g->n; a.C = C; a.KN = KN; a.CS = CS;
    a.lp0_w = g->lp0_w; a.lp0_b = g->lp0_b; a.lp_s = g->lp_bn_scale; a.lp_t = g->lp_bn_shift; a.lp3_w = g->lp3_w; a.lp3_b = g->lp3_b;
    a.w0_s = g->w0_bn_scale; a.w0_t = g->w0_bn_shift; a.w2_w = g->w2_w; a.w2_b = g->w2_b;
    a.w3_s = g->w3_bn_scale; a.w3_t = g->w3_bn_shift; a.w5_w = g->w5_w; a.w5_b = g->w5_b;
    a.out_s = g->out_scale; a.out_t = g->out_shift; a.relu = g->relu;
    const int cpl = (C + 63) / 64;
    const int ne = (KN + (64 / CS) - 1) / (64 / CS);
    a.w2_lds = C <= 256 ? 1 : 0;
    const int nwv = C <= 256 ? 4 : 1;                // the 512-channel level has N/256 points per sample: one wave per block
    const size_t lds = (size_t)((a.w2_lds ? C * CS : 0) + CS * CS + nwv * (KN * (C + 1) + 2 * KN * CS)) * sizeof(float);
    int quads = (g->n + nwv - 1) / nwv;
    const dim3 grid(quads < 2048 ? quads : 2048), block(64 * nwv);
    hipStream_t s = (hipStream_t)stream;
    AfmProf prof(AFM_PROF_PTATTN, 2.0 * g->n * KN * ((double)C * CS + CS * CS + 5.0 * C), s);
#define AFM_PT(CPL_, KN_, NE_)                                                                                             \
    do {                                                                                                                   \
        if (lds > 64 * 1024)                                                                                               \
            (void)hipFuncSetAttribute((const void*)pt_attention_kernel<CPL_, KN_, NE_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                                 \
        hipLaunchKernelGGL((pt_attention_kernel<CPL_, KN_, NE_>), grid, block, lds, s, a);                                 \
    } while (0)
    if (KN == 8) {
        if (ne > 4 || cpl > 4 || !a.w2_lds) return AFM_E_UNSUPPORTED;
        if (cpl == 1 && ne <= 1) AFM_PT(1, 8, 1);
        else if (cpl <= 2 && ne <= 2) AFM_PT(2, 8, 2);
        else AFM_PT(4, 8, 4);
    } else {
        if (ne > 16 || cpl > 8) return AFM_E_UNSUPPORTED;
        if (cpl == 1 && ne <= 2) AFM_PT(1, 16, 2);
        else if (cpl <= 2 && ne <= 4) AFM_PT(2, 16, 4);
        else if (cpl <= 4 && ne <= 8) AFM_PT(4, 16, 8);
        else AFM_PT(8, 16, 16);
    }
#undef AFM_PT
    AFM_CHECK_LAUNCH();
    return 0;
}
