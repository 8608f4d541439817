// CDM / ContactPerceiver denoiser, the ROW-LESS form of the sampling loop (round 3; reference models/cdm.py:155-188,474-513, modules.py:234-661):
// the 256-wide rows the reference builds for every point - encoder keys / values, decoder queries, attention output, the MLP's input - are
// LayerNorms and linear maps of the point's K = 12 or 44 inputs [x_t | features | 1] and of its 16 decoder attention weights, so they are never
// generated: enc_point_kernel (the two latents' reduction over the points), lat_head_kernel (head of the latent chain), lat_dectables_kernel
// (the per-sample tables of a step) and dec_point_kernel (the whole decoder of a point).  Same function as the layer-by-layer form up to f32
// re-association (tests/test_gpu_cdm.py); the host side of the tables is afm/cdm.py:_weights().
#include "perceiver_internal.h"
#include "bf16split.h"

using namespace afm_cdm;

namespace {

// ---------------------------------------------------------------- the encoder's reduction over the points without rows
// The rows the two latents attend over are LayerNorm_kv of a linear map of the point's K inputs x = [x_t | features | 1 | 0..] (K = 12: the
// H3D variant, K = 44: 32 scene features per point, the HUMANISE variant), so, exactly as in the decoder (dec_point_kernel): var = x Qe x^T
// (step-invariant K x K form), score[q] = rstd (x . EU[:, q]) + const with EU = Ec U'^T (Ec = the adapter map G_enc minus its row means: afm_cdm_weights.enc_ec; K x 16 per
// sample and step, built in the prologue), and the attention-weighted sum of the normalised rows is linear in sum_n p[n, q] rstd[n] x[n]: a
// wave accumulates 16 x K numbers instead of 16 x 256 and never generates a row.  Per 16 points at K = 12: 10 MFMAs (16x16x4) instead of
// 176; the partial (max, sum, K-vector) records are merged and taken through v-proj and o-proj by lat_head_kernel.
constexpr int EP_WAVES = 8, EP_SPLIT = NPART / EP_WAVES;
template <int NKS>
__global__ __launch_bounds__(64 * EP_WAVES) void enc_point_kernel(afm_ln kvn, const float* __restrict__ u_text, const float* __restrict__ cu_text,
                                                                  const float* __restrict__ u_time, const float* __restrict__ cu_time,
                                                                  const int64_t* __restrict__ t, int n_t, int N, float* __restrict__ pm, float* __restrict__ pl,
                                                                  float* __restrict__ pacc12, const float* __restrict__ xt, int cd, const float* __restrict__ feat,
                                                                  int fd, const float* __restrict__ ec, const float* __restrict__ qee) {
    constexpr int NQ = 16, K = RowLess<NKS>::K, NT = RowLess<NKS>::NT;
    constexpr int UGS_LD = 260;
    __shared__ float EUs[K * 16], QEs[K * 16 * NT], ccs[16], trs[EP_WAVES][16];
    __shared__ __align__(16) float ugs[16 * UGS_LD];
    constexpr int LD = K + 1;
    __shared__ float tiles[EP_WAVES][16 * LD];
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, p16 = lane & 15, g = lane >> 4;
    float* tr = trs[wave];
    int64_t ti = t[b];
    ti = ti < 0 ? 0 : (ti >= n_t ? n_t - 1 : ti);
    const int per = (N + EP_SPLIT - 1) / EP_SPLIT;
    const int n0 = blockIdx.x * per, n1 = min(N, n0 + per);
    const int wper = ((per + EP_WAVES - 1) / EP_WAVES + 15) & ~15;      // points per wave, whole tiles
    const int w0 = n0 + wave * wper, w1 = min(n1, w0 + wper);
    // The inputs of a 16-point tile are two contiguous pieces of global memory (16 fd floats of the features, 16 cd of x_t): a wave reads
    // them as whole 256-byte lines, one tile ahead of the one it computes on (they are cold: the previous step's dec_point wrote x_t from
    // other XCDs), and re-shapes them through a wave-private [16][LD] tile in LDS into the two operand layouts - instead of 2 (NKS + 4 NT)
    // scattered dword loads per tile at the head of each tile's dependent chain.
    float* T = tiles[wave];
    constexpr int NLF = (16 * (K - 1) + 63) / 64, NLX = 2;        // floats per lane of a tile's features (fd <= K - 1) and contacts (cd <= 8)
    int of[NLF], ox[NLX];                                          // where this lane's j-th float goes in T (-1: nowhere)
#pragma unroll
    for (int j = 0; j < NLF; ++j) {
        const int e = lane + 64 * j, pt = e / fd, col = e - pt * fd;
        of[j] = (e < 16 * fd && col >= cd) ? pt * LD + col : -1;
    }
#pragma unroll
    for (int j = 0; j < NLX; ++j) {
        const int e = lane + 64 * j, pt = e / cd, col = e - pt * cd;
        ox[j] = e < 16 * cd ? pt * LD + col : -1;
    }
    for (int i = lane; i < 16 * LD; i += 64) T[i] = (i % LD) == fd ? 1.0f : 0.0f;      // the constant input and the padding: written once
    float fr[NLF], xr[NLX];
    const unsigned flast = (unsigned)(b * N + n1) * (unsigned)fd - 1u, xlast = (unsigned)(b * N + n1) * (unsigned)cd - 1u;
    auto fetch = [&](int nb) {                                     // clamped to the sample's last float: points past w1 are masked below
        const unsigned fb = (unsigned)(b * N + nb) * (unsigned)fd + lane, xb = (unsigned)(b * N + nb) * (unsigned)cd + lane;
#pragma unroll
        for (int j = 0; j < NLF; ++j) fr[j] = feat[min(fb + 64u * j, flast)];
#pragma unroll
        for (int j = 0; j < NLX; ++j) xr[j] = xt[min(xb + 64u * j, xlast)];
    };
    fetch(min(w0, n1 - 1));
    for (int i = threadIdx.x; i < K * 16 * NT; i += 64 * EP_WAVES) QEs[i] = qee[i];
    for (int q = wave; q < NQ; q += EP_WAVES) {                    // one wave per folded query: u' = gamma * u_q (staged for the dots below), beta . u_q
        const float* up = q < NQ / 2 ? u_text + ((int64_t)b * (NQ / 2) + q) * 256 : u_time + (ti * (NQ / 2) + (q - NQ / 2)) * 256;
        const float4 u = *reinterpret_cast<const float4*>(up + lane * 4), gm = *reinterpret_cast<const float4*>(kvn.g + lane * 4),
                     bt = *reinterpret_cast<const float4*>(kvn.b + lane * 4);
        *reinterpret_cast<float4*>(&ugs[q * UGS_LD + lane * 4]) = make_float4(u.x * gm.x, u.y * gm.y, u.z * gm.z, u.w * gm.w);
        const float d = wave_sum((u.x * bt.x + u.y * bt.y) + (u.z * bt.z + u.w * bt.w));
        if (lane == 0) ccs[q] = d + (q < NQ / 2 ? cu_text[(int64_t)b * (NQ / 2) + q] : cu_time[ti * (NQ / 2) + (q - NQ / 2)]);
    }
    __syncthreads();
    // EU[k][q] = Ec[k] . u'_q: one thread per dot, its 64 float4 of Ec all independent loads (one wave per query with a cross-lane sum per k
    // was K dependent L2 round trips: 68 -> 50 us for the kernel at K = 44, nothing at K = 12); 16 lanes share an Ec row (broadcast), the
    // staged u' rows are UGS_LD = 260 floats apart so that the 16 queries of a wave read 16 different bank groups
    for (int idx = threadIdx.x; idx < K * 16; idx += 64 * EP_WAVES) {
        const int k = idx >> 4, q = idx & 15;
        const float4* e4 = reinterpret_cast<const float4*>(ec + k * 256);
        const float4* u4 = reinterpret_cast<const float4*>(&ugs[q * UGS_LD]);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 16
        for (int j = 0; j < 64; ++j) {
            const float4 e = e4[j], u = u4[j];
            a0 = fmaf(e.x, u.x, a0); a1 = fmaf(e.y, u.y, a1); a2 = fmaf(e.z, u.z, a2); a3 = fmaf(e.w, u.w, a3);
        }
        EUs[idx] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    const float cconst = ccs[p16];


    f32x4 wacc[NT];                                               // lane (q = p16, g), tile t: sum_n p[n, q] rstd[n] x[n][k = 16 t + 4 g + r]
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) wacc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;                          // of query p16, replicated over g
    for (int nb = w0; nb < w1; nb += 16) {
        float xin[NKS], xT[NT][4];
#pragma unroll
        for (int j = 0; j < NLF; ++j)
            if (of[j] >= 0) T[of[j]] = fr[j];
#pragma unroll
        for (int j = 0; j < NLX; ++j)
            if (ox[j] >= 0) T[ox[j]] = xr[j];
        fetch(min(nb + 16, n1 - 1));
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) xin[ks] = T[p16 * LD + 4 * ks + g];           // lane (p, g): inputs 4 ks + g of point p
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) xT[tt][r] = (16 * tt + p16 < K) ? T[(4 * g + r) * LD + min(16 * tt + p16, K - 1)] : 0.f;      // lane (k = p16, g): input 16 t + k of point 4 g + r
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        float varq = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(xin[ks], EUs[(4 * ks + g) * 16 + p16], sc, 0, 0, 0);       // lane (q = p16, g) reg r: point 4 g + r
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            f32x4 yq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) yq = __builtin_amdgcn_mfma_f32_16x16x4f32(QEs[(4 * ks + g) * (16 * NT) + 16 * tt + p16], xin[ks], yq, 0, 0, 0);       // reg r: input 4 (4 t + r) + g
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * tt + r < NKS) varq += yq[r] * xin[4 * tt + r];
        }
        varq += xor16(varq); varq += xor32(varq);
        if (g == 0) tr[p16] = 1.0f / sqrtf(fmaxf(varq, 0.f) + 1e-5f);
        float rq[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) rq[r] = tr[4 * g + r];
        const int nvalid = w1 - nb;                                // points 4 g + r >= nvalid do not exist
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = (4 * g + r < nvalid) ? rq[r] * sc[r] + cconst : -INFINITY;
            mt = fmaxf(mt, sc[r]);
        }
        mt = fmaxf(mt, xor16(mt)); mt = fmaxf(mt, xor32(mt));
        const float mn = fmaxf(m_run, mt);                         // finite: every processed tile has a valid point
        const float alpha = __expf(m_run - mn);
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = __expf(sc[r] - mn); ls += sc[r]; }
        ls += xor16(ls); ls += xor32(ls);
        l_run = l_run * alpha + ls;
        m_run = mn;
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            wacc[tt][0] *= alpha; wacc[tt][1] *= alpha; wacc[tt][2] *= alpha; wacc[tt][3] *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) wacc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xT[tt][r], sc[r] * rq[r], wacc[tt], 0, 0, 0);
        }
    }
    const int part = blockIdx.x * EP_WAVES + wave;
    const int64_t base = ((int64_t)b * NPART + part) * NQ;
    if (g == 0) { pm[base + p16] = m_run; pl[base + p16] = l_run; }
#pragma unroll
    for (int tt = 0; tt < NT; ++tt)
        *reinterpret_cast<float4*>(pacc12 + (base + p16) * (16 * NT) + 16 * tt + 4 * g) = make_float4(wacc[tt][0], wacc[tt][1], wacc[tt][2], wacc[tt][3]);
}

// The head of the chain in one launch (fused sampling form): merge enc_point_kernel's partials of the 8 heads of a latent token into the
// 8 x K numbers a12[h][k] = sum_n a[n] rstd[n] x[n][k], then apply everything that is linear behind them at once -
//   x1 = q0 + o_proj(v_proj(gamma_kv * (a12 Ec) + beta_kv)) = q0 + c1 + sum_{h, k} a12[h][k] WOVE[K h + k]
// with WOVE [8 K][dq] = W_o (per-head blocks) W_v (gamma_kv * Ec)^T and c1 = b_o + W_o (W_v beta_kv + b_v) from the host (float64).
// Replaces lat_combine + the v-proj and o-proj toklin launches.  grid (B, 2 latents, dq / 256), block 256.
template <int NKS>
__global__ __launch_bounds__(256) void lat_head_kernel(const float* __restrict__ pm, const float* __restrict__ pl, const float* __restrict__ pacc12,
                                                       const float* __restrict__ q0_text, const float* __restrict__ q0_time,
                                                       const int64_t* __restrict__ t, int n_t, int dq, const float* __restrict__ wove,
                                                       const float* __restrict__ c1, float* __restrict__ x1) {
    constexpr int K = RowLess<NKS>::K, NT = RowLess<NKS>::NT;
    __shared__ __align__(16) float a12[8 * K];
    const int b = blockIdx.x, i = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // lane = one of the NPART = 64 partials; a wave merges heads wave and wave + 4.  Every load of both heads is issued before the first
    // reduction (they are cold in L2 - dec_point streamed the whole cloud since they were written - and were 22 dependent round trips otherwise)
    float mm[2], ll[2];
    float4 v[2][NKS];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int64_t base = ((int64_t)b * NPART + lane) * 16 + (i * 8 + wave + 4 * hh);
        mm[hh] = pm[base]; ll[hh] = pl[base];
        const float4* pa = reinterpret_cast<const float4*>(pacc12 + base * (16 * NT));
#pragma unroll
        for (int k4 = 0; k4 < NKS; ++k4) v[hh][k4] = pa[k4];
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int h = wave + 4 * hh;
        float M = mm[hh];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
        const float ww = (mm[hh] == -INFINITY) ? 0.f : __expf(mm[hh] - M);
        const float L = wave_sum(ll[hh] * ww);
        const float wq = ww * (1.0f / L);
#pragma unroll
        for (int k4 = 0; k4 < NKS; ++k4) {
            const float s0 = wave_sum(wq * v[hh][k4].x), s1 = wave_sum(wq * v[hh][k4].y), s2 = wave_sum(wq * v[hh][k4].z), s3 = wave_sum(wq * v[hh][k4].w);
            if (lane == 0) *reinterpret_cast<float4*>(&a12[h * K + 4 * k4]) = make_float4(s0, s1, s2, s3);
        }
    }
    __syncthreads();
    int64_t ti = t[b];
    ti = ti < 0 ? 0 : (ti >= n_t ? n_t - 1 : ti);
    const float* q0 = i == 0 ? q0_text + (int64_t)b * dq : q0_time + ti * dq;
    const int n = blockIdx.z * 256 + threadIdx.x;                 // one output per thread: the 96 loads of its column are independent
    if (n < dq) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        constexpr int CH = NKS <= 4 ? 8 * K : 2 * K;              // loads in flight per thread: 96 (all of them) / 88 (four rounds)
        for (int j0 = 0; j0 < 8 * K; j0 += CH) {
#pragma unroll
            for (int jj = 0; jj < CH; jj += 4) {
                const int j = j0 + jj;
                v0 += a12[j] * wove[(int64_t)j * dq + n]; v1 += a12[j + 1] * wove[(int64_t)(j + 1) * dq + n];
                v2 += a12[j + 2] * wove[(int64_t)(j + 2) * dq + n]; v3 += a12[j + 3] * wove[(int64_t)(j + 3) * dq + n];
            }
        }
        x1[((int64_t)b * 2 + i) * dq + n] = (q0[n] + c1[n]) + ((v0 + v1) + (v2 + v3));
    }
}

// ---------------------------------------------------------------- the whole decoder of a point in ONE kernel (written for K = 12 inputs; K = 44: the same with NT = 3 input tiles)
// After the attention weights a[p, jh] of a point are known, everything up to linear1 is LINEAR in u[p] = [a[p, 0..15] | x_t | features | 1]:
//   h1[p, c] = sum_jh P[jh, c] a[p, jh] + b_o[c] + e[p, c],  e = the generated query row       = sum_m u[p, m] T[m, c]   (T = [P ; G_dec (+ b_o)], G_dec = both adapters as one map of the inputs)
//   z = LayerNorm_mlp(h1): h1 - mean_c(h1) = sum_m u[p, m] Tc[m, c]  (Tc = T minus its row means),  var_c(h1) = u Qc u^T,  Qc = Tc Tc^T / 256
//   linear1(z)[n] = rstd[p] * sum_m u[p, m] TWc[m, n] + C[n],   TWc = Tc (W1 * gamma_mlp)^T,  C = b1 + W1 beta_mlp
// so the 256-wide rows h1 and z never exist and linear1 is a K = 28 product instead of K = 256 (x 9 bf16 products): per 16 points
// 112 + 14 f32 MFMAs against 64 (P V) + 576 (linear1 on the bf16 pipe).  The hidden row GELU(linear1) lives one 16-channel tile at a time
// and goes straight into the row-dots with w2 = contact_layer.w fc2.w (64 MFMAs, the contact channel as the output row), to which the
// attention part (WP a) and the query part (gen_qe . inputs) of contact_layer.w . h1 are added in the same accumulator: the kernel reads
// 9 floats per point and writes the 6 of x_0 / x_{t-1}.  Per sample and step the rows of P enter through one small launch in front,
// lat_dectables_kernel (the TWc rows of the attention weights, the [28 x 28] quadratic form in operand order, EG, the score constants, WP);
// the rows of the inputs are step-invariant and come from the host (afm_cdm_weights.dec_*).  Same function as the layer-by-layer form up
// to f32 re-association (tests/test_gpu_cdm.py).
// The attention scores fold the same way: LayerNorm_q(e) . G'[jh] = rstd_q (x . EG[:, jh]) + const with x = the 12 inputs, EG = Dc G'^T
// (Dc = G_dec minus its row means, per sample and step: lat_dectables_kernel) and var_q = x Qd x^T (step-invariant): the query row e is
// never generated either.
constexpr int DP_LDW = 260;

// All per-sample tables of dec_point_kernel straight from the decoder keys / values of the sample's two latents (lat_kv, 2 x 2 x 256 numbers) and
// step-invariant matrices (afm_cdm_weights.dec_*; o = 32 h + r runs over a head's 32 key / value entries, jh = 8 j + h, k = an input):
//   EG[k][jh]     = scd sum_r kd_j[o] DWQ[k][o]            gconst[jh] = scd sum_r kd_j[o] wqb[o]            WP[r'][jh] = sum_r vd_j[o] WCO[r'][o]
//   TWc[jh][n]    = sum_r vd_j[o] WOW[o][n]                (centred P rows times (W1 gamma)^T, P itself is never formed)
//   Qc[jh][jh']   = sum_{r, r'} vd_j[o] WoG[o][o'] vd_j'[o'],  Qc[jh][16 + k] = sum_r vd_j[o] XWO[k][o],  Qc[16 + k][16 + k'] = qxx
// Qc goes out in MFMA operand order (RowLess<NKS>): entry (cs, 16 t + i) = Qc[m'(t, i)][m(cs)] - K index cs < 16: attention weight cs, cs = 16 +
// XS g + ks: input 4 ks + g; output rows of tile 0: attention weights, of tile 1 + t: input 4 (4 t + (i & 3)) + (i >> 2).
// Replaces lat_decfold + the TWc launch of the fused form.  grid B, block 1024 (the two [256][256] products: thread = (column, pair of heads),
// 64 independent row loads each).
template <int NKS>
__global__ __launch_bounds__(1024) void lat_dectables_kernel(const float* __restrict__ lat_kv, const float* __restrict__ dwq, const float* __restrict__ wqb,
                                                            const float* __restrict__ wco, const float* __restrict__ wow, const float* __restrict__ wog,
                                                            const float* __restrict__ xwo, const float* __restrict__ qxx, int cd,
                                                            float* __restrict__ twp, float* __restrict__ tab) {
    // grid (B, 4): workgroup w of a sample owns the columns [64 w, 64 w + 64) of the two [256][256] products (a quarter of the 512 KB of weights
    // each: one workgroup per sample pulled all of it through ONE CU's vector-load path, 10.9 us), the attention-weight block of Qc whose second
    // index lies in heads 2 w, 2 w + 1, and every fourth entry of EG / the mixed block; workgroup 0 also owns the constants.  Every table entry is
    // written by exactly one workgroup.
    using RL = RowLess<NKS>;
    constexpr int K = RL::K, NQ = 16 + K;
    __shared__ float kd[2][256], vd[2][256];
    __shared__ float tv[16][65];
    __shared__ float Q[NQ][NQ + 1];
    const int b = blockIdx.x, w = blockIdx.y, tid = threadIdx.x;
    const float scd = 0.17677669529663687f;                       // 1 / sqrt(32)
    if (tid < 512) {
        const int j = tid >> 8, c = tid & 255;
        kd[j][c] = lat_kv[((int64_t)b * 2 + j) * 512 + c];
        vd[j][c] = lat_kv[((int64_t)b * 2 + j) * 512 + 256 + c];
    }
    __syncthreads();
    float* T = tab + (int64_t)b * RL::TAB;
    if (tid < 512) {   // column 64 w + cl of the two matrices against the value vectors of head h: TWc rows (to memory), tv = vd WoG (to LDS)
        const int cl = tid & 63, h = tid >> 6, c = 64 * w + cl;
        float t0 = 0.f, t1 = 0.f, g0 = 0.f, g1 = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int o = 32 * h + r;
            const float w1 = wow[o * 256 + c], w2 = wog[o * 256 + c], v0 = vd[0][o], v1 = vd[1][o];
            t0 += v0 * w1; t1 += v1 * w1;
            g0 += v0 * w2; g1 += v1 * w2;
        }
        twp[((int64_t)b * 16 + h) * 256 + c] = t0; twp[((int64_t)b * 16 + 8 + h) * 256 + c] = t1;
        tv[h][cl] = g0; tv[8 + h][cl] = g1;
    } else {           // meanwhile: this workgroup's quarter of EG [K][16] and of the attention-weight x input block of Qc (entries e = 4 i + w)
        for (int e = 4 * (tid - 512) + w; e < K * 16; e += 4 * 512) {
            const int k = e >> 4, jh = e & 15, j = jh >> 3, h = jh & 7;
            float eg = 0.f, q = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) { eg += kd[j][32 * h + r] * dwq[k * 256 + 32 * h + r]; q += vd[j][32 * h + r] * xwo[k * 256 + 32 * h + r]; }
            T[RL::QTAB + e] = eg * scd;
            Q[jh][16 + k] = q; Q[16 + k][jh] = q;
        }
        if (w == 0) {
            if (tid >= 1024 - 16) {
                const int jh = tid - (1024 - 16), j = jh >> 3, h = jh & 7;
                float gq = 0.f;
                for (int r = 0; r < 32; ++r) gq += kd[j][32 * h + r] * wqb[32 * h + r];
                T[RL::QTAB + K * 16 + jh] = gq * scd;
            }
            if (tid >= 512 + 256 && tid < 512 + 256 + 128) {      // WP [8][16]
                const int e = tid - (512 + 256), rr = e >> 4, jh = e & 15, j = jh >> 3, h = jh & 7;
                float wp = 0.f;
                if (rr < cd)
                    for (int r = 0; r < 32; ++r) wp += vd[j][32 * h + r] * wco[rr * 256 + 32 * h + r];
                T[RL::QTAB + K * 16 + 16 + e] = wp;
            }
            for (int e = tid - 512; e < K * K; e += 512) Q[16 + e / K][16 + e % K] = qxx[e];
        }
    }
    __syncthreads();
    if (tid < 64) {    // Qc[jh][jh2] = tv[jh] (head block of jh2) . vd_j2 for the four jh2 of this workgroup's heads
        const int jh = tid >> 2, j2 = (tid >> 1) & 1, hl = tid & 1, h2 = 2 * w + hl;
        float q = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) q += tv[jh][32 * hl + r] * vd[j2][32 * h2 + r];
        Q[jh][8 * j2 + h2] = q;
    }
    __syncthreads();
    for (int e = tid; e < RL::QTAB; e += 1024) {                  // operand order; an entry is written by the workgroup that holds its source
        const int cs = e / RL::LDQ, col = e - cs * RL::LDQ;
        int m = -1, m2 = -1;
        if (col < 16 + 16 * RL::NT) {
            const int i = col & 15, tt = (col >> 4) - 1, x = cs - 16, xg = x / RL::XS, xk = x - xg * RL::XS;
            m2 = col < 16 ? i : (4 * tt + (i & 3) < NKS ? 16 + 4 * (4 * tt + (i & 3)) + (i >> 2) : -1);
            m = cs < 16 ? cs : (xk < NKS ? 16 + 4 * xk + xg : -1);
        }
        const bool live = m >= 0 && m2 >= 0;
        int owner = 0;                                            // zeros and the input x input block (qxx): workgroup 0
        if (live && m < 16 && m2 < 16) owner = (m & 7) >> 1;      // Qc[jh = m2][jh2 = m]: the heads of jh2
        else if (live && (m < 16) != (m2 < 16)) owner = (m < 16 ? m : m2) & 3;      // mixed block, entry e = 16 k + jh: e & 3 = jh & 3
        if (owner == w) T[e] = live ? Q[m2][m] : 0.f;
    }
}

template <int NKS> constexpr int dp_lds_floats() {
    using RL = RowLess<NKS>;
    return 16 * RL::NSTEP * 3 * 64 * 4 + RL::TAB + 8 * DP_LDW + 256 + RL::K * 16 * RL::NT + 8 * 16 * RL::NT + 16 + RL::NW * 16 * 17 + RL::NW * 16;
}

// NPROD (round 6): cross products of linear1's three-term bf16 operands - 9 (all, exact f32 products) or 6 (the six largest: the arithmetic of
// afm_linear's AFM_ARITH_BF16X6 / AFM_ARITH_DEFAULT, tests/test_gpu_arith.py), chosen by afm_cdm_weights.gemm_arith like every GEMM of the denoiser.
template <int NKS, int NPROD>
__global__ __launch_bounds__(64 * RowLess<NKS>::NW, NKS <= 4 ? 2 : 1)
void dec_point_kernel(const float* __restrict__ twp, const float* __restrict__ qtab, const float* __restrict__ qdd, const float* __restrict__ twx,
                      const float* __restrict__ cvec, const float* __restrict__ w2f, const float* __restrict__ gen_qe, const float* __restrict__ c0, int N, int cd,
                      const float* xt, const float* __restrict__ feat, int fd, float* __restrict__ x0_out, const float* __restrict__ noise, float* x_next,
                      const float* __restrict__ c1, const float* __restrict__ c2, const float* __restrict__ sigma, int clip) {
    using RL = RowLess<NKS>;
    constexpr int K = RL::K, NT = RL::NT, LDQ = RL::LDQ, XS = RL::XS, NSTEP = RL::NSTEP, NW = RL::NW, NTH = 64 * NW, QEW = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) float dp_sm[];
    // linear1 runs on the bf16 pipe with the exact three-way split (csrc/bf16split.h): the f32 MFMA issues at the vector rate and does not
    // overlap with the VALU work of the GELUs (122 us with all products in f32: VALU + f32 MFMA cycles add up), v_mfma_f32_16x16x32_bf16 does.
    // Operand order of its K = 32 steps: lane group g carries k = {a[4 g .. 4 g + 3], x[ks = 0 .. 3]} in step 0 and x[ks = 4 .. 11] in step 1 (x[ks] =
    // input 4 ks + g, zeros past NKS), i.e. what lane (p, g) already holds.
    uint4* TWP = reinterpret_cast<uint4*>(dp_sm);                 // [16 tiles][NSTEP][3 planes][64 lanes] 8 bf16: TWc rows in that order, split once per workgroup
    float* Qs = dp_sm + 16 * NSTEP * 3 * 64 * 4;                  // [QCOL][LDQ] quadratic form of the MLP's LayerNorm, operand order (lat_dectables_kernel)
    float* EGs = Qs + RL::QTAB;                                   // [K][16]    scores: row k = input, column jh
    float* gcs = EGs + K * 16;                                    // [16]       beta_q . G[jh] + cb[jh]
    float* WPs = gcs + 16;                                        // [8][16]    contact_layer.w . P
    float* W2s = WPs + 8 * 16;                                    // [8][LDW]   contact_layer.w fc2.w  (rows >= cd: 0)
    float* Cv = W2s + 8 * DP_LDW;                                 // [256]      b1 + W1 beta_mlp
    float* QDs = Cv + 256;                                        // [K][16 NT] quadratic form of the query's LayerNorm, operand order (host)
    float* QEs = QDs + K * 16 * NT;                               // [8][16 NT] contact_layer.w . G_dec^T  (columns >= feat_dim + 1: 0)
    float* c0s = QEs + 8 * QEW;                                   // [16]
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, p16 = lane & 15, g = lane >> 4;
    float* aT = c0s + 16 + wave * 16 * 17;                        // [16 points][17] attention weights of the tile, transposed
    float* tr = c0s + 16 + NW * 16 * 17 + wave * 16;              // [16] a per-point scalar from lanes (p, .) to lanes (., g)
    for (int it = threadIdx.x; it < 16 * NSTEP * 64; it += NTH) { // (channel tile, K32 step, lane) items: 8 operand values -> three planes
        const int l = it & 63, st = (it >> 6) % NSTEP, tt = it / (64 * NSTEP), n = 16 * tt + (l & 15), gg = l >> 4;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int slot = 8 * st + e;                           // slots 0..3: attention weights 4 g + e; slot 4 + ks: input 4 ks + g
            v[e] = slot < 4 ? twp[((int64_t)b * 16 + 4 * gg + slot) * 256 + n] : (slot - 4 < NKS ? twx[(4 * (slot - 4) + gg) * 256 + n] : 0.f);
        }
        uint4 p1, p2, p3;
        split2(v[0], v[1], p1.x, p2.x, p3.x); split2(v[2], v[3], p1.y, p2.y, p3.y); split2(v[4], v[5], p1.z, p2.z, p3.z); split2(v[6], v[7], p1.w, p2.w, p3.w);
        uint4* d = TWP + ((tt * NSTEP + st) * 3) * 64 + l;
        d[0] = p1; d[64] = p2; d[128] = p3;
    }
    for (int i = threadIdx.x; i < RL::TAB; i += NTH) Qs[i] = qtab[(int64_t)b * RL::TAB + i];      // Qs | EGs | gcs | WPs are contiguous, like the table
    for (int i = threadIdx.x; i < 8 * 256; i += NTH) W2s[(i >> 8) * DP_LDW + (i & 255)] = (i >> 8) < cd ? w2f[i] : 0.f;
    for (int i = threadIdx.x; i < 256; i += NTH) Cv[i] = cvec[i];
    for (int i = threadIdx.x; i < K * 16 * NT; i += NTH) QDs[i] = qdd[i];
    for (int i = threadIdx.x; i < 8 * QEW; i += NTH) {
        const int j = i / QEW, k = i - j * QEW;
        QEs[i] = (j < cd && k < K) ? gen_qe[j * K + k] : 0.f;
    }
    if (threadIdx.x < 16) c0s[threadIdx.x] = (int)threadIdx.x < cd ? c0[threadIdx.x] : 0.f;
    __syncthreads();
    const float gconst = gcs[p16];

    const int per = (N + gridDim.x - 1) / gridDim.x;
    const int n0 = blockIdx.x * per, n1 = min(N, n0 + per);
    const int wper = ((per + NW - 1) / NW + 15) & ~15;            // points per wave, whole tiles
    const int w0 = n0 + wave * wper, w1 = min(n1, w0 + wper);

    float xin[NKS], xnext[NKS];
    auto fetch = [&](int nb, float (&dst)[NKS]) {                  // inputs k = 4 ks + g of point nb + p16
        const unsigned pti = (unsigned)(b * N + min(nb + p16, n1 - 1));
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            const int k = 4 * ks + g;
            // one load per input, from the array that holds it (round 6: both arrays were read unconditionally and one value selected - twice the
            // requests and 34.5 MB of traffic per launch against 15.7 MB algorithmic; the lanes of a group take the same branch)
            float v = k == fd ? 1.0f : 0.0f;
            if (k < cd) v = xt[pti * (unsigned)cd + (unsigned)k];
            else if (k < fd) v = feat[pti * (unsigned)fd + (unsigned)k];
            dst[ks] = v;
        }
    };
    if (w0 < w1) fetch(w0, xnext);
    for (int nb = w0; nb < w1; nb += 16) {
        const int64_t pt = (int64_t)b * N + nb + p16;
        const bool pvalid = nb + p16 < w1;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) xin[ks] = xnext[ks];
        if (nb + 16 < w1) fetch(nb + 16, xnext);
        // ---- LayerNorm statistics of the (never generated) query row: var_q = x Qd x^T; scores = rstd_q (x . EG) + const; softmax over the
        // two keys of a head (jh and jh ^ 8: eight lanes apart)
        f32x4 sc = {0.f, 0.f, 0.f, 0.f};
        float varq = 0.f;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) sc = __builtin_amdgcn_mfma_f32_16x16x4f32(xin[ks], EGs[(4 * ks + g) * 16 + p16], sc, 0, 0, 0);       // lane (jh = p16, g) reg r: point 4 g + r
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            f32x4 yq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) yq = __builtin_amdgcn_mfma_f32_16x16x4f32(QDs[(4 * ks + g) * (16 * NT) + 16 * tt + p16], xin[ks], yq, 0, 0, 0);       // reg r: input 4 (4 t + r) + g
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * tt + r < NKS) varq += yq[r] * xin[4 * tt + r];
        }
        varq += xor16(varq); varq += xor32(varq);
        if (g == 0) tr[p16] = 1.0f / sqrtf(fmaxf(varq, 0.f) + 1e-5f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s_own = tr[4 * g + r] * sc[r] + gconst, s_oth = lane_xor<8>(s_own);
            const float mx = fmaxf(s_own, s_oth);
            const float e_own = __expf(s_own - mx), e_oth = __expf(s_oth - mx);
            aT[(4 * g + r) * 17 + p16] = e_own / (e_own + e_oth);
        }
        float aB[4];
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) aB[sI] = aT[p16 * 17 + 4 * g + sI];       // lane (p = p16, g): a[p, jh = 4 g + s]
        // ---- variance of the MLP's LayerNorm input: u Qc u^T (y = Qc u on the matrix pipe, the dot with u in the lane + across g)
        float var = 0.f;
#pragma unroll
        for (int tt = 0; tt < 1 + NT; ++tt) {                      // output tile 0: attention weights; 1 + t: inputs 4 (4 t + r) + g
            f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) y = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[(4 * g + sI) * LDQ + 16 * tt + p16], aB[sI], y, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) y = __builtin_amdgcn_mfma_f32_16x16x4f32(Qs[(16 + XS * g + ks) * LDQ + 16 * tt + p16], xin[ks], y, 0, 0, 0);
            if (tt == 0) var += (y[0] * aB[0] + y[1] * aB[1]) + (y[2] * aB[2] + y[3] * aB[3]);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * (tt - 1) + r < NKS) var += y[r] * xin[4 * (tt - 1) + r];
            }
        }
        var += xor16(var); var += xor32(var);
        const float rstd2 = 1.0f / sqrtf(fmaxf(var, 0.f) + 1e-5f);
        // ---- linear1 (K = 16 + 4 NKS) -> GELU -> row-dots with w2; then the attention and query parts of contact_layer.w . h1.
        // TG 16-channel tiles at a time, phase by phase (operand reads, interleaved MFMA chains, independent GELUs, separate row-dot accumulators):
        // tile by tile the wave would sit through an LDS round trip, a dependent MFMA chain and a GELU dependency chain per tile.
        uint4 ub[NSTEP][3];                                        // u = [a | x] of this lane's point, three bf16 planes in linear1's operand order
        {
            float v[8 * NSTEP];
#pragma unroll
            for (int e = 0; e < 8 * NSTEP; ++e) v[e] = e < 4 ? aB[e] : (e - 4 < NKS ? xin[e - 4 < NKS ? e - 4 : 0] : 0.f);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                split2(v[8 * st + 0], v[8 * st + 1], ub[st][0].x, ub[st][1].x, ub[st][2].x); split2(v[8 * st + 2], v[8 * st + 3], ub[st][0].y, ub[st][1].y, ub[st][2].y);
                split2(v[8 * st + 4], v[8 * st + 5], ub[st][0].z, ub[st][1].z, ub[st][2].z); split2(v[8 * st + 6], v[8 * st + 7], ub[st][0].w, ub[st][1].w, ub[st][2].w);
            }
        }
        constexpr int TG = NSTEP == 1 ? 4 : 2;                     // channel tiles in flight (registers: TG x NSTEP x 3 operand vectors)
        f32x4 sa[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) sa[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t4 = 0; t4 < 16; t4 += TG) {
            uint4 wp[TG][NSTEP][3];
            float4 cv[TG], w4[TG];
#pragma unroll
            for (int q = 0; q < TG; ++q) {
                const int tt = t4 + q;
#pragma unroll
                for (int st = 0; st < NSTEP; ++st)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) wp[q][st][pl] = TWP[((tt * NSTEP + st) * 3 + pl) * 64 + lane];
                cv[q] = *reinterpret_cast<const float4*>(Cv + 16 * tt + 4 * g);
                w4[q] = p16 < 8 ? *reinterpret_cast<const float4*>(W2s + p16 * DP_LDW + 16 * tt + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 acc[TG];
#pragma unroll
            for (int q = 0; q < TG; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < NSTEP; ++st)
#pragma unroll
                for (int pq = 9 - NPROD; pq < 9; ++pq)
#pragma unroll
                    for (int q = 0; q < TG; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wp[q][st][AFM_PA[pq]]), __builtin_bit_cast(bf16x8, ub[st][AFM_PB[pq]]), acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            float hid[TG][4];
#pragma unroll
            for (int q = 0; q < TG; ++q) {
                hid[q][0] = gelu_erf_fast(rstd2 * acc[q][0] + cv[q].x); hid[q][1] = gelu_erf_fast(rstd2 * acc[q][1] + cv[q].y);
                hid[q][2] = gelu_erf_fast(rstd2 * acc[q][2] + cv[q].z); hid[q][3] = gelu_erf_fast(rstd2 * acc[q][3] + cv[q].w);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < TG; ++q) sa[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[q].x, hid[q][0], sa[q & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < TG; ++q) sa[(q + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[q].y, hid[q][1], sa[(q + 2) & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < TG; ++q) sa[q & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[q].z, hid[q][2], sa[q & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < TG; ++q) sa[(q + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[q].w, hid[q][3], sa[(q + 2) & 3], 0, 0, 0);
        }
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) sa[sI & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p16 < 8 ? WPs[p16 * 16 + 4 * g + sI] : 0.f, aB[sI], sa[sI & 1], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) sa[2 + (ks & 1)] = __builtin_amdgcn_mfma_f32_16x16x4f32(p16 < 8 ? QEs[p16 * QEW + 4 * ks + g] : 0.f, xin[ks], sa[2 + (ks & 1)], 0, 0, 0);
        const f32x4 sat = (sa[0] + sa[1]) + (sa[2] + sa[3]);
        if (pvalid) {                                              // lane (point p16, g): contact channels 4 g + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = 4 * g + r;
                if (j < cd) {
                    const int64_t i = pt * cd + j;
                    float v = sat[r] + c0s[j];
                    if (clip) v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);      // clip_denoised
                    if (x0_out) x0_out[i] = v;
                    if (x_next) x_next[i] = (c1[b] * v + c2[b] * xt[i]) + sigma[b] * noise[i];
                }
            }
        }
    }
}

template <int NKS>
int launch_dec_point_t(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, const float* feat, float* x0_out,
                       const afm_ddpm_args* ddpm, hipStream_t s, bool with_tables);
template <int NKS>
int launch_dec_tables_t(const afm_cdm_weights& w, int B, const CdmWs& ws, hipStream_t s) {
    hipLaunchKernelGGL(lat_dectables_kernel<NKS>, dim3(B, 4), dim3(1024), 0, s, ws.lat_kv, w.dec_dwq, w.dec_wqb, w.dec_wco, w.dec_wow, w.dec_wog, w.dec_xwo,
                       w.dec_qxx, w.contact_dim, ws.twp, ws.qtab);
    AFM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

namespace afm_cdm {

int launch_enc_point(const afm_cdm_weights& w, const float* text_u, const float* text_cu, const int64_t* t, int B, int N, const CdmWs& ws,
                     const float* x_t, const float* feat, hipStream_t s) {
    AfmProf prof(AFM_PROF_CDM_ENC, 0.0, s);           // K-vector partials in ws.pacc (stride 16 NT)
    if (rowless_nks(w.feat_dim) == 3)
        hipLaunchKernelGGL(enc_point_kernel<3>, dim3(EP_SPLIT, B), dim3(64 * EP_WAVES), 0, s, w.enc_kv_norm, text_u, text_cu, w.time_u, w.time_cu, t, w.n_timesteps, N,
                           ws.pm, ws.pl, ws.pacc, x_t, w.contact_dim, feat, w.feat_dim, w.enc_ec, w.enc_qee);
    else
        hipLaunchKernelGGL(enc_point_kernel<11>, dim3(EP_SPLIT, B), dim3(64 * EP_WAVES), 0, s, w.enc_kv_norm, text_u, text_cu, w.time_u, w.time_cu, t, w.n_timesteps, N,
                           ws.pm, ws.pl, ws.pacc, x_t, w.contact_dim, feat, w.feat_dim, w.enc_ec, w.enc_qee);
    AFM_CHECK_LAUNCH();
    return 0;
}

// head of the latent chain: combine + v-proj + o-proj as one launch (called by cdm_latent_chain, inside its profile bracket)
int launch_lat_head(const afm_cdm_weights& w, const float* text_q0, const int64_t* t, const CdmWs& ws, int B, hipStream_t s) {
    const int dq = w.dq;
    if (rowless_nks(w.feat_dim) == 3)
        hipLaunchKernelGGL(lat_head_kernel<3>, dim3(B, 2, (dq + 255) / 256), dim3(256), 0, s, ws.pm, ws.pl, ws.pacc, text_q0, w.time_q0, t, w.n_timesteps, dq, w.enc_wove, w.enc_c1, ws.lat_x);
    else
        hipLaunchKernelGGL(lat_head_kernel<11>, dim3(B, 2, (dq + 255) / 256), dim3(256), 0, s, ws.pm, ws.pl, ws.pacc, text_q0, w.time_q0, t, w.n_timesteps, dq, w.enc_wove, w.enc_c1, ws.lat_x);
    AFM_CHECK_LAUNCH();
    return 0;
}

// the fused decoder (mode 3): the per-sample tables of the step (one launch), then one kernel over the points
int launch_dec_point(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, const float* feat, float* x0_out,
                     const afm_ddpm_args* ddpm, hipStream_t s, bool with_tables) {
    if (with_tables) AFM_TRY(launch_dec_tables(w, B, ws, s));        // (its own profile bracket: the tables belong to the chain's time)
    AfmProf prof(AFM_PROF_CDM_DEC, 0.0, s);
    return rowless_nks(w.feat_dim) == 3 ? launch_dec_point_t<3>(w, B, N, ws, x_t, feat, x0_out, ddpm, s, false)
                                        : launch_dec_point_t<11>(w, B, N, ws, x_t, feat, x0_out, ddpm, s, false);
}

int launch_dec_tables(const afm_cdm_weights& w, int B, const CdmWs& ws, hipStream_t s) {
    AfmProf prof(AFM_PROF_CDM_CHAIN, 0.0, s);
    return rowless_nks(w.feat_dim) == 3 ? launch_dec_tables_t<3>(w, B, ws, s) : launch_dec_tables_t<11>(w, B, ws, s);
}

}  // namespace afm_cdm

namespace {

template <int NKS, int NPROD>
int launch_dec_point_p(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, const float* feat, float* x0_out,
                       const afm_ddpm_args* ddpm, hipStream_t s, bool with_tables) {
    constexpr int LDS = dp_lds_floats<NKS>() * (int)sizeof(float);
    static_assert(LDS <= 160 * 1024, "dec_point_kernel's tables fit the LDS");
    static const int attr = []() { return (int)hipFuncSetAttribute((const void*)dec_point_kernel<NKS, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); }();
    if (attr != 0) return attr;
    if (with_tables) AFM_TRY(launch_dec_tables_t<NKS>(w, B, ws, s));
    int chunks = (N + 511) / 512;                     // 512 points per workgroup: the per-sample tables are staged once per 8 tiles and wave
    const int cap = (w.flags >> AFM_CDM_DEC_CHUNKS_SHIFT) & 0x3F;      // tuning knob (sub-batch streams: half a batch wants twice the chunks to keep two workgroups per CU)
    if (cap) chunks = cap;
    else if (chunks > 16) chunks = 16;                // (three workgroups per CU measured slower: the kernel is bound by VALU + f32 MFMA issue, not by latency)
    if (chunks > (N + 15) / 16) chunks = (N + 15) / 16;
    hipLaunchKernelGGL((dec_point_kernel<NKS, NPROD>), dim3(chunks, B), dim3(64 * RowLess<NKS>::NW), LDS, s, ws.twp, ws.qtab, w.dec_qdd, w.dec_twx, w.dec_c,
                       w.fold_w2, w.gen_qe, w.fold_c0, N, w.contact_dim, x_t, feat, w.feat_dim, x0_out, ddpm ? ddpm->noise : nullptr,
                       ddpm ? ddpm->x_next : nullptr, ddpm ? ddpm->c1 : nullptr, ddpm ? ddpm->c2 : nullptr, ddpm ? ddpm->sigma : nullptr,
                       (ddpm && (w.flags & AFM_CDM_CLIP_X0)) ? 1 : 0);
    AFM_CHECK_LAUNCH();
    return 0;
}

template <int NKS>
int launch_dec_point_t(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, const float* feat, float* x0_out,
                       const afm_ddpm_args* ddpm, hipStream_t s, bool with_tables) {
    // six products for AFM_ARITH_DEFAULT / AFM_ARITH_BF16X6, all nine otherwise: a function of the pack's arithmetic only
    if (w.gemm_arith == AFM_ARITH_DEFAULT || w.gemm_arith == AFM_ARITH_BF16X6) return launch_dec_point_p<NKS, 6>(w, B, N, ws, x_t, feat, x0_out, ddpm, s, with_tables);
    return launch_dec_point_p<NKS, 9>(w, B, N, ws, x_t, feat, x0_out, ddpm, s, with_tables);
}

}  // namespace
