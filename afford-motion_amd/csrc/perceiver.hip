// CDM / ContactPerceiver denoiser (reference models/cdm.py:155-188,474-513; Perceiver-IO blocks of
// models/modules.py:234-661).
//
// Per sample: N = 8192 points x 256 channels on the key/value side, but only TWO latent queries.
//   latent_token : (hoisted: once per text / once per timestep) enc_q0 row = adapter(input), q = q_proj(LN(enc_q0)) and the
//                  FOLDED queries u[h] = W_k[h]^T q[h] (8 vectors of 256 per latent), c[h] = q[h].b_k[h]
//   (afm_linear) : enc_kv = encoder_adapter(feat)                                         [B*N, 256]
//   enc_reduce   : flash-style reduction over the points: scores = LN_kv(enc_kv).u + c, online softmax,
//                  s[i,h] = sum_n a_n LN_kv(enc_kv_n)  -> per-wave partials (m, l, s)   (K / V never exist)
//   latent_post  : combine partials, o = W_v s + b_v, o_proj, residual, MLP, 2 self-attention layers on the 2
//                  latents, then the decoder's folded keys/values G[j,h] = W_q[h]^T k[j,h], P[j,h] = W_o[:,h] v[j,h]
//   (afm_linear) : dec_q0 = decoder_adapter(enc_kv)                                        [B*N, 256]
//   dec_attend   : per point: qn = LN(dec_q0); 16 scores qn.G + cb; softmax over the 2 keys per head; attention
//                  output = sum a P + b_o; + residual; LN of the MLP -> h1, z
//   (afm_linear) : t = GELU(fc1 z), h2 = fc2 t + h1, out = contact_layer h2 (+ fused DDPM update)
// The three dense 256x256 per-point layers are the FLOPs (103 of the 115 GFLOP/step folded work at B = 32)
// and run on the f32-MFMA GEMM; the kernels here are streaming / latency kernels (one wave per point).
#include "common.h"
#include "profile.h"
#include "bf16split.h"

extern "C" int afm_linear(const afm_linear_args*, void*);

namespace {

typedef float pf32x2 __attribute__((ext_vector_type(2)));

constexpr int NSPLIT = 16;          // workgroups per sample in enc_reduce (x4 waves = 64 partials per sample)
constexpr int NPART = NSPLIT * 4;
constexpr int MAXD = 512;           // dq upper bound for the latent kernels' LDS vectors
// per-sample record written by latent_post: G [njh][256] | P [njh][256] | cb [njh] | WP [8][njh] (contact_layer.w . P, folded form)
#define DEC_LAT_STRIDE(njh) (2 * (njh) * 256 + (njh) + 8 * (njh))

inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

// ---------------------------------------------------------------- small device helpers (latent kernels)
// out[tok][o] = b[o] + sum_k W[o][k] * in[tok][k]   for 2 tokens.  Each wave takes 4 output rows at a time
// (8 independent accumulators, float4 weight loads in flight for all 4 rows) and finishes them with ONE
// 8-value halving reduction.  ind must be a multiple of 4; W rows 16-byte aligned.
__device__ void matvec2(const float* __restrict__ W, const float* __restrict__ b, const float* in, float* out, int outd, int ind,
                        int in_stride, int out_stride) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int o0 = wave * 4; o0 < outd; o0 += nw * 4) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = 0.f;
        for (int k = lane * 4; k < ind; k += 256) {
            const float4 x0 = *reinterpret_cast<const float4*>(in + k);
            const float4 x1 = *reinterpret_cast<const float4*>(in + in_stride + k);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = min(o0 + r, outd - 1);
                const float4 wv = *reinterpret_cast<const float4*>(W + (int64_t)o * ind + k);
                a[r] += (wv.x * x0.x + wv.y * x0.y) + (wv.z * x0.z + wv.w * x0.w);
                a[4 + r] += (wv.x * x1.x + wv.y * x1.y) + (wv.z * x1.z + wv.w * x1.w);
            }
        }
        const float tot = wave_reduce_multi<8>(a, lane);
        if ((lane & 7) == 0) {                       // one lane per owned index: idx = tok * 4 + r
            const int idx = multi_owned_index<8>(lane), tok = idx >> 2, o = o0 + (idx & 3);
            if (o < outd) out[tok * out_stride + o] = tot + (b ? b[o] : 0.f);
        }
    }
}

// LayerNorm of 2 tokens (waves 0 and 1), eps 1e-5
__device__ void ln2(const float* in, float* out, afm_ln p, int dim, int stride) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 2) {
        const float* x = in + wave * stride;
        float s = 0.f;
        for (int k = lane; k < dim; k += 64) s += x[k];
        const float mean = wave_sum(s) / dim;
        float q = 0.f;
        for (int k = lane; k < dim; k += 64) { const float d = x[k] - mean; q += d * d; }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / dim + 1e-5f);
        for (int k = lane; k < dim; k += 64) out[wave * stride + k] = (x[k] - mean) * rstd * p.g[k] + p.b[k];
    }
}

// ---------------------------------------------------------------- latent_token
// One latent token per workgroup (grid = number of tokens, block 1024): adapter -> enc_q0 row, LN_q, q_proj, dp_scale,
// and the folded queries u[h][c] = sum_r W_k[h*hd + r][c] q[h*hd + r], cu[h] = q_h . b_k[h].
// Both latents are per-step invariant given their input: the text token depends on the sample's text only (once per
// sampling run) and the time token on t only (tabulated for every timestep when the weights are packed), so this
// kernel is OFF the per-step path; the per-step kernels gather its outputs.
__global__ __launch_bounds__(1024) void latent_token_kernel(const afm_cdm_weights w, const float* __restrict__ in, int in_dim,
                                                           afm_lin adapter, float* __restrict__ q0_out, float* __restrict__ u_out,
                                                           float* __restrict__ cu_out) {
    __shared__ __attribute__((aligned(16))) float vin[2][MAXD], q0[2][MAXD], qn[2][MAXD], q[2][MAXD];
    const int tok = blockIdx.x, dq = w.dq, dkv = w.dkv, He = w.enc_heads, hd = dq / He;
    for (int i = threadIdx.x; i < MAXD; i += blockDim.x) { vin[0][i] = i < in_dim ? in[(int64_t)tok * in_dim + i] : 0.f; vin[1][i] = 0.f; }
    __syncthreads();
    matvec2(adapter.w, adapter.b, &vin[0][0], &q0[0][0], dq, in_dim, MAXD, MAXD);      // row 1 is a dummy token
    __syncthreads();
    for (int i = threadIdx.x; i < dq; i += blockDim.x) q0_out[(int64_t)tok * dq + i] = q0[0][i];
    ln2(&q0[0][0], &qn[0][0], w.enc_q_norm, dq, MAXD);
    __syncthreads();
    matvec2(w.enc_attn.q.w, w.enc_attn.q.b, &qn[0][0], &q[0][0], dq, dq, MAXD, MAXD);
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)hd);                                          // q * dp_scale (modules.py:330)
    for (int e = threadIdx.x; e < He * dkv; e += blockDim.x) {
        const int h = e / dkv, c = e % dkv;
        float acc = 0.f;
        for (int r = 0; r < hd; ++r) acc += w.enc_attn.k.w[(int64_t)(h * hd + r) * dkv + c] * (q[0][h * hd + r] * scale);
        u_out[((int64_t)tok * He + h) * dkv + c] = acc;
    }
    for (int h = threadIdx.x; h < He; h += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < hd; ++r) acc += w.enc_attn.k.b[h * hd + r] * (q[0][h * hd + r] * scale);
        cu_out[(int64_t)tok * He + h] = acc;
    }
}

// ---------------------------------------------------------------- enc_reduce on the matrix pipe
// The form above spends ~500 issue slots per point and wave on two contractions a matrix core does natively: the 16 scores of a
// point (LN(e_n) . u_q over 256 channels) and the weighted row sums (sum_n p_nq LN(e_n)).  Here a wave takes 16 points at a time:
//   layout A  lane (p = lane & 15, g = lane >> 4) holds channels {16 j + 4 g + e} of point p (16 float4 loads straight from memory);
//   LayerNorm statistics on the VALU in layout A (64 values per lane + two cross-lane steps), y = (e - mean) * rstd WITHOUT the affine
//             part: gamma is folded into the queries (u'_q = gamma * u_q), beta into the score constants and into the final sums;
//   scores    S[p, q] = sum_c y[p, c] u'[q, c]: 64 x v_mfma_f32_16x16x4_f32, A = y and B = u' both in layout A (k = channel);
//   softmax   online over the point tiles in the accumulator layout (lane (q, g) holds points 4 g + r): tile maximum / sum across g;
//   sums      Acc[q, c] += sum_p P[p, q] y[p, c]: 64 MFMAs, A = P (the score accumulators as they are), B = y with the POINT as k:
//             the one transposition of the tile, through a per-wave LDS tile [16][260] (conflict-free b128 writes, b32 reads).
// The next tile's rows are fetched while the second product runs.  ~40 VALU + 8 MFMA per point instead of ~320 VALU + ~170 SALU.
// Exact f32 products, f32 accumulation (v_mfma_f32_16x16x4_f32); a re-association of the same arithmetic (tests: 2e-5 vs the VALU form).
constexpr int ERM_WAVES = 8, ERM_SPLIT = NPART / ERM_WAVES;       // workgroups per sample x waves = NPART partials, as in the VALU form
constexpr int ERM_LDY = 260;
constexpr int ERM_LDS_FLOATS = ERM_WAVES * 16 * ERM_LDY + 16 * ERM_LDY + 12 * 256 + 16 + ERM_WAVES * 16;

// One workgroup of 8 waves per CU (158 KB of LDS: eight transposition tiles, the 16 folded queries u' = gamma * u_q shared by the
// waves - they are all of one sample -, the contact columns of the adapter); registers: 64 (rows) + 64 (sums) per lane.
// MODE 0: rows read from `enc_kv`; MODE 1 (FOLD): rows = enc_kv[n] + sum_j x_t[n, j] xu[j] (step-invariant part materialised once per
// loop).  (Inputs of at most 11 channels do not come here at all: enc_point_kernel.)
// The row-less forms see a point as 4 NKS inputs [x_t | features | 1 | 0 ...]: NKS = 3 (12 inputs: the H3D variant's 9 channels) or 11 (44: the
// HUMANISE variant's 41).  Lane (p, g) of a 16-point tile holds inputs 4 ks + g, ks < NKS.  An MFMA output indexed by inputs has NT 16-row tiles;
// row i of tile t stands for input 4 (4 t + (i & 3)) + (i >> 2), so that register r of lane (p, g) meets the lane's own input 4 (4 t + r) + g.
template <int NKS> struct RowLess {
    static_assert(NKS == 3 || NKS == 11, "NKS");
    static constexpr int K = 4 * NKS;                             // inputs, zero-padded
    static constexpr int NT = (K + 15) / 16;                      // input tiles of an MFMA output
    static constexpr int XS = NKS <= 4 ? 4 : 12;                  // operand-order column of input 4 ks + g: 16 + XS g + ks (XS LDQ = 16 mod 32: no bank conflicts)
    static constexpr int QCOL = 16 + 4 * XS;                      // K index of the decoder's variance form: 16 attention weights, then the inputs
    static constexpr int LDQ = 16 + 16 * NT + 4;                  // its output columns (one tile of attention weights + NT input tiles), padded
    static constexpr int QTAB = QCOL * LDQ;
    static constexpr int TAB = QTAB + K * 16 + 16 + 8 * 16;       // + EG [K][16] + gconst [16] + WP [8][16]: a sample's table of the step (lat_dectables_kernel)
    static constexpr int NSTEP = NKS <= 4 ? 1 : 2;                // K = 32 steps of linear1 on the bf16 pipe: lane group g carries {a[4 g .. + 3], x[ks < 4]} | {x[4 <= ks < 12]}
    static constexpr int NW = NKS <= 4 ? 4 : 8;                   // waves of a dec_point workgroup (LDS: two workgroups per CU / one)
};
__host__ __device__ constexpr int rowless_nks(int feat_dim) { return feat_dim + 1 <= 12 ? 3 : (feat_dim + 1 <= 44 ? 11 : 0); }
template <int MODE>
__global__ __launch_bounds__(64 * ERM_WAVES, 1) void enc_reduce_mfma_kernel(const float* __restrict__ enc_kv, afm_ln kvn, const float* __restrict__ u_text,
                                                                           const float* __restrict__ cu_text, const float* __restrict__ u_time,
                                                                           const float* __restrict__ cu_time, const int64_t* __restrict__ t, int n_t,
                                                                           int N, float* __restrict__ pm, float* __restrict__ pl,
                                                                           float* __restrict__ pacc, const float* __restrict__ xt,
                                                                           const float* __restrict__ xu, int cd) {
    constexpr bool FOLD = MODE == 1;
    constexpr int NQ = 16, LDY = ERM_LDY, NT = 64 * ERM_WAVES;
    extern __shared__ __attribute__((aligned(16))) float er_sm[];
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, p16 = lane & 15, g = lane >> 4;
    float* ytile = er_sm + wave * 16 * LDY;                       // this wave's transposition tile
    float* ups = er_sm + ERM_WAVES * 16 * LDY;                    // [16][LDY] u'_q = gamma * u_q
    float* xus = ups + 16 * LDY;                                  // FOLD: [8][256] contact columns of the adapter
    float* ccs = xus + 12 * 256;                                  // [16] beta . u_q + c_q
    float* tr = ccs + 16 + wave * 16;                             // 16 floats per wave: a 16-vector from lanes (q, .) to lanes (., g)
    int64_t ti = t[b];
    ti = ti < 0 ? 0 : (ti >= n_t ? n_t - 1 : ti);
    // folded query q < 8: text latent of the sample, else the time latent of t[b]
    auto uptr = [&](int q) { return q < NQ / 2 ? u_text + ((int64_t)b * (NQ / 2) + q) * 256 : u_time + (ti * (NQ / 2) + (q - NQ / 2)) * 256; };
    for (int i = threadIdx.x; i < NQ * 64; i += NT) {              // (query, float4) items
        const int q = i >> 6, c = (i & 63) * 4;
        const float4 u = *reinterpret_cast<const float4*>(uptr(q) + c), gm = *reinterpret_cast<const float4*>(kvn.g + c);
        *reinterpret_cast<float4*>(ups + q * LDY + c) = make_float4(u.x * gm.x, u.y * gm.y, u.z * gm.z, u.w * gm.w);
    }
    for (int q = wave; q < NQ; q += ERM_WAVES) {                  // one wave per dot product beta . u_q
        const float4 u = *reinterpret_cast<const float4*>(uptr(q) + lane * 4), bt = *reinterpret_cast<const float4*>(kvn.b + lane * 4);
        const float d = wave_sum((u.x * bt.x + u.y * bt.y) + (u.z * bt.z + u.w * bt.w));
        if (lane == 0) ccs[q] = d + (q < NQ / 2 ? cu_text[(int64_t)b * (NQ / 2) + q] : cu_time[ti * (NQ / 2) + (q - NQ / 2)]);
    }
    if (FOLD) {
        for (int i = threadIdx.x; i < 8 * 256; i += NT) xus[i] = i < cd * 256 ? xu[i] : 0.f;
    }
    __syncthreads();
    const float cconst = ccs[p16];

    const int per = (N + ERM_SPLIT - 1) / ERM_SPLIT;
    const int n0 = blockIdx.x * per, n1 = min(N, n0 + per);
    const int wper = ((per + ERM_WAVES - 1) / ERM_WAVES + 15) & ~15;      // points per wave, whole tiles
    const int w0 = n0 + wave * wper, w1 = min(n1, w0 + wper);

    f32x4 acc[16];                                                // Acc[q = 4 g + r][c = 16 t + p16]
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;                          // of query p16, replicated over g

    float4 e[16];
    float xrow[8];
    auto fetch = [&](int nb) {                                     // rows of tile [nb, nb + 16): this lane's 64 channels of point nb + p16
        const int64_t pt = (int64_t)b * N + min(nb + p16, n1 - 1);
#pragma unroll
        for (int j = 0; j < 16; ++j) e[j] = *reinterpret_cast<const float4*>(enc_kv + pt * 256 + 16 * j + 4 * g);
        if (FOLD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xrow[j] = xt[pt * cd + min(j, cd - 1)];
        }
    };
    if (w0 < w1) fetch(w0);
    for (int nb = w0; nb < w1; nb += 16) {
        if (FOLD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j < cd) {                                      // wave-uniform
                    const float xj = xrow[j];
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        const float4 w = *reinterpret_cast<const float4*>(xus + j * 256 + 16 * jj + 4 * g);
                        e[jj].x += xj * w.x; e[jj].y += xj * w.y; e[jj].z += xj * w.z; e[jj].w += xj * w.w;
                    }
                }
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += (e[j].x + e[j].y) + (e[j].z + e[j].w);
        sum += xor16(sum); sum += xor32(sum);
        const float mean = sum * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            e[j].x -= mean; e[j].y -= mean; e[j].z -= mean; e[j].w -= mean;
            sq += (e[j].x * e[j].x + e[j].y * e[j].y) + (e[j].z * e[j].z + e[j].w * e[j].w);
        }
        sq += xor16(sq); sq += xor32(sq);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 256.0f) + 1e-5f);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sc1 = sc, sc2 = sc, sc3 = sc;      // four independent accumulation chains
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            e[j].x *= rstd; e[j].y *= rstd; e[j].z *= rstd; e[j].w *= rstd;
            *reinterpret_cast<float4*>(ytile + p16 * LDY + 16 * j + 4 * g) = e[j];
            const float4 u4 = *reinterpret_cast<const float4*>(ups + p16 * LDY + 16 * j + 4 * g);      // lane (q = p16, g): u'_q of the same channels
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32(e[j].x, u4.x, sc, 0, 0, 0);
            sc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e[j].y, u4.y, sc1, 0, 0, 0);
            sc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(e[j].z, u4.z, sc2, 0, 0, 0);
            sc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(e[j].w, u4.w, sc3, 0, 0, 0);
        }
        sc = (sc + sc1) + (sc2 + sc3);
        const int nvalid = w1 - nb;                                // points 4 g + r >= nvalid do not exist
        if (nb + 16 < w1) fetch(nb + 16);                          // e[] is free: the next tile's rows fly under the second product
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sc[r] = (4 * g + r < nvalid) ? sc[r] + cconst : -INFINITY;
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
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {     // some query's maximum moved: rescale its sums (rows q = 4 g + r)
            if (g == 0) tr[p16] = alpha;
            const float a0 = tr[4 * g], a1 = tr[4 * g + 1], a2 = tr[4 * g + 2], a3 = tr[4 * g + 3];
#pragma unroll
            for (int tt = 0; tt < 16; ++tt) { acc[tt][0] *= a0; acc[tt][1] *= a1; acc[tt][2] *= a2; acc[tt][3] *= a3; }
        }
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sc[r], ytile[(4 * g + r) * LDY + 16 * tt + p16], acc[tt], 0, 0, 0);
            if ((tt & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keep the LDS operand reads of four column tiles at a time in flight, not of all 16
        }
    }
    // partial of this wave: true sums = gamma_c * Acc + beta_c * l_q
    const int part = blockIdx.x * ERM_WAVES + wave;
    const int64_t base = ((int64_t)b * NPART + part) * NQ;
    if (g == 0) { pm[base + p16] = m_run; pl[base + p16] = l_run; tr[p16] = l_run; }
    const float l0 = tr[4 * g], l1 = tr[4 * g + 1], l2 = tr[4 * g + 2], l3 = tr[4 * g + 3];
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
        const int c = 16 * tt + p16;
        const float gm = kvn.g[c], bt = kvn.b[c];
        pacc[(base + 4 * g + 0) * 256 + c] = gm * acc[tt][0] + bt * l0;
        pacc[(base + 4 * g + 1) * 256 + c] = gm * acc[tt][1] + bt * l1;
        pacc[(base + 4 * g + 2) * 256 + c] = gm * acc[tt][2] + bt * l2;
        pacc[(base + 4 * g + 3) * 256 + c] = gm * acc[tt][3] + bt * l3;
    }
}


// ---------------------------------------------------------------- enc_reduce without rows (round 3; GEN inputs)
// The rows the two latents attend over are LayerNorm_kv of a linear map of the point's K inputs x = [x_t | features | 1 | 0..] (K = 12: the
// H3D variant, K = 44: 32 scene features per point, the HUMANISE variant), so, exactly as in the decoder (dec_point_kernel): var = x Qe x^T
// (step-invariant K x K form), score[q] = rstd (x . EU[:, q]) + const with EU = Ec U'^T (Ec = gen_enc minus its row means; K x 16 per
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

// ---------------------------------------------------------------- latent chain, batched over the samples
// The 2-latent chain (cross-attention output, o_proj, MLP, self-attention blocks, decoder K / V folding) is ~16 dependent
// matrix-vector stages per sample.  It runs batched over all 2 B latent tokens as a sequence of small launches: a stage is Y[tok, o] = epi(b[o] + W[o, :] . pro(X[tok, :])) for all tokens, N / 8
// workgroups per stage (every weight row is read once per token block, by one workgroup), ~5 us per launch.
//   toklin_kernel: ONE WAVE per 16 tokens x 16 outputs, everything in registers, no LDS, no barrier (round 3; the phase timelines of the
//   earlier forms - VALU with LDS-staged rows and weights: 15 us per work item; matrix pipe with LDS-staged rows: staging 4-8 us,
//   LayerNorm 4 us, product 3-8 us - are in profiles/r03_cdm_chain.md).  The 16 x 16 output tile is K / 4 v_mfma_f32_16x16x4_f32; both
//   operands want "row (l & 15), four consecutive k at 16 u + 4 (l >> 4)" per lane, which is how a lane reads its 16-byte pieces of an
//   input row and of a weight row straight from global memory: K / 16 float4 each, all issued at kernel entry.  A token's row is then
//   spread over the four lanes (l & 15) + 16 g, so the LayerNorm statistics are a per-lane sum plus two cross-lane steps and the
//   normalisation happens in registers.
constexpr int TL_TOK = 16, TL_OB = 16;
struct TokLin {
    const float* X; int ldx;                 // input rows: token tok at X + tok * ldx (+ head offset)
    int head_out, x_head_stride;             // head_out > 0: outputs [h * head_out, (h + 1) * head_out) read X + h * x_head_stride (per-head inputs)
    const float* W[3]; const float* b[3];    // up to three stacked weight matrices [ncol, K] (q | k | v), ncol outputs each
    int ncol;
    afm_ln ln; int use_ln;                   // 1: LayerNorm (eps 1e-5) of the input rows; 2: the same FOLDED - W carries gamma, b carries W beta,
    const float* gsum[3];                    //    gsum[part][o] = sum_k W[o][k]: Y = rstd (W x - mean gsum) + b, the products do not wait for the statistics
    int act;                                 // AFM_ACT_*
    const float* R; int ldr;                 // residual rows or NULL (may be Y: every element is read and written by the same lane)
    float* Y; int ldy;
    int ntok, N, K;                          // K % 16 == 0, K <= MAXD
};

#ifdef AFM_TOKLIN_TIMELINE      // tools/probes/toklin_timeline.py only (a debug build of this file); never compiled into the library
__device__ unsigned long long afm_tk_tl[16 * 8];      // [launch slot][stamp]
__device__ int afm_tk_slot = 0;
#define TKTL(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { afm_tk_tl[(afm_tk_slot & 15) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 4) afm_tk_slot = afm_tk_slot + 1; } } while (0)
#else
#define TKTL(i)
#endif

// NK16 = K / 16 at compile time (a run-time bound on the unrolled register arrays turns every step into compute-and-select)
template <int NK16>
__global__ __launch_bounds__(64) void toklin_kernel(const TokLin p) {
    TKTL(0);
    const int lane = threadIdx.x, p16 = lane & 15, g = lane >> 4;
    const int tb = blockIdx.y * TL_TOK, o0 = blockIdx.x * TL_OB;
    const int part0 = o0 / p.ncol, oc0 = o0 - part0 * p.ncol;    // the TL_OB outputs of a wave lie in one weight part (ncol % TL_OB == 0)
    const bool ovalid = o0 + p16 < p.N;
    // A operand: token tb + p16 (clamped: rows past the end are computed and dropped); B operand: weight row o0 + p16
    const float* xrow = p.X + (p.head_out ? (o0 / p.head_out) * p.x_head_stride : 0) + (int64_t)min(tb + p16, p.ntok - 1) * p.ldx + 4 * g;
    const float* wrow = p.W[part0] + (int64_t)(ovalid ? oc0 + p16 : 0) * p.K + 4 * g;
    float4 xr[NK16], wr[NK16];
#pragma unroll
    for (int u = 0; u < NK16; ++u) {
        xr[u] = *reinterpret_cast<const float4*>(xrow + 16 * u);
        wr[u] = *reinterpret_cast<const float4*>(wrow + 16 * u);
    }
    TKTL(1);
    float ln_mean = 0.f, ln_rstd = 1.f;                          // folded form: statistics of token p16 (all four lanes of the row hold them)
    if (p.use_ln) {                                              // uniform.  Statistics over the row's four lanes (same p16, g = 0..3), two passes
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < NK16; ++u) sum += (xr[u].x + xr[u].y) + (xr[u].z + xr[u].w);
        sum += xor16(sum); sum += xor32(sum);
        const float mean = sum / (float)p.K;
        float sq = 0.f;
#pragma unroll
        for (int u = 0; u < NK16; ++u) { const float a = xr[u].x - mean, b = xr[u].y - mean, c = xr[u].z - mean, d = xr[u].w - mean; sq += (a * a + b * b) + (c * c + d * d); }
        sq += xor16(sq); sq += xor32(sq);
        const float rstd = 1.0f / sqrtf(sq / (float)p.K + 1e-5f);
        ln_mean = mean; ln_rstd = rstd;
#pragma unroll
        for (int u = 0; u < NK16; ++u) {
            if (p.use_ln == 2) break;                            // uniform: the rows stay raw
            const float4 gg = *reinterpret_cast<const float4*>(p.ln.g + 16 * u + 4 * g), bb = *reinterpret_cast<const float4*>(p.ln.b + 16 * u + 4 * g);
            xr[u] = make_float4((xr[u].x - mean) * rstd * gg.x + bb.x, (xr[u].y - mean) * rstd * gg.y + bb.y,
                                (xr[u].z - mean) * rstd * gg.z + bb.z, (xr[u].w - mean) * rstd * gg.w + bb.w);
        }
    }
    TKTL(2);
    // ---- D[i = token][j = output] += X[i][k] W[j][k]: MFMA e of step u takes k = 16 u + 4 (l >> 4) + e on both operands; four accumulators
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NK16; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[u].x, wr[u].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[u].y, wr[u].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[u].z, wr[u].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[u].w, wr[u].w, acc[3], 0, 0, 0);
    }
    TKTL(3);
    float mt[4] = {0.f, 0.f, 0.f, 0.f}, rt[4] = {1.f, 1.f, 1.f, 1.f};
    if (p.use_ln == 2) {                                         // uniform: the statistics of token 4 g + r live in lane 4 g + r
#pragma unroll
        for (int r = 0; r < 4; ++r) { mt[r] = __shfl(ln_mean, 4 * g + r); rt[r] = __shfl(ln_rstd, 4 * g + r); }
    }
    if (ovalid) {                                                // lane (output p16; tokens 4 g + r of the tile)
        const float bias = p.b[part0] ? p.b[part0][oc0 + p16] : 0.f;
        const float gs = p.use_ln == 2 ? p.gsum[part0][oc0 + p16] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tok = tb + 4 * g + r;
            float v = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);
            if (p.use_ln == 2) v = rt[r] * (v - mt[r] * gs);
            if (tok >= p.ntok) continue;
            v += bias;
            if (p.act) v = apply_act(v, p.act);
            if (p.R) v += p.R[(int64_t)tok * p.ldr + o0 + p16];
            p.Y[(int64_t)tok * p.ldy + o0 + p16] = v;
        }
    }
    TKTL(4);
}

// combine the per-wave partials of enc_reduce into s [ntok][He][dkv] (token = 2 b + i, i = 0 text latent, 1 time latent) and set the
// latent state x0 [ntok][dq].  grid (B, 2 He), block dkv = 256.
__global__ __launch_bounds__(256) void lat_combine_kernel(const float* __restrict__ pm, const float* __restrict__ pl,
                                                          const float* __restrict__ pacc, int nih, int dkv, const float* __restrict__ q0_text,
                                                          const float* __restrict__ q0_time, const int64_t* __restrict__ t, int n_t, int dq,
                                                          float* __restrict__ sbuf, float* __restrict__ x0) {
    __shared__ float wq[NPART];
    const int b = blockIdx.x, ih = blockIdx.y;
    if (threadIdx.x < 64) {                                       // wave 0: NPART = 64 partial (max, sum) pairs
        const int pi = threadIdx.x;
        const float mm = pi < NPART ? pm[((int64_t)b * NPART + pi) * nih + ih] : -INFINITY;
        float M = mm;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
        const float ww = (mm == -INFINITY) ? 0.f : __expf(mm - M);
        const float L = wave_sum(pi < NPART ? pl[((int64_t)b * NPART + pi) * nih + ih] * ww : 0.f);
        if (pi < NPART) wq[pi] = ww * (1.0f / L);
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < dkv) {
        float a = 0.f;
        for (int pi = 0; pi < NPART; ++pi) a += wq[pi] * pacc[(((int64_t)b * NPART + pi) * nih + ih) * dkv + c];
        const int He = nih / 2, i = ih / He, h = ih % He;
        sbuf[(((int64_t)b * 2 + i) * He + h) * dkv + c] = a;
    }
    if (ih < 2) {                                                 // latent token i = ih of this sample
        int64_t ti = t[b];
        ti = ti < 0 ? 0 : (ti >= n_t ? n_t - 1 : ti);
        const float* src = ih == 0 ? q0_text + (int64_t)b * dq : q0_time + ti * dq;
        for (int k = threadIdx.x; k < dq; k += blockDim.x) x0[((int64_t)b * 2 + ih) * dq + k] = src[k];
    }
}

// The head of the chain in one launch (fused sampling form): merge enc_point_kernel's partials of the 8 heads of a latent token into the
// 8 x 12 numbers a12[h][k] = sum_n a[n] rstd[n] x[n][k], then apply everything that is linear behind them at once -
//   x1 = q0 + o_proj(v_proj(gamma_kv * (a12 Ec) + beta_kv)) = q0 + c1 + sum_{h, k} a12[h][k] WOVE[12 h + k]
// with WOVE [96][dq] = W_o (per-head blocks) W_v (gamma_kv * Ec)^T and c1 = b_o + W_o (W_v beta_kv + b_v) from the host (float64).
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

// self-attention of the two latent tokens of a sample (modules.py:544-648): qkv [ntok][3 dq] -> out [ntok][dq].  grid B, block 256.
__global__ __launch_bounds__(256) void lat_selfattn_kernel(const float* __restrict__ qkv, int dq, int He, float* __restrict__ out) {
    __shared__ float sc[64], aw[64];
    const int b = blockIdx.x, hd = dq / He;
    const float* q = qkv + (int64_t)b * 2 * 3 * dq;               // token rows 2 b, 2 b + 1: [q | k | v]
    const float scale = 1.0f / sqrtf((float)hd);
    if (threadIdx.x < He * 4) {                                   // (h, i, j) scores
        const int h = threadIdx.x >> 2, i = (threadIdx.x >> 1) & 1, j = threadIdx.x & 1;
        float a = 0.f;
        for (int r = 0; r < hd; ++r) a += (q[i * 3 * dq + h * hd + r] * scale) * q[j * 3 * dq + dq + h * hd + r];
        sc[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x < He * 2) {                                   // softmax over the 2 keys
        const int base = threadIdx.x * 2;
        const float a0 = sc[base], a1 = sc[base + 1], mx = fmaxf(a0, a1);
        const float e0 = __expf(a0 - mx), e1 = __expf(a1 - mx), inv = 1.0f / (e0 + e1);
        aw[base] = e0 * inv; aw[base + 1] = e1 * inv;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * dq; e += blockDim.x) {
        const int i = e / dq, c = e % dq, h = c / hd;
        out[((int64_t)b * 2 + i) * dq + c] = aw[(h * 2 + i) * 2 + 0] * q[2 * dq + c] + aw[(h * 2 + i) * 2 + 1] * q[3 * dq + 2 * dq + c];
    }
}

// decoder keys / values of the two latents folded through W_q / W_o of the decoder attention (+ contact_layer for the folded form):
// kv [ntok][2 dkv] (k | v) -> dec_lat record of the sample.  grid (B, Hd), block dkv = 256: workgroup (b, h) owns jh = h and Hd + h.
__global__ __launch_bounds__(256) void lat_decfold_kernel(const afm_cdm_weights w, const float* __restrict__ kv, float* __restrict__ dec_lat) {
    __shared__ float red[4][16];
    const int b = blockIdx.x, h = blockIdx.y, c = threadIdx.x, dkv = w.dkv, Hd = w.dec_heads, hdd = dkv / Hd, njh = 2 * Hd;
    const float scd = 1.0f / sqrtf((float)hdd);
    float* G = dec_lat + (int64_t)b * DEC_LAT_STRIDE(njh);
    float* P = G + njh * dkv;
    float* cb = P + njh * dkv;
    float* WP = cb + njh;
    // Both latents of the sample share the weight reads: W_q rows h*hdd .. (coalesced over c) and the 32-float run of W_o's row c
    // (eight 16-byte loads per thread, issued together, instead of 2 x 32 dependent 4-byte loads at a 1 KB stride).
    // The 4 x hdd key / value entries of this (sample, head) are staged in LDS and read back into VGPRs, NOT read with scalar loads.
    // The scalar-load form of this kernel (operands of the packed-f32 products in SGPR pairs) was non-deterministic on MI355X whenever
    // the other sub-batch stream had kernels in flight: in ~1/4 of 50-step loops single waves lost ONE product of the P sum in lanes
    // 48..63 (always one with an odd-indexed SGPR operand).  Stale caches, instruction alignment and SGPR write-after-read were ruled
    // out by experiment and two instruction-level replays did not reproduce it, so the mechanism is open; this form ran 80/80 loops
    // bit-identical in the same harness (profiles/r02_decfold_nondeterminism.md, tests/test_gpu_cdm.py::test_two_sub_batch_loop_repeats).
    __shared__ __attribute__((aligned(16))) float kvs[4][32];     // [k of latent 0 | k of latent 1 | v of latent 0 | v of latent 1][hdd <= 32]
    if (c < 4 * hdd) {
        const int a = c / hdd, r = c - a * hdd;
        kvs[a][r] = kv[((int64_t)b * 2 + (a & 1)) * 2 * dkv + (a >> 1) * dkv + h * hdd + r];
    }
    __syncthreads();
    float pj[2];
    {
        const float *kd0 = kvs[0], *kd1 = kvs[1], *vd0 = kvs[2], *vd1 = kvs[3];
        float4 ow[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ow[u] = (4 * u < hdd) ? *reinterpret_cast<const float4*>(w.dec_attn.o.w + (int64_t)c * dkv + h * hdd + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
        float a0 = 0.f, a1 = 0.f, p0 = 0.f, p1 = 0.f;
#pragma unroll 8
        for (int r = 0; r < hdd; ++r) {
            const float wq = w.dec_attn.q.w[(int64_t)(h * hdd + r) * dkv + c];
            a0 += wq * kd0[r];
            a1 += wq * kd1[r];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (4 * u < hdd) {
                p0 += (ow[u].x * vd0[4 * u] + ow[u].y * vd0[4 * u + 1]) + (ow[u].z * vd0[4 * u + 2] + ow[u].w * vd0[4 * u + 3]);
                p1 += (ow[u].x * vd1[4 * u] + ow[u].y * vd1[4 * u + 1]) + (ow[u].z * vd1[4 * u + 2] + ow[u].w * vd1[4 * u + 3]);
            }
        }
        G[h * dkv + c] = a0 * scd; G[(Hd + h) * dkv + c] = a1 * scd;
        P[h * dkv + c] = p0; P[(Hd + h) * dkv + c] = p1;
        pj[0] = p0; pj[1] = p1;
        if (c < 2) {
            const float* kd = c == 0 ? kd0 : kd1;
            float cbv = 0.f;
            for (int r = 0; r < hdd; ++r) cbv += w.dec_attn.q.b[h * hdd + r] * kd[r];
            cb[c * Hd + h] = cbv * scd;
        }
    }
    if (w.fold_xu && w.contact_dim <= 8) {                        // WP[r, jh] = contact_layer.w[r] . P[jh]
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int r = 0; r < w.contact_dim; ++r) {
            const float wc = w.contact_layer.w[(int64_t)r * dkv + c];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float s = wave_sum(wc * pj[j]);
                if (lane == 0) red[wave][r * 2 + j] = s;
            }
        }
        __syncthreads();
        if (threadIdx.x < w.contact_dim * 2) {
            const int r = threadIdx.x >> 1, j = threadIdx.x & 1;
            WP[r * njh + j * Hd + h] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        }
    }
}

// ---------------------------------------------------------------- dec_attend on the matrix pipe
// Same idea as enc_reduce_mfma_kernel, per tile of 16 points and wave, everything row-shaped stays in layout A (lane (p, g): channels
// {16 j + 4 g + e} of point p):
//   scores    S[p, jh] = sum_c yhat[p, c] G'[jh, c] with G' = gamma_q * G (LayerNorm's affine part folded into the keys and the score
//             constants): 64 MFMAs, result in lanes (jh, g) for points 4 g + r; the two keys of a head are 8 lanes apart (one DPP step);
//   output    O^T[c, p] = sum_jh P[jh, c] a[p, jh]: 64 MFMAs with the CHANNEL as the output row, so that the result lands in layout A
//             (lane (p, g), channel 16 t + 4 g + r) next to the query row it is added to; the only transposition is the 16 x 16
//             matrix of attention weights (1 KB per wave through LDS);
//   folded    contact_layer.w . (attention output) = sum_jh a[p, jh] WP[r, jh]: 4 more MFMAs (the 8 output rows are the contact channels);
//   residual + second LayerNorm + affine on the VALU in layout A, rows written with 16-byte stores.
// G', P, the three channel vectors (b_o, gamma_mlp, beta_mlp) and the contact columns live in LDS once per workgroup (one sample).
constexpr int DAM_LDG = 260;
constexpr int DAM_LDS_FLOATS = 2 * 16 * DAM_LDG + 3 * 256 + 16 + 8 * 16 + 12 * 256 + 4 * 16 * 17;

// MODE 0 / 1 as in enc_reduce_mfma_kernel: query rows from memory / + the contact columns (FOLD).  Inputs of at most 11 channels: dec_point_kernel.
template <int MODE>
__global__ __launch_bounds__(256, 2) void dec_attend_mfma_kernel(const float* __restrict__ dec_q0, const float* __restrict__ dec_lat, afm_ln qn,
                                                                const float* __restrict__ bo, afm_ln mlpn, int N, float* __restrict__ h1,
                                                                float* __restrict__ z, const float* __restrict__ xt,
                                                                const float* __restrict__ xv, int cd, float* __restrict__ s1) {
    constexpr bool FOLD = MODE == 1;                              // h1 is not stored, s1 is
    constexpr int NJH = 16, LDG = DAM_LDG;
    extern __shared__ __attribute__((aligned(16))) float da_sm[];
    float* Gs = da_sm;                                            // [16][LDG]  gamma_q * G
    float* Ps = Gs + 16 * LDG;                                    // [16][LDG]  P
    float* vec3 = Ps + 16 * LDG;                                  // [3][256]   b_o, gamma_mlp, beta_mlp
    float* gcs = vec3 + 3 * 256;                                  // [16]       beta_q . G[jh] + cb[jh]
    float* WPs = gcs + 16;                                        // [8][16]    contact_layer.w . P  (rows >= cd: 0)
    float* xvs = WPs + 8 * 16;                                    // FOLD: [8][256] contact columns of the decoder query
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, p16 = lane & 15, g = lane >> 4;
    float* aT = xvs + 12 * 256 + wave * 16 * 17;                  // [16 points][17] attention weights of the tile, transposed
    const float* rec = dec_lat + (int64_t)b * DEC_LAT_STRIDE(NJH);
    for (int i = threadIdx.x; i < NJH * 64; i += 256) {            // (jh, float4) items
        const int jh = i >> 6, c = (i & 63) * 4;
        const float4 gv = *reinterpret_cast<const float4*>(rec + jh * 256 + c), pv = *reinterpret_cast<const float4*>(rec + NJH * 256 + jh * 256 + c),
                     gm = *reinterpret_cast<const float4*>(qn.g + c);
        *reinterpret_cast<float4*>(Gs + jh * LDG + c) = make_float4(gv.x * gm.x, gv.y * gm.y, gv.z * gm.z, gv.w * gm.w);
        *reinterpret_cast<float4*>(Ps + jh * LDG + c) = pv;
    }
    for (int jh = wave; jh < NJH; jh += 4) {
        const float4 gv = *reinterpret_cast<const float4*>(rec + jh * 256 + lane * 4), bt = *reinterpret_cast<const float4*>(qn.b + lane * 4);
        const float d = wave_sum((gv.x * bt.x + gv.y * bt.y) + (gv.z * bt.z + gv.w * bt.w));
        if (lane == 0) gcs[jh] = d + rec[2 * NJH * 256 + jh];
    }
    for (int i = threadIdx.x; i < 256; i += 256) { vec3[i] = bo[i]; vec3[256 + i] = mlpn.g[i]; vec3[512 + i] = mlpn.b[i]; }
    if (threadIdx.x < 8 * 16) WPs[threadIdx.x] = (FOLD && (int)(threadIdx.x >> 4) < cd) ? rec[2 * NJH * 256 + NJH + threadIdx.x] : 0.f;
    if (FOLD) {
        for (int i = threadIdx.x; i < 8 * 256; i += 256) xvs[i] = i < cd * 256 ? xv[i] : 0.f;
    }
    __syncthreads();
    const float gconst = gcs[p16];

    const int per = (N + gridDim.x - 1) / gridDim.x;
    const int n0 = blockIdx.x * per, n1 = min(N, n0 + per);
    const int wper = ((per + 3) / 4 + 15) & ~15;                  // points per wave, whole tiles
    const int w0 = n0 + wave * wper, w1 = min(n1, w0 + wper);

    float4 e[16];
    float xrow[8];
    auto fetch = [&](int nb) {                                     // 32-bit element offsets from the uniform bases (one address register per load)
        const unsigned pti = (unsigned)(b * N + min(nb + p16, n1 - 1));
        const unsigned ro = pti * 256u + 4u * (unsigned)g;
#pragma unroll
        for (int j = 0; j < 16; ++j) e[j] = *reinterpret_cast<const float4*>(dec_q0 + (ro + 16u * j));
        if (FOLD) {
            const unsigned xo = pti * (unsigned)cd;
#pragma unroll
            for (int j = 0; j < 8; ++j) xrow[j] = xt[xo + (unsigned)min(j, cd - 1)];
        }
    };
    if (w0 < w1) fetch(w0);
    for (int nb = w0; nb < w1; nb += 16) {
        const int64_t pt = (int64_t)b * N + nb + p16;
        const bool pvalid = nb + p16 < w1;
        if (FOLD) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j >= cd) break;                                // wave-uniform
                const float xj = xrow[j];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const float4 w = *reinterpret_cast<const float4*>(xvs + j * 256 + 16 * jj + 4 * g);
                    e[jj].x += xj * w.x; e[jj].y += xj * w.y; e[jj].z += xj * w.z; e[jj].w += xj * w.w;
                }
                __builtin_amdgcn_sched_barrier(0);               // one contact column's LDS reads in flight at a time (registers)
            }
        }
        __builtin_amdgcn_sched_barrier(0);                         // phase boundary: keeps the scheduler from overlapping phases (registers)
        // LayerNorm statistics of the query row; scores against the folded keys (affine part inside G' / gconst)
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum += (e[j].x + e[j].y) + (e[j].z + e[j].w);
        sum += xor16(sum); sum += xor32(sum);
        const float mean = sum * (1.0f / 256.0f);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float a0 = e[j].x - mean, a1 = e[j].y - mean, a2 = e[j].z - mean, a3 = e[j].w - mean;
            sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        sq += xor16(sq); sq += xor32(sq);
        const float rstd = 1.0f / sqrtf(sq * (1.0f / 256.0f) + 1e-5f);
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sc1 = sc, sc2 = sc, sc3 = sc;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 g4 = *reinterpret_cast<const float4*>(Gs + p16 * LDG + 16 * j + 4 * g);        // lane (jh = p16, g)
            sc = __builtin_amdgcn_mfma_f32_16x16x4f32((e[j].x - mean) * rstd, g4.x, sc, 0, 0, 0);
            sc1 = __builtin_amdgcn_mfma_f32_16x16x4f32((e[j].y - mean) * rstd, g4.y, sc1, 0, 0, 0);
            sc2 = __builtin_amdgcn_mfma_f32_16x16x4f32((e[j].z - mean) * rstd, g4.z, sc2, 0, 0, 0);
            sc3 = __builtin_amdgcn_mfma_f32_16x16x4f32((e[j].w - mean) * rstd, g4.w, sc3, 0, 0, 0);
        }
        sc = (sc + sc1) + (sc2 + sc3);
        __builtin_amdgcn_sched_barrier(0);                         // phase boundary: keeps the scheduler from overlapping phases (registers)
        // softmax over the two keys of a head (jh and jh ^ 8: eight lanes apart), then the weights transposed to (point, jh)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s_own = sc[r] + gconst, s_oth = lane_xor<8>(s_own);
            const float mx = fmaxf(s_own, s_oth);
            const float e_own = __expf(s_own - mx), e_oth = __expf(s_oth - mx);
            aT[(4 * g + r) * 17 + p16] = e_own / (e_own + e_oth);
        }
        float aB[4];
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) aB[sI] = aT[p16 * 17 + 4 * g + sI];       // lane (p = p16, g): a[p, jh = 4 g + s]
        f32x4 acc[16];
#pragma unroll
        for (int tt = 0; tt < 16; ++tt) {
            acc[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sI = 0; sI < 4; ++sI)
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Ps[(4 * g + sI) * LDG + 16 * tt + p16], aB[sI], acc[tt], 0, 0, 0);
            if ((tt & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (FOLD) {                                                // contact_layer.w . (attention output): rows = contact channels
            f32x4 sa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int sI = 0; sI < 4; ++sI)
                sa = __builtin_amdgcn_mfma_f32_16x16x4f32(p16 < 8 ? WPs[p16 * 16 + 4 * g + sI] : 0.f, aB[sI], sa, 0, 0, 0);
            if (pvalid) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < cd) s1[pt * cd + 4 * g + r] = sa[r];
            }
        }
        __builtin_amdgcn_sched_barrier(0);                         // phase boundary: keeps the scheduler from overlapping phases (registers)
        // residual (adds the raw query row) in layout A, then e[] is free for the next tile's rows
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 ob = *reinterpret_cast<const float4*>(vec3 + 16 * j + 4 * g);
            acc[j][0] = (acc[j][0] + ob.x) + e[j].x; acc[j][1] = (acc[j][1] + ob.y) + e[j].y;
            acc[j][2] = (acc[j][2] + ob.z) + e[j].z; acc[j][3] = (acc[j][3] + ob.w) + e[j].w;
            // (the residual stream is stored in the layered form only: h1 == NULL in the folded one.  The test stays a run-time one on
            // purpose: with the store compiled out hipcc schedules this phase into 40 more live registers and spills)
            if (pvalid && h1) *reinterpret_cast<float4*>(h1 + pt * 256 + 16 * j + 4 * g) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
        }
        __builtin_amdgcn_sched_barrier(0);                         // phase boundary: keeps the scheduler from overlapping phases (registers)
        if (nb + 16 < w1) fetch(nb + 16);
        __builtin_amdgcn_sched_barrier(0);                         // phase boundary: keeps the scheduler from overlapping phases (registers)
        float sum2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) sum2 += (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
        sum2 += xor16(sum2); sum2 += xor32(sum2);
        const float mean2 = sum2 * (1.0f / 256.0f);
        float sq2 = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j][0] -= mean2; acc[j][1] -= mean2; acc[j][2] -= mean2; acc[j][3] -= mean2;
            sq2 += (acc[j][0] * acc[j][0] + acc[j][1] * acc[j][1]) + (acc[j][2] * acc[j][2] + acc[j][3] * acc[j][3]);
        }
        sq2 += xor16(sq2); sq2 += xor32(sq2);
        const float rstd2 = 1.0f / sqrtf(sq2 * (1.0f / 256.0f) + 1e-5f);
        if (pvalid) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 g2 = *reinterpret_cast<const float4*>(vec3 + 256 + 16 * j + 4 * g), b2 = *reinterpret_cast<const float4*>(vec3 + 512 + 16 * j + 4 * g);
                *reinterpret_cast<float4*>(z + pt * 256 + 16 * j + 4 * g) =
                    make_float4(acc[j][0] * rstd2 * g2.x + b2.x, acc[j][1] * rstd2 * g2.y + b2.y, acc[j][2] * rstd2 * g2.z + b2.z, acc[j][3] * rstd2 * g2.w + b2.w);
            }
        }
    }
}


// ---------------------------------------------------------------- the whole decoder of a point in ONE kernel (round 3; GEN inputs)
// After the attention weights a[p, jh] of a point are known, everything up to linear1 is LINEAR in u[p] = [a[p, 0..15] | x_t | features | 1]:
//   h1[p, c] = sum_jh P[jh, c] a[p, jh] + b_o[c] + e[p, c],  e = the generated query row       = sum_m u[p, m] T[m, c]   (T = [P ; gen_dec (+ b_o)])
//   z = LayerNorm_mlp(h1): h1 - mean_c(h1) = sum_m u[p, m] Tc[m, c]  (Tc = T minus its row means),  var_c(h1) = u Qc u^T,  Qc = Tc Tc^T / 256
//   linear1(z)[n] = rstd[p] * sum_m u[p, m] TWc[m, n] + C[n],   TWc = Tc (W1 * gamma_mlp)^T,  C = b1 + W1 beta_mlp
// so the 256-wide rows h1 and z never exist and linear1 is a K = 28 product instead of K = 256 (x 9 bf16 products): per 16 points
// 112 + 14 f32 MFMAs against 64 (P V) + 576 (linear1 on the bf16 pipe).  The hidden row GELU(linear1) lives one 16-channel tile at a time
// and goes straight into the row-dots with w2 = contact_layer.w fc2.w (64 MFMAs, the contact channel as the output row), to which the
// attention part (WP a) and the query part (gen_qe . inputs) of contact_layer.w . h1 are added in the same accumulator: the kernel reads
// 9 floats per point and writes the 6 of x_0 / x_{t-1}.  Per sample and step the rows of P enter through two small launches in front:
// lat_dectab_kernel (centred rows Pc, the [28 x 28] quadratic form in operand order) and one toklin launch (TWc rows of P = Pc W1g^T);
// the rows of the inputs are step-invariant and come from the host (afm_cdm_weights.dec_*).  Same function as the layer-by-layer form up
// to f32 re-association (tests/test_gpu_cdm.py).
// The attention scores fold the same way: LayerNorm_q(e) . G'[jh] = rstd_q (x . EG[:, jh]) + const with x = the 12 inputs, EG = Dc G'^T
// (Dc = gen_dec minus its row means, per sample and step: lat_dectab_kernel) and var_q = x Qd x^T (step-invariant): the query row e is
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
    using RL = RowLess<NKS>;
    constexpr int K = RL::K, NQ = 16 + K;
    __shared__ float kd[2][256], vd[2][256];
    __shared__ float tv[16][257];
    __shared__ float Q[NQ][NQ + 1];
    const int b = blockIdx.x, c = threadIdx.x & 255, gq4 = threadIdx.x >> 8;
    const float scd = 0.17677669529663687f;                       // 1 / sqrt(32)
    if (gq4 < 2) {
        kd[gq4][c] = lat_kv[((int64_t)b * 2 + gq4) * 512 + c];
        vd[gq4][c] = lat_kv[((int64_t)b * 2 + gq4) * 512 + 256 + c];
    }
    __syncthreads();
    {   // column c of the two [256][256] matrices against the value vectors of heads 2 gq4, 2 gq4 + 1: TWc rows (to memory), tv = vd WoG (to LDS)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * gq4 + hh;
            float t0 = 0.f, t1 = 0.f, g0 = 0.f, g1 = 0.f;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const int o = 32 * h + r;
                const float w1 = wow[o * 256 + c], w2 = wog[o * 256 + c], v0 = vd[0][o], v1 = vd[1][o];
                t0 += v0 * w1; t1 += v1 * w1;
                g0 += v0 * w2; g1 += v1 * w2;
            }
            twp[((int64_t)b * 16 + h) * 256 + c] = t0; twp[((int64_t)b * 16 + 8 + h) * 256 + c] = t1;
            tv[h][c] = g0; tv[8 + h][c] = g1;
        }
    }
    float* T = tab + (int64_t)b * RL::TAB;
    for (int e = threadIdx.x; e < K * 16; e += 1024) {            // EG [K][16] and the attention-weight x input block of Qc
        const int k = e >> 4, jh = e & 15, j = jh >> 3, h = jh & 7;
        float eg = 0.f, q = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) { eg += kd[j][32 * h + r] * dwq[k * 256 + 32 * h + r]; q += vd[j][32 * h + r] * xwo[k * 256 + 32 * h + r]; }
        T[RL::QTAB + e] = eg * scd;
        Q[jh][16 + k] = q; Q[16 + k][jh] = q;
    }
    if (threadIdx.x < 16) {
        const int j = threadIdx.x >> 3, h = threadIdx.x & 7;
        float gq = 0.f;
        for (int r = 0; r < 32; ++r) gq += kd[j][32 * h + r] * wqb[32 * h + r];
        T[RL::QTAB + K * 16 + threadIdx.x] = gq * scd;
    }
    if (threadIdx.x >= 256 && threadIdx.x < 256 + 128) {          // WP [8][16]
        const int e = threadIdx.x - 256, rr = e >> 4, jh = e & 15, j = jh >> 3, h = jh & 7;
        float wp = 0.f;
        if (rr < cd)
            for (int r = 0; r < 32; ++r) wp += vd[j][32 * h + r] * wco[rr * 256 + 32 * h + r];
        T[RL::QTAB + K * 16 + 16 + e] = wp;
    }
    for (int e = threadIdx.x; e < K * K; e += 1024) Q[16 + e / K][16 + e % K] = qxx[e];
    __syncthreads();
    if (threadIdx.x < 256) {   // Qc[jh][jh'] = tv[jh] (head block of jh') . vd_j'
        const int jh = c >> 4, jh2 = c & 15, j2 = jh2 >> 3, h2 = jh2 & 7;
        float q = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) q += tv[jh][32 * h2 + r] * vd[j2][32 * h2 + r];
        Q[jh][jh2] = q;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < RL::QTAB; e += 1024) {
        const int cs = e / RL::LDQ, col = e - cs * RL::LDQ;
        float v = 0.f;
        if (col < 16 + 16 * RL::NT) {
            const int i = col & 15, tt = (col >> 4) - 1, x = cs - 16, xg = x / RL::XS, xk = x - xg * RL::XS;
            const int m2 = col < 16 ? i : (4 * tt + (i & 3) < NKS ? 16 + 4 * (4 * tt + (i & 3)) + (i >> 2) : -1);
            const int m = cs < 16 ? cs : (xk < NKS ? 16 + 4 * xk + xg : -1);
            if (m >= 0 && m2 >= 0) v = Q[m2][m];
        }
        T[e] = v;
    }
}

template <int NKS> constexpr int dp_lds_floats() {
    using RL = RowLess<NKS>;
    return 16 * RL::NSTEP * 3 * 64 * 4 + RL::TAB + 8 * DP_LDW + 256 + RL::K * 16 * RL::NT + 8 * 16 * RL::NT + 16 + RL::NW * 16 * 17 + RL::NW * 16;
}

template <int NKS>
__global__ __launch_bounds__(64 * RowLess<NKS>::NW, NKS <= 4 ? 2 : 1)
void dec_point_kernel(const float* __restrict__ twp, const float* __restrict__ qtab, const float* __restrict__ qdd, const float* __restrict__ twx,
                      const float* __restrict__ cvec, const float* __restrict__ w2f, const float* __restrict__ gen_qe, const float* __restrict__ c0, int N, int cd,
                      const float* xt, const float* __restrict__ feat, int fd, float* __restrict__ x0_out, const float* __restrict__ noise, float* x_next,
                      const float* __restrict__ c1, const float* __restrict__ c2, const float* __restrict__ sigma) {
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
            const float vx = xt[pti * (unsigned)cd + (unsigned)min(k, cd - 1)], vf = feat[pti * (unsigned)fd + (unsigned)min(k, fd - 1)];
            dst[ks] = k < cd ? vx : (k < fd ? vf : (k == fd ? 1.0f : 0.0f));
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
                for (int pq = 0; pq < 9; ++pq)
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
                    const float v = sat[r] + c0s[j];
                    if (x0_out) x0_out[i] = v;
                    if (x_next) x_next[i] = (c1[b] * v + c2[b] * xt[i]) + sigma[b] * noise[i];
                }
            }
        }
    }
}

struct CdmWs {
    float *enc_kv, *bufB, *h1, *z, *pm, *pl, *pacc, *dec_lat, *s1, *rdot, *qe;
    float *lat_s, *lat_x, *lat_t1, *lat_t2, *lat_qkv, *lat_kv;     // batched latent chain: [2B] token rows
    float *twp, *qtab;                                             // fused decoder: TWc rows of the attention weights [16 B][256], per-sample tables [B][RowLess::TAB]
    int64_t bytes;
};

CdmWs carve(const afm_cdm_weights& w, int B, int N, void* base) {
    char* p = (char*)base;
    int64_t off = 0;
    auto take = [&](int64_t n) { char* r = p ? p + off : nullptr; off += align256(n); return (float*)r; };
    const int64_t M = (int64_t)B * N, nih = 2 * w.enc_heads, njh = 2 * w.dec_heads;
    CdmWs s;
    s.enc_kv = take(M * w.dkv * 4); s.bufB = take(M * w.dkv * 4); s.h1 = take(M * w.dkv * 4); s.z = take(M * w.dkv * 4);
    s.pm = take((int64_t)B * NPART * nih * 4); s.pl = take((int64_t)B * NPART * nih * 4);
    s.pacc = take((int64_t)B * NPART * nih * w.dkv * 4);
    s.dec_lat = take((int64_t)B * DEC_LAT_STRIDE(njh) * 4);
    s.s1 = take(M * 8 * 4);                                  // folded path: contact_layer . h1 per point (<= 8 channels)
    s.rdot = take(M * (w.dkv / 64) * 8 * 4);                 // folded path: row-dot partials of the fc1 GEMM
    s.qe = take(M * 8 * 4);                                  // folded path: contact_layer . (step-invariant part of the decoder query)
    const int64_t ntok = 2 * (int64_t)B;
    s.lat_s = take(ntok * w.enc_heads * w.dkv * 4); s.lat_x = take(ntok * w.dq * 4); s.lat_t1 = take(ntok * w.dq * 4);
    s.lat_t2 = take(ntok * w.dq * 4); s.lat_qkv = take(ntok * 3 * w.dq * 4); s.lat_kv = take(ntok * 2 * w.dkv * 4);
    s.twp = take((int64_t)B * 16 * 256 * 4); s.qtab = take((int64_t)B * RowLess<11>::TAB * 4);
    s.bytes = off;
    return s;
}

int validate(const afm_cdm_weights* w, int B, int N) {
    if (!w || B < 0 || N <= 0) return AFM_E_BADARG;
    if (w->dkv != 256 || w->dq <= 0 || w->dq > MAXD || (w->dq & 3) || w->text_dim > MAXD || w->time_dim > MAXD) return AFM_E_UNSUPPORTED;
    if (w->enc_heads != 8 || w->dec_heads != 8 || w->n_self < 0 || w->n_self > 4) return AFM_E_UNSUPPORTED;
    if (w->feat_dim <= 0 || w->contact_dim <= 0 || w->n_timesteps <= 0) return AFM_E_BADARG;
    // the batched latent chain: one wave per 16 tokens x 16 outputs (a head's outputs are whole tiles)
    if ((w->dq % 128) != 0 || (w->dq / w->enc_heads) % TL_OB != 0) return AFM_E_UNSUPPORTED;
    return 0;
}

#define AFM_TRY(expr) do { int rc__ = (expr); if (rc__ != 0) return rc__; } while (0)

}  // namespace

#ifdef AFM_TOKLIN_TIMELINE
extern "C" int afm_debug_toklin_timeline(unsigned long long* host_out128) {
    return (int)hipMemcpyFromSymbol(host_out128, HIP_SYMBOL(afm_tk_tl), 128 * sizeof(unsigned long long));
}
#endif

extern "C" int64_t afm_cdm_workspace_bytes(const afm_cdm_weights* w, int32_t B, int32_t N) {
    if (validate(w, B, N) != 0) return AFM_E_BADARG;
    return carve(*w, B, N, nullptr).bytes;
}

extern "C" int afm_cdm_latent_tokens(const afm_cdm_weights* wp, int32_t which, const float* in, int32_t n, float* q0_out,
                                     float* u_out, float* cu_out, void* stream) {
    AFM_TRY(validate(wp, 0, 1));
    if (!in || !q0_out || !u_out || !cu_out || n < 0 || (which != 0 && which != 1)) return AFM_E_BADARG;
    if (n == 0) return 0;
    const afm_cdm_weights& w = *wp;
    const int in_dim = which == 0 ? w.text_dim : w.time_dim;
    if (in_dim & 3) return AFM_E_UNSUPPORTED;
    AfmProf prof(AFM_PROF_CDM, 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(latent_token_kernel, dim3(n), dim3(1024), 0, (hipStream_t)stream, w, in, in_dim,
                       which == 0 ? w.language_adapter : w.time_embedding_adapter, q0_out, u_out, cu_out);
    AFM_CHECK_LAUNCH();
    return 0;
}

namespace {

// out[n, j] = (((p0 + p1) + p2) + p3) + s1[n, j] + (E[n, j] + q[j] . x_t[n]) + c0[j] from the row-dot partials of the fc1 GEMM, optional DDPM
// update IN PLACE.  Every output channel needs the point's whole contact row (through q), so a block owns WHOLE rows (256 / cd of
// them per trip, one thread per element) and all its reads of x_t happen before a barrier, its writes after.
__global__ __launch_bounds__(256) void cdm_output_kernel(const float* __restrict__ rdot, int ngrp, const float* __restrict__ s1,
                                                         const float* __restrict__ qe, const float* __restrict__ fq,
                                                         const float* __restrict__ c0, int cd, int64_t rows, int rows_per_sample,
                                                         float* __restrict__ x0_out, const float* xt, const float* __restrict__ noise,
                                                         float* x_next, const float* __restrict__ c1, const float* __restrict__ c2,
                                                         const float* __restrict__ sigma) {
    const int rpb = 256 / cd;                                     // rows per block and trip
    const int lr = threadIdx.x / cd, j = threadIdx.x - lr * cd;
    const bool act = lr < rpb;
    for (int64_t r0 = (int64_t)blockIdx.x * rpb; r0 < rows; r0 += (int64_t)gridDim.x * rpb) {      // uniform trip count per block
        const int64_t r = r0 + lr;
        const bool ok = act && r < rows;
        const int64_t i = r * cd + j;
        float v = 0.f, xj = 0.f;
        if (ok) {
            v = rdot[(r * ngrp) * cd + j];                                         // w2 . GELU(linear1 z), 64 columns per partial
            for (int g = 1; g < ngrp; ++g) v += rdot[(r * ngrp + g) * cd + j];
            float q;                                                                // contact_layer.w . decoder query = invariant part + x_t part
            {
                q = qe[i];
                for (int k = 0; k < cd; ++k) {
                    const float xk = xt[r * cd + k];
                    q += xk * fq[j * cd + k];
                    if (k == j) xj = xk;
                }
            }
            v = ((v + s1[i]) + q) + c0[j];                                          // + attention part of contact_layer.w . h1 + constants
        }
        __syncthreads();                                                            // every read of this block's rows of x_t is done
        if (ok) {
            if (x0_out) x0_out[i] = v;
            if (x_next) {
                const int b = (int)(r / rows_per_sample);
                x_next[i] = (c1[b] * v + c2[b] * xj) + sigma[b] * noise[i];
            }
        }
    }
}

// per-point kernels: mode 0 = rows from memory (layer-by-layer form), 1 = FOLD (step-invariant part materialised once per loop + contact
// columns), 3 = no rows at all (enc_point_kernel / dec_point_kernel: inputs of at most 11 channels)
int launch_enc_reduce(const afm_cdm_weights& w, const float* rows, const float* text_u, const float* text_cu, const int64_t* t, int B, int N,
                      const CdmWs& ws, const float* x_t, const float* feat, int mode, hipStream_t s) {
    AfmProf prof(AFM_PROF_CDM, 0.0, s);
    constexpr int LDS = ERM_LDS_FLOATS * (int)sizeof(float);
    static const int attr = []() {
        int rc = (int)hipFuncSetAttribute((const void*)enc_reduce_mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        return rc ? rc : (int)hipFuncSetAttribute((const void*)enc_reduce_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }();
    if (attr != 0) return attr;
    if (mode == 3) {                                  // no rows at all: 12-vector partials in ws.pacc (stride 16)
        if (rowless_nks(w.feat_dim) == 3)
            hipLaunchKernelGGL(enc_point_kernel<3>, dim3(EP_SPLIT, B), dim3(64 * EP_WAVES), 0, s, w.enc_kv_norm, text_u, text_cu, w.time_u, w.time_cu, t, w.n_timesteps, N,
                               ws.pm, ws.pl, ws.pacc, x_t, w.contact_dim, feat, w.feat_dim, w.enc_ec, w.enc_qee);
        else
            hipLaunchKernelGGL(enc_point_kernel<11>, dim3(EP_SPLIT, B), dim3(64 * EP_WAVES), 0, s, w.enc_kv_norm, text_u, text_cu, w.time_u, w.time_cu, t, w.n_timesteps, N,
                               ws.pm, ws.pl, ws.pacc, x_t, w.contact_dim, feat, w.feat_dim, w.enc_ec, w.enc_qee);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    const dim3 grid(ERM_SPLIT, B), block(64 * ERM_WAVES);
    if (mode == 1) hipLaunchKernelGGL(enc_reduce_mfma_kernel<1>, grid, block, LDS, s, rows, w.enc_kv_norm, text_u, text_cu, w.time_u, w.time_cu, t,
                                      w.n_timesteps, N, ws.pm, ws.pl, ws.pacc, x_t, w.fold_xu, w.contact_dim);
    else hipLaunchKernelGGL(enc_reduce_mfma_kernel<0>, grid, block, LDS, s, rows, w.enc_kv_norm, text_u, text_cu, w.time_u, w.time_cu, t,
                            w.n_timesteps, N, ws.pm, ws.pl, ws.pacc, (const float*)nullptr, (const float*)nullptr, 0);
    AFM_CHECK_LAUNCH();
    return 0;
}

int launch_dec_attend(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, const float* feat, int mode, hipStream_t s) {
    AfmProf prof(AFM_PROF_CDM, 0.0, s);
    constexpr int LDS = DAM_LDS_FLOATS * (int)sizeof(float);
    static const int attr = []() {
        int rc = (int)hipFuncSetAttribute((const void*)dec_attend_mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        return rc ? rc : (int)hipFuncSetAttribute((const void*)dec_attend_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }();
    if (attr != 0) return attr;
    int chunks = (N + 511) / 512;                                  // 512 points per workgroup: the per-sample tables are staged once per 8 tiles and wave
    if (chunks > 16) chunks = 16;
    const dim3 grid(chunks, B), block(256);
    if (mode == 1) hipLaunchKernelGGL(dec_attend_mfma_kernel<1>, grid, block, LDS, s, ws.bufB, ws.dec_lat, w.dec_q_norm, w.dec_attn.o.b, w.dec_mlp.norm, N,
                                      (float*)nullptr, ws.z, x_t, w.fold_xv, w.contact_dim, ws.s1);
    else hipLaunchKernelGGL(dec_attend_mfma_kernel<0>, grid, block, LDS, s, ws.bufB, ws.dec_lat, w.dec_q_norm, w.dec_attn.o.b, w.dec_mlp.norm, N,
                            ws.h1, ws.z, (const float*)nullptr, (const float*)nullptr, 0, (float*)nullptr);
    AFM_CHECK_LAUNCH();
    return 0;
}

int launch_toklin(const TokLin& p, hipStream_t s);

template <int NKS>
int launch_dec_point_t(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, const float* feat, float* x0_out,
                       const afm_ddpm_args* ddpm, hipStream_t s);

// the fused decoder (mode 3): the per-sample tables of the step (one launch), then one kernel over the points
int launch_dec_point(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, const float* feat, float* x0_out,
                     const afm_ddpm_args* ddpm, hipStream_t s) {
    AfmProf prof(AFM_PROF_CDM, 0.0, s);
    return rowless_nks(w.feat_dim) == 3 ? launch_dec_point_t<3>(w, B, N, ws, x_t, feat, x0_out, ddpm, s) : launch_dec_point_t<11>(w, B, N, ws, x_t, feat, x0_out, ddpm, s);
}

template <int NKS>
int launch_dec_point_t(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, const float* feat, float* x0_out,
                       const afm_ddpm_args* ddpm, hipStream_t s) {
    constexpr int LDS = dp_lds_floats<NKS>() * (int)sizeof(float);
    static_assert(LDS <= 160 * 1024, "dec_point_kernel's tables fit the LDS");
    static const int attr = []() { return (int)hipFuncSetAttribute((const void*)dec_point_kernel<NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); }();
    if (attr != 0) return attr;
    hipLaunchKernelGGL(lat_dectables_kernel<NKS>, dim3(B), dim3(1024), 0, s, ws.lat_kv, w.dec_dwq, w.dec_wqb, w.dec_wco, w.dec_wow, w.dec_wog, w.dec_xwo,
                       w.dec_qxx, w.contact_dim, ws.twp, ws.qtab);
    AFM_CHECK_LAUNCH();
    int chunks = (N + 511) / 512;                     // 512 points per workgroup: the per-sample tables are staged once per 8 tiles and wave
    if (chunks > 16) chunks = 16;                     // (three workgroups per CU measured slower: the kernel is bound by VALU + f32 MFMA issue, not by latency)
    hipLaunchKernelGGL(dec_point_kernel<NKS>, dim3(chunks, B), dim3(64 * RowLess<NKS>::NW), LDS, s, ws.twp, ws.qtab, w.dec_qdd, w.dec_twx, w.dec_c,
                       w.fold_w2, w.gen_qe, w.fold_c0, N, w.contact_dim, x_t, feat, w.feat_dim, x0_out, ddpm ? ddpm->noise : nullptr,
                       ddpm ? ddpm->x_next : nullptr, ddpm ? ddpm->c1 : nullptr, ddpm ? ddpm->c2 : nullptr, ddpm ? ddpm->sigma : nullptr);
    AFM_CHECK_LAUNCH();
    return 0;
}

int launch_toklin(const TokLin& p, hipStream_t s) {
    const dim3 grid((p.N + TL_OB - 1) / TL_OB, (p.ntok + TL_TOK - 1) / TL_TOK);
    switch (p.K) {                                // widths of the Perceiver's latents / point features (validate: dkv == 256, dq a multiple of 128)
        case 128: hipLaunchKernelGGL(toklin_kernel<8>, grid, dim3(64), 0, s, p); break;
        case 256: hipLaunchKernelGGL(toklin_kernel<16>, grid, dim3(64), 0, s, p); break;
        case 384: hipLaunchKernelGGL(toklin_kernel<24>, grid, dim3(64), 0, s, p); break;
        case 512: hipLaunchKernelGGL(toklin_kernel<32>, grid, dim3(64), 0, s, p); break;
        default: return AFM_E_UNSUPPORTED;
    }
    AFM_CHECK_LAUNCH();
    return 0;
}

// enc_reduce partials -> dec_lat records, as 17 small launches over all 2 B latent tokens (see toklin_kernel; 14-18 us each, ~2 us apart)
int cdm_latent_chain(const afm_cdm_weights& w, const float* text_q0, const int64_t* t, const CdmWs& ws, int B, hipStream_t s, bool enc12 = false) {
    AfmProf prof(AFM_PROF_CDM, 0.0, s);
    const int ntok = 2 * B, dq = w.dq, dkv = w.dkv, He = w.enc_heads;
    const bool head = enc12;                          // fused form: combine + v-proj + o-proj as one launch (lat_head_kernel)
    if (head && rowless_nks(w.feat_dim) == 3)
        hipLaunchKernelGGL(lat_head_kernel<3>, dim3(B, 2, (dq + 255) / 256), dim3(256), 0, s, ws.pm, ws.pl, ws.pacc, text_q0, w.time_q0, t, w.n_timesteps, dq, w.enc_wove, w.enc_c1, ws.lat_x);
    else if (head)
        hipLaunchKernelGGL(lat_head_kernel<11>, dim3(B, 2, (dq + 255) / 256), dim3(256), 0, s, ws.pm, ws.pl, ws.pacc, text_q0, w.time_q0, t, w.n_timesteps, dq, w.enc_wove, w.enc_c1, ws.lat_x);
    else hipLaunchKernelGGL(lat_combine_kernel, dim3(B, 2 * He), dim3(256), 0, s, ws.pm, ws.pl, ws.pacc, 2 * He, dkv, text_q0, w.time_q0, t, w.n_timesteps, dq,
                            ws.lat_s, ws.lat_x);
    AFM_CHECK_LAUNCH();
    auto lin = [&](const float* X, int ldx, int K, const afm_lin& l, int N, float* Y, int ldy) {
        TokLin p = {};
        p.X = X; p.ldx = ldx; p.W[0] = l.w; p.b[0] = l.b; p.ncol = N; p.Y = Y; p.ldy = ldy; p.ntok = ntok; p.N = N; p.K = K;
        return p;
    };
    const bool lnf = w.lat_fold != nullptr;          // LayerNorm folded into the weights of the stages that follow one (afm_cdm_weights.lat_fold)
    auto folded = [&](TokLin& p, int part, int slot) {            // slot: index into lat_fold ([wg, g, c] triples)
        p.W[part] = w.lat_fold[3 * slot]; p.gsum[part] = w.lat_fold[3 * slot + 1]; p.b[part] = w.lat_fold[3 * slot + 2]; p.use_ln = 2;
    };
    auto mlp = [&](const afm_mlp_w& m, int slot) {    // x <- x + fc2(GELU(fc1(LN(x))))
        TokLin p = lin(ws.lat_x, dq, dq, m.fc1, dq, ws.lat_t2, dq);
        p.ln = m.norm; p.use_ln = 1; p.act = AFM_ACT_GELU;
        if (lnf) folded(p, 0, slot);
        AFM_TRY(launch_toklin(p, s));
        p = lin(ws.lat_t2, dq, dq, m.fc2, dq, ws.lat_x, dq);
        p.R = ws.lat_x; p.ldr = dq;
        return launch_toklin(p, s);
    };
    if (head) {
        AFM_TRY(mlp(w.enc_mlp, 0));
    } else {   // attention output of the encoder cross-attention: o[tok, h hd + r] = W_v[h hd + r] . s[tok, h] + b_v, then o_proj + residual, MLP
        TokLin p = lin(ws.lat_s, He * dkv, dkv, w.enc_attn.v, dq, ws.lat_t1, dq);
        p.head_out = dq / He; p.x_head_stride = dkv;
        AFM_TRY(launch_toklin(p, s));
        p = lin(ws.lat_t1, dq, dq, w.enc_attn.o, dq, ws.lat_x, dq);
        p.R = ws.lat_x; p.ldr = dq;
        AFM_TRY(launch_toklin(p, s));
        AFM_TRY(mlp(w.enc_mlp, 0));
    }
    for (int li = 0; li < w.n_self; ++li) {        // self-attention block on the two latents of every sample (modules.py:544-648)
        TokLin p = lin(ws.lat_x, dq, dq, w.self_attn[li].q, 3 * dq, ws.lat_qkv, 3 * dq);
        p.W[1] = w.self_attn[li].k.w; p.b[1] = w.self_attn[li].k.b; p.W[2] = w.self_attn[li].v.w; p.b[2] = w.self_attn[li].v.b; p.ncol = dq;
        p.ln = w.self_norm[li]; p.use_ln = 1;
        if (lnf) { folded(p, 0, 1 + 4 * li); folded(p, 1, 2 + 4 * li); folded(p, 2, 3 + 4 * li); }
        AFM_TRY(launch_toklin(p, s));
        hipLaunchKernelGGL(lat_selfattn_kernel, dim3(B), dim3(256), 0, s, ws.lat_qkv, dq, He, ws.lat_t1);
        AFM_CHECK_LAUNCH();
        p = lin(ws.lat_t1, dq, dq, w.self_attn[li].o, dq, ws.lat_x, dq);
        p.R = ws.lat_x; p.ldr = dq;
        AFM_TRY(launch_toklin(p, s));
        AFM_TRY(mlp(w.self_mlp[li], 4 + 4 * li));
    }
    {   // decoder keys / values of the two latents, folded through W_q / W_o of the decoder attention
        TokLin p = lin(ws.lat_x, dq, dq, w.dec_attn.k, 2 * dkv, ws.lat_kv, 2 * dkv);
        p.W[1] = w.dec_attn.v.w; p.b[1] = w.dec_attn.v.b; p.ncol = dkv;
        p.ln = w.dec_kv_norm; p.use_ln = 1;
        if (lnf) { folded(p, 0, 17); folded(p, 1, 18); }
        AFM_TRY(launch_toklin(p, s));
        if (!enc12) {                                 // the fused form builds its tables from lat_kv itself (lat_dectables_kernel)
            hipLaunchKernelGGL(lat_decfold_kernel, dim3(B, w.dec_heads), dim3(256), 0, s, w, ws.lat_kv, ws.dec_lat);
            AFM_CHECK_LAUNCH();
        }
    }
    return 0;
}

int cdm_latents(const afm_cdm_weights& w, const float* text_q0, const int64_t* t, const CdmWs& ws, int B, hipStream_t s, bool enc12 = false) {
    return cdm_latent_chain(w, text_q0, t, ws, B, s, enc12);
}

// sampling form of the per-point kernels: 3 = no rows (enc_point_kernel, lat_head_kernel, lat_dectables_kernel, dec_point_kernel: every fused
// table present and feat_dim + 1 <= GEN_K), 1 = FOLD (round 2: step-invariant adapter parts materialised once per loop), 0 = layer by layer
inline int cdm_mode(const afm_cdm_weights& w) {
    const bool folded = w.fold_xu && w.fold_xv && w.fold_w2 && w.fold_q && w.fold_c0 && w.contact_dim <= 8 && w.feat_dim > w.contact_dim && (w.dkv % 64) == 0;
    if (!folded) return 0;
    const bool fused = w.gen_qe && w.dec_c && w.dec_twx && w.dec_qxx && w.dec_qdd && w.enc_ec && w.enc_qee && w.enc_wove && w.enc_c1 && w.dec_dwq && w.dec_wqb &&
                       w.dec_wco && w.dec_wow && w.dec_wog && w.dec_xwo && rowless_nks(w.feat_dim) != 0 && w.enc_heads == 8 && w.dec_heads == 8 && w.dkv == 256 &&
                       !(w.flags & AFM_CDM_NO_GEN);
    return fused ? 3 : 1;
}
inline bool cdm_folded(const afm_cdm_weights& w) { return cdm_mode(w) != 0; }

// the step-invariant parts of the two adapters: C = encoder_adapter(input with x = 0) -> ws.enc_kv, D = decoder_adapter(C) -> ws.bufB
int cdm_prepare_invariants(const afm_cdm_weights& w, const float* feat, int B, int N, const CdmWs& ws, hipStream_t s) {
    const int M = B * N, dkv = w.dkv, cd = w.contact_dim;
    afm_linear_args a = {};
    a.A = feat + cd; a.lda = w.feat_dim; a.W = w.encoder_adapter.w + cd; a.ldw = w.feat_dim; a.C = ws.enc_kv; a.ldc = dkv;
    a.M = M; a.N = dkv; a.K = w.feat_dim - cd; a.bias = w.encoder_adapter.b;
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    AFM_TRY(afm_linear(&a, s));
    a = {};
    a.A = ws.enc_kv; a.lda = dkv; a.W = w.decoder_adapter.w; a.ldw = dkv; a.C = ws.bufB; a.ldc = dkv;
    a.M = M; a.N = dkv; a.K = dkv; a.bias = w.decoder_adapter.b;
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    AFM_TRY(afm_linear(&a, s));
    a = {};                                  // E = contact_layer.w . D: the step-invariant part of what the output layer sees of the query
    a.A = ws.bufB; a.lda = dkv; a.W = w.contact_layer.w; a.ldw = dkv; a.C = ws.qe; a.ldc = cd;
    a.M = M; a.N = cd; a.K = dkv;
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    return afm_linear(&a, s);
}

// one denoiser evaluation in the folded (mode 1) or generated (mode 2) form; `prepared` (mode 1): ws.enc_kv / ws.bufB already hold C / D
int cdm_forward_folded(const afm_cdm_weights& w, const float* feat, const float* x_t, const int64_t* t, const float* text_q0,
                       const float* text_u, const float* text_cu, float* x0_out, const afm_ddpm_args* ddpm, int B, int N, const CdmWs& ws,
                       bool prepared, hipStream_t s) {
    const int M = B * N, dkv = w.dkv, cd = w.contact_dim, mode = cdm_mode(w);
    if (mode == 1 && !prepared) AFM_TRY(cdm_prepare_invariants(w, feat, B, N, ws, s));
    AFM_TRY(launch_enc_reduce(w, ws.enc_kv, text_u, text_cu, t, B, N, ws, x_t, feat, mode, s));
    AFM_TRY(cdm_latents(w, text_q0, t, ws, B, s, mode == 3));
    if (mode == 3) return launch_dec_point(w, B, N, ws, x_t, feat, x0_out, ddpm, s);
    AFM_TRY(launch_dec_attend(w, B, N, ws, x_t, feat, mode, s));
    afm_linear_args a = {};                 // GELU(linear1 z) . w2 per 64-column group; the hidden activations are never stored
    a.A = ws.z; a.lda = dkv; a.W = w.dec_mlp.fc1.w; a.ldw = dkv; a.M = M; a.N = dkv; a.K = dkv; a.bias = w.dec_mlp.fc1.b; a.act = AFM_ACT_GELU;
    a.rowdot_w = w.fold_w2; a.rowdot_out = ws.rdot; a.rowdot_n = cd;
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    a.tune = ((w.flags >> AFM_CDM_TILE_SHIFT) & 0xF) << AFM_TUNE_TILE_SHIFT;
    AFM_TRY(afm_linear(&a, s));
    {
        AfmProf prof(AFM_PROF_CDM, 0.0, s);
        const int rpb = 256 / cd;
        int64_t g = ((int64_t)M + rpb - 1) / rpb; if (g > 8192) g = 8192;
        hipLaunchKernelGGL(cdm_output_kernel, dim3((unsigned)g), dim3(256), 0, s, ws.rdot, dkv / 64, ws.s1, ws.qe, w.fold_q, w.fold_c0, cd, (int64_t)M, N, x0_out, x_t,
                           ddpm ? ddpm->noise : nullptr, ddpm ? ddpm->x_next : nullptr, ddpm ? ddpm->c1 : nullptr, ddpm ? ddpm->c2 : nullptr,
                           ddpm ? ddpm->sigma : nullptr);
        AFM_CHECK_LAUNCH();
    }
    return 0;
}

}  // namespace

static int cdm_forward_impl(const afm_cdm_weights* wp, const float* feat, const float* x_t, const int64_t* t,
                            const float* text_q0, const float* text_u, const float* text_cu, float* x0_out,
                            const afm_ddpm_args* ddpm, int32_t B, int32_t N, void* workspace, int64_t workspace_bytes,
                            void* side_stream, void* stream, bool prepared = false) {
    AFM_TRY(validate(wp, B, N));
    if (!feat || !t || !text_q0 || !text_u || !text_cu || !workspace || (!x0_out && !ddpm)) return AFM_E_BADARG;
    if (!wp->time_q0 || !wp->time_u || !wp->time_cu) return AFM_E_BADARG;
    if (ddpm && (!ddpm->x_next || !ddpm->c1 || !ddpm->c2 || !ddpm->sigma || !ddpm->noise || !x_t)) return AFM_E_BADARG;
    if (B == 0) return 0;
    const afm_cdm_weights& w = *wp;
    hipStream_t s = (hipStream_t)stream;
    const CdmWs ws = carve(w, B, N, workspace);
    if (ws.bytes > workspace_bytes) return AFM_E_WORKSPACE;
    const int M = B * N, dkv = w.dkv;
    if (cdm_folded(w) && x_t) return cdm_forward_folded(w, feat, x_t, t, text_q0, text_u, text_cu, x0_out, ddpm, B, N, ws, prepared, s);

    afm_linear_args a = {};
    a.A = feat; a.lda = w.feat_dim; a.W = w.encoder_adapter.w; a.ldw = w.feat_dim; a.C = ws.enc_kv; a.ldc = dkv;
    a.M = M; a.N = dkv; a.K = w.feat_dim; a.bias = w.encoder_adapter.b;
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    AFM_TRY(afm_linear(&a, s));
    // The decoder adapter GEMM (34 GFLOP at B = 32) only needs enc_kv, while the latent chain (enc_reduce -> latent_post: one
    // workgroup per SAMPLE, a serial 0.5 ms dependency chain that leaves the chip idle) only produces the 2 latent tokens:
    // with a side stream the GEMM runs under the latent chain and the two join in front of dec_attend.
    hipStream_t side = (hipStream_t)side_stream;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    AFM_TRY(launch_enc_reduce(w, ws.enc_kv, text_u, text_cu, t, B, N, ws, nullptr, nullptr, 0, s));
    if (side) {           // fork AFTER enc_reduce (a full-chip kernel): the GEMM shares the chip with latent_post only
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
        (void)hipEventRecord(ev_fork, s);
    }
    AFM_TRY(cdm_latents(w, text_q0, t, ws, B, s));
    if (side) {           // enqueued after latent_post so that its 32 workgroups get their CUs first
        (void)hipStreamWaitEvent(side, ev_fork, 0);
        afm_linear_args d = {};
        d.A = ws.enc_kv; d.lda = dkv; d.W = w.decoder_adapter.w; d.ldw = dkv; d.C = ws.bufB; d.ldc = dkv;
        d.M = M; d.N = dkv; d.K = dkv; d.bias = w.decoder_adapter.b;
        d.arith = w.gemm_arith; d.arith_min_n = w.gemm_arith_min_n;
        const int rc = afm_linear(&d, side);
        (void)hipEventRecord(ev_join, side);
        if (rc) { (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join); return rc; }
    }
    if (side) {
        (void)hipStreamWaitEvent(s, ev_join, 0);
        (void)hipEventDestroy(ev_fork); (void)hipEventDestroy(ev_join);
    } else {
        a = {};
        a.A = ws.enc_kv; a.lda = dkv; a.W = w.decoder_adapter.w; a.ldw = dkv; a.C = ws.bufB; a.ldc = dkv;
        a.M = M; a.N = dkv; a.K = dkv; a.bias = w.decoder_adapter.b;
        a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    AFM_TRY(afm_linear(&a, s));
    }
    AFM_TRY(launch_dec_attend(w, B, N, ws, nullptr, nullptr, 0, s));
    a = {};
    a.A = ws.z; a.lda = dkv; a.W = w.dec_mlp.fc1.w; a.ldw = dkv; a.C = ws.bufB; a.ldc = dkv;
    a.M = M; a.N = dkv; a.K = dkv; a.bias = w.dec_mlp.fc1.b; a.act = AFM_ACT_GELU;
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    AFM_TRY(afm_linear(&a, s));
    a = {};
    a.A = ws.bufB; a.lda = dkv; a.W = w.dec_mlp.fc2.w; a.ldw = dkv; a.C = ws.z; a.ldc = dkv;
    a.M = M; a.N = dkv; a.K = dkv; a.bias = w.dec_mlp.fc2.b; a.residual = ws.h1; a.ldr = dkv;
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    AFM_TRY(afm_linear(&a, s));
    a = {};
    a.A = ws.z; a.lda = dkv; a.W = w.contact_layer.w; a.ldw = dkv; a.C = x0_out; a.ldc = w.contact_dim;
    a.M = M; a.N = w.contact_dim; a.K = dkv; a.bias = w.contact_layer.b;
    if (ddpm) {
        a.ddpm_xt = x_t; a.ddpm_noise = ddpm->noise; a.ddpm_out = ddpm->x_next; a.ldx = w.contact_dim;
        a.ddpm_c1 = ddpm->c1; a.ddpm_c2 = ddpm->c2; a.ddpm_sigma = ddpm->sigma; a.rows_per_sample = N;
    }
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    AFM_TRY(afm_linear(&a, s));
    return 0;
}

extern "C" int afm_cdm_forward(const afm_cdm_weights* wp, const float* feat, const float* x_t, const int64_t* t,
                               const float* text_q0, const float* text_u, const float* text_cu, float* x0_out,
                               const afm_ddpm_args* ddpm, int32_t B, int32_t N, void* workspace, int64_t workspace_bytes,
                               void* stream) {
    return cdm_forward_impl(wp, feat, x_t, t, text_q0, text_u, text_cu, x0_out, ddpm, B, N, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int afm_cdm_forward_overlap(const afm_cdm_weights* wp, const float* feat, const float* x_t, const int64_t* t,
                                       const float* text_q0, const float* text_u, const float* text_cu, float* x0_out,
                                       const afm_ddpm_args* ddpm, int32_t B, int32_t N, void* workspace, int64_t workspace_bytes,
                                       void* side_stream, void* stream) {
    return cdm_forward_impl(wp, feat, x_t, t, text_q0, text_u, text_cu, x0_out, ddpm, B, N, workspace, workspace_bytes, side_stream, stream);
}

// ------------------------------------------------------------------------------------------------ native sampling loop
namespace {

__global__ void cdm_expand_schedule_kernel(const int64_t* __restrict__ tmap, const float* __restrict__ c1, const float* __restrict__ c2,
                                           const float* __restrict__ sg, int n_steps, int B, int64_t* __restrict__ t_all,
                                           float* __restrict__ c1_all, float* __restrict__ c2_all, float* __restrict__ sg_all) {
    const int64_t n = (int64_t)n_steps * B;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = n_steps - 1 - (int)(e / B);
        t_all[e] = tmap[i]; c1_all[e] = c1[i]; c2_all[e] = c2[i]; sg_all[e] = sg[i];
    }
}

// feat[r, 0:cd] = x[r, :]  (the noisy contact map is the leading block of the encoder input, cdm.py:167-171)
__global__ __launch_bounds__(256) void pack_x_kernel(const float* __restrict__ x, float* __restrict__ feat, int64_t rows, int cd, int fd) {
    const int64_t n = rows * cd;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cd;
        feat[r * fd + (i - r * cd)] = x[i];
    }
}

inline void cdm_sub_range(int B, int nsub, int s, int* start, int* count) {
    const int base = B / nsub, extra = B % nsub;
    *start = s * base + (s < extra ? s : extra);
    *count = base + (s < extra ? 1 : 0);
}

}  // namespace

extern "C" int64_t afm_cdm_loop_workspace_bytes(const afm_cdm_weights* w, int32_t B, int32_t N, int32_t n_sub) {
    if (validate(w, B, N) != 0 || n_sub < 0) return AFM_E_BADARG;
    int nsub = n_sub > 1 ? (n_sub < B ? n_sub : B) : 1;
    if (nsub > 8) nsub = 8;
    int64_t total = 0;
    for (int s = 0; s < nsub; ++s) {
        int st, cnt;
        cdm_sub_range(B, nsub, s, &st, &cnt);
        total += carve(*w, cnt, N, nullptr).bytes + align256((int64_t)cnt * N * w->contact_dim * 4);
    }
    return total;
}

// Whole p_sample_loop of the ADM (gaussian_diffusion.py:442-536) enqueued natively: x [B,N,contact_dim] holds x_T on entry and the
// sample on exit; feat [B,N,feat_dim] holds the step-invariant columns (point features, xyz) - its leading contact_dim columns are
// rewritten from x every step.  Sub-batch s runs on streams[2s] with streams[2s+1] as the side stream of its decoder-adapter GEMM
// (n_sub <= 1: everything on `stream`, streams[0] = optional side stream).
static int cdm_sample_loop_impl(const afm_cdm_weights* w, float* x, float* feat, const float* text_q0, const float* text_u,
                                const float* text_cu, const float* step_noise, const int64_t* d_timestep_map, const float* d_c1,
                                const float* d_c2, const float* d_sigma, int32_t n_steps, int32_t first_step, uint64_t seed,
                                int64_t sample_index0, int32_t B, int32_t N, void* sched_scratch, void* workspace, int64_t workspace_bytes,
                                int32_t n_sub, void* const* streams, void* stream) {
    AFM_TRY(validate(w, B, N));
    if (!x || !feat || !text_q0 || !text_u || !text_cu || !d_timestep_map || !d_c1 || !d_c2 || !d_sigma || n_steps <= 0 || !sched_scratch ||
        !workspace || n_sub < 0 || (n_sub > 1 && !streams))
        return AFM_E_BADARG;
    if (B == 0) return 0;
    hipStream_t s0 = (hipStream_t)stream;
    int nsub = n_sub > 1 ? (n_sub < B ? n_sub : B) : 1;
    if (nsub > 8) nsub = 8;
    const int cd = w->contact_dim, fd = w->feat_dim, He = w->enc_heads, dkv = w->dkv, dq = w->dq;

    char* sp = (char*)sched_scratch;
    const int64_t nb = (int64_t)n_steps * B;
    int64_t* t_all = (int64_t*)sp; sp += align256(nb * 8);
    float* c1_all = (float*)sp; sp += align256(nb * 4);
    float* c2_all = (float*)sp; sp += align256(nb * 4);
    float* sg_all = (float*)sp;
    hipLaunchKernelGGL(cdm_expand_schedule_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s0, d_timestep_map, d_c1, d_c2, d_sigma, n_steps, B,
                       t_all, c1_all, c2_all, sg_all);
    AFM_CHECK_LAUNCH();

    int start[8], count[8];
    char* wsp[8];
    int64_t wsb[8];
    float* noise[8];
    hipStream_t mainst[8], sidest[8];
    {
        char* base = (char*)workspace;
        int64_t off = 0;
        for (int s = 0; s < nsub; ++s) {
            cdm_sub_range(B, nsub, s, &start[s], &count[s]);
            wsb[s] = carve(*w, count[s], N, nullptr).bytes;
            wsp[s] = base + off; off += wsb[s];
            noise[s] = (float*)(base + off); off += align256((int64_t)count[s] * N * cd * 4);
            if (nsub > 1) { mainst[s] = (hipStream_t)streams[2 * s]; sidest[s] = (hipStream_t)streams[2 * s + 1]; }
            else { mainst[s] = s0; sidest[s] = streams ? (hipStream_t)streams[0] : nullptr; }
        }
        if (off > workspace_bytes) return AFM_E_WORKSPACE;
    }
    hipEvent_t fork = nullptr;
    if (nsub > 1) {
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
        (void)hipEventRecord(fork, s0);
        for (int s = 0; s < nsub; ++s) (void)hipStreamWaitEvent(mainst[s], fork, 0);
    }
    const int64_t per = (int64_t)N * cd;
    int rc = 0;
    // folded form: the step-invariant parts of the two adapters are computed once for the whole range of steps and x_t is read where
    // it is needed - no per-step rewrite of the input block, no adapter GEMMs inside the loop
    const bool folded = cdm_folded(*w);
    if (cdm_mode(*w) == 1) {                   // (the generated form has nothing to prepare)
        for (int s = 0; s < nsub && rc == 0; ++s) {
            if (count[s] == 0) continue;
            rc = cdm_prepare_invariants(*w, feat + (int64_t)start[s] * N * fd, count[s], N, carve(*w, count[s], N, wsp[s]), mainst[s]);
        }
    }
    for (int j = 0; j < n_steps && rc == 0; ++j) {
        for (int s = 0; s < nsub && rc == 0; ++s) {
            if (count[s] == 0) continue;
            float* xs = x + (int64_t)start[s] * per;
            float* fs = feat + (int64_t)start[s] * N * fd;
            const int64_t rows = (int64_t)count[s] * N;
            int64_t gx = (rows * cd + 255) / 256; if (gx > 2048) gx = 2048;
            if (!folded) hipLaunchKernelGGL(pack_x_kernel, dim3((unsigned)gx), dim3(256), 0, mainst[s], xs, fs, rows, cd, fd);
            afm_ddpm_args dd = {};
            if (step_noise) dd.noise = step_noise + ((int64_t)j * B + start[s]) * per;
            else {
                rc = afm_randn(noise[s], count[s], per, seed, sample_index0 + start[s], first_step + j, mainst[s]);
                if (rc) break;
                dd.noise = noise[s];
            }
            dd.x_next = xs;                               // in place: each element is read then written by the same lane
            dd.c1 = c1_all + (int64_t)j * B + start[s]; dd.c2 = c2_all + (int64_t)j * B + start[s]; dd.sigma = sg_all + (int64_t)j * B + start[s];
            dd.seed = seed; dd.sample_index0 = sample_index0 + start[s]; dd.step = first_step + j;
            rc = cdm_forward_impl(w, fs, xs, t_all + (int64_t)j * B + start[s], text_q0 + (int64_t)start[s] * dq,
                                  text_u + (int64_t)start[s] * He * dkv, text_cu + (int64_t)start[s] * He, nullptr, &dd, count[s], N, wsp[s], wsb[s],
                                  sidest[s], mainst[s], folded);
        }
    }
    if (nsub > 1) {
        for (int s = 0; s < nsub; ++s) {
            hipEvent_t done;
            if (hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess) {
                (void)hipEventRecord(done, mainst[s]);
                (void)hipStreamWaitEvent(s0, done, 0);
                (void)hipEventDestroy(done);
            }
        }
        (void)hipEventDestroy(fork);
    }
    return rc;
}

extern "C" int afm_cdm_sample_loop(const afm_cdm_weights* w, float* x, float* feat, const float* text_q0, const float* text_u,
                                   const float* text_cu, const float* step_noise, const int64_t* d_timestep_map, const float* d_c1,
                                   const float* d_c2, const float* d_sigma, int32_t n_steps, uint64_t seed, int64_t sample_index0, int32_t B,
                                   int32_t N, void* sched_scratch, void* workspace, int64_t workspace_bytes, int32_t n_sub, void* const* streams,
                                   void* stream) {
    return cdm_sample_loop_impl(w, x, feat, text_q0, text_u, text_cu, step_noise, d_timestep_map, d_c1, d_c2, d_sigma, n_steps, 0, seed,
                                sample_index0, B, N, sched_scratch, workspace, workspace_bytes, n_sub, streams, stream);
}

extern "C" int afm_cdm_sample_loop_range(const afm_cdm_weights* w, float* x, float* feat, const float* text_q0, const float* text_u,
                                         const float* text_cu, const float* step_noise, const int64_t* d_timestep_map, const float* d_c1,
                                         const float* d_c2, const float* d_sigma, int32_t n_steps, int32_t first_step, uint64_t seed,
                                         int64_t sample_index0, int32_t B, int32_t N, void* sched_scratch, void* workspace,
                                         int64_t workspace_bytes, int32_t n_sub, void* const* streams, void* stream) {
    if (first_step < 0) return AFM_E_BADARG;
    return cdm_sample_loop_impl(w, x, feat, text_q0, text_u, text_cu, step_noise, d_timestep_map, d_c1, d_c2, d_sigma, n_steps, first_step,
                                seed, sample_index0, B, N, sched_scratch, workspace, workspace_bytes, n_sub, streams, stream);
}
