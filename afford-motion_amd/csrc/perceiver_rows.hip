// CDM / ContactPerceiver denoiser: the per-point kernels of the forms that READ rows (reference models/cdm.py:155-188, modules.py:234-661) -
// layer-by-layer (mode 0: the training-shaped forward, any feature width) and FOLD (mode 1, round 2: the step-invariant parts of the two
// adapters materialised once per loop).  The row-less form of the sampling loop is perceiver_points.hip.
#include "perceiver_internal.h"

using namespace afm_cdm;

namespace {

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

// out[n, j] = (((p0 + p1) + p2) + p3) + s1[n, j] + (E[n, j] + q[j] . x_t[n]) + c0[j] from the row-dot partials of the fc1 GEMM, optional DDPM
// update IN PLACE.  Every output channel needs the point's whole contact row (through q), so a block owns WHOLE rows (256 / cd of
// them per trip, one thread per element) and all its reads of x_t happen before a barrier, its writes after.
__global__ __launch_bounds__(256) void cdm_output_kernel(const float* __restrict__ rdot, int ngrp, const float* __restrict__ s1,
                                                         const float* __restrict__ qe, const float* __restrict__ fq,
                                                         const float* __restrict__ c0, int cd, int64_t rows, int rows_per_sample,
                                                         float* __restrict__ x0_out, const float* xt, const float* __restrict__ noise,
                                                         float* x_next, const float* __restrict__ c1, const float* __restrict__ c2,
                                                         const float* __restrict__ sigma, int clip) {
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
            if (clip) v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);               // clip_denoised
            if (x0_out) x0_out[i] = v;
            if (x_next) {
                const int b = (int)(r / rows_per_sample);
                x_next[i] = (c1[b] * v + c2[b] * xj) + sigma[b] * noise[i];
            }
        }
    }
}

}  // namespace

namespace afm_cdm {

int launch_enc_reduce(const afm_cdm_weights& w, const float* rows, const float* text_u, const float* text_cu, const int64_t* t, int B, int N,
                      const CdmWs& ws, const float* x_t, int mode, hipStream_t s) {
    AfmProf prof(AFM_PROF_CDM, 0.0, s);
    constexpr int LDS = ERM_LDS_FLOATS * (int)sizeof(float);
    static const int attr = []() {
        int rc = (int)hipFuncSetAttribute((const void*)enc_reduce_mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        return rc ? rc : (int)hipFuncSetAttribute((const void*)enc_reduce_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    }();
    if (attr != 0) return attr;
    const dim3 grid(ERM_SPLIT, B), block(64 * ERM_WAVES);
    if (mode == 1) hipLaunchKernelGGL(enc_reduce_mfma_kernel<1>, grid, block, LDS, s, rows, w.enc_kv_norm, text_u, text_cu, w.time_u, w.time_cu, t,
                                      w.n_timesteps, N, ws.pm, ws.pl, ws.pacc, x_t, w.fold_xu, w.contact_dim);
    else hipLaunchKernelGGL(enc_reduce_mfma_kernel<0>, grid, block, LDS, s, rows, w.enc_kv_norm, text_u, text_cu, w.time_u, w.time_cu, t,
                            w.n_timesteps, N, ws.pm, ws.pl, ws.pacc, (const float*)nullptr, (const float*)nullptr, 0);
    AFM_CHECK_LAUNCH();
    return 0;
}

int launch_dec_attend(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, int mode, hipStream_t s) {
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

int launch_cdm_output(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, float* x0_out, const afm_ddpm_args* ddpm, hipStream_t s) {
    AfmProf prof(AFM_PROF_CDM, 0.0, s);
    const int M = B * N, cd = w.contact_dim, rpb = 256 / cd;
    int64_t g = ((int64_t)M + rpb - 1) / rpb; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(cdm_output_kernel, dim3((unsigned)g), dim3(256), 0, s, ws.rdot, w.dkv / 64, ws.s1, ws.qe, w.fold_q, w.fold_c0, cd, (int64_t)M, N, x0_out, x_t,
                       ddpm ? ddpm->noise : nullptr, ddpm ? ddpm->x_next : nullptr, ddpm ? ddpm->c1 : nullptr, ddpm ? ddpm->c2 : nullptr,
                       ddpm ? ddpm->sigma : nullptr, (ddpm && (w.flags & AFM_CDM_CLIP_X0)) ? 1 : 0);
    AFM_CHECK_LAUNCH();
    return 0;
}

int launch_lat_combine(const afm_cdm_weights& w, const float* text_q0, const int64_t* t, const CdmWs& ws, int B, hipStream_t s) {
    const int He = w.enc_heads;
    hipLaunchKernelGGL(lat_combine_kernel, dim3(B, 2 * He), dim3(256), 0, s, ws.pm, ws.pl, ws.pacc, 2 * He, w.dkv, text_q0, w.time_q0, t, w.n_timesteps, w.dq,
                       ws.lat_s, ws.lat_x);
    AFM_CHECK_LAUNCH();
    return 0;
}

int launch_lat_decfold(const afm_cdm_weights& w, const CdmWs& ws, int B, hipStream_t s) {
    hipLaunchKernelGGL(lat_decfold_kernel, dim3(B, w.dec_heads), dim3(256), 0, s, w, ws.lat_kv, ws.dec_lat);
    AFM_CHECK_LAUNCH();
    return 0;
}

}  // namespace afm_cdm
