// CDM / ContactPerceiver denoiser: the chain of the two latent tokens of every sample between the reduction over the points and the decoder
// (reference models/modules.py:234-661: cross-attention output, o_proj, MLP, self-attention blocks, decoder keys / values), batched over
// the samples as a sequence of small launches.
#include "perceiver_internal.h"

using namespace afm_cdm;

namespace {

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
struct TokLin {
    const float* X; int ldx;                 // input rows: token tok at X + tok * ldx (+ head offset)
    int head_out, x_head_stride;             // head_out > 0: outputs [h * head_out, (h + 1) * head_out) read X + h * x_head_stride (per-head inputs)
    const float* W[3]; const float* b[3];    // up to three stacked weight matrices [ncol, K] (q | k | v), ncol outputs each
    int ncol;
    afm_ln ln; int use_ln;                   // 1: LayerNorm (eps 1e-5) of the input rows; 2: the same FOLDED - W carries gamma, b carries W beta,
    const float* gsum[3];                    //    gsum[part][o] = sum_k W[o][k]: Y = rstd (W x - mean gsum) + b, the products do not wait for the statistics
    int act;                                 // AFM_ACT_*
    int attn_hd;                             // > 0: X is [q | k | v] rows (ldx = 3 K) of token PAIRS and the input row is the self-attention of the pair (head size attn_hd)
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

// NK16 = K / 16 at compile time (a run-time bound on the unrolled register arrays turns every step into compute-and-select).
// Workgroup = FOUR waves on one 16 x 16 output tile, wave w taking the K quarter [w K / 4, (w + 1) K / 4): a single wave spent 1.3 us issuing
// its 2 x K / 16 loads, 1.8 us issuing K / 4 MFMAs on one SIMD (32 cycles each) and 1.3 us on an epilogue whose side inputs (bias, row sums,
// residual) were requested after the product (profiles/r03_toklin_timeline.txt).  Now every wave issues a quarter of the loads and of the
// MFMAs, the epilogue's inputs are requested at kernel entry, the partial tiles and the per-quarter LayerNorm statistics (mean, M2 about that
// mean: combined with Chan's formula in wave order) meet in LDS behind ONE barrier and wave w finishes token 4 g + w of every lane's four.
template <int NK16, bool ATTN = false>
__global__ __launch_bounds__(256) void toklin_kernel(const TokLin p) {
    TKTL(0);
    static_assert(NK16 % 4 == 0, "K is split over the four waves in whole K16 steps");
    constexpr int NQ = NK16 / 4;
    __shared__ float part[4][64][4], stat[4][16][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, p16 = lane & 15, g = lane >> 4;
    const int tb = blockIdx.y * TL_TOK, o0 = blockIdx.x * TL_OB;
    const int part0 = o0 / p.ncol, oc0 = o0 - part0 * p.ncol;    // the TL_OB outputs of a workgroup lie in one weight part (ncol % TL_OB == 0)
    const bool ovalid = o0 + p16 < p.N;
    // A operand: token tb + p16 (clamped: rows past the end are computed and dropped); B operand: weight row o0 + p16; this wave's K quarter
    const float* xrow = p.X + (p.head_out ? (o0 / p.head_out) * p.x_head_stride : 0) + (int64_t)min(tb + p16, p.ntok - 1) * p.ldx + 4 * g + 16 * NQ * wave;
    const float* wrow = p.W[part0] + (int64_t)(ovalid ? oc0 + p16 : 0) * p.K + 4 * g + 16 * NQ * wave;
    float4 xr[NQ], wr[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        xr[u] = *reinterpret_cast<const float4*>(xrow + 16 * u);
        wr[u] = *reinterpret_cast<const float4*>(wrow + 16 * u);
    }
    if constexpr (ATTN) {
        // Self-attention of the two latent tokens of a sample (modules.py:544-648) as the prologue of its o_proj: the input row of token 2 b + i is
        // sum_j softmax_j(q_i . k_j / sqrt(hd)) v_j per head, the K quarter of a wave is whole heads (K / 4 = 2 heads of the reference's 8), and a
        // token's channels of a head sit in the four lanes of its row: xr holds q_i here; k and v of both tokens of the pair come the same way.
        constexpr int SH = NQ / 2;                                // K16 steps per head: NQ steps = two heads
        const float* kb = xrow - (int64_t)(min(tb + p16, p.ntok - 1) & 1) * p.ldx + p.K;      // k of token 2 b (same channels); + ldx: token 2 b + 1; + K: v
        float4 k0[NQ], k1[NQ], v0[NQ], v1[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            k0[u] = *reinterpret_cast<const float4*>(kb + 16 * u);
            k1[u] = *reinterpret_cast<const float4*>(kb + p.ldx + 16 * u);
            v0[u] = *reinterpret_cast<const float4*>(kb + p.K + 16 * u);
            v1[u] = *reinterpret_cast<const float4*>(kb + p.ldx + p.K + 16 * u);
        }
        const float scale = 1.0f / sqrtf((float)p.attn_hd);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int u = h * SH; u < (h + 1) * SH; ++u) {
                const float4 q = make_float4(xr[u].x * scale, xr[u].y * scale, xr[u].z * scale, xr[u].w * scale);      // q * dp_scale (modules.py:330)
                s0 += (q.x * k0[u].x + q.y * k0[u].y) + (q.z * k0[u].z + q.w * k0[u].w);
                s1 += (q.x * k1[u].x + q.y * k1[u].y) + (q.z * k1[u].z + q.w * k1[u].w);
            }
            s0 += xor16(s0); s0 += xor32(s0);
            s1 += xor16(s1); s1 += xor32(s1);
            const float mx = fmaxf(s0, s1), e0 = __expf(s0 - mx), e1 = __expf(s1 - mx), inv = 1.0f / (e0 + e1), a0 = e0 * inv, a1 = e1 * inv;
#pragma unroll
            for (int u = h * SH; u < (h + 1) * SH; ++u)
                xr[u] = make_float4(a0 * v0[u].x + a1 * v1[u].x, a0 * v0[u].y + a1 * v1[u].y, a0 * v0[u].z + a1 * v1[u].z, a0 * v0[u].w + a1 * v1[u].w);
        }
    }
    // the epilogue's side inputs of this lane's element (output o0 + p16 of token tb + 4 g + wave), requested with the operands
    const int tok = tb + 4 * g + wave, ocol = ovalid ? oc0 + p16 : 0;
    const bool tvalid = ovalid && tok < p.ntok;
    const float bias = p.b[part0] ? p.b[part0][ocol] : 0.f;
    const float gs = p.use_ln == 2 ? p.gsum[part0][ocol] : 0.f;
    const float res = (p.R && tvalid) ? p.R[(int64_t)tok * p.ldr + o0 + p16] : 0.f;
    TKTL(1);
    if (p.use_ln) {                                              // uniform.  This quarter of the row of token p16: (mean, M2 about it) over the row's four lanes
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < NQ; ++u) sum += (xr[u].x + xr[u].y) + (xr[u].z + xr[u].w);
        sum += xor16(sum); sum += xor32(sum);
        const float mq = sum / (float)(16 * NQ);
        float sq = 0.f;
#pragma unroll
        for (int u = 0; u < NQ; ++u) { const float a = xr[u].x - mq, b = xr[u].y - mq, c = xr[u].z - mq, d = xr[u].w - mq; sq += (a * a + b * b) + (c * c + d * d); }
        sq += xor16(sq); sq += xor32(sq);
        if (g == 0) { stat[wave][p16][0] = mq; stat[wave][p16][1] = sq; }
        if (p.use_ln == 1) {                                     // uniform: unfolded form, the rows are normalised before the product
            __syncthreads();
            const float mean = ((stat[0][p16][0] + stat[1][p16][0]) + (stat[2][p16][0] + stat[3][p16][0])) * 0.25f;
            float m2 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) { const float d = stat[w][p16][0] - mean; m2 += stat[w][p16][1] + (float)(16 * NQ) * (d * d); }
            const float rstd = 1.0f / sqrtf(m2 / (float)p.K + 1e-5f);
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const float4 gg = *reinterpret_cast<const float4*>(p.ln.g + 16 * (NQ * wave + u) + 4 * g), bb = *reinterpret_cast<const float4*>(p.ln.b + 16 * (NQ * wave + u) + 4 * g);
                xr[u] = make_float4((xr[u].x - mean) * rstd * gg.x + bb.x, (xr[u].y - mean) * rstd * gg.y + bb.y,
                                    (xr[u].z - mean) * rstd * gg.z + bb.z, (xr[u].w - mean) * rstd * gg.w + bb.w);
            }
        }
    }
    TKTL(2);
    // ---- D[i = token][j = output] += X[i][k] W[j][k]: MFMA e of step u takes k = 16 u + 4 (l >> 4) + e on both operands; four accumulators
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[u].x, wr[u].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[u].y, wr[u].y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[u].z, wr[u].z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(xr[u].w, wr[u].w, acc[3], 0, 0, 0);
    }
    TKTL(3);
    *reinterpret_cast<float4*>(&part[wave][lane][0]) = make_float4((acc[0][0] + acc[1][0]) + (acc[2][0] + acc[3][0]), (acc[0][1] + acc[1][1]) + (acc[2][1] + acc[3][1]),
                                                                   (acc[0][2] + acc[1][2]) + (acc[2][2] + acc[3][2]), (acc[0][3] + acc[1][3]) + (acc[2][3] + acc[3][3]));
    __syncthreads();
    // lane (output p16, token group g) of wave w finishes token 4 g + w: the K quarters in wave order
    float v = (part[0][lane][wave] + part[1][lane][wave]) + (part[2][lane][wave] + part[3][lane][wave]);
    if (p.use_ln == 2) {                                         // uniform: folded LayerNorm, statistics of token 4 g + wave
        const int t16 = 4 * g + wave;
        const float mean = ((stat[0][t16][0] + stat[1][t16][0]) + (stat[2][t16][0] + stat[3][t16][0])) * 0.25f;
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const float d = stat[w][t16][0] - mean; m2 += stat[w][t16][1] + (float)(16 * NQ) * (d * d); }
        const float rstd = 1.0f / sqrtf(m2 / (float)p.K + 1e-5f);
        v = rstd * (v - mean * gs);
    }
    if (tvalid) {
        v += bias;
        if (p.act) v = apply_act(v, p.act);
        if (p.R) v += res;
        p.Y[(int64_t)tok * p.ldy + o0 + p16] = v;
    }
    TKTL(4);
}

int launch_toklin(const TokLin& p, hipStream_t s) {
    const dim3 grid((p.N + TL_OB - 1) / TL_OB, (p.ntok + TL_TOK - 1) / TL_TOK);
    if (p.attn_hd) {                              // self-attention prologue: a wave's K quarter is two whole heads
        if (p.K != 8 * p.attn_hd || p.use_ln || p.head_out || p.ldx != 3 * p.K || (p.ntok & 1)) return AFM_E_UNSUPPORTED;
        switch (p.K) {
            case 128: hipLaunchKernelGGL((toklin_kernel<8, true>), grid, dim3(256), 0, s, p); break;
            case 256: hipLaunchKernelGGL((toklin_kernel<16, true>), grid, dim3(256), 0, s, p); break;
            case 384: hipLaunchKernelGGL((toklin_kernel<24, true>), grid, dim3(256), 0, s, p); break;
            case 512: hipLaunchKernelGGL((toklin_kernel<32, true>), grid, dim3(256), 0, s, p); break;
            default: return AFM_E_UNSUPPORTED;
        }
        AFM_CHECK_LAUNCH();
        return 0;
    }
    switch (p.K) {                                // widths of the Perceiver's latents / point features (validate: dkv == 256, dq a multiple of 128)
        case 128: hipLaunchKernelGGL(toklin_kernel<8>, grid, dim3(256), 0, s, p); break;
        case 256: hipLaunchKernelGGL(toklin_kernel<16>, grid, dim3(256), 0, s, p); break;
        case 384: hipLaunchKernelGGL(toklin_kernel<24>, grid, dim3(256), 0, s, p); break;
        case 512: hipLaunchKernelGGL(toklin_kernel<32>, grid, dim3(256), 0, s, p); break;
        default: return AFM_E_UNSUPPORTED;
    }
    AFM_CHECK_LAUNCH();
    return 0;
}
}  // namespace

#ifdef AFM_TOKLIN_TIMELINE
extern "C" int afm_debug_toklin_timeline(unsigned long long* host_out128) {
    return (int)hipMemcpyFromSymbol(host_out128, HIP_SYMBOL(afm_tk_tl), 128 * sizeof(unsigned long long));
}
#endif
namespace afm_cdm {

// enc_reduce partials -> dec_lat records, as 17 small launches over all 2 B latent tokens (see toklin_kernel; 14-18 us each, ~2 us apart)
int cdm_latent_chain(const afm_cdm_weights& w, const float* text_q0, const int64_t* t, const CdmWs& ws, int B, hipStream_t s, bool enc12) {
    AfmProf prof(enc12 ? AFM_PROF_CDM_CHAIN : AFM_PROF_CDM, 0.0, s);
    const int ntok = 2 * B, dq = w.dq, dkv = w.dkv, He = w.enc_heads;
    const bool head = enc12;                          // fused form: combine + v-proj + o-proj as one launch (lat_head_kernel)
    if (head) AFM_TRY(launch_lat_head(w, text_q0, t, ws, B, s));
    else AFM_TRY(launch_lat_combine(w, text_q0, t, ws, B, s));
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
        p = lin(ws.lat_qkv, 3 * dq, dq, w.self_attn[li].o, dq, ws.lat_x, dq);      // the attention of the pair is the prologue of its o_proj (toklin_kernel<., true>)
        p.attn_hd = dq / He;
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
        if (!enc12) AFM_TRY(launch_lat_decfold(w, ws, B, s));      // the row-less form builds its tables from lat_kv itself (lat_dectables_kernel)
    }
    return 0;
}

}  // namespace afm_cdm
