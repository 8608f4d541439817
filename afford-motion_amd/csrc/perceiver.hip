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
//
// This file: the C-ABI entry points (include/afm_hip.h: afm_cdm_*), the workspace, the latent-token kernel (off the per-step path), the choice
// of the sampling form and the forward / native-loop drivers.  The kernels of a step live in perceiver_rows.hip (forms that read per-point
// rows: layer by layer, FOLD), perceiver_points.hip (the row-less form of the sampling loop) and perceiver_chain.hip (the latent chain).
#include "perceiver_internal.h"

extern "C" int afm_linear(const afm_linear_args*, void*);

using namespace afm_cdm;

namespace {

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

// rows = false (the native loop in the row-less sampling form): the four [B N, dkv] row buffers and the folded form's per-point scratch are
// never touched and are not carved (about 1 GB at B = 32, N = 8192)
CdmWs carve(const afm_cdm_weights& w, int B, int N, void* base, bool rows = true) {
    char* p = (char*)base;
    int64_t off = 0;
    auto take = [&](int64_t n) { char* r = p ? p + off : nullptr; off += align256(n); return (float*)r; };
    const int64_t M = (int64_t)B * N, nih = 2 * w.enc_heads, njh = 2 * w.dec_heads;
    CdmWs s;
    const int64_t Mr = rows ? M : 0;
    s.enc_kv = take(Mr * w.dkv * 4); s.bufB = take(Mr * w.dkv * 4); s.h1 = take(Mr * w.dkv * 4); s.z = take(Mr * w.dkv * 4);
    s.pm = take((int64_t)B * NPART * nih * 4); s.pl = take((int64_t)B * NPART * nih * 4);
    s.pacc = take((int64_t)B * NPART * nih * w.dkv * 4);
    s.dec_lat = take((int64_t)B * DEC_LAT_STRIDE(njh) * 4);
    s.s1 = take(Mr * 8 * 4);                                 // folded path: contact_layer . h1 per point (<= 8 channels)
    s.rdot = take(Mr * (w.dkv / 64) * 8 * 4);                // folded path: row-dot partials of the fc1 GEMM
    s.qe = take(Mr * 8 * 4);                                 // folded path: contact_layer . (step-invariant part of the decoder query)
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
}  // namespace

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

// sampling form of the per-point kernels: 3 = no rows (enc_point_kernel, lat_head_kernel, lat_dectables_kernel, dec_point_kernel: every fused
// table present and rowless_nks(feat_dim) != 0: at most 43 input channels), 1 = FOLD (round 2: step-invariant adapter parts materialised once per loop), 0 = layer by layer
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
                       bool prepared, hipStream_t s, const CdmChainSide* cs = nullptr) {
    const int M = B * N, dkv = w.dkv, cd = w.contact_dim, mode = cdm_mode(w);
    if (mode == 1 && !prepared) AFM_TRY(cdm_prepare_invariants(w, feat, B, N, ws, s));
    if (mode == 3) AFM_TRY(launch_enc_point(w, text_u, text_cu, t, B, N, ws, x_t, feat, s));
    else AFM_TRY(launch_enc_reduce(w, ws.enc_kv, text_u, text_cu, t, B, N, ws, x_t, mode, s));
    if (mode == 3 && cs && cs->chain) {
        // the chain (13 small launches for 2 B tokens) on the sub-batch's side stream: enc_point -> [event] -> chain + tables -> [event] -> dec_point
        (void)hipEventRecord(cs->forked, s);
        (void)hipStreamWaitEvent(cs->chain, cs->forked, 0);
        AFM_TRY(cdm_latent_chain(w, text_q0, t, ws, B, cs->chain, true));
        AFM_TRY(launch_dec_tables(w, B, ws, cs->chain));
        (void)hipEventRecord(cs->joined, cs->chain);
        (void)hipStreamWaitEvent(s, cs->joined, 0);
        return launch_dec_point(w, B, N, ws, x_t, feat, x0_out, ddpm, s, false);
    }
    AFM_TRY(cdm_latent_chain(w, text_q0, t, ws, B, s, mode == 3));
    if (mode == 3) return launch_dec_point(w, B, N, ws, x_t, feat, x0_out, ddpm, s);
    AFM_TRY(launch_dec_attend(w, B, N, ws, x_t, mode, s));
    afm_linear_args a = {};                 // GELU(linear1 z) . w2 per 64-column group; the hidden activations are never stored
    a.A = ws.z; a.lda = dkv; a.W = w.dec_mlp.fc1.w; a.ldw = dkv; a.M = M; a.N = dkv; a.K = dkv; a.bias = w.dec_mlp.fc1.b; a.act = AFM_ACT_GELU;
    a.rowdot_w = w.fold_w2; a.rowdot_out = ws.rdot; a.rowdot_n = cd;
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    a.tune = ((w.flags >> AFM_CDM_TILE_SHIFT) & 0xF) << AFM_TUNE_TILE_SHIFT;
    AFM_TRY(afm_linear(&a, s));
    return launch_cdm_output(w, B, N, ws, x_t, x0_out, ddpm, s);
}

}  // namespace

static int cdm_forward_impl(const afm_cdm_weights* wp, const float* feat, const float* x_t, const int64_t* t,
                            const float* text_q0, const float* text_u, const float* text_cu, float* x0_out,
                            const afm_ddpm_args* ddpm, int32_t B, int32_t N, void* workspace, int64_t workspace_bytes,
                            void* side_stream, void* stream, bool prepared = false, bool rowless_ws = false, const CdmChainSide* cs = nullptr) {
    AFM_TRY(validate(wp, B, N));
    if (!feat || !t || !text_q0 || !text_u || !text_cu || !workspace || (!x0_out && !ddpm)) return AFM_E_BADARG;
    if (!wp->time_q0 || !wp->time_u || !wp->time_cu) return AFM_E_BADARG;
    if (ddpm && (!ddpm->x_next || !ddpm->c1 || !ddpm->c2 || !ddpm->sigma || !ddpm->noise || !x_t)) return AFM_E_BADARG;
    if (B == 0) return 0;
    const afm_cdm_weights& w = *wp;
    hipStream_t s = (hipStream_t)stream;
    if (rowless_ws && !(cdm_mode(w) == 3 && x_t)) return AFM_E_BADARG;          // a workspace without row buffers serves the row-less form only
    const CdmWs ws = carve(w, B, N, workspace, !rowless_ws);
    if (ws.bytes > workspace_bytes) return AFM_E_WORKSPACE;
    const int M = B * N, dkv = w.dkv;
    if (cdm_folded(w) && x_t) return cdm_forward_folded(w, feat, x_t, t, text_q0, text_u, text_cu, x0_out, ddpm, B, N, ws, prepared, s, cs);

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
    AFM_TRY(launch_enc_reduce(w, ws.enc_kv, text_u, text_cu, t, B, N, ws, nullptr, 0, s));
    if (side) {           // fork AFTER enc_reduce (a full-chip kernel): the GEMM shares the chip with latent_post only
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
        (void)hipEventRecord(ev_fork, s);
    }
    AFM_TRY(cdm_latent_chain(w, text_q0, t, ws, B, s, false));
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
    AFM_TRY(launch_dec_attend(w, B, N, ws, nullptr, 0, s));
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
        a.ddpm_clip = (w.flags & AFM_CDM_CLIP_X0) ? 1 : 0;
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

constexpr int NOISE_STEPS = 16;               // steps of Philox noise generated per launch of the native loop (workspace: NOISE_STEPS x B x N x contact_dim floats)

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
        total += carve(*w, cnt, N, nullptr, cdm_mode(*w) != 3).bytes + align256((int64_t)NOISE_STEPS * cnt * N * w->contact_dim * 4);
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

    const bool rowless = cdm_mode(*w) == 3;
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
            wsb[s] = carve(*w, count[s], N, nullptr, !rowless).bytes;
            wsp[s] = base + off; off += wsb[s];
            noise[s] = (float*)(base + off); off += align256((int64_t)NOISE_STEPS * count[s] * N * cd * 4);
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
    // AFM_CDM_CHAIN_SIDE: the latent chain of sub-batch s on sidest[s] (one event pair per sub-batch, reused by every step: an event
    // re-recorded on a stream orders behind the waits already enqueued on its previous record)
    CdmChainSide side[8] = {};
    const bool chain_side = rowless && nsub > 1 && (w->flags & AFM_CDM_CHAIN_SIDE);
    for (int s = 0; chain_side && s < nsub; ++s) {
        side[s].chain = sidest[s];
        if (hipEventCreateWithFlags(&side[s].forked, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&side[s].joined, hipEventDisableTiming) != hipSuccess) {
            const int rc_ev = (int)hipGetLastError();
            for (int u = 0; u <= s; ++u) {                 // nothing created so far may leak (the first event of a failed pair did)
                if (side[u].forked) (void)hipEventDestroy(side[u].forked);
                if (side[u].joined) (void)hipEventDestroy(side[u].joined);
            }
            if (fork) (void)hipEventDestroy(fork);
            return rc_ev;
        }
        (void)hipStreamWaitEvent(sidest[s], fork, 0);
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
    // ---- AFM_CDM_PIPELINE (round 6; row-less form, sub-batches): the step of a sub-batch is heavy - chain - heavy: enc_point (fills the chip,
    // ~15 us per 32 samples), the 13-launch latent chain (~70 us of launch latency on a handful of CUs), dec_point (fills the chip, ~88 us).
    // Independent sub-batch streams fall into lockstep (a stream that is behind gets the chip to itself and catches up: both chains end up
    // under each other, profiles/r04_cdm_streams.jsonl).  Here the phase is FIXED by construction: the heavy kernels of ALL sub-batches run on
    // ONE stream - the caller's - in round-robin order - dec(s, j), enc(s, j + 1) for s = 0 .. nsub - 1 - and the chain of sub-batch s on its
    // own side stream (streams[2 s + 1]) between two events, so that the chain of one sub-batch always sits under the point kernels of the
    // others.  1 + nsub streams: the runtime has four hardware queues, and two streams on one queue do not overlap whatever the events say.
    // Per-sample arithmetic does not depend on the sub-batching: bit-identical.
    const bool pipe = rowless && nsub > 1 && (w->flags & AFM_CDM_PIPELINE) && !chain_side;
    if (pipe) {
        hipStream_t H = s0;
        hipEvent_t e1[8] = {}, e2[8] = {};
        CdmWs wss[8];
        for (int s = 0; s < nsub && rc == 0; ++s) {
            wss[s] = carve(*w, count[s], N, wsp[s], false);
            if (hipEventCreateWithFlags(&e1[s], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e2[s], hipEventDisableTiming) != hipSuccess)
                rc = (int)hipGetLastError();
        }
        auto part1 = [&](int s, int j) -> int {           // noise (every NOISE_STEPS steps), enc_point on H; the chain + decoder tables on the sub-batch's side stream
            if (count[s] == 0) return 0;
            if (!step_noise && j % NOISE_STEPS == 0)
                AFM_TRY(afm_randn_steps(noise[s], count[s], per, seed, sample_index0 + start[s], first_step + j, n_steps - j < NOISE_STEPS ? n_steps - j : NOISE_STEPS, H));
            const int64_t* tj = t_all + (int64_t)j * B + start[s];
            AFM_TRY(launch_enc_point(*w, text_u + (int64_t)start[s] * He * dkv, text_cu + (int64_t)start[s] * He, tj, count[s], N, wss[s],
                                     x + (int64_t)start[s] * per, feat + (int64_t)start[s] * N * fd, H));
            if (hipEventRecord(e1[s], H) != hipSuccess || hipStreamWaitEvent(sidest[s], e1[s], 0) != hipSuccess) return (int)hipGetLastError();
            AFM_TRY(cdm_latent_chain(*w, text_q0 + (int64_t)start[s] * dq, tj, wss[s], count[s], sidest[s], true));
            AFM_TRY(launch_dec_tables(*w, count[s], wss[s], sidest[s]));
            if (hipEventRecord(e2[s], sidest[s]) != hipSuccess) return (int)hipGetLastError();
            return 0;
        };
        auto part2 = [&](int s, int j) -> int {           // dec_point (+ DDPM update, in place) on H once the sub-batch's tables are there
            if (count[s] == 0) return 0;
            if (hipStreamWaitEvent(H, e2[s], 0) != hipSuccess) return (int)hipGetLastError();
            afm_ddpm_args dd = {};
            dd.noise = step_noise ? step_noise + ((int64_t)j * B + start[s]) * per : noise[s] + (int64_t)(j % NOISE_STEPS) * count[s] * per;
            float* xs = x + (int64_t)start[s] * per;
            dd.x_next = xs;
            dd.c1 = c1_all + (int64_t)j * B + start[s]; dd.c2 = c2_all + (int64_t)j * B + start[s]; dd.sigma = sg_all + (int64_t)j * B + start[s];
            dd.seed = seed; dd.sample_index0 = sample_index0 + start[s]; dd.step = first_step + j;
            return launch_dec_point(*w, count[s], N, wss[s], xs, feat + (int64_t)start[s] * N * fd, nullptr, &dd, H, false);
        };
        for (int s = 0; s < nsub && rc == 0; ++s) rc = part1(s, 0);
        for (int j = 0; j < n_steps && rc == 0; ++j)
            for (int s = 0; s < nsub && rc == 0; ++s) {
                rc = part2(s, j);
                if (rc == 0 && j + 1 < n_steps) rc = part1(s, j + 1);
            }
        for (int s = 0; s < nsub; ++s) {
            if (e1[s]) (void)hipEventDestroy(e1[s]);
            if (e2[s]) (void)hipEventDestroy(e2[s]);
        }
    }
    for (int j = 0; j < n_steps && rc == 0 && !pipe; ++j) {
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
                if (j % NOISE_STEPS == 0) {               // the noise of the next NOISE_STEPS steps in one launch (6 us of launch per step otherwise)
                    rc = afm_randn_steps(noise[s], count[s], per, seed, sample_index0 + start[s], first_step + j, n_steps - j < NOISE_STEPS ? n_steps - j : NOISE_STEPS, mainst[s]);
                    if (rc) break;
                }
                dd.noise = noise[s] + (int64_t)(j % NOISE_STEPS) * count[s] * per;
            }
            dd.x_next = xs;                               // in place: each element is read then written by the same lane
            dd.c1 = c1_all + (int64_t)j * B + start[s]; dd.c2 = c2_all + (int64_t)j * B + start[s]; dd.sigma = sg_all + (int64_t)j * B + start[s];
            dd.seed = seed; dd.sample_index0 = sample_index0 + start[s]; dd.step = first_step + j;
            rc = cdm_forward_impl(w, fs, xs, t_all + (int64_t)j * B + start[s], text_q0 + (int64_t)start[s] * dq,
                                  text_u + (int64_t)start[s] * He * dkv, text_cu + (int64_t)start[s] * He, nullptr, &dd, count[s], N, wsp[s], wsb[s],
                                  sidest[s], mainst[s], folded, rowless, chain_side ? &side[s] : nullptr);
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
    for (int s = 0; chain_side && s < nsub; ++s) { (void)hipEventDestroy(side[s].forked); (void)hipEventDestroy(side[s].joined); }
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
