// CMDM (`trans_enc`) denoiser step and the native p_sample_loop driver.
//
// Token layout per sample (reference cmdm.py:161): [time | n_cond step-invariant tokens | L motion].
// Per step only the time token and the L motion tokens change, so the host pre-computes the
// n_cond condition tokens (text + contact groups, adapters and positional encoding applied) once
// per sampling run; the time token is a table lookup (TimestepEmbedder depends on t only).
//
// Launch sequence per step (all on one stream, no host sync):
//   prologue (time token + cond copy + key mask) -> motion_adapter GEMM (scatter into the token
//   buffer, +bias +positional rows) -> n_layers x { in_proj GEMM, flash MHA, out_proj GEMM(+bias
//   +residual), LN, FFN1 GEMM(+bias+GELU), FFN2 GEMM(+bias+residual), LN } -> motion_layer GEMM
//   (gather motion tokens, +bias, fused DDPM posterior update).
#include "common.h"

extern "C" int afm_linear(const afm_linear_args*, void*);
extern "C" int afm_linear_pair(const afm_linear_args*, const afm_linear_args*, void*);
extern "C" int afm_mha_fwd_grouped(const float*, const uint8_t*, float*, int32_t, int32_t, int32_t, int32_t, int32_t, void*);
extern "C" int afm_mha_fwd_rows(const float*, const uint8_t*, float*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, void*);
extern "C" int afm_mha_fwd_arith(const float*, const uint8_t*, float*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, void*);
extern "C" int afm_layernorm(const float*, const float*, const float*, float*, int64_t, int32_t, float, void*);
extern "C" int afm_layernorm_rows(const float*, const float*, const float*, float*, int64_t, int32_t, float, int32_t, int32_t, int32_t, void*);
extern "C" int afm_randn(float*, int32_t, int64_t, uint64_t, int64_t, int32_t, void*);
__attribute__((visibility("hidden"))) int afm_randn_steps(float*, int32_t, int64_t, uint64_t, int64_t, int32_t, int32_t, void*);      // elementwise.hip: [nsteps][B][per_sample]

namespace {

inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

constexpr int NOISE_STEPS = 16;
struct Workspace {
    float *seq0, *y, *x1, *tmp, *qkv, *qkv0, *att, *hid, *noise, *xpad;
    uint8_t* keymask;
    uint32_t* lncnt;          // tickets of the fused LayerNorm (one word per 32 output rows; zero between launches)
    float *stat1, *stat2;     // folded LayerNorm: (mean, M2) per row and 64-column group of the raw out_proj / linear2 outputs
    int64_t bytes;
};

Workspace carve(const afm_cmdm_weights& w, int B, int L, void* base, int noise_steps = 1) {
    const int64_t T = 1 + w.n_cond + L, M = (int64_t)B * T, d = w.d;
    char* p = (char*)base;
    int64_t off = 0;
    auto take = [&](int64_t nbytes) { char* r = p ? p + off : nullptr; off += align256(nbytes); return r; };
    Workspace ws;
    ws.seq0 = (float*)take(M * d * 4);
    ws.y = (float*)take(M * d * 4);
    ws.x1 = (float*)take(M * d * 4);
    ws.tmp = (float*)take(M * d * 4);
    ws.qkv = (float*)take(M * 3 * d * 4);
    ws.qkv0 = (float*)take(M * 3 * d * 4);            // layer 0's in_proj output: its condition-token rows persist across the steps of a loop
    ws.att = (float*)take(M * d * 4);
    ws.hid = (float*)take(M * (int64_t)w.ff * 4);
    ws.noise = (float*)take((int64_t)noise_steps * B * L * w.motion_dim * 4);      // the native loop draws the Philox noise of NOISE_STEPS steps per launch (single-step forward: 1)
    ws.keymask = (uint8_t*)take(M);
    ws.lncnt = (uint32_t*)take(((M + 31) / 32) * 4);
    ws.stat1 = (float*)take(M * (d / 64 + 1) * 2 * 4); ws.stat2 = (float*)take(M * (d / 64 + 1) * 2 * 4);
    ws.xpad = w.motion_adapter_kpad > 0 ? (float*)take((int64_t)B * L * w.motion_adapter_kpad * 4) : nullptr;      // x_t with rows padded to the GEMM's K
    ws.bytes = off;
    return ws;
}

// grid (B, 1 + n_cond): token 0 = time_table[t] + pos[0]; tokens 1..n_cond = cond copy; also key mask.  With xpad != NULL the
// blocks of a sample also copy its x_t rows into rows of kpad floats (zero padded): the motion adapter's K = 263 is not a multiple
// of 16, its padded copy is (the A operand of the bf16-split GEMM needs 16-byte rows and whole K16 steps).
__global__ __launch_bounds__(128) void prologue_kernel(float* __restrict__ seq0, const float* __restrict__ time_table,
                                                       const float* __restrict__ pos_table, const int64_t* __restrict__ t,
                                                       const float* __restrict__ cond, const uint8_t* __restrict__ frame_mask,
                                                       uint8_t* __restrict__ keymask, int T, int L, int n_cond, int d,
                                                       int n_timesteps, int copy_cond, const float* __restrict__ x_t,
                                                       float* __restrict__ xpad, int md, int kpad) {
    const int b = blockIdx.x, tok = blockIdx.y;
    if (xpad) {
        for (int l = tok; l < L; l += gridDim.y) {
            const float* src = x_t + ((int64_t)b * L + l) * md;
            float* dstp = xpad + ((int64_t)b * L + l) * kpad;
            for (int c = threadIdx.x; c < kpad; c += blockDim.x) dstp[c] = c < md ? src[c] : 0.0f;
        }
    }
    float* dst = seq0 + ((int64_t)b * T + tok) * d;
    if (tok == 0) {
        int64_t ti = t[b];
        ti = ti < 0 ? 0 : (ti >= n_timesteps ? n_timesteps - 1 : ti);
        const float* src = time_table + ti * d;
        for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4) {
            const float4 a = *reinterpret_cast<const float4*>(src + c);
            const float4 p = *reinterpret_cast<const float4*>(pos_table + c);
            *reinterpret_cast<float4*>(dst + c) = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
        }
        if (keymask)
            for (int i = threadIdx.x; i < T; i += blockDim.x)
                keymask[(int64_t)b * T + i] = (i < 1 + n_cond) ? 0 : frame_mask[(int64_t)b * L + (i - 1 - n_cond)];
    } else if (copy_cond) {
        const float* src = cond + ((int64_t)b * n_cond + (tok - 1)) * d;
        for (int c = threadIdx.x * 4; c < d; c += blockDim.x * 4)
            *reinterpret_cast<float4*>(dst + c) = *reinterpret_cast<const float4*>(src + c);
    }
}

// per-step per-sample schedule rows for the whole loop: row j <-> spaced timestep i = n_steps-1-j
__global__ void expand_schedule_kernel(const int64_t* __restrict__ tmap, const float* __restrict__ c1,
                                       const float* __restrict__ c2, const float* __restrict__ sg, int n_steps, int B,
                                       int64_t* __restrict__ t_all, float* __restrict__ c1_all, float* __restrict__ c2_all,
                                       float* __restrict__ sg_all) {
    const int64_t n = (int64_t)n_steps * B;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = n_steps - 1 - (int)(e / B);
        t_all[e] = tmap[i]; c1_all[e] = c1[i]; c2_all[e] = c2[i]; sg_all[e] = sg[i];
    }
}

#define AFM_TRY(expr) do { int rc__ = (expr); if (rc__ != 0) return rc__; } while (0)

// ---- the launches of one step as DATA (round 6, AFM_CMDM_PAIR_LAUNCH): with a Recorder, forward_impl does not launch - it lists the step's
// launches in order, tagged, and the paired schedule (issue_paired below) interleaves the lists of the two sub-batches on their streams and
// fuses sub-batch A's out_proj with sub-batch B's linear1 into ONE afm_linear_pair launch per layer.
enum { OP_LINEAR = 0, OP_MHA = 1 };
enum { TAG_NONE = 0, TAG_OUT_PROJ = 1, TAG_LINEAR1 = 2 };
struct Op {
    int kind, tag;
    afm_linear_args a;
    const float* qkv; const uint8_t* keymask; float* out; int B, T, H, dh, q_first, group_waves, arith;
};
struct Recorder { Op ops[8 * AFM_MAX_LAYERS + 8]; int n = 0; };

inline int launch_op(const Op& o, hipStream_t s) {
    if (o.kind == OP_LINEAR) return afm_linear(&o.a, s);
    return afm_mha_fwd_arith(o.qkv, o.keymask, o.out, o.B, o.T, o.H, o.dh, o.q_first, o.group_waves, o.arith, s);
}

// every nn.Linear of the denoiser runs with the arithmetic the caller put into the weight pack (ABI v3: no process-wide switch)
inline int run_linear(const afm_cmdm_weights& w, afm_linear_args& a, hipStream_t s, Recorder* rec = nullptr, int tag = TAG_NONE) {
    a.arith = w.gemm_arith; a.arith_min_n = w.gemm_arith_min_n;
    const int tile = (w.flags >> AFM_CMDM_WIDE_TILE_SHIFT) & 0xF;          // measurement knob (tile shapes of one arithmetic are bit-identical)
    if (tile && a.N >= 512 && a.M >= 2048) a.tune = tile << AFM_TUNE_TILE_SHIFT;
    if (rec) {
        if (rec->n >= (int)(sizeof(rec->ops) / sizeof(rec->ops[0]))) return AFM_E_UNSUPPORTED;
        Op& o = rec->ops[rec->n++];
        o = Op{};
        o.kind = OP_LINEAR; o.tag = tag; o.a = a;
        return 0;
    }
    return afm_linear(&a, s);
}

inline int run_mha(const afm_cmdm_weights& w, const float* qkv, const uint8_t* keymask, float* out, int B, int T, int q_first, hipStream_t s, Recorder* rec) {
    if (rec) {
        if (rec->n >= (int)(sizeof(rec->ops) / sizeof(rec->ops[0]))) return AFM_E_UNSUPPORTED;
        Op& o = rec->ops[rec->n++];
        o = Op{};
        o.kind = OP_MHA; o.qkv = qkv; o.keymask = keymask; o.out = out; o.B = B; o.T = T; o.H = w.heads; o.dh = w.d / w.heads; o.q_first = q_first;
        o.group_waves = w.attn_group_waves; o.arith = w.gemm_arith;
        return 0;
    }
    return afm_mha_fwd_arith(qkv, keymask, out, B, T, w.heads, w.d / w.heads, q_first, w.attn_group_waves, w.gemm_arith, s);
}

int forward_impl(const afm_cmdm_weights& w, const float* x_t, const int64_t* t, const float* cond,
                 const uint8_t* frame_mask, float* x0_out, const afm_ddpm_args* ddpm, int B, int L, const Workspace& ws,
                 bool copy_cond, hipStream_t s, Recorder* rec = nullptr) {
    const int d = w.d, T = 1 + w.n_cond + L;
    const int M = B * T;
    uint8_t* keymask = frame_mask ? ws.keymask : nullptr;

    // Steps after the first of a native loop need no prologue launch: the condition tokens and the key mask persist in the workspace, the
    // K-padded copy of x_t was written by the previous step's DDPM update (ddpm_out2), and the time tokens ride on the motion adapter's
    // launch (aux_*) - one launch less per step (~8 us of a 430 us step at one sample per GPU).
    const bool riders = !copy_cond && ddpm && ws.xpad && !(w.flags & AFM_CMDM_NO_RIDERS) && w.gemm_arith != AFM_ARITH_F32 &&
                        (w.gemm_arith == AFM_ARITH_DEFAULT || w.gemm_arith_min_n <= d) && B <= ((B * L + 127) / 128) * ((d + 127) / 128);
    if (rec && !riders) return AFM_E_UNSUPPORTED;       // (a recorded step has no prologue launch: the caller records riders-steps only)
    if (!riders) {
        hipLaunchKernelGGL(prologue_kernel, dim3(B, 1 + w.n_cond), dim3(128), 0, s, ws.seq0, w.time_table, w.pos_table, t, cond,
                           frame_mask, keymask, T, L, w.n_cond, d, w.n_timesteps, copy_cond ? 1 : 0, x_t, ws.xpad, w.motion_dim, w.motion_adapter_kpad);
        AFM_CHECK_LAUNCH();
    }

    {   // motion_adapter (cmdm.py:159) scattered to token rows 1+n_cond.., + positional encoding (cmdm.py:162)
        afm_linear_args a = {};
        const int kp = w.motion_adapter_kpad;           // > 0: motion_adapter_w is [d, kp] (zero-padded columns) and the A rows are ws.xpad
        a.A = kp ? ws.xpad : x_t; a.lda = kp ? kp : w.motion_dim; a.W = w.motion_adapter_w; a.ldw = a.lda;
        a.C = ws.seq0; a.ldc = d; a.M = B * L; a.N = d; a.K = (int)a.lda;
        a.bias = w.motion_adapter_b;
        a.rowtab = w.pos_table + (int64_t)(1 + w.n_cond) * d; a.rowtab_period = L;
        a.c_grp = L; a.c_stride = T; a.c_off = 1 + w.n_cond;
        if (riders) {       // time token of every sample: seq0[b T] = time_table[t_b] + pos[0]
            a.aux_src = w.time_table; a.aux_idx = t; a.aux_idx_max = w.n_timesteps; a.aux_add = w.pos_table; a.aux_dst = ws.seq0;
            a.aux_dst_ld = (int64_t)T * d; a.aux_rows = B; a.aux_cols = d;
        }
        AFM_TRY(run_linear(w, a, s, rec));
    }

    // LayerNorm folded across the kernel boundaries (round 3): with the folded tensors in the pack (eval mode) and every GEMM of the layer on
    // the bf16-split kernels, norm1 / norm2 are never launched and their outputs never exist: out_proj / linear2 store the RAW residual
    // sums plus (mean, M2) per row and 64-column group, linear1 / the next in_proj / motion_layer run on the raw rows with gamma folded
    // into their weights and apply (mean, rstd) in the epilogue, the residual adds normalise their (raw) input on the fly
    // (afm_linear_args.stat_out / a_stat / res_stat).  10 launches and ~170 MB of traffic less per step at B = 32.
    // Every folded GEMM must run on the bf16-split kernels (the native ones do not carry the row statistics): K = d or ff >= 128 and a
    // multiple of 16, N >= the arithmetic's minimum width for all five shapes (N = 3d, d, ff, d, motion_dim) - otherwise the layers fall
    // back to the separate LayerNorm launches instead of failing with AFM_E_UNSUPPORTED.
    const int split_min_n = w.gemm_arith == AFM_ARITH_DEFAULT ? 32 : w.gemm_arith_min_n;
    bool fold = !(w.flags & (AFM_CMDM_NO_LN_FOLD | AFM_CMDM_FUSED_LN)) && w.motion_layer_wg && w.motion_layer_g && w.motion_layer_c && (d % 64) == 0 &&
                w.gemm_arith != AFM_ARITH_F32 && d >= 128 && w.ff >= 128 && (w.ff % 16) == 0 && d >= split_min_n && w.ff >= split_min_n &&
                w.motion_dim >= split_min_n;
    for (int li = 0; li < w.n_layers && fold; ++li)
        fold = w.layer[li].lin1_wg && w.layer[li].lin1_g && w.layer[li].lin1_c && (li == 0 || (w.layer[li].in_proj_wg && w.layer[li].in_proj_g && w.layer[li].in_proj_c));
    const int sg = d / 64;                            // statistic groups per row
    const float* X = ws.seq0;
    for (int li = 0; li < w.n_layers; ++li) {
        const afm_encoder_layer_weights& lw = w.layer[li];
        // Layer 0 reads the token buffer itself, whose condition rows (text + contact groups) do not change between the steps of a
        // sampling loop: their q | k | v rows are computed on the loop's first step only (copy_cond) and kept in qkv0; afterwards the
        // in_proj runs on the time-token rows and the motion rows (a GEMM row depends on its own input row only: bit-identical).
        float* qkv = li == 0 ? ws.qkv0 : ws.qkv;
        afm_linear_args a = {};
        a.A = X; a.lda = d; a.W = lw.in_proj_w; a.ldw = d; a.C = qkv; a.ldc = 3 * d;
        a.N = 3 * d; a.K = d; a.bias = lw.in_proj_b;
        if (fold && li > 0) {                         // X = the previous layer's raw linear2 output: its norm2 is folded into W / bias
            a.W = lw.in_proj_wg; a.bias = lw.in_proj_c; a.a_stat = ws.stat2; a.a_stat_groups = sg; a.a_fold_g = lw.in_proj_g; a.ln_eps2 = 1e-5f;
        }
        const bool no_l0_cache = (w.flags & AFM_CMDM_NO_L0_CACHE) != 0;                  // measurement knob
        if (li == 0 && !copy_cond && w.n_cond > 0 && !no_l0_cache) {
            // ONE launch over the time token and the L motion tokens of every sample: groups of 1 + L rows with a hole of n_cond rows
            // behind the first (afm_linear_args.a_skip; round 3 ran two launches - a launch is ~17 us of a small-batch step)
            a.M = B * (1 + L);
            a.a_grp = 1 + L; a.a_stride = T; a.a_off = 0; a.a_skip_after = 1; a.a_skip = w.n_cond;
            a.c_grp = 1 + L; a.c_stride = T; a.c_off = 0; a.c_skip_after = 1; a.c_skip = w.n_cond;
            AFM_TRY(run_linear(w, a, s, rec));
        } else {
            a.M = M;
            AFM_TRY(run_linear(w, a, s, rec));
        }
        // After the LAST layer only the L motion tokens of each sample are read (motion_layer, cmdm.py:169,195), and
        // everything after the attention's key / value side is row-local: the attention computes the motion tokens' QUERY rows only
        // (7 query blocks instead of 11 at T = 326) and the stages behind it run on the B*L motion rows only (token rows gathered /
        // scattered by the row maps; the other rows of att/tmp/x1/y keep stale values nobody reads).
        const bool last = (li == w.n_layers - 1);
        if (last && w.n_cond > 0 && !(w.flags & AFM_CMDM_ALL_QUERIES))
            AFM_TRY(run_mha(w, qkv, keymask, ws.att, B, T, 1 + w.n_cond, s, rec));
        else
            AFM_TRY(run_mha(w, qkv, keymask, ws.att, B, T, 0, s, rec));
        const int rows = last ? B * L : M;
        const int g = last ? L : 0, gs = last ? T : 0, go = last ? 1 + w.n_cond : 0;
        a = {};
        a.A = ws.att; a.lda = d; a.W = lw.out_proj_w; a.ldw = d; a.C = ws.tmp; a.ldc = d;
        a.M = rows; a.N = d; a.K = d; a.bias = lw.out_proj_b; a.residual = X; a.ldr = d;
        a.a_grp = g; a.a_stride = gs; a.a_off = go; a.c_grp = g; a.c_stride = gs; a.c_off = go;
        // norm1 / norm2 CAN run inside the GEMM that produces their input (afm_linear_args.ln_*: the workgroup finishing the last column
        // tile of a block of rows normalises it; bit-identical).  Measured SLOWER than the separate launch on MI355X (B = 32: +20 us per
        // GEMM launch against 10.7 us per LayerNorm launch; B = 4: 0.679 vs 0.628 ms/step; profiles/r03_ln_fusion.md): every tile's
        // workgroup has to drain its write-through stores and wait for its ticket before it can retire.  Opt-in: AFM_CMDM_FUSED_LN.
        const bool fuse_ln = (w.flags & AFM_CMDM_FUSED_LN) != 0;
        if (fold) {
            a.stat_out = ws.stat1; a.ln_eps2 = 1e-5f;
            if (li > 0) { a.res_stat = ws.stat2; a.res_gamma = w.layer[li - 1].norm2_w; a.res_beta = w.layer[li - 1].norm2_b; }     // residual = LayerNorm(raw X)
            AFM_TRY(run_linear(w, a, s, rec, TAG_OUT_PROJ));
            a = {};                                   // linear1 on the raw rows, norm1 folded
            a.A = ws.tmp; a.lda = d; a.W = lw.lin1_wg; a.ldw = d; a.C = ws.hid; a.ldc = w.ff;
            a.M = rows; a.N = w.ff; a.K = d; a.bias = lw.lin1_c; a.act = AFM_ACT_GELU;
            a.a_stat = ws.stat1; a.a_stat_groups = sg; a.a_fold_g = lw.lin1_g; a.ln_eps2 = 1e-5f;
            a.a_grp = g; a.a_stride = gs; a.a_off = go;
            AFM_TRY(run_linear(w, a, s, rec, TAG_LINEAR1));
            a = {};                                   // linear2 + LayerNorm1(raw) as the residual -> raw output + its statistics
            a.A = ws.hid; a.lda = w.ff; a.W = lw.lin2_w; a.ldw = w.ff; a.C = ws.y; a.ldc = d;
            a.M = rows; a.N = d; a.K = w.ff; a.bias = lw.lin2_b; a.residual = ws.tmp; a.ldr = d;
            a.res_stat = ws.stat1; a.res_gamma = lw.norm1_w; a.res_beta = lw.norm1_b; a.stat_out = ws.stat2; a.ln_eps2 = 1e-5f;
            a.c_grp = g; a.c_stride = gs; a.c_off = go;
            AFM_TRY(run_linear(w, a, s, rec));
            X = ws.y;
            continue;
        }
        if (rec) return AFM_E_UNSUPPORTED;             // (the paired schedule exists for the folded-LayerNorm step only)
        if (fuse_ln) { a.ln_gamma = lw.norm1_w; a.ln_beta = lw.norm1_b; a.ln_out = ws.x1; a.ldo = d; a.ln_eps = 1e-5f; a.ln_counters = ws.lncnt; }
        AFM_TRY(run_linear(w, a, s));
        if (!fuse_ln) AFM_TRY(afm_layernorm_rows(ws.tmp, lw.norm1_w, lw.norm1_b, ws.x1, rows, d, 1e-5f, g, gs, go, s));
        a = {};
        a.A = ws.x1; a.lda = d; a.W = lw.lin1_w; a.ldw = d; a.C = ws.hid; a.ldc = w.ff;
        a.M = rows; a.N = w.ff; a.K = d; a.bias = lw.lin1_b; a.act = AFM_ACT_GELU;
        a.a_grp = g; a.a_stride = gs; a.a_off = go;                       // hid is written compactly [rows, ff]
        AFM_TRY(run_linear(w, a, s));
        a = {};
        a.A = ws.hid; a.lda = w.ff; a.W = lw.lin2_w; a.ldw = w.ff; a.C = ws.tmp; a.ldc = d;
        a.M = rows; a.N = d; a.K = w.ff; a.bias = lw.lin2_b; a.residual = ws.x1; a.ldr = d;
        a.c_grp = g; a.c_stride = gs; a.c_off = go;
        if (fuse_ln) { a.ln_gamma = lw.norm2_w; a.ln_beta = lw.norm2_b; a.ln_out = ws.y; a.ldo = d; a.ln_eps = 1e-5f; a.ln_counters = ws.lncnt; }
        AFM_TRY(run_linear(w, a, s));
        if (!fuse_ln) AFM_TRY(afm_layernorm_rows(ws.tmp, lw.norm2_w, lw.norm2_b, ws.y, rows, d, 1e-5f, g, gs, go, s));
        X = ws.y;
    }

    {   // motion_layer (cmdm.py:195) on the motion tokens only (cmdm.py:169), optional DDPM update
        afm_linear_args a = {};
        a.A = X; a.lda = d; a.W = w.motion_layer_w; a.ldw = d;
        a.C = x0_out; a.ldc = w.motion_dim; a.M = B * L; a.N = w.motion_dim; a.K = d;
        a.bias = w.motion_layer_b;
        if (fold) {                                   // X = the last layer's raw linear2 output
            a.W = w.motion_layer_wg; a.bias = w.motion_layer_c; a.a_stat = ws.stat2; a.a_stat_groups = sg; a.a_fold_g = w.motion_layer_g; a.ln_eps2 = 1e-5f;
        }
        a.a_grp = L; a.a_stride = T; a.a_off = 1 + w.n_cond;
        if (ddpm) {
            const float* nz = ddpm->noise;
            if (rec && !nz) return AFM_E_UNSUPPORTED;      // (recorded steps get their noise from the loop)
            if (!nz) {
                AFM_TRY(afm_randn(ws.noise, B, (int64_t)L * w.motion_dim, ddpm->seed, ddpm->sample_index0, ddpm->step, s));
                nz = ws.noise;
            }
            a.ddpm_xt = x_t; a.ddpm_noise = nz; a.ddpm_out = ddpm->x_next; a.ldx = w.motion_dim;
            a.ddpm_c1 = ddpm->c1; a.ddpm_c2 = ddpm->c2; a.ddpm_sigma = ddpm->sigma; a.rows_per_sample = L;
            a.ddpm_clip = (w.flags & AFM_CMDM_CLIP_X0) ? 1 : 0;
            if (ws.xpad) { a.ddpm_out2 = ws.xpad; a.ldx2 = w.motion_adapter_kpad; }      // x_next also as the NEXT step's K-padded A rows (the padding columns stay zero)
        }
        AFM_TRY(run_linear(w, a, s, rec));
    }
    return 0;
}

int validate(const afm_cmdm_weights* w, int B, int L) {
    if (!w || B < 0 || L <= 0) return AFM_E_BADARG;
    if (w->d <= 0 || (w->d & 3) || w->heads <= 0 || w->d % w->heads || w->ff <= 0 || (w->ff & 3)) return AFM_E_BADARG;
    if (w->n_layers <= 0 || w->n_layers > AFM_MAX_LAYERS || w->n_cond < 0 || w->motion_dim <= 0) return AFM_E_BADARG;
    if (w->d / w->heads != 64) return AFM_E_UNSUPPORTED;
    if (w->gemm_arith != AFM_ARITH_DEFAULT && w->gemm_arith != AFM_ARITH_F32 && w->gemm_arith != AFM_ARITH_BF16X6 && w->gemm_arith != AFM_ARITH_BF16X9 &&
        w->gemm_arith != AFM_ARITH_BF16X1) return AFM_E_BADARG;
    if (w->gemm_arith_min_n < 0 || w->attn_group_waves < 0) return AFM_E_BADARG;
    if (w->motion_adapter_kpad != 0 && (w->motion_adapter_kpad < w->motion_dim || (w->motion_adapter_kpad & 3))) return AFM_E_BADARG;
    if (!w->motion_adapter_w || !w->motion_layer_w || !w->time_table || !w->pos_table) return AFM_E_BADARG;
    return 0;
}

}  // namespace

extern "C" int afm_version(void) { return AFM_ABI_VERSION; }

extern "C" int64_t afm_cmdm_workspace_bytes(const afm_cmdm_weights* w, int32_t B, int32_t L) {
    if (validate(w, B, L) != 0) return AFM_E_BADARG;
    return carve(*w, B, L, nullptr).bytes;
}

extern "C" int afm_cmdm_forward(const afm_cmdm_weights* w, const float* x_t, const int64_t* t, const float* cond_tokens,
                                const uint8_t* frame_mask, float* x0_out, const afm_ddpm_args* ddpm, int32_t B, int32_t L,
                                void* workspace, int64_t workspace_bytes, void* stream) {
    AFM_TRY(validate(w, B, L));
    if (!x_t || !t || (w->n_cond > 0 && !cond_tokens) || !workspace) return AFM_E_BADARG;
    if (!x0_out && !ddpm) return AFM_E_BADARG;
    if (ddpm && (!ddpm->x_next || !ddpm->c1 || !ddpm->c2 || !ddpm->sigma)) return AFM_E_BADARG;
    if (B == 0) return 0;
    const Workspace ws = carve(*w, B, L, workspace);
    if (ws.bytes > workspace_bytes) return AFM_E_WORKSPACE;
    if ((w->flags & AFM_CMDM_FUSED_LN) &&           // ticket words of the opt-in fused LayerNorm only
        hipMemsetAsync(ws.lncnt, 0, (size_t)(((int64_t)B * (1 + w->n_cond + L) + 31) / 32) * 4, (hipStream_t)stream) != hipSuccess) return (int)hipGetLastError();
    return forward_impl(*w, x_t, t, cond_tokens, frame_mask, x0_out, ddpm, B, L, ws, true, (hipStream_t)stream);
}

extern "C" int64_t afm_cmdm_sched_scratch_bytes(int32_t n_steps, int32_t B) {
    if (n_steps <= 0 || B < 0) return AFM_E_BADARG;
    return align256((int64_t)n_steps * B * 8) + 3 * align256((int64_t)n_steps * B * 4);
}

namespace {
inline void sub_range(int B, int n, int s, int* start, int* count) {
    const int base = B / n, extra = B % n;
    *count = base + (s < extra ? 1 : 0);
    *start = s * base + (s < extra ? s : extra);
}
}  // namespace

extern "C" int64_t afm_cmdm_loop_workspace_bytes(const afm_cmdm_weights* w, int32_t B, int32_t L, int32_t n_streams) {
    if (validate(w, B, L) != 0 || n_streams < 0) return AFM_E_BADARG;
    const int n = n_streams > 1 ? (n_streams < B ? n_streams : (B > 0 ? B : 1)) : 1;
    int64_t total = 0;
    for (int s = 0; s < n; ++s) {
        int st, cnt;
        sub_range(B, n, s, &st, &cnt);
        total += carve(*w, cnt, L, nullptr, NOISE_STEPS).bytes;
    }
    return total;
}

// ---- the paired schedule of one step (AFM_CMDM_PAIR_LAUNCH; two sub-batches A, B on streams sa, sb).  Both lists hold the same launch
// sequence.  Per layer: sub-batch B runs up to and including its out_proj on sb; then ONE launch on sa computes A's out_proj AND B's linear1
// (afm_linear_pair: 164 + 328 = 492 tiles of 128 x 128 at 16 samples per sub-batch - one full resident round); B continues with linear2 on
// sb, A with linear1 on sa.  Two cross-stream edges per layer: sb -> sa before the pair (B's out_proj output and statistics), sa -> sb after
// it (B's hidden rows).  Every element is computed by the same tile program on the same operands: bit-identical to the unpaired schedule.
static int issue_paired(const Recorder& A, const Recorder& B, hipStream_t sa, hipStream_t sb, hipEvent_t* ev, int nev) {
    if (A.n != B.n) return AFM_E_UNSUPPORTED;
    int ia = 0, ib = 0, e = 0;
    while (ia < A.n) {
        const bool pair = A.ops[ia].tag == TAG_OUT_PROJ && ia + 1 < B.n && B.ops[ia + 1].tag == TAG_LINEAR1 && ib <= ia + 1 && e + 2 <= nev;
        if (pair) {
            while (ib <= ia) AFM_TRY(launch_op(B.ops[ib++], sb));                     // B up to and including its out_proj
            if (hipEventRecord(ev[e], sb) != hipSuccess || hipStreamWaitEvent(sa, ev[e], 0) != hipSuccess) return (int)hipGetLastError();
            const int rc = afm_linear_pair(&A.ops[ia].a, &B.ops[ia + 1].a, sa);
            if (rc == AFM_E_UNSUPPORTED) {                                             // shapes the paired form does not take: two plain launches
                AFM_TRY(launch_op(A.ops[ia], sa));
                AFM_TRY(launch_op(B.ops[ia + 1], sa));
            } else if (rc != 0) return rc;
            if (hipEventRecord(ev[e + 1], sa) != hipSuccess || hipStreamWaitEvent(sb, ev[e + 1], 0) != hipSuccess) return (int)hipGetLastError();
            e += 2;
            ia += 1; ib = ia + 1;                                                      // B's linear1 is done
        } else {
            AFM_TRY(launch_op(A.ops[ia++], sa));
        }
    }
    while (ib < B.n) AFM_TRY(launch_op(B.ops[ib++], sb));
    return 0;
}

static int sample_loop_impl(const afm_cmdm_weights* w, float* x, const float* cond_tokens, const uint8_t* frame_mask,
                            const float* step_noise, const int64_t* d_timestep_map, const float* d_c1,
                            const float* d_c2, const float* d_sigma, int32_t n_steps, int32_t first_step, uint64_t seed,
                            int64_t sample_index0, int32_t B, int32_t L, void* sched_scratch, void* workspace,
                            int64_t workspace_bytes, int32_t n_streams, void* const* side_streams, void* stream) {
    AFM_TRY(validate(w, B, L));
    if (!x || (w->n_cond > 0 && !cond_tokens) || !d_timestep_map || !d_c1 || !d_c2 || !d_sigma || n_steps <= 0 ||
        !sched_scratch || !workspace || n_streams < 0 || (n_streams > 1 && !side_streams))
        return AFM_E_BADARG;
    if (B == 0) return 0;
    hipStream_t s0 = (hipStream_t)stream;
    int nsub = n_streams > 1 ? (n_streams < B ? n_streams : B) : 1;
    if (nsub > 16) nsub = 16;

    char* sp = (char*)sched_scratch;
    const int64_t nb = (int64_t)n_steps * B;
    int64_t* t_all = (int64_t*)sp; sp += align256(nb * 8);
    float* c1_all = (float*)sp; sp += align256(nb * 4);
    float* c2_all = (float*)sp; sp += align256(nb * 4);
    float* sg_all = (float*)sp;
    hipLaunchKernelGGL(expand_schedule_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, s0, d_timestep_map, d_c1, d_c2,
                       d_sigma, n_steps, B, t_all, c1_all, c2_all, sg_all);
    AFM_CHECK_LAUNCH();

    // carve one workspace per sub-batch
    Workspace ws[16];
    int start[16], count[16];
    hipStream_t st[16];
    {
        char* base = (char*)workspace;
        int64_t off = 0;
        for (int s = 0; s < nsub; ++s) {
            sub_range(B, nsub, s, &start[s], &count[s]);
            ws[s] = carve(*w, count[s], L, base + off, NOISE_STEPS);
            off += ws[s].bytes;
            st[s] = nsub > 1 ? (hipStream_t)side_streams[s] : s0;
        }
        if (off > workspace_bytes) return AFM_E_WORKSPACE;
    }
    hipEvent_t fork = nullptr;
    if (nsub > 1) {      // side streams start after everything already queued on `stream` (inputs, schedule rows)
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return (int)hipGetLastError();
        (void)hipEventRecord(fork, s0);
        for (int s = 0; s < nsub; ++s) (void)hipStreamWaitEvent(st[s], fork, 0);
    }

    const int T = 1 + w->n_cond + L;
    // Tile shape of the wide encoder GEMMs inside the multi-stream loop (round 6).  afm_linear's own rule prices ONE launch: 128 x 128 tiles only
    // when their last resident round is >= 90 % full (in_proj), 64 x 64 otherwise.  Inside this loop a second sub-batch's kernels fill the
    // slots a partial round leaves, and what counts is the work per matrix instruction (the loop runs at the board's power limit): with the
    // six-product arithmetic, 128 x 128 on EVERY GEMM with N >= 512 measured 588-590 against 568-574 steps/s at 16 + 16 samples in the same
    // calls, but 853 against 995 at 8 + 8 and 1168 against 1470 at 4 + 4 (profiles/r06_tile_rule.md) - so: sub-batches of >= 4096 rows only.
    // Tile shapes of one arithmetic are bit-identical; a caller's explicit AFM_CMDM_WIDE_TILE code wins.
    afm_cmdm_weights wl = *w;
    if (nsub >= 2 && ((wl.flags >> AFM_CMDM_WIDE_TILE_SHIFT) & 0xF) == 0) {
        bool big = true;
        for (int s = 0; s < nsub; ++s) big = big && (int64_t)count[s] * T >= 4096;
        if (big) wl.flags |= 5 << AFM_CMDM_WIDE_TILE_SHIFT;
    }
    w = &wl;
    for (int s = 0; s < nsub; ++s)        // ticket words of the fused LayerNorm: zero once, every launch leaves them zero
        if (count[s] > 0 && (w->flags & AFM_CMDM_FUSED_LN) && hipMemsetAsync(ws[s].lncnt, 0, (size_t)(((int64_t)count[s] * T + 31) / 32) * 4, st[s]) != hipSuccess) return (int)hipGetLastError();
    const int64_t row = (int64_t)L * w->motion_dim;
    int rc = 0;
    // paired schedule (AFM_CMDM_PAIR_LAUNCH, two sub-batches): every step after the first is recorded per sub-batch and issued interleaved
    const bool paired = (w->flags & AFM_CMDM_PAIR_LAUNCH) && nsub == 2 && count[0] > 0 && count[1] > 0;
    constexpr int NEV = 2 * AFM_MAX_LAYERS;
    hipEvent_t pev[NEV] = {};
    Recorder* recs = nullptr;
    if (paired) {
        for (int i = 0; i < NEV; ++i)
            if (hipEventCreateWithFlags(&pev[i], hipEventDisableTiming) != hipSuccess) {
                const int rc_ev = (int)hipGetLastError();
                for (int u = 0; u < i; ++u) (void)hipEventDestroy(pev[u]);          // nothing created so far may leak
                if (fork) (void)hipEventDestroy(fork);
                return rc_ev;
            }
        recs = new Recorder[2];
    }
    for (int j = 0; j < n_steps && rc == 0; ++j) {
        const bool rec_step = paired && j > 0;
        if (rec_step) recs[0].n = recs[1].n = 0;
        for (int s = 0; s < nsub && rc == 0; ++s) {
            if (count[s] == 0) continue;
            afm_ddpm_args dd = {};
            if (step_noise) dd.noise = step_noise + ((int64_t)j * B + start[s]) * row;
            else {
                if (j % NOISE_STEPS == 0) {               // one launch per NOISE_STEPS steps instead of one per step (a launch is ~5 us of a small-batch step)
                    rc = afm_randn_steps(ws[s].noise, count[s], row, seed, sample_index0 + start[s], first_step + j, n_steps - j < NOISE_STEPS ? n_steps - j : NOISE_STEPS, st[s]);
                    if (rc) break;
                }
                dd.noise = ws[s].noise + (int64_t)(j % NOISE_STEPS) * count[s] * row;
            }
            dd.x_next = x + (int64_t)start[s] * row;      // in place: each element is read then written by the same lane
            dd.c1 = c1_all + (int64_t)j * B + start[s]; dd.c2 = c2_all + (int64_t)j * B + start[s];
            dd.sigma = sg_all + (int64_t)j * B + start[s];
            dd.seed = seed; dd.sample_index0 = sample_index0 + start[s]; dd.step = first_step + j;
            rc = forward_impl(*w, x + (int64_t)start[s] * row, t_all + (int64_t)j * B + start[s],
                              cond_tokens ? cond_tokens + (int64_t)start[s] * w->n_cond * w->d : nullptr,
                              frame_mask ? frame_mask + (int64_t)start[s] * L : nullptr, nullptr, &dd, count[s], L, ws[s], j == 0,
                              st[s], rec_step ? &recs[s] : nullptr);
        }
        if (rec_step && rc == 0) rc = issue_paired(recs[0], recs[1], st[0], st[1], pev, NEV);
    }
    if (paired) {
        delete[] recs;
        for (int i = 0; i < NEV; ++i) (void)hipEventDestroy(pev[i]);
    }
    if (nsub > 1) {      // join: `stream` continues only after every sub-batch loop has finished
        for (int s = 0; s < nsub; ++s) {
            hipEvent_t done;
            if (hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess) {
                (void)hipEventRecord(done, st[s]);
                (void)hipStreamWaitEvent(s0, done, 0);
                (void)hipEventDestroy(done);
            }
        }
        (void)hipEventDestroy(fork);
    }
    return rc;
}

extern "C" int afm_cmdm_sample_loop(const afm_cmdm_weights* w, float* x, const float* cond_tokens, const uint8_t* frame_mask,
                                    const float* step_noise, const int64_t* d_timestep_map, const float* d_c1,
                                    const float* d_c2, const float* d_sigma, int32_t n_steps, uint64_t seed,
                                    int64_t sample_index0, int32_t B, int32_t L, void* sched_scratch, void* workspace,
                                    int64_t workspace_bytes, int32_t n_streams, void* const* side_streams, void* stream) {
    return sample_loop_impl(w, x, cond_tokens, frame_mask, step_noise, d_timestep_map, d_c1, d_c2, d_sigma, n_steps, 0, seed,
                            sample_index0, B, L, sched_scratch, workspace, workspace_bytes, n_streams, side_streams, stream);
}

extern "C" int afm_cmdm_sample_loop_range(const afm_cmdm_weights* w, float* x, const float* cond_tokens,
                                          const uint8_t* frame_mask, const float* step_noise, const int64_t* d_timestep_map,
                                          const float* d_c1, const float* d_c2, const float* d_sigma, int32_t n_steps,
                                          int32_t first_step, uint64_t seed, int64_t sample_index0, int32_t B, int32_t L,
                                          void* sched_scratch, void* workspace, int64_t workspace_bytes, int32_t n_streams,
                                          void* const* side_streams, void* stream) {
    if (first_step < 0) return AFM_E_BADARG;
    return sample_loop_impl(w, x, cond_tokens, frame_mask, step_noise, d_timestep_map, d_c1, d_c2, d_sigma, n_steps, first_step,
                            seed, sample_index0, B, L, sched_scratch, workspace, workspace_bytes, n_streams, side_streams, stream);
}
