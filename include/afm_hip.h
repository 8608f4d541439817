/*
 * afm_hip.h - C-ABI of libafm_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * afford-motion diffusion denoising hot path.
 *
 * The reference (pure Python on PyTorch) has no FFI of its own; its only native dependency on
 * this path is the external CUDA extension `pointops_cuda` (reference
 * models/scene_models/pointops.py:7) and ATen's fused transformer kernels.  Every entry point
 * below names the reference interface it replaces (file:line under /root/reference).
 *
 * Conventions ("C-ABI face"):
 *   - extern "C", plain pointers and sizes; all pointers are DEVICE pointers unless named h_*.
 *   - every function returns 0 on success, a positive hipError_t, or a negative AFM_E_* code.
 *   - no allocation, no host synchronisation, no global state: work is enqueued on `stream`
 *     (a hipStream_t passed as void*); the caller owns all memory and keeps it alive until the
 *     stream is synchronised.  Thread-safe by construction.  In particular the library reads NO
 *     environment variables and has no setters: every arithmetic or tuning choice is a field of
 *     the argument structs (ABI v3; the one exception is the opt-in measurement profiler at the end).
 *   - all matrices are row-major float32; indices int32; masks uint8 (1 = padded/ignored).
 */
#ifndef AFM_HIP_H
#define AFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFM_ABI_VERSION 7

#define AFM_E_BADARG   (-1)   /* shape / pointer validation failed              */
#define AFM_E_WORKSPACE (-2)  /* workspace too small                            */
#define AFM_E_UNSUPPORTED (-3)

/* activation codes for afm_linear */
#define AFM_ACT_NONE 0
#define AFM_ACT_GELU 1        /* exact erf GELU (nn.GELU / activation='gelu')   */
#define AFM_ACT_RELU 2
#define AFM_ACT_SILU 3

int afm_version(void);

/* ------------------------------------------------------------------------------------------
 * afm_linear: C = act_post(act(scale * (A @ W^T) + bias) + residual + rowtab[row % period])
 * Replaces F.linear / nn.Linear (+ the elementwise op that follows it) wherever the reference
 * calls one on this path: cmdm.py:146,156,159,195; nn.TransformerEncoderLayer's in_proj /
 * out_proj / linear1 / linear2 (cmdm.py:66-77); modules.py:52-53,317-319,377,651-661;
 * pointtransformer.py:28,49,51,115-120 (with eval-mode BatchNorm folded into scale/bias).
 * f32 MFMA (v_mfma_f32_32x32x2_f32): exact-f32 products, f32 accumulate.
 *
 *   A [M,K] (row stride lda), W [N,K] (row stride ldw, the nn.Linear weight layout),
 *   C [M,N] (row stride ldc).  bias/scale [N] or NULL.  residual [M,N] (stride ldr) or NULL,
 *   indexed by the OUTPUT row.  rowtab [period,N] or NULL.
 *   Row remaps (0 = identity): logical row r reads A row (r / a_grp) * a_stride + a_off + r % a_grp
 *   and writes C row (r / c_grp) * c_stride + c_off + r % c_grp  (token gather / scatter of
 *   cmdm.py:161,169).
 */
typedef struct {
    const float* A; int64_t lda;
    const float* W; int64_t ldw;
    float* C; int64_t ldc;
    int32_t M, N, K;
    const float* bias;
    const float* scale;
    const float* residual; int64_t ldr;
    const float* rowtab; int32_t rowtab_period;
    int32_t act;                    /* AFM_ACT_* applied before the residual               */
    int32_t act_post;               /* AFM_ACT_* applied after residual / rowtab (ReLU(bn3(.) + identity), pointtransformer.py:120-122) */
    int32_t a_grp, a_stride, a_off;
    int32_t c_grp, c_stride, c_off;
    /* optional fused DDPM update on the output (gaussian_diffusion.py:209-231,431-439):
     * x_next[r,n] = c1[r / rows_per_sample] * C[r,n] + c2[..] * x_t[r,n] + sigma[..] * noise[r,n]
     * (all [M,N] with row stride ldx).  C itself (pred_xstart) is still written if C != NULL. */
    const float* ddpm_xt; const float* ddpm_noise; float* ddpm_out; int64_t ldx;
    const float* ddpm_c1; const float* ddpm_c2; const float* ddpm_sigma; int32_t rows_per_sample;
    /* ---- training hooks (all optional, zero = off; ABI v2).  Order inside the epilogue:
     *   v = scale*acc + bias; preact <- v; v = act(v); v = dropout(v) [drop_after == 0]; v *= act'(dact_z) [dact];
     *   v += residual + rowtab; v = act_post(v); v = dropout(v) [drop_after == 1]; C <- v
     * preact [M,N] (stride ldp): the pre-activation saved for the backward pass (linear1 of the encoder layer,
     *   time_embed.0).  dact / dact_z [M,N] (stride ldz): multiply by the derivative of AFM_ACT_* at the saved
     *   pre-activation - the input-gradient GEMM of the NEXT linear produces dz directly (GELU/SiLU backward fused).
     * dropout: keep-mask from a counter hash of (drop_seed, drop_id, output_row * N + col), scaled by 1/(1-p)
     *   (nn.Dropout of the encoder layer / PositionalEncoding, modules.py:43-45; regenerated, never stored). */
    float* preact; int64_t ldp;
    const float* dact_z; int64_t ldz; int32_t dact;
    float drop_p; uint64_t drop_seed; uint32_t drop_id; int32_t drop_after;
    /* ---- arithmetic of the product (ABI v3; replaces the process-wide afm_linear_set_split of v2).  A GEMM is ELIGIBLE for the
     * bf16-split path when K >= 128, K % 16 == 0 and A / W are 16-byte aligned with lda / ldw % 4 == 0; everything else always runs
     * the native f32 MFMA kernels (v_mfma_f32_32x32x2_f32, 157 TF peak on gfx950).
     *   AFM_ARITH_DEFAULT  every eligible GEMM with N >= 32 takes AFM_ARITH_BF16X6 (ABI v7; v3-v6: X9), all others AFM_ARITH_F32 (arith_min_n
     *                      ignored).  Decided by a worst-case measurement, not an rms: over adversarial operand families (catastrophic
     *                      cancellation, 2^+-60 dynamic range inside a row, mantissas that maximise the dropped terms with every product of one
     *                      sign, K = 128 ... 4096) the worst output element of X6 is never above X9's and both sit in the error class of the
     *                      native f32 MFMA kernel and of an unfused f32 multiply-add chain (tests/test_gpu_arith.py, profiles/r06_arith_worstcase.json).
     *                      Domain of both split forms: operands of magnitude >= 2^-110 (or zero) - a split term below 2^-126 is a bf16
     *                      subnormal and is flushed; AFM_ARITH_F32 has no such limit;
     *   AFM_ARITH_F32      native f32 MFMA;
     *   AFM_ARITH_BF16X9   eligible GEMMs with N >= arith_min_n: every f32 operand is split exactly into three bf16 terms inside the
     *                      kernel and all nine cross products run on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 2500 TF peak)
     *                      with f32 accumulation: the same exact products as the f32 MFMA in a different summation order
     *                      (measured error vs float64 <= the native kernel's, tools/kernel_sweep.cpp);
     *   AFM_ARITH_BF16X6   as X9 without the three smallest products a2 w3, a3 w2, a3 w3 (together <= 2^-23 |a||w|, i.e. two f32 roundings of
     *                      that product; what oneMKL calls float_to_bf16x3 and XLA's "highest" precision on bf16 matrix units).
     *   AFM_ARITH_BF16X1   INFORMATIONAL, not f32 arithmetic: only the product of the two leading bf16 terms (what a plain bf16 x bf16 GEMM with f32
     *                      accumulation computes; relative error ~2^-9 per product).  Exists to MEASURE what bf16 would cost in accuracy
     *                      (tests/test_gpu_cmdm.py::test_bf16_one_product_drift...); never selected by the library.
     * The kernel choice is a function of (arith, arith_min_n, N, K, alignment) - never of M - and every tile shape of one
     * arithmetic sums a given output element in the same order, so a batch and its shards compute identical bits. */
    int32_t arith; int32_t arith_min_n;
    /* performance-only knobs for experiments (bit-neutral; 0 = the library's heuristics): AFM_TUNE_* bits */
    int32_t tune;
    /* ---- fused row-dot epilogue (ABI v4; NULL = off): for a linear layer that is followed by a NARROW linear one (the CDM's
     * contact_layer, contact_dim <= 8 outputs) the wide output never has to exist: with R = rowdot_n vectors rowdot_w [R, N] the kernel
     * writes rowdot_out[row, g, r] = sum over the 64-column group g of C[row, col] * rowdot_w[r, col]  ([M, ceil(N / 64), R], rows
     * through the c_* remap) in a fixed summation order that does not depend on the tile shape; the consumer adds the groups left to
     * right.  C may be NULL.  Needs N % 4 == 0 and 16-byte aligned side inputs. */
    const float* rowdot_w; float* rowdot_out; int32_t rowdot_n;
    /* ---- fused LayerNorm of the output rows (ABI v5; ln_out NULL = off): out_proj / linear2 of a post-LN encoder layer are followed by
     * nn.LayerNorm over the whole output row (cmdm.py:66-77), which no column tile owns.  With ln_out set, the workgroup that finishes the
     * LAST column tile of a block of output rows ("last arriver": one agent-scope release per tile, one ticket per tile on ln_counters, one
     * agent-scope acquire by the last) reads those rows of C back and writes ln_out[row] = LayerNorm(C[row]) * ln_gamma + ln_beta with
     * the arithmetic of afm_layernorm (bit-identical to the two-launch form).  ln_out rows follow the c_* remap, row stride ldo.
     * ln_counters: >= ceil(M / 32) device words, ZERO on entry, zero again on exit.  Needs C, N % 4 == 0, N <= 1024, 16-byte rows; ln_out
     * must not alias C, A or residual. */
    const float* ln_gamma; const float* ln_beta; float* ln_out; int64_t ldo; float ln_eps; uint32_t* ln_counters;
    /* ---- LayerNorm folded ACROSS kernel boundaries (ABI v5; all NULL = off).  In a post-LN encoder layer the LayerNorm output is only
     * ever (i) the A operand of the next linear layer and (ii) a residual input, so it need not exist:
     *   stat_out   [C rows][N / 64][2]  the producer writes, per stored output row and 64-column group, (mean, M2 = sum of squared
     *              deviations from that mean).  Needs N % 64 == 0.  A fixed butterfly per group: independent of the tile shape.
     *   a_stat     statistics of the (raw) A rows as written by their producer ([A rows][a_stat_groups][2], a_stat_groups = K / 64).
     *              W must carry the LayerNorm weight (W[n][k] * gamma[k]), bias the term b[n] + sum_k W[n][k] beta[k], and
     *              a_fold_g [N] = sum_k W[n][k] gamma[k]:  out = rstd * (acc - mean * a_fold_g[n]) + bias[n]  ( = W LN(a) + b ).
     *   res_stat   statistics of the (raw) residual rows ([C rows][N / 64][2]): the residual added is LayerNorm(residual) with
     *              res_gamma / res_beta [N].
     * ln_eps2 = the LayerNorm epsilon.  Plain forward inputs only (no preact / dact / dropout / rowtab / act_post / scale); bf16-split
     * kernels only (AFM_E_UNSUPPORTED when the arithmetic selects the native kernels). */
    float* stat_out; const float* a_stat; int32_t a_stat_groups; const float* a_fold_g;
    const float* res_stat; const float* res_gamma; const float* res_beta; float ln_eps2;
    /* ---- a hole inside every group of the row remaps (ABI v6; 0 = none): with a_skip > 0 the members j >= a_skip_after of a group sit
     * a_skip rows further on: A row = (r / a_grp) * a_stride + a_off + j + (j >= a_skip_after ? a_skip : 0), j = r % a_grp (c_* alike).
     * One launch then covers the time token AND the L motion tokens of every sample while skipping the n_cond step-invariant condition
     * tokens between them (layer 0's in_proj of the sampling loop: a_grp = 1 + L, a_stride = T, a_skip_after = 1, a_skip = n_cond). */
    int32_t a_skip_after, a_skip, c_skip_after, c_skip;
    /* ---- clip_denoised (ABI v6; with ddpm_out only): the epilogue's value (pred_xstart) is clamped to [-1, 1] before it is stored to C
     * and enters the DDPM update - `process_xstart` of gaussian_diffusion.py:289-294 with clip_denoised=True, the reference's default. */
    int32_t ddpm_clip;
    /* ---- riders of the sampling loop's first and last launch of a step (ABI v6; NULL = off; bf16-split kernels only): what used to be a
     * launch of its own per step (prologue_kernel: ~8 us of a 430 us step at one sample per GPU).
     *   ddpm_out2 / ldx2   a second copy of the DDPM update's x_next with row stride ldx2 >= N (the K-padded copy of x_t the NEXT step's
     *                      motion adapter reads; its padding columns are zeroed once per loop)
     *   aux_*              aux_rows rows of aux_cols floats: aux_dst[r * aux_dst_ld + c] = aux_src[clamp(aux_idx[r], 0, aux_idx_max - 1) * aux_cols + c]
     *                      + aux_add[c]  - the time token of every sample (TimestepEmbedder table row of t[r] + positional row 0), written by the
     *                      first aux_rows workgroups of the launch behind their own tile. */
    float* ddpm_out2; int64_t ldx2;
    const float* aux_src; const int64_t* aux_idx; const float* aux_add; float* aux_dst; int64_t aux_dst_ld;
    int32_t aux_rows, aux_cols, aux_idx_max;
} afm_linear_args;

#define AFM_ARITH_DEFAULT 0
#define AFM_ARITH_F32     1
#define AFM_ARITH_BF16X1  3
#define AFM_ARITH_BF16X6  6
#define AFM_ARITH_BF16X9  9

#define AFM_TUNE_NO_DMA      0x1     /* register-staged operand loads instead of global_load_lds                         */
#define AFM_TUNE_TILE_SHIFT  4       /* bits 4..7: force the workgroup tile: 1 = 32x32, 2 = 32x64, 3 = 64x64, 4 = 64x128, 5 = 128x128, 7 = 64x64 with the K segments split over wave groups, 8 = weight-stationary 64-column slabs (row-dot launches with K = 256), 9 = 64x64 on three LDS stages, 10 = split-K on three LDS stages (K = 1024: two groups x two segments), 11 = split-K two groups x two segments on two stages, 12 = 64x64 tiles WALKED by 768 resident workgroups (bf16-split arithmetic), 13 = 256x128 tiles on 512 threads (bf16-split arithmetic; round 6, measurement) */
#define AFM_TUNE_TILE_MASK   0xF0

int afm_linear(const afm_linear_args* args, void* stream);
/* Two independent afm_linear launches as ONE grid of 128 x 128 tiles of the bf16-split tile program (ABI v7): workgroups [0, tiles0) compute
 * args0's output, the rest args1's.  Both must take the bf16-split path with the same arithmetic and K > 256 (AFM_E_UNSUPPORTED otherwise -
 * the caller then issues two afm_linear calls).  Bit-identical to the two calls; the sampling loop pairs one sub-batch's out_proj with the
 * other's linear1 (cmdm.py:66-77: 164 + 328 tiles = one resident round of 512). */
int afm_linear_pair(const afm_linear_args* args0, const afm_linear_args* args1, void* stream);


/* ------------------------------------------------------------------------------------------
 * afm_mha_fwd: multi-head self-attention core, softmax(QK^T / sqrt(dh) + key mask) V.
 * Replaces the attention inside nn.TransformerEncoderLayer's fast path
 * (torch._transformer_encoder_layer_fwd, reached from cmdm.py:167).
 *   qkv [B*T, 3*H*dh] packed as in_proj output (q | k | v), out [B*T, H*dh];
 *   key_mask [B,T] uint8, 1 = padded key (-inf), or NULL.  dh must be 64.
 * Flash-style single pass, S^T = K Q^T and O^T = V^T P^T on f32 MFMA, K/V tiles staged in LDS.
 */
int afm_mha_fwd(const float* qkv, const uint8_t* key_mask, float* out,
                int32_t B, int32_t T, int32_t H, int32_t dh, void* stream);
/* Same with an explicit workgroup shape: group_waves in {1, 2, 4, 6, 8, 12} 32-query waves per workgroup (the grid is
 * (sample, head, query group)), or 100 + {2, 4}: the two key segments of every query block on two waves (workgroups of twice that many
 * waves - what small launches take: the serial chain of a wave halves); 0 = the library's choice from B*H and T.  The softmax of a
 * row is computed over two key segments (blocks [0, ceil(nkb / 2)) and the rest) whose states are merged in a fixed operation order, so
 * a query row's arithmetic does not depend on the grouping: results are bit-identical. */
int afm_mha_fwd_grouped(const float* qkv, const uint8_t* key_mask, float* out,
                        int32_t B, int32_t T, int32_t H, int32_t dh, int32_t group_waves, void* stream);
/* Same, for the query rows q_first .. T - 1 of every sample only (all T keys): rows 0 .. q_first - 1 of `out` are not written.  The CMDM's
 * last encoder layer is read on its L motion tokens only (cmdm.py:169,195), so its attention skips the 2 + n_groups condition-token
 * queries (q_first = 130 of T = 326: 7 query blocks instead of 11).  A query row's arithmetic does not depend on q_first (bit-identical
 * to afm_mha_fwd_grouped on the rows it computes).  ABI v6. */
int afm_mha_fwd_rows(const float* qkv, const uint8_t* key_mask, float* out,
                     int32_t B, int32_t T, int32_t H, int32_t dh, int32_t q_first, int32_t group_waves, void* stream);
/* Same with the arithmetic of the two products S = Q K^T and O = P V as an argument (ABI v7), `arith` as in afm_linear_args: the inference
 * kernel splits Q, K, V and the probabilities P exactly into three bf16 terms; AFM_ARITH_DEFAULT / AFM_ARITH_BF16X6 run the six largest cross
 * products of every operand pair, every other setting all nine (exact f32 products).  q_first = 0: all query rows.  The entry points above
 * (and afm_mha_cross_fwd) run AFM_ARITH_DEFAULT.  A query row's arithmetic depends on `arith` only - never on the grouping, q_first or B. */
int afm_mha_fwd_arith(const float* qkv, const uint8_t* key_mask, float* out,
                      int32_t B, int32_t T, int32_t H, int32_t dh, int32_t q_first, int32_t group_waves, int32_t arith, void* stream);

/* Cross-attention core of nn.TransformerDecoderLayer (CMDM `trans_dec`, cmdm.py:78-113,171-191): Tq queries q [B*Tq, H*dh] over a
 * packed memory kv [B*Tk, 2*H*dh] (k | v), key_mask [B,Tk] or NULL.  Same kernel as afm_mha_fwd (dh = 64). */
int afm_mha_cross_fwd(const float* q, const float* kv, const uint8_t* key_mask, float* out,
                      int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t dh, void* stream);

/* afm_layernorm: y = LN(x) * gamma + beta over the last dim (eps 1e-5).  Replaces nn.LayerNorm
 * (norm1 / norm2 of the encoder layer, modules.py:399-400,459,653).  In-place allowed. */
int afm_layernorm(const float* x, const float* gamma, const float* beta, float* y,
                  int64_t rows, int32_t dim, float eps, void* stream);
/* Same, on a strided subset of token rows: logical row r -> (r / grp) * stride + off + r % grp of both x and y
 * (grp == 0: identity).  Used to normalise only the L motion tokens of each sample in the last encoder layer. */
int afm_layernorm_rows(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t dim,
                       float eps, int32_t grp, int32_t stride, int32_t off, void* stream);

/* ------------------------------------------------------------------------------------------
 * afm_ddpm_step: x_next = (c1[b] * x0 + c2[b] * x_t) + sigma[b] * noise, evaluated WITHOUT fma
 * contraction so it is bit-identical to the reference's float32 expression
 * (gaussian_diffusion.py:222-225 and :439) given the same inputs.  c1/c2/sigma are per-sample
 * [B] device arrays the host gathers from the float32-cast schedule tables
 * (sigma = (t != 0) * exp(0.5 * posterior_log_variance_clipped[t])).
 * If noise == NULL, counter-based Philox4x32-10 + Box-Muller noise keyed by
 * (seed, sample_index0 + b, step, element) is generated in-kernel (sharding-invariant).
 */
int afm_ddpm_step(const float* x0, const float* x_t, const float* noise, float* x_next,
                  const float* c1, const float* c2, const float* sigma,
                  int32_t B, int64_t per_sample, uint64_t seed, int64_t sample_index0,
                  int32_t step, void* stream);

/* afm_clamp: x <- min(max(x, lo), hi) in place (NaN propagates, as torch.clamp): `process_xstart` with clip_denoised=True on the
 * step-by-step path (gaussian_diffusion.py:289-294).  ABI v6. */
int afm_clamp(float* x, int64_t n, float lo, float hi, void* stream);

/* afm_randn: standalone Philox normal generator with the same keying as afm_ddpm_step
 * (replaces th.randn / th.randn_like, gaussian_diffusion.py:431,514). */
int afm_randn(float* out, int32_t B, int64_t per_sample, uint64_t seed, int64_t sample_index0,
              int32_t step, void* stream);

/* afm_bn_fold (ABI v5): eval-mode nn.BatchNorm1d as y = x * scale + shift (pointtransformer.py:31-36,56-57,111-113 call it after a
 * Linear): scale = w / sqrt(running_var + eps), shift = b - running_mean * scale (+ lin_bias * scale when the Linear has a bias),
 * every operation individually rounded.  All vectors [C]. */
int afm_bn_fold(const float* w, const float* b, const float* running_mean, const float* running_var, float eps, const float* lin_bias,
                float* scale, float* shift, int32_t C, void* stream);

/* afm_contact_glue (ABI v5): the ADM -> AMDM hand-off of the two-stage pipeline kept in HBM.  The reference writes
 * dist = sqrt(-2 ln(clip(sample * std + mean, 1e-20, 1)) sigma^2) to H3D/pred_contact/<id>.npy (utils/evaluate.py:41-82) and reads it back as
 * exp(-dist^2 / (2 sigma^2)) (datasets/humanml3d.py:763-774); out[i] is that condition value, n elements.  sigma_sq = sigma^2 rounded to
 * float32 once by the caller (the reference's Python scalar `sigma ** 2`). */
int afm_contact_glue(const float* sample, float* out, int64_t n, float sigma_sq, float mean, float std, void* stream);

/* afm_masked_mse: out[b] = sum((target - pred)^2 * keep) / (sum(keep) * D) over [L, D], keep = !frame_mask.
 * Replaces the loss reduction of training_losses (gaussian_diffusion.py:815-818, sum_flat nn.py:93-97). */
int afm_masked_mse(const float* target, const float* pred, const uint8_t* frame_mask, float* out,
                   int32_t B, int32_t L, int32_t D, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training (backward) entry points - the kernels behind `loss.backward()` of training_losses
 * (gaussian_diffusion.py:757-823 called from utils/training.py:140-152).  Gradients are
 * deterministic (fixed-order split reductions, no atomics on floats).  The two scatter-adds of the point branch whose reference CUDA
 * kernels use f32 atomics - the backward of a row gather / of the neighbour grouping and of the 3-NN interpolation - run as segmented sums
 * over an inverse index built once per index list (afm_scatter_plan + afm_segment_sum_rows, ABI v7); the atomic forms
 * (afm_scatter_add_rows, afm_interpolate_bwd) remain for callers without a plan and are the only order-dependent entry points.
 */

/* out[c][r] = in[r][c] (rows x cols -> cols x rows).  The input-gradient GEMM dX = dY @ W is run as
 * afm_linear(A = dY, W = W^T), so the [N,K] nn.Linear weights are transposed once per optimiser step. */
int afm_transpose(const float* in, float* out, int32_t rows, int32_t cols, void* stream);

/* Weight / bias gradient of y = x @ W^T + b:  dW[N,K] = dY^T @ X,  db[N] = colsum(dY)  (db may be NULL).
 * dY [M,N] (row stride lddy), X [M,K] (row stride ldx); logical row r of dY / X lives at row
 * (r / grp) * stride + off + r % grp when the corresponding grp != 0 (token subsets, as in afm_linear).
 * f32 MFMA with both operands reduction-major; the M reduction is split over workgroups into `ws`
 * (afm_linear_wgrad_workspace_bytes) and summed in a fixed order.  accumulate != 0: dW += / db +=. */
typedef struct {
    const float* dY; int64_t lddy;
    const float* X; int64_t ldx;
    float* dW; int64_t lddw;
    float* db;
    int32_t M, N, K;
    int32_t dy_grp, dy_stride, dy_off;
    int32_t x_grp, x_stride, x_off;
    int32_t accumulate;
    void* ws; int64_t ws_bytes;
} afm_linear_wgrad_args;
int64_t afm_linear_wgrad_workspace_bytes(int32_t M, int32_t N, int32_t K);
int afm_linear_wgrad(const afm_linear_wgrad_args* args, void* stream);

/* nn.LayerNorm backward.  x = the LayerNorm INPUT (statistics are recomputed), dy = grad of the output.
 *   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma
 *   dgamma = sum_rows dy * xhat, dbeta = sum_rows dy  (two-stage fixed-order reduction through ws)
 * dx_drop (optional): dx with the dropout keep-mask of (drop_p, drop_seed, drop_id) applied - the gradient of the
 * residual BRANCH when the forward was  LN(x + dropout(branch))  (encoder layer dropout1 / dropout2). */
int64_t afm_layernorm_bwd_workspace_bytes(int64_t rows, int32_t dim);
int afm_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dx_drop,
                      float* dgamma, float* dbeta, int64_t rows, int32_t dim, float eps,
                      float drop_p, uint64_t drop_seed, uint32_t drop_id, void* ws, int64_t ws_bytes, void* stream);

/* Training-mode attention: as afm_mha_fwd plus lse [B,H,T] (log-sum-exp of the scaled, masked logits, saved for the
 * backward) and attention-probability dropout (F.multi_head_attention_forward dropout_p; 0 = off). */
int afm_mha_fwd_train(const float* qkv, const uint8_t* key_mask, float* out, float* lse,
                      int32_t B, int32_t T, int32_t H, int32_t dh,
                      float drop_p, uint64_t drop_seed, uint32_t drop_id, void* stream);
/* Attention backward: dqkv [B*T, 3*H*dh] packed like qkv, from qkv / out / lse of the forward and dout [B*T, H*dh].
 * Two flash-style passes on f32 MFMA (probabilities recomputed from lse, nothing [T,T]-sized is stored):
 * dQ with one wave per 32-query block, then dK/dV with one wave per 32-key block.  ws: B*H*T floats. */
int afm_mha_bwd(const float* qkv, const uint8_t* key_mask, const float* out, const float* dout, const float* lse,
                float* dqkv, int32_t B, int32_t T, int32_t H, int32_t dh,
                float drop_p, uint64_t drop_seed, uint32_t drop_id, void* ws, int64_t ws_bytes, void* stream);
/* The same pair for cross-attention (nn.TransformerDecoderLayer.multihead_attn of the CMDM's `trans_dec` variant under autograd, cmdm.py:78-113,
 * 171-191): q [B,Tq,H*dh], kv [B,Tk,2*H*dh] = packed K | V projections of the memory, key_mask [B,Tk] or NULL; the forward also writes
 * lse [B*H, Tq]; the backward fills dq [B,Tq,H*dh] and dkv [B,Tk,2*H*dh].  ws: B*H*Tq floats.  Dropout as afm_mha_fwd_train. */
int afm_mha_cross_fwd_train(const float* q, const float* kv, const uint8_t* key_mask, float* out, float* lse, int32_t B, int32_t Tq, int32_t Tk,
                            int32_t H, int32_t dh, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* stream);
int afm_mha_cross_bwd(const float* q, const float* kv, const uint8_t* key_mask, const float* out, const float* dout, const float* lse, float* dq,
                      float* dkv, int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t dh, float drop_p, uint64_t drop_seed, uint32_t drop_id,
                      void* ws, int64_t ws_bytes, void* stream);

/* d(afm_masked_mse)/d(pred): dpred[b,l,:] = dloss[b] * 2 * (pred - target) * keep[b,l] / (sum(keep[b]) * D). */
int afm_masked_mse_bwd(const float* target, const float* pred, const uint8_t* frame_mask, const float* dloss,
                       float* dpred, int32_t B, int32_t L, int32_t D, void* stream);

/* out[r,c] = (x[r,c] + rowtab[r % period, c]) * act'(z[r,c]) * keep(r,c)/(1-p)  (rowtab, z optional; in-place allowed).
 * The elementwise glue of the training graph: PositionalEncoding add + dropout (modules.py:43-45) and the
 * activation / dropout backward in front of a stand-alone linear's gradient GEMMs. */
int afm_rowop(const float* x, const float* rowtab, int32_t period, const float* z, int32_t act, float* out,
              int64_t rows, int32_t cols, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* stream);

/* ---- train-mode point-cloud operators (BatchNorm on batch statistics; models/scene_models/pointtransformer.py:26-69,
 * 102-123 under model.train()).  Batch statistics force a full pass between every BatchNorm and its consumer, so the
 * training graph is composed from bandwidth-bound passes over materialised [n, k, c] tensors. */

/* Column statistics of a row-major [rows, C] matrix (C <= 512, rows >= 1), fixed summation order, taken about the
 * matrix's first row K (shifted-data variance: no E[x^2] - mean^2 cancellation):
 *   stats[0..C) = sum_r (x[r,c] - K[c]),  stats[C..2C) = sum_r (x[r,c] - K[c])^2,  stats[2C..3C) = K. */
int64_t afm_colstats_workspace_bytes(int64_t rows, int32_t C);
int afm_colstats(const float* x, int64_t rows, int32_t C, float* stats, void* ws, int64_t ws_bytes, void* stream);
/* nn.BatchNorm1d / nn.SyncBatchNorm training-mode bookkeeping from the statistics of `world` ranks ([world][3C], as
 * produced by afm_colstats on every rank and all-gathered; world = 1 without synchronisation), rows_per_rank rows each:
 * per-rank means / M2 are merged with the parallel-variance formula; outputs mean, rstd = 1/sqrt(biased var + eps),
 * scale = gamma*rstd, shift = beta - mean*scale, and the running-statistics update
 * running = (1-momentum)*running + momentum*batch (unbiased variance); running_* may be NULL. */
int afm_bn_finalize(const float* stats, int32_t world, int64_t rows_per_rank, const float* gamma, const float* beta, float eps,
                    float momentum, float* running_mean, float* running_var, float* mean, float* rstd, float* scale,
                    float* shift, int32_t C, void* stream);
/* y = relu?(x * scale[c] + shift[c] + residual)   (BatchNorm apply [+ identity] [+ ReLU]; residual may be NULL) */
int afm_colaffine(const float* x, const float* scale, const float* shift, const float* residual, int32_t relu, float* y,
                  int64_t rows, int32_t C, void* stream);
/* BatchNorm backward, stage 1: stats[0..C) = sum g, stats[C..2C) = sum g*xhat with g = dy * (y > 0 if y given) - these
 * are dbeta and dgamma.  Stage 2: dx = gamma*rstd*(g - sum_g/count - xhat*sum_gx/count), dres = g (optional). */
int afm_bn_bwd_stats(const float* dy, const float* x, const float* y, const float* mean, const float* rstd, int64_t rows,
                     int32_t C, float* stats, void* ws, int64_t ws_bytes, void* stream);
int afm_bn_bwd_apply(const float* dy, const float* x, const float* y, const float* mean, const float* rstd,
                     const float* gamma, const float* stats, int64_t count, float* dx, float* dres, int64_t rows,
                     int32_t C, void* stream);
/* out[r, 0:3] = xyz[idx[r]] - new_xyz[r / k], out[r, 3:3+C] = feat[idx[r]]  (pointops.queryandgroup, pointops.py:79-100;
 * C = 0: relative coordinates only). */
int afm_group_points(const float* xyz, const float* new_xyz, const float* feat, const int32_t* idx, float* out,
                     int64_t rows, int32_t k, int32_t C, void* stream);
/* dst[idx[r], c] += src[r*ld + col_offset + c], c < C  (backward of a row gather; f32 atomics like the CUDA original) */
int afm_scatter_add_rows(const float* src, int64_t ld, int32_t col_offset, const int32_t* idx, float* dst, int64_t rows,
                         int32_t C, void* stream);
/* Deterministic form (ABI v7).  afm_scatter_plan: the inverse of an index list idx [entries] with values in [0, n_dst) - plan = int32 words
 * [n_dst + 1 segment offsets | entries: the entries of every destination, ASCENDING | scratch], afm_scatter_plan_words(entries, n_dst) words in
 * all; integer atomics only, the result does not depend on any arrival order.  afm_segment_sum_rows: dst[d, c] = sum over the entries e of
 * destination d, in ascending order, of weight(e) * src[(e / row_div) * ld + col_offset + c]  (every destination row is WRITTEN: no zero-fill;
 * row_div = 1: the backward of a row gather, entry = source row; row_div = k with dist2 [rows, k]: the backward of afm_interpolate, weight(e) =
 * w_e / sum_j w_(row, j), w = 1 / (sqrt(dist2) + 1e-8)).  One plan serves every operator that scatters through the same index list. */
int64_t afm_scatter_plan_words(int64_t entries, int64_t n_dst);
int afm_scatter_plan(const int32_t* idx, int64_t entries, int64_t n_dst, int32_t* plan, void* stream);
int afm_segment_sum_rows(const float* src, int64_t ld, int32_t col_offset, int32_t row_div, const float* dist2, const int32_t* plan, float* dst,
                         int64_t n_dst, int32_t C, void* stream);
/* nn.MaxPool1d(k) over [m, k, C] with argmax, and its backward (pointtransformer.py:66-68) */
int afm_group_max(const float* x, float* y, int32_t* arg, int64_t m, int32_t k, int32_t C, void* stream);
int afm_group_max_bwd(const float* dy, const int32_t* arg, float* dx, int64_t m, int32_t k, int32_t C, void* stream);
/* out[g,c] = scale * sum_j x[g,j,c] */
int afm_group_sum(const float* x, float* out, int64_t m, int32_t k, int32_t C, float scale, void* stream);
/* vector-attention glue of PointTransformerLayer (pointtransformer.py:34-37):
 * w0 = k_g - q[:,None] + p_r;  sw = softmax_k(w2);  out[g, s*Cs+j] = sum_k (v_g + p_r)[g,k,s*Cs+j] * sw[g,k,j] */
int afm_pt_w0(const float* kg, const float* q, const float* pr, float* out, int64_t m, int32_t k, int32_t C, void* stream);
int afm_pt_aggregate(const float* vg, const float* pr, const float* w2, float* out, float* sw, int64_t m, int32_t k,
                     int32_t C, int32_t share_planes, void* stream);
int afm_pt_aggregate_bwd(const float* vg, const float* pr, const float* sw, const float* dout, float* da, float* dw2,
                         int64_t m, int32_t k, int32_t C, int32_t share_planes, void* stream);

/* ---- training-path attention of the CDM ContactPerceiver (models/cdm.py:155-188 under model.train()).
 * Few-query cross-attention (encoder, 2 latent queries over N point keys): Q [B,2,C], K/V [B,N,C], H heads;
 * P [B, H*2, N] (row h*2+q) receives the softmax probabilities (saved for the backward), O [B,2,C];
 * attention-probability dropout by the (seed, id, row of P, n) counter hash.  C in {256, 512}. */
int64_t afm_xq_workspace_bytes(int32_t B, int32_t N, int32_t C);
int afm_xq_attention_fwd(const float* Q, const float* K, const float* V, float* P, float* O, int32_t B, int32_t N, int32_t H,
                         int32_t C, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* ws, int64_t ws_bytes, void* stream);
int afm_xq_attention_bwd(const float* Q, const float* K, const float* V, const float* P, const float* dO, float* dS,
                         float* dQ, float* dK, float* dV, int32_t B, int32_t N, int32_t H, int32_t C, float drop_p,
                         uint64_t drop_seed, uint32_t drop_id, void* ws, int64_t ws_bytes, void* stream);
/* Few-key cross-attention (decoder, N point queries over 2 latent keys): Q/O [B,N,C], K/V [B,2,C]; probabilities are
 * recomputed in the backward; dK/dV are fixed-order sums of per-workgroup partials (ws). */
int64_t afm_xk_workspace_bytes(int32_t B, int32_t N, int32_t C);
int afm_xk_attention_fwd(const float* Q, const float* K, const float* V, float* O, int32_t B, int32_t N, int32_t H, int32_t C,
                         float drop_p, uint64_t drop_seed, uint32_t drop_id, void* stream);
int afm_xk_attention_bwd(const float* Q, const float* K, const float* V, const float* dO, float* dQ, float* dK, float* dV,
                         int32_t B, int32_t N, int32_t H, int32_t C, float drop_p, uint64_t drop_seed, uint32_t drop_id,
                         void* ws, int64_t ws_bytes, void* stream);

/* Fused AdamW over one flat parameter (torch.optim.AdamW semantics, utils/training.py:48-53):
 *   p *= 1 - lr*wd; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps) */
int afm_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
              float eps, float weight_decay, int32_t step, void* stream);
/* The same update for every parameter tensor of the model in ONE launch: d_table is a DEVICE array of n_tensors
 * descriptors (all tensors at the same step count), max_n the largest element count. */
typedef struct { float* p; const float* g; float* m; float* v; int64_t n; } afm_adamw_tensor;
int afm_adamw_multi(const afm_adamw_tensor* d_table, int32_t n_tensors, int64_t max_n, float lr, float beta1, float beta2,
                    float eps, float weight_decay, int32_t step, void* stream);

/* ------------------------------------------------------------------------------------------
 * Point-cloud operators.  Every sample holds the same number of points, so the reference's
 * offset arrays are implicit: sample b owns rows [b*n, (b+1)*n).  Returned indices are GLOBAL rows.
 *
 * afm_fps replaces pointops_cuda.furthestsampling_cuda (pointops.py:10-27; called from
 * TransitionDown.forward, pointtransformer.py:61): start at the sample's first point, running
 * min squared distance initialised to 1e10, arg-max each round (ties -> lowest index).
 * afm_knn replaces pointops_cuda.knnquery_cuda (pointops.py:30-45): brute force, neighbours in
 * ascending (dist2, index) order; dist2 is the SQUARED distance (the wrapper's sqrt is the caller's).
 * Both evaluate d2 = (dx*dx + dy*dy) + dz*dz in float32 without fma contraction (bit-exact indices
 * against oracle/pointops_ref.py).  k in {3, 8, 16}.
 */
int afm_fps(const float* xyz, int32_t B, int32_t n, int32_t m, int32_t* idx_out, void* stream);
int afm_knn(int32_t k, const float* xyz, const float* new_xyz, int32_t B, int32_t n, int32_t m,
            int32_t* idx_out, float* dist2_out, void* stream);
/* Same result through an EXACT spatial pruning (ABI v7): the candidates of a sample are sorted along a Morton curve and cut into tiles of 64 with
 * bounding boxes, the queries are sorted the same way, and a wave scans a tile only if some lane's distance to the tile's box (same float operations
 * as the point distances) is not above its current k-th distance; the k best are kept as (distance bits, index) keys, so the neighbours and
 * their order are those of afm_knn bit for bit.  workspace: afm_knn_workspace_bytes(k, B, n, m) bytes, 16-byte aligned (0 = the form does not apply and
 * afm_knn_ws runs afm_knn: it is taken for 4096 <= n <= 8192 candidates with m >= n queries and k in {3, 8, 16} - the large self-searches, the only
 * shapes of this path where it measured faster, 0.60 vs 0.83 ms at 32 x 8192 points, k = 8). */
int64_t afm_knn_workspace_bytes(int32_t k, int32_t B, int32_t n, int32_t m);
int afm_knn_ws(int32_t k, const float* xyz, const float* new_xyz, int32_t B, int32_t n, int32_t m, int32_t* idx_out, float* dist2_out,
               void* workspace, int64_t workspace_bytes, void* stream);
/* out[r, :] = src[idx[r], :]  (the `p[idx.long(), :]` of pointtransformer.py:62) */
int afm_gather_rows(const float* src, const int32_t* idx, float* out, int64_t rows, int32_t c, void* stream);

/* afm_interpolate: inverse-distance-weighted feature upsampling of pointops.interpolation (pointops.py:164-178):
 *   out[i,:] = base[i,:] + sum_j w_ij feat[idx[i,j],:],  w_ij = r_ij / sum_j r_ij,  r_ij = 1 / (sqrt(dist2[i,j]) + 1e-8)
 * idx / dist2 [n,k] come from afm_knn (k = 3); base may be NULL.  Fuses the `linear1(x1) +` of TransitionUp
 * (pointtransformer.py:98).  afm_segment_mean: per-sample mean over n rows (TransitionUp head mode, :90). */
int afm_interpolate(const float* feat, const int32_t* idx, const float* dist2, const float* base, float* out,
                    int64_t n, int32_t c, int32_t k, void* stream);
int afm_segment_mean(const float* x, float* out, int32_t B, int32_t n, int32_t c, void* stream);
/* backward of afm_interpolate with respect to feat (autograd of pointops.interpolation reached from loss.backward() of the PointTrans
 * denoisers, utils/training.py:152): dfeat [m, c] = 0, then dfeat[idx[i,j], :] += w_ij * dout[i, :] over the n fine points (f32 atomics,
 * as the reference's CUDA backward).  The gradient of `base` is dout itself. */
int afm_interpolate_bwd(const float* dout, const int32_t* idx, const float* dist2, float* dfeat, int64_t n, int64_t m, int32_t c,
                        int32_t k, void* stream);

/* afm_transition_down: fused "set abstraction" of TransitionDown.forward (pointtransformer.py:53-69,
 * stride != 1, eval-mode BN folded to scale/shift):
 *   out[i, o] = max_j ReLU(scale[o] * (W[o,:] . [p[knn[i,j]] - new_p[i] ; x[knn[i,j]]]) + shift[o])
 * p [R,3], x [R,c], new_p [M,3], knn_idx [M,nsample] (nsample must be 16), weight [cout, 3+c].
 * The grouped (M, nsample, 3+c) tensor of pointops.queryandgroup (pointops.py:79-100) is never built. */
int afm_transition_down(const float* p, const float* x, int32_t c, const float* new_p, const int32_t* knn_idx,
                        int32_t nsample, const float* weight, int32_t cout, const float* scale, const float* shift,
                        float* out, int32_t M, void* stream);

/* afm_pt_attention: PointTransformerLayer.forward after the q/k/v projections (pointtransformer.py:26-38):
 *   p_r = lp3(ReLU(BN(lp0(p_j - p_i)))),  w = softmax_j(w5(ReLU(BN(w2(ReLU(BN(k_j - q_i + p_r))))))),
 *   out[i, c] = sum_j (v_j + p_r)[c] * w[j, c mod (C/share_planes)],  then optional out*out_scale+out_shift, ReLU.
 * qkv [n, 3C] = [linear_q(x) | linear_k(x) | linear_v(x)], knn_idx [n, nsample] (self-kNN, nsample 8 or 16).
 * BatchNorms are passed folded (scale, shift). */
typedef struct {
    const float* p; const float* qkv; const int32_t* knn_idx; float* out;
    int32_t n, channels, nsample, share_planes;
    const float* lp0_w; const float* lp0_b;              /* linear_p.0  [3,3],[3]      */
    const float* lp_bn_scale; const float* lp_bn_shift;  /* linear_p.1  BN(3)          */
    const float* lp3_w; const float* lp3_b;              /* linear_p.3  [C,3],[C]      */
    const float* w0_bn_scale; const float* w0_bn_shift;  /* linear_w.0  BN(C)          */
    const float* w2_w; const float* w2_b;                /* linear_w.2  [C/s, C],[C/s] */
    const float* w3_bn_scale; const float* w3_bn_shift;  /* linear_w.3  BN(C/s)        */
    const float* w5_w; const float* w5_b;                /* linear_w.5  [C/s,C/s],[C/s]*/
    const float* out_scale; const float* out_shift;      /* optional fused bn2 of the block, or NULL */
    int32_t relu;
} afm_pt_attention_args;
int afm_pt_attention(const afm_pt_attention_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * CMDM (`trans_enc`) denoiser forward, sampling form.  Replaces CMDM.forward (cmdm.py:118-196)
 * with the step-invariant condition tokens supplied pre-computed (they depend only on
 * c_text / c_pc_xyz / c_pc_contact, see afm_cmdm_cond_tokens in the Python host layer).
 */
typedef struct {
    const float* in_proj_w;  const float* in_proj_b;    /* [3d,d], [3d]  self_attn.in_proj_*      */
    const float* out_proj_w; const float* out_proj_b;   /* [d,d],  [d]   self_attn.out_proj.*     */
    const float* lin1_w; const float* lin1_b;           /* [ff,d], [ff]  linear1.*                */
    const float* lin2_w; const float* lin2_b;           /* [d,ff], [d]   linear2.*                */
    const float* norm1_w; const float* norm1_b;         /* [d]           norm1.*                  */
    const float* norm2_w; const float* norm2_b;         /* [d]           norm2.*                  */
    /* ABI v5, all six or none (the host builds them in eval mode, float64 products rounded once): the layer's two LayerNorm-fed linears
     * with the LayerNorm folded in (afm_linear_args.a_stat): W' = W * gamma (per input column), g = row sums of W', c = b + W beta.
     * lin1_* folds norm1 of THIS layer; in_proj_* folds norm2 of the PREVIOUS layer (unused in layer 0, whose input is not normalised). */
    const float* lin1_wg; const float* lin1_g; const float* lin1_c;             /* [ff,d], [ff], [ff]   */
    const float* in_proj_wg; const float* in_proj_g; const float* in_proj_c;    /* [3d,d], [3d], [3d]   */
} afm_encoder_layer_weights;

#define AFM_MAX_LAYERS 16

typedef struct {
    int32_t d, heads, ff, n_layers;        /* latent_dim, num_heads, dim_feedforward, sum(num_layers) */
    int32_t motion_dim;                    /* input_feats (263 'h3d', 66 'pos')                        */
    int32_t n_cond;                        /* number of step-invariant tokens after the time token      */
    const float* motion_adapter_w; const float* motion_adapter_b;   /* [d, motion_dim], [d]  */
    const float* motion_layer_w;   const float* motion_layer_b;     /* [motion_dim, d], [motion_dim] */
    const float* time_table;               /* [n_timesteps, d] TimestepEmbedder output for every t (modules.py:52-53) */
    int32_t n_timesteps;
    const float* pos_table;                /* [>= 1+n_cond+L, d] sinusoid table (modules.py:10-26)    */
    afm_encoder_layer_weights layer[AFM_MAX_LAYERS];
    /* ABI v3: arithmetic of every nn.Linear of the denoiser (afm_linear_args.arith / arith_min_n) and bit-neutral tuning */
    int32_t gemm_arith, gemm_arith_min_n;
    int32_t attn_group_waves;              /* afm_mha_fwd_grouped's group_waves (0 = auto)                              */
    int32_t flags;                         /* AFM_CMDM_* bits                                                           */
    /* ABI v5: 0, or the row length (a multiple of 4, >= motion_dim) motion_adapter_w is zero-padded to: [d, kpad].  The loop then keeps a
     * padded copy of x_t in its workspace so that the adapter (K = 263 for 'h3d') runs with K = 272 on the bf16-split GEMM. */
    int32_t motion_adapter_kpad;
    /* ABI v5: motion_layer with the LAST layer's norm2 folded in (see afm_encoder_layer_weights.lin1_wg); NULL = not folded.  With all
     * folded tensors present the step runs WITHOUT LayerNorm launches and without materialised LayerNorm outputs: out_proj / linear2
     * write per-row statistics next to their raw outputs, the consumers apply them (afm_linear_args.stat_out / a_stat / res_stat). */
    const float* motion_layer_wg; const float* motion_layer_g; const float* motion_layer_c;      /* [motion_dim,d], [motion_dim] x 2 */
} afm_cmdm_weights;

#define AFM_CMDM_NO_L0_CACHE 0x1           /* measurement: recompute layer 0's q|k|v rows of the condition tokens every step */
#define AFM_CMDM_NO_LN_FOLD  0x4           /* measurement: separate afm_layernorm launches although the folded tensors are present */
#define AFM_CMDM_WIDE_TILE_SHIFT 8        /* bits 8..11, measurement: AFM_TUNE_TILE code forced on the encoder GEMMs with N >= 512 and M >= 2048 (bit-neutral) */
#define AFM_CMDM_NO_RIDERS   0x20          /* measurement: the per-step prologue launch of round 3 instead of the riders on the first / last GEMM of a step (bit-identical) */
#define AFM_CMDM_CLIP_X0     0x10          /* clip_denoised=True (gaussian_diffusion.py:289-294): pred_xstart clamped to [-1, 1] inside the fused DDPM update */
#define AFM_CMDM_ALL_QUERIES 0x8           /* measurement: the last layer's attention computes all T query rows (bit-identical on the rows that are read) */
#define AFM_CMDM_PAIR_LAUNCH 0x40          /* ABI v7, native loop with two sub-batch streams: sub-batch A's out_proj and sub-batch B's linear1 of every layer as ONE 128 x 128-tile launch (afm_linear_pair; bit-identical) */
#define AFM_CMDM_FUSED_LN    0x2           /* norm1 / norm2 inside out_proj / linear2 (afm_linear_args.ln_*; bit-identical, measured slower: off by default) */

/* bytes of workspace afm_cmdm_forward needs for (B, L). */
int64_t afm_cmdm_workspace_bytes(const afm_cmdm_weights* w, int32_t B, int32_t L);

/* One denoiser evaluation (+ optional fused DDPM update).
 *   x_t [B,L,motion_dim]; t [B] int64 ORIGINAL timesteps (after respace.py:124-129 mapping);
 *   cond_tokens [B, n_cond, d] = adapters(conditions) + positional encoding of positions 1..n_cond;
 *   frame_mask [B,L] uint8 (1 = padded frame) or NULL (mask_motion False);
 *   x0_out [B,L,motion_dim] (may be NULL when ddpm != NULL).
 *   ddpm: if non-NULL, also writes x_next (see afm_ddpm_args).
 */
typedef struct {
    const float* noise;      /* [B,L,motion_dim] or NULL -> Philox */
    float* x_next;           /* [B,L,motion_dim] */
    const float* c1; const float* c2; const float* sigma;   /* [B] */
    uint64_t seed; int64_t sample_index0; int32_t step;
} afm_ddpm_args;

int afm_cmdm_forward(const afm_cmdm_weights* w, const float* x_t, const int64_t* t,
                     const float* cond_tokens, const uint8_t* frame_mask, float* x0_out,
                     const afm_ddpm_args* ddpm, int32_t B, int32_t L,
                     void* workspace, int64_t workspace_bytes, void* stream);

/* Whole p_sample_loop (gaussian_diffusion.py:442-536) enqueued from native code with no host
 * synchronisation and no host<->device traffic: for i = n_steps-1 .. 0: x <- p_sample(x, t = i).
 *   d_timestep_map [n_steps] int64 (respace.py timestep_map), d_c1/d_c2/d_sigma [n_steps] float32
 *   schedule rows (posterior_mean_coef1/2 and (i != 0) * exp(0.5 * posterior_log_variance_clipped)),
 *   all DEVICE arrays indexed by the spaced step i.
 *   x [B,L,motion_dim] holds x_T on entry and the final sample on exit.
 *   step_noise: [n_steps, B, L, motion_dim] device (row j = j-th executed step, t = n_steps-1-j)
 *   or NULL -> Philox keyed by (seed, sample_index0 + b, step = j).
 *   sched_scratch: device scratch of >= afm_cmdm_sched_scratch_bytes(n_steps, B) bytes.
 *   Sub-batching: samples are independent for the whole loop, so the batch may be split into
 *   n_streams contiguous sub-batches, sub-batch s running its own loop on side_streams[s] (caller-owned
 *   hipStream_t handles; NULL / 0 = everything on `stream`).  One sub-batch's kernels fill the
 *   wave-quantisation tails of the other's; results are bit-identical to the single-stream run.
 *   `stream` is joined with every side stream before the call returns its work to the caller
 *   (event wait, no host sync).  workspace must hold afm_cmdm_loop_workspace_bytes(w, B, L, n_streams).
 */
int64_t afm_cmdm_sched_scratch_bytes(int32_t n_steps, int32_t B);
int64_t afm_cmdm_loop_workspace_bytes(const afm_cmdm_weights* w, int32_t B, int32_t L, int32_t n_streams);

int afm_cmdm_sample_loop(const afm_cmdm_weights* w, float* x, const float* cond_tokens,
                         const uint8_t* frame_mask, const float* step_noise,
                         const int64_t* d_timestep_map, const float* d_c1, const float* d_c2,
                         const float* d_sigma, int32_t n_steps, uint64_t seed, int64_t sample_index0,
                         int32_t B, int32_t L, void* sched_scratch, void* workspace,
                         int64_t workspace_bytes, int32_t n_streams, void* const* side_streams, void* stream);

/* A contiguous slice of the same loop: runs executed steps first_step .. first_step+n_steps-1 of a longer chain (the
 * Philox step counter continues at first_step, so chunks chained over [0, T) reproduce afm_cmdm_sample_loop bit for bit).
 * The schedule rows and step_noise passed in are the SLICE's: d_*[i] for the slice's timesteps in ascending order (the
 * slice walks them from index n_steps-1 down to 0), step_noise row 0 = the slice's first executed step.  This is what
 * `progress=True` in the reference's test.py maps to (gaussian_diffusion.py:520-523: tqdm over the step indices): the host
 * advances the progress bar between slices while each slice stays one native enqueue. */
int afm_cmdm_sample_loop_range(const afm_cmdm_weights* w, float* x, const float* cond_tokens,
                               const uint8_t* frame_mask, const float* step_noise,
                               const int64_t* d_timestep_map, const float* d_c1, const float* d_c2,
                               const float* d_sigma, int32_t n_steps, int32_t first_step, uint64_t seed,
                               int64_t sample_index0, int32_t B, int32_t L, void* sched_scratch, void* workspace,
                               int64_t workspace_bytes, int32_t n_streams, void* const* side_streams, void* stream);

/* ------------------------------------------------------------------------------------------
 * CDM (`Perceiver`) denoiser forward.  Replaces CDM.forward + ContactPerceiver.forward
 * (models/cdm.py:474-513,155-188) and the Perceiver-IO blocks it uses (models/modules.py:234-661:
 * CrossAttentionLayer, SelfAttentionBlock, MLP, Residual; pad mask / rotary / kv-cache never used).
 *
 * The encoder has only TWO latent queries (text, time), so its cross-attention over the N points is
 * evaluated without materialising K/V: q.(W_k x + b_k) = (W_k^T q).x + q.b_k and
 * sum_n a_n (W_v x_n + b_v) = W_v (sum_n a_n x_n) + b_v (fp re-association only); the decoder's
 * two-key attention folds the same way.  The dense per-point layers (decoder adapter, MLP) run on afm_linear.
 *
 * All weights are the reference's tensors (state-dict names in the comments, prefix `contact_model.`).
 */
typedef struct { const float* w; const float* b; } afm_lin;                 /* nn.Linear weight [out,in], bias [out] */
typedef struct { const float* g; const float* b; } afm_ln;                  /* nn.LayerNorm weight, bias             */
typedef struct { afm_lin q, k, v, o; } afm_mha_w;                          /* attention.{q,k,v,o}_proj               */
typedef struct { afm_ln norm; afm_lin fc1, fc2; } afm_mlp_w;               /* MLP: module.0 (LN), .1, .3              */
typedef struct {
    int32_t contact_dim;       /* input_feats (6)                                                  */
    int32_t feat_dim;          /* encoder_adapter in_features = contact_dim + point_feat_dim + 3   */
    int32_t dq, dkv;           /* encoder_q_input_channels (512), encoder_kv_input_channels (256)  */
    int32_t enc_heads, dec_heads, n_self;   /* 8, 8, encoder_self_attn_num_layers (2)              */
    int32_t text_dim, time_dim, n_timesteps;
    /* per-timestep tables of the TIME latent, built once per weight version with afm_cdm_latent_tokens(which = 1) on
     * the TimestepEmbedder output of every t (modules.py:38-53): enc_q0 row, folded queries, folded key-bias terms */
    const float* time_q0;      /* [n_timesteps, dq]                 */
    const float* time_u;       /* [n_timesteps, enc_heads, dkv]     */
    const float* time_cu;      /* [n_timesteps, enc_heads]          */
    afm_lin language_adapter, time_embedding_adapter, encoder_adapter, decoder_adapter;
    afm_ln enc_q_norm, enc_kv_norm; afm_mha_w enc_attn; afm_mlp_w enc_mlp;          /* encoder_cross_attn.{0,1}.module */
    afm_ln self_norm[4]; afm_mha_w self_attn[4]; afm_mlp_w self_mlp[4];             /* encoder_self_attn.{l}.{0,1}.module */
    afm_ln dec_q_norm, dec_kv_norm; afm_mha_w dec_attn; afm_mlp_w dec_mlp;          /* decoder_cross_attn.{0,1}.module */
    afm_lin contact_layer;     /* [contact_dim, dkv]                                               */
    int32_t gemm_arith, gemm_arith_min_n;   /* ABI v3: afm_linear_args.arith / arith_min_n of the dense per-point layers */
    /* ABI v4 (all five or none; the host builds them in eval mode when contact_dim <= 8 < ... and feat_dim > contact_dim): weight
     * products that let the sampling form skip everything that is linear in step-invariant data.  Only the contact_dim leading
     * columns of the encoder input (the noisy contact map x_t) change between the steps of a loop and no nonlinearity separates
     * encoder_adapter from decoder_adapter, nor linear2 (+ residual) from contact_layer (cdm.py:176-186,509-510):
     *   enc_kv[n] = C[n] + sum_j x_t[n,j] xu[j],  C = encoder_adapter(input with x = 0)      computed once per loop,
     *   dec_q0[n] = D[n] + sum_j x_t[n,j] xv[j],  D = decoder_adapter(C)                     computed once per loop,
     *   out[n]    = w2 . GELU(linear1 z[n]) + contact_layer.w . h1[n] + c0                   (linear2 and h1 never materialise;
     *               contact_layer.w . h1 = sum_jh a[n,jh] (contact_layer.w . P[jh]) + contact_layer.w . dec_q0[n] + const, the last
     *               again split into an invariant part E[n] and q . x_t[n]).
     * Same function as the layer-by-layer form up to f32 re-association (tests: 2e-4 abs against the reference goldens). */
    const float* fold_xu;      /* [contact_dim, dkv]  encoder_adapter.w[:, j]                                 */
    const float* fold_xv;      /* [contact_dim, dkv]  decoder_adapter.w @ encoder_adapter.w[:, j]             */
    const float* fold_w2;      /* [contact_dim, dkv]  contact_layer.w @ dec_mlp.fc2.w                         */
    int32_t flags;             /* AFM_CDM_* bits (ABI v4)                                                       */
    const float* fold_q;       /* [contact_dim, contact_dim]  contact_layer.w @ fold_xv^T                     */
    const float* fold_c0;      /* [contact_dim]  contact_layer.w @ (dec_mlp.fc2.b + dec_attn.o.b) + contact_layer.b */
    /* ABI v5 (all or none; built by the host next to fold_* when feat_dim + 1 <= 44): the sampling form without per-point rows.  A point is
     * x = [x_t | point features, xyz | 1 | 0 ...], K numbers: K = 12 when feat_dim + 1 <= 12 (the H3D variant's 9 input channels), else K = 44 (the
     * HUMANISE variant's 41), NT = ceil(K / 16).  Everything between the nonlinearities is linear in x and in the 16 attention weights a[jh] of the
     * decoder:
     *   encoder side   the rows the two latents attend over are LayerNorm_kv(x G_enc): var = x Qe x^T, any dot with a vector u is rstd (x . (Ec u)),
     *                  the attention-weighted row sum is linear in sum_n a_n rstd_n x_n (enc_point_kernel accumulates 16 x K numbers per wave):
     *     enc_ec  [K, dkv]   G_enc = [encoder_adapter.w^T ; encoder_adapter.b ; 0] minus its row means
     *     enc_qee [K, 16 NT] enc_ec enc_ec^T / dkv in MFMA operand order: entry (k, 16 t + i) = Q[4 (4 t + (i & 3)) + (i >> 2)][k], 0 where that index is >= K
     *     enc_wove [8 K, dq], enc_c1 [dq]   head of the latent chain: x1 = q0 + o_proj(v_proj(.)) as one product with the 8 x K accumulated numbers
     *                  (row K h + k = o_proj.w[:, head h] v_proj.w[head h] (enc_kv_norm.w * enc_ec[k]); enc_c1 = o_proj.b + o_proj.w (v_proj.w enc_kv_norm.b + v_proj.b))
     *   decoder side   with G_dec = [(decoder_adapter.w encoder_adapter.w)^T ; decoder_adapter.w encoder_adapter.b + decoder_adapter.b ; 0]:
     *                  scores = rstd_q (x . EG) + const, var_q = x Qd x^T;  h1 = [a | x] T with T = [P ; G_dec + dec_attn.o.b on the constant's row];
     *                  LayerNorm_mlp(h1) W1^T = rstd ([a | x] TWc) + C, var = [a | x] Qc [a | x]^T: linear1 is a K + 16 product.  Step-invariant parts:
     *     dec_qdd [K, 16 NT] (G_dec - row means)(...)^T / dkv in operand order (as enc_qee)    dec_c   [dkv]      dec_mlp.fc1.b + dec_mlp.fc1.w dec_mlp.norm.b
     *     dec_twx [K, dkv]   Xc (dec_mlp.fc1.w * dec_mlp.norm.w)^T, Xc = centred input rows of T     dec_qxx [K, K]     Xc Xc^T / dkv
     *     gen_qe  [contact_dim, K]  contact_layer.w G_dec^T
     *                  and the parts that depend on the sample's latents come from their decoder keys / values (o = 32 h + r over a head's entries,
     *                  Woc = dec_attn.o.w minus its column means) through
     *     dec_dwq [K, dkv]  = ((G_dec - row means) * dec_q_norm.w) dec_attn.q.w^T      dec_wqb [dkv] = dec_attn.q.w dec_q_norm.b + dec_attn.q.b
     *     dec_wco [8, dkv]  = contact_layer.w dec_attn.o.w (zero rows >= contact_dim)   dec_wow [dkv, dkv] = Woc^T (dec_mlp.fc1.w * dec_mlp.norm.w)^T
     *     dec_wog [dkv, dkv] = Woc^T Woc / dkv                                          dec_xwo [K, dkv] = Xc Woc / dkv */
    const float* gen_qe;
    const float* dec_c; const float* dec_twx; const float* dec_qxx; const float* dec_qdd;
    const float* enc_ec; const float* enc_qee; const float* enc_wove; const float* enc_c1;
    const float* dec_dwq; const float* dec_wqb; const float* dec_wco; const float* dec_wow; const float* dec_wog; const float* dec_xwo;
    /* optional (any sampling form): LayerNorm folded into the six kinds of latent-chain stages that follow one.  lat_fold[3 s + 0 / 1 / 2] =
     * (W * gamma [N, K], g[n] = sum_k (W * gamma)[n, k], b + W beta) for slot s: 0 enc_mlp.fc1; 1 + 4 l .. 3 + 4 l the q / k / v projections of
     * self-attention layer l (self_norm[l]); 4 + 4 l self_mlp[l].fc1; 17 / 18 dec_attn.k / dec_attn.v (dec_kv_norm).  The stage then computes
     * rstd (W' x - mean g) + c on the RAW rows: its matrix products do not wait for the statistics. */
    const float* const* lat_fold;
} afm_cdm_weights;

#define AFM_CDM_TILE_SHIFT     8       /* bits 8..11, measurement: AFM_TUNE_TILE code forced on the linear1 GEMM of the sampling forms (bit-neutral) */
#define AFM_CDM_CHAIN_SIDE     0x4     /* row-less form with sub-batch streams: the 2-latent chain (lat_head .. lat_dectables) of a sub-batch runs on its SIDE stream (fork / join by events), so that it can sit on CUs of its own (a CU-masked stream) under the other sub-batch's point kernels; bit-identical */
#define AFM_CDM_PIPELINE       0x10    /* ABI v7, row-less form with n_sub > 1: the point kernels (enc_point, dec_point) of ALL sub-batches in round-robin order on streams[0], the 2-latent chain of sub-batch s on streams[2 s + 1] between two events - one sub-batch's launch-latency-bound chain always runs under the other sub-batches' point kernels; bit-identical */
#define AFM_CDM_DEC_CHUNKS_SHIFT 12    /* bits 12..17: workgroups per sample of enc_point's successor dec_point_kernel (0 = 16); a tuning knob, bit-identical (a point's arithmetic does not depend on its chunk) */
#define AFM_CDM_CLIP_X0        0x8     /* clip_denoised=True: pred_xstart clamped to [-1, 1] inside the fused DDPM update of every sampling form */
#define AFM_CDM_NO_GEN         0x2     /* measurement: round 2's folded form (step-invariant adapter parts materialised, per-point rows) although the row-less tables are present */

int64_t afm_cdm_workspace_bytes(const afm_cdm_weights* w, int32_t B, int32_t N);

/* One denoiser evaluation (+ optional fused DDPM update, same afm_ddpm_args as the CMDM).
 *   feat [B,N,feat_dim] = cat(x_t, (point features), xyz) as cdm.py:167-171 builds it; x_t itself is feat[..., :contact_dim]
 *   (needed separately, contiguous [B,N,contact_dim], only for the DDPM update); t [B] int64;
 *   text_q0 / text_u / text_cu: the TEXT latent of every sample from afm_cdm_latent_tokens(which = 0) on text_feat [B,text_dim];
 *   x0_out [B,N,contact_dim] (may be NULL when ddpm != NULL). */
int afm_cdm_forward(const afm_cdm_weights* w, const float* feat, const float* x_t, const int64_t* t,
                    const float* text_q0, const float* text_u, const float* text_cu, float* x0_out,
                    const afm_ddpm_args* ddpm, int32_t B, int32_t N, void* workspace, int64_t workspace_bytes, void* stream);
/* Same, with the decoder-adapter GEMM enqueued on `side_stream` underneath the per-sample latent chain (fork / join by events on
 * `stream`; results identical).  side_stream == NULL behaves like afm_cdm_forward. */
int afm_cdm_forward_overlap(const afm_cdm_weights* w, const float* feat, const float* x_t, const int64_t* t,
                    const float* text_q0, const float* text_u, const float* text_cu, float* x0_out,
                    const afm_ddpm_args* ddpm, int32_t B, int32_t N, void* workspace, int64_t workspace_bytes, void* side_stream, void* stream);

/* Whole ADM p_sample_loop (gaussian_diffusion.py:442-536 with the CDM Perceiver denoiser) enqueued natively, no host synchronisation:
 *   x [B,N,contact_dim]: x_T on entry, the sample on exit.  feat [B,N,feat_dim]: the encoder input; its step-invariant columns
 *   (per-point features, xyz) are filled by the caller, the leading contact_dim columns are rewritten from x every step.
 *   text_q0 / text_u / text_cu: afm_cdm_latent_tokens of the text features.  step_noise [n_steps,B,N,contact_dim] or NULL (Philox
 *   keyed by (seed, sample_index0 + b, step)).  Schedule rows / sched_scratch as for afm_cmdm_sample_loop
 *   (afm_cmdm_sched_scratch_bytes).  Sub-batching: n_sub > 1 splits the batch; sub-batch s runs on streams[2s] and uses streams[2s+1]
 *   as the side stream of its decoder-adapter GEMM (one half's per-sample latent chain overlaps the other half's GEMMs);
 *   n_sub <= 1: everything on `stream`, streams[0] (if given) = side stream.  workspace >= afm_cdm_loop_workspace_bytes. */
int64_t afm_cdm_loop_workspace_bytes(const afm_cdm_weights* w, int32_t B, int32_t N, int32_t n_sub);
int afm_cdm_sample_loop(const afm_cdm_weights* w, float* x, float* feat, const float* text_q0, const float* text_u,
                        const float* text_cu, const float* step_noise, const int64_t* d_timestep_map, const float* d_c1,
                        const float* d_c2, const float* d_sigma, int32_t n_steps, uint64_t seed, int64_t sample_index0,
                        int32_t B, int32_t N, void* sched_scratch, void* workspace, int64_t workspace_bytes, int32_t n_sub,
                        void* const* streams, void* stream);
/* Slice of the loop, as afm_cmdm_sample_loop_range. */
int afm_cdm_sample_loop_range(const afm_cdm_weights* w, float* x, float* feat, const float* text_q0, const float* text_u,
                              const float* text_cu, const float* step_noise, const int64_t* d_timestep_map, const float* d_c1,
                              const float* d_c2, const float* d_sigma, int32_t n_steps, int32_t first_step, uint64_t seed,
                              int64_t sample_index0, int32_t B, int32_t N, void* sched_scratch, void* workspace,
                              int64_t workspace_bytes, int32_t n_sub, void* const* streams, void* stream);

/* Latent-token precomputation (step-invariant, off the per-step path): for n input rows `in` [n, text_dim] (which = 0,
 * language_adapter) or [n, time_dim] (which = 1, time_embedding_adapter) compute the latent's enc_q0 row
 * q0_out [n, dq], q = dp_scale * q_proj(LN_q(q0)) folded through k_proj: u_out [n, enc_heads, dkv], cu_out [n, enc_heads]. */
int afm_cdm_latent_tokens(const afm_cdm_weights* w, int32_t which, const float* in, int32_t n,
                          float* q0_out, float* u_out, float* cu_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Opt-in profiler (measurement only, no reference counterpart): when enabled, every kernel launch of
 * this library is bracketed by HIP events on its own stream.  afm_profile_read synchronises, returns
 * per-kernel totals since the last read (work = algorithmic FLOPs, or bytes for streaming kernels) and
 * clears them.  Returns the number of entries written, or a negative AFM_E_* code. */
typedef struct {
    const char* name;       /* kernel name as it appears in rocprofv3 kernel traces (template args abbreviated) */
    int64_t launches;
    double total_ms;
    double total_work;
} afm_profile_entry;
int afm_profile_enable(int32_t on);
int afm_profile_read(afm_profile_entry* out, int32_t max_entries);

#ifdef __cplusplus
}
#endif
#endif /* AFM_HIP_H */
