// Internals shared by the translation units of the CDM / ContactPerceiver denoiser (perceiver.hip: C-ABI, workspace, forward and loop drivers;
// perceiver_rows.hip: the per-point kernels of the forms that read rows; perceiver_points.hip: the row-less form; perceiver_chain.hip: the
// latent chain).  Nothing here is part of the library's interface (include/afm_hip.h).
#pragma once
#include "common.h"
#include "profile.h"

// elementwise.hip: the Philox noise of consecutive steps in one launch, out [nsteps][B][per_sample] (same values as nsteps calls of afm_randn)
__attribute__((visibility("hidden"))) int afm_randn_steps(float* out, int32_t B, int64_t per_sample, uint64_t seed, int64_t sample_index0, int32_t step0, int32_t nsteps,
                                                          void* stream);

namespace afm_cdm {

constexpr int NSPLIT = 16;          // workgroups per sample in enc_reduce (x4 waves = 64 partials per sample)
constexpr int NPART = NSPLIT * 4;
constexpr int MAXD = 512;           // dq upper bound for the latent kernels' LDS vectors
constexpr int TL_TOK = 16, TL_OB = 16;      // toklin_kernel: one wave per 16 tokens x 16 outputs
// per-sample record written by lat_decfold_kernel: G [njh][256] | P [njh][256] | cb [njh] | WP [8][njh] (contact_layer.w . P, folded form)
#define DEC_LAT_STRIDE(njh) (2 * (njh) * 256 + (njh) + 8 * (njh))
#define AFM_TRY(expr) do { int rc__ = (expr); if (rc__ != 0) return rc__; } while (0)

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

// workspace of one denoiser evaluation (perceiver.hip: carve)
struct CdmWs {
    float *enc_kv, *bufB, *h1, *z, *pm, *pl, *pacc, *dec_lat, *s1, *rdot, *qe;
    float *lat_s, *lat_x, *lat_t1, *lat_t2, *lat_qkv, *lat_kv;     // batched latent chain: [2B] token rows
    float *twp, *qtab;                                             // fused decoder: TWc rows of the attention weights [16 B][256], per-sample tables [B][RowLess::TAB]
    int64_t bytes;
};

// AFM_CDM_CHAIN_SIDE: the sub-batch's chain stream and its fork / join events (created once per loop)
struct CdmChainSide { hipStream_t chain; hipEvent_t forked, joined; };

#pragma GCC visibility push(hidden)
// perceiver_rows.hip - per-point kernels that read rows: mode 0 = rows from memory (layer-by-layer form), 1 = FOLD (step-invariant part
// materialised once per loop + contact columns)
int launch_enc_reduce(const afm_cdm_weights& w, const float* rows, const float* text_u, const float* text_cu, const int64_t* t, int B, int N,
                      const CdmWs& ws, const float* x_t, int mode, hipStream_t s);
int launch_dec_attend(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, int mode, hipStream_t s);
int launch_cdm_output(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, float* x0_out, const afm_ddpm_args* ddpm, hipStream_t s);
int launch_lat_combine(const afm_cdm_weights& w, const float* text_q0, const int64_t* t, const CdmWs& ws, int B, hipStream_t s);
int launch_lat_decfold(const afm_cdm_weights& w, const CdmWs& ws, int B, hipStream_t s);
// perceiver_points.hip - the row-less form (mode 3: rowless_nks(feat_dim) != 0 and every fused table present)
int launch_enc_point(const afm_cdm_weights& w, const float* text_u, const float* text_cu, const int64_t* t, int B, int N, const CdmWs& ws,
                     const float* x_t, const float* feat, hipStream_t s);
int launch_lat_head(const afm_cdm_weights& w, const float* text_q0, const int64_t* t, const CdmWs& ws, int B, hipStream_t s);
int launch_dec_point(const afm_cdm_weights& w, int B, int N, const CdmWs& ws, const float* x_t, const float* feat, float* x0_out,
                     const afm_ddpm_args* ddpm, hipStream_t s, bool with_tables = true);
int launch_dec_tables(const afm_cdm_weights& w, int B, const CdmWs& ws, hipStream_t s);      // lat_dectables_kernel alone (chain stream)
// perceiver_chain.hip - enc_reduce / enc_point partials -> the decoder's view of the two latents (dec_lat records / lat_kv for the row-less form)
int cdm_latent_chain(const afm_cdm_weights& w, const float* text_q0, const int64_t* t, const CdmWs& ws, int B, hipStream_t s, bool rowless);
#pragma GCC visibility pop

}  // namespace afm_cdm
