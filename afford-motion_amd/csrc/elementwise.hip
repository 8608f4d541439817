// LayerNorm, DDPM posterior update, Philox normal generator.
// All three are HBM/L2-streaming kernels: float4 accesses, one wave per LayerNorm row.
#include "common.h"
#include "profile.h"

namespace {

// one wave per row; the row lives in registers (dim <= 64 * 4 * MAXV)
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        int64_t rows, int dim, float eps, int grp, int stride, int off) {
    const int lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    if (grp) row = (row / grp) * stride + off + row % grp;          // logical row -> strided token row
    layernorm_row<MAXV>(x + row * dim, gamma, beta, y + row * dim, dim, eps, lane);
}

// any dim (not a multiple of 4, e.g. the 646-wide input LayerNorm of the CDM 'MLP' arch, cdm.py:18-23): wave per row,
// lane-strided scalar accesses, three passes over the (L1/L2-resident) row
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y, int64_t rows, int dim,
                                                                float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xp = x + row * dim;
    float sum = 0.f;
    for (int c = lane; c < dim; c += 64) sum += xp[c];
    const float mean = wave_sum(sum) / (float)dim;
    float sq = 0.f;
    for (int c = lane; c < dim; c += 64) { const float d = xp[c] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)dim + eps);
    float* yp = y + row * dim;
    for (int c = lane; c < dim; c += 64) yp[c] = (xp[c] - mean) * rstd * gamma[c] + beta[c];
}

// x_next = (c1*x0 + c2*x_t) + sigma*noise with every product and sum individually rounded
// (bit-identical to the reference's float32 torch expression).
__global__ __launch_bounds__(256) void ddpm_step_kernel(const float* __restrict__ x0, const float* __restrict__ xt,
                                                        const float* __restrict__ noise, float* __restrict__ xn,
                                                        const float* __restrict__ c1, const float* __restrict__ c2,
                                                        const float* __restrict__ sigma, int64_t per_sample, uint64_t seed,
                                                        int64_t sample0, int step) {
#pragma clang fp contract(off)
    const int b = blockIdx.y;
    const float a1 = c1[b], a2 = c2[b], sg = sigma[b];
    const int64_t base = (int64_t)b * per_sample;
    const int64_t nquad = (per_sample + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x) {
        float z[4];
        if (!noise) philox_normal4(seed, sample0 + b, step, (uint64_t)q, z);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = q * 4 + e;
            if (i < per_sample) {
                const float nz = noise ? noise[base + i] : z[e];
                const float m1 = a1 * x0[base + i];
                const float m2 = a2 * xt[base + i];
                const float mean = m1 + m2;
                const float sn = sg * nz;
                xn[base + i] = mean + sn;
            }
        }
    }
}

// grid (x, B, steps): blockIdx.z = a further step of the same keying, its [B][per_sample] block behind the previous one
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, int64_t per_sample, uint64_t seed,
                                                    int64_t sample0, int step) {
    const int b = blockIdx.y;
    out += (int64_t)blockIdx.z * gridDim.y * per_sample;
    step += (int)blockIdx.z;
    const int64_t base = (int64_t)b * per_sample;
    const int64_t nquad = (per_sample + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nquad; q += (int64_t)gridDim.x * blockDim.x) {
        float z[4];
        philox_normal4(seed, sample0 + b, step, (uint64_t)q, z);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (q * 4 + e < per_sample) out[base + q * 4 + e] = z[e];
    }
}

// per-sample masked MSE: sum((a-b)^2 * keep) / (sum(keep) * D), one 1024-thread workgroup per sample: a thread walks
// whole frames (row l, then its D contiguous values are spread over the 16 lanes of its group -> coalesced, no division
// per element), fixed-order tree at the end
__global__ __launch_bounds__(1024) void masked_mse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          const uint8_t* __restrict__ mask, float* __restrict__ out, int L, int D) {
    __shared__ float red[2][16];
    const int s = blockIdx.x;
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;       // 64 groups of 16 lanes, one frame per group at a time
    float acc = 0.f, cnt = 0.f;
    for (int l = grp; l < L; l += 64) {
        if (mask && mask[(int64_t)s * L + l]) continue;
        const float* ap = a + ((int64_t)s * L + l) * D;
        const float* bp = b + ((int64_t)s * L + l) * D;
        for (int c = sub; c < D; c += 16) { const float d = ap[c] - bp[c]; acc += d * d; }
        if (sub == 0) cnt += (float)D;
    }
    acc = wave_sum(acc); cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = acc; red[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sa = 0.f, sc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sa += red[0][i]; sc += red[1][i]; }
        out[s] = sa / sc;            // sc == sum(keep) * D
    }
}

// eval-mode BatchNorm folded into y = x * scale + shift (optionally around a preceding nn.Linear's bias: BN(W x + lb) = scale W x + (lb scale + shift)):
// scale = w / sqrt(var + eps), shift = b - mean * scale.  Individually rounded operations (the torch expression, bit for bit).
__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ mean,
                                                      const float* __restrict__ var, float eps, const float* __restrict__ lin_bias,
                                                      float* __restrict__ scale, float* __restrict__ shift, int C) {
#pragma clang fp contract(off)
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float sc = w[c] / sqrtf(var[c] + eps);
    const float ms = mean[c] * sc;
    float sh = b[c] - ms;
    if (lin_bias) { const float lb = lin_bias[c] * sc; sh = lb + sh; }
    scale[c] = sc;
    shift[c] = sh;
}

// ADM -> AMDM hand-off in HBM (the reference goes through .npy files): contact = clip(sample * std + mean, 1e-20, 1)
// (datasets/humanml3d.py:494-511), dist = sqrt(-2 ln(contact) sigma^2) (utils/evaluate.py:56-66), condition = exp(-dist^2 / (2 sigma^2))
// (datasets/humanml3d.py:773-774) - the reference's chain of float32 operations, one element per thread.
__global__ __launch_bounds__(256) void contact_glue_kernel(const float* __restrict__ sample, float* __restrict__ out, int64_t n, float sigma2, float mean, float std) {
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float c = sample[i] * std;
        c = c + mean;
        c = fminf(fmaxf(c, 1e-20f), 1.0f);
        const float l = -2.0f * logf(c);
        const float d = sqrtf(l * sigma2);
        const float e = -0.5f * (d * d);
        out[i] = expf(e / sigma2);
    }
}

}  // namespace

namespace {
__global__ void clamp_kernel(float* __restrict__ x, int64_t n, float lo, float hi) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        x[i] = v < lo ? lo : (v > hi ? hi : v);
    }
}
}  // namespace

extern "C" int afm_clamp(float* x, int64_t n, float lo, float hi, void* stream) {
    if (n == 0) return 0;
    if (!x || n < 0 || !(lo <= hi)) return AFM_E_BADARG;
    int64_t g = (n + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, n, lo, hi);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_bn_fold(const float* w, const float* b, const float* mean, const float* var, float eps, const float* lin_bias, float* scale,
                           float* shift, int32_t C, void* stream) {
    if (C == 0) return 0;
    if (!w || !b || !mean || !var || !scale || !shift || C < 0 || !(eps >= 0.0f)) return AFM_E_BADARG;
    hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, b, mean, var, eps, lin_bias, scale, shift, C);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_contact_glue(const float* sample, float* out, int64_t n, float sigma_sq, float mean, float std, void* stream) {
    if (n == 0) return 0;
    if (!sample || !out || n < 0 || !(sigma_sq > 0.0f)) return AFM_E_BADARG;
    int64_t g = (n + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(contact_glue_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, sample, out, n, sigma_sq, mean, std);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_masked_mse(const float* target, const float* pred, const uint8_t* frame_mask, float* out, int32_t B,
                              int32_t L, int32_t D, void* stream) {
    if (!target || !pred || !out || B < 0 || L <= 0 || D <= 0) return AFM_E_BADARG;
    if (B == 0) return 0;
    hipLaunchKernelGGL(masked_mse_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, target, pred, frame_mask, out, L, D);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_layernorm_rows(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t dim,
                                  float eps, int32_t grp, int32_t stride, int32_t off, void* stream);

extern "C" int afm_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t dim,
                             float eps, void* stream) {
    return afm_layernorm_rows(x, gamma, beta, y, rows, dim, eps, 0, 0, 0, stream);
}

extern "C" int afm_layernorm_rows(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t dim,
                                  float eps, int32_t grp, int32_t stride, int32_t off, void* stream) {
    if (dim <= 0) return AFM_E_BADARG;
    if (rows == 0) return 0;                                  // empty batch (pointers may be null)
    if (!x || !gamma || !beta || !y || rows < 0) return AFM_E_BADARG;
    if ((dim & 3) || ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15)) {
        if (grp) return AFM_E_UNSUPPORTED;                     // the row-subset form is only used with d_model (multiple of 4)
        AfmProf prof(AFM_PROF_LN, 8.0 * rows * dim, (hipStream_t)stream);
        hipLaunchKernelGGL(layernorm_generic_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, rows, dim, eps);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    if (rows == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(256), grid((unsigned)((rows + 3) / 4));
    AfmProf prof(AFM_PROF_LN, 8.0 * rows * dim, s);
    if (dim <= 256) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, gamma, beta, y, rows, dim, eps, grp, stride, off);
    else if (dim <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, gamma, beta, y, rows, dim, eps, grp, stride, off);
    else if (dim <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, gamma, beta, y, rows, dim, eps, grp, stride, off);
    else if (dim <= 2048) hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, gamma, beta, y, rows, dim, eps, grp, stride, off);
    else return AFM_E_UNSUPPORTED;
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_ddpm_step(const float* x0, const float* x_t, const float* noise, float* x_next, const float* c1,
                             const float* c2, const float* sigma, int32_t B, int64_t per_sample, uint64_t seed,
                             int64_t sample_index0, int32_t step, void* stream) {
    if (!x0 || !x_t || !x_next || !c1 || !c2 || !sigma || B < 0 || per_sample <= 0) return AFM_E_BADARG;
    if (B == 0) return 0;
    const int64_t nquad = (per_sample + 3) >> 2;
    unsigned gx = (unsigned)((nquad + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(ddpm_step_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, x0, x_t, noise, x_next, c1, c2, sigma,
                       per_sample, seed, sample_index0, step);
    AFM_CHECK_LAUNCH();
    return 0;
}

// the noise of `nsteps` consecutive steps in one launch: out [nsteps][B][per_sample], step index step0 + i (same values as nsteps calls of afm_randn)
__attribute__((visibility("hidden"))) int afm_randn_steps(float* out, int32_t B, int64_t per_sample, uint64_t seed, int64_t sample_index0, int32_t step0, int32_t nsteps, void* stream) {
    if (!out || B < 0 || per_sample <= 0 || nsteps < 0 || nsteps > 65535) return AFM_E_BADARG;
    if (B == 0 || nsteps == 0) return 0;
    const int64_t nquad = (per_sample + 3) >> 2;
    unsigned gx = (unsigned)((nquad + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(randn_kernel, dim3(gx, B, nsteps), dim3(256), 0, (hipStream_t)stream, out, per_sample, seed, sample_index0, step0);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_randn(float* out, int32_t B, int64_t per_sample, uint64_t seed, int64_t sample_index0, int32_t step,
                         void* stream) {
    if (!out || B < 0 || per_sample <= 0) return AFM_E_BADARG;
    if (B == 0) return 0;
    const int64_t nquad = (per_sample + 3) >> 2;
    unsigned gx = (unsigned)((nquad + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(randn_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, out, per_sample, seed, sample_index0, step);
    AFM_CHECK_LAUNCH();
    return 0;
}
