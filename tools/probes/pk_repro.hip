// Round 4 - a stand-alone reproducer (no Python, no torch, no model) of the "wrong only while a second stream has kernels in flight"
// defect class of rounds 2 / 3 (profiles/r02_decfold_nondeterminism.md, profiles/r03_packed_f32_defect.md): packed-f32 VALU
// instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) deliver wrong values in lanes 48..63 of single waves.
//
//   mode `lib`   the library's own statistics-carrying GEMM chain (out_proj + stats -> linear1 with folded norm1 -> linear2 + LayerNorm(raw)
//                residual + stats -> in_proj with folded norm2) through the C-ABI, on two streams with buffers of their own; every buffer is
//                compared bit for bit with the same chain run alone.  Link it against a library built WITH packed-f32 ops
//                (tools/probes/build_packed.sh -> tools/probes/packed/libafm_hip.so) to see the defect, against the shipped (fenced) library to
//                see it gone.
//   mode `mini`  ONE synthetic kernel in this file that does what the failing epilogue does - a 64 x 64 f32 tile staged in LDS, per-row
//                (mean, rstd) from LDS, `(r - mean) * rstd * gamma + beta` on float4 rows as PACKED f32 arithmetic (ext_vector float2 fma:
//                hipcc emits v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, checked in the disassembly), 16-lane butterflies for the output
//                statistics - and a co-runner on the second stream, swept over {the same kernel on other buffers, a bf16-MFMA burner, a
//                VALU burner, a memory streamer, nothing}.  Prints which lanes / columns mismatch.
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude tools/probes/pk_repro.hip -ldl -o tools/probes/pk_repro        (tools/probes/build_packed.sh)
//   tools/probes/pk_repro lib <path to a libafm_hip.so> [iterations [co-runner]]      co-runner: chain (default) | mfma | valu | mem | none
//   tools/probes/pk_repro mini [iterations]
//   tools/probes/pk_repro opsel [repetitions [libafm_hip.so]]     the bare instruction forms in a loop next to simple co-runners (and the GEMM chain)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "afm_hip.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define AK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d\n", #x, r_); exit(2); } } while (0)
// the library under test is opened at run time (several builds are compared in one call)
static int (*p_afm_linear)(const afm_linear_args*, void*) = nullptr;
static int (*p_afm_version)(void) = nullptr;
#define afm_linear p_afm_linear
#define afm_version p_afm_version

static unsigned g_seed = 12345u;
static float frand() { g_seed = g_seed * 1664525u + 1013904223u; return ((g_seed >> 8) * (1.0f / 8388608.0f) - 1.0f); }
static float* dev_rand(size_t n, float scale, float shift = 0.f) {
    std::vector<float> h(n);
    for (auto& v : h) v = frand() * scale + shift;
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}
static float* dev_empty(size_t n) { float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0xFF, n * 4)); return d; }

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma_burner(float* sink, int iters) {          // bf16 matrix pipe, nothing else
    f16v acc = {};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (acc[0] == 12345.678f) sink[threadIdx.x] = acc[1];
}
__global__ __launch_bounds__(256) void valu_burner(float* sink, int iters) {
    float x = threadIdx.x * 0.001f, y = 1.0f;
    for (int it = 0; it < iters; ++it) { x = fmaf(x, 1.0000001f, 0.5f); y = fmaf(y, 0.9999999f, x); }
    if (y == 12345.678f) sink[threadIdx.x] = x;
}
__global__ __launch_bounds__(256) void mem_streamer(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// ------------------------------------------------------------------------------------------------ mode lib
struct Weights { float *wo, *bo, *g1, *b1, *w1g, *gsum, *c1f, *w2, *c2, *wing, *ginsum, *cinf; };
struct Bufs { float *st1, *st2, *t1, *h, *t2, *qkv; };
static const int M = 5216, D = 512, FF = 1024;

static void chain(const Weights& w, const float* att, const float* xin, const Bufs& b, hipStream_t s) {
    afm_linear_args a;
    memset(&a, 0, sizeof a); a.A = att; a.lda = D; a.W = w.wo; a.ldw = D; a.C = b.t1; a.ldc = D; a.M = M; a.N = D; a.K = D; a.bias = w.bo;
    a.residual = xin; a.ldr = D; a.stat_out = b.st1; a.ln_eps2 = 1e-5f; AK(afm_linear(&a, s));
    memset(&a, 0, sizeof a); a.A = b.t1; a.lda = D; a.W = w.w1g; a.ldw = D; a.C = b.h; a.ldc = FF; a.M = M; a.N = FF; a.K = D; a.bias = w.c1f; a.act = AFM_ACT_GELU;
    a.a_stat = b.st1; a.a_stat_groups = D / 64; a.a_fold_g = w.gsum; a.ln_eps2 = 1e-5f; AK(afm_linear(&a, s));
    memset(&a, 0, sizeof a); a.A = b.h; a.lda = FF; a.W = w.w2; a.ldw = FF; a.C = b.t2; a.ldc = D; a.M = M; a.N = D; a.K = FF; a.bias = w.c2;
    a.residual = b.t1; a.ldr = D; a.res_stat = b.st1; a.res_gamma = w.g1; a.res_beta = w.b1; a.stat_out = b.st2; a.ln_eps2 = 1e-5f; AK(afm_linear(&a, s));
    memset(&a, 0, sizeof a); a.A = b.t2; a.lda = D; a.W = w.wing; a.ldw = D; a.C = b.qkv; a.ldc = 3 * D; a.M = M; a.N = 3 * D; a.K = D; a.bias = w.cinf;
    a.a_stat = b.st2; a.a_stat_groups = D / 64; a.a_fold_g = w.ginsum; a.ln_eps2 = 1e-5f; AK(afm_linear(&a, s));
}
static Bufs mk() { return Bufs{dev_empty((size_t)M * 16), dev_empty((size_t)M * 16), dev_empty((size_t)M * D), dev_empty((size_t)M * FF), dev_empty((size_t)M * D), dev_empty((size_t)M * 3 * D)}; }
static std::vector<float> down(const float* d, size_t n) { std::vector<float> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost)); return h; }

static int mode_lib(const char* libpath, int iters, const char* co) {
    void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen %s: %s\n", libpath, dlerror()); return 2; }
    p_afm_linear = (int (*)(const afm_linear_args*, void*))dlsym(h, "afm_linear");
    p_afm_version = (int (*)(void))dlsym(h, "afm_version");
    if (!p_afm_linear || !p_afm_version) { fprintf(stderr, "symbols missing in %s\n", libpath); return 2; }
    printf("== %s (ABI %d), second stream runs: %s\n", libpath, afm_version(), co);
    float* sink = dev_empty(4096);
    f4 *big_in = (f4*)dev_rand((size_t)16 << 20, 1.f), *big_out = (f4*)dev_empty((size_t)16 << 20);
    Weights w{dev_rand((size_t)D * D, 0.044f), dev_rand(D, 0.1f), dev_rand(D, 0.2f, 1.f), dev_rand(D, 0.1f), dev_rand((size_t)FF * D, 0.044f), dev_rand(FF, 0.5f),
              dev_rand(FF, 0.1f), dev_rand((size_t)D * FF, 0.031f), dev_rand(D, 0.1f), dev_rand((size_t)3 * D * D, 0.044f), dev_rand(3 * D, 0.5f), dev_rand(3 * D, 0.1f)};
    hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
    const float* att[2] = {dev_rand((size_t)M * D, 1.f), dev_rand((size_t)M * D, 1.f)};
    const float* xin[2] = {dev_rand((size_t)M * D, 2.f, 0.7f), dev_rand((size_t)M * D, 2.f, 0.7f)};
    const char* names[6] = {"stat1", "stat2", "t1", "h", "t2", "qkv"};
    const size_t sizes[6] = {(size_t)M * 16, (size_t)M * 16, (size_t)M * D, (size_t)M * FF, (size_t)M * D, (size_t)M * 3 * D};
    const int widths[6] = {16, 16, D, FF, D, 3 * D};
    std::vector<std::vector<float>> ref[2];
    for (int s = 0; s < 2; ++s) {                // the chain ALONE: one stream, nothing else on the chip
        Bufs b = mk(); chain(w, att[s], xin[s], b, st[0]); CK(hipStreamSynchronize(st[0]));
        float* p[6] = {b.st1, b.st2, b.t1, b.h, b.t2, b.qkv};
        for (int i = 0; i < 6; ++i) ref[s].push_back(down(p[i], sizes[i]));
    }
    Bufs bufs[2] = {mk(), mk()};
    long bad_total = 0; int bad_iters = 0;
    std::map<std::string, long> lane_hist;
    for (int it = 0; it < iters; ++it) {
        for (int rep = 0; rep < 6; ++rep) {
            chain(w, att[0], xin[0], bufs[0], st[0]);
            if (!strcmp(co, "chain")) chain(w, att[1], xin[1], bufs[1], st[1]);                                         // both streams run the chain
            else if (!strcmp(co, "mfma")) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(mfma_burner, dim3(1024), dim3(256), 0, st[1], sink, 1500);
            else if (!strcmp(co, "valu")) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(valu_burner, dim3(1024), dim3(256), 0, st[1], sink, 8000);
            else if (!strcmp(co, "mem")) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(mem_streamer, dim3(2048), dim3(256), 0, st[1], big_in, big_out, (size_t)4 << 20);      // 16 M floats = 4 M float4
        }
        CK(hipDeviceSynchronize());
        bool bad_it = false;
        for (int s = 0; s < (strcmp(co, "chain") ? 1 : 2); ++s) {
            float* p[6] = {bufs[s].st1, bufs[s].st2, bufs[s].t1, bufs[s].h, bufs[s].t2, bufs[s].qkv};
            for (int i = 0; i < 6; ++i) {
                std::vector<float> got = down(p[i], sizes[i]);
                long nb = 0;
                for (size_t e = 0; e < sizes[i]; ++e) if (memcmp(&got[e], &ref[s][i][e], 4)) {
                    ++nb;
                    if (i >= 2) {                                        // data tensors: which quarter of the wave, which column parity
                        const long row = e / widths[i], col = e % widths[i];
                        char key[64]; snprintf(key, sizeof key, "%s row%%4=%ld col%%2=%ld", names[i], row % 4, col % 2);
                        lane_hist[key]++;
                    }
                }
                if (nb) { bad_it = true; bad_total += nb; if (bad_iters < 3) printf("iteration %d stream %d %s: %ld elements differ\n", it, s, names[i], nb); }
            }
        }
        bad_iters += bad_it;
    }
    printf("   chain on stream 0, `%s` on stream 1: %d of %d iterations with mismatches, %ld elements in total\n", co, bad_iters, iters, bad_total);
    for (auto& kv : lane_hist) printf("   %-28s %ld\n", kv.first.c_str(), kv.second);
    if (!bad_iters) printf("   mismatches: none\n");
    return 0;
}

// ------------------------------------------------------------------------------------------------ mode mini
__device__ __forceinline__ float lane_xor_f(float v, int m) { return __shfl_xor(v, m, 64); }

// the failing epilogue, alone: tile [64][64] of "accumulators" read from global into LDS (the library stages its MFMA accumulators there),
// per-row (mean, rstd) combined from 8 group records into LDS, then four trips of: v = tile row quad; t = residual quad;
// t = (t - mean) * rstd * gamma + beta as packed f32; v += bias + t; store; 16-lane statistics of v.
template <bool PACKED>
__global__ __launch_bounds__(256) void victim_kernel(const float* __restrict__ acc, const float* __restrict__ residual, const float* __restrict__ stat_in,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ bias,
                                                     float* __restrict__ out, float* __restrict__ stat_out, int Mrows, int N) {
    __shared__ __attribute__((aligned(16))) float tile[64 * 68];
    __shared__ float rowst[128];
    const int tid = threadIdx.x, nbn = N / 64, bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    for (int e = tid; e < 64 * 16; e += 256) {
        const int row = e / 16, cq = (e % 16) * 4, grow = min(bm * 64 + row, Mrows - 1);
        *reinterpret_cast<f4*>(tile + row * 68 + cq) = *reinterpret_cast<const f4*>(acc + (size_t)grow * N + bn * 64 + cq);
    }
    if (tid < 64) {
        const int grow = min(bm * 64 + tid, Mrows - 1), groups = N / 64;
        const float* st = stat_in + (size_t)grow * groups * 2;
        float ms = 0.f;
        for (int t = 0; t < groups; ++t) ms += st[2 * t];
        const float mean = ms / groups;
        float m2 = 0.f;
        for (int t = 0; t < groups; ++t) { const float d = st[2 * t] - mean; m2 += st[2 * t + 1] + 64.0f * (d * d); }
        rowst[2 * tid] = mean; rowst[2 * tid + 1] = 1.0f / sqrtf(m2 / (64.0f * groups) + 1e-5f);
    }
    __syncthreads();
    for (int e = tid; e < 64 * 16; e += 256) {
        const int row = e / 16, cq = (e % 16) * 4, grow = bm * 64 + row, gcol = bn * 64 + cq;
        const bool valid = grow < Mrows;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            v = *reinterpret_cast<const f4*>(tile + row * 68 + cq);
            const f4 t = *reinterpret_cast<const f4*>(residual + (size_t)grow * N + gcol);
            const f4 g = *reinterpret_cast<const f4*>(gamma + gcol), b = *reinterpret_cast<const f4*>(beta + gcol), bi = *reinterpret_cast<const f4*>(bias + gcol);
            const float mu = rowst[2 * row], rs = rowst[2 * row + 1];
            if (PACKED) {
                const f2 mu2 = {mu, mu}, rs2 = {rs, rs};
                f2 lo = {t.x, t.y}, hi = {t.z, t.w};
                lo = __builtin_elementwise_fma((lo - mu2) * rs2, f2{g.x, g.y}, f2{b.x, b.y});
                hi = __builtin_elementwise_fma((hi - mu2) * rs2, f2{g.z, g.w}, f2{b.z, b.w});
                const f2 vl = f2{v.x, v.y} + f2{bi.x, bi.y} + lo, vh = f2{v.z, v.w} + f2{bi.z, bi.w} + hi;
                v = f4{vl.x, vl.y, vh.x, vh.y};
            } else {
                v.x = (v.x + bi.x) + fmaf((t.x - mu) * rs, g.x, b.x); v.y = (v.y + bi.y) + fmaf((t.y - mu) * rs, g.y, b.y);
                v.z = (v.z + bi.z) + fmaf((t.z - mu) * rs, g.z, b.z); v.w = (v.w + bi.w) + fmaf((t.w - mu) * rs, g.w, b.w);
            }
            *reinterpret_cast<f4*>(out + (size_t)grow * N + gcol) = v;
        }
        float sm = (v.x + v.y) + (v.z + v.w);
        sm += lane_xor_f(sm, 1); sm += lane_xor_f(sm, 2); sm += lane_xor_f(sm, 4); sm += lane_xor_f(sm, 8);
        const float mu_o = sm * (1.0f / 64.0f);
        const float a = v.x - mu_o, b2 = v.y - mu_o, c = v.z - mu_o, d = v.w - mu_o;
        float m2 = (a * a + b2 * b2) + (c * c + d * d);
        m2 += lane_xor_f(m2, 1); m2 += lane_xor_f(m2, 2); m2 += lane_xor_f(m2, 4); m2 += lane_xor_f(m2, 8);
        if (valid && (cq & 63) == 0) { stat_out[((size_t)grow * nbn + bn) * 2] = mu_o; stat_out[((size_t)grow * nbn + bn) * 2 + 1] = m2; }
    }
}

static int mode_mini(int iters) {
    const int Mr = 5216, N = 512, nb = ((Mr + 63) / 64) * (N / 64);
    hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
    struct Set { float *acc, *res, *stat, *out, *so; } s[2];
    for (auto& x : s) x = Set{dev_rand((size_t)Mr * N, 1.f), dev_rand((size_t)Mr * N, 2.f, 0.7f), dev_rand((size_t)Mr * 16, 0.3f, 0.5f), dev_empty((size_t)Mr * N), dev_empty((size_t)Mr * 16)};
    float *gamma = dev_rand(N, 0.2f, 1.f), *beta = dev_rand(N, 0.1f), *bias = dev_rand(N, 0.1f), *sink = dev_empty(4096);
    f4 *big_in = (f4*)dev_rand((size_t)16 << 20, 1.f), *big_out = (f4*)dev_empty((size_t)16 << 20);
    auto victim = [&](int which, bool packed, hipStream_t q) {
        if (packed) hipLaunchKernelGGL(victim_kernel<true>, dim3(nb), dim3(256), 0, q, s[which].acc, s[which].res, s[which].stat, gamma, beta, bias, s[which].out, s[which].so, Mr, N);
        else hipLaunchKernelGGL(victim_kernel<false>, dim3(nb), dim3(256), 0, q, s[which].acc, s[which].res, s[which].stat, gamma, beta, bias, s[which].out, s[which].so, Mr, N);
    };
    const char* co_names[5] = {"the same kernel on other buffers", "bf16 MFMA burner", "VALU burner", "memory streamer", "nothing"};
    for (int packed = 1; packed >= 0; --packed) {
        victim(0, packed, st[0]); CK(hipDeviceSynchronize());
        const std::vector<float> ref = down(s[0].out, (size_t)Mr * N), ref_so = down(s[0].so, (size_t)Mr * 16);
        for (int co = 0; co < 5; ++co) {
            long bad = 0; int bad_iters = 0;
            std::map<std::string, long> hist;
            for (int it = 0; it < iters; ++it) {
                CK(hipMemsetAsync(s[0].out, 0xFF, (size_t)Mr * N * 4, st[0]));
                for (int rep = 0; rep < 8; ++rep) {
                    victim(0, packed, st[0]);
                    switch (co) {
                        case 0: victim(1, packed, st[1]); break;
                        case 1: hipLaunchKernelGGL(mfma_burner, dim3(1024), dim3(256), 0, st[1], sink, 600); break;
                        case 2: hipLaunchKernelGGL(valu_burner, dim3(1024), dim3(256), 0, st[1], sink, 3000); break;
                        case 3: hipLaunchKernelGGL(mem_streamer, dim3(2048), dim3(256), 0, st[1], big_in, big_out, (size_t)4 << 20); break;
                        default: break;
                    }
                }
                CK(hipDeviceSynchronize());
                const std::vector<float> got = down(s[0].out, (size_t)Mr * N);
                long nbad = 0;
                for (size_t e = 0; e < got.size(); ++e) if (memcmp(&got[e], &ref[e], 4)) {
                    ++nbad;
                    const long row = e / N, col = e % N;
                    char key[64]; snprintf(key, sizeof key, "row%%4=%ld (lanes %ld..%ld) col%%2=%ld", row % 4, 16 * (row % 4), 16 * (row % 4) + 15, col % 2);
                    hist[key]++;
                }
                const std::vector<float> gso = down(s[0].so, (size_t)Mr * 16);
                for (size_t e = 0; e < gso.size(); ++e) if (memcmp(&gso[e], &ref_so[e], 4)) ++nbad;
                bad += nbad; bad_iters += nbad != 0;
            }
            printf("mini %-9s arithmetic, co-runner = %-34s: %3d of %d iterations with mismatches, %ld elements\n", packed ? "PACKED" : "scalar", co_names[co], bad_iters, iters, bad);
            for (auto& kv : hist) printf("      %-40s %ld\n", kv.first.c_str(), kv.second);
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ mode opsel
// The instruction form that fails in the library (profiles/r04_packed_f32_repro.md), alone: a wave multiplies a register pair by the HIGH
// half of a second pair - `v_pk_mul_f32 d, a, p op_sel:[0,1]` - in a loop and checks every result against the two scalar products; the
// second stream runs a kernel that does nothing but packed multiplies of ANOTHER operand selection (or a scalar-only loop, or nothing).
// FORM 0: v_pk_mul_f32 d, a, p op_sel:[0,1]   (src1's high half feeds the low result: the failing form)
// FORM 1: v_pk_mul_f32 d, p, a op_sel:[1,0]   (the commuted form that passes in the library)
// FORM 2: v_pk_mul_f32 d, a, pp               (no op_sel: the factor replicated)
template <int FORM>
__global__ __launch_bounds__(256) void opsel_victim(unsigned* __restrict__ bad_by_lane, unsigned* __restrict__ bad_lo_hi, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned bad = 0, bad_lo = 0, bad_hi = 0;
    f2 a = {1.0f + 0.001f * threadIdx.x, 2.0f + 0.003f * threadIdx.x};
    const f2 p = {3.0f + 0.01f * lane, 0.5f + 0.001f * lane}, pp = {p.y, p.y};
    for (int it = 0; it < iters; ++it) {
        f2 d;
        if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(p));
        else if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(p));
        else asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(pp));
        const float e0 = __fmul_rn(a.x, p.y), e1 = __fmul_rn(a.y, p.y);
        if (__float_as_uint(d.x) != __float_as_uint(e0)) { ++bad; ++bad_lo; }
        if (__float_as_uint(d.y) != __float_as_uint(e1)) { ++bad; ++bad_hi; }
        a.x = a.x * 1.000001f + 1e-3f; a.y = a.y * 0.999999f + 2e-3f;       // new operands every trip (scalar ops)
        if (a.x > 1e6f) a.x = 1.0f;
    }
    if (bad) { atomicAdd(&bad_by_lane[lane], bad); atomicAdd(&bad_lo_hi[0], bad_lo); atomicAdd(&bad_lo_hi[1], bad_hi); }
}
// co-runners: packed multiplies with NO op_sel (0), with op_sel:[1,0] (1), with op_sel:[0,1] like the victim (2), scalar multiplies only (3)
template <int KIND>
__global__ __launch_bounds__(256) void opsel_corunner(float* sink, int iters) {
    f2 u = {1.0f + 0.001f * threadIdx.x, 1.0f}, w = {1.0000001f, 0.9999999f};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(u) : "v"(w));
        else if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[1,0]" : "+v"(u) : "v"(w));
        else if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]" : "+v"(u) : "v"(w));
        else { u.x *= w.x; u.y *= w.y; asm volatile("" : "+v"(u)); }
    }
    if (u.x == 12345.678f) sink[threadIdx.x] = u.y;
}

// synthetic co-runners, one ingredient of a GEMM kernel each (mode `opsel2`): which of them makes the victim's op_sel:[0,1] form fail?
//  0 DPP adds (quad_perm / row_ror / row_mirror: the statistics butterflies)   1 v_permlane32_swap / v_permlane16_swap
//  2 LDS write + read + s_barrier                                              3 ds_bpermute (__shfl)
//  4 global load + store stream                                                5 bf16 MFMA + v_accvgpr moves
//  6 v_cvt_pk_bf16_f32 + shifts (the operand split)                            7 v_readlane / v_readfirstlane / SALU
//  8 ds_read_b128 + bf16 MFMA fed from LDS (a K loop without the split)        9 many VGPRs live (a 200-register wave) + scalar FMAs
template <int KIND>
__global__ __launch_bounds__(256) void synth_corunner(float* sink, const f4* __restrict__ gin, f4* __restrict__ gout, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 68 * 2];
    const int tid = threadIdx.x, lane = tid & 63;
    float x = 1.0f + 0.001f * tid, y = 0.5f;
    if (KIND == 0) {
        for (int it = 0; it < iters; ++it) {
            x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));        // quad_perm:[1,0,3,2]
            x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x128, 0xF, 0xF, true));       // row_ror:8
            x += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));       // row_mirror
            x = x * 0.25f + 0.1f;
        }
    } else if (KIND == 1) {
        unsigned a = __float_as_uint(x), b = __float_as_uint(y);
        for (int it = 0; it < iters; ++it) {
            asm volatile("v_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
            a += 3; b ^= a;
        }
        x = __uint_as_float(a & 0x3fffffffu);
    } else if (KIND == 2) {
        for (int it = 0; it < iters; ++it) {
            *reinterpret_cast<f4*>(lds + (tid * 4) % (64 * 68)) = f4{x, y, x, y};
            __syncthreads();
            const f4 v = *reinterpret_cast<const f4*>(lds + ((tid * 4 + 1024) % (64 * 64)));
            x = v.x * 0.5f + 0.25f; y = v.w * 0.5f + 0.1f;
            __syncthreads();
        }
    } else if (KIND == 3) {
        for (int it = 0; it < iters; ++it) { x += __shfl_xor(x, 17, 64); x = x * 0.25f + 0.1f; }
    } else if (KIND == 4) {
        f4 acc = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            const size_t i = ((size_t)blockIdx.x * 256 + tid + (size_t)it * 65536) & ((1u << 22) - 1);
            acc += gin[i];
            if ((it & 7) == 7) gout[i] = acc;
        }
        x = acc.x;
    } else if (KIND == 5) {
        f16v acc = {};
        bf8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (tid + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
        for (int it = 0; it < iters; ++it) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
            asm volatile("v_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0" : "+v"(y) :: "a0");
        }
        x = acc[0] + y;
    } else if (KIND == 6) {
        for (int it = 0; it < iters; ++it) {
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            const f2 v = {x, y};
            const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
            x -= __uint_as_float(pk << 16); y -= __uint_as_float(pk & 0xffff0000u);
            x = x * 1000.0f + 1.001f; y = y * 1000.0f + 0.502f;
        }
    } else if (KIND == 7) {
        int s = 0;
        for (int it = 0; it < iters; ++it) {
            s += __builtin_amdgcn_readlane(__float_as_int(x), (it * 7) & 63) & 3;
            s += __builtin_amdgcn_readfirstlane(lane + it) & 1;
            x = x * 0.999f + 0.001f * (s & 7);
        }
    } else if (KIND == 8) {
        for (int i = tid; i < 64 * 68 * 2; i += 256) lds[i] = 0.001f * (i & 255);
        __syncthreads();
        f16v acc = {};
        for (int it = 0; it < iters; ++it) {
            const f4 av = *reinterpret_cast<const f4*>(lds + ((lane & 31) * 68 + (lane >> 5) * 8 + (it & 7) * 4) % (64 * 68));
            const f4 bv = *reinterpret_cast<const f4*>(lds + 64 * 68 + ((lane & 31) * 68 + (lane >> 5) * 8 + (it & 3) * 4) % (64 * 68));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, av), __builtin_bit_cast(bf8, bv), acc, 0, 0, 0);
        }
        x = acc[0];
    } else {
        float r[160];
#pragma unroll
        for (int i = 0; i < 160; ++i) r[i] = x + i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 160; ++i) r[i] = fmaf(r[i], 0.999f, 0.001f * i);
        }
        x = 0.f;
#pragma unroll
        for (int i = 0; i < 160; ++i) x += r[i];
    }
    if (x == 12345.678f) sink[tid] = x + y;
}

// mode `opsel3`: the one synthetic co-runner that triggers (bf16 MFMA + v_accvgpr moves) taken apart
//  0 v_accvgpr_write_b32 + v_accvgpr_read_b32 only          1 v_accvgpr_write_b32 only          2 v_accvgpr_read_b32 only
//  3 MFMA whose accumulator lives in AGPRs (no explicit moves)   4 MFMA (VGPR accumulator) + the moves, as in opsel2
//  5 v_accvgpr_mov_b32 a1, a0 only
template <int KIND>
__global__ __launch_bounds__(256) void acc_corunner(float* sink, int iters) {
    float y = 0.5f + 0.001f * threadIdx.x;
    f16v acc = {};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) asm volatile("v_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0" : "+v"(y) :: "a0");
        else if (KIND == 1) asm volatile("v_accvgpr_write_b32 a0, %0" :: "v"(y) : "a0");
        else if (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(y) :: "a0");
        else if (KIND == 3) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" :: "v"(a), "v"(b) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
        else if (KIND == 4) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0); asm volatile("v_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0" : "+v"(y) :: "a0"); }
        else asm volatile("v_accvgpr_mov_b32 a1, a0" ::: "a0", "a1");
        y = y * 0.999f + 0.001f;
    }
    if (y == 12345.678f) sink[threadIdx.x] = y + acc[0];
}
// victims of mode `opsel3`: other packed-f32 opcodes and selections next to co-runner 0
// (co-runner 4: MFMA + accvgpr moves)
//  0 v_pk_mul op_sel:[0,1]   1 v_pk_add op_sel:[0,1]   2 v_pk_fma op_sel:[0,1,0]   3 v_pk_mul op_sel_hi:[1,0] (src1 LOW half feeding the HIGH result)
//  4 v_pk_mul op_sel:[1,0] (src0 high half feeding the low result)   5 v_pk_mul without op_sel
template <int FORM>
__global__ __launch_bounds__(256) void opsel_victim2(unsigned* __restrict__ bad_by_lane, unsigned* __restrict__ bad_lo_hi, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned bad = 0, bad_lo = 0, bad_hi = 0, as_if_ignored = 0;
    f2 a = {1.0f + 0.001f * threadIdx.x, 2.0f + 0.003f * threadIdx.x};
    const f2 p = {3.0f + 0.01f * lane, 0.5f + 0.001f * lane}, c = {0.25f, 0.75f};
    for (int it = 0; it < iters; ++it) {
        f2 d;
        float e0, e1;
        if (FORM == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(p)); e0 = __fmul_rn(a.x, p.y); e1 = __fmul_rn(a.y, p.y); }
        else if (FORM == 1) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(p)); e0 = __fadd_rn(a.x, p.y); e1 = __fadd_rn(a.y, p.y); }
        else if (FORM == 2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(p), "v"(c)); e0 = __fmaf_rn(a.x, p.y, c.x); e1 = __fmaf_rn(a.y, p.y, c.y); }
        else if (FORM == 3) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(p)); e0 = __fmul_rn(a.x, p.x); e1 = __fmul_rn(a.y, p.x); }
        else if (FORM == 4) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(p)); e0 = __fmul_rn(a.y, p.x); e1 = __fmul_rn(a.y, p.y); }
        else { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(p)); e0 = __fmul_rn(a.x, p.x); e1 = __fmul_rn(a.y, p.y); }
        if (__float_as_uint(d.x) != __float_as_uint(e0)) {
            ++bad; ++bad_lo;
            // what WAS computed?  the same operation with the source's LOW half (the op_sel bit ignored)
            const float ign = FORM == 0 ? __fmul_rn(a.x, p.x) : FORM == 1 ? __fadd_rn(a.x, p.x) : FORM == 2 ? __fmaf_rn(a.x, p.x, c.x) : FORM == 4 ? __fmul_rn(a.x, p.x) : 0.f;
            if (__float_as_uint(d.x) == __float_as_uint(ign)) ++as_if_ignored;
        }
        if (__float_as_uint(d.y) != __float_as_uint(e1)) { ++bad; ++bad_hi; }
        a.x = a.x * 1.000001f + 1e-3f; a.y = a.y * 0.999999f + 2e-3f;
        if (a.x > 1e6f) a.x = 1.0f;
    }
    if (bad) { atomicAdd(&bad_by_lane[lane], bad); atomicAdd(&bad_lo_hi[0], bad_lo); atomicAdd(&bad_lo_hi[1], bad_hi); atomicAdd(&bad_lo_hi[2], as_if_ignored); }
}

static int mode_opsel3(int reps) {
    hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
    unsigned *bad_lane, *bad_lh; CK(hipMalloc(&bad_lane, 64 * 4)); CK(hipMalloc(&bad_lh, 16));
    float* sink = dev_empty(4096);
    auto report = [&](const char* what) {
        unsigned h[64], lh[4]; CK(hipMemcpy(h, bad_lane, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(lh, bad_lh, 16, hipMemcpyDeviceToHost));
        unsigned long q[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l) q[l >> 4] += h[l];
        printf("opsel3 %-92s: wrong results in lanes 0-15 / 16-31 / 32-47 / 48-63 = %lu / %lu / %lu / %lu (low result %u, high result %u; %u of the wrong low results equal the operation on the source's LOW half)\n", what, q[0], q[1], q[2], q[3], lh[0], lh[1], lh[2]);
        CK(hipMemset(bad_lane, 0, 64 * 4)); CK(hipMemset(bad_lh, 0, 16));
    };
    CK(hipMemset(bad_lane, 0, 64 * 4)); CK(hipMemset(bad_lh, 0, 16));
    const char* cn[6] = {"v_accvgpr_write + v_accvgpr_read", "v_accvgpr_write only", "v_accvgpr_read only", "MFMA with its accumulator in AGPRs", "MFMA + accvgpr moves", "v_accvgpr_mov only"};
    for (int k = 0; k < 6; ++k) {
        for (int rep = 0; rep < reps; ++rep) {
#define CO(K) case K: hipLaunchKernelGGL(acc_corunner<K>, dim3(512), dim3(256), 0, st[1], sink, 60000); break;
            for (int q = 0; q < 4; ++q) switch (k) { CO(0) CO(1) CO(2) CO(3) CO(4) CO(5) }
#undef CO
            hipLaunchKernelGGL(opsel_victim2<0>, dim3(512), dim3(256), 0, st[0], bad_lane, bad_lh, 400000);
            CK(hipDeviceSynchronize());
        }
        char what[160]; snprintf(what, sizeof what, "victim v_pk_mul op_sel:[0,1] | co-runner %s", cn[k]); report(what);
    }
    const char* vn[6] = {"v_pk_mul op_sel:[0,1]", "v_pk_add op_sel:[0,1]", "v_pk_fma op_sel:[0,1,0]", "v_pk_mul op_sel_hi:[1,0]", "v_pk_mul op_sel:[1,0]", "v_pk_mul (no op_sel)"};
    for (int f = 0; f < 6; ++f) {
        for (int rep = 0; rep < reps; ++rep) {
            for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(acc_corunner<4>, dim3(512), dim3(256), 0, st[1], sink, 60000);
#define VI(F) case F: hipLaunchKernelGGL(opsel_victim2<F>, dim3(512), dim3(256), 0, st[0], bad_lane, bad_lh, 400000); break;
            switch (f) { VI(0) VI(1) VI(2) VI(3) VI(4) VI(5) }
#undef VI
            CK(hipDeviceSynchronize());
        }
        char what[160]; snprintf(what, sizeof what, "victim %s | co-runner MFMA + accvgpr moves", vn[f]); report(what);
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ mode opsel4 (round 5: the two missing controls)
// (a) PADDING: the triggering co-runner (bf16 MFMA on a VGPR accumulator + v_accvgpr_write / read) with idle wait states around its moves - if the
//     failure were a missing hazard wait of the co-runner's own instruction stream, padding would remove it.
//       0 as in opsel3   1 s_nop 7 x 2 in front of the write and behind the read   2 three s_nop 15 between the MFMA and the moves (the MFMA has retired)
// (b) PLACEMENT: victim and co-runner on CU-masked streams (hipExtStreamCreateWithCUMask); every workgroup records (XCC id, HW_ID CU bits) so the
//     placement that was actually obtained is printed next to the counts.  same CUs / same XCD but disjoint CUs / different XCDs.
template <int PAD>
__global__ __launch_bounds__(256) void pad_corunner(float* sink, int iters, unsigned char* seen) {
    if (seen && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        seen[(xcc & 7) * 256 + ((hw >> 8) & 0xFF)] = 1;
    }
    float y = 0.5f + 0.001f * threadIdx.x;
    f16v acc = {};
    bf8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
    for (int it = 0; it < iters; ++it) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        if (PAD == 0) asm volatile("v_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0" : "+v"(y) :: "a0");
        else if (PAD == 1) asm volatile("s_nop 7\n\ts_nop 7\n\tv_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0\n\ts_nop 7\n\ts_nop 7" : "+v"(y) :: "a0");
        else asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0\n\ts_nop 15" : "+v"(y) :: "a0");
        y = y * 0.999f + 0.001f;
    }
    if (y == 12345.678f) sink[threadIdx.x] = y + acc[0];
}
__global__ __launch_bounds__(256) void placed_victim(unsigned* __restrict__ bad_by_lane, unsigned* __restrict__ bad_lo_hi, int iters, unsigned char* seen) {
    if (seen && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        seen[(xcc & 7) * 256 + ((hw >> 8) & 0xFF)] = 1;
    }
    const int lane = threadIdx.x & 63;
    unsigned bad = 0, bad_lo = 0, bad_hi = 0;
    f2 a = {1.0f + 0.001f * threadIdx.x, 2.0f + 0.003f * threadIdx.x};
    const f2 p = {3.0f + 0.01f * lane, 0.5f + 0.001f * lane};
    for (int it = 0; it < iters; ++it) {
        f2 d;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(p));
        const float e0 = __fmul_rn(a.x, p.y), e1 = __fmul_rn(a.y, p.y);
        if (__float_as_uint(d.x) != __float_as_uint(e0)) { ++bad; ++bad_lo; }
        if (__float_as_uint(d.y) != __float_as_uint(e1)) { ++bad; ++bad_hi; }
        a.x = a.x * 1.000001f + 1e-3f; a.y = a.y * 0.999999f + 2e-3f;
        if (a.x > 1e6f) a.x = 1.0f;
    }
    if (bad) { atomicAdd(&bad_by_lane[lane], bad); atomicAdd(&bad_lo_hi[0], bad_lo); atomicAdd(&bad_lo_hi[1], bad_hi); }
}

static int mode_opsel4(int reps) {
    unsigned *bad_lane, *bad_lh; CK(hipMalloc(&bad_lane, 64 * 4)); CK(hipMalloc(&bad_lh, 16));
    unsigned char *seen_v, *seen_c; CK(hipMalloc(&seen_v, 2048)); CK(hipMalloc(&seen_c, 2048));
    float* sink = dev_empty(4096);
    auto counts = [&](unsigned long q[4], unsigned lh[4]) {
        unsigned h[64]; CK(hipMemcpy(h, bad_lane, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(lh, bad_lh, 16, hipMemcpyDeviceToHost));
        q[0] = q[1] = q[2] = q[3] = 0;
        for (int l = 0; l < 64; ++l) q[l >> 4] += h[l];
        CK(hipMemset(bad_lane, 0, 64 * 4)); CK(hipMemset(bad_lh, 0, 16));
    };
    CK(hipMemset(bad_lane, 0, 64 * 4)); CK(hipMemset(bad_lh, 0, 16));
    {   // (a) padding around the co-runner's moves, both kernels unmasked
        hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
        const char* pn[3] = {"MFMA + accvgpr moves (as opsel3)", "... with s_nop 7 x 2 around the moves", "... with 3 x s_nop 15 between the MFMA and the moves"};
        for (int k = 0; k < 3; ++k) {
            for (int rep = 0; rep < reps; ++rep) {
                for (int q = 0; q < 4; ++q) {
                    if (k == 0) hipLaunchKernelGGL(pad_corunner<0>, dim3(512), dim3(256), 0, st[1], sink, 60000, (unsigned char*)nullptr);
                    else if (k == 1) hipLaunchKernelGGL(pad_corunner<1>, dim3(512), dim3(256), 0, st[1], sink, 45000, (unsigned char*)nullptr);
                    else hipLaunchKernelGGL(pad_corunner<2>, dim3(512), dim3(256), 0, st[1], sink, 30000, (unsigned char*)nullptr);
                }
                hipLaunchKernelGGL(placed_victim, dim3(512), dim3(256), 0, st[0], bad_lane, bad_lh, 400000, (unsigned char*)nullptr);
                CK(hipDeviceSynchronize());
            }
            unsigned long q[4]; unsigned lh[4]; counts(q, lh);
            printf("opsel4 padding   | co-runner %-58s: wrong results in lanes 0-15 / 16-31 / 32-47 / 48-63 = %lu / %lu / %lu / %lu (low result %u, high result %u)\n", pn[k], q[0], q[1], q[2], q[3], lh[0], lh[1]);
        }
    }
    // (b) placement by CONTIGUOUS ranges of CU-mask bits (sparse masks were not honoured by the runtime in the first attempt of this round: a mask of
    // every eighth bit ran on 192 CUs).  On this part the mask bits are shader-engine interleaved - bits 0..31 are 4 CUs in EACH of the 8 XCCs
    // (recorded XCC mask 0xff) - so disjoint bit ranges are disjoint CU sets on the SAME XCDs: the control isolates "same CU", not "same XCD".
    struct Case { const char* name; int v0, v1, c0, c1; } cases[6] = {
        {"victim bits [0,256), co-runner bits [0,256): everything shared", 0, 256, 0, 256}, {"victim bits [0,128), co-runner bits [0,128): shared half", 0, 128, 0, 128},
        {"victim bits [0,32), co-runner bits [0,32): shared eighth", 0, 32, 0, 32},          {"victim bits [0,128), co-runner bits [128,256): DISJOINT CUs", 0, 128, 128, 256},
        {"victim bits [128,256), co-runner bits [0,128): DISJOINT CUs", 128, 256, 0, 128},   {"victim bits [0,32), co-runner bits [32,256): DISJOINT CUs", 0, 32, 32, 256}};
    for (const Case& c : cases) {
        uint32_t mv[8] = {0}, mc[8] = {0};
        for (int b = c.v0; b < c.v1; ++b) mv[b >> 5] |= 1u << (b & 31);
        for (int b = c.c0; b < c.c1; ++b) mc[b >> 5] |= 1u << (b & 31);
        hipStream_t sv, sc;
        if (hipExtStreamCreateWithCUMask(&sv, 8, mv) != hipSuccess || hipExtStreamCreateWithCUMask(&sc, 8, mc) != hipSuccess) { printf("opsel4 placement: hipExtStreamCreateWithCUMask failed\n"); return 1; }
        CK(hipMemset(seen_v, 0, 2048)); CK(hipMemset(seen_c, 0, 2048));
        for (int rep = 0; rep < reps; ++rep) {
            for (int q = 0; q < 4; ++q) hipLaunchKernelGGL(pad_corunner<0>, dim3(2 * (c.c1 - c.c0)), dim3(256), 0, sc, sink, 60000, seen_c);
            hipLaunchKernelGGL(placed_victim, dim3(2 * (c.v1 - c.v0)), dim3(256), 0, sv, bad_lane, bad_lh, 400000, seen_v);
            CK(hipDeviceSynchronize());
        }
        unsigned char hv[2048], hc[2048]; CK(hipMemcpy(hv, seen_v, 2048, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc, seen_c, 2048, hipMemcpyDeviceToHost));
        int nv = 0, nc = 0, both = 0; unsigned xv = 0, xc = 0;
        for (int i = 0; i < 2048; ++i) { nv += hv[i]; nc += hc[i]; both += hv[i] && hc[i]; if (hv[i]) xv |= 1u << (i >> 8); if (hc[i]) xc |= 1u << (i >> 8); }
        unsigned long q[4]; unsigned lh[4]; counts(q, lh);
        printf("opsel4 placement | %-66s: victim ran on %3d CUs (XCC mask 0x%02x), co-runner on %3d CUs (XCC mask 0x%02x), %3d CUs shared, %.1e results checked: wrong in lanes 0-15 / 16-31 / 32-47 / 48-63 = %lu / %lu / %lu / %lu (low %u, high %u)\n",
               c.name, nv, xv, nc, xc, both, 2.0 * reps * 2 * (c.v1 - c.v0) * 256 * 400000.0, q[0], q[1], q[2], q[3], lh[0], lh[1]);
        CK(hipStreamDestroy(sv)); CK(hipStreamDestroy(sc));
    }
    return 0;
}

static int mode_opsel2(int reps) {
    hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
    unsigned *bad_lane, *bad_lh; CK(hipMalloc(&bad_lane, 64 * 4)); CK(hipMalloc(&bad_lh, 8));
    float* sink = dev_empty(4096);
    f4 *gin = (f4*)dev_rand((size_t)16 << 20, 1.f), *gout = (f4*)dev_empty((size_t)16 << 20);
    const char* names[10] = {"DPP adds", "v_permlane32/16_swap", "LDS write + read + s_barrier", "ds_bpermute", "global load + store stream", "bf16 MFMA + v_accvgpr moves",
                             "v_cvt_pk_bf16_f32 + shifts", "v_readlane / readfirstlane", "ds_read_b128 + MFMA from LDS", "160 live VGPRs + scalar FMAs"};
    const int its[10] = {60000, 60000, 12000, 20000, 4000, 20000, 40000, 60000, 20000, 600};
    for (int k = 0; k < 10; ++k) {
        CK(hipMemset(bad_lane, 0, 64 * 4)); CK(hipMemset(bad_lh, 0, 8));
        for (int rep = 0; rep < reps; ++rep) {
#define CO(K) case K: hipLaunchKernelGGL(synth_corunner<K>, dim3(512), dim3(256), 0, st[1], sink, gin, gout, its[K]); break;
            for (int q = 0; q < 4; ++q) switch (k) { CO(0) CO(1) CO(2) CO(3) CO(4) CO(5) CO(6) CO(7) CO(8) CO(9) }
#undef CO
            hipLaunchKernelGGL(opsel_victim<0>, dim3(512), dim3(256), 0, st[0], bad_lane, bad_lh, 400000);
            CK(hipDeviceSynchronize());
        }
        unsigned h[64], lh[2]; CK(hipMemcpy(h, bad_lane, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(lh, bad_lh, 8, hipMemcpyDeviceToHost));
        unsigned long q[4] = {0, 0, 0, 0};
        for (int l = 0; l < 64; ++l) q[l >> 4] += h[l];
        printf("opsel2 victim v_pk_mul op_sel:[0,1] | co-runner %-32s: wrong results in lanes 0-15 / 16-31 / 32-47 / 48-63 = %lu / %lu / %lu / %lu (low %u, high %u)\n", names[k], q[0], q[1], q[2],
               q[3], lh[0], lh[1]);
    }
    return 0;
}

static int mode_opsel(int iters, const char* libpath) {
    hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
    // optional sixth co-runner: the library's GEMM chain (the only co-runner that triggers the defect in mode `lib`) next to the bare instruction loop
    Weights w{}; Bufs cb{}; const float *catt = nullptr, *cxin = nullptr;
    if (libpath) {
        void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "dlopen %s: %s\n", libpath, dlerror()); return 2; }
        p_afm_linear = (int (*)(const afm_linear_args*, void*))dlsym(h, "afm_linear");
        w = Weights{dev_rand((size_t)D * D, 0.044f), dev_rand(D, 0.1f), dev_rand(D, 0.2f, 1.f), dev_rand(D, 0.1f), dev_rand((size_t)FF * D, 0.044f), dev_rand(FF, 0.5f),
                    dev_rand(FF, 0.1f), dev_rand((size_t)D * FF, 0.031f), dev_rand(D, 0.1f), dev_rand((size_t)3 * D * D, 0.044f), dev_rand(3 * D, 0.5f), dev_rand(3 * D, 0.1f)};
        cb = mk(); catt = dev_rand((size_t)M * D, 1.f); cxin = dev_rand((size_t)M * D, 2.f, 0.7f);
    }
    unsigned *bad_lane, *bad_lh; CK(hipMalloc(&bad_lane, 64 * 4)); CK(hipMalloc(&bad_lh, 8));
    float* sink = dev_empty(4096);
    const char* forms[3] = {"v_pk_mul d, a, p op_sel:[0,1]", "v_pk_mul d, p, a op_sel:[1,0]", "v_pk_mul d, a, pp (no op_sel)"};
    const char* cos[6] = {"v_pk_mul without op_sel", "v_pk_mul op_sel:[1,0]", "v_pk_mul op_sel:[0,1]", "scalar v_mul only", "nothing", "the library's GEMM chain"};
    for (int form = 0; form < 3; ++form)
        for (int co = 0; co < (libpath ? 6 : 5); ++co) {
            CK(hipMemset(bad_lane, 0, 64 * 4)); CK(hipMemset(bad_lh, 0, 8));
            for (int rep = 0; rep < iters; ++rep) {
                // both kernels at half the chip's wave slots each, so that waves of the two share SIMDs for the whole run
                switch (co) {
                    case 0: hipLaunchKernelGGL(opsel_corunner<0>, dim3(1024), dim3(256), 0, st[1], sink, 200000); break;
                    case 1: hipLaunchKernelGGL(opsel_corunner<1>, dim3(1024), dim3(256), 0, st[1], sink, 200000); break;
                    case 2: hipLaunchKernelGGL(opsel_corunner<2>, dim3(1024), dim3(256), 0, st[1], sink, 200000); break;
                    case 3: hipLaunchKernelGGL(opsel_corunner<3>, dim3(1024), dim3(256), 0, st[1], sink, 200000); break;
                    case 5: for (int k = 0; k < 12; ++k) chain(w, catt, cxin, cb, st[1]); break;
                    default: break;
                }
                // the victim at HALF a wave slot set (512 workgroups of 256 = 2 waves per SIMD) so that the chain's workgroups find room next to it
                const int vg = co == 5 ? 512 : 1024, vi = co == 5 ? 400000 : 100000;
                if (form == 0) hipLaunchKernelGGL(opsel_victim<0>, dim3(vg), dim3(256), 0, st[0], bad_lane, bad_lh, vi);
                else if (form == 1) hipLaunchKernelGGL(opsel_victim<1>, dim3(vg), dim3(256), 0, st[0], bad_lane, bad_lh, vi);
                else hipLaunchKernelGGL(opsel_victim<2>, dim3(vg), dim3(256), 0, st[0], bad_lane, bad_lh, vi);
                CK(hipDeviceSynchronize());
            }
            unsigned h[64], lh[2]; CK(hipMemcpy(h, bad_lane, 256, hipMemcpyDeviceToHost)); CK(hipMemcpy(lh, bad_lh, 8, hipMemcpyDeviceToHost));
            unsigned long q[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) q[l >> 4] += h[l];
            printf("opsel victim %-32s | co-runner %-26s: wrong results in lanes 0-15 / 16-31 / 32-47 / 48-63 = %lu / %lu / %lu / %lu  (low result %u, high result %u) of %.2e checked\n",
                   forms[form], cos[co], q[0], q[1], q[2], q[3], lh[0], lh[1], 2.0 * iters * 1024 * 256 * 100000.0);
        }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 2 && !strcmp(argv[1], "lib")) return mode_lib(argv[2], argc > 3 ? atoi(argv[3]) : 20, argc > 4 ? argv[4] : "chain");
    if (argc > 1 && !strcmp(argv[1], "mini")) return mode_mini(argc > 2 ? atoi(argv[2]) : 20);
    if (argc > 1 && !strcmp(argv[1], "opsel4")) return mode_opsel4(argc > 2 ? atoi(argv[2]) : 3);
    if (argc > 1 && !strcmp(argv[1], "opsel3")) return mode_opsel3(argc > 2 ? atoi(argv[2]) : 3);
    if (argc > 1 && !strcmp(argv[1], "opsel2")) return mode_opsel2(argc > 2 ? atoi(argv[2]) : 3);
    if (argc > 1 && !strcmp(argv[1], "opsel")) return mode_opsel(argc > 2 ? atoi(argv[2]) : 3, argc > 3 ? argv[3] : nullptr);
    fprintf(stderr, "usage: %s lib <libafm_hip.so> [iterations [chain|mfma|valu|mem|none]] | mini [iterations]\n", argv[0]);
    return 1;
}
