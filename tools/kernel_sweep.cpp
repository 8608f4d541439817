// Stand-alone sweep of the hot kernels through the C-ABI (no Python, no torch): every GEMM shape of the CMDM encoder at several
// batch sizes x {arithmetic, workgroup tile}, and the attention kernel x workgroup grouping.  HIP-event timing (median of
// `reps` launches timed one by one on the launch stream), bit-equality of the variants that must agree (tile shapes of one
// arithmetic, attention groupings), and the error of sampled GEMM outputs against a float64 host reference.
//   hipcc -O2 -Iinclude tools/kernel_sweep.cpp -Lafford-motion_amd/afm -lafm_hip -Wl,-rpath,'$ORIGIN/../afford-motion_amd/afm' -o tools/kernel_sweep
//   tools/kernel_sweep [gemm|mha|all] [batches, e.g. 32,4] > profiles/rNN_kernel_sweep.txt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "afm_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static hipStream_t st;
static hipEvent_t e0, e1;

template <class F>
static double time_us(F&& launch, int reps = 15) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipStreamSynchronize(st));
    std::vector<float> ts;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0, st));
        launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

struct GemmCfg { int arith, min_n, tile; const char* name; int flags = 0; };

static void gemm_sweep(const std::vector<int>& batches) {
    const int T = 326, L = 196;
    const GemmCfg cfgs[] = {{AFM_ARITH_F32, 0, 0, "f32 auto"},   {AFM_ARITH_F32, 0, 1, "f32 32x32"},   {AFM_ARITH_F32, 0, 2, "f32 32x64"},
                            {AFM_ARITH_F32, 0, 3, "f32 64x64"},  {AFM_ARITH_F32, 0, 4, "f32 64x128"},  {AFM_ARITH_F32, 0, 5, "f32 128x128"},
                            {AFM_ARITH_BF16X9, 0, 0, "x9 auto"}, {AFM_ARITH_BF16X9, 0, 3, "x9 64x64"}, {AFM_ARITH_BF16X9, 0, 7, "x9 64x64 split-K"}, {AFM_ARITH_BF16X9, 0, 10, "x9 split-K 3-stage"}, {AFM_ARITH_BF16X9, 0, 11, "x9 split-K 2x2seg"}, {AFM_ARITH_BF16X9, 0, 9, "x9 64x64 3-stage"}, {AFM_ARITH_BF16X9, 0, 12, "x9 64x64 walked"}, {AFM_ARITH_BF16X9, 0, 5, "x9 128x128"},
                            {AFM_ARITH_BF16X6, 0, 0, "x6 auto"}};
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int B : batches) {
        struct Shape { int M, N, K; const char* name; };
        const Shape shapes[] = {{B * T, 1536, 512, "in_proj"}, {B * T, 512, 512, "out_proj"}, {B * T, 1024, 512, "ffn1"}, {B * T, 512, 1024, "ffn2"},
                                {B * L, 512, 512, "out_proj(last)"}, {B * L, 263, 512, "motion_layer"}};
        for (const Shape& sh : shapes) {
            const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
            std::vector<float> A(na), W(nw), bias(sh.N), ref_out, out(nc);
            for (auto& v : A) v = nd(rng) * (1.0f + 3.0f * (rng() % 7 == 0));
            for (auto& v : W) v = nd(rng) * 0.05f;
            for (auto& v : bias) v = nd(rng) * 0.1f;
            float *dA, *dW, *dB, *dC, *dR;
            CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dW, nw * 4)); CK(hipMalloc(&dB, sh.N * 4)); CK(hipMalloc(&dC, nc * 4)); CK(hipMalloc(&dR, nc * 4));
            CK(hipMemcpy(dA, A.data(), na * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dW, W.data(), nw * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dB, bias.data(), sh.N * 4, hipMemcpyHostToDevice));
            CK(hipMemset(dR, 0, nc * 4));
            int last_arith = -1;
            for (const GemmCfg& c : cfgs) {
                afm_linear_args a;
                memset(&a, 0, sizeof a);
                a.A = dA; a.lda = sh.K; a.W = dW; a.ldw = sh.K; a.C = dC; a.ldc = sh.N; a.bias = dB; a.residual = dR; a.ldr = sh.N;
                a.M = sh.M; a.N = sh.N; a.K = sh.K; a.arith = c.arith; a.arith_min_n = c.min_n; a.tune = (c.tile << AFM_TUNE_TILE_SHIFT) | c.flags;
                CK(hipMemsetAsync(dC, 0xFF, nc * 4, st));
                int rc = afm_linear(&a, st);
                if (rc) { printf("B=%-2d %-15s %-11s rc=%d\n", B, sh.name, c.name, rc); continue; }
                const double us = time_us([&] { afm_linear(&a, st); });
                CK(hipMemcpy(out.data(), dC, nc * 4, hipMemcpyDeviceToHost));
                const char* same = "";
                if (c.arith != last_arith) { ref_out = out; last_arith = c.arith; }
                else same = memcmp(ref_out.data(), out.data(), nc * 4) == 0 ? "bit-identical to first variant" : "DIFFERS from first variant";
                double worst = 0;
                for (int t = 0; t < 512; ++t) {
                    const int m = rng() % sh.M, n = rng() % sh.N;
                    double ref = bias[n], scale = std::fabs((double)bias[n]);
                    for (int k = 0; k < sh.K; ++k) {
                        const double pr = (double)A[(size_t)m * sh.K + k] * (double)W[(size_t)n * sh.K + k];
                        ref += pr; scale += std::fabs(pr);
                    }
                    worst = std::fmax(worst, std::fabs((double)out[(size_t)m * sh.N + n] - ref) / scale);
                }
                printf("B=%-2d %-15s M=%5d N=%4d K=%4d  %-18s %8.1f us %6.1f TF  err %.1e  %s\n", B, sh.name, sh.M, sh.N, sh.K, c.name, us,
                       2.0 * sh.M * sh.N * sh.K / (us * 1e-6) / 1e12, worst, same);
            }
            CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dR));
        }
    }
}

static void mha_sweep(const std::vector<int>& batches) {
    const int T = 326, H = 8, dh = 64, D = H * dh;
    std::mt19937 rng(11);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int B : batches) {
        const size_t nq = (size_t)B * T * 3 * D, no = (size_t)B * T * D;
        std::vector<float> qkv(nq), ref, out(no);
        for (auto& v : qkv) v = nd(rng);
        std::vector<uint8_t> mask((size_t)B * T, 0);
        for (int b = 0; b < B; ++b)
            for (int t = T - (b % 4) * 20; t < T; ++t) mask[(size_t)b * T + t] = 1;       // ragged suffix padding on 3 of 4 samples
        float *dq, *dout; uint8_t* dm;
        CK(hipMalloc(&dq, nq * 4)); CK(hipMalloc(&dout, no * 4)); CK(hipMalloc(&dm, mask.size()));
        CK(hipMemcpy(dq, qkv.data(), nq * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dm, mask.data(), mask.size(), hipMemcpyHostToDevice));
        for (int masked = 0; masked < 2; ++masked) {
            bool have_ref = false;
            for (int g : {12, 8, 6, 4, 2, 1, 0}) {
                CK(hipMemsetAsync(dout, 0xFF, no * 4, st));
                int rc = afm_mha_fwd_grouped(dq, masked ? dm : nullptr, dout, B, T, H, dh, g, st);
                if (rc) { printf("B=%-2d mha group=%d rc=%d\n", B, g, rc); continue; }
                const double us = time_us([&] { afm_mha_fwd_grouped(dq, masked ? dm : nullptr, dout, B, T, H, dh, g, st); });
                CK(hipMemcpy(out.data(), dout, no * 4, hipMemcpyDeviceToHost));
                const char* same = "";
                if (!have_ref) { ref = out; have_ref = true; }
                else same = memcmp(ref.data(), out.data(), no * 4) == 0 ? "bit-identical" : "DIFFERS";
                printf("B=%-2d mha T=%d %s group_waves=%-2d %8.1f us %6.1f TF  %s\n", B, T, masked ? "ragged-mask" : "no-mask    ", g, us,
                       4.0 * B * H * (double)T * T * dh / (us * 1e-6) / 1e12, same);
            }
        }
        CK(hipFree(dq)); CK(hipFree(dout)); CK(hipFree(dm));
    }
}

// one x9 GEMM launched `reps` times (profiling target for rocprofv3 --pmc): kernel_sweep one M N K tile flags reps
static int one_gemm(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "usage: kernel_sweep one M N K tile flags reps\n"); return 1; }
    const int M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]), tile = atoi(argv[5]), flags = atoi(argv[6]), reps = atoi(argv[7]);
    std::mt19937 rng(3);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> A((size_t)M * K), W((size_t)N * K);
    for (auto& v : A) v = nd(rng);
    for (auto& v : W) v = nd(rng) * 0.05f;
    float *dA, *dW, *dC;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    afm_linear_args a;
    memset(&a, 0, sizeof a);
    a.A = dA; a.lda = K; a.W = dW; a.ldw = K; a.C = dC; a.ldc = N; a.M = M; a.N = N; a.K = K; a.arith = AFM_ARITH_BF16X9;
    a.tune = (tile << AFM_TUNE_TILE_SHIFT) | flags;
    for (int i = 0; i < reps; ++i)
        if (int rc = afm_linear(&a, st)) { fprintf(stderr, "rc=%d\n", rc); return 3; }
    CK(hipStreamSynchronize(st));
    const double us = time_us([&] { afm_linear(&a, st); });
    printf("M=%d N=%d K=%d tile=%d flags=%d: %.1f us %.1f TF\n", M, N, K, tile, flags, us, 2.0 * M * N * K / (us * 1e-6) / 1e12);
    return 0;
}

int main(int argc, char** argv) {
    const std::string what = argc > 1 ? argv[1] : "all";
    std::vector<int> batches = {32, 16, 8, 4, 1};
    if (argc > 2) {
        batches.clear();
        for (char* tok = strtok(argv[2], ","); tok; tok = strtok(nullptr, ",")) batches.push_back(atoi(tok));
    }
    if (afm_version() != AFM_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    CK(hipStreamCreate(&st));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    if (what == "one") return one_gemm(argc, argv);
    if (what == "gemm" || what == "all") gemm_sweep(batches);
    if (what == "mha" || what == "all") mha_sweep(batches);
    return 0;
}
