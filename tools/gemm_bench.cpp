// Stand-alone timing + accuracy check of afm_linear through the C-ABI (no Python): the encoder's GEMM shapes, HIP-event
// timing, and the error of sampled outputs against a float64 host reference.  Modes are the library's env knobs
// (AFM_GEMM_SPLIT=9|6, AFM_GEMM_SPLIT_TILE, AFM_GEMM_TILE), so run the binary once per mode.
//   hipcc -O2 -Iinclude tools/gemm_bench.cpp -Lafford-motion_amd/afm -lafm_hip -Wl,-rpath,'$ORIGIN/../afford-motion_amd/afm' -o tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "afm_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

int main(int argc, char** argv) {
    struct Shape { int M, N, K; const char* name; };
    std::vector<Shape> shapes = {{10432, 1536, 512, "in_proj"}, {10432, 512, 512, "out_proj"}, {10432, 1024, 512, "ffn1"},
                                 {10432, 512, 1024, "ffn2"},    {5216, 1536, 512, "in_proj/2"}, {5216, 512, 512, "out_proj/2"},
                                 {5216, 1024, 512, "ffn1/2"},   {5216, 512, 1024, "ffn2/2"},   {4096, 4096, 4096, "4096^3"}};
    if (argc == 4) shapes = {{atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), "cli"}};
    const char* mode = getenv("AFM_GEMM_SPLIT");
    printf("mode: AFM_GEMM_SPLIT=%s AFM_GEMM_SPLIT_TILE=%s\n", mode ? mode : "(default)", getenv("AFM_GEMM_SPLIT_TILE") ? getenv("AFM_GEMM_SPLIT_TILE") : "-");
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const size_t na = (size_t)sh.M * sh.K, nw = (size_t)sh.N * sh.K, nc = (size_t)sh.M * sh.N;
        std::vector<float> A(na), W(nw), bias(sh.N), Cc(nc);
        for (auto& v : A) v = nd(rng) * (1.0f + 3.0f * (rng() % 7 == 0));      // a few larger magnitudes
        for (auto& v : W) v = nd(rng) * 0.05f;
        for (auto& v : bias) v = nd(rng) * 0.1f;
        float *dA, *dW, *dB, *dC;
        CK(hipMalloc(&dA, na * 4)); CK(hipMalloc(&dW, nw * 4)); CK(hipMalloc(&dB, sh.N * 4)); CK(hipMalloc(&dC, nc * 4));
        CK(hipMemcpy(dA, A.data(), na * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, W.data(), nw * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, bias.data(), sh.N * 4, hipMemcpyHostToDevice));
        afm_linear_args a;
        memset(&a, 0, sizeof a);
        a.A = dA; a.lda = sh.K; a.W = dW; a.ldw = sh.K; a.C = dC; a.ldc = sh.N; a.bias = dB; a.M = sh.M; a.N = sh.N; a.K = sh.K;
        for (int i = 0; i < 3; ++i) { int rc = afm_linear(&a, st); if (rc) { fprintf(stderr, "afm_linear rc=%d\n", rc); return 3; } }
        CK(hipStreamSynchronize(st));
        const int reps = 20;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) afm_linear(&a, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(Cc.data(), dC, nc * 4, hipMemcpyDeviceToHost));
        // accuracy on sampled entries: error relative to sum_k |a||w| (the scale f32 rounding errors live on)
        double worst = 0, sum2 = 0, worst_rel = 0;
        const int samples = 4096;
        for (int t = 0; t < samples; ++t) {
            const int m = rng() % sh.M, n = rng() % sh.N;
            double ref = bias[n], scale = std::fabs((double)bias[n]);
            for (int k = 0; k < sh.K; ++k) {
                const double pr = (double)A[(size_t)m * sh.K + k] * (double)W[(size_t)n * sh.K + k];
                ref += pr; scale += std::fabs(pr);
            }
            const double err = std::fabs((double)Cc[(size_t)m * sh.N + n] - ref);
            worst = std::fmax(worst, err / scale);
            sum2 += (err / scale) * (err / scale);
            if (std::fabs(ref) > 1e-3 * scale) worst_rel = std::fmax(worst_rel, err / std::fabs(ref));
        }
        const double us = 1e3 * ms / reps, tf = 2.0 * sh.M * sh.N * sh.K / (us * 1e-6) / 1e12;
        printf("%-11s M=%5d N=%4d K=%4d  %8.1f us  %6.1f TF   err/sum|a||w|: max %.2e rms %.2e   max rel %.2e\n", sh.name, sh.M, sh.N, sh.K,
               us, tf, worst, std::sqrt(sum2 / samples), worst_rel);
        CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
