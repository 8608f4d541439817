// Stand-alone client of the C-ABI (include/afm_hip.h): no Python, no torch - plain HIP allocations and the extern "C" entry points,
// the way a host written in any other language would bind them.  Runs afm_linear (+GELU), afm_layernorm and afm_mha_fwd on small
// inputs and checks them against straightforward host loops.  Exit code 0 = all good.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "afm_hip.h"

#define HIP_OK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("hip error %d at line %d\n", (int)e, __LINE__); return 2; } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; }

int main() {
    if (afm_version() != AFM_ABI_VERSION) { std::printf("ABI version mismatch\n"); return 1; }
    unsigned seed = 7;
    const int M = 70, K = 64, N = 128;
    std::vector<float> A(M * K), W(N * K), b(N), C(M * N), ref(M * N);
    for (auto& v : A) v = frand(seed);
    for (auto& v : W) v = frand(seed) * 0.2f;
    for (auto& v : b) v = frand(seed);
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double acc = b[n];
            for (int k = 0; k < K; ++k) acc += (double)A[m * K + k] * W[n * K + k];
            ref[m * N + n] = (float)(0.5 * acc * (1.0 + std::erf(acc * 0.7071067811865476)));
        }
    float *dA, *dW, *db, *dC;
    HIP_OK(hipMalloc(&dA, A.size() * 4)); HIP_OK(hipMalloc(&dW, W.size() * 4)); HIP_OK(hipMalloc(&db, b.size() * 4)); HIP_OK(hipMalloc(&dC, C.size() * 4));
    HIP_OK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    afm_linear_args a = {};
    a.A = dA; a.lda = K; a.W = dW; a.ldw = K; a.C = dC; a.ldc = N; a.M = M; a.N = N; a.K = K; a.bias = db; a.act = AFM_ACT_GELU;
    int rc = afm_linear(&a, st);
    if (rc) { std::printf("afm_linear rc=%d\n", rc); return 1; }
    HIP_OK(hipMemcpyAsync(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    double err = 0;
    for (size_t i = 0; i < C.size(); ++i) err = std::fmax(err, std::fabs((double)C[i] - ref[i]));
    std::printf("afm_linear+GELU max|diff| = %.3e\n", err);
    if (!(err < 1e-4)) return 1;

    // LayerNorm over the GEMM output rows (dim 128)
    std::vector<float> g(N, 1.0f), be(N, 0.0f), Y(M * N);
    float *dg, *dbe, *dY;
    HIP_OK(hipMalloc(&dg, N * 4)); HIP_OK(hipMalloc(&dbe, N * 4)); HIP_OK(hipMalloc(&dY, Y.size() * 4));
    HIP_OK(hipMemcpy(dg, g.data(), N * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dbe, be.data(), N * 4, hipMemcpyHostToDevice));
    rc = afm_layernorm(dC, dg, dbe, dY, M, N, 1e-5f, st);
    if (rc) { std::printf("afm_layernorm rc=%d\n", rc); return 1; }
    HIP_OK(hipMemcpyAsync(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    err = 0;
    for (int m = 0; m < M; ++m) {
        double mean = 0, var = 0;
        for (int n = 0; n < N; ++n) mean += C[m * N + n];
        mean /= N;
        for (int n = 0; n < N; ++n) var += (C[m * N + n] - mean) * (C[m * N + n] - mean);
        var /= N;
        for (int n = 0; n < N; ++n) err = std::fmax(err, std::fabs((C[m * N + n] - mean) / std::sqrt(var + 1e-5) - Y[m * N + n]));
    }
    std::printf("afm_layernorm max|diff| = %.3e\n", err);
    if (!(err < 1e-4)) return 1;

    // attention: B=1, T=40, H=2, dh=64 with the last 5 keys masked
    const int T = 40, H = 2, D = 128;
    std::vector<float> qkv(T * 3 * D), out(T * D);
    std::vector<unsigned char> mask(T, 0);
    for (auto& v : qkv) v = frand(seed);
    for (int i = T - 5; i < T; ++i) mask[i] = 1;
    float *dq, *dout; unsigned char* dm;
    HIP_OK(hipMalloc(&dq, qkv.size() * 4)); HIP_OK(hipMalloc(&dout, out.size() * 4)); HIP_OK(hipMalloc(&dm, T));
    HIP_OK(hipMemcpy(dq, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(dm, mask.data(), T, hipMemcpyHostToDevice));
    rc = afm_mha_fwd(dq, dm, dout, 1, T, H, 64, st);
    if (rc) { std::printf("afm_mha_fwd rc=%d\n", rc); return 1; }
    HIP_OK(hipMemcpyAsync(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    err = 0;
    for (int h = 0; h < H; ++h)
        for (int i = 0; i < T; ++i) {
            std::vector<double> p(T);
            double mx = -1e30, sum = 0;
            for (int j = 0; j < T; ++j) {
                double s = 0;
                for (int c = 0; c < 64; ++c) s += (double)qkv[i * 3 * D + h * 64 + c] * qkv[j * 3 * D + D + h * 64 + c];
                p[j] = mask[j] ? -1e30 : s / 8.0;
                mx = std::fmax(mx, p[j]);
            }
            for (int j = 0; j < T; ++j) { p[j] = mask[j] ? 0.0 : std::exp(p[j] - mx); sum += p[j]; }
            for (int c = 0; c < 64; ++c) {
                double o = 0;
                for (int j = 0; j < T; ++j) o += p[j] / sum * qkv[j * 3 * D + 2 * D + h * 64 + c];
                err = std::fmax(err, std::fabs(o - out[i * D + h * 64 + c]));
            }
        }
    std::printf("afm_mha_fwd max|diff| = %.3e\n", err);
    if (!(err < 1e-4)) return 1;
    // argument validation is part of the contract
    if (afm_mha_fwd(dq, dm, dout, 1, T, 4, 32, st) != AFM_E_UNSUPPORTED) { std::printf("expected AFM_E_UNSUPPORTED\n"); return 1; }
    if (afm_linear(nullptr, st) != AFM_E_BADARG) { std::printf("expected AFM_E_BADARG\n"); return 1; }
    std::printf("C-ABI client OK\n");
    return 0;
}
