// Per-workgroup timeline of one afm_linear launch (native f32 MFMA LDS-DMA kernel): every workgroup records s_memrealtime (100 MHz,
// chip-global) at entry and exit plus its hardware id.  Answers "where does a launch's time go": how long tiles of the first
// resident round take vs the tail round, how many workgroups are resident over time, when the last CU goes idle.
// Compiles the library's own gemm sources with -DAFM_TIMELINE (instrumentation that is never part of libafm_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAFM_TIMELINE -Iinclude -Iafford-motion_amd/csrc tools/gemm_timeline.hip \
//         afford-motion_amd/csrc/profile.hip afford-motion_amd/csrc/gemm_slab.hip afford-motion_amd/csrc/gemm_thin.hip -o tools/gemm_timeline
//   tools/gemm_timeline [M N K [tile [arith]]]        arith: 1 = f32 MFMA (default), 9 / 6 = bf16 split
#include "../afford-motion_amd/csrc/gemm.hip"
#include "../afford-motion_amd/csrc/gemm_split.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// the producer of the sampling loop: the A rows (and the residual rows) were written by the PREVIOUS kernel, on other XCDs - rewrite them
// right before the measured launch so that they are where the loop finds them (written back to memory, in no reader's L2)
__global__ void produce_kernel(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
static int g_cold = 0;

static void one(int M, int N, int K, int tile, int arith = AFM_ARITH_F32, int pad = 0, int flags = 0, int lnfold = 0) {
    const int lda = K + pad, ldw = K + pad;           // pad != 0: row strides that are not a power of two (L2 channel spread)
    float *dA, *dW, *dC, *dR;
    CK(hipMalloc(&dA, (size_t)M * lda * 4)); CK(hipMalloc(&dW, (size_t)N * ldw * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dR, (size_t)M * N * 4));
    CK(hipMemset(dA, 0x11, (size_t)M * lda * 4)); CK(hipMemset(dW, 0x11, (size_t)N * ldw * 4)); CK(hipMemset(dR, 0, (size_t)M * N * 4));
    const int maxwg = ((M + 31) / 32) * ((N + 31) / 32);
    AfmTimelineRec* drec;
    CK(hipMalloc(&drec, (size_t)maxwg * sizeof(AfmTimelineRec)));
    CK(hipMemset(drec, 0, (size_t)maxwg * sizeof(AfmTimelineRec)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(afm_timeline), &drec, sizeof drec));
    afm_linear_args a = {};
    a.A = dA; a.lda = lda; a.W = dW; a.ldw = ldw; a.C = dC; a.ldc = N; a.residual = dR; a.ldr = N; a.M = M; a.N = N; a.K = K;
    a.arith = arith; a.tune = (tile << AFM_TUNE_TILE_SHIFT) | flags;
    float *dS1 = nullptr, *dS2 = nullptr, *dG = nullptr;
    if (lnfold) {       // the sampling loop's epilogues: 1 = out_proj / linear2 (raw residual normalised on the fly, statistics out), 2 = linear1 / in_proj (folded LayerNorm of A)
        CK(hipMalloc(&dS1, (size_t)M * 16 * 2 * 4)); CK(hipMalloc(&dS2, (size_t)M * 16 * 2 * 4)); CK(hipMalloc(&dG, (size_t)(N + K) * 4 * 2));
        CK(hipMemset(dS1, 0, (size_t)M * 16 * 2 * 4)); CK(hipMemset(dS2, 0, (size_t)M * 16 * 2 * 4)); CK(hipMemset(dG, 0, (size_t)(N + K) * 4 * 2));
        a.bias = dG; a.ln_eps2 = 1e-5f;
        if (lnfold == 1) { a.stat_out = dS1; a.res_stat = dS2; a.res_gamma = dG; a.res_beta = dG + N; }
        else { a.a_stat = dS2; a.a_stat_groups = K / 64; a.a_fold_g = dG; a.residual = nullptr; a.act = AFM_ACT_GELU; }
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) afm_linear(&a, st);
    CK(hipStreamSynchronize(st));
    CK(hipMemset(drec, 0, (size_t)maxwg * sizeof(AfmTimelineRec)));
    if (g_cold) {
        hipLaunchKernelGGL(produce_kernel, dim3(1024), dim3(256), 0, st, dA, (size_t)M * lda, 0.5f);
        hipLaunchKernelGGL(produce_kernel, dim3(1024), dim3(256), 0, st, dR, (size_t)M * N, 0.25f);
        if (dS2) hipLaunchKernelGGL(produce_kernel, dim3(256), dim3(256), 0, st, dS2, (size_t)M * 16 * 2, 0.0f);
    }
    CK(hipEventRecord(e0, st));
    afm_linear(&a, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<AfmTimelineRec> rec(maxwg);
    CK(hipMemcpy(rec.data(), drec, (size_t)maxwg * sizeof(AfmTimelineRec), hipMemcpyDeviceToHost));
    std::vector<AfmTimelineRec> r;
    for (auto& x : rec) if (x.t1) r.push_back(x);
    const int n = (int)r.size();
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto& x : r) { tmin = std::min(tmin, x.t0); tmax = std::max(tmax, x.t1); }
    const double tick_us = 0.01;                       // s_memrealtime: 100 MHz
    printf("== %s arith=%d flags=%d pad=%d lnfold=%d M=%d N=%d K=%d tile=%d: %d workgroups, event time %.1f us, first entry -> last exit %.1f us, %.1f TF\n", g_cold ? "COLD (A / residual / statistics just written by another kernel)" : "warm", arith, flags, pad, lnfold, M, N, K, tile, n, ms * 1e3,
           (tmax - tmin) * tick_us, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    std::map<unsigned, int> cus;
    for (auto& x : r) cus[(x.xcc_id & 0xF) << 16 | (x.hw_id & 0xFF00)]++;           // (xcc, se, sh, cu)
    int mn = 1 << 30, mx = 0;
    for (auto& kv : cus) { mn = std::min(mn, kv.second); mx = std::max(mx, kv.second); }
    printf("   distinct CUs %zu, workgroups per CU min %d max %d\n", cus.size(), mn, mx);
    // durations by start order: first resident round vs later
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return r[x].t0 < r[y].t0; });
    auto stats = [&](int lo, int hi, const char* name) {
        if (hi <= lo) return;
        std::vector<double> d;
        double last_end = 0, first_start = 1e30;
        for (int i = lo; i < hi; ++i) { const auto& x = r[order[i]]; d.push_back((x.t1 - x.t0) * tick_us); last_end = std::max(last_end, (x.t1 - tmin) * tick_us); first_start = std::min(first_start, (x.t0 - tmin) * tick_us); }
        std::sort(d.begin(), d.end());
        printf("   %-22s n=%4d  start %6.1f..  last end %6.1f us   duration min %5.1f  p50 %5.1f  p90 %5.1f  max %5.1f us\n", name, hi - lo, first_start, last_end,
               d.front(), d[d.size() / 2], d[d.size() * 9 / 10], d.back());
    };
    int first_round = 0;
    for (int i = 0; i < n; ++i) if ((r[order[i]].t0 - tmin) * tick_us < 3.0) first_round++;
    stats(0, first_round, "started in first 3 us");
    stats(first_round, n, "started later");
    // shader-clock view of wave 0 of every workgroup: effective clock and where the K loop's cycles go
    auto phases = [&](int lo, int hi, const char* name) {
        if (hi <= lo) return;
        double cyc = 0, us = 0, w = 0, b = 0, is = 0, mf = 0;
        for (int i = lo; i < hi; ++i) { const auto& x = r[order[i]]; cyc += (double)(x.c1 - x.c0); us += (x.t1 - x.t0) * tick_us; w += x.wait_c; b += x.barrier_c; is += x.issue_c; mf += x.mfma_c; }
        const int m = hi - lo;
        printf("   %-22s shader clock %.2f GHz; per workgroup: %.0f cycles = vmcnt wait %.0f + barrier %.0f + DMA issue %.0f + ds_read/MFMA section %.0f + prologue/epilogue %.0f  (MFMA-only floor per wave: %d)\n",
               name, cyc / us / 1e3, cyc / m, w / m, b / m, is / m, mf / m, (cyc - w - b - is - mf) / m, 64 * (K / 2) * (tile == 5 ? 4 : tile == 4 ? 2 : 1));
    };
    if (arith != AFM_ARITH_F32) {          // bf16-split kernels: the four trailing fields are phase spans in 10 ns ticks (gemm_split.hip)
        double a = 0, b = 0, c = 0, d = 0;
        for (auto& x : r) { a += x.wait_c; b += x.barrier_c; c += x.issue_c; d += x.mfma_c; }
        { double cyc = 0, us = 0; for (auto& x : r) { cyc += (double)(x.c1 - x.c0); us += (x.t1 - x.t0) * tick_us; }
          printf("   shader clock over the workgroups' lifetimes: %.2f GHz (cycle counter / s_memrealtime)\n", cyc / us / 1e3); }
        printf("   phases of thread 0, mean over workgroups (us): prologue (operand + statistic loads, first split) %.2f | K loop %.2f | K-group merge + staging %.2f | epilogue %.2f;  launch overhead = event time - (first entry -> last exit) = %.1f us\n",
               a / n * tick_us, b / n * tick_us, c / n * tick_us, d / n * tick_us, ms * 1e3 - (tmax - tmin) * tick_us);
    } else
    phases(0, first_round, "started in first 3 us");
    phases(first_round, n, "started later");
    const double span = (tmax - tmin) * tick_us, bucket = span / 24;
    printf("   resident workgroups over time (24 buckets of %.1f us):", bucket);
    for (int b = 0; b < 24; ++b) {
        const double t = (b + 0.5) * bucket;
        int c = 0;
        for (auto& x : r) if ((x.t0 - tmin) * tick_us <= t && (x.t1 - tmin) * tick_us > t) c++;
        printf(" %d", c);
    }
    printf("\n");
    CK(hipFree(dA)); CK(hipFree(dW)); CK(hipFree(dC)); CK(hipFree(dR)); CK(hipFree(drec));
    if (dS1) { CK(hipFree(dS1)); CK(hipFree(dS2)); CK(hipFree(dG)); }
}

int main(int argc, char** argv) {
    if (argc >= 4) { one(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 0, argc > 5 ? atoi(argv[5]) : AFM_ARITH_F32, argc > 6 ? atoi(argv[6]) : 0, 0, argc > 7 ? atoi(argv[7]) : 0); return 0; }
    if (argc == 2 && !strcmp(argv[1], "cold")) {       // the loop's situation: operands fresh from their producer
        g_cold = 1;
        for (int M : {1304, 326, 10432})
            for (int lf : {1, 2}) {
                if (lf == 1) { one(M, 512, 512, 0, AFM_ARITH_BF16X9, 0, 0, lf); one(M, 512, 1024, 0, AFM_ARITH_BF16X9, 0, 0, lf); }
                else { one(M, 1024, 512, 0, AFM_ARITH_BF16X9, 0, 0, lf); one(M, 1536, 512, 0, AFM_ARITH_BF16X9, 0, 0, lf); }
            }
        g_cold = 0;
        for (int lf : {1, 2}) one(10432, lf == 1 ? 512 : 1024, 512, 0, AFM_ARITH_BF16X9, 0, 0, lf);
        return 0;
    }
    if (argc == 2 && !strcmp(argv[1], "small")) {       // round 4: the small-launch regime (what each GPU runs under strong scaling), phase spans
        for (int M : {1304, 326})
            for (int lf : {0, 1, 2}) {
                one(M, 512, 512, 0, AFM_ARITH_BF16X9, 0, 0, lf);
                if (lf != 1) one(M, 1024, 512, 0, AFM_ARITH_BF16X9, 0, 0, lf);
                if (lf != 1) one(M, 1536, 512, 0, AFM_ARITH_BF16X9, 0, 0, lf);
                if (lf != 2) one(M, 512, 1024, 0, AFM_ARITH_BF16X9, 0, 0, lf);
            }
        return 0;
    }
    if (argc == 2 && !strcmp(argv[1], "r5")) {          // round 5: two-stage (7 / 3) against three-stage (10 / 9) forms, loop epilogues, warm and cold
        for (int cold : {0, 1}) {
            g_cold = cold;
            for (int M : {1304, 326}) {
                for (int tile : {7, 10}) { one(M, 512, 512, tile, AFM_ARITH_BF16X9, 0, 0, 1); one(M, 512, 1024, tile, AFM_ARITH_BF16X9, 0, 0, 1); }
                one(M, 512, 1024, 11, AFM_ARITH_BF16X9, 0, 0, 1);
                for (int tile : {7, 10, 9}) { one(M, 1024, 512, tile, AFM_ARITH_BF16X9, 0, 0, 2); one(M, 1536, 512, tile, AFM_ARITH_BF16X9, 0, 0, 2); }
            }
        }
        g_cold = 0;
        for (int M : {10432, 5216})
            for (int tile : {3, 9}) { one(M, 512, 512, tile, AFM_ARITH_BF16X9, 0, 0, 1); one(M, 512, 1024, tile, AFM_ARITH_BF16X9, 0, 0, 1); one(M, 1024, 512, tile, AFM_ARITH_BF16X9, 0, 0, 2); }
        return 0;
    }
    for (int tile : {3, 5}) one(10432, 512, 512, tile);
    one(10432, 512, 1024, 3);
    one(10432, 1536, 512, 3);
    one(6144, 512, 512, 3);          // exactly 768 tiles of 64x64: one full resident round (3 workgroups per CU)
    for (int tile : {1, 3}) one(1304, 512, 512, tile);
    for (int tile : {3, 5}) one(10432, 512, 512, tile, AFM_ARITH_BF16X9);
    for (int tile : {3, 5}) one(10432, 1536, 512, tile, AFM_ARITH_BF16X9);
    one(10432, 512, 1024, 3, AFM_ARITH_BF16X9);
    one(1304, 512, 512, 3, AFM_ARITH_BF16X9);
    return 0;
}
