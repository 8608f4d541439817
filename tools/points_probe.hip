// Stand-alone timing of the point-cloud kernels (FPS / kNN) at BASELINE configs[3]'s shapes, compiled from the library's own source with
// -DAFM_PROBE (adds a threads-per-workgroup override for FPS experiments; never part of libafm_hip.so):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAFM_PROBE -Iinclude -Iafford-motion_amd/csrc tools/points_probe.hip afford-motion_amd/csrc/profile.hip -o tools/points_probe
#include "../afford-motion_amd/csrc/pointops.hip"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

int main() {
    const int B = 32, n = 8192;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> U(-2.f, 2.f);
    std::vector<float> p((size_t)B * n * 3);
    for (auto& v : p) v = U(rng);
    float *dp, *dq, *dd;
    int *didx, *dk;
    CK(hipMalloc(&dp, p.size() * 4)); CK(hipMalloc(&dq, (size_t)B * 2048 * 3 * 4)); CK(hipMalloc(&didx, (size_t)B * 2048 * 4));
    CK(hipMalloc(&dk, (size_t)B * n * 16 * 4)); CK(hipMalloc(&dd, (size_t)B * n * 16 * 4));
    CK(hipMemcpy(dp, p.data(), p.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto&& f) {
        f(); CK(hipStreamSynchronize(st));
        std::vector<float> ts;
        for (int i = 0; i < 5; ++i) { CK(hipEventRecord(e0, st)); f(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms); }
        std::sort(ts.begin(), ts.end());
        return ts[2];
    };
    std::vector<int> ref((size_t)B * 2048), got((size_t)B * 2048);
    for (int m : {2048, 1024}) {
        {   // exact pruning of the scan (fps_pruned_kernel, the default at this size) against the plain kernel below
            afm_probe_fps_threads = 0;
            for (int PT : {1024, 512, 256}) {
                afm_probe_fps_prune = PT;
                const float ms = time([&] { int rc = afm_fps(dp, B, n, m, didx, st); if (rc) { printf("fps rc=%d\n", rc); } });
                CK(hipMemcpy(got.data(), didx, (size_t)B * m * 4, hipMemcpyDeviceToHost));
                if (PT == 1024) ref = got;
                printf("fps  B=%d n=%d m=%d pruned scan, %4d threads: %7.3f ms  %.3f us/round  %s\n", B, n, m, PT, ms, 1e3 * ms / (m - 1),
                       std::equal(got.begin(), got.begin() + (size_t)B * m, ref.begin()) ? "same indices" : "DIFFERENT");
            }
            afm_probe_fps_prune = 0;
        }
        for (int T : {0, 1024, 512, 256}) {
            afm_probe_fps_threads = T;
            const float ms = time([&] { int rc = afm_fps(dp, B, n, m, didx, st); if (rc) { printf("fps rc=%d\n", rc); } });
            CK(hipMemcpy(got.data(), didx, (size_t)B * m * 4, hipMemcpyDeviceToHost));
            printf("fps  B=%d n=%d m=%d threads=%4d: %7.3f ms  %.3f us/round  %s\n", B, n, m, T, ms, 1e3 * ms / (m - 1), std::equal(got.begin(), got.begin() + (size_t)B * m, ref.begin()) ? "same indices" : "DIFFERENT");
        }
        afm_probe_fps_threads = 0;
        afm_probe_fps_prune = -1;
#ifdef AFM_PROBE_TIMELINE
        {   // phase timeline of the default shape: summed cycles of wave 0 of workgroup 0 between the stamps (s_memtime: 100 MHz x ... no: shader clock)
            int rc = afm_fps(dp, B, n, m, didx, st); (void)rc;
            CK(hipStreamSynchronize(st));
            unsigned long long cyc[8];
            CK(hipMemcpyFromSymbol(cyc, HIP_SYMBOL(afm_probe_fps_cyc), sizeof(cyc)));
            const char* nm[6] = {"coords (ds_read)", "scan", "wave arg-max", "LDS hop + barrier", "cross-wave arg-max", "loop overhead / store"};
            double tot = 0; for (int k = 0; k < 6; ++k) tot += (double)cyc[k];
            printf("fps  phase timeline m=%d (counter ticks per round, wave 0 of workgroup 0; stamps cost ~5 x s_memtime):", m);
            for (int k = 0; k < 6; ++k) printf("  %s %.0f", nm[k], (double)cyc[k] / (m - 1));
            printf("  | total %.0f\n", tot / (m - 1));
        }
#endif
        CK(hipMemcpy(got.data(), didx, (size_t)B * m * 4, hipMemcpyDeviceToHost));
        afm_gather_rows(dp, didx, dq, (int64_t)B * m, 3, st);
        const float ms = time([&] { afm_knn(16, dp, dq, B, n, m, dk, dd, st); });
        printf("knn  B=%d n=%d m=%d k=16: %7.3f ms  (%.1f G pair evaluations/s)\n", B, n, m, ms, (double)B * m * n / (ms * 1e-3) / 1e9);
    }
    const float ms8 = time([&] { afm_knn(8, dp, dp, B, n, n, dk, dd, st); });
    printf("knn  B=%d n=%d m=%d k=8 (self): %7.3f ms  (%.1f G pair evaluations/s)\n", B, n, n, ms8, (double)B * n * n / (ms8 * 1e-3) / 1e9);
    return 0;
}
