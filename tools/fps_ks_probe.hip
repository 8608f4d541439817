// Rounds and time of fps_pruned_kernel for one value of AFM_FPS_KS (samples per round), on a scene cloud dumped by Python (float32 [B][n][3]):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAFM_PROBE -DAFM_FPS_KS=4 -Iinclude -Iafford-motion_amd/csrc tools/fps_ks_probe.hip afford-motion_amd/csrc/profile.hip -o tools/fps_ks_probe_4
//   tools/fps_ks_probe_4 /tmp/cloud.bin 32 8192
#include "../afford-motion_amd/csrc/pointops.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
int main(int argc, char** argv) {
    printf("AFM_FPS_CELLS=%d\n", AFM_FPS_CELLS);
    const int B = atoi(argv[2]), n = atoi(argv[3]);
    std::vector<float> p((size_t)B * n * 3);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(p.data(), 4, p.size(), f) != p.size()) { printf("cannot read %s\n", argv[1]); return 2; }
    fclose(f);
    float* dp; int* didx;
    CK(hipMalloc(&dp, p.size() * 4)); CK(hipMalloc(&didx, (size_t)B * n * 4));
    CK(hipMemcpy(dp, p.data(), p.size() * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int m : {2048, 1024}) {
        std::vector<float> ts;
        for (int i = 0; i < 6; ++i) {
            CK(hipEventRecord(e0, st)); int rc = afm_fps(dp, B, n, m, didx, st); if (rc) printf("rc %d\n", rc);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        int rounds[4] = {0, 0, 0, 0};
        CK(hipMemcpyFromSymbol(rounds, HIP_SYMBOL(afm_probe_fps_rounds), sizeof(rounds)));
        std::vector<int> got((size_t)B * m);
        CK(hipMemcpy(got.data(), didx, got.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long h = 1469598103934665603ull;
        for (int v : got) { h ^= (unsigned)v; h *= 1099511628211ull; }
        printf("KS=%d m=%d: %.3f ms, rounds of sample 0: %d of %d (%.3f us per round), index hash %016llx\n", AFM_FPS_KS, m, ts[2], rounds[0], m - 1,
               rounds[0] ? 1e3 * ts[2] / rounds[0] : 0.0, h);
    }
    return 0;
}
