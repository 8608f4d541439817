// Per-workgroup phase timeline of the attention kernel (library source compiled with -DAFM_TIMELINE; instrumentation never ships):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAFM_TIMELINE -Iinclude -Iafford-motion_amd/csrc tools/mha_timeline.hip afford-motion_amd/csrc/profile.hip -o tools/mha_timeline
#include "../afford-motion_amd/csrc/attention.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static void one(int B, int g) {
    const int T = 326, H = 8, dh = 64, D = H * dh;
    float *dq, *dout;
    CK(hipMalloc(&dq, (size_t)B * T * 3 * D * 4)); CK(hipMalloc(&dout, (size_t)B * T * D * 4));
    { std::vector<float> h((size_t)B * T * 3 * D); unsigned x = 12345u; for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((x >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1.7f; } CK(hipMemcpy(dq, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }   // uniform in (-1.7, 1.7): unit variance
    const int maxwg = B * H * 11;
    AfmMhaRec* drec;
    CK(hipMalloc(&drec, (size_t)maxwg * sizeof(AfmMhaRec)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(afm_mha_timeline), &drec, sizeof drec));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) afm_mha_fwd_grouped(dq, nullptr, dout, B, T, H, dh, g, st);
    CK(hipStreamSynchronize(st));
    CK(hipMemset(drec, 0, (size_t)maxwg * sizeof(AfmMhaRec)));
    CK(hipEventRecord(e0, st));
    afm_mha_fwd_grouped(dq, nullptr, dout, B, T, H, dh, g, st);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<AfmMhaRec> rec(maxwg), r;
    CK(hipMemcpy(rec.data(), drec, (size_t)maxwg * sizeof(AfmMhaRec), hipMemcpyDeviceToHost));
    for (auto& x : rec) if (x.t1) r.push_back(x);
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto& x : r) { tmin = std::min(tmin, x.t0); tmax = std::max(tmax, x.t1); }
    double cyc = 0, us = 0, s = 0, so = 0, pv = 0, sy = 0;
    std::vector<double> dur;
    for (auto& x : r) { cyc += (double)(x.c1 - x.c0); us += (x.t1 - x.t0) * 0.01; s += x.s_c; so += x.soft_c; pv += x.pv_c; sy += x.sync_c; dur.push_back((x.t1 - x.t0) * 0.01); }
    std::sort(dur.begin(), dur.end());
    const int n = (int)r.size();
    printf("== B=%d group_waves=%d: %d workgroups, event %.1f us, span %.1f us, %.1f TF; workgroup duration min %.1f p50 %.1f max %.1f us; clock %.2f GHz\n", B, g, n, ms * 1e3,
           (tmax - tmin) * 0.01, 4.0 * B * H * (double)T * T * dh / (ms * 1e-3) / 1e12, dur.front(), dur[n / 2], dur.back(), cyc / us / 1e3);
    printf("   wave 0 per workgroup: %.0f cycles = S phase (load issue + 12 ds_read_b128 + 36 MFMA) %.0f + softmax %.0f + PV (P split + 12 ds_read_b128 + 36 MFMA) %.0f + split/store/barrier %.0f + rest %.0f;  MFMA floor per wave: %d\n",
           cyc / n, s / n, so / n, pv / n, sy / n, (cyc - s - so - pv - sy) / n, 11 * 72 * 32);
    CK(hipFree(dq)); CK(hipFree(dout)); CK(hipFree(drec));
}

int main(int argc, char** argv) {
    if (argc > 1) {          // round 4: the small-launch forms (104 / 102 = the two key segments of a query block on two waves)
        for (int B : {4, 1}) for (int g : {4, 104, 2, 102, 1}) one(B, g);
        one(32, 12);
        return 0;
    }
    for (int g : {4, 6, 12, 2}) one(32, g);
    one(16, 4);
    one(4, 4);
    return 0;
}
