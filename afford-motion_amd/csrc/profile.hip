// Opt-in HIP-event profiler: brackets every kernel launch of the library with a pair of events on
// the launch stream and accumulates (count, milliseconds, algorithmic work) per kernel tag.
// Used by bench.py for the `roofline` object (average launch duration of the dominant kernel,
// measured on the stream the kernel runs on); disabled (zero overhead beyond one branch) otherwise.
#include <mutex>
#include <vector>
#include "common.h"
#include "profile.h"

namespace {
struct Rec { int tag; double work; hipEvent_t a, b; };
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_free;
const char* kNames[AFM_PROF_NTAGS] = {"gemm_f32_mfma<128,128>", "gemm_f32_mfma<64,128>", "gemm_f32_mfma<64,64>", "mha_fwd_kernel",
                                      "layernorm_kernel", "misc (ddpm, randn, thin linears)", "fps_kernel", "knn_kernel",
                                      "transition_down_kernel", "pt_attention_kernel", "cdm_perceiver", "gemm_f32_mfma_dma<128,128>",
                                      "gemm_f32_mfma_dma<64,128>", "gemm_f32_mfma_dma<64,64>", "wgrad_kernel", "layernorm_bwd_kernel",
                                      "mha_bwd_dq_kernel", "mha_bwd_dkv_kernel", "train_misc", "point_train_passes", "wgrad_skinny_kernel", "cdm_train_attention", "gemm_f32_split_bf16<128, 128>", "gemm_f32_split_bf16<64, 64>", "gemm_f32_mfma_dma<32,32>", "gemm_f32_mfma_dma<32,64>", "gemm_f32_split_bf16<64, 64, split-K>", "gemm_f32_split_rowdot_slab", "mha_fwd_split_kernel",
                                      "enc_point_kernel", "dec_point_kernel", "cdm_latent_chain (lat_head + toklin x 11 + lat_dectables)"};
}  // namespace

bool afm_prof_on() { return g_on; }

void afm_prof_begin(int tag, double work, hipStream_t s, void** handle) {
    *handle = nullptr;
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r; r.tag = tag; r.work = work;
    auto get = [&]() { hipEvent_t e; if (!g_free.empty()) { e = g_free.back(); g_free.pop_back(); } else { (void)hipEventCreate(&e); } return e; };
    r.a = get(); r.b = get();
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    *handle = (void*)(uintptr_t)g_recs.size();
}

void afm_prof_end(void* handle, hipStream_t s) {
    if (!handle) return;
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t i = (size_t)(uintptr_t)handle - 1;
    if (i < g_recs.size()) (void)hipEventRecord(g_recs[i].b, s);
}

extern "C" int afm_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return 0;
}

extern "C" int afm_profile_read(afm_profile_entry* out, int32_t max_entries) {
    if (!out || max_entries < 0) return AFM_E_BADARG;
    std::lock_guard<std::mutex> lk(g_mu);
    double ms[AFM_PROF_NTAGS] = {0}, work[AFM_PROF_NTAGS] = {0};
    int64_t cnt[AFM_PROF_NTAGS] = {0};
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.b);
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.tag] += t; work[r.tag] += r.work; cnt[r.tag]++; }
        g_free.push_back(r.a); g_free.push_back(r.b);
    }
    g_recs.clear();
    int n = 0;
    for (int t = 0; t < AFM_PROF_NTAGS && n < max_entries; ++t) {
        if (!cnt[t]) continue;
        out[n].name = kNames[t]; out[n].launches = cnt[t]; out[n].total_ms = ms[t]; out[n].total_work = work[t];
        ++n;
    }
    return n;
}
