#pragma once
#include <hip/hip_runtime.h>
enum {
    AFM_PROF_GEMM128 = 0, AFM_PROF_GEMM64x128, AFM_PROF_GEMM64, AFM_PROF_MHA, AFM_PROF_LN, AFM_PROF_MISC, AFM_PROF_FPS,
    AFM_PROF_KNN, AFM_PROF_TD, AFM_PROF_PTATTN, AFM_PROF_CDM, AFM_PROF_GEMM128_DMA, AFM_PROF_GEMM64x128_DMA, AFM_PROF_GEMM64_DMA,
    AFM_PROF_WGRAD, AFM_PROF_LN_BWD, AFM_PROF_MHA_BWD_DQ, AFM_PROF_MHA_BWD_DKV, AFM_PROF_TRAIN_MISC, AFM_PROF_POINT_TRAIN, AFM_PROF_WGRAD_SKINNY, AFM_PROF_CDM_TRAIN,
    AFM_PROF_GEMM_SPLIT128, AFM_PROF_GEMM_SPLIT64, AFM_PROF_GEMM32_DMA, AFM_PROF_GEMM32x64_DMA, AFM_PROF_GEMM_SPLIT64_KG, AFM_PROF_GEMM_SLAB, AFM_PROF_MHA_SPLIT,
    AFM_PROF_CDM_ENC, AFM_PROF_CDM_DEC, AFM_PROF_CDM_CHAIN,
    AFM_PROF_NTAGS
};
bool afm_prof_on();
void afm_prof_begin(int tag, double work, hipStream_t s, void** handle);
void afm_prof_end(void* handle, hipStream_t s);
// RAII bracket around one launch
struct AfmProf {
    void* h; hipStream_t s;
    AfmProf(int tag, double work, hipStream_t st) : h(nullptr), s(st) { if (afm_prof_on()) afm_prof_begin(tag, work, st, &h); }
    ~AfmProf() { if (h) afm_prof_end(h, s); }
};
