// Sustained f32-MFMA ceiling probe (measurement tooling, not part of the product):
// every wave issues back-to-back v_mfma_f32_32x32x2_f32 on NACC independent accumulators, no memory traffic.
// usage: mfma_peak [seconds]   -> prints TFLOP/s for 1, 2 and 4 waves per SIMD, short and sustained.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

static double run(int blocks_per_cu, int iters, int reps) {
    float* d; hipMalloc(&d, 4);
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, d, 16, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * grid * 4 /*waves*/ * iters * 8.0 * 4 /*NACC*/ * (2.0 * 32 * 32 * 2);
    hipFree(d);
    return flops / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    for (int bpc : {1, 2, 4}) {
        const double burst = run(bpc, 2000, 1);                      // ~a few ms
        // sustained: repeat launches for ~secs seconds
        const double per = 2000.0 * 8 * 4 * 64 / 2.4e9;               // rough seconds per launch at full clock (1 wave/SIMD)
        int reps = (int)(secs / (per * bpc)); if (reps < 1) reps = 1;
        const double sus = run(bpc, 2000, reps);
        printf("f32 mfma 32x32x2: %d wave(s)/SIMD  burst %.1f TF  sustained(%d launches) %.1f TF  (spec 157.3)\n", bpc, burst, reps, sus);
    }
    return 0;
}
