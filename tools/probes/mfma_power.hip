// What does the bf16 matrix pipe cost in board power?  (measurement tooling, not part of the product)
// Every wave issues v_mfma_f32_32x32x16_bf16 on four independent accumulators with pseudo-random operand bits and no memory traffic;
// `gap` s_sleep units (64 cycles each) between groups of 36 MFMAs set the duty cycle.  Runs each setting for `secs` seconds and prints the
// achieved rate; tools/gpu_power_mfma.sh samples rocm-smi next to it.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/mfma_power.hip -o tools/probes/mfma_power
//   tools/probes/mfma_power <waves_per_simd> <gap> <secs>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int GAP>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    u32x4 a[3], b[3];                                       // three planes per operand, as in the split kernels
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h = h * 1664525u + 1013904223u; a[p][j] = (h & 0x807f807fu) | 0x3f003f00u;        // bf16 pairs in [0.5, 2), random mantissas / signs
            h = h * 1664525u + 1013904223u; b[p][j] = (h & 0x807f807fu) | 0x3f003f00u;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q / 3]), __builtin_bit_cast(bf16x8, b[q % 3]), acc[i], 0, 0, 0);
        if (GAP > 0) __builtin_amdgcn_s_sleep(GAP);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

template <int GAP>
static void run(int wps, double secs) {
    float* d; hipMalloc(&d, 4);
    const int grid = 256 * wps, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<GAP>, dim3(grid), dim3(256), 0, 0, d, 16, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<GAP>, dim3(grid), dim3(256), 0, 0, d, iters, 2u);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms1; hipEventElapsedTime(&ms1, e0, e1);
    int reps = (int)(secs * 1e3 / ms1); if (reps < 1) reps = 1;
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<GAP>, dim3(grid), dim3(256), 0, 0, d, iters, 3u + r);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double)reps * grid * 4 * iters * 36.0 * (32 * 32 * 16);
    printf("bf16 32x32x16: %d wave(s)/SIMD gap %d: %.0f TFLOP/s = %.3e MAC/s over %.1f s (pipe busy %.2f at 2.4 GHz-equivalent)\n", wps, GAP, 2 * macs / (ms * 1e-3) / 1e12,
           macs / (ms * 1e-3), ms * 1e-3, macs / (ms * 1e-3) / (1024.0 * 512 * 2.4e9));
    hipFree(d);
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2, gap = argc > 2 ? atoi(argv[2]) : 0;
    const double secs = argc > 3 ? atof(argv[3]) : 4.0;
    switch (gap) {
        case 0: run<0>(wps, secs); break;
        case 9: run<9>(wps, secs); break;
        case 18: run<18>(wps, secs); break;
        case 36: run<36>(wps, secs); break;
        case 72: run<72>(wps, secs); break;
        default: fprintf(stderr, "gap must be 0, 9, 18, 36 or 72\n"); return 2;
    }
    return 0;
}
