// What does a wave pay for VALU work BETWEEN the MFMAs of one accumulator chain?  (round 5, the small-launch K loop: 9 v_mfma_f32_32x32x16_bf16 on ONE
// accumulator per K-tile with ~5-10 VALU instructions of operand split between them, two waves per SIMD.)
//   chains = 1: every MFMA depends on the previous one;  chains = 2: two accumulators alternate (same MFMA count)
//   fill   = VALU instructions (v_fma_f32 on private registers) between consecutive MFMAs
//   waves  = waves per SIMD (workgroups of 256 threads x waves, one workgroup per CU)
// Prints shader cycles per MFMA per SIMD (32 = the pipe's issue rate).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_dep_gap.hip -o tools/probes/mfma_dep_gap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, int FILL>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f + i); b[i] = (__bf16)(0.5f + i); }
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = 1.0f + threadIdx.x * 1e-6f * i;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 18; ++m) {
            acc[CHAINS == 2 ? (m & 1) : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[CHAINS == 2 ? (m & 1) : 0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < FILL; ++v) f[v & 7] = __builtin_fmaf(f[v & 7], 1.0001f, f[(v + 1) & 7]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS, int FILL>
static void run(int waves_per_simd) {
    const int iters = 2000, blocks = 256, threads = 256 * waves_per_simd;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, blocks * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<CHAINS, FILL>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= 256;
    // s_memtime / readcyclecounter counts at a fixed 100 MHz on gfx950?  report raw ticks per MFMA and let the caller compare rows
    printf("chains %d  fill %2d  waves/SIMD %d : %.1f ticks per MFMA issued on a SIMD (wave-level: %.1f per MFMA of a wave)\n", CHAINS, FILL, waves_per_simd,
           mean / (iters * 18.0 * waves_per_simd), mean / (iters * 18.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2}) {
        run<1, 0>(w); run<2, 0>(w);
        run<1, 3>(w); run<2, 3>(w);
        run<1, 6>(w); run<2, 6>(w);
        run<1, 10>(w); run<2, 10>(w);
    }
    return 0;
}
