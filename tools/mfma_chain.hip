// f32 MFMA issue-rate probe: NACC independent accumulators per wave, W waves per SIMD, optional LDS operand reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[64 * 36];
    for (int i = threadIdx.x; i < 64 * 36; i += 256) sm[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int lane = threadIdx.x & 63;
    const float* ap = sm + (lane & 31) * 36 + (lane >> 5) * 16;
    float a = 1.0f + lane * 1e-6f, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 av, bv;
            if (LDS) {
                av = *reinterpret_cast<const float4*>(ap + j * 4);
                bv = *reinterpret_cast<const float4*>(ap + 32 * 36 + j * 4);
            } else {
                av = make_float4(a, a, a, a); bv = make_float4(b, b, b, b);
            }
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i], 0, 0, 0);
            }
        }
        if (LDS) asm volatile("" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}

template <int NACC, bool LDS>
double run(int wg_per_cu, int iters) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(256 * wg_per_cu), dim3(256), 0, 0, d, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL((k<NACC, LDS>), dim3(256 * wg_per_cu), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return 20.0 * 256 * wg_per_cu * 4 * (double)iters * 16 * NACC * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12;
}

int main() {
    for (int w : {1, 2, 4}) {
        printf("waves/SIMD %d | regs: 1 acc %.1f  2 acc %.1f  4 acc %.1f TF | LDS operands: 1 acc %.1f  2 acc %.1f  4 acc %.1f TF\n", w,
               run<1, false>(w, 2000), run<2, false>(w, 1000), run<4, false>(w, 500), run<1, true>(w, 2000), run<2, true>(w, 1000), run<4, true>(w, 500));
    }
    return 0;
}
