// How many VALU instructions of the f32 -> 3 x bf16 split issue for free behind one v_mfma_f32_32x32x16_bf16, with one or two
// waves per SIMD?  Per loop trip: 4 independent MFMAs, each followed by NV split-style VALU instructions (pinned order).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t cvt(float a, float b) { f32x2 v = {a, b}; return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }

template <int NV, bool MF, bool PK>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    uint4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, 7, threadIdx.x};
    float r0[4], r1[4];
    for (int i = 0; i < 4; ++i) { r0[i] = in[threadIdx.x + i]; r1[i] = in[threadIdx.x + 4 + i]; }
    uint32_t sink = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MF) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[m], 0, 0, 0);
            // NV instructions: groups of {cvt, shl, and, sub, sub}
#pragma unroll
            for (int v = 0; v < NV; v += 5) {
                const uint32_t pk = cvt(r0[m], r1[m]);
                sink ^= pk;
                if (v + 1 < NV) {
                    const float lo = __uint_as_float(pk << 16);
                    if (v + 3 < NV) r0[m] -= lo; else sink ^= __float_as_uint(lo);
                }
                if (v + 2 < NV) {
                    const float hi = __uint_as_float(pk & 0xffff0000u);
                    if (v + 4 < NV) r1[m] -= hi; else sink ^= __float_as_uint(hi);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 4; ++i) s += r0[i] + r1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + __uint_as_float(sink);
}

template <int NV, bool MF>
void run(int blocks, float* out, const float* in) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, MF, false>), dim3(blocks), dim3(256), 0, 0, out, in, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, MF, false>), dim3(blocks), dim3(256), 0, 0, out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = blocks / 256.0;
    const double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0 * waves_per_simd);
    printf("NV=%2d mfma=%d waves/SIMD=%.0f : %.3f ms  %.1f cycles per (MFMA+VALU) slot per SIMD @2.4GHz\n", NV, (int)MF, waves_per_simd, ms, cyc);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&in, 4096); hipMemset(in, 0, 4096);
    for (int blocks : {256, 512, 1024}) {
        run<0, true>(blocks, out, in);  run<3, true>(blocks, out, in);  run<5, true>(blocks, out, in);
        run<10, true>(blocks, out, in); run<15, true>(blocks, out, in);
        run<5, false>(blocks, out, in); run<10, false>(blocks, out, in); run<15, false>(blocks, out, in);
    }
    return 0;
}
