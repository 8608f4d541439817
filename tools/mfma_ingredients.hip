// Which ingredient of the GEMM K-loop costs MFMA throughput?  One 32x32 tile per wave, 16 MFMAs per "K-tile",
// 4 waves per workgroup, WG workgroups per CU; ingredients toggled by template flags.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDK = 36;

template <bool BAR, bool LDSW, bool GLOAD, bool SEL>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ g, float* out, int iters, int K) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 128 * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 2 * 128 * LDK; i += 256) sm[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int c4 = tid & 7, r0 = tid >> 3;
    const float* rowp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) rowp[i] = g + (size_t)(((blockIdx.x * 64 + r0 + 32 * (i & 1)) % 4096) + (i >> 1) * 4096) * K;
    float4 reg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) reg[i] = make_float4(1, 2, 3, 4);
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        const int k = (it * 32 + c4 * 4) % K;
        if (GLOAD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) reg[i] = *reinterpret_cast<const float4*>(rowp[i] + k);
        }
        const float* ap = sm + cur * 128 * LDK + (wm * 32 + (lane & 31)) * LDK + (lane >> 5) * 16;
        const float* bp = sm + cur * 128 * LDK + 64 * LDK + (wn * 32 + (lane & 31)) * LDK + (lane >> 5) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 av = *reinterpret_cast<const float4*>(ap + j * 4);
            const float4 bv = *reinterpret_cast<const float4*>(bp + j * 4);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
        if (LDSW) {
            float* dst = sm + (cur ^ 1) * 128 * LDK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 v = reg[i];
                if (SEL) { const bool ok = k < K - 1; v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f; }
                *reinterpret_cast<float4*>(dst + (r0 + 32 * i) * LDK + c4 * 4) = v;
            }
        } else {
            asm volatile("" ::"v"(reg[0].x), "v"(reg[1].x), "v"(reg[2].x), "v"(reg[3].x));
        }
        if (BAR) __syncthreads();
        else asm volatile("" ::: "memory");
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 123.456f) out[0] = s;
}

template <bool BAR, bool LDSW, bool GLOAD, bool SEL>
double run(const float* g, int wg_per_cu, int iters) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<BAR, LDSW, GLOAD, SEL>), dim3(256 * wg_per_cu), dim3(256), 0, 0, g, d, 8, 512);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<BAR, LDSW, GLOAD, SEL>), dim3(256 * wg_per_cu), dim3(256), 0, 0, g, d, iters, 512);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return 10.0 * 256 * wg_per_cu * 4 * (double)iters * 16 * (2.0 * 32 * 32 * 2) / (ms * 1e-3) / 1e12;
}

int main() {
    float* g; hipMalloc(&g, (size_t)8192 * 512 * 4); hipMemset(g, 0, (size_t)8192 * 512 * 4);
    for (int w : {1, 2, 4}) {
        printf("WG/CU %d: reads only %.1f | +barrier %.1f | +ldswrite %.1f | +ldswrite+barrier %.1f | +gload %.1f | +gload+ldsw+bar %.1f | +select %.1f TF\n", w,
               run<false, false, false, false>(g, w, 4000), run<true, false, false, false>(g, w, 4000), run<false, true, false, false>(g, w, 4000),
               run<true, true, false, false>(g, w, 4000), run<false, false, true, false>(g, w, 4000), run<true, true, true, false>(g, w, 4000),
               run<true, true, true, true>(g, w, 4000));
    }
    return 0;
}
