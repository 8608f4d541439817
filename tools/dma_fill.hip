// L2 -> CU fill-rate microbenchmark with the GEMM's operand access pattern and NO matrix work: what does one CU sustain when 1 / 2 / 3 / 4
// workgroups of 4 waves each stream a 64-row A panel and a 64-row W panel K-tile by K-tile (32 floats per row per tile)?
//   mode 0: global_load_lds_dwordx4, 8 rows x 128 B per wave-instruction (the GEMM's pattern: row stride = K floats)
//   mode 1: the same addresses with global_load_dwordx4 into registers (+ ds_write_b128)
//   mode 2: global_load_lds_dwordx4 over a PRE-TILED operand: every wave-instruction reads 1 KiB contiguous
//   hipcc --offload-arch=gfx950 -O3 tools/dma_fill.hip -o tools/dma_fill && tools/dma_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int MODE, int LDS_KB>
__global__ __launch_bounds__(256) void fill_kernel(const float* __restrict__ A, const float* __restrict__ W, int M, int N, int K, int nbn, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_KB * 256];
    const int tid = threadIdx.x, wave = tid >> 6, c8 = tid & 7, r0 = tid >> 3;
    const int bm = blockIdx.x / nbn, bn = blockIdx.x % nbn;
    const float* src[4];
    for (int i = 0; i < 2; ++i) {
        const int ra = min(bm * 64 + r0 + 32 * i, M - 1), rw = min(bn * 64 + r0 + 32 * i, N - 1);
        if (MODE == 2) {        // pre-tiled: [row block of 8][k-tile][8 rows][32 floats]: a wave's 8 rows x 128 B are one contiguous KiB
            src[i] = A + ((size_t)(ra >> 3) * (K / 32)) * 256 + (ra & 7) * 32 + c8 * 4;
            src[2 + i] = W + ((size_t)(rw >> 3) * (K / 32)) * 256 + (rw & 7) * 32 + c8 * 4;
        } else {
            src[i] = A + (size_t)ra * K + c8 * 4;
            src[2 + i] = W + (size_t)rw * K + c8 * 4;
        }
    }
    const int kstep = MODE == 2 ? 256 : 32;
    float4 acc = make_float4(0, 0, 0, 0);
    const int nk = K / 32;
    for (int kt = 0; kt < nk; ++kt) {
        float* dst = lds + (kt % 3) * 4096 + wave * 1024;
        if (MODE == 1) {
            float4 v[4];
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4*>(src[i] + kt * kstep);
            for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(dst + i * 256 + (tid & 63) * 4) = v[i];
        } else {
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * kstep), (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
        }
        if (kt >= 2) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc.x = lds[tid];
    if (acc.x == 123.456f) sink[0] = acc.x;
}

template <int MODE, int LDS_KB>
static void run(const char* name, const float* dA, const float* dW, int M, int N, int K, float* sink, hipStream_t st) {
    const int nbm = (M + 63) / 64, nbn = N / 64;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill_kernel<MODE, LDS_KB>), dim3(nbm * nbn), dim3(256), 0, st, dA, dW, M, N, K, nbn, sink);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((fill_kernel<MODE, LDS_KB>), dim3(nbm * nbn), dim3(256), 0, st, dA, dW, M, N, K, nbn, sink);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, bytes = (double)nbm * nbn * (K / 32) * 16384.0;
    printf("%-58s LDS %3d KB/WG (%d WG/CU): %7.1f us  %6.2f TB/s L2->LDS = %5.1f B/clk/CU at 2.3 GHz  (GEMM at this rate: %5.1f TF)\n", name, LDS_KB, 160 / LDS_KB, us,
           bytes / (us * 1e-6) / 1e12, bytes / (us * 1e-6) / 256 / 2.3e9, 2.0 * M * N * K / (us * 1e-6) / 1e12);
}

int main() {
    const int M = 10432, N = 512;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    float *dA, *dW, *sink;
    CK(hipMalloc(&dA, (size_t)M * 1024 * 4 + 4096)); CK(hipMalloc(&dW, (size_t)N * 1024 * 4 + 4096)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(dA, 0, (size_t)M * 1024 * 4)); CK(hipMemset(dW, 0, (size_t)N * 1024 * 4));
    for (int K : {512, 1024}) {
        printf("M=%d N=%d K=%d, 64x64 tiles, 16 KiB per workgroup per K-tile\n", M, N, K);
        run<0, 48>("LDS-DMA, 8 rows x 128 B per instruction", dA, dW, M, N, K, sink, st);
        run<0, 36>("LDS-DMA, 8 rows x 128 B per instruction", dA, dW, M, N, K, sink, st);
        run<0, 80>("LDS-DMA, 8 rows x 128 B per instruction", dA, dW, M, N, K, sink, st);
        run<1, 48>("global_load_dwordx4 -> VGPR -> ds_write_b128", dA, dW, M, N, K, sink, st);
        run<1, 36>("global_load_dwordx4 -> VGPR -> ds_write_b128", dA, dW, M, N, K, sink, st);
        run<2, 48>("LDS-DMA, pre-tiled operands (1 KiB contiguous per instr)", dA, dW, M, N, K, sink, st);
        run<2, 36>("LDS-DMA, pre-tiled operands (1 KiB contiguous per instr)", dA, dW, M, N, K, sink, st);
    }
    return 0;
}
