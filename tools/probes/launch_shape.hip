// Round 4: what does a dependent launch cost on this box as a function of its SHAPE?  The small-launch GEMMs of the sampling loop spend
// 12-14 us between their first workgroup's entry and their last workgroup's exit (tools/gemm_timeline small) but occupy 21 us of the
// stream (rocprofv3): 7-9 us per launch are outside the kernel's own code.  This probe times chains of N back-to-back launches on one
// stream (wall time / N, stream drained before and after) for kernels that differ in one ingredient at a time: workgroup size, dynamic
// LDS, grid size, kernarg bytes, a body that loads and stores (plain / write-through sc1 stores), two kernels alternating.
//   hipcc -O2 --offload-arch=gfx950 tools/probes/launch_shape.hip -o tools/probes/launch_shape
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

struct Big { char b[560]; float* in; float* out; int n; };
struct Small { float* in; float* out; int n; };

// MODE 0: empty; 1: every thread loads one float4 and stores one float4 (plain); 2: the same with a write-through (sc1) store;
// 3: like 1 plus ~8 us of dependent ALU work per wave (a kernel with a real duration)
template <class Arg, int MODE>
__global__ void k(Arg a) {
    extern __shared__ float lds[];
    if (MODE == 0) return;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) % a.n;
    float4 v = reinterpret_cast<const float4*>(a.in)[i];
    if (MODE == 3) {
        float x = v.x;
        for (int it = 0; it < 4000; ++it) x = x * 1.0000001f + 0.5f;
        v.x = x;
    }
    lds[threadIdx.x] = v.y;
    __syncthreads();
    v.z += lds[threadIdx.x ^ 1];
    float4* p = reinterpret_cast<float4*>(a.out) + i;
    if (MODE == 2) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 d = {v.x, v.y, v.z, v.w};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
    } else *p = v;
}
template <class Arg, int MODE>
__global__ void k2(Arg a) {          // a second code object entry with the same body (alternating launches: instruction-cache effect)
    extern __shared__ float lds[];
    if (MODE == 0) return;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) % a.n;
    float4 v = reinterpret_cast<const float4*>(a.in)[i];
    lds[threadIdx.x] = v.y;
    __syncthreads();
    v.w += lds[threadIdx.x ^ 1];
    reinterpret_cast<float4*>(a.out)[i] = v;
}

template <class F>
static double chain_us(F&& f, int n, hipStream_t s) {
    for (int i = 0; i < 20; ++i) f(i);
    CK(hipStreamSynchronize(s));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f(i);
    CK(hipStreamSynchronize(s));
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}

template <class Arg, int MODE>
static void run(const char* what, int grid, int block, int lds, float* in, float* out, int n4, hipStream_t s, bool alternate = false) {
    CK(hipFuncSetAttribute((const void*)k<Arg, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)k2<Arg, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Arg a; memset(&a, 0, sizeof a); a.in = in; a.out = out; a.n = n4;
    const double us = chain_us([&](int i) {
        if (alternate && (i & 1)) hipLaunchKernelGGL((k2<Arg, MODE>), dim3(grid), dim3(block), lds, s, a);
        else hipLaunchKernelGGL((k<Arg, MODE>), dim3(grid), dim3(block), lds, s, a);
    }, 3000, s);
    printf("%-62s grid %4d x %4d threads, LDS %6d B, kernarg %3zu B: %6.2f us per launch\n", what, grid, block, lds, sizeof(Arg), us);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int n4 = 1 << 20;
    float *in, *out; CK(hipMalloc(&in, (size_t)n4 * 16)); CK(hipMalloc(&out, (size_t)n4 * 16));
    CK(hipMemset(in, 0, (size_t)n4 * 16));
    printf("-- empty body\n");
    run<Small, 0>("empty, small launch", 1, 64, 0, in, out, n4, s);
    run<Small, 0>("empty", 168, 256, 0, in, out, n4, s);
    run<Small, 0>("empty, 512-thread workgroups", 168, 512, 0, in, out, n4, s);
    run<Small, 0>("empty, 1024-thread workgroups", 168, 1024, 0, in, out, n4, s);
    run<Small, 0>("empty, 512 threads + 74 KB LDS", 168, 512, 74 * 1024, in, out, n4, s);
    run<Small, 0>("empty, 1024 threads + 148 KB LDS", 168, 1024, 148 * 1024, in, out, n4, s);
    run<Big, 0>("empty, 512 threads + 74 KB LDS + 584-byte kernarg", 168, 512, 74 * 1024, in, out, n4, s);
    run<Big, 0>("empty, 504 workgroups (in_proj at M = 1304)", 504, 512, 74 * 1024, in, out, n4, s);
    run<Big, 0>("empty, 48 workgroups (M = 326)", 48, 512, 74 * 1024, in, out, n4, s);
    run<Big, 0>("empty, 1304 workgroups of 256 (B = 32 form)", 1304, 256, 37 * 1024, in, out, n4, s);
    printf("-- load + store body (every thread one float4 in, one out)\n");
    run<Big, 1>("plain stores", 168, 512, 74 * 1024, in, out, n4, s);
    run<Big, 2>("write-through (sc1) stores", 168, 512, 74 * 1024, in, out, n4, s);
    run<Big, 1>("plain stores, two kernels alternating", 168, 512, 74 * 1024, in, out, n4, s, true);
    run<Big, 1>("plain stores, 504 workgroups", 504, 512, 74 * 1024, in, out, n4, s);
    run<Big, 2>("sc1 stores, 504 workgroups", 504, 512, 74 * 1024, in, out, n4, s);
    run<Big, 1>("plain stores, 1304 x 256 (B = 32 form)", 1304, 256, 37 * 1024, in, out, n4, s);
    run<Big, 2>("sc1 stores, 1304 x 256 (B = 32 form)", 1304, 256, 37 * 1024, in, out, n4, s);
    printf("-- ~8 us of ALU work per wave + load + store\n");
    run<Big, 3>("work + plain stores", 168, 512, 74 * 1024, in, out, n4, s);
    run<Small, 3>("work + plain stores, 256 threads, no LDS, small kernarg", 168, 256, 0, in, out, n4, s);
    run<Small, 3>("work, one workgroup of 64", 1, 64, 0, in, out, n4, s);
    return 0;
}
