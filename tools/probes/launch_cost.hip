// Host cost of a launch on this box: an empty kernel, an empty kernel with a 400-byte by-value argument, afm_linear (tiny GEMM) and
// afm_layernorm through the C-ABI; enqueue time only (the stream is drained between measurements).
//   hipcc -O2 --offload-arch=gfx950 -Iinclude tools/probes/launch_cost.hip -Lafford-motion_amd/afm -lafm_hip -Wl,-rpath,'$ORIGIN/../../afford-motion_amd/afm' -o tools/probes/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include "afm_hip.h"
struct Big { char b[400]; };
__global__ void k_empty() {}
__global__ void k_big(Big) {}
template <class F>
static double per_launch_us(F&& f, int n, hipStream_t s) {
    for (int i = 0; i < 50; ++i) f();
    hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) f();
    const auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
    hipStream_t s, s2; hipStreamCreate(&s); hipStreamCreate(&s2);
    float *A, *W, *C; hipMalloc(&A, 64 * 512 * 4); hipMalloc(&W, 64 * 512 * 4); hipMalloc(&C, 64 * 64 * 4);
    hipMemset(A, 0, 64 * 512 * 4); hipMemset(W, 0, 64 * 512 * 4);
    afm_linear_args a; memset(&a, 0, sizeof a);
    a.A = A; a.lda = 512; a.W = W; a.ldw = 512; a.C = C; a.ldc = 64; a.M = 64; a.N = 64; a.K = 512;
    Big big; memset(&big, 0, sizeof big);
    const int n = 2000;
    printf("empty kernel                 %.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); }, n, s));
    printf("empty kernel, 400-byte arg   %.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, big); }, n, s));
    printf("afm_linear (64 x 64 x 512)   %.2f us per launch\n", per_launch_us([&] { afm_linear(&a, s); }, n, s));
    int flip = 0;
    printf("empty kernel, two streams alternating  %.2f us per launch\n", per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, (flip++ & 1) ? s2 : s); }, n, s));
    hipStreamSynchronize(s2);
    return 0;
}
