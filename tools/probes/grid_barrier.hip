// Cost of an in-launch grid barrier on MI355X (round 3): G resident workgroups x NB barriers back to back, the barrier of
// csrc/perceiver.hip::chain_barrier (write-through payload stores, vmcnt drain, one relaxed agent-scope ticket + relaxed poll, one
// agent-scope acquire) and variants.   hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier.hip -o tools/probes/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

template <int SLEEP, bool ACQ, bool PAYLOAD>
__global__ __launch_bounds__(256) void barrier_kernel(unsigned* counters, float* payload, int nb, int G) {
    __shared__ int flag;
    for (int k = 0; k < nb; ++k) {
        if (PAYLOAD) __hip_atomic_store(payload + ((size_t)k * G + blockIdx.x) * 256 + threadIdx.x, (float)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counters + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(counters + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)G) {
                if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
                if (++spins > (1u << 22)) __builtin_trap();
            }
            if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            flag = 1;
        }
        __syncthreads();
    }
}

template <int SLEEP, bool ACQ, bool PAYLOAD>
static void run(const char* name, int G, int nb) {
    unsigned* c; float* p;
    CK(hipMalloc(&c, nb * sizeof(unsigned))); CK(hipMalloc(&p, (size_t)nb * G * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(c, 0, nb * sizeof(unsigned)));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((barrier_kernel<SLEEP, ACQ, PAYLOAD>), dim3(G), dim3(256), 0, 0, c, p, nb, G);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-44s G=%3d: %6.2f us per barrier (%d barriers, %.1f us launch)\n", name, G, best * 1e3f / nb, nb, best * 1e3f);
    CK(hipFree(c)); CK(hipFree(p));
}

int main() {
    for (int G : {64, 128, 256}) {
        run<4, true, true>("sleep 4, acquire, payload (chain_barrier)", G, 64);
        run<0, true, true>("no sleep, acquire, payload", G, 64);
        run<1, true, true>("sleep 1, acquire, payload", G, 64);
        run<1, false, true>("sleep 1, NO acquire, payload", G, 64);
        run<1, true, false>("sleep 1, acquire, no payload stores", G, 64);
    }
    run<1, true, true>("sleep 1, acquire, payload, 1 barrier", 64, 1);
    return 0;
}
