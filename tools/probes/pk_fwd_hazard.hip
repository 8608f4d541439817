// Instruction-level replay for the nondeterminism logged in profiles/r02_decfold_nondeterminism.md.  NEGATIVE RESULT: it does not
// reproduce it.  In the failing build of lat_decfold_kernel one product of the P sum (always one with an odd-indexed SGPR operand of a
// v_pk_mul_f32 / v_pk_fma_f32) evaluated to 0 in lanes 48..63 of single waves, only while another stream had kernels in flight.  This
// replays the instruction block of that term verbatim (registers as hipcc allocated them), alone and next to VALU / MFMA+LDS / HBM /
// scalar-load contenders on a second stream: 0 wrong results in 8e7 wave-iterations per mode (MI355X, gpurun 2026-09-27).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/pk_fwd_hazard.hip -o tools/probes/pk_fwd_hazard && tools/probes/pk_fwd_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(const float* __restrict__ pa, const float* __restrict__ pb, unsigned* bad_by_lane, int iters) {
    const int lane = threadIdx.x & 63;
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
        const float a = 1.0f + lane * 0.25f + i * 1e-3f, b = 2.0f - lane * 0.125f, c = 0.5f + lane * 0.0625f, d = 3.0f + lane * 0.5f - i * 1e-3f;
        float r14, r15, r16, r17;
        asm volatile(
            "v_mov_b32 v18, %4\n v_mov_b32 v19, %5\n v_mov_b32 v20, %6\n v_mov_b32 v21, %7\n"
            "s_load_dwordx4 s[28:31], %8, 0x0\n"
            "s_load_dwordx4 s[44:47], %9, 0x0\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_mov_b32 s26, s28\n"
            "s_mov_b32 s27, s45\n"
            "s_mov_b32 s45, s29\n"
            "s_mov_b32 s28, s30\n"
            "s_mov_b32 s29, s47\n"
            "v_pk_mul_f32 v[14:15], v[18:19], s[26:27]\n"
            "v_pk_mul_f32 v[16:17], v[20:21], s[28:29]\n"
            "s_mov_b32 s47, s31\n"
            "v_pk_fma_f32 v[14:15], v[18:19], s[44:45], v[14:15] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n"
            "v_pk_fma_f32 v[16:17], v[20:21], s[46:47], v[16:17] op_sel:[0,0,1] op_sel_hi:[1,1,0]\n"
            "s_nop 0\n"
            "v_mov_b32 %0, v14\n v_mov_b32 %1, v15\n v_mov_b32 %2, v16\n v_mov_b32 %3, v17\n"
            : "=v"(r14), "=v"(r15), "=v"(r16), "=v"(r17)
            : "v"(a), "v"(b), "v"(c), "v"(d), "s"(pa), "s"(pb)
            : "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "s26", "s27", "s28", "s29", "s30", "s31", "s44", "s45", "s46", "s47", "memory");
        const float A12 = pa[0], A13 = pa[1], A14 = pa[2], A15 = pa[3], B12 = pb[0], B13 = pb[1], B14 = pb[2], B15 = pb[3];
        const float e14 = __fmaf_rn(a, B12, __fmul_rn(b, B13)), e15 = __fmaf_rn(b, A13, __fmul_rn(a, A12));
        const float e16 = __fmaf_rn(c, B14, __fmul_rn(d, B15)), e17 = __fmaf_rn(d, A15, __fmul_rn(c, A14));
        bad += (r14 != e14 || r15 != e15 || r16 != e16 || r17 != e17) ? 1u : 0u;
    }
    if (bad) atomicAdd(&bad_by_lane[lane], bad);
}

__global__ __launch_bounds__(256) void contend_valu(float* sink, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    if (a == 123.456f) sink[0] = a + b;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void contend_mfma(float* sink, int iters) {
    __shared__ float buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc = {0};
    for (int i = 0; i < iters; ++i) {
        const float a = buf[(threadIdx.x * 4 + i) & 4095], b = buf[(threadIdx.x + 17 * i) & 4095];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (acc[0] == 123.456f) sink[0] = acc[1];
}

__global__ __launch_bounds__(256) void contend_mem(const float4* __restrict__ src, float* sink, int64_t n, int iters) {
    float4 s = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const float4 v = src[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (s.x == 123.456f) sink[0] = s.x + s.y + s.z + s.w;
}

__global__ __launch_bounds__(256) void contend_smem(const float* __restrict__ tab, float* sink, int iters) {      // scalar-load traffic into the same SGPR files
    float acc = 0.f;
    const float* p = tab + (blockIdx.x & 15) * 64;
    for (int i = 0; i < iters; ++i) {
        float r;
        asm volatile("s_load_dwordx16 s[48:63], %1, 0x0\n s_load_dwordx16 s[64:79], %1, 0x40\n s_waitcnt lgkmcnt(0)\n v_mov_b32 %0, s63\n"
                     : "=v"(r) : "s"(p)
                     : "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
                       "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "memory");
        acc += r;
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    hipStream_t s0, s1;
    (void)hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    unsigned* d_bad; float *d_sink, *d_a, *d_b; float4* d_big;
    const int64_t nbig = 64ll << 20;
    (void)hipMalloc(&d_bad, 64 * sizeof(unsigned)); (void)hipMalloc(&d_sink, 4); (void)hipMalloc(&d_a, 64); (void)hipMalloc(&d_b, 64);
    (void)hipMalloc(&d_big, nbig * sizeof(float4)); (void)hipMemset(d_big, 0, nbig * sizeof(float4));
    const float ha[4] = {0.75f, -1.5f, 1.25f, 1.470225f}, hb[4] = {-0.5f, 0.739016f, 2.25f, 1.508179f};
    (void)hipMemcpy(d_a, ha, 16, hipMemcpyHostToDevice); (void)hipMemcpy(d_b, hb, 16, hipMemcpyHostToDevice);
    const char* names[5] = {"probe alone", "VALU contender on stream 2", "MFMA+LDS contender on stream 2", "HBM-streaming contender on stream 2",
                            "scalar-load contender on stream 2"};
    for (int mode = 0; mode < 5; ++mode) {
        for (int wgs : {128, 1024}) {
            (void)hipMemsetAsync(d_bad, 0, 64 * sizeof(unsigned), s0);
            (void)hipStreamSynchronize(s0);
            for (int rep = 0; rep < 20; ++rep) {
                if (mode == 1) hipLaunchKernelGGL(contend_valu, dim3(1024), dim3(256), 0, s1, d_sink, 20000);
                if (mode == 2) hipLaunchKernelGGL(contend_mfma, dim3(1024), dim3(256), 0, s1, d_sink, 4000);
                if (mode == 3) hipLaunchKernelGGL(contend_mem, dim3(1024), dim3(256), 0, s1, d_big, d_sink, nbig, 1);
                if (mode == 4) hipLaunchKernelGGL(contend_smem, dim3(1024), dim3(256), 0, s1, (const float*)d_big, d_sink, 20000);
                hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, s0, d_a, d_b, d_bad, 4000);
            }
            (void)hipDeviceSynchronize();
            std::vector<unsigned> h(64);
            (void)hipMemcpy(h.data(), d_bad, 64 * sizeof(unsigned), hipMemcpyDeviceToHost);
            unsigned long long q[4] = {0, 0, 0, 0};
            for (int l = 0; l < 64; ++l) q[l >> 4] += h[l];
            printf("%-38s %4d probe workgroups: wrong results by lane quarter [0-15 | 16-31 | 32-47 | 48-63]: %llu %llu %llu %llu\n", names[mode], wgs, q[0], q[1], q[2], q[3]);
        }
    }
    return 0;
}
