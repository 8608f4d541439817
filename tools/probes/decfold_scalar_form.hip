// The FAILING form of lat_decfold_kernel (round 2, profiles/r02_decfold_nondeterminism.md), kept as the known-bad input of the ISA scan
// of afford-motion_amd/build_hip.py (tests/test_host_logic.py::test_isa_scan_flags_the_known_bad_kernel_form compiles this file for
// gfx950 and requires the scan to reject it).  The 4 x 32 key / value entries of a (sample, head) are read straight from global memory
// with wave-uniform addresses: hipcc turns them into s_load_dwordx4 + s_mov_b32 re-packs feeding v_pk_mul_f32 / v_pk_fma_f32 through
// SGPR pairs.  Never linked into the library.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(256) void decfold_scalar_form(const float* __restrict__ wq_, const float* __restrict__ wo_,
                                                                      const float* __restrict__ kv, float* __restrict__ G,
                                                                      float* __restrict__ P, int dkv, int Hd) {
    const int b = blockIdx.x, h = blockIdx.y, c = threadIdx.x, hdd = dkv / Hd;
    const float scd = 1.0f / sqrtf((float)hdd);
    const float* kd0 = kv + ((int64_t)b * 2 + 0) * 2 * dkv + h * hdd;
    const float* kd1 = kv + ((int64_t)b * 2 + 1) * 2 * dkv + h * hdd;
    const float *vd0 = kd0 + dkv, *vd1 = kd1 + dkv;
    float4 ow[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ow[u] = (4 * u < hdd) ? *reinterpret_cast<const float4*>(wo_ + (int64_t)c * dkv + h * hdd + 4 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
    float a0 = 0.f, a1 = 0.f, p0 = 0.f, p1 = 0.f;
#pragma unroll 8
    for (int r = 0; r < hdd; ++r) {
        const float wq = wq_[(int64_t)(h * hdd + r) * dkv + c];
        a0 += wq * kd0[r];
        a1 += wq * kd1[r];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (4 * u < hdd) {
            p0 += (ow[u].x * vd0[4 * u] + ow[u].y * vd0[4 * u + 1]) + (ow[u].z * vd0[4 * u + 2] + ow[u].w * vd0[4 * u + 3]);
            p1 += (ow[u].x * vd1[4 * u] + ow[u].y * vd1[4 * u + 1]) + (ow[u].z * vd1[4 * u + 2] + ow[u].w * vd1[4 * u + 3]);
        }
    }
    const int njh = 2 * Hd;
    float* g = G + (int64_t)b * njh * dkv;
    float* p = P + (int64_t)b * njh * dkv;
    g[h * dkv + c] = a0 * scd; g[(Hd + h) * dkv + c] = a1 * scd;
    p[h * dkv + c] = p0; p[(Hd + h) * dkv + c] = p1;
}
