// afm_mha_fwd: softmax(Q K^T / sqrt(dh) + key_mask) V for the CMDM encoder (T <= ~330 tokens, dh = 64).
//
// gfx950 design
//   * grid = (sample, head, query group); a workgroup is a group of 1 / 2 / 4 / 8 / 12 waves, one wave per 32-query block.
//     T = 326 -> 11 query blocks per (sample, head): groups of 4 waves give 3 workgroups per (sample, head) = 768 workgroups at
//     B = 32 (3 co-resident per CU, each with its own barriers, so one group's softmax phase overlaps another's MFMA phase);
//     at B = 4 per GPU (strong scaling over 8 GPUs) single-wave groups give 352 workgroups instead of 32.  A query row's
//     arithmetic does not depend on the grouping (bit-identical results).
//   * K/V are streamed in 32-key blocks through a double-buffered LDS stage shared by all waves
//     (coalesced float4 global loads, one barrier per block); Q lives in registers for the whole pass.
//   * "Swapped" products on v_mfma_f32_32x32x2_f32 so that nothing is ever transposed or shuffled:
//       S^T = K Q^T   (A = K block, B = Q^T): lane l holds query (l&31), 16 keys  -> softmax is lane-local
//       O^T = V^T P^T (A = V^T,    B = P^T): the P registers ARE the B operand, V is read row-wise
//     The MFMA k index of S^T is permuted (lane-half h uses head dims 32h .. 32h+31) so each lane reads
//     its K row as 8 x ds_read_b128; K rows are padded to 68 floats (conflict-free 16-lane groups).
//   * online softmax in f32 (running max / sum per query), masked keys get -inf exactly like
//     masked_fill(-inf) in the reference; fully masked 32-key blocks are skipped (their weight is 0).
#include <atomic>
#include "common.h"
#include "profile.h"
#include <math.h>

namespace {

#ifdef AFM_TIMELINE          // tools/mha_timeline.hip only: per-workgroup phase cycle totals of wave 0; never compiled into the library
struct AfmMhaRec { unsigned long long t0, t1, c0, c1, s_c, soft_c, pv_c, sync_c; unsigned hw_id, xcc_id; };
__device__ AfmMhaRec* afm_mha_timeline = nullptr;
__device__ __forceinline__ unsigned long long afm_cyc() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define TL(...) __VA_ARGS__
#else
#define TL(...)
#endif

constexpr int DH = 64;
constexpr int KB = 32;           // keys per block
constexpr int LDKK = 68;         // padded K row (floats)
constexpr int MAX_WAVES = 12;      // 3 waves per SIMD -> 168 VGPRs each, no spills

// TRAIN: also writes lse[b,h,q] = log-sum-exp of the scaled, masked logits (saved for afm_mha_bwd) and applies
// attention-probability dropout to the P used in P V (the softmax normaliser uses the undropped P, as in torch).
// NWC: waves per workgroup when known at compile time (NST * 64 * NWC == 1024: every thread stages exactly NST float4 of each K / V
// block, no predicates), 0 = read from blockDim (NST * blockDim >= 1024, predicated).
template <int NST, bool TRAIN, int NWC>
__global__ __launch_bounds__(NWC ? 64 * NWC : (NST == 4 ? 512 : 64 * MAX_WAVES)) void mha_fwd_kernel(const float* __restrict__ qp_, int ldq, const float* __restrict__ kp_,
                                                       const float* __restrict__ vp_, int ldkv, const uint8_t* __restrict__ key_mask,
                                                       float* __restrict__ out, int Tq, int T, int H, float scale,
                                                       float* __restrict__ lse, float drop_p, uint64_t drop_seed, uint32_t drop_id, int nchunk) {
    // Tq queries (rows of qp_, stride ldq) attend over T keys / values (rows of kp_ / vp_, stride ldkv): self-attention passes the
    // packed in_proj output three times (q | k | v, ld = 3D, Tq == T), cross-attention a [B,Tq,D] query and a packed [B,T,2D] memory.
    extern __shared__ __attribute__((aligned(16))) float smem[];
    TL(const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime(); const unsigned long long tl_c0 = afm_cyc();
       unsigned long long tl_s = 0, tl_soft = 0, tl_pv = 0, tl_sync = 0;)
    float* Ks = smem;                               // [2][KB][LDKK]
    float* Vs = smem + 2 * KB * LDKK;               // [2][KB][DH]
    float* madd = Vs + 2 * KB * DH;                 // [nkb*KB] additive mask (0 / -inf)
    int* blk_valid = reinterpret_cast<int*>(madd + ((T + KB - 1) / KB) * KB);   // [nkb]

    // XCD-aware order: block i runs on XCD i % 8; give each XCD a contiguous range of (sample, head, group) so the groups that
    // share one (sample, head)'s K / V read them through the same L2
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bh = bid / nchunk, chunk = bid % nchunk;
    const int b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = NWC ? NWC : (int)(blockDim.x >> 6);
    const int nthreads = NWC ? 64 * NWC : (int)blockDim.x;
    const int r32 = lane & 31, hh = lane >> 5;
    const int D = H * DH;
    const int nkb = (T + KB - 1) / KB, nqb = (Tq + 31) / 32;
    const float* qbase = qp_ + (int64_t)b * Tq * ldq + h * DH;
    const float* kbase = kp_ + (int64_t)b * T * ldkv + h * DH;
    const float* vbase = vp_ + (int64_t)b * T * ldkv + h * DH;
    const float NEG_INF = -INFINITY;

    for (int i = tid; i < nkb; i += nthreads) blk_valid[i] = 0;
    __syncthreads();
    for (int i = tid; i < nkb * KB; i += nthreads) {
        const bool ok = (i < T) && !(key_mask && key_mask[(int64_t)b * T + i]);
        madd[i] = ok ? 0.0f : NEG_INF;
        if (ok) blk_valid[i / KB] = 1;              // benign race: every writer stores 1
    }

    // cooperative K/V block loader: 1024 float4 per block (512 K + 512 V), NST per thread (NST * threads >= 1024).  Loads are
    // UNCONDITIONAL: rows past the last key are clamped to key T-1 (real, finite data) - those keys carry an additive -inf, so their
    // probabilities are exactly 0 whatever K / V hold.  (A predicated load made hipcc put `s_waitcnt vmcnt(0)` between the loads of
    // a block, i.e. a full memory round trip in front of every block's MFMAs.)
    f32x4 stage[NST];      // ext_vector_type, not HIP's float4 struct: struct copies in a register array can pin it in scratch
    auto load_block = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = NWC ? tid + i * nthreads : min(tid + i * nthreads, 1023);
            const int isv = e >> 9, row = (e & 511) >> 4, c4 = e & 15;
            const int key = min(kb * KB + row, T - 1);
            stage[i] = *reinterpret_cast<const f32x4*>((isv ? vbase : kbase) + (int64_t)key * ldkv + c4 * 4);
        }
    };
    auto store_block = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * nthreads;
            if (NWC || e < 1024) {
                const int isv = e >> 9, row = (e & 511) >> 4, c4 = e & 15;
                float* dst = isv ? (Vs + (buf * KB + row) * DH + c4 * 4) : (Ks + (buf * KB + row) * LDKK + c4 * 4);
                *reinterpret_cast<f32x4*>(dst) = stage[i];
            }
        }
    };

    for (int q0 = chunk * nw; q0 < nqb; q0 += nw * nchunk) {
        const int qb = q0 + wave;
        const bool active = qb < nqb;
        // Q fragment: query row (clamped), head dims 32*hh .. 32*hh+31, pre-scaled
        float q[32];
        {
            const int qrow = min(qb * 32 + r32, Tq - 1);
            const float* qp = qbase + (int64_t)(active ? qrow : 0) * ldq + hh * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(qp + i * 4);
                q[4 * i + 0] = v.x * scale; q[4 * i + 1] = v.y * scale; q[4 * i + 2] = v.z * scale; q[4 * i + 3] = v.w * scale;
            }
        }
        float m_run = NEG_INF, l_run = 0.0f;
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

        __syncthreads();                 // previous pass done with LDS; madd/blk_valid visible
        load_block(0);
        store_block(0);
        __syncthreads();

        for (int kb = 0; kb < nkb; ++kb) {
            const int buf = kb & 1;
            TL(const unsigned long long tl_a = afm_cyc();)
            if (kb + 1 < nkb) load_block(kb + 1);
            if (active && blk_valid[kb]) {
                // ---- S^T = K Q^T  (32 MFMA steps over the 64 head dims)
                f32x16 s;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
                const float* kp = Ks + (buf * KB + r32) * LDKK + hh * 32;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 kv = *reinterpret_cast<const float4*>(kp + i * 4);
                    s = mfma32(kv.x, q[4 * i + 0], s);
                    s = mfma32(kv.y, q[4 * i + 1], s);
                    s = mfma32(kv.z, q[4 * i + 2], s);
                    s = mfma32(kv.w, q[4 * i + 3], s);
                }
                // ---- mask + online softmax; reg r <-> key kb*32 + (r&3) + 8*(r>>2) + 4*hh
                TL(asm volatile("" : "+v"(s)); const unsigned long long tl_b = afm_cyc(); tl_s += tl_b - tl_a;)
                float mx = NEG_INF;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 ma = *reinterpret_cast<const float4*>(madd + kb * KB + 8 * g + 4 * hh);
                    s[4 * g + 0] += ma.x; s[4 * g + 1] += ma.y; s[4 * g + 2] += ma.z; s[4 * g + 3] += ma.w;
                    mx = fmaxf(mx, fmaxf(fmaxf(s[4 * g + 0], s[4 * g + 1]), fmaxf(s[4 * g + 2], s[4 * g + 3])));
                }
                mx = fmaxf(mx, xor32(mx));
                const float m_new = fmaxf(m_run, mx);
                const float m_safe = (m_new == NEG_INF) ? 0.0f : m_new;
                const float alpha = __expf(m_run - m_safe);
                float rs = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = __expf(s[r] - m_safe);
                    rs += s[r];
                }
                rs += xor32(rs);
                l_run = l_run * alpha + rs;
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
                // ---- O^T += V^T P^T : step r multiplies key (r&3) + 8*(r>>2) + 4*hh
                if (TRAIN && drop_p > 0.0f) {
                    const DropKey dk(drop_p, drop_seed, drop_id);
                    const uint32_t row_ix = (uint32_t)bh * Tq + min(qb * 32 + r32, Tq - 1), col0 = kb * KB + 4 * hh;
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[r] *= dk(row_ix, col0 + (r & 3) + 8 * (r >> 2));
                }
                TL(asm volatile("" : "+v"(s)); const unsigned long long tl_c = afm_cyc(); tl_soft += tl_c - tl_b;)
                const float* vp = Vs + buf * KB * DH + r32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const float v0 = vp[key * DH], v1 = vp[key * DH + 32];
                    o0 = mfma32(v0, s[r], o0);
                    o1 = mfma32(v1, s[r], o1);
                }
                TL(asm volatile("" : "+v"(o0), "+v"(o1)); const unsigned long long tl_d = afm_cyc(); tl_pv += tl_d - tl_c;)
            }
            TL(const unsigned long long tl_e = afm_cyc();)
            if (kb + 1 < nkb) store_block(buf ^ 1);
            __syncthreads();
            TL(tl_sync += afm_cyc() - tl_e;)
        }

        if (active) {
            const int qrow = qb * 32 + r32;
            if (qrow < Tq) {
                if (TRAIN && hh == 0) lse[(int64_t)bh * Tq + qrow] = m_run + __logf(l_run);
                const float inv = 1.0f / l_run;
                float* op = out + ((int64_t)b * Tq + qrow) * D + h * DH + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<float4*>(op + 8 * g) =
                        make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
                    *reinterpret_cast<float4*>(op + 32 + 8 * g) =
                        make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
                }
            }
        }
    }
    TL(if (afm_mha_timeline && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        afm_mha_timeline[blockIdx.x] = AfmMhaRec{tl_t0, (unsigned long long)__builtin_amdgcn_s_memrealtime(), tl_c0, afm_cyc(), tl_s, tl_soft, tl_pv, tl_sync, hw, xcc};
    })
}

// group_waves: waves (32-query blocks) per workgroup, one of 1 / 2 / 4 / 8 / 12; 0 = choose from the launch size; < 0 = one workgroup per
// (sample, head) that walks all query blocks (long-query cross-attention, training).
int mha_fwd_launch(const float* q, int ldq, const float* k, const float* v, int ldkv, const uint8_t* key_mask, float* out, float* lse, int32_t B,
                   int32_t Tq, int32_t T, int32_t H, int32_t dh, float drop_p, uint64_t drop_seed, uint32_t drop_id, bool train, int group_waves,
                   void* stream) {
    if (dh != DH) return AFM_E_UNSUPPORTED;
    if (B == 0) return 0;                                     // empty batch (pointers may be null)
    if (!q || !k || !v || !out || B < 0 || T <= 0 || Tq <= 0 || H <= 0) return AFM_E_BADARG;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) return AFM_E_BADARG;
    if (train && (!lse || drop_p < 0.0f || drop_p >= 1.0f)) return AFM_E_BADARG;
    const int nqb = (Tq + 31) / 32, nkb = (T + 31) / 32;
    int nw, nchunk = 1;
    if (group_waves < 0) {
        nw = nqb < 4 ? 4 : (nqb > MAX_WAVES ? MAX_WAVES : nqb);
    } else {
        if (group_waves != 0 && group_waves != 1 && group_waves != 2 && group_waves != 4 && group_waves != 8 && group_waves != 12) return AFM_E_BADARG;
        nw = group_waves;
        // 4-wave groups everywhere (profiles/r02_kernel_sweep.txt, T = 326, us for 12 / 8 / 4 / 2 / 1 waves per group): B = 32: 93 / 119 / 94 / 128 / 150,
        // B = 16: 85 / 62 / 63 / 69 / 98, B = 4: 79 / 57 / 36 / 41 / 49, B = 1: 78 / 56 / 36 / 40 / 48 - the four waves share every K / V
        // tile they stage (smaller groups re-stage it per wave), and three groups fit one CU (larger ones leave SIMDs unevenly loaded)
        if (nw == 0) nw = 4;
        if (nw > nqb) nw = nqb >= 8 ? nqb : (nqb >= 4 ? 4 : (nqb >= 2 ? 2 : 1));      // never more waves than query blocks
        nchunk = (nqb + nw - 1) / nw;
    }
    const size_t lds = (size_t)(2 * KB * LDKK + 2 * KB * DH + nkb * KB) * sizeof(float) + (size_t)nkb * sizeof(int);
    if (lds > 160 * 1024) return AFM_E_UNSUPPORTED;           // T <= ~30000 keys
    if (lds > 64 * 1024) {                                     // long memories (cross-attention over N = 8192 points): opt in once
        static std::atomic<bool> attr_set{false};      // idempotent attribute: a race only repeats the call
        if (!attr_set.load(std::memory_order_acquire)) {
            const void* fns[8] = {(const void*)mha_fwd_kernel<2, false, 0>, (const void*)mha_fwd_kernel<4, false, 0>, (const void*)mha_fwd_kernel<2, true, 0>,
                                  (const void*)mha_fwd_kernel<4, true, 0>, (const void*)mha_fwd_kernel<8, false, 2>, (const void*)mha_fwd_kernel<16, false, 1>,
                                  (const void*)mha_fwd_kernel<2, false, 8>, (const void*)mha_fwd_kernel<4, false, 4>};
            for (int i = 0; i < 8; ++i) {
                hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return (int)e;
            }
            attr_set.store(true, std::memory_order_release);
        }
    }
    const float scale = 1.0f / sqrtf((float)dh);
    hipStream_t s = (hipStream_t)stream;
    AfmProf prof(AFM_PROF_MHA, 4.0 * B * H * (double)Tq * T * dh, s);
#define AFM_MHA(NST, TR, NWC) hipLaunchKernelGGL((mha_fwd_kernel<NST, TR, NWC>), dim3(B * H * nchunk), dim3(nw * 64), lds, s, q, ldq, k, v, ldkv, key_mask, out, Tq, T, H, scale, lse, drop_p, drop_seed, drop_id, nchunk)
    if (train) { if (nw >= 8) AFM_MHA(2, true, 0); else AFM_MHA(4, true, 0); }
    else if (nw == 8) AFM_MHA(2, false, 8);
    else if (nw > 8) AFM_MHA(2, false, 0);
    else if (nw == 4) AFM_MHA(4, false, 4);
    else if (nw > 4) AFM_MHA(4, false, 0);
    else if (nw == 2) AFM_MHA(8, false, 2);
    else AFM_MHA(16, false, 1);
#undef AFM_MHA
    AFM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int afm_mha_fwd_grouped(const float* qkv, const uint8_t* key_mask, float* out, int32_t B, int32_t T, int32_t H,
                                   int32_t dh, int32_t group_waves, void* stream) {
    if (group_waves < 0) return AFM_E_BADARG;
    const int D = H * dh;
    return mha_fwd_launch(qkv, 3 * D, qkv ? qkv + D : nullptr, qkv ? qkv + 2 * D : nullptr, 3 * D, key_mask, out, nullptr, B, T, T, H, dh, 0.0f, 0, 0, false,
                          group_waves, stream);
}

extern "C" int afm_mha_fwd(const float* qkv, const uint8_t* key_mask, float* out, int32_t B, int32_t T, int32_t H,
                           int32_t dh, void* stream) {
    return afm_mha_fwd_grouped(qkv, key_mask, out, B, T, H, dh, 0, stream);
}

extern "C" int afm_mha_fwd_train(const float* qkv, const uint8_t* key_mask, float* out, float* lse, int32_t B, int32_t T, int32_t H,
                                 int32_t dh, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* stream) {
    const int D = H * dh;
    return mha_fwd_launch(qkv, 3 * D, qkv ? qkv + D : nullptr, qkv ? qkv + 2 * D : nullptr, 3 * D, key_mask, out, lse, B, T, T, H, dh, drop_p, drop_seed,
                          drop_id, true, -1, stream);
}

extern "C" int afm_mha_cross_fwd(const float* q, const float* kv, const uint8_t* key_mask, float* out, int32_t B, int32_t Tq, int32_t Tk, int32_t H,
                                 int32_t dh, void* stream) {
    const int D = H * dh;
    return mha_fwd_launch(q, D, kv, kv ? kv + D : nullptr, 2 * D, key_mask, out, nullptr, B, Tq, Tk, H, dh, 0.0f, 0, 0, false, -1, stream);
}
