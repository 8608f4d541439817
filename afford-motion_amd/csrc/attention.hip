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
#include "bf16split.h"
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
constexpr int AFM_MHA_DEFAULT_GROUP = 4;      // waves per workgroup of the inference kernel when the caller does not choose

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


// ------------------------------------------------------------------------------------------------------------------------------
// Inference attention on the bf16 matrix pipe with f32 results (round 3).  gfx950's f32 MFMA runs at 1/16 of its bf16 MFMA, so -
// exactly as in gemm_split.hip - every f32 operand is split into three bf16 terms (8 + 8 + 8 significant bits, exact) and the nine
// exact cross products accumulate in f32 inside v_mfma_f32_32x32x16_bf16: 9 x 32 = 288 matrix-pipe cycles per (32 x 32 x 16) block
// product instead of 8 x 64 = 512 on v_mfma_f32_32x32x2_f32, the same f32 products in another summation order.
//   Q   split once per wave (registers: three planes x four K16 steps);
//   K/V split once per WORKGROUP while a 32-key block is staged (the waves of a group share the planes through LDS):
//       K planes  [32 keys][64 dims] bf16, rows padded to 144 B (conflict-free ds_read_b128 lane groups, 64-bank rule);
//       V planes  TRANSPOSED [64 dims][32 key slots] bf16, rows padded to 80 B - the bf16 MFMA wants eight consecutive k per lane, and
//                 for O^T = V^T P^T the k index is the key.  A staging thread owns 4 keys x 4 dims: its four bf16 of one dim are
//                 one 8-byte store (16 lanes = 8 key quads x 2 dim-quad parities: 16 distinct 8-byte slots of the 32 write banks);
//   P   split in registers right after the softmax: the S^T accumulator layout (lane = query, 16 keys) IS the B operand of the second
//       product once the key <-> k-slot map is chosen accordingly: slot 16 t + 8 hh + e <-> key 16 t + 8 (e >> 2) + 4 hh + (e & 3),
//       i.e. key bits 2 and 3 swapped - the V planes are written in that slot order.
// Everything else (swapped products, online softmax, exact -inf masking, skipped fully-masked blocks, grid = (sample, head, query
// group), double-buffered stage with one barrier per block) is the f32 kernel's.  A query row's arithmetic does not depend on the
// grouping (bit-identical results for every NW).
constexpr int KROWB = 144, VROWB = 80;                 // plane row bytes
constexpr int KPLANE = KB * KROWB, VPLANE = DH * VROWB;
constexpr int KVSTAGE = 3 * KPLANE + 3 * VPLANE;       // 29184 bytes per stage

// ---- Two key segments (round 4).  The softmax of a query row is computed over TWO segments of the key blocks - segment 0 = blocks
// [0, n0), segment 1 = [n0, nkb), n0 = ceil(nkb / 2) - each with its own online-softmax state (m, l, O) started from (-inf, 0, 0), and the two
// states are merged once at the end:  m = max(m0, m1), a_s = exp2(m_s - m), l = l0 a0 + l1 a1, O = O0 a0 + O1 a1, out = O / l  (seg_merge
// below pins the operation order).  What this buys: a small launch (B H < 128: strong scaling runs 4 samples per GPU) puts the two segments
// of a query block on TWO waves (SEG2: a workgroup of 2 NW waves, waves NW .. 2 NW - 1 take segment 1 with K / V stages of their own) - the
// serial chain of a wave drops from 11 to 6 key blocks at T = 326 (26.5 -> ~17 us per launch at B = 4), while a large launch walks the
// segments one after the other in one wave (O0 parked in LDS in between).  Both forms execute the same float operations in the same
// order for every query row: a batch and its shards stay bit-identical (tests/test_gpu_ops.py, test_gpu_cmdm.py).  One segment when nkb == 1.
__device__ __forceinline__ float seg_merge(float x0, float a0, float x1, float a1) { return __fmaf_rn(x0, a0, __fmul_rn(x1, a1)); }
// a wave's parked segment state in LDS: O as [32 registers][64 lanes] (conflict-free), then (m, l) per query [32][2] (written by the lanes of
// the first half-wave, read by both): 8448 bytes - twelve of them fit next to the K / V stages of the 12-wave form (160 KB)
constexpr int PARK_FLOATS = 32 * 64 + 64;

// second launch bound = waves per SIMD the grid needs: the sequential forms as in round 3 (12 waves: one workgroup of three waves per SIMD =
// 168 VGPRs; 4 / 8 waves: two; 6: three), SEG2: one workgroup per CU (its LDS) of 2 NW waves
// NPROD (round 6): 9 = all cross products of the three-term operands (exact f32 products), 6 = without the three smallest (a2 b3, a3 b2, a3 b3:
// each <= 2^-24 |a||b|) - the arithmetic of afm_linear's AFM_ARITH_BF16X6, chosen by the same argument (afm_mha_fwd_arith).
template <int NW, bool SEG2, int NPROD>
__global__ __launch_bounds__(64 * NW * (SEG2 ? 2 : 1), SEG2 ? (NW >= 2 ? NW / 2 : 1) : (NW == 4 || NW == 8 ? 2 : (NW == 6 || NW == 12 ? 3 : 1)))
void mha_fwd_split_kernel(const float* __restrict__ qp_, int ldq, const float* __restrict__ kp_, const float* __restrict__ vp_, int ldkv,
                          const uint8_t* __restrict__ key_mask, float* __restrict__ out, int Tq, int T, int H, float scale, int nchunk, int q_first) {
    // `scale` = log2(e) / sqrt(dh): the logits are kept in base-2 units, so the softmax is exp2(s - m) = one v_exp_f32 per element
    // q_first: only the query rows q_first .. Tq - 1 are computed (query blocks start at q_first; the rows in front keep whatever `out` held)
    static_assert(!SEG2 || NW * PARK_FLOATS * 4 <= 2 * KVSTAGE, "SEG2: segment 1's parked state fits its own K / V stages");
    constexpr int NSEGW = SEG2 ? 2 : 1;                 // key segments in flight per workgroup
    constexpr int NTH = 64 * NW;                        // threads that stage one segment's blocks
    constexpr int NT = NTH * NSEGW;
    constexpr int NIT = (256 + NTH - 1) / NTH;          // staging items per thread (256 items per block: 128 of K, 128 of V)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TL(const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime(); const unsigned long long tl_c0 = afm_cyc();
       unsigned long long tl_s = 0, tl_soft = 0, tl_pv = 0, tl_sync = 0;)
    const int tid = threadIdx.x, lane = tid & 63, wave_all = tid >> 6;
    const int seg = SEG2 ? __builtin_amdgcn_readfirstlane(wave_all / NW) : 0;      // SEG2: which segment this wave works on
    const int wave = SEG2 ? wave_all - seg * NW : wave_all, htid = tid - seg * NTH;
    unsigned char* kv = smem_raw + seg * (2 * KVSTAGE);  // [2][KVSTAGE] per segment in flight
    float* madd = reinterpret_cast<float*>(smem_raw + NSEGW * 2 * KVSTAGE);         // [nkb * KB] additive mask (0 / -inf)
    int* blk_valid = reinterpret_cast<int*>(madd + ((T + KB - 1) / KB) * KB);       // [nkb]
    // parked state: SEG2 - segment 1's final (O, m, l) on its way to the merging wave, in segment 1's (then idle) K / V stages;
    //               sequential - O0 of every wave while it walks segment 1, behind the mask
    float* park = SEG2 ? reinterpret_cast<float*>(smem_raw + 2 * KVSTAGE) : reinterpret_cast<float*>(blk_valid + ((T + KB - 1) / KB + 3) / 4 * 4);

    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bh = bid / nchunk, chunk = bid % nchunk;
    const int b = bh / H, h = bh % H;
    const int r32 = lane & 31, hh = lane >> 5;
    const int D = H * DH;
    const int nkb = (T + KB - 1) / KB, nqb = (Tq - q_first + 31) / 32;
    const int n0 = (nkb + 1) / 2, n1 = nkb - n0;        // blocks of segment 0 / 1 (n0 >= n1; n1 == 0: one segment, no merge)
    const float* qbase = qp_ + (int64_t)b * Tq * ldq + h * DH;
    const float* kbase = kp_ + (int64_t)b * T * ldkv + h * DH;
    const float* vbase = vp_ + (int64_t)b * T * ldkv + h * DH;
    const float NEG_INF = -INFINITY;

    // blk_valid[kb]: bit 0 = the block has a valid key, bit 1 = it has a masked key, bit 2 = it has a valid key among its keys 16 .. 31
    // (blocks without bit 0 are skipped, blocks without bit 1 skip the mask addition, blocks without bit 2 - the ragged tail of T = 326:
    // 6 keys - skip the second K16 step of P V, whose probabilities are exact zeros); written with atomic ORs by whichever threads see such a key
    auto setup_mask = [&]() {
        for (int i = tid; i < nkb; i += NT) blk_valid[i] = 0;
        __syncthreads();
        for (int i = tid; i < nkb * KB; i += NT) {
            const bool ok = (i < T) && !(key_mask && key_mask[(int64_t)b * T + i]);
            madd[i] = ok ? 0.0f : NEG_INF;
            atomicOr(&blk_valid[i / KB], ok ? (1 | ((i & 16) ? 4 : 0)) : 2);
        }
    };
    // small groups request their first K / V block BEFORE the mask is built (the mask bytes are a memory round trip of their own - the
    // key mask was written by the step's prologue on another XCD - and a small launch has nothing else in flight to hide it behind:
    // ~1.5 us at the head of every launch, profiles/r04_mha_timeline_small.txt); the 168-register forms cannot afford the live range
    if (NW > 4) setup_mask();

    // Staging items (all loads unconditional, rows past the last key clamped to key T-1: those keys carry an additive -inf):
    //   item i < 128  (K): float4 c4 = i & 15 of keys (i >> 4) + 8 q, q = 0..3  -> per load instruction 4 key rows x 256 contiguous bytes
    //   item 128 + i  (V): keys 4 kq .. 4 kq + 3 (kq = i & 7) x dims 4 c .. 4 c + 3 (c = ((i >> 3) & 1) + 2 (i >> 4))
    f32x4 st[NIT][4];
    auto load_block = [&](int kb) {
#pragma unroll
        for (int n = 0; n < NIT; ++n) {
            const int item = htid + n * NTH;
            if (NTH * NIT > 256 && item >= 256) continue;
            if (item < 128) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int key = min(kb * KB + (item >> 4) + 8 * q, T - 1);
                    st[n][q] = *reinterpret_cast<const f32x4*>(kbase + (int64_t)key * ldkv + (item & 15) * 4);
                }
            } else {
                const int i = item - 128, kq = i & 7, c = ((i >> 3) & 1) + 2 * (i >> 4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int key = min(kb * KB + 4 * kq + q, T - 1);
                    st[n][q] = *reinterpret_cast<const f32x4*>(vbase + (int64_t)key * ldkv + c * 4);
                }
            }
        }
    };
    auto store_block = [&](int buf) {
        unsigned char* base = kv + buf * KVSTAGE;
#pragma unroll
        for (int n = 0; n < NIT; ++n) {
            const int item = htid + n * NTH;
            if (NTH * NIT > 256 && item >= 256) continue;
            if (item < 128) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t a1, a2, a3, b1, b2, b3;
                    split2(st[n][q][0], st[n][q][1], a1, a2, a3);
                    split2(st[n][q][2], st[n][q][3], b1, b2, b3);
                    unsigned char* d = base + ((item >> 4) + 8 * q) * KROWB + (item & 15) * 8;
                    *reinterpret_cast<u32x2*>(d) = u32x2{a1, b1};
                    *reinterpret_cast<u32x2*>(d + KPLANE) = u32x2{a2, b2};
                    *reinterpret_cast<u32x2*>(d + 2 * KPLANE) = u32x2{a3, b3};
                }
            } else {
                const int i = item - 128, kq = i & 7, c = ((i >> 3) & 1) + 2 * (i >> 4);
                const int sq = (kq & 4) | ((kq & 1) << 1) | ((kq >> 1) & 1);        // key bits 2 <-> 3 swapped: slot quad of key quad kq
                unsigned char* d0 = base + 3 * KPLANE + (4 * c) * VROWB + sq * 8;
#pragma unroll
                for (int j = 0; j < 4; ++j) {                                       // dim 4 c + j: its four keys are one 8-byte store per plane
                    uint32_t a1, a2, a3, b1, b2, b3;
                    split2(st[n][0][j], st[n][1][j], a1, a2, a3);
                    split2(st[n][2][j], st[n][3][j], b1, b2, b3);
                    unsigned char* d = d0 + j * VROWB;
                    *reinterpret_cast<u32x2*>(d) = u32x2{a1, b1};
                    *reinterpret_cast<u32x2*>(d + VPLANE) = u32x2{a2, b2};
                    *reinterpret_cast<u32x2*>(d + 2 * VPLANE) = u32x2{a3, b3};
                }
            }
        }
    };

    // the blocks this wave walks: SEG2 - its segment's (padded to n0 trips so that both halves meet at every barrier); sequential - all
    const int kb_first = SEG2 ? seg * n0 : 0, ntrip = SEG2 ? n0 : nkb, kb_end = SEG2 ? (seg ? nkb : n0) : nkb;

    bool first_pass = true;
    for (int q0 = chunk * NW; q0 < nqb; q0 += NW * nchunk) {
        const int qb = q0 + wave;
        const bool active = qb < nqb;
        if (NW <= 4) load_block(kb_first);      // in flight under the mask setup, the Q loads and the Q split (the 168-register variants cannot afford the live range)
        if (NW <= 4 && first_pass) { setup_mask(); first_pass = false; }
        // Q planes: query row (clamped), K16 step s covers head dims 16 s + 8 hh .. + 7, pre-scaled (1 / sqrt(64) is a power of two)
        u32x4 qpl[4][3];
        {
            const int qrow = min(q_first + qb * 32 + r32, Tq - 1);
            const float* qp = qbase + (int64_t)(active ? qrow : 0) * ldq + hh * 8;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(qp + 16 * s), v1 = *reinterpret_cast<const f32x4*>(qp + 16 * s + 4);
                uint32_t w1[4], w2[4], w3[4];
                split2(v0[0] * scale, v0[1] * scale, w1[0], w2[0], w3[0]);
                split2(v0[2] * scale, v0[3] * scale, w1[1], w2[1], w3[1]);
                split2(v1[0] * scale, v1[1] * scale, w1[2], w2[2], w3[2]);
                split2(v1[2] * scale, v1[3] * scale, w1[3], w2[3], w3[3]);
                qpl[s][0] = u32x4{w1[0], w1[1], w1[2], w1[3]};
                qpl[s][1] = u32x4{w2[0], w2[1], w2[2], w2[3]};
                qpl[s][2] = u32x4{w3[0], w3[1], w3[2], w3[3]};
            }
        }
        float m_run = NEG_INF, l_run = 0.0f;
        f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }

        __syncthreads();                 // previous pass done with LDS (stages and parked state); madd / blk_valid visible
        if (NW > 4) load_block(kb_first);
        store_block(0);
        __syncthreads();

        for (int it = 0; it < ntrip; ++it) {
            const int kb = kb_first + it, buf = it & 1;
            const bool have = kb < kb_end;                       // SEG2: segment 1 may have one block less (an idle trip, barriers only)
            TL(const unsigned long long tl_a = afm_cyc();)
            if (kb + 1 < kb_end) load_block(kb + 1);
            if (!SEG2 && n1 > 0 && kb == n0 && active) {
                // sequential form, first block of segment 1: park (O0, m0, l0) (LDS), start over
                float* pk = park + wave * PARK_FLOATS + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) { pk[r * 64] = o0[r]; pk[(16 + r) * 64] = o1[r]; o0[r] = 0.f; o1[r] = 0.f; }
                if (hh == 0) *reinterpret_cast<float2*>(park + wave * PARK_FLOATS + 32 * 64 + 2 * r32) = make_float2(m_run, l_run);
                m_run = NEG_INF; l_run = 0.0f;
            }
            const int bflag = have ? blk_valid[kb] : 0;
            if (active && (bflag & 1)) {
                const unsigned char* kpl = kv + buf * KVSTAGE + r32 * KROWB + hh * 16;
                const unsigned char* vpl = kv + buf * KVSTAGE + 3 * KPLANE + r32 * VROWB + hh * 16;
                // ---- S^T = K Q^T: A = K block (row = key), B = Q^T (column = query), 4 K16 steps x 9 products
                f32x16 s;
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
                for (int st_ = 0; st_ < 4; ++st_) {
                    u32x4 ak[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) ak[pl] = *reinterpret_cast<const u32x4*>(kpl + pl * KPLANE + st_ * 32);
#pragma unroll
                    for (int q = 9 - NPROD; q < 9; ++q) s = mfma_bf16(ak[AFM_PA[q]], qpl[st_][AFM_PB[q]], s);
                }
                // ---- mask + online softmax; reg r <-> key kb*32 + (r&3) + 8*(r>>2) + 4*hh
                TL(asm volatile("" : "+v"(s)); const unsigned long long tl_b = afm_cyc(); tl_s += tl_b - tl_a;)
                float mx = NEG_INF;
                if (bflag & 2) {                 // wave-uniform: only blocks with a masked key pay the addition
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 ma = *reinterpret_cast<const float4*>(madd + kb * KB + 8 * g + 4 * hh);
                        s[4 * g + 0] += ma.x; s[4 * g + 1] += ma.y; s[4 * g + 2] += ma.z; s[4 * g + 3] += ma.w;
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) mx = fmaxf(mx, fmaxf(fmaxf(s[4 * g + 0], s[4 * g + 1]), fmaxf(s[4 * g + 2], s[4 * g + 3])));
                mx = fmaxf(mx, xor32(mx));
                const float m_new = fmaxf(m_run, mx);
                const float m_safe = (m_new == NEG_INF) ? 0.0f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
                float rs = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[r] = __builtin_amdgcn_exp2f(s[r] - m_safe);
                    rs += s[r];
                }
                rs += xor32(rs);
                l_run = l_run * alpha + rs;
                m_run = m_new;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
                TL(asm volatile("" : "+v"(s)); const unsigned long long tl_c = afm_cyc(); tl_soft += tl_c - tl_b;)
                // ---- O^T += V^T P^T: K16 step t takes the P registers 8 t .. 8 t + 7 of this lane (k-slot 8 hh + e <-> register 8 t + e)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (t == 1 && !(bflag & 4)) continue;          // wave-uniform: no valid key in slots 16 .. 31 - every P there is +0, the step adds nothing
                    uint32_t w1[4], w2[4], w3[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) split2(s[8 * t + 2 * i], s[8 * t + 2 * i + 1], w1[i], w2[i], w3[i]);
                    const u32x4 pp[3] = {u32x4{w1[0], w1[1], w1[2], w1[3]}, u32x4{w2[0], w2[1], w2[2], w2[3]}, u32x4{w3[0], w3[1], w3[2], w3[3]}};
                    u32x4 av0[3], av1[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        av0[pl] = *reinterpret_cast<const u32x4*>(vpl + pl * VPLANE + t * 32);
                        av1[pl] = *reinterpret_cast<const u32x4*>(vpl + pl * VPLANE + 32 * VROWB + t * 32);
                    }
#pragma unroll
                    for (int q = 9 - NPROD; q < 9; ++q) {
                        o0 = mfma_bf16(av0[AFM_PA[q]], pp[AFM_PB[q]], o0);
                        o1 = mfma_bf16(av1[AFM_PA[q]], pp[AFM_PB[q]], o1);
                    }
                }
                TL(asm volatile("" : "+v"(o0), "+v"(o1)); const unsigned long long tl_d = afm_cyc(); tl_pv += tl_d - tl_c;)
            }
            TL(const unsigned long long tl_e = afm_cyc();)
            if (kb + 1 < kb_end) store_block(buf ^ 1);
            __syncthreads();
            TL(tl_sync += afm_cyc() - tl_e;)
        }

        // ---- merge the two segments' states (n1 > 0), normalise, store
        if (SEG2 && n1 > 0) {
            if (seg == 1 && active) {              // segment 1's state -> LDS (its own, now idle, stages)
                float* pk = park + wave * PARK_FLOATS + lane;
#pragma unroll
                for (int r = 0; r < 16; ++r) { pk[r * 64] = o0[r]; pk[(16 + r) * 64] = o1[r]; }
                if (hh == 0) *reinterpret_cast<float2*>(park + wave * PARK_FLOATS + 32 * 64 + 2 * r32) = make_float2(m_run, l_run);
            }
            __syncthreads();
        }
        if (active && seg == 0) {
            const int qrow = q_first + qb * 32 + r32;
            float l_fin = l_run;
            if (n1 > 0) {
                // the parked state is read operand by operand (no second accumulator set in registers: the 168-register forms have none to spare)
                const float* pk = park + wave * PARK_FLOATS + lane;
                const float2 mlp = *reinterpret_cast<const float2*>(park + wave * PARK_FLOATS + 32 * 64 + 2 * r32);
                const float mp = mlp.x, lp = mlp.y;
                const float m0 = SEG2 ? m_run : mp, l0 = SEG2 ? l_run : lp, m1 = SEG2 ? mp : m_run, l1 = SEG2 ? lp : l_run;
                const float m = fmaxf(m0, m1), ms = (m == NEG_INF) ? 0.0f : m;
                const float a0 = __builtin_amdgcn_exp2f(m0 - ms), a1 = __builtin_amdgcn_exp2f(m1 - ms);
                l_fin = seg_merge(l0, a0, l1, a1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (SEG2) {        // registers hold segment 0, LDS segment 1
                        o0[r] = seg_merge(o0[r], a0, pk[r * 64], a1);
                        o1[r] = seg_merge(o1[r], a0, pk[(16 + r) * 64], a1);
                    } else {           // LDS holds segment 0, registers segment 1
                        o0[r] = seg_merge(pk[r * 64], a0, o0[r], a1);
                        o1[r] = seg_merge(pk[(16 + r) * 64], a0, o1[r], a1);
                    }
                }
            }
            if (qrow < Tq) {
                const float inv = 1.0f / l_fin;
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

// group_waves: waves (32-query blocks) per workgroup, one of 1 / 2 / 4 / 6 / 8 / 12 (one wave walks both key segments of its query block),
// or 100 + {2, 4}: the two key segments of a query block on two waves (workgroups of 2 x that many waves); 0 = the library's choice;
// < 0 = one workgroup per (sample, head) that walks all query blocks (long-query cross-attention, training).  Inference runs the
// bf16-split kernel, training (lse output, attention dropout) the f32-MFMA kernel.
template <int NW, bool SEG2, int NPROD>
int launch_split_mha(const float* q, int ldq, const float* k, const float* v, int ldkv, const uint8_t* key_mask, float* out, int B, int Tq, int T, int H,
                     float scale, int nchunk, size_t lds, hipStream_t s, int q_first) {
    static const int attr = (int)hipFuncSetAttribute((const void*)mha_fwd_split_kernel<NW, SEG2, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != 0) return attr;
    hipLaunchKernelGGL((mha_fwd_split_kernel<NW, SEG2, NPROD>), dim3(B * H * nchunk), dim3(NW * 64 * (SEG2 ? 2 : 1)), lds, s, q, ldq, k, v, ldkv, key_mask, out, Tq, T, H, scale,
                       nchunk, q_first);
    AFM_CHECK_LAUNCH();
    return 0;
}

// dynamic LDS of the split kernel: stages + additive mask + block flags (+ the sequential form's parked segment-0 state: 8448 B per wave)
inline size_t split_mha_lds(int nkb, int nw, bool seg2) {
    const size_t mask = (size_t)nkb * KB * sizeof(float) + (size_t)((nkb + 3) / 4 * 4) * sizeof(int);
    return seg2 ? (size_t)4 * KVSTAGE + mask : (size_t)2 * KVSTAGE + mask + (nkb > 1 ? (size_t)nw * PARK_FLOATS * sizeof(float) : 0);
}

int mha_fwd_launch(const float* q, int ldq, const float* k, const float* v, int ldkv, const uint8_t* key_mask, float* out, float* lse, int32_t B,
                   int32_t Tq, int32_t T, int32_t H, int32_t dh, float drop_p, uint64_t drop_seed, uint32_t drop_id, bool train, int group_waves,
                   void* stream, int q_first = 0, int arith = AFM_ARITH_DEFAULT) {
    if (dh != DH) return AFM_E_UNSUPPORTED;
    if (arith != AFM_ARITH_DEFAULT && arith != AFM_ARITH_F32 && arith != AFM_ARITH_BF16X6 && arith != AFM_ARITH_BF16X9 && arith != AFM_ARITH_BF16X1) return AFM_E_BADARG;
    // six products for AFM_ARITH_DEFAULT / AFM_ARITH_BF16X6, all nine (exact f32 products) for every other setting - a function of `arith` only
    const bool six = arith == AFM_ARITH_DEFAULT || arith == AFM_ARITH_BF16X6;
    if (B == 0) return 0;                                     // empty batch (pointers may be null)
    if (!q || !k || !v || !out || B < 0 || T <= 0 || Tq <= 0 || H <= 0) return AFM_E_BADARG;
    if ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)out)) & 15) return AFM_E_BADARG;
    if ((ldq & 3) || (ldkv & 3)) return AFM_E_BADARG;
    if (train && (!lse || drop_p < 0.0f || drop_p >= 1.0f)) return AFM_E_BADARG;
    if (q_first < 0 || q_first >= Tq || (train && q_first)) return AFM_E_BADARG;
    const int nqb = (Tq - q_first + 31) / 32, nkb = (T + 31) / 32;
    const float scale = 1.0f / sqrtf((float)dh);
    const float scale2 = 1.4426950408889634f * scale;         // log2(e) / sqrt(dh): the inference kernel's logits are in base-2 units
    hipStream_t s = (hipStream_t)stream;
    if (train) {
        const int nw = nqb < 4 ? 4 : (nqb > MAX_WAVES ? MAX_WAVES : nqb);
        const size_t lds = (size_t)(2 * KB * LDKK + 2 * KB * DH + nkb * KB) * sizeof(float) + (size_t)nkb * sizeof(int);
        if (lds > 160 * 1024) return AFM_E_UNSUPPORTED;
        if (lds > 64 * 1024) {
            static std::atomic<bool> attr_set{false};      // idempotent attribute: a race only repeats the call
            if (!attr_set.load(std::memory_order_acquire)) {
                const void* fns[2] = {(const void*)mha_fwd_kernel<2, true, 0>, (const void*)mha_fwd_kernel<4, true, 0>};
                for (int i = 0; i < 2; ++i) {
                    hipError_t e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e != hipSuccess) return (int)e;
                }
                attr_set.store(true, std::memory_order_release);
            }
        }
        AfmProf prof(AFM_PROF_MHA, 4.0 * B * H * (double)Tq * T * dh, s);
        if (nw >= 8) hipLaunchKernelGGL((mha_fwd_kernel<2, true, 0>), dim3(B * H), dim3(nw * 64), lds, s, q, ldq, k, v, ldkv, key_mask, out, Tq, T, H, scale, lse, drop_p, drop_seed, drop_id, 1);
        else hipLaunchKernelGGL((mha_fwd_kernel<4, true, 0>), dim3(B * H), dim3(nw * 64), lds, s, q, ldq, k, v, ldkv, key_mask, out, Tq, T, H, scale, lse, drop_p, drop_seed, drop_id, 1);
        AFM_CHECK_LAUNCH();
        return 0;
    }
    int nw, nchunk = 1;
    bool seg2 = false;
    const size_t lds_cap = 160 * 1024;
    if (group_waves < 0) {
        nw = nqb <= 4 ? 4 : (nqb <= 6 ? 6 : (nqb <= 8 ? 8 : 12));
    } else {
        int gw = group_waves;
        if (gw >= 100) {
            seg2 = true; gw -= 100;
            if (gw != 2 && gw != 4) return AFM_E_BADARG;
        } else if (gw != 0 && gw != 1 && gw != 2 && gw != 4 && gw != 6 && gw != 8 && gw != 12)
            return AFM_E_BADARG;
        // one 12-wave workgroup per (sample, head) when those alone fill the chip (every K / V block is split and staged once, three waves per
        // SIMD: profiles/r03_mha_timeline.txt); small launches (strong scaling: B = 4 per GPU): groups of four query blocks with the two key
        // segments of a block on two waves (8-wave workgroups, one per CU) while they fit the chip in one round, plain 4-wave groups otherwise.
        // Bit-identical either way.
        nw = gw ? gw : ((int64_t)B * H >= 128 ? 12 : AFM_MHA_DEFAULT_GROUP);
        if (nw > nqb) nw = nqb > 8 ? 12 : (nqb > 6 ? 8 : (nqb > 4 ? 6 : (nqb > 2 ? 4 : nqb)));      // the smallest group that covers the query blocks
        nchunk = (nqb + nw - 1) / nw;
        if (!gw && (int64_t)B * H < 128 && nkb >= 2) {
            // the two key segments of a query block on two waves: groups of two query blocks (4-wave workgroups, one wave per SIMD) while the
            // launch still fits the 256 CUs in one round (B = 4: 192 workgroups, 21.9 us against 24.9 for groups of four and 29.2 for round 3's
            // form in one call, profiles/r04_mha_timeline_small.txt), groups of four query blocks above that
            const int n2 = (nqb + 1) / 2, n4 = (nqb + 3) / 4;
            if (nqb >= 2 && (int64_t)B * H * n2 <= 256) { nw = 2; nchunk = n2; seg2 = true; }
            else if (nqb >= 3 && (int64_t)B * H * n4 <= 256) { nw = 4; nchunk = n4; seg2 = true; }
        }
        if (seg2 && nw != 2 && nw != 4) seg2 = false;                                              // (an explicit 100 + g clipped by nqb)
    }
    // the sequential form parks 8 KB per wave in LDS: long key sequences (cross-attention over N points) take fewer waves per workgroup
    while (!seg2 && split_mha_lds(nkb, nw, false) > lds_cap && nw > 1) nw = nw > 8 ? 8 : (nw > 6 ? 6 : (nw > 4 ? 4 : nw / 2));
    if (group_waves >= 0) nchunk = (nqb + nw - 1) / nw;
    const size_t lds = split_mha_lds(nkb, nw, seg2);
    if (lds > lds_cap) return AFM_E_UNSUPPORTED;
    AfmProf prof(AFM_PROF_MHA_SPLIT, 4.0 * B * H * (double)Tq * T * dh, s);
#define AFM_MHA_GO(NWV, S2) do { if (six) return launch_split_mha<NWV, S2, 6>(q, ldq, k, v, ldkv, key_mask, out, B, Tq, T, H, scale2, nchunk, lds, s, q_first); \
                                 return launch_split_mha<NWV, S2, 9>(q, ldq, k, v, ldkv, key_mask, out, B, Tq, T, H, scale2, nchunk, lds, s, q_first); } while (0)
    if (seg2) {
        if (nw == 2) AFM_MHA_GO(2, true);
        AFM_MHA_GO(4, true);
    }
    switch (nw) {
        case 1: AFM_MHA_GO(1, false);
        case 2: AFM_MHA_GO(2, false);
        case 4: AFM_MHA_GO(4, false);
        case 6: AFM_MHA_GO(6, false);
        case 8: AFM_MHA_GO(8, false);
        default: AFM_MHA_GO(12, false);
    }
#undef AFM_MHA_GO
}

}  // namespace

extern "C" int afm_mha_fwd_grouped(const float* qkv, const uint8_t* key_mask, float* out, int32_t B, int32_t T, int32_t H,
                                   int32_t dh, int32_t group_waves, void* stream) {
    if (group_waves < 0) return AFM_E_BADARG;
    const int D = H * dh;
    return mha_fwd_launch(qkv, 3 * D, qkv ? qkv + D : nullptr, qkv ? qkv + 2 * D : nullptr, 3 * D, key_mask, out, nullptr, B, T, T, H, dh, 0.0f, 0, 0, false,
                          group_waves, stream);
}

extern "C" int afm_mha_fwd_rows(const float* qkv, const uint8_t* key_mask, float* out, int32_t B, int32_t T, int32_t H, int32_t dh, int32_t q_first,
                                int32_t group_waves, void* stream) {
    if (group_waves < 0) return AFM_E_BADARG;
    const int D = H * dh;
    return mha_fwd_launch(qkv, 3 * D, qkv ? qkv + D : nullptr, qkv ? qkv + 2 * D : nullptr, 3 * D, key_mask, out, nullptr, B, T, T, H, dh, 0.0f, 0, 0, false,
                          group_waves, stream, q_first);
}

extern "C" int afm_mha_fwd_arith(const float* qkv, const uint8_t* key_mask, float* out, int32_t B, int32_t T, int32_t H, int32_t dh, int32_t q_first,
                                 int32_t group_waves, int32_t arith, void* stream) {
    if (group_waves < 0) return AFM_E_BADARG;
    const int D = H * dh;
    return mha_fwd_launch(qkv, 3 * D, qkv ? qkv + D : nullptr, qkv ? qkv + 2 * D : nullptr, 3 * D, key_mask, out, nullptr, B, T, T, H, dh, 0.0f, 0, 0, false,
                          group_waves, stream, q_first, arith);
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

extern "C" int afm_mha_cross_fwd_train(const float* q, const float* kv, const uint8_t* key_mask, float* out, float* lse, int32_t B, int32_t Tq,
                                       int32_t Tk, int32_t H, int32_t dh, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* stream) {
    const int D = H * dh;
    return mha_fwd_launch(q, D, kv, kv ? kv + D : nullptr, 2 * D, key_mask, out, lse, B, Tq, Tk, H, dh, drop_p, drop_seed, drop_id, true, -1, stream);
}
