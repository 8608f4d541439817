// afm_mha_bwd: backward of softmax(Q K^T / sqrt(dh) + key_mask) V for the CMDM encoder layers (training path).
//
// Flash-style: the [T,T] probabilities are never stored; both passes recompute P = exp(S - lse) from the forward's
// log-sum-exp.  With D_q = rowsum(dO o O):
//     dP = dO V^T (o keep/(1-p) under attention dropout),  dS = P o (dP - D),
//     dQ = scale * dS K,   dK = scale * dS^T Q,   dV = P_drop^T dO
//
// gfx950 design (same conventions as attention.hip: one workgroup per (sample, head), v_mfma_f32_32x32x2_f32,
// "swapped" products so the softmax axis is lane-local and P / dS registers are directly the next MFMA's B operand):
//   pass 1 (dQ): one wave per 32-query block, K/V streamed through double-buffered LDS blocks
//       S^T = K Q^T, dP^T = V dO^T   (A = K / V rows from LDS as ds_read_b128, B = Q / dO rows in registers)
//       dQ^T += K^T dS^T             (A = K columns from LDS, B = the dS registers)
//     also writes D_q for pass 2.
//   pass 2 (dK, dV): one wave per 32-key block, Q/dO streamed through double-buffered LDS blocks, the wave's own
//     (pre-scaled) K block in a private LDS region, its V rows in registers
//       S = Q K^T, dP = dO V^T       (A = Q / dO rows from LDS, B = K from LDS / V from registers)
//       dV^T += dO^T P_drop, dK^T += Q^T dS   (A = dO / Q columns from LDS, B = the P / dS registers)
//   Two passes recompute S twice (7 products instead of 5) but need no atomics and no cross-wave reduction: every
//   gradient element has exactly one owner, so the result is deterministic.
#include <atomic>
#include "common.h"
#include "profile.h"
#include <math.h>

namespace {

constexpr int DH = 64;
constexpr int KB = 32;
constexpr int LDP = 68;            // padded LDS row (floats): conflict-free ds_read_b128 rows and b32 columns
constexpr int MAX_WAVES = 8;        // 2 waves per SIMD -> 256 VGPRs each (the dK/dV pass holds 4 accumulator tiles + V + S + dP)

// Tq queries (rows of q, stride ldq) over Tk keys / values (rows of k / v, stride ldkv).  Self-attention passes the packed in_proj output and
// its gradient three times (q | k | v, every stride 3D, Tq == Tk); cross-attention (nn.TransformerDecoderLayer.multihead_attn of the CMDM's
// trans_dec variant, cmdm.py:78-113) a [B, Tq, D] query and a packed [B, Tk, 2D] memory projection.  out / dout are [B, Tq, D], lse / Dws [B H, Tq].
struct MhaBwd {
    const float* q; int ldq;
    const float* k; const float* v; int ldkv;
    const uint8_t* key_mask;                 // [B, Tk] or null
    const float* out; const float* dout; const float* lse;
    float* dq; int lddq;
    float* dk; float* dv; int lddkv;
    float* Dws;
    int Tq, Tk, H;
    float scale, drop_p; uint64_t drop_seed; uint32_t drop_id;
};

// 32x32 product over the 64 head dims with the k index permuted per lane half (as in attention.hip):
// acc[reg r] (+)= sum_d rows[row (l&31)][32*hh + d] * breg[d]
__device__ __forceinline__ void prod_rows(const float* __restrict__ rows_lds /* + r32*LDP + hh*32 */, const float (&breg)[32], f32x16& acc) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(rows_lds + i * 4);
        acc = mfma32(a.x, breg[4 * i + 0], acc);
        acc = mfma32(a.y, breg[4 * i + 1], acc);
        acc = mfma32(a.z, breg[4 * i + 2], acc);
        acc = mfma32(a.w, breg[4 * i + 3], acc);
    }
}

// ------------------------------------------------------------------------------------------------ pass 1: dQ, D
template <int NST, bool DROP>
__global__ __launch_bounds__(64 * MAX_WAVES) void mha_bwd_dq_kernel(const MhaBwd a) {
    const uint8_t* __restrict__ key_mask = a.key_mask;
    const float* __restrict__ out = a.out;
    const float* __restrict__ dout = a.dout;
    const float* __restrict__ lse = a.lse;
    float* __restrict__ Dws = a.Dws;
    const int Tq = a.Tq, T = a.Tk, H = a.H;
    const float scale = a.scale, drop_p = a.drop_p;
    const uint64_t drop_seed = a.drop_seed;
    const uint32_t drop_id = a.drop_id;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                               // [2][KB][LDP]
    float* Vs = smem + 2 * KB * LDP;                // [2][KB][LDP]
    float* madd = Vs + 2 * KB * LDP;                // [nkb*KB]
    int* blk_valid = reinterpret_cast<int*>(madd + ((T + KB - 1) / KB) * KB);

    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int r32 = lane & 31, hh = lane >> 5;
    const int D = H * DH;
    const int nkb = (T + KB - 1) / KB, nqb = (Tq + KB - 1) / KB;
    const float* qbase = a.q + (int64_t)b * Tq * a.ldq + h * DH;
    const float* kbase = a.k + (int64_t)b * T * a.ldkv + h * DH;
    const float* vbase = a.v + (int64_t)b * T * a.ldkv + h * DH;
    const float NEG_INF = -INFINITY;
    const DropKey dk(DROP ? drop_p : 0.0f, drop_seed, drop_id);

    for (int i = tid; i < nkb; i += blockDim.x) blk_valid[i] = 0;
    __syncthreads();
    for (int i = tid; i < nkb * KB; i += blockDim.x) {
        const bool ok = (i < T) && !(key_mask && key_mask[(int64_t)b * T + i]);
        madd[i] = ok ? 0.0f : NEG_INF;
        if (ok) blk_valid[i / KB] = 1;
    }

    float4 stage[NST];
    auto load_block = [&](int kb) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * blockDim.x;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 1024) {
                const int isv = e >> 9, row = (e & 511) >> 4, c4 = e & 15;
                const int key = kb * KB + row;
                if (key < T) v = *reinterpret_cast<const float4*>((isv ? vbase : kbase) + (int64_t)key * a.ldkv + c4 * 4);
            }
            stage[i] = v;
        }
    };
    auto store_block = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * blockDim.x;
            if (e < 1024) {
                const int isv = e >> 9, row = (e & 511) >> 4, c4 = e & 15;
                float* dst = (isv ? Vs : Ks) + (buf * KB + row) * LDP + c4 * 4;
                *reinterpret_cast<float4*>(dst) = stage[i];
            }
        }
    };

    for (int q0 = 0; q0 < nqb; q0 += nw) {
        const int qb = q0 + wave;
        const bool active = qb < nqb;
        const int qrow = min((active ? qb : 0) * 32 + r32, Tq - 1);
        float q[32], dO[32];
        float lse_q, D_q;
        {
            const float* qp = qbase + (int64_t)qrow * a.ldq + hh * 32;
            const float* dop = dout + ((int64_t)b * Tq + qrow) * D + h * DH + hh * 32;
            const float* op = out + ((int64_t)b * Tq + qrow) * D + h * DH + hh * 32;
            float dsum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(qp + i * 4);
                q[4 * i + 0] = v.x * scale; q[4 * i + 1] = v.y * scale; q[4 * i + 2] = v.z * scale; q[4 * i + 3] = v.w * scale;
                const float4 g = *reinterpret_cast<const float4*>(dop + i * 4);
                dO[4 * i + 0] = g.x; dO[4 * i + 1] = g.y; dO[4 * i + 2] = g.z; dO[4 * i + 3] = g.w;
                const float4 o = *reinterpret_cast<const float4*>(op + i * 4);
                dsum += (g.x * o.x + g.y * o.y) + (g.z * o.z + g.w * o.w);
            }
            D_q = dsum + xor32(dsum);
            lse_q = lse[(int64_t)blockIdx.x * Tq + qrow];
            if (active && hh == 0 && qb * 32 + r32 < Tq) Dws[(int64_t)blockIdx.x * Tq + qrow] = D_q;
        }
        f32x16 dq0, dq1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dq0[r] = 0.f; dq1[r] = 0.f; }

        __syncthreads();
        load_block(0);
        store_block(0);
        __syncthreads();

        for (int kb = 0; kb < nkb; ++kb) {
            const int buf = kb & 1;
            if (kb + 1 < nkb) load_block(kb + 1);
            if (active && blk_valid[kb]) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                prod_rows(Ks + (buf * KB + r32) * LDP + hh * 32, q, s);          // S^T  (already scaled)
                __builtin_amdgcn_sched_barrier(0);
                prod_rows(Vs + (buf * KB + r32) * LDP + hh * 32, dO, dp);        // dP^T
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t row_ix = blockIdx.x * Tq + qrow, col0 = kb * KB + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 ma = *reinterpret_cast<const float4*>(madd + kb * KB + 8 * g + 4 * hh);
                    const float mm[4] = {ma.x, ma.y, ma.z, ma.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const float p = __expf(s[r] + mm[e] - lse_q);
                        float dpe = dp[r];
                        if (DROP) dpe *= dk(row_ix, col0 + 8 * g + e);
                        s[r] = p * (dpe - D_q);                                   // dS^T
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const float* kp = Ks + buf * KB * LDP + r32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const float k0 = kp[key * LDP], k1 = kp[key * LDP + 32];
                    dq0 = mfma32(k0, s[r], dq0);
                    dq1 = mfma32(k1, s[r], dq1);
                }
            }
            if (kb + 1 < nkb) store_block(buf ^ 1);
            __syncthreads();
        }

        if (active && qb * 32 + r32 < Tq) {
            float* gp = a.dq + ((int64_t)b * Tq + qb * 32 + r32) * a.lddq + h * DH + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float4*>(gp + 8 * g) =
                    make_float4(dq0[4 * g] * scale, dq0[4 * g + 1] * scale, dq0[4 * g + 2] * scale, dq0[4 * g + 3] * scale);
                *reinterpret_cast<float4*>(gp + 32 + 8 * g) =
                    make_float4(dq1[4 * g] * scale, dq1[4 * g + 1] * scale, dq1[4 * g + 2] * scale, dq1[4 * g + 3] * scale);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ pass 2: dK, dV
template <int NST, bool DROP>
__global__ __launch_bounds__(64 * MAX_WAVES) void mha_bwd_dkv_kernel(const MhaBwd a) {
    const uint8_t* __restrict__ key_mask = a.key_mask;
    const float* __restrict__ dout = a.dout;
    const float* __restrict__ lse = a.lse;
    const float* __restrict__ Dws = a.Dws;
    const int Tq = a.Tq, T = a.Tk, H = a.H;
    const float scale = a.scale, drop_p = a.drop_p;
    const uint64_t drop_seed = a.drop_seed;
    const uint32_t drop_id = a.drop_id;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nw = blockDim.x >> 6;
    float* Qs = smem;                               // [2][KB][LDP]
    float* Gs = smem + 2 * KB * LDP;                // [2][KB][LDP]   dO blocks
    float* lseS = Gs + 2 * KB * LDP;                // [2][KB]
    float* DS = lseS + 2 * KB;                      // [2][KB]
    float* Kw = DS + 2 * KB;                        // [nw][KB][LDP]  per-wave K block, pre-scaled

    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r32 = lane & 31, hh = lane >> 5;
    const int D = H * DH;
    const int nqb = (Tq + KB - 1) / KB, nkb = (T + KB - 1) / KB;
    const float* qbase = a.q + (int64_t)b * Tq * a.ldq + h * DH;
    const float* kbase = a.k + (int64_t)b * T * a.ldkv + h * DH;
    const float* vbase = a.v + (int64_t)b * T * a.ldkv + h * DH;
    const float* gbase = dout + (int64_t)b * Tq * D + h * DH;
    const float NEG_INF = -INFINITY;
    const DropKey dk(DROP ? drop_p : 0.0f, drop_seed, drop_id);

    // cooperative Q / dO block loader (+ lse / D of the block's queries)
    float4 stage[NST];
    float stage_x = 0.f;
    auto load_block = [&](int qb) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * blockDim.x;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < 1024) {
                const int isg = e >> 9, row = (e & 511) >> 4, c4 = e & 15;
                const int qr = qb * KB + row;
                if (qr < Tq) v = isg ? *reinterpret_cast<const float4*>(gbase + (int64_t)qr * D + c4 * 4)
                                     : *reinterpret_cast<const float4*>(qbase + (int64_t)qr * a.ldq + c4 * 4);
            }
            stage[i] = v;
        }
        if (tid < 64) {
            const int qr = qb * KB + (tid & 31);
            // padded queries: lse = +inf makes their probability exactly 0
            stage_x = (qr < Tq) ? ((tid < 32) ? lse[(int64_t)blockIdx.x * Tq + qr] : Dws[(int64_t)blockIdx.x * Tq + qr]) : ((tid < 32) ? INFINITY : 0.f);
        }
    };
    auto store_block = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int e = tid + i * blockDim.x;
            if (e < 1024) {
                const int isg = e >> 9, row = (e & 511) >> 4, c4 = e & 15;
                float* dst = (isg ? Gs : Qs) + (buf * KB + row) * LDP + c4 * 4;
                *reinterpret_cast<float4*>(dst) = stage[i];
            }
        }
        if (tid < 64) ((tid < 32) ? lseS : DS)[buf * KB + (tid & 31)] = stage_x;
    };

    for (int k0 = 0; k0 < nkb; k0 += nw) {
        const int kblk = k0 + wave;
        const bool active = kblk < nkb;
        const int key = min((active ? kblk : 0) * KB + r32, T - 1);
        const bool key_ok = active && (kblk * KB + r32 < T) && !(key_mask && key_mask[(int64_t)b * T + key]);
        const float madd_key = key_ok ? 0.0f : NEG_INF;
        const bool any_valid = __any(key_ok);
        float v[32];
        {
            const float* vp = vbase + (int64_t)key * a.ldkv + hh * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(vp + i * 4);
                v[4 * i + 0] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
            }
        }
        __syncthreads();                 // previous pass done with Kw / the Q blocks
        {
            // this wave's K block -> its private LDS region, pre-scaled: lane (row r32, half hh) copies 32 floats
            const float* kp = kbase + (int64_t)key * a.ldkv + hh * 32;
            float* kd = Kw + (wave * KB + r32) * LDP + hh * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float4 t = *reinterpret_cast<const float4*>(kp + i * 4);
                t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale;
                *reinterpret_cast<float4*>(kd + i * 4) = t;
            }
        }
        f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk0[r] = 0.f; dk1[r] = 0.f; dv0[r] = 0.f; dv1[r] = 0.f; }

        load_block(0);
        store_block(0);
        __syncthreads();

        for (int qb = 0; qb < nqb; ++qb) {
            const int buf = qb & 1;
            if (qb + 1 < nqb) load_block(qb + 1);
            if (active && any_valid) {
                f32x16 s, dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
                const float* qrows = Qs + (buf * KB + r32) * LDP + hh * 32;
                const float* grows = Gs + (buf * KB + r32) * LDP + hh * 32;
                const float* krow = Kw + (wave * KB + r32) * LDP + hh * 32;
                // S = Q K^T: both operands from LDS (A = Q rows, B = this lane's scaled K row)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float4 a = *reinterpret_cast<const float4*>(qrows + i * 4);
                    const float4 kk = *reinterpret_cast<const float4*>(krow + i * 4);
                    s = mfma32(a.x, kk.x, s);
                    s = mfma32(a.y, kk.y, s);
                    s = mfma32(a.z, kk.z, s);
                    s = mfma32(a.w, kk.w, s);
                }
                __builtin_amdgcn_sched_barrier(0);
                prod_rows(grows, v, dp);                                          // dP = dO V^T
                __builtin_amdgcn_sched_barrier(0);
                // reg r <-> query qb*32 + (r&3) + 8*(r>>2) + 4*hh, key = this lane's
                const uint32_t row0 = blockIdx.x * Tq + qb * KB + 4 * hh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 l4 = *reinterpret_cast<const float4*>(lseS + buf * KB + 8 * g + 4 * hh);
                    const float4 d4 = *reinterpret_cast<const float4*>(DS + buf * KB + 8 * g + 4 * hh);
                    const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * g + e;
                        const float p = __expf(s[r] + madd_key - ll[e]);
                        if (DROP) {
                            const float keep = dk(row0 + 8 * g + e, (uint32_t)key);
                            dp[r] = p * (dp[r] * keep - dd[e]);                   // dS
                            s[r] = p * keep;                                      // P_drop
                        } else {
                            dp[r] = p * (dp[r] - dd[e]);
                            s[r] = p;
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const float* gcol = Gs + buf * KB * LDP + r32;
                const float* qcol = Qs + buf * KB * LDP + r32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const float g0 = gcol[qi * LDP], g1 = gcol[qi * LDP + 32];
                    const float q0 = qcol[qi * LDP], q1 = qcol[qi * LDP + 32];
                    dv0 = mfma32(g0, s[r], dv0);
                    dv1 = mfma32(g1, s[r], dv1);
                    dk0 = mfma32(q0, dp[r], dk0);
                    dk1 = mfma32(q1, dp[r], dk1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (qb + 1 < nqb) store_block(buf ^ 1);
            __syncthreads();
        }

        if (active && kblk * KB + r32 < T) {
            float* kp = a.dk + ((int64_t)b * T + kblk * KB + r32) * a.lddkv + h * DH + 4 * hh;
            float* vp = a.dv + ((int64_t)b * T + kblk * KB + r32) * a.lddkv + h * DH + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float4*>(kp + 8 * g) =
                    make_float4(dk0[4 * g] * scale, dk0[4 * g + 1] * scale, dk0[4 * g + 2] * scale, dk0[4 * g + 3] * scale);
                *reinterpret_cast<float4*>(kp + 32 + 8 * g) =
                    make_float4(dk1[4 * g] * scale, dk1[4 * g + 1] * scale, dk1[4 * g + 2] * scale, dk1[4 * g + 3] * scale);
                *reinterpret_cast<float4*>(vp + 8 * g) = make_float4(dv0[4 * g], dv0[4 * g + 1], dv0[4 * g + 2], dv0[4 * g + 3]);
                *reinterpret_cast<float4*>(vp + 32 + 8 * g) = make_float4(dv1[4 * g], dv1[4 * g + 1], dv1[4 * g + 2], dv1[4 * g + 3]);
            }
        }
    }
}

}  // namespace

namespace {

int mha_bwd_launch(MhaBwd a, int B, int dh, void* ws, int64_t ws_bytes, hipStream_t s) {
    if (dh != DH) return AFM_E_UNSUPPORTED;
    if (B == 0) return 0;
    if (!a.q || !a.k || !a.v || !a.out || !a.dout || !a.lse || !a.dq || !a.dk || !a.dv || B < 0 || a.Tq <= 0 || a.Tk <= 0 || a.H <= 0) return AFM_E_BADARG;
    if ((((uintptr_t)a.q) | ((uintptr_t)a.k) | ((uintptr_t)a.v) | ((uintptr_t)a.out) | ((uintptr_t)a.dout) | ((uintptr_t)a.dq) | ((uintptr_t)a.dk) |
         ((uintptr_t)a.dv)) & 15) return AFM_E_BADARG;
    if ((a.ldq | a.ldkv | a.lddq | a.lddkv) & 3) return AFM_E_BADARG;
    if (a.drop_p < 0.0f || a.drop_p >= 1.0f) return AFM_E_BADARG;
    if (!ws || ws_bytes < (int64_t)B * a.H * a.Tq * (int64_t)sizeof(float)) return AFM_E_WORKSPACE;
    a.Dws = (float*)ws;
    const int nqb = (a.Tq + 31) / 32, nkb = (a.Tk + 31) / 32;
    const int nw1 = nqb < 4 ? 4 : (nqb > MAX_WAVES ? MAX_WAVES : nqb);             // pass 1: a wave per query block
    const int nw2 = nkb < 4 ? 4 : (nkb > MAX_WAVES ? MAX_WAVES : nkb);             // pass 2: a wave per key block
    const size_t lds1 = (size_t)(4 * KB * LDP + nkb * KB) * sizeof(float) + (size_t)nkb * sizeof(int);
    const size_t lds2 = (size_t)(4 * KB * LDP + 4 * KB + nw2 * KB * LDP) * sizeof(float);
    if (lds1 > 160 * 1024 || lds2 > 160 * 1024) return AFM_E_UNSUPPORTED;
    a.scale = 1.0f / sqrtf((float)dh);
    const double flops_prod = 2.0 * B * a.H * (double)a.Tq * a.Tk * dh;
    static std::atomic<bool> attr_set{false};         // > 64 KB of dynamic LDS must be opted into once per kernel (idempotent)
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipSuccess;
        const void* fns[8] = {(const void*)mha_bwd_dkv_kernel<2, false>, (const void*)mha_bwd_dkv_kernel<4, false>,
                              (const void*)mha_bwd_dkv_kernel<2, true>, (const void*)mha_bwd_dkv_kernel<4, true>,
                              (const void*)mha_bwd_dq_kernel<2, false>, (const void*)mha_bwd_dq_kernel<4, false>,
                              (const void*)mha_bwd_dq_kernel<2, true>, (const void*)mha_bwd_dq_kernel<4, true>};
        for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set.store(true, std::memory_order_release);
    }
    const bool drop = a.drop_p > 0.0f;
#define AFM_DQ(NST, DR) hipLaunchKernelGGL((mha_bwd_dq_kernel<NST, DR>), dim3(B * a.H), dim3(nw1 * 64), lds1, s, a)
#define AFM_DKV(NST, DR) hipLaunchKernelGGL((mha_bwd_dkv_kernel<NST, DR>), dim3(B * a.H), dim3(nw2 * 64), lds2, s, a)
    {
        AfmProf prof(AFM_PROF_MHA_BWD_DQ, 3.0 * flops_prod, s);
        if (nw1 >= 8) { if (drop) AFM_DQ(2, true); else AFM_DQ(2, false); }
        else { if (drop) AFM_DQ(4, true); else AFM_DQ(4, false); }
        AFM_CHECK_LAUNCH();
    }
    {
        AfmProf prof(AFM_PROF_MHA_BWD_DKV, 4.0 * flops_prod, s);
        if (nw2 >= 8) { if (drop) AFM_DKV(2, true); else AFM_DKV(2, false); }
        else { if (drop) AFM_DKV(4, true); else AFM_DKV(4, false); }
        AFM_CHECK_LAUNCH();
    }
#undef AFM_DQ
#undef AFM_DKV
    return 0;
}

}  // namespace

extern "C" int afm_mha_bwd(const float* qkv, const uint8_t* key_mask, const float* out, const float* dout, const float* lse, float* dqkv,
                           int32_t B, int32_t T, int32_t H, int32_t dh, float drop_p, uint64_t drop_seed, uint32_t drop_id, void* ws,
                           int64_t ws_bytes, void* stream) {
    if (!qkv || !dqkv || H <= 0) return B == 0 ? 0 : AFM_E_BADARG;
    const int D = H * dh;
    MhaBwd a = {};
    a.q = qkv; a.ldq = 3 * D; a.k = qkv + D; a.v = qkv + 2 * D; a.ldkv = 3 * D; a.key_mask = key_mask; a.out = out; a.dout = dout; a.lse = lse;
    a.dq = dqkv; a.lddq = 3 * D; a.dk = dqkv + D; a.dv = dqkv + 2 * D; a.lddkv = 3 * D; a.Tq = T; a.Tk = T; a.H = H;
    a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_id = drop_id;
    return mha_bwd_launch(a, B, dh, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int afm_mha_cross_bwd(const float* q, const float* kv, const uint8_t* key_mask, const float* out, const float* dout, const float* lse,
                                 float* dq, float* dkv, int32_t B, int32_t Tq, int32_t Tk, int32_t H, int32_t dh, float drop_p, uint64_t drop_seed,
                                 uint32_t drop_id, void* ws, int64_t ws_bytes, void* stream) {
    if (!q || !kv || !dq || !dkv || H <= 0) return B == 0 ? 0 : AFM_E_BADARG;
    const int D = H * dh;
    MhaBwd a = {};
    a.q = q; a.ldq = D; a.k = kv; a.v = kv + D; a.ldkv = 2 * D; a.key_mask = key_mask; a.out = out; a.dout = dout; a.lse = lse;
    a.dq = dq; a.lddq = D; a.dk = dkv; a.dv = dkv + D; a.lddkv = 2 * D; a.Tq = Tq; a.Tk = Tk; a.H = H;
    a.drop_p = drop_p; a.drop_seed = drop_seed; a.drop_id = drop_id;
    return mha_bwd_launch(a, B, dh, ws, ws_bytes, (hipStream_t)stream);
}
