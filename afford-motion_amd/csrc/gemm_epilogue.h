// Shared GEMM epilogue (bias / activation / dropout / residual / row table / DDPM update) of afm_linear's kernels.
// The accumulators are staged in LDS as a [BM][BN + 4] f32 tile by the kernel; rows are then streamed out with 16-byte
// accesses so every side input is read coalesced and the epilogue math exists once per kernel.
#pragma once
#include "common.h"

#ifndef AFM_PK_PROBE        // instrumentation of the packed-f32 defect hunt (tools/probes/build_packed.sh builds probe libraries; never in the product)
#define AFM_PK_PROBE 0
#endif

namespace {

struct RowMap {
    int grp, stride, off, skip_after = 0, skip = 0;      // skip: the members j >= skip_after of a group sit `skip` rows further on (afm_linear_args.a_skip)
    __device__ __forceinline__ int64_t operator()(int r) const {
        if (!grp) return (int64_t)r;
        const int j = r % grp;
        return (int64_t)(r / grp) * stride + off + j + (j >= skip_after ? skip : 0);
    }
};

// 16-byte WRITE-THROUGH store (sc1): the tile of a fused-LayerNorm launch is read back by another workgroup, possibly on another XCD
// (private L2s).  Written through, it needs no release fence (cdna_hip_programming.md section 6 Guideline 16, recipe R1: a per-tile
// `buffer_wbl2` release cost 45 us per launch here).  The trailing s_nop keeps the data registers intact until the store has read them.
__device__ __forceinline__ void store_f4_sc1(float* ptr, const float4& v) {
    const f32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(d) : "memory");
}

// ---- LayerNorm folded across kernel boundaries (afm_linear_args.stat_out / a_stat / res_stat, ABI v5).  A producer writes, per output
// row and 64-column group, (mean, M2 = sum of squared deviations from that mean) of what it stores; consumers combine the groups of a row
// in index order (Chan's parallel-variance formula): mean = avg(mean_t), M2 = sum_t M2_t + 64 sum_t (mean_t - mean)^2.
// All records of the row are REQUESTED before the first is used: the loop over a run-time `groups` made hipcc wait for every load before it
// issued the next - 8 (d = 512) or 16 (ff = 1024) dependent round trips to rows the previous kernel has just written on other XCDs, 1.3 us
// at the head of every folded launch (profiles/r04_gemm_timeline_small.txt: prologue 1.1 -> 2.5 us).  Records past `groups` re-read the
// last one (unconditional loads, no predicate); the sums still run over the first `groups` records in index order: same bits as before.
constexpr int AFM_STAT_MAXG = 16;
__device__ __forceinline__ void row_stat_combine(const float* __restrict__ st, int groups, float eps, float& mean, float& rstd) {
    if (groups > AFM_STAT_MAXG) {                   // wider rows than the encoder's (not on the shipped path): the plain loop
        float ms = 0.f;
        for (int t = 0; t < groups; ++t) ms += st[2 * t];
        mean = ms / (float)groups;
        float m2 = 0.f;
        for (int t = 0; t < groups; ++t) { const float d = st[2 * t] - mean; m2 += st[2 * t + 1] + 64.0f * (d * d); }
        rstd = 1.0f / sqrtf(m2 / (64.0f * (float)groups) + eps);
        return;
    }
    float2 rec[AFM_STAT_MAXG];
#pragma unroll
    for (int t = 0; t < AFM_STAT_MAXG; ++t) rec[t] = *reinterpret_cast<const float2*>(st + 2 * min(t, groups - 1));
    float ms = 0.f;
#pragma unroll
    for (int t = 0; t < AFM_STAT_MAXG; ++t) if (t < groups) ms += rec[t].x;
    mean = ms / (float)groups;
    float m2 = 0.f;
#pragma unroll
    for (int t = 0; t < AFM_STAT_MAXG; ++t) if (t < groups) { const float d = rec[t].x - mean; m2 += rec[t].y + 64.0f * (d * d); }
    rstd = 1.0f / sqrtf(m2 / (64.0f * (float)groups) + eps);
}

// (mean, rstd) of the tile's A rows (thread r < BM -> rowst[r]) and of its residual rows (thread BM + r -> rowst[BM + r]), computed in
// the kernel's PROLOGUE into an LDS region of their own (2 * BM float2 behind the operand stages): the 16 loads of a row's group
// statistics fly with the first operand loads and the kernel's first barrier publishes them.  (As a separate phase in front of the
// epilogue they cost every tile a dependent memory round trip - GEMM time at B = 4: 0.49 -> 0.54 ms per step; parked in registers
// across the K loop they spilled the 128 x 128 kernel.)
template <int BM>
__device__ __forceinline__ void gemm_rowstats(const afm_linear_args& p, float* rowst, int bm) {
    if (!p.a_stat && !p.res_stat) return;                               // uniform
    const int r = threadIdx.x;
    if (r >= 2 * BM) return;
    const RowMap amap{p.a_grp, p.a_stride, p.a_off, p.a_skip_after, p.a_skip}, cmap{p.c_grp, p.c_stride, p.c_off, p.c_skip_after, p.c_skip};
    const bool is_res = r >= BM;
    const int grow = bm * BM + (is_res ? r - BM : r);
    const float* st = is_res ? p.res_stat : p.a_stat;
    float mean = 0.f, rstd = 0.f;
    if (st && grow < p.M) {
        const int groups = is_res ? p.N / 64 : p.a_stat_groups;
        row_stat_combine(st + (is_res ? cmap(grow) : amap(grow)) * (2 * groups), groups, p.ln_eps2, mean, rstd);
    }
    rowst[2 * r] = mean; rowst[2 * r + 1] = rstd;
}

// epilogue math for one output element (compact: instantiated once, looped over, never unrolled 64x)
__device__ __forceinline__ float epilogue(const afm_linear_args& p, float v, int grow, int64_t orow, int gcol, const float* rowst = nullptr, int row = 0, int BM = 0) {
    if (p.a_stat) v = rowst[2 * row + 1] * (v - rowst[2 * row] * p.a_fold_g[gcol]);       // LayerNorm of the A rows, folded: W carries gamma, bias carries W beta
    if (p.scale) v *= p.scale[gcol];
    if (p.bias) v += p.bias[gcol];
    if (p.preact) p.preact[orow * p.ldp + gcol] = v;
    if (p.act) v = apply_act(v, p.act);
    if (p.drop_p > 0.0f && !p.drop_after) v *= DropKey(p.drop_p, p.drop_seed, p.drop_id)((uint32_t)orow, (uint32_t)gcol);
    if (p.dact) v *= act_grad(p.dact_z[orow * p.ldz + gcol], p.dact);
    if (p.residual) {
        float rv = p.residual[orow * p.ldr + gcol];
        if (p.res_stat) rv = (rv - rowst[2 * (BM + row)]) * rowst[2 * (BM + row) + 1] * p.res_gamma[gcol] + p.res_beta[gcol];      // the residual rows are raw: LayerNorm on the fly
        v += rv;
    }
    if (p.rowtab) v += p.rowtab[(int64_t)(grow % p.rowtab_period) * p.N + gcol];
    if (p.act_post) v = apply_act(v, p.act_post);
    if (p.drop_p > 0.0f && p.drop_after) v *= DropKey(p.drop_p, p.drop_seed, p.drop_id)((uint32_t)orow, (uint32_t)gcol);
    return v;
}

// Shared epilogue: the accumulators were staged in `lds` as a [BM][BN + 4] tile; stream rows out with 16-byte accesses.
template <int BM, int BN, int NT = 256, int LDC = BN + 4>
__device__ __forceinline__ void gemm_epilogue(const afm_linear_args& p, const float* lds, int bm, int bn, int tid, const float* rowst = nullptr) {
    const RowMap cmap{p.c_grp, p.c_stride, p.c_off, p.c_skip_after, p.c_skip};
    const int col0 = bn * BN;
    const bool vec_out = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && ((p.ldr & 3) == 0) && ((p.ldp & 3) == 0) && ((p.ldz & 3) == 0) && !p.ddpm_out &&
                         ((((uintptr_t)p.C) | ((uintptr_t)p.residual) | ((uintptr_t)p.bias) | ((uintptr_t)p.scale) | ((uintptr_t)p.rowtab) |
                           ((uintptr_t)p.preact) | ((uintptr_t)p.dact_z)) & 15) == 0;
    const bool drop = p.drop_p > 0.0f;
    const DropKey dk(drop ? p.drop_p : 0.0f, p.drop_seed, p.drop_id);
    if (vec_out && p.rowdot_w) {
        // row-dot form: every 64-column group of a row is owned by 16 consecutive lanes (4 columns each); all lanes stay in the loop
        // (out-of-range ones contribute zeros) so that the xor-butterfly below is a full-wave operation
        static_assert((BM * (BN / 4)) % NT == 0, "row-dot epilogue: every lane makes the same number of trips");
        const int ngrp = (p.N + 63) / 64;
        // a thread's column quad is the same on every trip (NT is a multiple of BN / 4): its slices of the R vectors live in registers
        float4 rw[8];
        {
            const int gc = col0 + (tid % (BN / 4)) * 4;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                rw[r] = (r < p.rowdot_n && gc < p.N) ? *reinterpret_cast<const float4*>(p.rowdot_w + (int64_t)r * p.N + gc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int e = tid; (BN / 4) % 16 == 0 && e < BM * (BN / 4); e += NT) {      // 32-column tiles are never dispatched with a row-dot
            const int row = e / (BN / 4), cq = (e % (BN / 4)) * 4;
            const int grow = bm * BM + row, gcol = col0 + cq;
            const bool valid = grow < p.M && gcol < p.N;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int64_t orow = 0;
            if (valid) {
                v = *reinterpret_cast<const float4*>(lds + row * LDC + cq);
                orow = cmap(grow);
                if (p.scale) { const float4 t = *reinterpret_cast<const float4*>(p.scale + gcol); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
                if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + gcol); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                if (p.act) { v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act); }
                if (p.residual) { const float4 t = *reinterpret_cast<const float4*>(p.residual + orow * p.ldr + gcol); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                if (p.act_post) { v.x = apply_act(v.x, p.act_post); v.y = apply_act(v.y, p.act_post); v.z = apply_act(v.z, p.act_post); v.w = apply_act(v.w, p.act_post); }
                if (p.C) *reinterpret_cast<float4*>(p.C + orow * p.ldc + gcol) = v;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (r < p.rowdot_n) {                       // wave-uniform
                    float d = (v.x * rw[r].x + v.y * rw[r].y) + (v.z * rw[r].z + v.w * rw[r].w);      // v == 0 on out-of-range lanes
                    d += lane_xor<1>(d); d += lane_xor<2>(d); d += lane_xor<4>(d); d += lane_xor<8>(d);      // DPP inside the row of 16 lanes
                    if (valid && (cq & 63) == 0) p.rowdot_out[(orow * ngrp + gcol / 64) * p.rowdot_n + r] = d;
                }
            }
        }
    } else if (vec_out && (p.stat_out || p.a_stat || p.res_stat)) {
        // folded-LayerNorm form (plain forward inputs only): every lane stays in the loop so that the group statistics are full 16-lane
        // butterflies (a 64-column group of a row = 16 consecutive lanes x 4 columns; N % 64 == 0)
        static_assert((BM * (BN / 4)) % NT == 0, "folded-LayerNorm epilogue: every lane makes the same number of trips");
        const int ngrp = p.N / 64;
        for (int e = tid; (BN / 4) % 16 == 0 && e < BM * (BN / 4); e += NT) {
            const int row = e / (BN / 4), cq = (e % (BN / 4)) * 4;
            const int grow = bm * BM + row, gcol = col0 + cq;
            const bool valid = grow < p.M && gcol < p.N;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            int64_t orow = 0;
            if (valid) {
                v = *reinterpret_cast<const float4*>(lds + row * LDC + cq);
                orow = cmap(grow);
                if (p.a_stat) {
                    const float mu = rowst[2 * row], rs = rowst[2 * row + 1];
                    const float4 g = *reinterpret_cast<const float4*>(p.a_fold_g + gcol);
                    v.x = rs * (v.x - mu * g.x); v.y = rs * (v.y - mu * g.y); v.z = rs * (v.z - mu * g.z); v.w = rs * (v.w - mu * g.w);
                }
                if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + gcol); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                if (p.act) { v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act); }
                if (p.residual) {
                    float4 t = *reinterpret_cast<const float4*>(p.residual + orow * p.ldr + gcol);
                    if (p.res_stat) {
                        const float mu = rowst[2 * (BM + row)], rs = rowst[2 * (BM + row) + 1];
                        const float4 g = *reinterpret_cast<const float4*>(p.res_gamma + gcol), b = *reinterpret_cast<const float4*>(p.res_beta + gcol);
#if AFM_PK_PROBE & 1       // (tools/probes/build_packed.sh only) every load of the trip has landed, plus 16 idle cycles, before the packed arithmetic starts
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#endif
#if AFM_PK_PROBE & 4       // the packed arithmetic's inputs pinned in registers of their own (no register reuse between the loads and the packed ops)
                        asm volatile("" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
#endif
#if AFM_PK_PROBE & (16 | 32 | 64 | 128)
                        // the instruction forms themselves, written out (profiles/r04_packed_f32_repro.md): (r - mean) * rstd on the pair {mean, rstd}
                        //   16  the form hipcc emits and that fails: v_pk_add (mean = low half of src1 for both results), then v_pk_mul with
                        //       op_sel:[0,1] (rstd = HIGH half of src1 feeding the LOW result)
                        //   32  the same with four idle cycles between the adds and the multiplies
                        //   64  the commuted form of the passing build: the pair as src0, op_sel:[1,0]
                        //  128  no op_sel at all: rstd replicated into a register pair of its own
                        {
                            typedef float pk2 __attribute__((ext_vector_type(2)));
                            pk2 t01 = {t.x, t.y}, t23 = {t.z, t.w};
                            const pk2 ms = {mu, rs}, rr = {rs, rs};
#if AFM_PK_PROBE & 16
                            asm volatile("v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                                         "v_pk_mul_f32 %0, %0, %2 op_sel:[0,1]\n\tv_pk_mul_f32 %1, %1, %2 op_sel:[0,1]" : "+v"(t01), "+v"(t23) : "v"(ms), "v"(rr));
#elif AFM_PK_PROBE & 32
                            asm volatile("v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 3\n\t"
                                         "v_pk_mul_f32 %0, %0, %2 op_sel:[0,1]\n\tv_pk_mul_f32 %1, %1, %2 op_sel:[0,1]" : "+v"(t01), "+v"(t23) : "v"(ms), "v"(rr));
#elif AFM_PK_PROBE & 64
                            asm volatile("v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                                         "v_pk_mul_f32 %0, %2, %0 op_sel:[1,0]\n\tv_pk_mul_f32 %1, %2, %1 op_sel:[1,0]" : "+v"(t01), "+v"(t23) : "v"(ms), "v"(rr));
#else
                            asm volatile("v_pk_add_f32 %0, %0, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                                         "v_pk_mul_f32 %0, %0, %3\n\tv_pk_mul_f32 %1, %1, %3" : "+v"(t01), "+v"(t23) : "v"(ms), "v"(rr));
#endif
                            t.x = t01.x * g.x + b.x; t.y = t01.y * g.y + b.y; t.z = t23.x * g.z + b.z; t.w = t23.y * g.w + b.w;
                        }
#else
                        t.x = (t.x - mu) * rs * g.x + b.x; t.y = (t.y - mu) * rs * g.y + b.y; t.z = (t.z - mu) * rs * g.z + b.z; t.w = (t.w - mu) * rs * g.w + b.w;
#endif
#if AFM_PK_PROBE & 8       // 16 idle cycles BEHIND the packed arithmetic: nothing may overwrite its source registers while it is still reading them
                        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(t.x), "+v"(t.y), "+v"(t.z), "+v"(t.w));
#endif
                    }
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                *reinterpret_cast<float4*>(p.C + orow * p.ldc + gcol) = v;
#if AFM_PK_PROBE & 2       // the trip's stores drained and no load of the NEXT trip issued before this point
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            }
            if (p.stat_out) {                       // uniform
                float sm = (v.x + v.y) + (v.z + v.w);
                sm += lane_xor<1>(sm); sm += lane_xor<2>(sm); sm += lane_xor<4>(sm); sm += lane_xor<8>(sm);
                const float mu = sm * (1.0f / 64.0f);
                const float a = v.x - mu, b = v.y - mu, c = v.z - mu, d = v.w - mu;
                float m2 = (a * a + b * b) + (c * c + d * d);
                m2 += lane_xor<1>(m2); m2 += lane_xor<2>(m2); m2 += lane_xor<4>(m2); m2 += lane_xor<8>(m2);
                if (valid && (cq & 63) == 0) *reinterpret_cast<float2*>(p.stat_out + (orow * ngrp + gcol / 64) * 2) = make_float2(mu, m2);
            }
        }
    } else if (vec_out) {
        for (int e = tid; e < BM * (BN / 4); e += NT) {
            const int row = e / (BN / 4), cq = (e % (BN / 4)) * 4;
            const int grow = bm * BM + row, gcol = col0 + cq;
            if (grow >= p.M || gcol >= p.N) continue;
            float4 v = *reinterpret_cast<const float4*>(lds + row * LDC + cq);
            const int64_t orow = cmap(grow);
            if (p.scale) { const float4 t = *reinterpret_cast<const float4*>(p.scale + gcol); v.x *= t.x; v.y *= t.y; v.z *= t.z; v.w *= t.w; }
            if (p.bias) { const float4 t = *reinterpret_cast<const float4*>(p.bias + gcol); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (p.preact) *reinterpret_cast<float4*>(p.preact + orow * p.ldp + gcol) = v;
            if (p.act) { v.x = apply_act(v.x, p.act); v.y = apply_act(v.y, p.act); v.z = apply_act(v.z, p.act); v.w = apply_act(v.w, p.act); }
            if (drop && !p.drop_after) { const uint32_t ro = (uint32_t)orow, co = (uint32_t)gcol; v.x *= dk(ro, co); v.y *= dk(ro, co + 1); v.z *= dk(ro, co + 2); v.w *= dk(ro, co + 3); }
            if (p.dact) { const float4 t = *reinterpret_cast<const float4*>(p.dact_z + orow * p.ldz + gcol);
                          v.x *= act_grad(t.x, p.dact); v.y *= act_grad(t.y, p.dact); v.z *= act_grad(t.z, p.dact); v.w *= act_grad(t.w, p.dact); }
            if (p.residual) { const float4 t = *reinterpret_cast<const float4*>(p.residual + orow * p.ldr + gcol); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (p.rowtab) { const float4 t = *reinterpret_cast<const float4*>(p.rowtab + (int64_t)(grow % p.rowtab_period) * p.N + gcol); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (p.act_post) { v.x = apply_act(v.x, p.act_post); v.y = apply_act(v.y, p.act_post); v.z = apply_act(v.z, p.act_post); v.w = apply_act(v.w, p.act_post); }
            if (drop && p.drop_after) { const uint32_t ro = (uint32_t)orow, co = (uint32_t)gcol; v.x *= dk(ro, co); v.y *= dk(ro, co + 1); v.z *= dk(ro, co + 2); v.w *= dk(ro, co + 3); }
            if (p.ln_out) store_f4_sc1(p.C + orow * p.ldc + gcol, v);           // uniform: published to the row block's last arriver
            else *reinterpret_cast<float4*>(p.C + orow * p.ldc + gcol) = v;
        }
    } else if (p.ddpm_out && !p.scale && !p.preact && !p.act && !p.dact && !drop && !p.residual && !p.rowtab && !p.act_post && !p.ln_out) {
        // The sampling loop's output layer (N = 263: rows are not 16-byte multiples, so this is the scalar branch; at one sample per GPU its
        // 8 dependent trips of ~1 us each were half of the launch).  Four elements per trip: every load of the four is issued before the
        // first is used (addresses clamped instead of predicated), the arithmetic per element is the general branch's, expression for expression.
        constexpr int U = 4;
        for (int e0 = tid; e0 < BM * BN; e0 += U * NT) {
            float v[U], xt[U], nz[U], c1[U], c2[U], sg[U], ga[U], bi[U], mu[U], rs[U];
            int64_t orow[U], ix[U];
            int gc[U];
            bool ok[U];
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int e = min(e0 + j * NT, BM * BN - 1);
                const int row = e / BN, c = e % BN;
                const int grow = bm * BM + row, gcol = col0 + c;
                ok[j] = e0 + j * NT < BM * BN && grow < p.M && gcol < p.N;
                const int gr = min(grow, p.M - 1);
                gc[j] = min(gcol, p.N - 1);
                orow[j] = cmap(gr);
                ix[j] = orow[j] * p.ldx + gc[j];
                const int b = gr / p.rows_per_sample;
                v[j] = lds[row * LDC + c];
                mu[j] = p.a_stat ? rowst[2 * row] : 0.f;
                rs[j] = p.a_stat ? rowst[2 * row + 1] : 0.f;
                ga[j] = p.a_stat ? p.a_fold_g[gc[j]] : 0.f;
                bi[j] = p.bias ? p.bias[gc[j]] : 0.f;
                xt[j] = p.ddpm_xt[ix[j]]; nz[j] = p.ddpm_noise[ix[j]];
                c1[j] = p.ddpm_c1[b]; c2[j] = p.ddpm_c2[b]; sg[j] = p.ddpm_sigma[b];
            }
#pragma unroll
            for (int j = 0; j < U; ++j) {
                float w = v[j];
                if (p.a_stat) w = rs[j] * (w - mu[j] * ga[j]);
                if (p.bias) w += bi[j];
                if (p.ddpm_clip) w = w < -1.0f ? -1.0f : (w > 1.0f ? 1.0f : w);
                const float xn = (c1[j] * w + c2[j] * xt[j]) + sg[j] * nz[j];
                if (ok[j]) {
                    if (p.C) p.C[orow[j] * p.ldc + gc[j]] = w;
                    p.ddpm_out[ix[j]] = xn;
                    if (p.ddpm_out2) p.ddpm_out2[orow[j] * p.ldx2 + gc[j]] = xn;
                }
            }
        }
    } else {
        for (int e = tid; e < BM * BN; e += NT) {
            const int row = e / BN, c = e % BN;
            const int grow = bm * BM + row, gcol = col0 + c;
            if (grow >= p.M || gcol >= p.N) continue;
            const int64_t orow = cmap(grow);
            float v = epilogue(p, lds[row * LDC + c], grow, orow, gcol, rowst, row, BM);
            if (p.ddpm_clip) v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);          // clip_denoised (NaN passes through, as torch.clamp)
            if (p.C && p.ln_out) __hip_atomic_store(p.C + orow * p.ldc + gcol, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // write-through (sc1), scalar form
            else if (p.C) p.C[orow * p.ldc + gcol] = v;
            if (p.ddpm_out) {
                const int b = grow / p.rows_per_sample;
                const int64_t ix = orow * p.ldx + gcol;
                const float xn = (p.ddpm_c1[b] * v + p.ddpm_c2[b] * p.ddpm_xt[ix]) + p.ddpm_sigma[b] * p.ddpm_noise[ix];
                p.ddpm_out[ix] = xn;
                if (p.ddpm_out2) p.ddpm_out2[orow * p.ldx2 + gcol] = xn;          // the next step's K-padded copy of x_t
            }
        }
    }
}

// Fused LayerNorm of the output rows ("last arriver", afm_linear_args.ln_*): called by EVERY thread of the workgroup after its tile's
// epilogue.  The tile is published with the placement-independent hand-off of cdna_hip_programming.md section 6 Guideline 16 (R1 with a
// ticket counter): the tile was stored write-through (sc1), every storing wave drains its stores, ONE lane takes a ticket on the row
// block's counter (relaxed, agent scope); the workgroup that draws the last ticket does ONE agent-scope acquire and then normalises the
// block's rows with plain loads - one wave per row, the arithmetic of layernorm_kernel.  `flag` is a word of the kernel's (now idle)
// LDS staging area.
template <int BM>
__device__ __forceinline__ void gemm_ln_tail(const afm_linear_args& p, int bm, int nbn, int* flag) {
    if (!p.ln_out) return;                                           // uniform
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave: its write-through stores have landed
    __syncthreads();                                                 // (also: every read of the staging area is done)
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(p.ln_counters + bm, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == (unsigned)(nbn - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(p.ln_counters + bm, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        }
        *flag = last;
    }
    __syncthreads();
    if (!*flag) return;
    const RowMap cmap{p.c_grp, p.c_stride, p.c_off, p.c_skip_after, p.c_skip};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = wave; r < BM; r += nw) {
        const int grow = bm * BM + r;
        if (grow >= p.M) break;
        const int64_t orow = cmap(grow);
        const float* xp = p.C + orow * p.ldc;
        float* yp = p.ln_out + orow * p.ldo;
        if (p.N <= 256) layernorm_row<1>(xp, p.ln_gamma, p.ln_beta, yp, p.N, p.ln_eps, lane);
        else if (p.N <= 512) layernorm_row<2>(xp, p.ln_gamma, p.ln_beta, yp, p.N, p.ln_eps, lane);
        else layernorm_row<4>(xp, p.ln_gamma, p.ln_beta, yp, p.N, p.ln_eps, lane);
    }
}

}  // namespace
