// afm_linear with the row-dot epilogue (the CDM's linear1: GELU(z W1^T + b) . w2 per 64-column group, M = B N points, N = K = 256), nine-product
// bf16 arithmetic (gemm_split.hip's header), WEIGHT-STATIONARY form.
//
// The LDS-staged 64x64 kernel spends a K-tile on 2 vector loads, ~50 split VALU, 6 ds_write_b64, a barrier and 6 ds_read_b128 per wave for 9 MFMAs
// (without a single MFMA it still takes 56 of its 59 us: profiles/r03_power_limit.md), and at K = 256 a tile is only 16 such K-tiles between a
// prologue and an epilogue.  Here a workgroup of 16 waves owns a SLAB of 64 output columns: W[64][256] is split once into the three bf16
// planes and stays in LDS for the whole launch (96 KB, operand-ready: one ds_read_b128 per plane, column half and K16 step).  Every wave then
// works ALONE on 32-row tiles taken from a workgroup counter: its A rows go global -> registers (each lane reads the 8 floats per K16 step it
// needs as MFMA operand), are split in registers and multiplied against both column halves of the slab - two independent accumulator chains,
// one split per 18 MFMAs.  No LDS stores and no barrier after the prologue; M / 32 tiles x N / 64 slabs are many rounds of work per wave.
// (A general-purpose form of this kernel - 32-column slabs, K = 512, folded-LayerNorm epilogue - lost to the staged kernels on the CMDM's
// shapes, where a launch is 1.3 - 2.5 tiles per wave: profiles/r03_gemm_slab.md.)
//
// Arithmetic: the products and their order are gemm_split.hip's (operand-to-lane mapping of v_mfma_f32_32x32x16_bf16, AFM_PA / AFM_PB); the
// row-dot partial of a 64-column group is summed in this kernel's own fixed tree (columns c and c + 32 in the lane, then c ^ 1, ^ 2, ^ 4, ^ 8,
// ^ 16 across lanes).  Which form runs is a function of the argument set only, never of M.
#include <type_traits>
#include "common.h"
#include "profile.h"
#include "gemm_epilogue.h"
#include "bf16split.h"

namespace {

constexpr int SLAB_N = 64;                            // output columns of a workgroup
constexpr int SLAB_K = 256;
constexpr int SLAB_STEPS = SLAB_K / 16;
constexpr int SLAB_WAVES = 16;
constexpr int SLAB_PLANE = SLAB_N * 32;               // bytes of one (K16 step, plane): [column][16 bf16], the 16-byte halves of a row swapped on odd (column >> 3)
constexpr int SLAB_BYTES = SLAB_STEPS * 3 * SLAB_PLANE;
constexpr int SLAB_SCR = 256;                         // per wave: output row offsets of the tile (32 x int64)

// Sum over the 32 lanes (l & 31) of a half wave for the 16 values a lane holds, tree l ^ 1, ^ 2, ^ 4, ^ 8, ^ 16.  Halving butterfly:
// 8 + 4 + 2 + 1 exchanges + one plain step; on return lane l holds the total of value index 8 (l & 1) + 4 ((l >> 1) & 1) + 2 ((l >> 2) & 1)
// + ((l >> 3) & 1), in both lanes l and l ^ 16.
__device__ __forceinline__ float group_reduce16(float (&v)[16], int lane) {
    halve_step<16, 1>(v, lane); halve_step<8, 2>(v, lane); halve_step<4, 4>(v, lane); halve_step<2, 8>(v, lane);
    return v[0] + lane_xor<16>(v[0]);
}
__device__ __forceinline__ int group_owned_row(int lane) { return 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1); }

template <int NPROD>
__global__ __launch_bounds__(64 * SLAB_WAVES) void gemm_f32_split_rowdot_slab(const afm_linear_args p, int nslab, int nchunk, int tiles_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sl_raw[];
    int* counter = reinterpret_cast<int*>(sl_raw + SLAB_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int64_t* t_orow = reinterpret_cast<int64_t*>(sl_raw + SLAB_BYTES + 16 + wave * SLAB_SCR);

    // workgroup -> (slab, row chunk).  The workgroups of one row chunk (one per slab) read the same A rows at about the same time: they share
    // an XCD (blockIdx & 7), i.e. an L2; nchunk % 8 == 0 gives every XCD the same number of chunks.
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int slab_i = jb % nslab, chunk = (jb / nslab) * 8 + xcd;
    const int n0 = slab_i * SLAB_N;

    // ---- prologue: W[n0 .. n0 + 63][0 .. 256) -> three bf16 planes in LDS.  Items are (column, 4 consecutive k); 16 consecutive lanes take the
    // four quarters of four consecutive columns of one K16 step: 64 contiguous bytes per column from memory, 128 contiguous bytes of LDS per
    // 16-lane store group (conflict-free ds_write_b64).
    {
        constexpr int ITEMS = SLAB_N * SLAB_K / 4 / (64 * SLAB_WAVES);
        f32x4 wv[ITEMS];
        int dst[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + 64 * SLAB_WAVES * i, q = it & 3, c_lo = (it >> 2) & 3, s = (it >> 4) & (SLAB_STEPS - 1), c = (it / (16 * SLAB_STEPS)) * 4 + c_lo;
            wv[i] = *reinterpret_cast<const f32x4*>(p.W + (int64_t)(n0 + c) * p.ldw + 16 * s + 4 * q);
            dst[i] = s * 3 * SLAB_PLANE + c * 32 + (((q >> 1) ^ ((c >> 3) & 1)) * 16) + (q & 1) * 8;
        }
        if (tid == 0) *counter = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            uint32_t a1, a2, a3, b1, b2, b3;
            split2(wv[i][0], wv[i][1], a1, a2, a3);
            split2(wv[i][2], wv[i][3], b1, b2, b3);
            unsigned char* d = sl_raw + dst[i];
            *reinterpret_cast<u32x2*>(d) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(d + SLAB_PLANE) = u32x2{a2, b2};
            *reinterpret_cast<u32x2*>(d + 2 * SLAB_PLANE) = u32x2{a3, b3};
        }
    }
    __syncthreads();

    const int r32 = lane & 31, hh = lane >> 5;
    const unsigned char* bbase = sl_raw + r32 * 32 + ((hh ^ ((r32 >> 3) & 1)) * 16);      // this lane's 16 bytes of (step 0, plane 0, column half 0)
    const RowMap amap{p.a_grp, p.a_stride, p.a_off, p.a_skip_after, p.a_skip}, cmap{p.c_grp, p.c_stride, p.c_off, p.c_skip_after, p.c_skip};
    const int ntile_all = (p.M + 31) / 32;
    const int tile0 = chunk * tiles_per_chunk;
    const int ntile = min(tiles_per_chunk, ntile_all - tile0);
    const int ngrp = (p.N + 63) / 64, R = p.rowdot_n;
    const float c_bias0 = p.bias ? p.bias[n0 + r32] : 0.f, c_bias1 = p.bias ? p.bias[n0 + 32 + r32] : 0.f;
    float w0[8], w1[8];                               // this lane's two columns of the R row-dot vectors
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        w0[j] = j < R ? p.rowdot_w[(int64_t)j * p.N + n0 + r32] : 0.f;
        w1[j] = j < R ? p.rowdot_w[(int64_t)j * p.N + n0 + 32 + r32] : 0.f;
    }

    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(counter, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= ntile) break;
        const int row0 = (tile0 + t) * 32;
        const float* aptr = p.A + amap(min(row0 + r32, p.M - 1)) * p.lda + 8 * hh;

        // A rows: a ring of four K16 steps per lane (8 floats each), three steps in flight ahead of the one being multiplied (the two lanes
        // of a row and two consecutive steps share one 128-byte line)
        f32x4 g[4][2];
        auto fetch = [&](auto SLOTC, int st) {
            constexpr int S = decltype(SLOTC)::value;
            const float* src = aptr + 16 * min(st, SLAB_STEPS - 1);
            g[S][0] = *reinterpret_cast<const f32x4*>(src);
            g[S][1] = *reinterpret_cast<const f32x4*>(src + 4);
        };
        using Slot0 = std::integral_constant<int, 0>;
        using Slot1 = std::integral_constant<int, 1>;
        using Slot2 = std::integral_constant<int, 2>;
        using Slot3 = std::integral_constant<int, 3>;
        fetch(Slot0{}, 0); fetch(Slot1{}, 1); fetch(Slot2{}, 2);
        if (hh == 0) t_orow[r32] = cmap(min(row0 + r32, p.M - 1));

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        auto step = [&](const f32x4& lo, const f32x4& hi, int s) {      // one K16 step: this lane's floats k = 16 s + 8 hh .. + 7
            u32x4 b[2][3];
            const unsigned char* bs = bbase + s * (3 * SLAB_PLANE);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                b[0][pl] = *reinterpret_cast<const u32x4*>(bs + pl * SLAB_PLANE);
                b[1][pl] = *reinterpret_cast<const u32x4*>(bs + pl * SLAB_PLANE + 32 * 32);
            }
            u32x4 a[3];
            uint32_t p1, p2, p3;
            split2(lo[0], lo[1], p1, p2, p3); a[0][0] = p1; a[1][0] = p2; a[2][0] = p3;
            split2(lo[2], lo[3], p1, p2, p3); a[0][1] = p1; a[1][1] = p2; a[2][1] = p3;
            split2(hi[0], hi[1], p1, p2, p3); a[0][2] = p1; a[1][2] = p2; a[2][2] = p3;
            split2(hi[2], hi[3], p1, p2, p3); a[0][3] = p1; a[1][3] = p2; a[2][3] = p3;
#pragma unroll
            for (int q = 9 - NPROD; q < 9; ++q) {
                acc0 = mfma_bf16(a[AFM_PA[q]], b[0][AFM_PB[q]], acc0);
                acc1 = mfma_bf16(a[AFM_PA[q]], b[1][AFM_PB[q]], acc1);
            }
        };
        for (int st = 0; st < SLAB_STEPS; st += 4) {
            fetch(Slot3{}, st + 3); step(g[0][0], g[0][1], st);
            fetch(Slot0{}, st + 4); step(g[1][0], g[1][1], st + 1);
            fetch(Slot1{}, st + 5); step(g[2][0], g[2][1], st + 2);
            fetch(Slot2{}, st + 6); step(g[3][0], g[3][1], st + 3);
        }

        // ---- epilogue from registers: lane (column r32 of each half, row half hh) holds rows (r & 3) + 8 (r >> 2) + 4 hh of the tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v0 = acc0[r], v1 = acc1[r];
            if (p.bias) { v0 += c_bias0; v1 += c_bias1; }
            if (p.act) { v0 = apply_act(v0, p.act); v1 = apply_act(v1, p.act); }
            if (p.C) {                                // uniform
                const int row = mfma_row(r, lane);
                if (row0 + row < p.M) {
                    float* c = p.C + t_orow[row] * p.ldc + n0 + r32;
                    c[0] = v0; c[32] = v1;
                }
            }
            acc0[r] = v0; acc1[r] = v1;
        }
        float res[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            res[j] = 0.f;
            if (j < R) {                              // uniform
                float pj[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) pj[r] = acc0[r] * w0[j] + acc1[r] * w1[j];
                res[j] = group_reduce16(pj, lane);
            }
        }
        const int row = mfma_row(group_owned_row(lane), lane);
        if (!(lane & 16) && row0 + row < p.M) {
            float* o = p.rowdot_out + (t_orow[row] * ngrp + slab_i) * R;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < R) o[j] = res[j];
        }
    }
}

}  // namespace

// Can this launch take the weight-stationary row-dot form?  A function of the argument set only (never of M).
bool afm_linear_rowdot_slab_ok(const afm_linear_args& a) {
    if (!a.rowdot_w || !a.rowdot_out || a.rowdot_n <= 0 || a.rowdot_n > 8 || a.K != SLAB_K || (a.N % SLAB_N) || a.N > 64 * SLAB_N) return false;
    if (a.residual || a.scale || a.preact || a.dact_z || a.drop_p > 0.0f || a.rowtab || a.act_post || a.ddpm_out || a.ln_out || a.stat_out || a.a_stat || a.res_stat) return false;
    if ((a.lda & 3) || (a.ldw & 3) || (((uintptr_t)a.A | (uintptr_t)a.W) & 15)) return false;
    return true;
}

template <int NPROD>
static int rowdot_slab_launch(const afm_linear_args& a, hipStream_t s) {
    constexpr int LDS_BYTES = SLAB_BYTES + 16 + SLAB_WAVES * SLAB_SCR;
    static const int attr = []() {
        return (int)hipFuncSetAttribute((const void*)gemm_f32_split_rowdot_slab<NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }();
    if (attr != 0) return attr;
    const int nslab = a.N / SLAB_N, ntile = (a.M + 31) / 32;
    // Row chunks: a multiple of 8 of them (see the kernel), as many as keep the launch within one resident round of 256 workgroups (one per CU:
    // 16 waves, ~100 KB of LDS); fewer for small M so that a workgroup still has a tile per wave
    int nchunk = (256 / nslab) & ~7;
    if (nchunk < 8) nchunk = 8;
    while (nchunk > 8 && ntile < nchunk * 16) nchunk -= 8;
    const int tpc = (ntile + nchunk - 1) / nchunk;
    AfmProf prof(AFM_PROF_GEMM_SLAB, 2.0 * a.M * a.N * a.K, s);
    hipLaunchKernelGGL(gemm_f32_split_rowdot_slab<NPROD>, dim3(nslab * nchunk), dim3(64 * SLAB_WAVES), LDS_BYTES, s, a, nslab, nchunk, tpc);
    AFM_CHECK_LAUNCH();
    return 0;
}

// products = 9 (all cross products) or 6 (gemm_split.hip's NPROD): the same products in the same order as the tile kernels of that arithmetic
int afm_linear_rowdot_slab(const afm_linear_args& a, int products, hipStream_t s) {
    return products == 6 ? rowdot_slab_launch<6>(a, s) : rowdot_slab_launch<9>(a, s);
}
