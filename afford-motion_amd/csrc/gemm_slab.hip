// afm_linear, nine-product bf16 arithmetic (gemm_split.hip's header), WEIGHT-STATIONARY form for the large launches of the sampling loop.
//
// Why a second form (profiles/r03_power_limit.md): in the LDS-staged 64x64 kernel a K-tile is 2 vector loads, ~50 split VALU, 6 ds_write_b64,
// a barrier and 6 ds_read_b128 per wave for 9 MFMAs - without a single MFMA that kernel still takes 56 of its 59 us.  Here a workgroup of 16
// waves owns a SLAB of 32 output columns: W[32][K] (K <= 512) is split once into the three bf16 planes and stays in LDS for the whole launch
// (96 KB, operand-ready: one ds_read_b128 per plane and K16 step).  Every wave then works ALONE on 32-row tiles of its workgroup's row range:
// its A rows go global -> registers (each lane reads the 8 floats of its row it needs as MFMA operand: no LDS, no barrier, no sharing with
// another wave), are split in registers and fed to the nine MFMAs against the slab.  Per wave and K16 step: 2 vector loads, 44 split VALU,
// 3 ds_read_b128, 9 MFMAs; no stores to LDS, no barrier after the prologue.  Waves take tiles from a workgroup counter, so the work unit is
// one 32x32 tile of one wave (a launch is one resident round of <= 256 workgroups with equal row ranges).
//
// Arithmetic: identical to gemm_split.hip element by element - same operand-to-lane mapping of v_mfma_f32_32x32x16_bf16, same product order
// inside a K16 step (AFM_PA / AFM_PB), same K segments of 256 accumulated from zero and added left to right - so a launch may take either
// form (the choice depends on M) without changing a bit; tests/test_gpu_ops.py checks it for every epilogue input this form supports.
// Epilogue: from registers (lane = one output column, 16 rows): folded-LayerNorm inputs (a_stat / res_stat / stat_out, gemm_epilogue.h),
// bias, activation, residual, row maps.  Everything else (row-dot, DDPM update, dropout, ...) stays on the staged kernels.
#include <type_traits>
#include "common.h"
#include "profile.h"
#include "gemm_epilogue.h"
#include "bf16split.h"

namespace {

constexpr int SLAB_N = 32;                            // output columns of a workgroup
constexpr int SLAB_WAVES = 16;
constexpr int SLAB_SCR = 768;                         // per-wave epilogue tables: 32 x (row offset int64, A-row (mean, rstd), residual-row (mean, rstd))

template <int N, int MASK>
__device__ __forceinline__ void unhalve_step(float* m, int lane) {      // inverse of halve_step<2 N, MASK>: N values per lane -> 2 N
    const bool up = (lane & MASK) != 0;
#pragma unroll
    for (int q = 0; q < N; ++q) {
        const float recv = lane_xor<MASK>(m[q]);
        const float lo = up ? recv : m[q], hi = up ? m[q] : recv;
        m[q] = lo; m[q + N] = hi;
    }
}

// Sum over the 32 columns of a row (= the 32 lanes l & 31 of a half wave) for the 16 rows a lane holds, tree c ^ 1, ^ 2, ^ 4, ^ 8, ^ 16
// (gemm_epilogue.h).  Halving butterfly: 8 + 4 + 2 + 1 exchanges + one plain step; on return v[0] of lane l is the total of row index
// 8 (l & 1) + 4 ((l >> 1) & 1) + 2 ((l >> 2) & 1) + ((l >> 3) & 1), in both lanes l and l ^ 16.
__device__ __forceinline__ float group_reduce16(float (&v)[16], int lane) {
    halve_step<16, 1>(v, lane); halve_step<8, 2>(v, lane); halve_step<4, 4>(v, lane); halve_step<2, 8>(v, lane);
    return v[0] + lane_xor<16>(v[0]);
}
__device__ __forceinline__ int group_owned_row(int lane) { return 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1); }

template <int NSEG>                                   // K = 256 NSEG
__global__ __launch_bounds__(64 * SLAB_WAVES) void gemm_f32_split_slab(const afm_linear_args p, int nslab, int nchunk, int tiles_per_chunk) {
    constexpr int K = 256 * NSEG, NSTEP = K / 16;
    constexpr int SLAB_BYTES = NSTEP * 3 * SLAB_N * 32;                // [K16 step][plane][column][16 bf16], the 16-byte halves of a row swapped on odd (column >> 3)
    extern __shared__ __attribute__((aligned(16))) unsigned char sl_raw[];
    int* counter = reinterpret_cast<int*>(sl_raw + SLAB_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* scr = sl_raw + SLAB_BYTES + 16 + wave * SLAB_SCR;

    // workgroup -> (slab, row chunk).  The workgroups of one row chunk read the same A rows: keep them on one XCD (blockIdx & 7), i.e. in one L2
    int slab_i, chunk;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3;          // nchunk % 8 == 0 (launch_slab)
        slab_i = j % nslab; chunk = (j / nslab) * 8 + xcd;
    }
    const int n0 = slab_i * SLAB_N;

    // ---- prologue: W[n0 .. n0 + 31][0 .. K) -> three bf16 planes in LDS.  Items are (column, 4 consecutive k); 16 consecutive lanes take the
    // four quarters of four consecutive columns of one K16 step: 64 contiguous bytes per column from memory, 128 contiguous bytes of LDS per
    // 16-lane store group (conflict-free ds_write_b64).
    {
        constexpr int ITEMS = SLAB_N * K / 4 / (64 * SLAB_WAVES);
        f32x4 wv[ITEMS];
        int dst[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const int it = tid + 64 * SLAB_WAVES * i, q = it & 3, c_lo = (it >> 2) & 3, s = (it >> 4) & (NSTEP - 1), c = (it / (16 * NSTEP)) * 4 + c_lo;
            wv[i] = *reinterpret_cast<const f32x4*>(p.W + (int64_t)(n0 + c) * p.ldw + 16 * s + 4 * q);
            dst[i] = (s * 3 * SLAB_N + c) * 32 + (((q >> 1) ^ ((c >> 3) & 1)) * 16) + (q & 1) * 8;
        }
        if (tid == 0) *counter = 0;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            uint32_t a1, a2, a3, b1, b2, b3;
            split2(wv[i][0], wv[i][1], a1, a2, a3);
            split2(wv[i][2], wv[i][3], b1, b2, b3);
            unsigned char* d = sl_raw + dst[i];
            *reinterpret_cast<u32x2*>(d) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(d + SLAB_N * 32) = u32x2{a2, b2};
            *reinterpret_cast<u32x2*>(d + 2 * SLAB_N * 32) = u32x2{a3, b3};
        }
    }
    __syncthreads();

    const int r32 = lane & 31, hh = lane >> 5;
    const unsigned char* bbase = sl_raw + r32 * 32 + ((hh ^ ((r32 >> 3) & 1)) * 16);      // this lane's 16 bytes of (step 0, plane 0)
    const RowMap amap{p.a_grp, p.a_stride, p.a_off}, cmap{p.c_grp, p.c_stride, p.c_off};
    const int ntile_all = (p.M + 31) / 32;
    const int tile0 = chunk * tiles_per_chunk;
    const int ntile = min(tiles_per_chunk, ntile_all - tile0);
    const int gcol = n0 + r32;
    const float c_bias = p.bias ? p.bias[gcol] : 0.f;
    const float c_g = p.a_stat ? p.a_fold_g[gcol] : 0.f;
    const float c_rg = p.res_stat ? p.res_gamma[gcol] : 0.f, c_rb = p.res_stat ? p.res_beta[gcol] : 0.f;
    int64_t* t_orow = reinterpret_cast<int64_t*>(scr);
    float2* t_sa = reinterpret_cast<float2*>(scr + 256);
    float2* t_sr = reinterpret_cast<float2*>(scr + 512);

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
            const float* src = aptr + 16 * min(st, NSTEP - 1);
            g[S][0] = *reinterpret_cast<const f32x4*>(src);
            g[S][1] = *reinterpret_cast<const f32x4*>(src + 4);
        };
        using Slot0 = std::integral_constant<int, 0>;
        using Slot1 = std::integral_constant<int, 1>;
        using Slot2 = std::integral_constant<int, 2>;
        using Slot3 = std::integral_constant<int, 3>;
        fetch(Slot0{}, 0); fetch(Slot1{}, 1); fetch(Slot2{}, 2);

        {   // per-wave tables of the tile's 32 rows (lane < 32: output row offset and A-row statistics; lane >= 32: residual-row statistics)
            const int grow = min(row0 + r32, p.M - 1);
            if (hh == 0) {
                t_orow[r32] = cmap(grow);
                if (p.a_stat) {
                    float mean, rstd;
                    row_stat_combine(p.a_stat + amap(grow) * (2 * p.a_stat_groups), p.a_stat_groups, p.ln_eps2, mean, rstd);
                    t_sa[r32] = make_float2(mean, rstd);
                }
            } else if (p.res_stat) {
                const int groups = p.N / AFM_STAT_GROUP;
                float mean, rstd;
                row_stat_combine(p.res_stat + cmap(grow) * (2 * groups), groups, p.ln_eps2, mean, rstd);
                t_sr[r32] = make_float2(mean, rstd);
            }
        }

        f32x16 acc, tot;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; tot[r] = 0.f; }
        auto step = [&](const f32x4& lo, const f32x4& hi, int s) {      // one K16 step: this lane's floats k = 16 s + 8 hh .. + 7
            u32x4 a[3], b[3];
            uint32_t p1, p2, p3;
            split2(lo[0], lo[1], p1, p2, p3); a[0][0] = p1; a[1][0] = p2; a[2][0] = p3;
            split2(lo[2], lo[3], p1, p2, p3); a[0][1] = p1; a[1][1] = p2; a[2][1] = p3;
            split2(hi[0], hi[1], p1, p2, p3); a[0][2] = p1; a[1][2] = p2; a[2][2] = p3;
            split2(hi[2], hi[3], p1, p2, p3); a[0][3] = p1; a[1][3] = p2; a[2][3] = p3;
            const unsigned char* bs = bbase + s * (3 * SLAB_N * 32);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4*>(bs + pl * (SLAB_N * 32));
#pragma unroll
            for (int q = 0; q < 9; ++q) acc = mfma_bf16(a[AFM_PA[q]], b[AFM_PB[q]], acc);
        };
#pragma unroll
        for (int seg = 0; seg < NSEG; ++seg) {
            for (int st = 16 * seg; st < 16 * seg + 16; st += 4) {
                fetch(Slot3{}, st + 3); step(g[0][0], g[0][1], st);
                fetch(Slot0{}, st + 4); step(g[1][0], g[1][1], st + 1);
                fetch(Slot1{}, st + 5); step(g[2][0], g[2][1], st + 2);
                fetch(Slot2{}, st + 6); step(g[3][0], g[3][1], st + 3);
            }
            if (seg + 1 < NSEG) {                     // segment finished, more to come: bank it, restart from zero
#pragma unroll
                for (int r = 0; r < 16; ++r) { tot[r] = seg == 0 ? acc[r] : tot[r] + acc[r]; acc[r] = 0.f; }
            }
        }
        if (NSEG > 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = tot[r] + acc[r];
        }

        // ---- epilogue from registers: lane (column r32, half hh) holds rows (r & 3) + 8 (r >> 2) + 4 hh of the tile
        float out[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma_row(r, lane);
            const bool valid = row0 + row < p.M;
            const int64_t orow = t_orow[row];
            float v = acc[r];
            if (p.a_stat) { const float2 st = t_sa[row]; v = ep_fold_a(v, st.x, st.y, c_g); }
            if (p.bias) v += c_bias;
            if (p.act) v = apply_act(v, p.act);
            if (p.residual) {
                float rv = valid ? p.residual[orow * p.ldr + gcol] : 0.f;
                if (p.res_stat) { const float2 st = t_sr[row]; rv = ep_norm_res(rv, st.x, st.y, c_rg, c_rb); }
                v += rv;
            }
            if (valid) p.C[orow * p.ldc + gcol] = v;
            out[r] = v;
        }
        if (p.stat_out) {                             // uniform.  (mean, M2) of the 32 stored columns of every row: gemm_epilogue.h's tree
            float m[16], sq[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) m[r] = out[r];
            const float mu = group_reduce16(m, lane) * (1.0f / STAT_GROUP_F);
            m[0] = mu;
            unhalve_step<1, 8>(m, lane); unhalve_step<2, 4>(m, lane); unhalve_step<4, 2>(m, lane); unhalve_step<8, 1>(m, lane);      // every lane: the 16 row means
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = out[r] - m[r]; sq[r] = ep_sq(d); }
            const float m2 = group_reduce16(sq, lane);
            const int row = mfma_row(group_owned_row(lane), lane);
            if (!(lane & 16) && row0 + row < p.M)
                *reinterpret_cast<float2*>(p.stat_out + (t_orow[row] * (p.N / AFM_STAT_GROUP) + slab_i) * 2) = make_float2(mu, m2);
        }
    }
}

template <int NSEG>
int launch_slab(const afm_linear_args& a, hipStream_t s) {
    constexpr int LDS_BYTES = (256 * NSEG / 16) * 3 * SLAB_N * 32 + 16 + SLAB_WAVES * SLAB_SCR;
    static const int attr = []() {
        return (int)hipFuncSetAttribute((const void*)gemm_f32_split_slab<NSEG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }();
    if (attr != 0) return attr;
    const int nslab = a.N / SLAB_N, ntile = (a.M + 31) / 32;
    // Row chunks: a multiple of 8 of them (a chunk's workgroups - one per slab, all reading the same A rows - share an XCD, i.e. an L2, and the
    // XCDs get equal shares), as many as keep the launch within one resident round of 256 workgroups (one per CU: 16 waves, ~110 KB of LDS)
    int nchunk = (256 / nslab) & ~7;
    if (nchunk < 8) nchunk = 8;
    while (nchunk > 8 && ntile < nchunk * 4) nchunk -= 8;
    const int tpc = (ntile + nchunk - 1) / nchunk;
    AfmProf prof(AFM_PROF_GEMM_SLAB, 2.0 * a.M * a.N * a.K, s);
    hipLaunchKernelGGL((gemm_f32_split_slab<NSEG>), dim3(nslab * nchunk), dim3(64 * SLAB_WAVES), LDS_BYTES, s, a, nslab, nchunk, tpc);
    AFM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// Can this launch take the weight-stationary form?  A function of the argument set only (never of M beyond "large enough to fill it"): the
// results are bit-identical to the staged kernels', so the caller may mix the two.
bool afm_linear_slab_ok(const afm_linear_args& a) {
    if (!(a.K == 256 || a.K == 512) || (a.N % SLAB_N) || a.N > 256 * SLAB_N) return false;
    if (!a.C || a.scale || a.preact || a.dact_z || a.drop_p > 0.0f || a.rowtab || a.act_post || a.ddpm_out || a.rowdot_w || a.ln_out) return false;
    if ((a.lda & 3) || (a.ldw & 3) || (((uintptr_t)a.A | (uintptr_t)a.W) & 15)) return false;
    return true;
}

int afm_linear_slab(const afm_linear_args& a, hipStream_t s) {
    return a.K == 256 ? launch_slab<1>(a, s) : launch_slab<2>(a, s);
}
