// afm_linear: C = act(scale * (A @ W^T) + bias) + residual + rowtab, optional fused DDPM update.
//
// gfx950 design
//   * v_mfma_f32_32x32x2_f32 (exact f32 products + f32 accumulate; 157 TF peak = 1/16 of bf16).
//     The 64-cycle MFMA is so long that LDS / L2 bandwidth is never the limiter; what matters is
//     keeping every SIMD issuing MFMAs back to back and filling 256 CUs (tile choice below).
//   * 256 threads = 4 waves in a 2x2 grid; each wave owns a (BM/2)x(BN/2) block of 32x32 MFMA tiles.
//   * K is consumed in tiles of 32.  Inside a tile the K index is PERMUTED: lane-half h of the wave
//     multiplies k = 16*h + s in MFMA step s (both operands use the same map, the sum over k is
//     order-free), so a lane's 16 operands of a row are contiguous -> 4 x ds_read_b128 per row.
//   * LDS rows are padded to 36 floats (144 B): conflict-free for ds_read_b128's 16-lane groups.
//   * global -> registers (next tile) is issued before the MFMA block of the current tile and
//     written to the other LDS buffer afterwards: one barrier per K tile.
//   * XCD-aware tile order: the 8 XCDs each walk a contiguous range of tiles (A row-panels are
//     reused out of the XCD's own L2; W (<= 3 MB) stays L2 resident).
#include <type_traits>
#include "common.h"
#include "profile.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = 36;     // padded LDS row (floats)

// One thread stages ROWS/32 rows x 4 consecutive k of a [ROWS x 32] tile.  Row base pointers are computed ONCE
// (row remap / clamping out of the K loop); rows past the matrix edge are clamped to the last valid row (their
// products are never stored).  Loads are UNCONDITIONAL (k clamped into the row) so the compiler emits no
// zero-init + branch + `s_waitcnt vmcnt(0)` in front of them; the K tail is zero-filled when the tile is
// written to LDS (store_tile), i.e. after the wait that the data needs anyway.
template <int ROWS, bool VEC>
__device__ __forceinline__ void load_tile(const float* const (&rowp)[ROWS / 32], int k, int K, float4 (&reg)[ROWS / 32]) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        if (VEC) {
            reg[i] = *reinterpret_cast<const float4*>(rowp[i] + min(k, K - 4));
        } else {
            const float* p = rowp[i];
            reg[i] = make_float4(p[min(k, K - 1)], p[min(k + 1, K - 1)], p[min(k + 2, K - 1)], p[min(k + 3, K - 1)]);
        }
    }
}

template <int ROWS>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, int tid, const float4 (&reg)[ROWS / 32], int k, int K) {
    const int c4 = tid & 7, r0 = tid >> 3;
    const bool k0 = k < K, k1 = k + 1 < K, k2 = k + 2 < K, k3 = k + 3 < K;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        float4 v = reg[i];
        v.x = k0 ? v.x : 0.f; v.y = k1 ? v.y : 0.f; v.z = k2 ? v.z : 0.f; v.w = k3 ? v.w : 0.f;
        *reinterpret_cast<float4*>(lds + (r0 + 32 * i) * LDK + c4 * 4) = v;
    }
}

template <int BM, int BN, bool VEC>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma(const afm_linear_args p, int nbm, int nbn) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int STAGE = (BM + BN) * LDK;          // floats per LDS stage: A tile then W tile
    constexpr int LDC = BN + 4;                     // padded row of the epilogue staging tile
    constexpr int LDS_FLOATS = (2 * STAGE > BM * LDC) ? 2 * STAGE : BM * LDC;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    // XCD-aware bijective remap: block b runs on XCD b % 8; give each XCD a contiguous tile range.
    const int nblk = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / nbn, bn = bid % nbn;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;
    const int c4 = tid & 7, r0 = tid >> 3;

    const RowMap amap{p.a_grp, p.a_stride, p.a_off, p.a_skip_after, p.a_skip};
    const float* arow[BM / 32];
    const float* wrow[BN / 32];
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) arow[i] = p.A + amap(min(bm * BM + r0 + 32 * i, p.M - 1)) * p.lda;
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) wrow[i] = p.W + (int64_t)min(bn * BN + r0 + 32 * i, p.N - 1) * p.ldw;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Register ring of depth 2: the loads of K-tile kt+2 are issued before the MFMAs of tile kt, so a load has
    // TWO tiles of MFMA time (>= 2 x 1024 cycles per resident wave) to land before its s_waitcnt (one tile was not
    // enough under load: SQ_WAIT_ANY 23 %).  The loop is unrolled by two so both register sets keep static names.
    float4 ra0[BM / 32], rw0[BN / 32], ra1[BM / 32], rw1[BN / 32];
    const int nk = (p.K + BK - 1) / BK;
    const int kc = c4 * 4;
    load_tile<BM, VEC>(arow, kc, p.K, ra0);
    load_tile<BN, VEC>(wrow, kc, p.K, rw0);
    if (nk > 1) {
        load_tile<BM, VEC>(arow, BK + kc, p.K, ra1);
        load_tile<BN, VEC>(wrow, BK + kc, p.K, rw1);
    }
    store_tile<BM>(lds, tid, ra0, kc, p.K);
    store_tile<BN>(lds + BM * LDK, tid, rw0, kc, p.K);
    __syncthreads();

    auto compute = [&](int cur) {
        const float* a_base = lds + cur * STAGE + (wm * (BM / 2) + r32) * LDK + hh * 16;
        const float* w_base = lds + cur * STAGE + BM * LDK + (wn * (BN / 2) + r32) * LDK + hh * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float af[TM][4], bf[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(a_base + i * 32 * LDK + j * 4);
                af[i][0] = t.x; af[i][1] = t.y; af[i][2] = t.z; af[i][3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const float4 t = *reinterpret_cast<const float4*>(w_base + i * 32 * LDK + j * 4);
                bf[i][0] = t.x; bf[i][1] = t.y; bf[i][2] = t.z; bf[i][3] = t.w;
            }
            // independent accumulators back to back (never two dependent MFMAs in a row when TM*TN > 1)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(af[tm][e], bf[tn][e], acc[tm][tn]);
        }
    };

    for (int kt = 0; kt < nk; kt += 2) {
        // ---- even tile kt: reads stage 0; set 0 is free (stored), refill it with tile kt+2; then store set 1 (tile kt+1)
        if (kt + 2 < nk) {
            load_tile<BM, VEC>(arow, (kt + 2) * BK + kc, p.K, ra0);
            load_tile<BN, VEC>(wrow, (kt + 2) * BK + kc, p.K, rw0);
        }
        compute(0);
        if (kt + 1 < nk) {
            store_tile<BM>(lds + STAGE, tid, ra1, (kt + 1) * BK + kc, p.K);
            store_tile<BN>(lds + STAGE + BM * LDK, tid, rw1, (kt + 1) * BK + kc, p.K);
        }
        __syncthreads();
        if (kt + 1 >= nk) break;
        // ---- odd tile kt+1: reads stage 1; refill set 1 with tile kt+3; then store set 0 (tile kt+2)
        if (kt + 3 < nk) {
            load_tile<BM, VEC>(arow, (kt + 3) * BK + kc, p.K, ra1);
            load_tile<BN, VEC>(wrow, (kt + 3) * BK + kc, p.K, rw1);
        }
        compute(1);
        if (kt + 2 < nk) {
            store_tile<BM>(lds, tid, ra0, (kt + 2) * BK + kc, p.K);
            store_tile<BN>(lds + BM * LDK, tid, rw0, (kt + 2) * BK + kc, p.K);
        }
        __syncthreads();
    }

    // ---- epilogue.  Stage the accumulators through LDS (lane holds column l&31, rows
    // (r&3) + 8*(r>>2) + 4*(l>>5) of each 32x32 tile -> conflict-free column-contiguous writes), then
    // stream rows out with 16-byte accesses: bias / residual / row-table reads and the C store are
    // fully coalesced and the epilogue math exists once instead of 16*TM*TN times.
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                lds[(wm * (BM / 2) + tm * 32 + mfma_row(r, lane)) * LDC + wn * (BN / 2) + tn * 32 + r32] = acc[tm][tn][r];
    __syncthreads();

    gemm_epilogue<BM, BN>(p, lds, bm, bn, tid);
    gemm_ln_tail<BM>(p, bm, nbn, reinterpret_cast<int*>(lds));
}

// ------------------------------------------------------------------------------------------------------------
// DMA variant (K % 32 == 0, 16-byte aligned operands): operands go global -> LDS directly (global_load_lds_dwordx4),
// no staging VGPRs, no ds_write, no K-tail select; a 3-stage LDS ring gives a true prefetch distance of two K-tiles
// with hand-counted `s_waitcnt vmcnt(N)` and ONE raw s_barrier per K-tile (the compiler's own waits cannot express
// "all but the newest tile").  Measured on the ingredient microbenchmark (tools/mfma_ingredients.hip): register-staged
// loads cost 150 -> 125-130 TF and the K-tail select another 8 %.
// LDS image: [row][8 chunks of 16 B] UNPADDED (the DMA destination is wave-uniform base + lane*16, so the image must be
// lane-linear); bank conflicts are removed by an XOR swizzle applied on the SOURCE address: LDS slot (row, c) holds
// logical chunk c ^ ((row >> 1) & 7), and a reader of logical chunk q reads slot q ^ ((row >> 1) & 7) (same involution both sides).
// Two 128-byte rows share a 256-byte bank row and ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}
// (+32): (row & 1, (row >> 1) & 7) is distinct for the 16 rows of each group, row & 7 alone was a 2-way conflict on every read
// (SQ_LDS_BANK_CONFLICT 3.4 M cycles per launch).
// Stage layout: every wave owns one contiguous region of PT = PA + PW pieces of 1 KiB (8 rows x 128 B): [A pass 0 .. | W pass 0 ..].
// The instruction's immediate offset is added to BOTH the global and the LDS address, so the pieces of one group of four share ONE
// LDS base: M0 is written once per group instead of once per instruction (the global pointer is pre-decremented by the same offset).
// With one M0 write per piece the issue phase measured ~290 cycles per DMA instruction (tools/gemm_timeline, 20.9 k of a workgroup's
// 69.8 k cycles at M = 10432, N = K = 512: an M0 write waits for the LDS-DMA instructions in flight that still need the old value).
template <int OFF>
__device__ __forceinline__ void glds16(const float* g, float* l) {
    static_assert(OFF >= 0 && OFF < 4096, "immediate offset range of global_load_lds");
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g) - OFF),
                                     (__attribute__((address_space(3))) void*)l, 16, OFF, 0);
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// Workgroup = WM x WN waves, each wave owns TM x TN MFMA tiles of 32x32: tile BM x BN = 32 WM TM x 32 WN TN.  Every shape sums an
// output element in the same order (K-tiles of 32 in sequence, the permuted k inside a tile), so shapes are interchangeable bit for
// bit and the dispatcher may pick one from M (small batches: 32x32 single-wave workgroups keep 256 CUs busy at B*T = 1304 rows).
#ifdef AFM_TIMELINE          // tools/gemm_timeline.hip only: per-workgroup (start, end, hardware id) records; never compiled into the library
struct AfmTimelineRec { unsigned long long t0, t1; unsigned hw_id, xcc_id; unsigned long long c0, c1, wait_c, barrier_c, issue_c, mfma_c; };
__device__ __forceinline__ unsigned long long afm_cycles() {      // shader clock; drained immediately so hand-counted lgkmcnt waits stay valid
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ AfmTimelineRec* afm_timeline = nullptr;
#endif

template <int WM, int WN, int TM, int TN, int NSTG = 3>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_mfma_dma(const afm_linear_args p, int nbm, int nbn) {
    constexpr int NT = 64 * WM * WN, BM = 32 * WM * TM, BN = 32 * WN * TN;
#ifdef AFM_TIMELINE
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime(), tl_c0 = afm_cycles();
    unsigned long long tl_wait = 0, tl_barrier = 0, tl_issue = 0, tl_mfma = 0;
#endif
    constexpr int RP = NT / 8;                       // rows one DMA pass of the whole workgroup covers (8 x 16 B per row)
    constexpr int PA = BM / RP, PW = BN / RP;        // passes per K-tile for the A / W panels
    static_assert(BM % RP == 0 && BN % RP == 0, "panel rows must be a multiple of the rows per DMA pass");
    constexpr int STAGE = (BM + BN) * BK;            // floats per stage: A tile then W tile, 32 floats per row
    constexpr int LDC = BN + 4;
    constexpr int LDS_FLOATS = (NSTG * STAGE > BM * LDC) ? NSTG * STAGE : BM * LDC;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const int nblk = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r32 = lane & 31, hh = lane >> 5;
    const int c8 = tid & 7, r0 = tid >> 3;           // this thread's (row-in-pass, 16-byte slot) of every DMA pass

    const RowMap amap{p.a_grp, p.a_stride, p.a_off, p.a_skip_after, p.a_skip};
    // per-pass source pointers with the swizzle folded in: slot c8 of row r receives logical chunk c8 ^ ((r >> 1) & 7)
    const float* asrc[PA];
    const float* wsrc[PW];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int r = r0 + RP * i;
        asrc[i] = p.A + amap(min(bm * BM + r, p.M - 1)) * p.lda + ((c8 ^ ((r >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int r = r0 + RP * i;
        wsrc[i] = p.W + (int64_t)min(bn * BN + r, p.N - 1) * p.ldw + ((c8 ^ ((r >> 1) & 7)) << 2);
    }
    constexpr int PT = PA + PW;                      // 1-KiB pieces per wave per K-tile = DMA instructions per thread per K-tile
    constexpr int PER_TILE = PT;
    const int wave_off = __builtin_amdgcn_readfirstlane(wave) * (PT * 256);      // floats: this wave's region of a stage
    auto issue = [&](int kt, int stage) {
        float* dst = lds + stage * STAGE + wave_off;
        static_for<0, PT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const float* src = i < PA ? asrc[i < PA ? i : 0] : wsrc[i < PA ? 0 : i - PA];
            glds16<(i & 3) * 1024>(src + kt * BK, dst + (i >> 2) * 1024);            // group base every 4 KiB, piece = immediate offset
        });
    };
    // LDS float offset of logical row R of the A tile (piece0 = 0) / the W tile (piece0 = PA): the wave that staged it, the pass, the row of 8
    auto row_off = [&](int R, int piece0) { return ((R % RP) >> 3) * (PT * 256) + (piece0 + R / RP) * 256 + (R & 7) * BK; };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nk = p.K / BK;
    issue(0, 0);
    if (NSTG == 3 && nk > 1) issue(1, 1);
    const int swz = (r32 >> 1) & 7;
    unsigned a_row[TM], w_row[TN];                   // byte offsets (inside a stage) of the rows this lane feeds to the MFMAs
#pragma unroll
    for (int i = 0; i < TM; ++i) a_row[i] = (unsigned)(row_off((wm * TM + i) * 32 + r32, 0) * 4);
#pragma unroll
    for (int i = 0; i < TN; ++i) w_row[i] = (unsigned)(row_off((wn * TN + i) * 32 + r32, PA) * 4);
    int stage = 0;
    for (int kt = 0; kt < nk; ++kt) {
#ifdef AFM_TIMELINE
        const unsigned long long tl_a = afm_cycles();
#endif
        // this wave's part of tile kt has landed once at most one newer tile (kt+1) is still in flight
        if (NSTG == 3 && kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef AFM_TIMELINE
        const unsigned long long tl_b = afm_cycles();
#endif
        __builtin_amdgcn_s_barrier();                // every wave's part landed; every wave is done with tile kt-1
#ifdef AFM_TIMELINE
        const unsigned long long tl_c = afm_cycles();
#endif
        if (NSTG == 3) { if (kt + 2 < nk) issue(kt + 2, stage == 0 ? 2 : stage - 1); }      // (kt+2) % 3 == (stage + 2) % 3
        else if (kt + 1 < nk) issue(kt + 1, stage ^ 1);                                     // 2-stage ring: 32 KB, more workgroups per CU
#ifdef AFM_TIMELINE
        const unsigned long long tl_d = afm_cycles();
#endif
        // LDS operand reads are issued through inline asm: with an LDS-DMA in flight hipcc otherwise guards every
        // ds_read with `s_waitcnt vmcnt(0)` (it cannot prove the DMA target does not alias), which drains the ring.
        // All reads of the K-tile are issued up front (they return in order), then each MFMA group waits for exactly
        // the reads it needs (counted lgkmcnt) - guide section 5.7 forms (ii)/(iii), rule 18 (sched_barrier after a wait).
        const unsigned st_addr = (unsigned)(stage * STAGE * 4);
        f32x4 af[4][TM], bf[4][TN];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned slot = (unsigned)((((hh * 4 + j) ^ swz) << 2) * 4);
#pragma unroll
            for (int i = 0; i < TM; ++i)
                asm volatile("ds_read_b128 %0, %1" : "=v"(af[j][i]) : "v"(st_addr + a_row[i] + slot));
#pragma unroll
            for (int i = 0; i < TN; ++i)
                asm volatile("ds_read_b128 %0, %1" : "=v"(bf[j][i]) : "v"(st_addr + w_row[i] + slot));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // reads still allowed in flight after group j's operands arrived: (3 - j) * (TM + TN)
            if (j == 0) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * (TM + TN)) : "memory");
            if (j == 1) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (TM + TN)) : "memory");
            if (j == 2) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(1 * (TM + TN)) : "memory");
            if (j == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[j][i]));
#pragma unroll
            for (int i = 0; i < TN; ++i) asm volatile("" : "+v"(bf[j][i]));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma32(af[j][tm][e], bf[j][tn][e], acc[tm][tn]);
        }
        stage = NSTG == 3 ? (stage == 2 ? 0 : stage + 1) : (stage ^ 1);
#ifdef AFM_TIMELINE
        const unsigned long long tl_e = afm_cycles();
        tl_wait += tl_b - tl_a; tl_barrier += tl_c - tl_b; tl_issue += tl_d - tl_c; tl_mfma += tl_e - tl_d;
#endif
    }
    __syncthreads();                                  // all waves done reading the last stage before it is reused

#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                lds[(wm * TM * 32 + tm * 32 + mfma_row(r, lane)) * LDC + wn * TN * 32 + tn * 32 + r32] = acc[tm][tn][r];
    __syncthreads();
    gemm_epilogue<BM, BN, NT>(p, lds, bm, bn, tid);
    gemm_ln_tail<BM>(p, bm, nbn, reinterpret_cast<int*>(lds));
#ifdef AFM_TIMELINE
    if (afm_timeline && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        afm_timeline[blockIdx.x] = AfmTimelineRec{tl_t0, (unsigned long long)__builtin_amdgcn_s_memrealtime(), hw, xcc, tl_c0, afm_cycles(), tl_wait, tl_barrier, tl_issue, tl_mfma};
    }
#endif
}

template <int WM, int WN, int TM, int TN>
int launch_dma(const afm_linear_args& a, int tag, hipStream_t s) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    AfmProf prof(tag, 2.0 * a.M * a.N * a.K, s);
    hipLaunchKernelGGL((gemm_f32_mfma_dma<WM, WN, TM, TN>), dim3(nbm * nbn), dim3(64 * WM * WN), 0, s, a, nbm, nbn);
    AFM_CHECK_LAUNCH();
    return 0;
}

int launch_regs(const afm_linear_args& a, bool vec, hipStream_t s) {       // register-staged 64x64 kernel: any K, any alignment
    const int nbm = (a.M + 63) / 64, nbn = (a.N + 63) / 64;
    AfmProf prof(AFM_PROF_GEMM64, 2.0 * a.M * a.N * a.K, s);
    if (vec) hipLaunchKernelGGL((gemm_f32_mfma<64, 64, true>), dim3(nbm * nbn), dim3(256), 0, s, a, nbm, nbn);
    else hipLaunchKernelGGL((gemm_f32_mfma<64, 64, false>), dim3(nbm * nbn), dim3(256), 0, s, a, nbm, nbn);
    AFM_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int afm_linear_split_mode(const afm_linear_args& a);                  // gemm_split.hip
int afm_linear_split(const afm_linear_args& a, int mode, hipStream_t s);
int afm_linear_thin_mode(const afm_linear_args& a);                   // gemm_thin.hip
int afm_linear_thin(const afm_linear_args& a, int mode, hipStream_t s);

// argument validation shared by afm_linear and afm_linear_pair (a.M > 0); 0 = fine
static int validate_linear(const afm_linear_args& a) {
    if (!a.A || !a.W || a.M < 0 || a.N <= 0 || a.K <= 0) return AFM_E_BADARG;
    if (!a.C && !a.ddpm_out && !a.rowdot_out) return AFM_E_BADARG;
    if (a.rowdot_w || a.rowdot_out) {            // row-dot epilogue: the 16-byte-row form of the shared epilogue, plain forward inputs only
        const uintptr_t ptrs = (uintptr_t)a.C | (uintptr_t)a.residual | (uintptr_t)a.bias | (uintptr_t)a.scale | (uintptr_t)a.rowdot_w;
        if (!a.rowdot_w || !a.rowdot_out || a.rowdot_n <= 0 || a.rowdot_n > 8 || (a.N & 3) || (a.ldc & 3) || (a.ldr & 3) || (ptrs & 15) ||
            a.ddpm_out || a.rowtab || a.preact || a.dact_z || a.drop_p > 0.0f)
            return AFM_E_BADARG;
    }
    if (a.ddpm_clip && !a.ddpm_out) return AFM_E_BADARG;
    if (a.ddpm_out2 && (!a.ddpm_out || a.ldx2 < a.N)) return AFM_E_BADARG;
    if (a.aux_dst && (!a.aux_src || a.aux_rows <= 0 || a.aux_cols <= 0 || a.aux_idx_max <= 0 || a.aux_dst_ld < a.aux_cols ||
                      a.aux_rows > (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128))) return AFM_E_BADARG;      // (one rider row per workgroup of the coarsest tiling)
    if (a.ddpm_out && (!a.ddpm_xt || !a.ddpm_noise || !a.ddpm_c1 || !a.ddpm_c2 || !a.ddpm_sigma || a.rows_per_sample <= 0))
        return AFM_E_BADARG;
    if (a.rowtab && a.rowtab_period <= 0) return AFM_E_BADARG;
    if (a.a_skip < 0 || a.c_skip < 0 || a.a_skip_after < 0 || a.c_skip_after < 0 || (a.a_skip && !a.a_grp) || (a.c_skip && !a.c_grp)) return AFM_E_BADARG;
    if (a.ln_out) {                               // fused LayerNorm of the output rows (ABI v5)
        const uintptr_t lp = (uintptr_t)a.C | (uintptr_t)a.ln_out | (uintptr_t)a.ln_gamma | (uintptr_t)a.ln_beta;
        if (!a.C || !a.ln_gamma || !a.ln_beta || !a.ln_counters || (a.N & 3) || (a.ldc & 3) || (a.ldo & 3) || (lp & 15) || a.N > 1024 ||
            a.ln_out == a.C || a.ln_out == a.A || a.ln_out == a.residual || a.ddpm_out || a.rowdot_w)
            return AFM_E_BADARG;
    }
    const bool lnfold = a.stat_out || a.a_stat || a.res_stat;
    if (lnfold) {                                 // LayerNorm folded across kernel boundaries (ABI v5)
        if ((a.stat_out && (a.N % 64)) || (a.a_stat && (!a.a_fold_g || a.a_stat_groups <= 0)) || (a.res_stat && (!a.residual || !a.res_gamma || !a.res_beta || (a.N % 64))) ||
            !(a.ln_eps2 > 0.0f) || a.scale || a.rowtab || a.preact || a.dact_z || a.drop_p > 0.0f || a.act_post || a.rowdot_w || a.ln_out)
            return AFM_E_BADARG;
        if (a.stat_out && (!a.C || (a.N & 3) || (a.ldc & 3) || (a.ldr & 3) || a.ddpm_out)) return AFM_E_BADARG;
        // stat_out is written by the 16-byte-row branch of the shared epilogue only: a side input that is not 16-byte aligned would send
        // the launch down the scalar branch (which applies a_stat / res_stat but never writes stat_out) and the next consumer would
        // normalise with uninitialised statistics - refuse it instead of returning success
        if (a.stat_out && ((((uintptr_t)a.C | (uintptr_t)a.residual | (uintptr_t)a.bias) & 15) || (a.ldp & 3) || (a.ldz & 3))) return AFM_E_BADARG;
        // the per-column fold vectors are read 16 bytes at a time, the statistic records 8
        if ((((uintptr_t)a.a_fold_g | (uintptr_t)a.res_gamma | (uintptr_t)a.res_beta) & 15) ||
            (((uintptr_t)a.stat_out | (uintptr_t)a.a_stat | (uintptr_t)a.res_stat) & 7)) return AFM_E_BADARG;
    }
    if (a.arith != AFM_ARITH_DEFAULT && a.arith != AFM_ARITH_F32 && a.arith != AFM_ARITH_BF16X6 && a.arith != AFM_ARITH_BF16X9 &&
        a.arith != AFM_ARITH_BF16X1) return AFM_E_BADARG;
    if (a.arith_min_n < 0) return AFM_E_BADARG;
    return 0;
}

extern "C" int afm_linear(const afm_linear_args* args, void* stream) {
    if (!args) return AFM_E_BADARG;
    const afm_linear_args& a = *args;
    if (a.M == 0) return 0;                                   // empty batch: nothing to do (pointers may be null)
    if (const int rc = validate_linear(a)) return rc;
    const bool lnfold = a.stat_out || a.a_stat || a.res_stat;
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (a.K % 4 == 0) && (a.lda % 4 == 0) && (a.ldw % 4 == 0) &&
                     (((uintptr_t)a.A & 15) == 0) && (((uintptr_t)a.W & 15) == 0);
    if (const int mode = afm_linear_split_mode(a)) return afm_linear_split(a, mode, s);
    if (const int thin = afm_linear_thin_mode(a)) return afm_linear_thin(a, thin, s);       // one side <= 16 wide, flat rows: a stream, not a GEMM
    if (lnfold || a.aux_dst) return AFM_E_UNSUPPORTED;         // the native kernels carry neither the row statistics nor the riders
    // Native f32 MFMA.  Tile choice (measured on MI355X, profiles/r01_gemm_investigation.md): with the 64-cycle f32 MFMA neither LDS
    // nor L2 bandwidth limits; what limits is keeping every SIMD's matrix pipe busy across the barrier / load phases of its waves
    // and filling 256 CUs.  64x64 tiles (4 workgroups = 4 waves per SIMD, fine-grained tails) beat 64x128 and 128x128 on every
    // encoder shape at B = 32 (2.79 vs 2.95 vs 3.30 ms/step); with few rows (strong scaling: B = 4 per GPU -> M = 1304) the same
    // kernel runs with 2-wave 32x64 or single-wave 32x32 workgroups so that the launch still covers the chip.  All shapes are
    // bit-identical (same summation order per output element), which is why M may enter this choice.
    const bool dma = vec && (a.K % BK) == 0 && !(a.tune & AFM_TUNE_NO_DMA);
    if (!dma) return launch_regs(a, vec, s);
    int tile = (a.tune & AFM_TUNE_TILE_MASK) >> AFM_TUNE_TILE_SHIFT;
    if (tile == 0) {
        const int64_t t64 = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64), t3264 = (int64_t)((a.M + 31) / 32) * ((a.N + 63) / 64);
        tile = t64 >= 512 ? 3 : (t3264 >= 512 ? 2 : 1);
    }
    if (a.rowdot_w && tile == 1) tile = 2;          // a row-dot group is 64 columns of ONE tile
    switch (tile) {
        case 1: return launch_dma<1, 1, 1, 1>(a, AFM_PROF_GEMM32_DMA, s);
        case 2: return launch_dma<1, 2, 1, 1>(a, AFM_PROF_GEMM32x64_DMA, s);
        case 3: return launch_dma<2, 2, 1, 1>(a, AFM_PROF_GEMM64_DMA, s);
        case 4: return launch_dma<2, 2, 1, 2>(a, AFM_PROF_GEMM64x128_DMA, s);
        case 5: return launch_dma<2, 2, 2, 2>(a, AFM_PROF_GEMM128_DMA, s);
        default: return AFM_E_BADARG;
    }
}

// ---- afm_linear_pair (ABI v7): TWO independent linear launches as ONE grid of 128 x 128 tiles of the bf16-split kernel (gemm_split.hip).
// The sampling loop's two sub-batch streams each run out_proj (164 tiles at 16 samples) and linear1 (328): neither fills the 512 resident
// slots of the 128 x 128 tile program, one sub-batch's out_proj together with the other's linear1 does (492).  Every output element is
// computed by the tile program of afm_linear with the same operands in the same order: bit-identical to two afm_linear calls.
int afm_linear_pair_split(const afm_linear_args& a0, const afm_linear_args& a1, int mode, hipStream_t s);       // gemm_split.hip
extern "C" int afm_linear_pair(const afm_linear_args* args0, const afm_linear_args* args1, void* stream) {
    if (!args0 || !args1) return AFM_E_BADARG;
    if (args0->M == 0) return afm_linear(args1, stream);
    if (args1->M == 0) return afm_linear(args0, stream);
    if (const int rc = validate_linear(*args0)) return rc;
    if (const int rc = validate_linear(*args1)) return rc;
    const int m0 = afm_linear_split_mode(*args0), m1 = afm_linear_split_mode(*args1);
    if (m0 == 0 || m0 != m1 || args0->K <= 256 || args1->K <= 256) return AFM_E_UNSUPPORTED;          // the paired form exists for the bf16-split tile program, one arithmetic per launch
    return afm_linear_pair_split(*args0, *args1, m0, (hipStream_t)stream);
}
