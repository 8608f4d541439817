// afm_linear on the bf16 matrix pipe with f32 results: every f32 operand is split EXACTLY into three bf16 terms
// (x = x1 + x2 + x3, 8 + 8 + 8 significant bits, round-to-nearest residual chain), the products x_i * w_j are exact in
// f32 inside v_mfma_f32_32x32x16_bf16 and accumulate in f32, so the sum of all nine products is the f32 dot product
// with f32 accumulation - the arithmetic of v_mfma_f32_32x32x2_f32 up to summation order.  gfx950's f32 MFMA runs at
// 1/16 of the bf16 rate (157 vs 2500 TF), so nine bf16 MFMAs cost 9/16 of one f32 MFMA's time.
//   NPROD = 9: all products (error = f32 accumulation order only).
//   NPROD = 6: drops x2*w3, x3*w2, x3*w3 (|x2| <= 2^-8 |x|, |x3| <= 2^-16 |x|: each dropped product <= 2^-24 |x||w|, rms ~2^-26, zero-mean because
//              the split rounds to nearest - an order of magnitude below the rounding noise of the f32 accumulation itself).
// Measured against float64 (tools/kernel_sweep.cpp, K = 512..4096): error / sum|x||w| rms 2.9e-8 for both variants vs 3.5e-8 for the
// native f32 MFMA kernel.  Default (AFM_ARITH_DEFAULT, since round 2): the exact 9-product variant on EVERY eligible GEMM (K >= 128,
// K % 16 == 0, 16-byte aligned operands, N >= 32) - in one call it beats the native kernel on every encoder shape (out_proj 65 vs 75 us,
// linear2 118 vs 134, in_proj 161 vs 186) and +7.5-8 % in the loop on every box seen (profiles/r02_gemm_investigation.md).  Narrow
// outputs (N < 32: the CDM's contact_layer products, N = contact_dim) stay on the native small-tile kernels: a 64 x 64 split tile would
// do nine times the work on a mostly empty tile.  afm_linear_args.arith = AFM_ARITH_F32 selects the native kernels everywhere;
// AFM_ARITH_BF16X6 (informational) is the fastest setting.  The choice never depends on M.
//
// Kernel: 256 threads = 2x2 waves, wave tile (BM/2)x(BN/2) of 32x32 MFMA tiles, K consumed BK at a time.
//   global f32 -> registers (next K-tile, issued before the MFMAs of the current one) -> split in VALU, interleaved with
//   the MFMA stream -> three bf16 planes in LDS ([row][BK] bf16, rows padded to 48 B: the 16 rows of each ds_read_b128 lane group
//   {0-3,12-15,20-27} / {4-11,16-19,28-31} hit 16 distinct 16-byte slots; the 8-byte stores of a 16-lane group cover four rows of
//   equal parity, also conflict-free) -> MFMA operands are one ds_read_b128 per (tile, plane, K16 step).
//   One barrier per K-tile (double-buffered LDS), shared epilogue of gemm.hip.
#include <type_traits>
#include "common.h"
#include "profile.h"
#include "gemm_epilogue.h"
#include "bf16split.h"

bool afm_linear_rowdot_slab_ok(const afm_linear_args& a);       // gemm_slab.hip
int afm_linear_rowdot_slab(const afm_linear_args& a, hipStream_t s);

namespace {

// K is summed in SEGMENTS of KSEG = 256: every segment accumulates from zero and the segment sums are added left to right,
// ((s0 + s1) + s2) + s3.  With KG == 1 a workgroup walks all segments itself (a second accumulator set, one add per segment: free);
// with KG > 1 (small launches: strong scaling runs 4 samples per GPU, M = 1304) KG groups of 256 threads take one segment each and
// the sums meet in LDS - the serial chain of a 32x32 MFMA tile drops from K to 256 deep.  Both forms add exactly the same numbers in
// the same order, so a shard computed by the split form is bit-identical to the full batch computed by the sequential form.
constexpr int KSEG = 256;

// Energy attribution builds (tools/gpu_power_ablate.sh only, never the library): -DAFM_ABLATE=<bits> removes one ingredient of the K loop at a time -
// wrong results, same control flow - so that time, clock and board power can be read per ingredient (profiles/r03_power_limit.md).
//   1 no global loads after the first two K-tiles, 2 no split arithmetic, 4 no LDS stores, 8 operand ds_reads only for the first K-tile, 16 no MFMAs
//   32 (round 4) the W operand's whole staging path removed after the first two K-tiles - no loads, no split, no LDS stores for the W items: the
//      upper bound of what weights handed over as ready-made bf16 planes (LDS-DMA, no VGPR / VALU work) could save
//   64 (round 4) the same for the A operand: the floor of a GEMM whose A rows are already split in LDS (out_proj fused behind the attention)
#ifndef AFM_ABLATE
#define AFM_ABLATE 0
#endif
#ifndef AFM_SPLITK_PIPELINED          // 1: launches of <= 256 tiles take the three-stage split-K forms (round 5); 0: round 4's two-stage forms everywhere
#define AFM_SPLITK_PIPELINED 1
#endif

// RING = 3 (round 5, small launches): THREE LDS stages and two sets of operand fragment registers.  With two stages a K-tile is a serial chain
// barrier -> 6 ds_read_b128 -> wait -> 9 dependent MFMAs -> last ds_write -> wait -> barrier (profiles/r04_gemm_timeline_small.txt: 0.48 us per
// K-tile against 0.14 us of MFMA issue when a CU holds one workgroup).  With three, tile kt + 2 is split into the third stage while the MFMAs of
// tile kt run on fragments that were read during kt - 1, and the fragments of tile kt + 1 (published by the previous barrier) are requested right
// behind the first MFMA: nothing but the barrier itself stands between two K-tiles' matrix work, and the split is finished two MFMAs before the
// barrier so that its LDS stores land under them.  Same MFMAs on the same accumulators in the same order: bit-identical.
// GSEG = 2 (split-K groups only): every group walks TWO consecutive K segments (K = 1024 on two groups instead of four: the three stages of four
// groups do not fit the LDS); group g > 0 hands its two segment sums over separately, so the sum order ((s0 + s1) + s2) + s3 is unchanged.
template <int BM, int BN, int BKS, int NPROD, int KG = 1, int RING = 2, int GSEG = 1>
__global__ __launch_bounds__(256 * KG, KG == 1 ? 2 : 1) void gemm_f32_split_bf16(const afm_linear_args p, int nbm, int nbn) {
    static_assert(BKS == 16, "one K16 MFMA step per K-tile");
    static_assert(RING == 2 || RING == 3, "two or three LDS stages");
    static_assert(GSEG == 1 || (GSEG == 2 && KG > 1), "two segments per group: split-K forms only");
    constexpr int TM = BM / 64, TN = BN / 64;
    static_assert(KG == 1 || (BM == 64 && BN == 64), "the split-K form exists for 64x64 tiles");
    constexpr int ROWB = BKS * 2 + 16;                // LDS row bytes (bf16 + pad)
    constexpr int ROWS = BM + BN;                     // A rows then W rows
    constexpr int PLANE = ROWS * ROWB;
    constexpr int STAGE = 3 * PLANE;
    constexpr int NA = BM / 64, NW = BN / 64;         // (row, 4 consecutive k) items of A / of W per thread and K-tile
    constexpr int NI = NA + NW;
    constexpr int LDC = BN + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int grp = KG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;      // K segment of this 256-thread group
    unsigned char* lds = lds_raw + grp * (RING * STAGE);
#if defined(AFM_TIMELINE) && defined(AFM_DESYNC_TICKS)
    // experiment (tools/gemm_timeline builds only): half of the workgroups of a CU start AFM_DESYNC_TICKS x 10 ns late, so that the prologue /
    // epilogue phases of one half meet the K loops of the other - what a desynchronised (persistent, stream-K like) schedule would buy
    if ((blockIdx.x >> 8) & 1) {
        const unsigned long long td = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - td < AFM_DESYNC_TICKS) __builtin_amdgcn_s_sleep(20);
    }
#endif
#ifdef AFM_TIMELINE          // tools/gemm_timeline.hip only (single translation unit with gemm.hip, which defines the record type)
    const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime(), tl_c0 = afm_cycles();
#endif

    const int nblk = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nblk >> 3, r = nblk & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int bm = bid / nbn, bn = bid % nbn;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // position inside the 256-thread group
    const int wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, hh = lane >> 5;

    // Staging items are (row, quarter) = 4 floats: the four lanes of a row fetch its 64 contiguous bytes of the K-tile with ONE
    // load instruction (16 cache lines per wave-instruction; an 8-float item needs two instructions of 32 lines each:
    // TCP_TOTAL_CACHE_ACCESSES 11.7 M -> 6.0 M per out_proj launch, 2-3 % faster), and every thread carries the same mix of A and W
    // work.  16 consecutive lanes (the unit ds_write_b64 is serviced in) take the four quarters of four rows of equal parity inside a
    // block of 8 rows: with rows 48 B apart those are four disjoint 32-byte windows of the 128-byte bank row.
    const RowMap amap{p.a_grp, p.a_stride, p.a_off, p.a_skip_after, p.a_skip};
    const float* src[NI];
    int dst[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int it = tid + 256 * (i < NA ? i : i - NA), q = it & 3, v = it >> 2, row = (v & ~7) + ((v & 3) << 1) + ((v >> 2) & 1);
        src[i] = (i < NA ? p.A + amap(min(bm * BM + row, p.M - 1)) * p.lda
                         : p.W + (int64_t)min(bn * BN + row, p.N - 1) * p.ldw) + q * 4 + grp * (GSEG * KSEG);
        dst[i] = ((i < NA ? 0 : BM) + row) * ROWB + q * 8;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Two register sets, always indexed with compile-time constants: at the top of K-tile kt the set (kt & 1) is free (its
    // tile went to LDS during kt - 1) and receives tile kt + 2; the other set holds tile kt + 1 (loaded one full K-tile
    // ago) and is split into the other LDS stage between the MFMAs of tile kt.
    // RING == 3 keeps NSET = 4 register sets: tile j lives in set j % 4 and is requested THREE K-tiles before its split (round 5: with one
    // K-tile of lead - ~950 cycles on a small launch - the split waited for operands that the previous kernel had just written on other
    // XCDs, an HBM / Infinity-Cache round trip of the same length; removing all split arithmetic moved a small launch by 0.3 us, removing one
    // operand's loads by 1.2: profiles/r04_gemm_w_ablation.txt).  A small launch holds two waves per SIMD: the registers are there.
    constexpr int NSET = RING == 3 ? 4 : 2;
    f32x4 g[NSET][NI];
    const int nk = KG > 1 ? GSEG * KSEG / BKS : p.K / BKS;
    constexpr int SEGT = KSEG / BKS;                  // K-tiles per segment
    f32x16 tot[TM][TN];                               // sum of the finished segments (KG == 1 with K > KSEG only)
    bool have_tot = false;
    auto load = [&](auto SETC, int kt) {
        constexpr int S = decltype(SETC)::value;
        const int k = min(kt, nk - 1) * BKS;          // past the end: re-load the last tile (never consumed)
        if ((AFM_ABLATE & 1) && kt >= 2) {
#pragma unroll
            for (int i = 0; i < NI; ++i) asm volatile("" : "+v"(g[S][i]));
            return;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if ((AFM_ABLATE & 32) && i >= NA && kt >= 2) { asm volatile("" : "+v"(g[S][i])); continue; }
            if ((AFM_ABLATE & 64) && i < NA && kt >= 2) { asm volatile("" : "+v"(g[S][i])); continue; }
            g[S][i] = *reinterpret_cast<const f32x4*>(src[i] + k);
        }
    };
    using Set0 = std::integral_constant<int, 0>;
    using Set1 = std::integral_constant<int, 1>;
    load(Set0{}, 0);
    load(Set1{}, 1);
    float* rowst = reinterpret_cast<float*>(lds_raw + KG * RING * STAGE);   // folded LayerNorm: row statistics, an LDS region of their own
    gemm_rowstats<BM>(p, rowst, bm);                  // (loads in flight with the operands'; published by the barrier below)
    auto split_set = [&](auto SETC, int stage) {      // prologue only: a whole register set -> one LDS stage, no interleaving
        constexpr int S = decltype(SETC)::value;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            unsigned char* d = lds + stage * STAGE + dst[i];
            uint32_t a1, a2, a3, b1, b2, b3;
            split2(g[S][i][0], g[S][i][1], a1, a2, a3);
            split2(g[S][i][2], g[S][i][3], b1, b2, b3);
            *reinterpret_cast<u32x2*>(d) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(d + PLANE) = u32x2{a2, b2};
            *reinterpret_cast<u32x2*>(d + 2 * PLANE) = u32x2{a3, b3};
        }
    };
    using Set2 = std::integral_constant<int, 2>;
    using Set3 = std::integral_constant<int, 3>;
    if constexpr (RING == 3) {                        // tiles 0 .. 4 requested, tiles 0 and 1 in LDS (K-tile 0 splits set 2 and refills set 1 with tile 5)
        load(Set2{}, 2);
        load(Set3{}, 3);
        split_set(Set0{}, 0);
        load(Set0{}, 4);
        split_set(Set1{}, 1);
    } else {
        split_set(Set0{}, 0);
    }
    __syncthreads();
#ifdef AFM_TIMELINE
    const unsigned long long tl_pro = __builtin_amdgcn_s_memrealtime();          // prologue done: first K-tile split and published
#endif

    const int a_off = (wm * (BM / 2) + r32) * ROWB + hh * 16;
    const int w_off = (BM + wn * (BN / 2) + r32) * ROWB + hh * 16;
    constexpr int NMFMA = NPROD * TM * TN;                    // MFMAs per wave per K-tile
    constexpr int NPIECE = NI * 2 * 3;                        // split pieces per thread per K-tile (pair of floats x residual level)

    // One K-tile.  The instruction order is written out and pinned with sched_barrier fences (hipcc otherwise hoists all
    // MFMAs in front of the split and chains the nine MFMAs of one accumulator back to back): after every MFMA a piece of
    // the next tile's split (~5 VALU) issues in the shadow of the 32-cycle matrix op.
    uint4 af[TM][3], bf[TN][3];
    auto body = [&](auto CURC, int kt) {          // K-tile kt with kt & 1 == cur: register set cur is free, set cur ^ 1 holds tile kt + 1
        constexpr int cur = decltype(CURC)::value;
        load(CURC, kt + 2);
        const unsigned char* base = lds + cur * STAGE;
        unsigned char* wbase = lds + (cur ^ 1) * STAGE;
        float r0[NI * 2], r1[NI * 2];
        uint32_t sp[NI * 2][3];
        int piece = 0, m = 0;
        auto do_piece = [&](int t) {
            const int u = t / 3, lvl = t % 3, i = u / 2, c = u % 2;
            if ((AFM_ABLATE & 32) && i >= NA && kt >= 1) return;          // W items: nothing to split, nothing to store
            if ((AFM_ABLATE & 64) && i < NA && kt >= 1) return;           // A items likewise (the floor of "A already in LDS": out_proj fused behind attention)
            if (lvl == 0) {
                r0[u] = g[cur ^ 1][i][2 * c];
                r1[u] = g[cur ^ 1][i][2 * c + 1];
            }
            const uint32_t pk = (AFM_ABLATE & 2) ? __float_as_uint(lvl == 1 ? r1[u] : r0[u]) : cvt_pk_bf16(r0[u], r1[u]);
            sp[u][lvl] = pk;
            if (lvl < 2) {
                if (!(AFM_ABLATE & 2)) {
                    r0[u] = sub_bf16_lo(r0[u], pk);              // (one v_dot2c_f32_bf16 each: bf16split.h)
                    r1[u] = sub_bf16_hi(r1[u], pk);
                }
            } else if (c == 1) {
                unsigned char* d = wbase + dst[i];
                if (AFM_ABLATE & 4) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) asm volatile("" :: "v"(sp[2 * i][pl]), "v"(sp[2 * i + 1][pl]), "v"(d));
                } else {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2*>(d + pl * PLANE) = u32x2{sp[2 * i][pl], sp[2 * i + 1][pl]};
                }
            }
        };
        if (!(AFM_ABLATE & 8) || kt == 0)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i][pl] = *reinterpret_cast<const uint4*>(base + pl * PLANE + a_off + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j][pl] = *reinterpret_cast<const uint4*>(base + pl * PLANE + w_off + j * 32 * ROWB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 9 - NPROD; q < 9; ++q)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    if (AFM_ABLATE & 16) asm volatile("" : "+v"(acc[tm][tn][q]) : "v"(af[tm][AFM_PA[q]].x), "v"(af[tm][AFM_PA[q]].w), "v"(bf[tn][AFM_PB[q]].x), "v"(bf[tn][AFM_PB[q]].w));
                    else acc[tm][tn] = mfma_bf16(af[tm][AFM_PA[q]], bf[tn][AFM_PB[q]], acc[tm][tn]);
                    ++m;
#pragma unroll
                    for (int t = 0; t < NPIECE; ++t)
                        if (t >= piece && t < (m * NPIECE) / NMFMA) do_piece(t);
                    piece = (m * NPIECE) / NMFMA;
                    __builtin_amdgcn_sched_barrier(0);
                }
        if ((KG == 1 || GSEG > 1) && ((kt + 1) % SEGT) == 0 && kt + 1 < nk) {        // segment finished, more to come: bank it, restart from zero
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        tot[tm][tn][r] = have_tot ? tot[tm][tn][r] + acc[tm][tn][r] : acc[tm][tn][r];
                        acc[tm][tn][r] = 0.f;
                    }
            have_tot = true;
        }
        __syncthreads();
    };
    // RING == 3: K-tile kt runs its MFMAs on fragment set kt & 1 (read during kt - 1), requests tile kt + 1's fragments from stage
    // (kt + 1) % 3 (written during kt - 1, published by that K-tile's barrier) into the other set, and splits tile kt + 2 (register set
    // kt & 1, loaded during kt - 1) into stage (kt + 2) % 3, whose previous tile (kt - 1) was last read during kt - 2.
    u32x4 fa[2][TM][3], fb[2][TN][3];
    auto read_frag = [&](auto FC, int stage) {
        constexpr int F = decltype(FC)::value;
        const unsigned char* base = lds + stage * STAGE;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[F][i][pl] = *reinterpret_cast<const u32x4*>(base + pl * PLANE + a_off + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[F][j][pl] = *reinterpret_cast<const u32x4*>(base + pl * PLANE + w_off + j * 32 * ROWB);
        }
    };
#ifndef AFM_R3_TAIL
#define AFM_R3_TAIL 2         // MFMAs at the end of a K-tile that carry no split work: the last LDS stores land under them
#endif
    auto body3 = [&](auto SC, int kt, int st_next, int st_wr) {          // K-tile kt, S = kt % 4
        constexpr int S = decltype(SC)::value;
        constexpr int cur = S & 1;                    // fragment set of this K-tile
        constexpr int SS = (S + 2) & 3;               // register set that is split (tile kt + 2)
        using LoadSet = std::integral_constant<int, (S + 1) & 3>;       // ... and the one that is refilled (tile kt + 5)
        using FragNext = std::integral_constant<int, cur ^ 1>;
        unsigned char* wbase = lds + st_wr * STAGE;
        float r0[NI * 2], r1[NI * 2];
        uint32_t sp[NI * 2][3];
        int piece = 0, m = 0;
        auto do_piece = [&](int t) {
            const int u = t / 3, lvl = t % 3, i = u / 2, c = u % 2;
            if (lvl == 0) {
                r0[u] = g[SS][i][2 * c];
                r1[u] = g[SS][i][2 * c + 1];
            }
            const uint32_t pk = cvt_pk_bf16(r0[u], r1[u]);
            sp[u][lvl] = pk;
            if (lvl < 2) {
                r0[u] = sub_bf16_lo(r0[u], pk);
                r1[u] = sub_bf16_hi(r1[u], pk);
            } else if (c == 1) {
                unsigned char* d = wbase + dst[i];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2*>(d + pl * PLANE) = u32x2{sp[2 * i][pl], sp[2 * i + 1][pl]};
            }
        };
        constexpr int NFILL = NMFMA - 1 - AFM_R3_TAIL > 0 ? NMFMA - 1 - AFM_R3_TAIL : 1;      // MFMAs 2 .. NFILL + 1 carry the split
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 9 - NPROD; q < 9; ++q)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc[tm][tn] = mfma_bf16(fa[cur][tm][AFM_PA[q]], fb[cur][tn][AFM_PB[q]], acc[tm][tn]);
                    ++m;
                    if (m == 1) {               // behind the first MFMA: the requests of this K-tile (nothing in it waits for them)
                        __builtin_amdgcn_sched_barrier(0);
                        load(LoadSet{}, kt + 5);
                        read_frag(FragNext{}, st_next);
                    } else {
                        const int want = ((m - 1) * NPIECE + NFILL - 1) / NFILL;
#pragma unroll
                        for (int t = 0; t < NPIECE; ++t)
                            if (t >= piece && t < want) do_piece(t);
                        piece = want < NPIECE ? want : NPIECE;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
        for (int t = 0; t < NPIECE; ++t)              // (NPROD == 1: a single MFMA per K-tile has no slot for the split)
            if (t >= piece) do_piece(t);
        if ((KG == 1 || GSEG > 1) && ((kt + 1) % SEGT) == 0 && kt + 1 < nk) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        tot[tm][tn][r] = have_tot ? tot[tm][tn][r] + acc[tm][tn][r] : acc[tm][tn][r];
                        acc[tm][tn][r] = 0.f;
                    }
            have_tot = true;
        }
        __syncthreads();
    };
    if constexpr (RING == 2) {
        for (int kt = 0; kt < nk; kt += 2) {
            body(Set0{}, kt);
            if (kt + 1 < nk) body(Set1{}, kt + 1);
        }
    } else {
        read_frag(Set0{}, 0);
        int s1 = 1, s2 = 2;                           // (kt + 1) % 3, (kt + 2) % 3
        for (int kt = 0; kt < nk; kt += 4) {         // nk % 4 == 0 (the launcher checks K)
            body3(Set0{}, kt, s1, s2);
            s1 = s2; s2 = (s1 + 1) % 3;
            body3(Set1{}, kt + 1, s1, s2);
            s1 = s2; s2 = (s1 + 1) % 3;
            body3(Set2{}, kt + 2, s1, s2);
            s1 = s2; s2 = (s1 + 1) % 3;
            body3(Set3{}, kt + 3, s1, s2);
            s1 = s2; s2 = (s1 + 1) % 3;
        }
    }

#ifdef AFM_TIMELINE
    const unsigned long long tl_kloop = __builtin_amdgcn_s_memrealtime();        // K loop done
#endif
    if (have_tot && (KG == 1 || grp == 0)) {          // ((s0 + s1) + ...) + s_last (of a later split-K group: both segment sums travel, see below)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = tot[tm][tn][r] + acc[tm][tn][r];
    }
    if (KG > 1) {
        // groups 1 .. KG-1 hand their segment sums (GSEG of them, in segment order) to group 0 through their own (now idle) operand
        // regions: [segment][r / 4][thread] float4
        static_assert(RING * STAGE >= GSEG * 4 * 256 * 16, "the segment sums of a group fit its operand stages");
        f32x4* part = reinterpret_cast<f32x4*>(lds);
        if (grp > 0) {
            if (GSEG == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) part[q * 256 + tid] = f32x4{tot[0][0][4 * q], tot[0][0][4 * q + 1], tot[0][0][4 * q + 2], tot[0][0][4 * q + 3]};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) part[((GSEG - 1) * 4 + q) * 256 + tid] = f32x4{acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]};
        }
        __syncthreads();
        if (grp == 0) {
            for (int gI = 1; gI < KG; ++gI) {
                const f32x4* pg = reinterpret_cast<const f32x4*>(lds_raw + gI * (RING * STAGE));
#pragma unroll
                for (int sI = 0; sI < GSEG; ++sI)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = pg[(sI * 4 + q) * 256 + tid];
                        acc[0][0][4 * q] += v[0]; acc[0][0][4 * q + 1] += v[1]; acc[0][0][4 * q + 2] += v[2]; acc[0][0][4 * q + 3] += v[3];
                    }
            }
        }
    }
    float* ldsf = reinterpret_cast<float*>(lds_raw);
    if (grp == 0) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ldsf[(wm * (BM / 2) + tm * 32 + mfma_row(r, lane)) * LDC + wn * (BN / 2) + tn * 32 + r32] = acc[tm][tn][r];
    }
    __syncthreads();
#ifdef AFM_TIMELINE
    const unsigned long long tl_staged = __builtin_amdgcn_s_memrealtime();       // K groups merged, accumulators staged in LDS
#endif
    // The split-K form's epilogue runs on ALL its wave groups (512 / 1024 threads: two trips / one over the 64 x 64 tile instead of four -
    // a small launch has nothing else to hide the trips' memory round trips behind; profiles/r04_gemm_timeline_small.txt: epilogue 3.5 of
    // 14.8 us).  An output element's arithmetic does not depend on which thread handles it: bit-identical.
    constexpr int ENT = (KG == 2 || KG == 4) ? 256 * KG : 256;
    if (ENT > 256) gemm_epilogue<BM, BN, ENT>(p, ldsf, bm, bn, (int)threadIdx.x, rowst);
    else if (grp == 0) gemm_epilogue<BM, BN>(p, ldsf, bm, bn, tid, rowst);
    gemm_ln_tail<BM>(p, bm, nbn, reinterpret_cast<int*>(ldsf));
    if (p.aux_dst && (int)blockIdx.x < p.aux_rows) {          // rider: one row copy (+ add) per workgroup, behind its own tile
        const int r = blockIdx.x;
        int64_t ix = p.aux_idx ? p.aux_idx[r] : r;
        ix = ix < 0 ? 0 : (ix >= p.aux_idx_max ? p.aux_idx_max - 1 : ix);
        for (int c = threadIdx.x; c < p.aux_cols; c += blockDim.x)
            p.aux_dst[(int64_t)r * p.aux_dst_ld + c] = p.aux_src[ix * p.aux_cols + c] + (p.aux_add ? p.aux_add[c] : 0.0f);
    }
#ifdef AFM_TIMELINE
    if (afm_timeline && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the epilogue's stores have left the wave
        const unsigned long long tl_end = __builtin_amdgcn_s_memrealtime();
        // phase spans in s_memrealtime ticks (10 ns): prologue | K loop | merge + staging | epilogue
        afm_timeline[blockIdx.x] = AfmTimelineRec{tl_t0, tl_end, hw, xcc, tl_c0, afm_cycles(), tl_pro - tl_t0, tl_kloop - tl_pro, tl_staged - tl_kloop, tl_end - tl_staged};
    }
#endif
}

template <int BM, int BN, int BKS, int NPROD, int KG = 1, int RING = 2, int GSEG = 1>
int launch_split(const afm_linear_args& a, hipStream_t s) {
    constexpr int STAGE = 3 * (BM + BN) * (BKS * 2 + 16);
    static_assert(RING * STAGE >= BM * (BN + 4) * 4, "the staged accumulators fit group 0's operand stages");
    constexpr int LDS_BYTES = KG * RING * STAGE + 2 * BM * 2 * 4;          // operand stages (reused for the staged accumulators) + row statistics of the folded LayerNorm
    static_assert(LDS_BYTES <= 160 * 1024, "LDS of one CU");
    static const int attr = []() {
        return (int)hipFuncSetAttribute((const void*)gemm_f32_split_bf16<BM, BN, BKS, NPROD, KG, RING, GSEG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }();
    if (attr != 0) return attr;
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    AfmProf prof(BM == 128 ? AFM_PROF_GEMM_SPLIT128 : (KG > 1 ? AFM_PROF_GEMM_SPLIT64_KG : AFM_PROF_GEMM_SPLIT64), 2.0 * a.M * a.N * a.K, s);
    hipLaunchKernelGGL((gemm_f32_split_bf16<BM, BN, BKS, NPROD, KG, RING, GSEG>), dim3(nbm * nbn), dim3(256 * KG), LDS_BYTES, s, a, nbm, nbn);
    AFM_CHECK_LAUNCH();
    return 0;
}

template <int NPROD>
int dispatch_split(const afm_linear_args& a, hipStream_t s) {
    const int tile = (a.tune & AFM_TUNE_TILE_MASK) >> AFM_TUNE_TILE_SHIFT;      // 3 = 64x64, 5 = 128x128, 7 = 64x64 split-K, 8 = weight-stationary slabs, 0 = heuristic
    // Row-dot launches with K = 256 (the CDM's linear1): the weight-stationary form (gemm_slab.hip) whatever M is
    if (NPROD == 9 && (tile == 0 || tile == 8) && afm_linear_rowdot_slab_ok(a)) return afm_linear_rowdot_slab(a, s);
    if (tile == 8) return AFM_E_UNSUPPORTED;
    // Small launches (every 64x64 tile resident at once, at most two per CU): the launch is bound by the serial K chain of one MFMA
    // tile, so the K segments of a tile go to separate 256-thread groups of one workgroup (bit-identical, see the kernel's header).
    // Measured (profiles/r02_kernel_sweep_splitk.txt, us, sequential -> split): M = 1304: out_proj 20.0 -> 16.6, ffn2 (K = 1024, four groups)
    // 32.7 -> 24.9, in_proj 30.4 -> 27.7; M = 326: 18.9 -> 15.4, 30.8 -> 22.4; M = 2608: out_proj (328 tiles) 26.0 -> 23.9 but ffn2 45.0 -> 47.5
    // (four groups = 147 KB of LDS = one workgroup per CU, 328 tiles are 1.3 rounds) -> two groups up to 512 tiles, three / four up to 256.
    {
        const int64_t tiles64 = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
        const int nseg = a.K / KSEG;
        const bool splittable = (a.K % KSEG) == 0 && nseg >= 2 && nseg <= 4;
        // Round 5: the pipelined forms (RING = 3: three LDS stages, operand fragments requested one K-tile ahead).  Two wave groups with three
        // stages are 111 KB - one workgroup per CU - so they take the launches of at most one tile per CU (<= 256 tiles); K = 1024 runs on
        // two groups of two segments each there (four groups x three stages do not fit the LDS).  tile 10 forces them, 7 the two-stage forms.
        const bool pipelined = AFM_SPLITK_PIPELINED && (nseg == 2 || nseg == 4) && tiles64 <= 256;
        if (splittable && nseg != 3 && (tile == 10 || (tile == 0 && pipelined))) {
            if (nseg == 2) return launch_split<64, 64, 16, NPROD, 2, 3>(a, s);
            return launch_split<64, 64, 16, NPROD, 2, 3, 2>(a, s);
        }
        if (tile == 10) return AFM_E_UNSUPPORTED;
        if (splittable && nseg == 4 && tile == 11) return launch_split<64, 64, 16, NPROD, 2, 2, 2>(a, s);      // measurement: two groups x two segments on two stages
        if (tile == 11) return AFM_E_UNSUPPORTED;
        if (splittable && (tile == 7 || (tile == 0 && tiles64 <= (nseg == 2 ? 512 : 256)))) {
            if (nseg == 2) return launch_split<64, 64, 16, NPROD, 2>(a, s);
            if (nseg == 3) return launch_split<64, 64, 16, NPROD, 3>(a, s);
            return launch_split<64, 64, 16, NPROD, 4>(a, s);
        }
        if (tile == 7) return AFM_E_UNSUPPORTED;
    }
    if (tile == 9 && (a.K % 64) != 0) return AFM_E_UNSUPPORTED;              // (its K loop is unrolled four K-tiles deep)
    if (tile == 9) return launch_split<64, 64, 16, NPROD, 1, 3>(a, s);      // measurement: the sequential 64x64 form on three stages (55 KB: two workgroups per CU)
    // 128x128 amortises the split best (each thread splits 16 floats per 36 MFMAs of its wave) but holds 2 workgroups per CU = 512
    // resident tiles, so it only pays when its last resident round is nearly full; otherwise 64x64 tiles fill the chip better.
    // Measured (profiles/r02_kernel_sweep.txt, x9, us): N=1536 M=10432 (984 tiles, 96 % full) 161 vs 174 for 64x64; M=5216 (492, 96 %) 80 vs 86;
    // M=2608 (252, 49 %) 51 vs 47; N=1024 M=10432 (656, 64 %) 118 vs 113; N=512 (328, 64 %) 73 vs 65.  Both tile shapes add the
    // products of an output element in the same order (bit-identical), so M may enter the choice.
    const int64_t tiles128 = (int64_t)((a.M + 127) / 128) * ((a.N + 127) / 128);
    const int64_t resident = 512, rounds = (tiles128 + resident - 1) / resident;
    // K <= 256 (one K segment, 16 K-tiles: the CDM's linear1, M = 262144, N = K = 256): the 128x128 workgroup's prologue / epilogue at two waves
    // per SIMD weigh twice as much as at K = 512 and the 64x64 kernel (four waves per SIMD) wins, 0.367 vs 0.384 ms (configs[2]: 1323 vs 1284
    // steps/s in one call).  Round 3 also measured 128x64 / 64x128 tiles in the CMDM loop: 431 / 437 steps/s against 449 for this rule.
    const bool full_rounds = tiles128 * 10 >= rounds * resident * 9 && a.K > KSEG;      // >= 90 % of the resident slots used over all rounds
    if (tile == 3 || (tile != 5 && !full_rounds)) return launch_split<64, 64, 16, NPROD>(a, s);
    return launch_split<128, 128, 16, NPROD>(a, s);
}

}  // namespace

// Which GEMMs take the split path is a function of (arith, arith_min_n, N, K) and operand alignment only - never of M - so a batch
// and its shards run the same arithmetic (sharding / sub-batch invariance stays bit-exact).  Measured in the sampling loop (B = 32,
// two sub-batch streams): native everywhere 406 steps/s, x9 on the N >= 1024 GEMMs (in_proj, linear1) 435, x9 everywhere 447 (385 on
// a box with less power headroom, where the wide-only default still matched native), x6 everywhere 529.
int afm_linear_split_mode(const afm_linear_args& a) {
    int mode, min_n;
    switch (a.arith) {
        case AFM_ARITH_DEFAULT: mode = 9; min_n = 32; break;       // narrow outputs: native small-tile kernels (a function of N only, never of M)
        case AFM_ARITH_BF16X9: mode = 9; min_n = a.arith_min_n; break;
        case AFM_ARITH_BF16X6: mode = 6; min_n = a.arith_min_n; break;
        case AFM_ARITH_BF16X1: mode = 1; min_n = a.arith_min_n; break;
        default: return 0;
    }
    const bool ok = (a.K % 16 == 0) && a.K >= 128 && a.N >= min_n && (a.lda % 4 == 0) && (a.ldw % 4 == 0) && (((uintptr_t)a.A & 15) == 0) &&
                    (((uintptr_t)a.W & 15) == 0);
    return ok ? mode : 0;
}

int afm_linear_split(const afm_linear_args& a, int mode, hipStream_t s) {
    if (mode == 1) return dispatch_split<1>(a, s);        // informational: plain bf16 x bf16 (top terms only), NOT f32 arithmetic
    return mode == 9 ? dispatch_split<9>(a, s) : dispatch_split<6>(a, s);
}
