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
//   One barrier per K-tile (double-buffered LDS; the small-launch forms of round 5 run on three stages, RING = 3 below), shared epilogue of gemm.hip.
//   The tile program itself is csrc/gemm_split_body.inc, included as text by the kernels that run it.
#include <type_traits>
#include "common.h"
#include "profile.h"
#include "gemm_epilogue.h"
#include "bf16split.h"

bool afm_linear_rowdot_slab_ok(const afm_linear_args& a);       // gemm_slab.hip
int afm_linear_rowdot_slab(const afm_linear_args& a, int products, hipStream_t s);

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
#define AFM_WG ((int)blockIdx.x)
#define AFM_TIDX threadIdx.x
#define AFM_TIMELINE_SLOT blockIdx.x
#include "gemm_split_body.inc"
#undef AFM_WG
#undef AFM_TIDX
#undef AFM_TIMELINE_SLOT
}

// ---- WALK (round 5; MEASUREMENT ONLY, AFM_TUNE tile code 12): `gridDim.x` RESIDENT workgroups walk the tiles of a launch (tile wg = blockIdx.x,
// + gridDim.x, ...) instead of one workgroup per tile - the first half of the "layer-resident" schedule VERDICT r4 asked for: no lockstep
// rounds, no tail round.  Built, bit-identical (same tile program, same tile order), and SLOWER where it counts (profiles/r05_layer_resident.md):
// single launches 63.6 -> 69.1 us (linear1 at M = 5216), the two-stream B = 32 loop 478-479 -> 452-457 steps/s for every grid size and register
// budget tried (hipcc needs 168-202 VGPRs for the looped tile program against 93 for the plain one: 2-3 workgroups per CU instead of 4-5).
// The library never selects it.
template <int BM, int BN, int BKS, int NPROD>
#ifndef AFM_WALK_LB
#define AFM_WALK_LB 3
#endif
#ifndef AFM_WALK_OPAQUE
#define AFM_WALK_OPAQUE 1
#endif
__global__ __launch_bounds__(256, AFM_WALK_LB) void gemm_f32_split_bf16_walk(const afm_linear_args p, int nbm, int nbn) {
    constexpr int KG = 1, RING = 2, GSEG = 1;
    const int ntiles_ = nbm * nbn;
    for (int wg_ = blockIdx.x; wg_ < ntiles_; wg_ += gridDim.x) {
        if (wg_ != (int)blockIdx.x) __syncthreads();         // the previous tile's epilogue has left the LDS
        // an opaque copy of the thread index per tile: hipcc otherwise hoists every thread-dependent address of prologue AND epilogue out of
        // the tile loop (202 VGPRs, two waves per SIMD; capped at 128 it spilled 260 bytes per lane) - recomputed per tile they cost nothing
        int tidx_ = (int)threadIdx.x;
        if (AFM_WALK_OPAQUE) asm volatile("" : "+v"(tidx_));
#define AFM_WG wg_
#define AFM_TIDX tidx_
#define AFM_TIMELINE_SLOT wg_
#include "gemm_split_body.inc"
#undef AFM_WG
#undef AFM_TIDX
#undef AFM_TIMELINE_SLOT
    }
}

// ---- WIDE (round 6; MEASUREMENT, tile code 13): a 256 x 128 tile on 512 threads = 4 x 2 waves, each wave the 64 x 64 wave tile of the 128 x 128 program.
// Per matrix instruction the workgroup stages 25 % fewer operand rows (384 rows for 32 wave-tiles against 256 for 16: global loads, split VALU, LDS stores);
// the price is one workgroup per CU (139 KB of LDS: the staged accumulators outgrow the operand stages) whose eight waves meet at every K16 barrier.
// Same tile program text, same products in the same order: bit-identical to every other tile shape.
template <int NPROD>
__global__ __launch_bounds__(512, 1) void gemm_f32_split_bf16_wide(const afm_linear_args p, int nbm, int nbn) {
    constexpr int BM = 256, BN = 128, BKS = 16, KG = 1, RING = 2, GSEG = 1;
#define AFM_BODY_WM 4
#define AFM_WG ((int)blockIdx.x)
#define AFM_TIDX threadIdx.x
#define AFM_TIMELINE_SLOT blockIdx.x
#include "gemm_split_body.inc"
#undef AFM_WG
#undef AFM_TIDX
#undef AFM_TIMELINE_SLOT
#undef AFM_BODY_WM
}

template <int NPROD>
int launch_split_wide(const afm_linear_args& a, hipStream_t s) {
    constexpr int BM = 256, BN = 128, BKS = 16;
    constexpr int STAGE = 3 * (BM + BN) * (BKS * 2 + 16);
    constexpr int OPER = 2 * STAGE > BM * (BN + 4) * 4 ? 2 * STAGE : BM * (BN + 4) * 4;
    constexpr int LDS_BYTES = OPER + 2 * BM * 2 * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS of one CU");
    if (a.ln_out) return AFM_E_UNSUPPORTED;                 // (the fused LayerNorm's tickets are per 128-row block at most)
    if (a.aux_dst && a.aux_rows > (int64_t)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN)) return AFM_E_UNSUPPORTED;      // one rider row per workgroup
    static const int attr = []() {
        return (int)hipFuncSetAttribute((const void*)gemm_f32_split_bf16_wide<NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }();
    if (attr != 0) return attr;
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    AfmProf prof(AFM_PROF_GEMM_SPLIT128, 2.0 * a.M * a.N * a.K, s);
    hipLaunchKernelGGL((gemm_f32_split_bf16_wide<NPROD>), dim3(nbm * nbn), dim3(512), LDS_BYTES, s, a, nbm, nbn);
    AFM_CHECK_LAUNCH();
    return 0;
}

// ---- PAIR (round 6): two independent problems in ONE grid of 128 x 128 tiles.  Workgroups [0, tiles0) run problem 0's tiles, workgroups
// [start1, start1 + tiles1) problem 1's (start1 = tiles0 rounded up to a multiple of 8, so that a tile's XCD - blockIdx.x & 7 - is what the
// tile program's XCD-aware tile order assumes; the < 8 workgroups in between exit at once).  The tile program is the text every other kernel
// of this file runs; which problem a workgroup belongs to is wave-uniform, its arguments are read from the kernel-argument segment with scalar
// loads as before.  Why: the sampling loop's sub-batch launches of out_proj (164 tiles) and linear1 (328) do not fill the 512 resident slots
// of this tile program, one sub-batch's out_proj + the other's linear1 do (492) - afm_linear_pair, csrc/cmdm.hip's paired schedule.
struct afm_linear_pair_args {
    afm_linear_args a[2];
    int nbm[2], nbn[2];
    int tiles0, start1;
};
template <int BM, int BN, int BKS, int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_f32_split_bf16_pair(const afm_linear_pair_args pa) {
    constexpr int KG = 1, RING = 2, GSEG = 1;
    const int which = (int)blockIdx.x >= pa.start1 ? 1 : 0;
    const int wg_ = (int)blockIdx.x - (which ? pa.start1 : 0);
    if (!which && wg_ >= pa.tiles0) return;
    const afm_linear_args& p = pa.a[which];
    const int nbm = pa.nbm[which], nbn = pa.nbn[which];
#define AFM_WG wg_
#define AFM_TIDX threadIdx.x
#define AFM_TIMELINE_SLOT blockIdx.x
#include "gemm_split_body.inc"
#undef AFM_WG
#undef AFM_TIDX
#undef AFM_TIMELINE_SLOT
}

template <int NPROD>
int launch_split_pair(const afm_linear_args& a0, const afm_linear_args& a1, hipStream_t s) {
    constexpr int BM = 128, BN = 128, BKS = 16;
    constexpr int STAGE = 3 * (BM + BN) * (BKS * 2 + 16);
    constexpr int LDS_BYTES = 2 * STAGE + 2 * BM * 2 * 4;
    static const int attr = []() {
        return (int)hipFuncSetAttribute((const void*)gemm_f32_split_bf16_pair<BM, BN, BKS, NPROD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }();
    if (attr != 0) return attr;
    afm_linear_pair_args g;
    g.a[0] = a0; g.a[1] = a1;
    for (int i = 0; i < 2; ++i) { g.nbm[i] = (g.a[i].M + BM - 1) / BM; g.nbn[i] = (g.a[i].N + BN - 1) / BN; }
    g.tiles0 = g.nbm[0] * g.nbn[0];
    g.start1 = (g.tiles0 + 7) & ~7;
    const int grid = g.start1 + g.nbm[1] * g.nbn[1];
    AfmProf prof(AFM_PROF_GEMM_SPLIT128, 2.0 * a0.M * a0.N * a0.K + 2.0 * a1.M * a1.N * a1.K, s);
    hipLaunchKernelGGL((gemm_f32_split_bf16_pair<BM, BN, BKS, NPROD>), dim3(grid), dim3(256), LDS_BYTES, s, g);
    AFM_CHECK_LAUNCH();
    return 0;
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

#ifndef AFM_WALK_GRID
#define AFM_WALK_GRID 0            // > 0 (A/B builds only): every 64x64 launch with more tiles than this walks them on this many resident workgroups
#endif
#ifndef AFM_WALK_GRID_FORCED
#define AFM_WALK_GRID_FORCED 768   // resident workgroups of the forced form (tile code 12): three per CU
#endif
template <int NPROD>
int launch_split_walk(const afm_linear_args& a, hipStream_t s) {
    constexpr int BM = 64, BN = 64, BKS = 16;
    constexpr int STAGE = 3 * (BM + BN) * (BKS * 2 + 16);
    constexpr int LDS_BYTES = 2 * STAGE + 2 * BM * 2 * 4;
    const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
    const int64_t tiles = (int64_t)nbm * nbn;
    AfmProf prof(AFM_PROF_GEMM_SPLIT64, 2.0 * a.M * a.N * a.K, s);
    const int64_t grid = AFM_WALK_GRID > 0 ? AFM_WALK_GRID : AFM_WALK_GRID_FORCED;
    hipLaunchKernelGGL((gemm_f32_split_bf16_walk<BM, BN, BKS, NPROD>), dim3((unsigned)(tiles < grid ? tiles : grid)), dim3(256), LDS_BYTES, s, a, nbm, nbn);
    AFM_CHECK_LAUNCH();
    return 0;
}

template <int NPROD>
int dispatch_split(const afm_linear_args& a, hipStream_t s) {
    const int tile = (a.tune & AFM_TUNE_TILE_MASK) >> AFM_TUNE_TILE_SHIFT;      // 3 = 64x64, 5 = 128x128, 7 = 64x64 split-K, 8 = weight-stationary slabs, 0 = heuristic
    // Row-dot launches with K = 256 (the CDM's linear1): the weight-stationary form (gemm_slab.hip) whatever M is
    if ((NPROD == 9 || NPROD == 6) && (tile == 0 || tile == 8) && afm_linear_rowdot_slab_ok(a)) return afm_linear_rowdot_slab(a, NPROD, s);
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
    if (tile == 12) return launch_split_walk<NPROD>(a, s);          // measurement: force the walking form
    if (tile == 13) return launch_split_wide<NPROD>(a, s);          // measurement: the 256 x 128 tile on 512 threads
    if (tile == 3 || (tile != 5 && !full_rounds)) {
        const int64_t tiles64w = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
        if (tile != 3 && AFM_WALK_GRID > 0 && tiles64w > AFM_WALK_GRID) return launch_split_walk<NPROD>(a, s);      // (A/B builds only: AFM_WALK_GRID is 0 in the library)
        return launch_split<64, 64, 16, NPROD>(a, s);
    }
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
        case AFM_ARITH_DEFAULT: mode = 6; min_n = 32; break;       // round 6: six products (tests/test_gpu_arith.py); narrow outputs: native small-tile kernels (a function of N only, never of M)
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

int afm_linear_pair_split(const afm_linear_args& a0, const afm_linear_args& a1, int mode, hipStream_t s) {
    if (mode == 1) return launch_split_pair<1>(a0, a1, s);
    return mode == 9 ? launch_split_pair<9>(a0, a1, s) : launch_split_pair<6>(a0, a1, s);
}
