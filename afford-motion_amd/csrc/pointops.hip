// Furthest point sampling, brute-force kNN and row gather: replacements for the external CUDA
// extension `pointops_cuda` the reference calls (models/scene_models/pointops.py:23,42,91-94).
//
// Index semantics (bit-exact against oracle/pointops_ref.py):
//   d2 = (dx*dx + dy*dy) + dz*dz in float32 with fma contraction OFF;
//   FPS: start at the first point of the sample, tmp initialised to 1e10, ties -> lowest index;
//   kNN: neighbours ascending in (d2, index) lexicographic order.
// Every sample has the same point count, so batch offsets are implicit (sample b owns rows
// [b*n, (b+1)*n)); returned indices are GLOBAL row numbers like the reference's.
//
// gfx950 design
//   * FPS is latency-bound (m-1 dependent rounds): one workgroup per sample, every point and its
//     running min-distance live in registers, arg-max = 64-bit (dist bits | ~index) keys reduced with
//     4 DPP row steps + 4 readlanes per wave (no ds_bpermute), one LDS hop across <= 16 waves, ONE barrier per round.
//   * kNN: one lane per query, candidates streamed through an LDS tile (broadcast reads), the k best
//     kept as a sorted register array with a branch-free insertion.
#include "common.h"
#include "profile.h"
#pragma clang fp contract(off)

#ifdef AFM_PROBE
int afm_probe_fps_threads = 0;
int afm_probe_fps_prune = -1;                           // 0: plain fps_kernel even where the pruned one applies; 256 / 512 / 1024: its threads
__device__ unsigned long long afm_probe_fps_cyc[8];     // summed phase cycles of wave 0 of workgroup 0 (tools/points_probe.hip)
__device__ int afm_probe_fps_rounds[4];                 // [rounds of workgroup 0, picks taken as 1st / 2nd+ ...] (tools/fps_ks_probe.hip)
#endif
#ifdef AFM_PROBE_TIMELINE      // with -DAFM_PROBE: per-phase cycle stamps inside the FPS round (they lengthen the round by ~15 %)
#define AFM_FPS_STAMP(k) do { if (probe_) { const unsigned long long t_ = __builtin_readcyclecounter(); cyc_[k] += t_ - last_; last_ = t_; } } while (0)
#else
#define AFM_FPS_STAMP(k) do { } while (0)
#endif

#ifndef AFM_FPS_KS
#define AFM_FPS_KS 1             // samples per round of fps_pruned_kernel: 1 in the library.  KS = 2 / 4 / 8 (tools/fps_ks_probe.hip builds) are exact - same indices,
                                 // 2047 -> 1082 / 663 / 571 rounds at n = 8192, m = 2048 - and NOT faster: 1.48 -> 1.54 / 1.53 / 1.63 ms (profiles/r06_fps_speculation.md)
#endif

#ifndef AFM_FPS_CELLS
#define AFM_FPS_CELLS 1          // cells per thread of the pruned FPS: 1 = fps_pruned_kernel on 1024 threads (the library); 2 / 4 = fps_cells_kernel on 512 / 256 threads
                                 // (probe builds): identical indices, 1.48 -> 1.72 / 2.30 ms at n = 8192, m = 2048 (profiles/r06_fps_speculation.md)
#endif

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Per-point work is kept minimal (the round's VALU time is 16 waves x PPT points on ONE CU): coordinates of two slots are
// processed as packed f32 pairs (v_pk_add / v_pk_mul: same IEEE results, half the instructions), the per-thread arg-max is a
// 32-bit (distance bits, slot) select - slots are visited in increasing point index, so a strict '>' keeps the lowest index on
// ties - and the 64-bit (dist | ~index) key that the wave / workgroup reduction orders by is built once per thread per round.
template <int PPT, int MAXT, bool IN_LDS>
__global__ __launch_bounds__(MAXT) void fps_kernel(const float* __restrict__ xyz, int n, int m, int* __restrict__ idx_out) {
    constexpr int NP = (PPT + 1) / 2;                 // packed pairs (PPT == 1: second half is a dead slot)
    __shared__ unsigned long long keys[2][16];
    extern __shared__ float pts[];                    // [3n] copy of the sample for the winner's coordinates (critical path)
    const int b = blockIdx.x, T = blockDim.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
    const float* P = xyz + (int64_t)b * n * 3;
    f32x2 px[NP], py[NP], pz[NP];
    // running minimum distance of every slot, kept as BITS: distances are sums of squares (>= +0, never NaN), so unsigned order is
    // float order and v_min_u32 replaces fminf (no canonicalisation of the operands, the bits feed the arg-max directly).  A slot
    // without a point starts at 0 and stays there: it can never beat the strict '>' below (what the per-slot validity select did).
    unsigned tmpb[2 * NP];
    bool ok0 = false;
#pragma unroll
    for (int s = 0; s < 2 * NP; ++s) {
        const int i = s * T + tid;
        const bool ok = (s < PPT) && (i < n);
        if (s == 0) ok0 = ok;
        px[s >> 1][s & 1] = ok ? P[i * 3 + 0] : 0.f; py[s >> 1][s & 1] = ok ? P[i * 3 + 1] : 0.f; pz[s >> 1][s & 1] = ok ? P[i * 3 + 2] : 0.f;
        tmpb[s] = ok ? __float_as_uint(1e10f) : 0u;
    }
    if (IN_LDS) {
        for (int i0 = tid; i0 < 3 * n; i0 += 8 * T) {                // eight loads in flight per thread (not a load -> store chain per element)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = i0 + u * T < 3 * n ? P[i0 + u * T] : 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * T < 3 * n) pts[i0 + u * T] = v[u];
        }
        __syncthreads();
    }
    int cur = 0;
    if (tid == 0 && m > 0) idx_out[(int64_t)b * m] = b * n;
#ifdef AFM_PROBE_TIMELINE
    const bool probe_ = blockIdx.x == 0 && wave == 0;
    unsigned long long cyc_[6] = {0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
    for (int j = 1; j < m; ++j) {
        AFM_FPS_STAMP(5);
        // the winner's coordinates head the round's dependent chain: an explicit LDS read (ds_read), not a generic-pointer load
        // (a run-time `in_lds ? pts : P` compiles to flat_load, which waits on both memory counters and takes the slow aperture path)
        float cx, cy, cz;
        if (IN_LDS) { cx = pts[cur * 3 + 0]; cy = pts[cur * 3 + 1]; cz = pts[cur * 3 + 2]; }
        else { cx = P[cur * 3 + 0]; cy = P[cur * 3 + 1]; cz = P[cur * 3 + 2]; }
        AFM_FPS_STAMP(0);                             // winner's coordinates arrived
        const f32x2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
        unsigned bd = 0u, bs = 0u;                    // best distance bits (distances are >= 0: bits order like values), its slot
#pragma unroll
        for (int p2 = 0; p2 < NP; ++p2) {
            const f32x2 dx = px[p2] - cx2, dy = py[p2] - cy2, dz = pz[p2] - cz2;
            const f32x2 d = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int s = 2 * p2 + h;
                if (s < PPT) {
                    const unsigned db = min(tmpb[s], __float_as_uint(d[h]));
                    tmpb[s] = db;
                    const bool better = db > bd;
                    bd = better ? db : bd;
                    bs = better ? (unsigned)s : bs;
                }
            }
        }
        // slot 0 wins an all-zero tie, which is also the lowest index of this thread; invalid threads carry key 0.
        // The key orders by (distance bits, lowest point index).  Wave level: a 32-bit maximum of the distances (DPP), then the lowest
        // index among the lanes that hold it - one ballot and one v_readlane when a single lane does (the usual case), a second 32-bit
        // reduction only on a tie.  (A 64-bit key carried through the DPP steps costs a v_cmp_u64 + two selects per step: 250 cycles
        // of the ~2200-cycle round, profiles/r02_points_probe.txt.)
        const unsigned bi = bs * (unsigned)T + (unsigned)tid;
        AFM_FPS_STAMP(1);                             // per-thread scan
        const unsigned wmax = wave_max_u32(ok0 ? bd : 0u);
        const unsigned long long holders = __builtin_amdgcn_ballot_w64(ok0 && bd == wmax);
        unsigned long long best = 0ull;
        if (holders) {                                // wave-uniform
            unsigned wbi;
            if ((holders & (holders - 1)) == 0) wbi = (unsigned)__builtin_amdgcn_readlane((int)bi, __builtin_ctzll(holders));
            else wbi = wave_min_u32((ok0 && bd == wmax) ? bi : 0xFFFFFFFFu);
            best = ((unsigned long long)wmax << 32) | (unsigned long long)(0xFFFFFFFFu - wbi);
        }
        AFM_FPS_STAMP(2);                             // wave arg-max
        if (lane == 0) keys[j & 1][wave] = best;
        __syncthreads();
        AFM_FPS_STAMP(3);                             // LDS hop + barrier
        // workgroup level, same scheme on the <= 16 wave keys (lane l of every row reads key l; one LDS round trip)
        const bool has = (lane & 15) < nw;
        const unsigned long long kw = keys[j & 1][has ? (lane & 15) : 0];
        const unsigned khi = has ? (unsigned)(kw >> 32) : 0u, klo = has ? (unsigned)kw : 0u;       // lo = ~index: larger is the lower index
        unsigned gm = khi;
        gm = max(gm, dpp_u32<0xB1>(gm)); gm = max(gm, dpp_u32<0x4E>(gm)); gm = max(gm, dpp_u32<0x141>(gm)); gm = max(gm, dpp_u32<0x140>(gm));
        const unsigned long long hold2 = __builtin_amdgcn_ballot_w64(has && lane < 16 && khi == gm);       // row 0 is enough: the rows are copies
        unsigned wlo;
        if ((hold2 & (hold2 - 1)) == 0) wlo = (unsigned)__builtin_amdgcn_readlane((int)klo, __builtin_ctzll(hold2 | (1ull << 63)));
        else wlo = wave_max_u32((has && khi == gm) ? klo : 0u);
        cur = __builtin_amdgcn_readfirstlane((int)(0xFFFFFFFFu - wlo));
        AFM_FPS_STAMP(4);                             // cross-wave arg-max
        if (tid == 0) idx_out[(int64_t)b * m + j] = b * n + cur;
    }
#ifdef AFM_PROBE_TIMELINE
    if (probe_ && lane == 0) for (int k = 0; k < 6; ++k) afm_probe_fps_cyc[k] = cyc_[k];
#endif
}

// ---------------------------------------------------------------- FPS with an exact pruning of the per-round scan
// The plain kernel above updates the running minimum of ALL n points every round: ~1100 of its ~2100 cycles per round, on the one compute
// unit a sample occupies.  Most of those updates are no-ops: a point's minimum only changes if the new sample is closer than its current
// minimum.  Here the points of a sample are sorted along a Morton curve once (bitonic sort of (key, index) in LDS, ~20 us) and a thread owns
// PPT CONSECUTIVE sorted points - a compact cell with a bounding box - and a wave 64 neighbouring cells.  Per round a thread evaluates the
// squared distance from the new sample to its box with the SAME float operations and association as the point distances
// (ex = max(lo - c, c - hi, 0) per axis, (ex^2 + ey^2) + ez^2): rounding is monotonic, so this is a lower bound of every COMPUTED distance
// of the cell, and if it is not below the cell's current maximum no minimum can change - the wave skips the cell's update when none of its 64
// cells needs one (16 % of the waves scan per round at n = 8192, m = 2048).  The kept state (running minima, per-cell maximum and its lowest
// ORIGINAL index) is exactly what the plain kernel would hold, so the sampled indices are identical, ties included (lowest original index).
__device__ __forceinline__ unsigned morton_part10(unsigned x) {
    x &= 0x3ffu; x = (x | (x << 16)) & 0x30000ffu; x = (x | (x << 8)) & 0x300f00fu; x = (x | (x << 4)) & 0x30c30c3u; x = (x | (x << 2)) & 0x9249249u;
    return x;
}

// KS > 1 (round 6): up to KS samples PER ROUND.  A round's cost is its dependent chain (winner's coordinates -> box test -> scan -> wave arg-max ->
// LDS hop -> barrier -> cross-wave arg-max: ~1500 cycles), and the chain of m - 1 rounds is the kernel.  But the point with the SECOND largest
// running minimum is almost always far from the one with the largest: if its distance to the new sample is not below its own running minimum, that
// minimum does not change, every other minimum can only shrink, and it IS the next sample - provably, not heuristically (keys are (distance
// bits, lowest index), all unique).  So a round takes the best point, then the next best while that one is unaffected by every point taken before it
// in the round (checked with the same float expression the scan uses), and the next round scans with all of them as centres.  What a wave
// publishes is its best key and the distance bits of its SECOND best point: a candidate from another wave is only taken while its distance is
// strictly above the second-best distance of every wave whose best was already taken (whose other points are not published).  Measured on
// the synthetic scenes (profiles/r06_fps_speculation.md): 2047 rounds -> 1082 / 663 / 571 with KS = 2 / 4 / 8 (n = 8192, m = 2048), the sampled
// indices identical by construction and by test (tests/test_gpu_points.py ran green on a KS = 4 build) - and the kernel NO faster: a round grows
// from 0.72 to 1.42 / 2.31 / 2.85 us.  The 16 waves of a sample share ONE compute unit, whose four SIMDs issue every wave's per-round
// instructions: the uniform selection logic of an extra sample (~50 instructions, executed by all 16 waves, plus a box test per centre and
// wave) costs the CU as many issue slots as the whole one-sample round.  Kept as a measured form (AFM_FPS_KS), not compiled into the library.
template <int PPT, int T, int KS = 1>
__global__ __launch_bounds__(T) void fps_pruned_kernel(const float* __restrict__ xyz, int n, int m, int* __restrict__ idx_out) {
    constexpr int CAP = T * PPT, NP = PPT / 2;
    static_assert(PPT % 2 == 0 && PPT <= 32 && T % 64 == 0 && T <= 1024, "cell sizes");
    __shared__ unsigned long long keys[2][16];
    __shared__ float bb[6][16];
    extern __shared__ float pts[];                    // first the sort arrays (2 x CAP words), then the [3n] copy of the sample
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nw = T / 64;
    const float* P = xyz + (int64_t)b * n * 3;
    // ---- bounding box of the sample (for the Morton quantisation only: any permutation gives the same samples)
    float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += T) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = P[i * 3 + a]; lo3[a] = fminf(lo3[a], v); hi3[a] = fmaxf(hi3[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo3[a] = fminf(lo3[a], __shfl_xor(lo3[a], o)); hi3[a] = fmaxf(hi3[a], __shfl_xor(hi3[a], o)); }
        if (lane == 0) { bb[a][wave] = lo3[a]; bb[3 + a][wave] = hi3[a]; }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = bb[a][0], h = bb[3 + a][0];
        for (int w2 = 1; w2 < nw; ++w2) { l = fminf(l, bb[a][w2]); h = fmaxf(h, bb[3 + a][w2]); }
        lo3[a] = l; hi3[a] = 1023.0f / fmaxf(h - l, 1e-12f);          // hi3 now holds the quantisation scale
    }
    // ---- Morton keys and the bitonic sort of (key, original index); entries beyond n sort to the end
    unsigned* kk = reinterpret_cast<unsigned*>(pts);
    unsigned* vv = kk + CAP;
    for (int i = tid; i < CAP; i += T) {
        unsigned key = 0xFFFFFFFFu;
        if (i < n) {
            unsigned q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float f = (P[i * 3 + a] - lo3[a]) * hi3[a]; q[a] = (unsigned)min(1023, max(0, (int)f)); }
            key = morton_part10(q[0]) | (morton_part10(q[1]) << 1) | (morton_part10(q[2]) << 2);
        }
        kk[i] = key; vv[i] = i < n ? (unsigned)i : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int k = 2; k <= CAP; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int i = tid; i < CAP; i += T) {
                const int x = i ^ jj;
                if (x > i) {
                    const unsigned ka = kk[i], kb = kk[x];
                    if ((ka > kb) == ((i & k) == 0)) { kk[i] = kb; kk[x] = ka; const unsigned va = vv[i]; vv[i] = vv[x]; vv[x] = va; }
                }
            }
            __syncthreads();
        }
    }
    // ---- this thread's cell: PPT consecutive sorted points, their original indices, the cell's box
    unsigned oi[PPT];
    f32x2 px[NP], py[NP], pz[NP];
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool cell_ok = false;
#pragma unroll
    for (int s2 = 0; s2 < PPT; ++s2) {
        oi[s2] = vv[tid * PPT + s2];
        const bool ok = oi[s2] != 0xFFFFFFFFu;
        cell_ok = cell_ok || ok;
        const float x = ok ? P[oi[s2] * 3 + 0] : 0.f, y = ok ? P[oi[s2] * 3 + 1] : 0.f, z = ok ? P[oi[s2] * 3 + 2] : 0.f;
        px[s2 >> 1][s2 & 1] = x; py[s2 >> 1][s2 & 1] = y; pz[s2 >> 1][s2 & 1] = z;
        if (ok) { clo[0] = fminf(clo[0], x); chi[0] = fmaxf(chi[0], x); clo[1] = fminf(clo[1], y); chi[1] = fmaxf(chi[1], y); clo[2] = fminf(clo[2], z); chi[2] = fmaxf(chi[2], z); }
    }
    __syncthreads();                                  // the sort arrays are dead: the same LDS becomes the copy of the sample
    for (int i0 = tid; i0 < 3 * n; i0 += 8 * T) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + u * T < 3 * n ? P[i0 + u * T] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + u * T < 3 * n) pts[i0 + u * T] = v[u];
    }
    __syncthreads();
    // running minima as bits (see fps_kernel); an empty slot holds 0 and the index 0xFFFFFFFF: it can never win
    unsigned tmpb[PPT];
#pragma unroll
    for (int s2 = 0; s2 < PPT; ++s2) tmpb[s2] = oi[s2] != 0xFFFFFFFFu ? __float_as_uint(1e10f) : 0u;
    unsigned bd = 0u, boi = 0xFFFFFFFFu;              // the cell's maximum and the lowest original index that holds it
    auto cell_argmax = [&]() {                        // two passes: the maximum (v_max3), then the lowest original index that holds it (v_min3)
        bd = 0u;
#pragma unroll
        for (int s2 = 0; s2 < PPT; s2 += 2) bd = max(bd, max(tmpb[s2], tmpb[s2 + 1]));
        boi = 0xFFFFFFFFu;
#pragma unroll
        for (int s2 = 0; s2 < PPT; s2 += 2)
            boi = min(boi, min(tmpb[s2] == bd ? oi[s2] : 0xFFFFFFFFu, tmpb[s2 + 1] == bd ? oi[s2 + 1] : 0xFFFFFFFFu));
    };
    cell_argmax();
    if constexpr (KS > 1) {
        __shared__ uint4 rec[2][16];                  // per wave: (best distance bits, ~lowest index, second-best distance bits of the wave, -)
        unsigned long long bestk = 0ull;              // this wave's best key and second-best distance, kept across the rounds in which the wave skips
        unsigned sec = 0u;
        int ncen = 1;
        int cen[KS];
        float ccx[KS], ccy[KS], ccz[KS];
#pragma unroll
        for (int c = 0; c < KS; ++c) { cen[c] = 0; ccx[c] = pts[0]; ccy[c] = pts[1]; ccz[c] = pts[2]; }
        if (tid == 0 && m > 0) idx_out[(int64_t)b * m] = b * n;
        int j = 1, round = 0;
        while (j < m) {
            bool scanned = false;
#pragma unroll
            for (int c = 0; c < KS; ++c) {
                if (c < ncen) {                                       // wave-uniform
                    const float cx = ccx[c], cy = ccy[c], cz = ccz[c];
                    // lower bound of every computed distance of the cell (same operations, same association as the scan); `bd` may be stale
                    // (too large) after an earlier centre of this round: the test is then only more often true
                    const float ex = fmaxf(fmaxf(clo[0] - cx, cx - chi[0]), 0.f), ey = fmaxf(fmaxf(clo[1] - cy, cy - chi[1]), 0.f), ez = fmaxf(fmaxf(clo[2] - cz, cz - chi[2]), 0.f);
                    const float lb = (ex * ex + ey * ey) + ez * ez;
                    const bool need = cell_ok && __float_as_uint(lb) < bd;
                    if (__builtin_amdgcn_ballot_w64(need) != 0ull) {
                        const f32x2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
#pragma unroll
                        for (int p2 = 0; p2 < NP; ++p2) {
                            const f32x2 dx = px[p2] - cx2, dy = py[p2] - cy2, dz = pz[p2] - cz2;
                            const f32x2 d = (dx * dx + dy * dy) + dz * dz;
                            tmpb[2 * p2] = min(tmpb[2 * p2], __float_as_uint(d[0]));
                            tmpb[2 * p2 + 1] = min(tmpb[2 * p2 + 1], __float_as_uint(d[1]));
                        }
                        scanned = true;
                    }
                }
            }
            if (scanned) cell_argmax();
            if (scanned || round == 0) {
                const unsigned wmax = wave_max_u32(cell_ok ? bd : 0u);
                const unsigned long long holders = __builtin_amdgcn_ballot_w64(cell_ok && bd == wmax);
                bestk = 0ull; sec = 0u;
                if (holders) {
                    unsigned wbi;
                    if ((holders & (holders - 1)) == 0) wbi = (unsigned)__builtin_amdgcn_readlane((int)boi, __builtin_ctzll(holders));
                    else wbi = wave_min_u32((cell_ok && bd == wmax) ? boi : 0xFFFFFFFFu);
                    bestk = ((unsigned long long)wmax << 32) | (unsigned long long)(0xFFFFFFFFu - wbi);
                    // second-best DISTANCE of the wave: every point but the best one (wmax, wbi); a second point at the same distance counts
                    unsigned v2 = cell_ok ? bd : 0u;
                    if (cell_ok && bd == wmax && boi == wbi) {
                        v2 = 0u;
#pragma unroll
                        for (int s2 = 0; s2 < PPT; ++s2) v2 = max(v2, oi[s2] == wbi ? 0u : tmpb[s2]);
                    }
                    sec = wave_max_u32(v2);
                }
            }
            if (lane == 0) rec[round & 1][wave] = make_uint4((unsigned)(bestk >> 32), (unsigned)bestk, sec, 0u);
            __syncthreads();
            // ---- cross-wave: up to KS picks in key order (every wave computes them redundantly from the published records: no second barrier)
            const bool has = (lane & 15) < nw;
            const uint4 rw = rec[round & 1][has ? (lane & 15) : 0];
            unsigned khi = has ? rw.x : 0u, klo = has ? rw.y : 0u;
            const unsigned ksec = has ? rw.z : 0u;
            unsigned thr = 0u;                            // largest second-best distance among the waves whose best was taken this round
            int nacc = 0;
            const int room = m - j < KS ? m - j : KS;
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                if (i < room && (i == 0 || nacc == i)) {              // wave-uniform: still extending the prefix
                    unsigned gm = khi;
                    gm = max(gm, dpp_u32<0xB1>(gm)); gm = max(gm, dpp_u32<0x4E>(gm)); gm = max(gm, dpp_u32<0x141>(gm)); gm = max(gm, dpp_u32<0x140>(gm));
                    unsigned long long hold = __builtin_amdgcn_ballot_w64(has && lane < 16 && khi == gm);
                    if (hold & (hold - 1)) {                          // several waves at this distance: the lowest index (largest ~index)
                        const unsigned wlo = wave_max_u32((has && lane < 16 && khi == gm) ? klo : 0u);
                        hold = __builtin_amdgcn_ballot_w64(has && lane < 16 && khi == gm && klo == wlo);
                    }
                    const int wl = __builtin_ctzll(hold | (1ull << 63)) & 15;
                    const unsigned hi_i = (unsigned)__builtin_amdgcn_readlane((int)khi, wl), lo_i = (unsigned)__builtin_amdgcn_readlane((int)klo, wl);
                    const unsigned sec_i = (unsigned)__builtin_amdgcn_readlane((int)ksec, wl);
                    const int ci = (int)(0xFFFFFFFFu - lo_i);
                    bool ok = true;
                    float qx = 0.f, qy = 0.f, qz = 0.f;
                    if (i > 0) ok = hi_i > thr;                       // (strictly above: a point at the same distance in a taken wave could come first; also excludes distance 0)
                    if (ok) {
                        qx = pts[ci * 3 + 0]; qy = pts[ci * 3 + 1]; qz = pts[ci * 3 + 2];
#pragma unroll
                        for (int a = 0; a < KS; ++a) {
                            if (a < i && ok) {                        // unaffected by every point taken before it: the scan's own expression, point minus centre
                                const float dx = qx - ccx[a], dy = qy - ccy[a], dz = qz - ccz[a];
                                const float d = (dx * dx + dy * dy) + dz * dz;
                                ok = __float_as_uint(d) >= hi_i;
                            }
                        }
                    }
                    if (ok) {
                        cen[i] = ci; ccx[i] = qx; ccy[i] = qy; ccz[i] = qz;
                        nacc = i + 1;
                        thr = max(thr, sec_i);
                        if ((lane & 15) == wl) { khi = 0u; klo = 0u; }         // taken (all four row copies)
                    }
                }
            }
            if (tid == 0) {
#pragma unroll
                for (int i = 0; i < KS; ++i)
                    if (i < nacc) idx_out[(int64_t)b * m + j + i] = b * n + cen[i];
            }
            ncen = nacc;
            j += nacc;
            ++round;
        }
#ifdef AFM_PROBE
        if (blockIdx.x == 0 && tid == 0) afm_probe_fps_rounds[0] = round;
#endif
        return;
    }
    int cur = 0;
    unsigned long long best = 0ull;                   // this wave's key, kept across the rounds in which the wave skips
    if (tid == 0 && m > 0) idx_out[(int64_t)b * m] = b * n;
    for (int j = 1; j < m; ++j) {
        const float cx = pts[cur * 3 + 0], cy = pts[cur * 3 + 1], cz = pts[cur * 3 + 2];
        // lower bound of every computed distance of the cell (same operations, same association as below)
        const float ex = fmaxf(fmaxf(clo[0] - cx, cx - chi[0]), 0.f), ey = fmaxf(fmaxf(clo[1] - cy, cy - chi[1]), 0.f), ez = fmaxf(fmaxf(clo[2] - cz, cz - chi[2]), 0.f);
        const float lb = (ex * ex + ey * ey) + ez * ez;
        const bool need = cell_ok && __float_as_uint(lb) < bd;
        const bool scan = __builtin_amdgcn_ballot_w64(need) != 0ull;
        if (scan) {                                               // wave-uniform: some cell of the wave can change
            const f32x2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) {
                const f32x2 dx = px[p2] - cx2, dy = py[p2] - cy2, dz = pz[p2] - cz2;
                const f32x2 d = (dx * dx + dy * dy) + dz * dz;
                tmpb[2 * p2] = min(tmpb[2 * p2], __float_as_uint(d[0]));
                tmpb[2 * p2 + 1] = min(tmpb[2 * p2 + 1], __float_as_uint(d[1]));
            }
            cell_argmax();
        }
        // (distance bits, lowest original index): wave level, then the 16 wave keys - as in fps_kernel.  A wave that skipped the update holds
        // the cells it held the round before: its key is unchanged and only re-published (the key buffers alternate)
        if (scan || j == 1) {
            const unsigned wmax = wave_max_u32(cell_ok ? bd : 0u);
            const unsigned long long holders = __builtin_amdgcn_ballot_w64(cell_ok && bd == wmax);
            best = 0ull;
            if (holders) {
                unsigned wbi;
                if ((holders & (holders - 1)) == 0) wbi = (unsigned)__builtin_amdgcn_readlane((int)boi, __builtin_ctzll(holders));
                else wbi = wave_min_u32((cell_ok && bd == wmax) ? boi : 0xFFFFFFFFu);
                best = ((unsigned long long)wmax << 32) | (unsigned long long)(0xFFFFFFFFu - wbi);
            }
        }
        if (lane == 0) keys[j & 1][wave] = best;
        __syncthreads();
        const bool has = (lane & 15) < nw;
        const unsigned long long kw = keys[j & 1][has ? (lane & 15) : 0];
        const unsigned khi = has ? (unsigned)(kw >> 32) : 0u, klo = has ? (unsigned)kw : 0u;
        unsigned gm = khi;
        gm = max(gm, dpp_u32<0xB1>(gm)); gm = max(gm, dpp_u32<0x4E>(gm)); gm = max(gm, dpp_u32<0x141>(gm)); gm = max(gm, dpp_u32<0x140>(gm));
        const unsigned long long hold2 = __builtin_amdgcn_ballot_w64(has && lane < 16 && khi == gm);
        unsigned wlo;
        if ((hold2 & (hold2 - 1)) == 0) wlo = (unsigned)__builtin_amdgcn_readlane((int)klo, __builtin_ctzll(hold2 | (1ull << 63)));
        else wlo = wave_max_u32((has && khi == gm) ? klo : 0u);
        cur = __builtin_amdgcn_readfirstlane((int)(0xFFFFFFFFu - wlo));
        if (tid == 0) idx_out[(int64_t)b * m + j] = b * n + cur;
    }
}

// ---------------------------------------------------------------- FPS, several cells per thread (round 6)
// A TEST of the hypothesis that fps_pruned_kernel's round is bound by the instruction slots of the one compute unit a sample occupies (of ~1300 slots per round,
// ~720 are work every wave repeats whatever it owns: the cross-wave arg-max, publish + barrier, the box test) - REFUTED by this kernel: with the same cells on
// half / a quarter of the waves the kernel is 16 % / 55 % SLOWER (1.48 -> 1.72 / 2.30 ms).  The round is a dependent latency chain per wave (winner's coordinates ->
// box test -> scan -> arg-max -> LDS -> barrier -> cross-wave arg-max), and sixteen waves walk their cells in parallel where eight or four walk them in turn.
// MEASUREMENT form (AFM_FPS_CELLS, probe builds only).  A thread owns CELLS cells of PPT sorted points each (cell c of thread t = sorted points [(c T + t) PPT, + PPT): slot c of a wave's 64 lanes is what one
// wave of the T x CELLS-thread layout owns), so the pruning granularity and the wave-level skip are unchanged while the per-wave overheads are paid by T / 64
// waves instead of CELLS T / 64.  Same state, same arithmetic, same (distance bits, lowest original index) order: identical indices.
template <int PPT, int T, int CELLS>
__global__ __launch_bounds__(T) void fps_cells_kernel(const float* __restrict__ xyz, int n, int m, int* __restrict__ idx_out) {
    constexpr int CAP = T * PPT * CELLS, NP = PPT / 2;
    static_assert(PPT % 2 == 0 && PPT <= 16 && T % 64 == 0 && T <= 1024 && CELLS >= 1 && CELLS <= 8, "cell sizes");
    __shared__ unsigned long long keys[2][16];
    __shared__ float bb[6][16];
    extern __shared__ float pts[];                    // first the sort arrays (2 x CAP words), then the [3n] copy of the sample
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int nw = T / 64;
    const float* P = xyz + (int64_t)b * n * 3;
    float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += T) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = P[i * 3 + a]; lo3[a] = fminf(lo3[a], v); hi3[a] = fmaxf(hi3[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo3[a] = fminf(lo3[a], __shfl_xor(lo3[a], o)); hi3[a] = fmaxf(hi3[a], __shfl_xor(hi3[a], o)); }
        if (lane == 0) { bb[a][wave] = lo3[a]; bb[3 + a][wave] = hi3[a]; }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = bb[a][0], h = bb[3 + a][0];
        for (int w2 = 1; w2 < nw; ++w2) { l = fminf(l, bb[a][w2]); h = fmaxf(h, bb[3 + a][w2]); }
        lo3[a] = l; hi3[a] = 1023.0f / fmaxf(h - l, 1e-12f);
    }
    unsigned* kk = reinterpret_cast<unsigned*>(pts);
    unsigned* vv = kk + CAP;
    for (int i = tid; i < CAP; i += T) {
        unsigned key = 0xFFFFFFFFu;
        if (i < n) {
            unsigned q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float f = (P[i * 3 + a] - lo3[a]) * hi3[a]; q[a] = (unsigned)min(1023, max(0, (int)f)); }
            key = morton_part10(q[0]) | (morton_part10(q[1]) << 1) | (morton_part10(q[2]) << 2);
        }
        kk[i] = key; vv[i] = i < n ? (unsigned)i : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int k = 2; k <= CAP; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int i = tid; i < CAP; i += T) {
                const int x = i ^ jj;
                if (x > i) {
                    const unsigned ka = kk[i], kb = kk[x];
                    if ((ka > kb) == ((i & k) == 0)) { kk[i] = kb; kk[x] = ka; const unsigned va = vv[i]; vv[i] = vv[x]; vv[x] = va; }
                }
            }
            __syncthreads();
        }
    }
    // ---- this thread's cells
    unsigned oi[CELLS][PPT];
    f32x2 px[CELLS][NP], py[CELLS][NP], pz[CELLS][NP];
    float clo[CELLS][3], chi[CELLS][3];
    bool cell_ok[CELLS];
#pragma unroll
    for (int c = 0; c < CELLS; ++c) {
        cell_ok[c] = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) { clo[c][a] = INFINITY; chi[c][a] = -INFINITY; }
#pragma unroll
        for (int s2 = 0; s2 < PPT; ++s2) {
            oi[c][s2] = vv[(c * T + tid) * PPT + s2];
            const bool ok = oi[c][s2] != 0xFFFFFFFFu;
            cell_ok[c] = cell_ok[c] || ok;
            const float x = ok ? P[oi[c][s2] * 3 + 0] : 0.f, y = ok ? P[oi[c][s2] * 3 + 1] : 0.f, z = ok ? P[oi[c][s2] * 3 + 2] : 0.f;
            px[c][s2 >> 1][s2 & 1] = x; py[c][s2 >> 1][s2 & 1] = y; pz[c][s2 >> 1][s2 & 1] = z;
            if (ok) { clo[c][0] = fminf(clo[c][0], x); chi[c][0] = fmaxf(chi[c][0], x); clo[c][1] = fminf(clo[c][1], y); chi[c][1] = fmaxf(chi[c][1], y);
                      clo[c][2] = fminf(clo[c][2], z); chi[c][2] = fmaxf(chi[c][2], z); }
        }
    }
    __syncthreads();                                  // the sort arrays are dead: the same LDS becomes the copy of the sample
    for (int i0 = tid; i0 < 3 * n; i0 += 8 * T) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i0 + u * T < 3 * n ? P[i0 + u * T] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + u * T < 3 * n) pts[i0 + u * T] = v[u];
    }
    __syncthreads();
    unsigned tmpb[CELLS][PPT], bd[CELLS], boi[CELLS];
    auto cell_argmax = [&](int c) {
        unsigned d = 0u;
#pragma unroll
        for (int s2 = 0; s2 < PPT; s2 += 2) d = max(d, max(tmpb[c][s2], tmpb[c][s2 + 1]));
        unsigned o = 0xFFFFFFFFu;
#pragma unroll
        for (int s2 = 0; s2 < PPT; s2 += 2)
            o = min(o, min(tmpb[c][s2] == d ? oi[c][s2] : 0xFFFFFFFFu, tmpb[c][s2 + 1] == d ? oi[c][s2 + 1] : 0xFFFFFFFFu));
        bd[c] = d; boi[c] = o;
    };
    bool any_ok = false;
#pragma unroll
    for (int c = 0; c < CELLS; ++c) {
#pragma unroll
        for (int s2 = 0; s2 < PPT; ++s2) tmpb[c][s2] = oi[c][s2] != 0xFFFFFFFFu ? __float_as_uint(1e10f) : 0u;
        cell_argmax(c);
        any_ok = any_ok || cell_ok[c];
    }
    int cur = 0;
    unsigned long long best = 0ull;                   // this wave's key, kept across the rounds in which the wave skips
    if (tid == 0 && m > 0) idx_out[(int64_t)b * m] = b * n;
    for (int j = 1; j < m; ++j) {
        const float cx = pts[cur * 3 + 0], cy = pts[cur * 3 + 1], cz = pts[cur * 3 + 2];
        bool scanned = false;
#pragma unroll
        for (int c = 0; c < CELLS; ++c) {
            const float ex = fmaxf(fmaxf(clo[c][0] - cx, cx - chi[c][0]), 0.f), ey = fmaxf(fmaxf(clo[c][1] - cy, cy - chi[c][1]), 0.f),
                        ez = fmaxf(fmaxf(clo[c][2] - cz, cz - chi[c][2]), 0.f);
            const float lb = (ex * ex + ey * ey) + ez * ez;
            const bool need = cell_ok[c] && __float_as_uint(lb) < bd[c];
            if (__builtin_amdgcn_ballot_w64(need) != 0ull) {          // wave-uniform: some cell of slot c can change
                const f32x2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
#pragma unroll
                for (int p2 = 0; p2 < NP; ++p2) {
                    const f32x2 dx = px[c][p2] - cx2, dy = py[c][p2] - cy2, dz = pz[c][p2] - cz2;
                    const f32x2 d = (dx * dx + dy * dy) + dz * dz;
                    tmpb[c][2 * p2] = min(tmpb[c][2 * p2], __float_as_uint(d[0]));
                    tmpb[c][2 * p2 + 1] = min(tmpb[c][2 * p2 + 1], __float_as_uint(d[1]));
                }
                cell_argmax(c);
                scanned = true;
            }
        }
        if (scanned || j == 1) {
            // this thread's best over its cells: (distance bits, lowest original index)
            unsigned tb = bd[0], to = boi[0];
#pragma unroll
            for (int c = 1; c < CELLS; ++c) {
                const bool better = bd[c] > tb || (bd[c] == tb && boi[c] < to);
                tb = better ? bd[c] : tb; to = better ? boi[c] : to;
            }
            const unsigned wmax = wave_max_u32(any_ok ? tb : 0u);
            const unsigned long long holders = __builtin_amdgcn_ballot_w64(any_ok && tb == wmax);
            best = 0ull;
            if (holders) {
                unsigned wbi;
                if ((holders & (holders - 1)) == 0) wbi = (unsigned)__builtin_amdgcn_readlane((int)to, __builtin_ctzll(holders));
                else wbi = wave_min_u32((any_ok && tb == wmax) ? to : 0xFFFFFFFFu);
                best = ((unsigned long long)wmax << 32) | (unsigned long long)(0xFFFFFFFFu - wbi);
            }
        }
        if (lane == 0) keys[j & 1][wave] = best;
        __syncthreads();
        const bool has = (lane & 15) < nw;
        const unsigned long long kw = keys[j & 1][has ? (lane & 15) : 0];
        const unsigned khi = has ? (unsigned)(kw >> 32) : 0u, klo = has ? (unsigned)kw : 0u;
        unsigned gm = khi;
        gm = max(gm, dpp_u32<0xB1>(gm)); gm = max(gm, dpp_u32<0x4E>(gm)); gm = max(gm, dpp_u32<0x141>(gm)); gm = max(gm, dpp_u32<0x140>(gm));
        const unsigned long long hold2 = __builtin_amdgcn_ballot_w64(has && lane < 16 && khi == gm);
        unsigned wlo;
        if ((hold2 & (hold2 - 1)) == 0) wlo = (unsigned)__builtin_amdgcn_readlane((int)klo, __builtin_ctzll(hold2 | (1ull << 63)));
        else wlo = wave_max_u32((has && khi == gm) ? klo : 0u);
        cur = __builtin_amdgcn_readfirstlane((int)(0xFFFFFFFFu - wlo));
        if (tid == 0) idx_out[(int64_t)b * m + j] = b * n + cur;
    }
}

constexpr int KNN_TILE = 1024;

// One lane per query, candidates tiled through LDS.  A candidate that beats the lane's current k-th distance is NOT inserted
// at once: a wave executes a divergent branch whenever ANY of its 64 lanes takes it, and with 64 independent queries some lane
// nearly always does (each query takes ~K ln(n / K) ~ 100 candidates of 8192, i.e. ~0.8 takes per candidate per wave), so the 6K-instruction
// sorted insert would run for almost every candidate.  Instead a qualifying candidate is APPENDED to a per-lane buffer in LDS: one
// branch-free ds_write_b64 whose row is the lane's fill count (non-qualifying lanes write to a dump row), and the wave drains the
// buffers with one wave-uniform branch when some lane is nearly full.  Round 1 kept a 3-entry buffer in registers (six v_cndmask per
// candidate); the LDS form costs 4 VALU + 1 LDS store per candidate and leaves room for 8 entries, so the drain runs half as often.
// Candidates are consumed four per trip (three broadcast ds_read_b128, four independent distance chains): the kernel runs ONE wave
// per SIMD (65536 queries are 1024 waves), so instruction-level parallelism is all the latency hiding there is.
// Insertion order per lane is still the scan order and uses strict '<', so the result is identical to the direct form:
// neighbours ordered by (d2, index).
template <int K>
__global__ __launch_bounds__(256) void knn_kernel(const float* __restrict__ xyz, const float* __restrict__ qxyz, int n, int m,
                                                  int* __restrict__ idx_out, float* __restrict__ d2_out) {
    constexpr int BUF = 8;
    __shared__ __attribute__((aligned(16))) float tile[3 * KNN_TILE];      // interleaved x,y,z exactly as in memory (coalesced fill)
    __shared__ float2 buf[BUF + 1][256];                                    // [entry][thread] (conflict-free), row BUF = dump row
    const int b = blockIdx.y, tid = threadIdx.x;
    const int q = blockIdx.x * blockDim.x + tid;
    const bool valid = q < m;
    const float* Q = qxyz + ((int64_t)b * m + (valid ? q : 0)) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2];
    const float* P = xyz + (int64_t)b * n * 3;
    float bd[K];
    int bi[K];
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = INFINITY; bi[j] = -1; }
    auto flush = [&]() {
        for (int e = 0; e < BUF; ++e) {
            if (!__any(e < cnt)) break;                              // wave-uniform
            const float2 v = buf[e][tid];
            const float d = e < cnt ? v.x : INFINITY;                // unused slots can never beat anything
            const int id = __float_as_int(v.y);
            // sorted insert; strict '<' keeps the earlier (lower) index in front on equal distances
#pragma unroll
            for (int j = K - 1; j > 0; --j) {
                const bool up = d < bd[j - 1];
                const bool here = d < bd[j];
                bd[j] = up ? bd[j - 1] : (here ? d : bd[j]);
                bi[j] = up ? bi[j - 1] : (here ? id : bi[j]);
            }
            const bool first = d < bd[0];
            bd[0] = first ? d : bd[0];
            bi[0] = first ? id : bi[0];
        }
        cnt = 0;
    };
    auto offer = [&](float d, int id) {                              // append if it beats the (possibly stale) k-th distance: re-checked at the drain
        const bool take = d < bd[K - 1];
        buf[take ? cnt : BUF][tid] = make_float2(d, __int_as_float(id));
        cnt += take ? 1 : 0;
    };
    // tile fill through registers, one tile ahead: all of a thread's loads are in flight before the first store (a load -> store loop
    // waits a full memory round trip per element), and the next tile's loads are issued before the current tile is scanned
    constexpr int NF = 3 * KNN_TILE / 256;
    float stage[NF];
    auto fetch = [&](int t0_) {
        const int lim = (min(KNN_TILE, n - t0_)) * 3;
#pragma unroll
        for (int u = 0; u < NF; ++u) { const int f = tid + u * 256; stage[u] = (t0_ < n && f < lim) ? P[(int64_t)t0_ * 3 + f] : 0.f; }
    };
    fetch(0);
    for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
        const int tc = min(KNN_TILE, n - t0);
#pragma unroll
        for (int u = 0; u < NF; ++u) { const int f = tid + u * 256; if (f < tc * 3) tile[f] = stage[u]; }
        __syncthreads();
        fetch(t0 + KNN_TILE);
        int i = 0;
        for (; i + 4 <= tc; i += 4) {
            const float4 c0 = *reinterpret_cast<const float4*>(tile + 3 * i), c1 = *reinterpret_cast<const float4*>(tile + 3 * i + 4),
                         c2 = *reinterpret_cast<const float4*>(tile + 3 * i + 8);
            const float ax = qx - c0.x, ay = qy - c0.y, az = qz - c0.z;
            const float bx = qx - c0.w, by = qy - c1.x, bz = qz - c1.y;
            const float ex = qx - c1.z, ey = qy - c1.w, ez = qz - c2.x;
            const float fx = qx - c2.y, fy = qy - c2.z, fz = qz - c2.w;
            const float d0 = (ax * ax + ay * ay) + az * az, d1 = (bx * bx + by * by) + bz * bz;
            const float d2 = (ex * ex + ey * ey) + ez * ez, d3 = (fx * fx + fy * fy) + fz * fz;
            offer(d0, t0 + i); offer(d1, t0 + i + 1); offer(d2, t0 + i + 2); offer(d3, t0 + i + 3);
            if (__any(cnt > BUF - 4)) flush();                       // room for the next four appends in every lane
        }
        for (; i < tc; ++i) {
            const float dx = qx - tile[3 * i], dy = qy - tile[3 * i + 1], dz = qz - tile[3 * i + 2];
            offer((dx * dx + dy * dy) + dz * dz, t0 + i);
            if (__any(cnt > BUF - 4)) flush();
        }
        __syncthreads();
    }
    flush();
    if (valid) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j > 0 && bi[j] < 0) { bi[j] = bi[j - 1]; bd[j] = bd[j - 1]; }     // fewer than K points: repeat the last
            idx_out[((int64_t)b * m + q) * K + j] = b * n + bi[j];
            d2_out[((int64_t)b * m + q) * K + j] = bd[j];
        }
    }
}

// ---------------------------------------------------------------- kNN with an EXACT spatial pruning (round 6)
// knn_kernel above evaluates every (query, candidate) pair: 2048 x 8192 x 32 samples = 537 M distances for configs[3]'s neighbour lists (0.59 ms),
// 2.1 G for the level-1 lists (0.82 ms).  Here the candidates of a sample are sorted along a Morton curve once (knn_sort_kernel: bitonic sort in LDS, as
// fps_pruned_kernel does) and cut into TILES of 64 consecutive sorted points with a bounding box each; the queries are sorted the same way, so the 64
// queries of a wave are neighbours in space.  A wave first scans the tiles whose box meets the box of its own queries, then walks the remaining
// tiles and scans one only if, for some lane, the squared distance from the query to the tile's box - computed with the SAME float operations and
// association as the point distances (rounding is monotonic: a lower bound of every computed distance in the tile) - is not above the lane's current
// k-th distance.  Visiting order no longer is index order, so the k best are kept as 64-bit keys (distance bits << 32 | original index): unsigned
// order = (d2, index) lexicographic order (distances are >= +0), the order the plain kernel's strict '<' on ascending indices produces.  Same
// distances bit for bit, same neighbours in the same order - tests/test_gpu_points.py compares both kernels with the oracle and with each other.
constexpr int KNN_TS = 64;                              // candidates per tile: one wave-wide load

// grid (B, 2): y = 0 sorts the candidates (and writes the tile boxes), y = 1 the queries (skipped when they are the candidates).  out4[i] = (x, y, z, bits of the
// original index) of the i-th point along the curve.  Dynamic LDS: 2 x cap words (cap = n rounded up to a power of two, <= 8192).
__global__ __launch_bounds__(1024) void knn_sort_kernel(const float* __restrict__ xyz, int n, const float* __restrict__ qxyz, int m, int cap_c, int cap_q,
                                                        float4* __restrict__ cs4, float* __restrict__ tbox, float4* __restrict__ qs4) {
    constexpr int T = 1024, nw = T / 64;
    __shared__ float bb[6][16];
    extern __shared__ unsigned knn_sort_lds[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool isq = blockIdx.y == 1;
    if (isq && !qs4) return;
    const int cnt = isq ? m : n, CAP = isq ? cap_q : cap_c;
    const float* P = (isq ? qxyz : xyz) + (int64_t)b * cnt * 3;
    float4* out4 = (isq ? qs4 : cs4) + (int64_t)b * cnt;
    float lo3[3] = {INFINITY, INFINITY, INFINITY}, hi3[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < cnt; i += T) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = P[i * 3 + a]; lo3[a] = fminf(lo3[a], v); hi3[a] = fmaxf(hi3[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo3[a] = fminf(lo3[a], __shfl_xor(lo3[a], o)); hi3[a] = fmaxf(hi3[a], __shfl_xor(hi3[a], o)); }
        if (lane == 0) { bb[a][wave] = lo3[a]; bb[3 + a][wave] = hi3[a]; }
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = bb[a][0], h = bb[3 + a][0];
        for (int w2 = 1; w2 < nw; ++w2) { l = fminf(l, bb[a][w2]); h = fmaxf(h, bb[3 + a][w2]); }
        lo3[a] = l; hi3[a] = 1023.0f / fmaxf(h - l, 1e-12f);          // hi3 now holds the quantisation scale (any order gives the same neighbours)
    }
    unsigned* kk = knn_sort_lds;
    unsigned* vv = kk + CAP;
    for (int i = tid; i < CAP; i += T) {
        unsigned key = 0xFFFFFFFFu;
        if (i < cnt) {
            unsigned q[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float f = (P[i * 3 + a] - lo3[a]) * hi3[a]; q[a] = (unsigned)min(1023, max(0, (int)f)); }
            key = morton_part10(q[0]) | (morton_part10(q[1]) << 1) | (morton_part10(q[2]) << 2);
        }
        kk[i] = key; vv[i] = i < cnt ? (unsigned)i : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int k = 2; k <= CAP; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int i = tid; i < CAP; i += T) {
                const int x = i ^ jj;
                if (x > i) {
                    const unsigned ka = kk[i], kb = kk[x];
                    if ((ka > kb) == ((i & k) == 0)) { kk[i] = kb; kk[x] = ka; const unsigned va = vv[i]; vv[i] = vv[x]; vv[x] = va; }
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < cnt; i += T) {
        const unsigned o = vv[i];
        out4[i] = make_float4(P[o * 3 + 0], P[o * 3 + 1], P[o * 3 + 2], __uint_as_float(o));
    }
    if (!isq) {                                          // boxes of the tiles of 64 consecutive sorted candidates: one wave per tile
        const int nt = (cnt + KNN_TS - 1) / KNN_TS;
        for (int t = wave; t < nt; t += nw) {
            const int i = t * KNN_TS + lane;
            const bool ok = i < cnt;
            const unsigned o = ok ? vv[i] : 0u;
            float l[3], h[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { const float v = P[o * 3 + a]; l[a] = ok ? v : INFINITY; h[a] = ok ? v : -INFINITY; }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int o2 = 32; o2 > 0; o2 >>= 1) { l[a] = fminf(l[a], __shfl_xor(l[a], o2)); h[a] = fmaxf(h[a], __shfl_xor(h[a], o2)); }
            }
            if (lane == 0) {
                float* tb = tbox + ((int64_t)b * nt + t) * 6;
                tb[0] = l[0]; tb[1] = l[1]; tb[2] = l[2]; tb[3] = h[0]; tb[4] = h[1]; tb[5] = h[2];
            }
        }
    }
}

// A wave = 16 queries x 4 candidate SUB-SLOTS (lane = query + 16 * sub-slot): the sixteen queries are consecutive along the curve (a box sixteen
// points wide prunes far better than one 64 points wide: the first form of this kernel, 64 queries per wave, scanned 35-46 of 128 tiles and was slower than
// the plain kernel), every lane evaluates a quarter of a scanned tile's 64 candidates and keeps its own K best; the bound a lane prunes with is the
// smallest k-th key of its query's four lists (K candidates are known to lie at or below each of them).  At the end the four sorted lists of a query
// are merged by rank: a key's position in the union is its position in its own list plus the number of smaller keys in the other three.
template <int K>
__global__ __launch_bounds__(256) void knn_pruned_kernel(const float4* __restrict__ cs4, const float* __restrict__ tbox, const float4* __restrict__ qs4, int n, int m,
                                                         int nt, int* __restrict__ idx_out, float* __restrict__ d2_out) {
    constexpr int BUF = 8;
    constexpr unsigned long long EMPTY = ((unsigned long long)0x7F800000u << 32) | 0xFFFFFFFFull;      // (+inf, no index): nothing real is above it
    __shared__ float4 tile[4][KNN_TS];
    __shared__ unsigned long long buf[4][BUF + 1][64];                  // [wave][entry][lane], row BUF = dump row
    __shared__ unsigned long long mk[4][K][64];                         // the sorted lists at the end: [wave][position][lane]
    const int b = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, qi = lane & 15, sub = lane >> 4;
    const int qs = (blockIdx.x * 4 + wave) * 16 + qi;
    const bool valid = qs < m;
    const float4 Q = qs4[(int64_t)b * m + (valid ? qs : m - 1)];        // slots past the end repeat the last query (a neighbour in space: prunes like it)
    const float qx = Q.x, qy = Q.y, qz = Q.z;
    const int qorig = (int)__float_as_uint(Q.w);
    unsigned long long bk[K];
#pragma unroll
    for (int j = 0; j < K; ++j) bk[j] = EMPTY;
    int cnt = 0;
    auto flush = [&]() {
        for (int e = 0; e < BUF; ++e) {
            if (!__any(e < cnt)) break;                              // wave-uniform
            const unsigned long long key = e < cnt ? buf[wave][e][lane] : ~0ull;
#pragma unroll
            for (int j = K - 1; j > 0; --j) {                        // sorted insert on (distance bits, index): keys are unique
                const bool up = key < bk[j - 1];
                const bool here = key < bk[j];
                bk[j] = up ? bk[j - 1] : (here ? key : bk[j]);
            }
            bk[0] = key < bk[0] ? key : bk[0];
        }
        cnt = 0;
    };
    auto offer = [&](float d, unsigned idbits) {                     // append if it beats the (possibly stale) k-th key: re-checked at the drain
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | idbits;
        const bool take = key < bk[K - 1];
        buf[wave][take ? cnt : BUF][lane] = key;
        cnt += take ? 1 : 0;
    };
    const float4* C = cs4 + (int64_t)b * n;
    const float* TB = tbox + (int64_t)b * nt * 6;
    auto scan_tile = [&](int t) {
        const int i0 = t * KNN_TS;
        // candidate j of the tile goes to slot (j & 15) * 4 + (j >> 4): the four sub-slots read four ADJACENT float4 per step (one 64-byte broadcast read)
        tile[wave][qi * 4 + sub] = i0 + lane < n ? C[i0 + lane] : make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0xFFFFFFFFu));
        __builtin_amdgcn_wave_barrier();                             // (LDS operations of one wave complete in order: the reads below see the tile)
#pragma unroll
        for (int i = 0; i < 16; i += 4) {
            const float4 c0 = tile[wave][i * 4 + sub], c1 = tile[wave][(i + 1) * 4 + sub], c2 = tile[wave][(i + 2) * 4 + sub], c3 = tile[wave][(i + 3) * 4 + sub];
            const float ax = qx - c0.x, ay = qy - c0.y, az = qz - c0.z;
            const float bx = qx - c1.x, by = qy - c1.y, bz = qz - c1.z;
            const float ex = qx - c2.x, ey = qy - c2.y, ez = qz - c2.z;
            const float fx = qx - c3.x, fy = qy - c3.y, fz = qz - c3.z;
            const float d0 = (ax * ax + ay * ay) + az * az, d1 = (bx * bx + by * by) + bz * bz;
            const float d2 = (ex * ex + ey * ey) + ez * ez, d3 = (fx * fx + fy * fy) + fz * fz;
            offer(d0, __float_as_uint(c0.w)); offer(d1, __float_as_uint(c1.w)); offer(d2, __float_as_uint(c2.w)); offer(d3, __float_as_uint(c3.w));
            if (__any(cnt > BUF - 4)) flush();
        }
        __builtin_amdgcn_wave_barrier();                             // every lane has read the tile before the next one overwrites it
    };
    // ---- box of this wave's sixteen queries (every sub-slot holds the same sixteen)
    float wl[3] = {qx, qy, qz}, wh[3] = {qx, qy, qz};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        wl[a] = fminf(wl[a], lane_xor<1>(wl[a])); wl[a] = fminf(wl[a], lane_xor<2>(wl[a])); wl[a] = fminf(wl[a], lane_xor<4>(wl[a])); wl[a] = fminf(wl[a], lane_xor<8>(wl[a]));
        wh[a] = fmaxf(wh[a], lane_xor<1>(wh[a])); wh[a] = fmaxf(wh[a], lane_xor<2>(wh[a])); wh[a] = fmaxf(wh[a], lane_xor<4>(wh[a])); wh[a] = fmaxf(wh[a], lane_xor<8>(wh[a]));
    }
    // ---- pass 1: the tiles whose box meets the wave's box (its own neighbourhood: fills the k best with near points)
    unsigned long long seen[2] = {0ull, 0ull};                       // nt <= 128
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int t = h * 64 + lane;
        bool hit = false;
        if (t < nt) {
            const float* tb = TB + t * 6;
            hit = tb[0] <= wh[0] && tb[3] >= wl[0] && tb[1] <= wh[1] && tb[4] >= wl[1] && tb[2] <= wh[2] && tb[5] >= wl[2];
        }
        seen[h] = __builtin_amdgcn_ballot_w64(hit);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned long long mkk = seen[h];
        while (mkk) {                                                // wave-uniform
            const int t = h * 64 + __builtin_ctzll(mkk);
            mkk &= mkk - 1;
            scan_tile(t);
        }
    }
    flush();
    // the bound of a query: the smallest k-th distance of its four lists (refreshed after every scan; stale = only ever too large)
    auto bound = [&]() {
        float kd = __uint_as_float((unsigned)(bk[K - 1] >> 32));
        kd = fminf(kd, xor16(kd));
        return fminf(kd, xor32(kd));
    };
    float kd = bound();
    // ---- pass 2: every other tile, scanned only if some lane's lower bound is not above its query's bound
    for (int t = 0; t < nt; ++t) {
        if ((seen[t >> 6] >> (t & 63)) & 1ull) continue;             // wave-uniform
        const float* tb = TB + t * 6;                                // uniform address: scalar loads
        const float ex = fmaxf(fmaxf(tb[0] - qx, qx - tb[3]), 0.f), ey = fmaxf(fmaxf(tb[1] - qy, qy - tb[4]), 0.f), ez = fmaxf(fmaxf(tb[2] - qz, qz - tb[5]), 0.f);
        const float lb = (ex * ex + ey * ey) + ez * ez;
        if (__builtin_amdgcn_ballot_w64(lb <= kd) != 0ull) {
            scan_tile(t);
            flush();
            kd = bound();
        }
    }
    // ---- merge the four lists of every query by rank
#pragma unroll
    for (int j = 0; j < K; ++j) mk[wave][j][lane] = bk[j];
    __builtin_amdgcn_wave_barrier();
    int rank[K];
#pragma unroll
    for (int j = 0; j < K; ++j) rank[j] = j;
#pragma unroll
    for (int o = 1; o < 4; ++o) {
        const int other = qi + 16 * ((sub + o) & 3);
        unsigned long long ok[K];
#pragma unroll
        for (int i = 0; i < K; ++i) ok[i] = mk[wave][i][other];
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int i = 0; i < K; ++i) rank[j] += ok[i] < bk[j] ? 1 : 0;
    }
    if (valid) {
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (rank[j] < K && (unsigned)bk[j] != 0xFFFFFFFFu) {         // (n >= 1024 > K: the K best of the union are real points)
                idx_out[((int64_t)b * m + qorig) * K + rank[j]] = b * n + (int)(unsigned)bk[j];
                d2_out[((int64_t)b * m + qorig) * K + rank[j]] = __uint_as_float((unsigned)(bk[j] >> 32));
            }
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                          float* __restrict__ out, int64_t rows, int c) {
    const int64_t total = rows * c;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = e / c;
        out[e] = src[(int64_t)idx[r] * c + (e - r * c)];
    }
}

// out[i,:] = base[i,:] + sum_j w_ij * feat[idx[i,j],:],  w_ij = (1 / (sqrt(d2_ij) + 1e-8)) / sum_j (...)   (k neighbours)
__global__ __launch_bounds__(256) void interpolate_kernel(const float* __restrict__ feat, const int* __restrict__ idx,
                                                          const float* __restrict__ d2, const float* __restrict__ base,
                                                          float* __restrict__ out, int64_t n, int c, int k) {
    const int c4n = c >> 2;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * c4n; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / c4n;
        const int cc = (int)(e - i * c4n) * 4;
        float w[8], norm = 0.f;
        for (int j = 0; j < k; ++j) { w[j] = 1.0f / (sqrtf(d2[i * k + j]) + 1e-8f); norm += w[j]; }
        float4 acc = base ? *reinterpret_cast<const float4*>(base + i * c + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 up = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < k; ++j) {
            const float4 f = *reinterpret_cast<const float4*>(feat + (int64_t)idx[i * k + j] * c + cc);
            const float wj = w[j] / norm;
            up.x += f.x * wj; up.y += f.y * wj; up.z += f.z * wj; up.w += f.w * wj;
        }
        *reinterpret_cast<float4*>(out + i * c + cc) = make_float4(acc.x + up.x, acc.y + up.y, acc.z + up.z, acc.w + up.w);
    }
}

// backward of interpolate_kernel with respect to feat: dfeat[idx[i,j], :] += w_ij * dout[i, :] (f32 atomics: a coarse point is the neighbour
// of many fine points - the reference's CUDA interpolation backward does the same); the gradient of `base` is dout itself.
__global__ __launch_bounds__(256) void interpolate_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ idx, const float* __restrict__ d2,
                                                              float* __restrict__ dfeat, int64_t n, int c, int k) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * c; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / c;
        const int ch = (int)(e - i * c);
        float w[8], norm = 0.f;
        for (int j = 0; j < k; ++j) { w[j] = 1.0f / (sqrtf(d2[i * k + j]) + 1e-8f); norm += w[j]; }
        const float g = dout[e];
        for (int j = 0; j < k; ++j) atomicAdd(dfeat + (int64_t)idx[i * k + j] * c + ch, g * (w[j] / norm));
    }
}

// out[b,:] = mean over the n rows of sample b  (grid B, block 256; c <= 4096)
__global__ __launch_bounds__(256) void segment_mean_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int c) {
    const int b = blockIdx.x;
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += x[((int64_t)b * n + i) * c + ch];
        out[(int64_t)b * c + ch] = s / (float)n;
    }
}

}  // namespace

extern "C" int afm_interpolate(const float* feat, const int32_t* idx, const float* dist2, const float* base, float* out,
                               int64_t n, int32_t c, int32_t k, void* stream) {
    if (c <= 0 || (c & 3) || k <= 0 || k > 8 || n < 0) return AFM_E_BADARG;
    if (n == 0) return 0;
    if (!feat || !idx || !dist2 || !out) return AFM_E_BADARG;
    const int64_t total = n * (c >> 2);
    unsigned g = (unsigned)((total + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(interpolate_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, feat, idx, dist2, base, out, n, c, k);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_interpolate_bwd(const float* dout, const int32_t* idx, const float* dist2, float* dfeat, int64_t n, int64_t m, int32_t c,
                                   int32_t k, void* stream) {
    if (c <= 0 || k <= 0 || k > 8 || n < 0 || m < 0) return AFM_E_BADARG;
    if (m == 0) return 0;
    if (!dfeat) return AFM_E_BADARG;
    hipError_t e = hipMemsetAsync(dfeat, 0, (size_t)m * c * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return 0;
    if (!dout || !idx || !dist2) return AFM_E_BADARG;
    const int64_t total = n * c;
    unsigned g = (unsigned)((total + 255) / 256);
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(interpolate_bwd_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, dout, idx, dist2, dfeat, n, c, k);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_segment_mean(const float* x, float* out, int32_t B, int32_t n, int32_t c, void* stream) {
    if (B < 0 || n <= 0 || c <= 0) return AFM_E_BADARG;
    if (B == 0) return 0;
    if (!x || !out) return AFM_E_BADARG;
    hipLaunchKernelGGL(segment_mean_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, out, n, c);
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_fps(const float* xyz, int32_t B, int32_t n, int32_t m, int32_t* idx_out, void* stream) {
    if (!xyz || !idx_out || B < 0 || n <= 0 || m < 0 || m > n) return AFM_E_BADARG;
    if (B == 0 || m == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    {   // exact pruning of the scan (fps_pruned_kernel): PT threads, np2 / PT Morton-consecutive points per thread
        int np2 = 2048;
        while (np2 < n) np2 <<= 1;
        bool prune = n >= 1536 && np2 <= 8192 && m >= 32;
        int PT = 1024;
#ifdef AFM_PROBE
        if (afm_probe_fps_prune == 0) prune = false;
        if (afm_probe_fps_prune > 0) PT = afm_probe_fps_prune;
#endif
        if (prune && AFM_FPS_KS == 1 && AFM_FPS_CELLS > 1 && PT == 1024) {
            // round 6: CELLS cells of 8 sorted points per thread on 1024 / CELLS threads (same cells, same pruning, fewer waves to pay the per-wave overheads)
            AfmProf prof(AFM_PROF_FPS, (double)B * (m - 1) * n, s);
            const size_t lds = (size_t)max(3 * n, 2 * np2) * sizeof(float);
            constexpr int TC = 1024 / AFM_FPS_CELLS;
#define AFM_FPS_CELLS_GO(P)                                                                                                                 \
    do {                                                                                                                            \
        if (lds > 48 * 1024) {                                                                                                      \
            hipError_t e__ = hipFuncSetAttribute((const void*)fps_cells_kernel<P, TC, AFM_FPS_CELLS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
            if (e__ != hipSuccess) return (int)e__;                                                                                 \
        }                                                                                                                           \
        hipLaunchKernelGGL((fps_cells_kernel<P, TC, AFM_FPS_CELLS>), dim3(B), dim3(TC), lds, s, xyz, n, m, idx_out);                \
    } while (0)
            if (np2 == 2048) AFM_FPS_CELLS_GO(2);
            else if (np2 == 4096) AFM_FPS_CELLS_GO(4);
            else AFM_FPS_CELLS_GO(8);
#undef AFM_FPS_CELLS_GO
            AFM_CHECK_LAUNCH();
            return 0;
        }
        if (prune) {
            AfmProf prof(AFM_PROF_FPS, (double)B * (m - 1) * n, s);
            const size_t lds = (size_t)max(3 * n, 2 * np2) * sizeof(float);
#define AFM_FPS_PRUNED(P, PT_)                                                                                                          \
    do {                                                                                                                            \
        if (lds > 48 * 1024) {                                                                                                      \
            hipError_t e__ = hipFuncSetAttribute((const void*)fps_pruned_kernel<P, PT_, AFM_FPS_KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
            if (e__ != hipSuccess) return (int)e__;                                                                                 \
        }                                                                                                                           \
        hipLaunchKernelGGL((fps_pruned_kernel<P, PT_, AFM_FPS_KS>), dim3(B), dim3(PT_), lds, s, xyz, n, m, idx_out);                \
    } while (0)
#define AFM_FPS_PRUNED_T(PT_)                                                                                                           \
    do {                                                                                                                            \
        if (np2 == 2048) AFM_FPS_PRUNED(2048 / PT_, PT_);                                                                           \
        else if (np2 == 4096) AFM_FPS_PRUNED(4096 / PT_, PT_);                                                                      \
        else AFM_FPS_PRUNED(8192 / PT_, PT_);                                                                                       \
    } while (0)
            if (PT == 256) AFM_FPS_PRUNED_T(256);
            else if (PT == 512) AFM_FPS_PRUNED_T(512);
            else AFM_FPS_PRUNED_T(1024);
#undef AFM_FPS_PRUNED_T
#undef AFM_FPS_PRUNED
            AFM_CHECK_LAUNCH();
            return 0;
        }
    }
    // Threads per workgroup: a round is a dependent chain (per-thread scan -> wave arg-max -> LDS hop -> barrier -> broadcast) on ONE
    // compute unit per sample.  Its VALU work (n points x 8 instructions: ~1100 issue cycles of the ~2100-cycle round at n = 8192) is
    // the same for every split, the reductions are not: 0.94 / 0.96 / 0.91 us per round with 1024 / 512 / 256 threads
    // (profiles/r02_points_probe.txt).  Results do not depend on the split.
    int T = (((n + 31) / 32 + 63) / 64) * 64;
    if (T < 64) T = 64;
    if (T > 512) T = 512;
    if ((n + T - 1) / T > 32) T = 1024;              // n > 16384: the round-1 shape (up to 16 points per thread)
#ifdef AFM_PROBE               // tools/points_probe.hip only: threads per workgroup override for experiments
    if (afm_probe_fps_threads > 0) T = afm_probe_fps_threads;
#endif
    const int ppt = (n + T - 1) / T;
    AfmProf prof(AFM_PROF_FPS, (double)B * (m - 1) * n, s);
    const size_t lds = (size_t)3 * n * sizeof(float);
    const int in_lds = lds <= 150 * 1024 ? 1 : 0;
#define AFM_FPS(P, MT)                                                                                                              \
    do {                                                                                                                            \
        if (in_lds) {                                                                                                               \
            if (lds > 48 * 1024) {                                                                                                  \
                hipError_t e__ = hipFuncSetAttribute((const void*)fps_kernel<P, MT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
                if (e__ != hipSuccess) return (int)e__;                                                                             \
            }                                                                                                                       \
            hipLaunchKernelGGL((fps_kernel<P, MT, true>), dim3(B), dim3(T), lds, s, xyz, n, m, idx_out);                            \
        } else {                                                                                                                    \
            hipLaunchKernelGGL((fps_kernel<P, MT, false>), dim3(B), dim3(T), 0, s, xyz, n, m, idx_out);                             \
        }                                                                                                                           \
    } while (0)
    if (T <= 512 && ppt > 16 && ppt <= 32) AFM_FPS(32, 512);
    else if (ppt <= 1) AFM_FPS(1, 1024);
    else if (ppt <= 2) AFM_FPS(2, 1024);
    else if (ppt <= 4) AFM_FPS(4, 1024);
    else if (ppt <= 8) AFM_FPS(8, 1024);
    else if (ppt <= 16) AFM_FPS(16, 1024);
    else return AFM_E_UNSUPPORTED;
#undef AFM_FPS
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_knn(int32_t k, const float* xyz, const float* new_xyz, int32_t B, int32_t n, int32_t m, int32_t* idx_out,
                       float* dist2_out, void* stream) {
    if (!xyz || !new_xyz || !idx_out || !dist2_out || B < 0 || n <= 0 || m < 0) return AFM_E_BADARG;
    if (B == 0 || m == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((m + 255) / 256, B), block(256);
    AfmProf prof(AFM_PROF_KNN, (double)B * m * n, s);
    switch (k) {
        case 3: hipLaunchKernelGGL(knn_kernel<3>, grid, block, 0, s, xyz, new_xyz, n, m, idx_out, dist2_out); break;
        case 8: hipLaunchKernelGGL(knn_kernel<8>, grid, block, 0, s, xyz, new_xyz, n, m, idx_out, dist2_out); break;
        case 16: hipLaunchKernelGGL(knn_kernel<16>, grid, block, 0, s, xyz, new_xyz, n, m, idx_out, dist2_out); break;
        default: return AFM_E_UNSUPPORTED;
    }
    AFM_CHECK_LAUNCH();
    return 0;
}

// Workspace of afm_knn_ws: sorted candidates (float4) + tile boxes + sorted queries (float4); 0 = the pruned form does not apply (plain kernel, no workspace)
// Measured (round 6, B = 32, one call; pruned vs plain): self-search 8192 x 8192, k = 8: 0.604 vs 0.827 ms - the only shape of this path where the pruned form wins;
// 8192 -> 2048, k = 16: 0.858 vs 0.586; self 2048, k = 16: 0.395 vs 0.181; 2048 <- 8192 queries, k = 3: 0.284 vs 0.196 (sort + box walk + key bookkeeping cost more
// than the ~4-6 x fewer pair evaluations save: 8192 points in a room are too few for tiles of 64 to be small against the k-NN radius).  So: large self-searches only.
static bool knn_pruned_applies(int k, int n, int m) { return (k == 3 || k == 8 || k == 16) && n >= 4096 && n <= 8192 && m >= n && m <= 8192; }
extern "C" int64_t afm_knn_workspace_bytes(int32_t k, int32_t B, int32_t n, int32_t m) {
    if (B < 0 || n <= 0 || m < 0) return AFM_E_BADARG;
    if (!knn_pruned_applies(k, n, m)) return 0;
    const int64_t nt = (n + KNN_TS - 1) / KNN_TS;
    return (int64_t)B * n * 16 + ((int64_t)B * nt * 6 * 4 + 255) / 256 * 256 + (int64_t)B * m * 16;
}

extern "C" int afm_knn_ws(int32_t k, const float* xyz, const float* new_xyz, int32_t B, int32_t n, int32_t m, int32_t* idx_out, float* dist2_out,
                          void* workspace, int64_t workspace_bytes, void* stream) {
    if (!xyz || !new_xyz || !idx_out || !dist2_out || B < 0 || n <= 0 || m < 0) return AFM_E_BADARG;
    if (B == 0 || m == 0) return 0;
    if (!knn_pruned_applies(k, n, m) || !workspace) return afm_knn(k, xyz, new_xyz, B, n, m, idx_out, dist2_out, stream);
    if (workspace_bytes < afm_knn_workspace_bytes(k, B, n, m) || ((uintptr_t)workspace & 15)) return AFM_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nt = (n + KNN_TS - 1) / KNN_TS;
    char* wp = (char*)workspace;
    float4* cs4 = (float4*)wp; wp += (int64_t)B * n * 16;
    float* tbox = (float*)wp; wp += ((int64_t)B * nt * 6 * 4 + 255) / 256 * 256;
    const bool self = new_xyz == xyz && m == n;
    float4* qs4 = self ? nullptr : (float4*)wp;
    int cap_c = 1024, cap_q = 1024;
    while (cap_c < n) cap_c <<= 1;
    while (cap_q < m) cap_q <<= 1;
    const size_t lds = (size_t)2 * (cap_c > cap_q ? cap_c : cap_q) * sizeof(unsigned);
    AfmProf prof(AFM_PROF_KNN, (double)B * m * n, s);
    if (lds > 48 * 1024) {
        static const int attr = (int)hipFuncSetAttribute((const void*)knn_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8192 * 4);
        if (attr != 0) return attr;
    }
    hipLaunchKernelGGL(knn_sort_kernel, dim3(B, self ? 1 : 2), dim3(1024), lds, s, xyz, n, new_xyz, m, cap_c, cap_q, cs4, tbox, qs4);
    const dim3 grid((m + 63) / 64, B), block(256);              // 16 queries per wave, 4 waves per workgroup
    const float4* q4 = self ? cs4 : qs4;
    switch (k) {
        case 3: hipLaunchKernelGGL(knn_pruned_kernel<3>, grid, block, 0, s, cs4, tbox, q4, n, m, nt, idx_out, dist2_out); break;
        case 8: hipLaunchKernelGGL(knn_pruned_kernel<8>, grid, block, 0, s, cs4, tbox, q4, n, m, nt, idx_out, dist2_out); break;
        default: hipLaunchKernelGGL(knn_pruned_kernel<16>, grid, block, 0, s, cs4, tbox, q4, n, m, nt, idx_out, dist2_out); break;
    }
    AFM_CHECK_LAUNCH();
    return 0;
}

extern "C" int afm_gather_rows(const float* src, const int32_t* idx, float* out, int64_t rows, int32_t c, void* stream) {
    if (!src || !idx || !out || rows < 0 || c <= 0) return AFM_E_BADARG;
    if (rows == 0) return 0;
    const int64_t total = rows * c;
    unsigned g = (unsigned)((total + 255) / 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src, idx, out, rows, c);
    AFM_CHECK_LAUNCH();
    return 0;
}
