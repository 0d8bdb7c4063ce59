// out[i][tap, c] += sum_m A[amap(m)][i] * B[bmap(m, tap)][c]       (weight gradients)
//
// Both operands are reduced over their ROW index, i.e. both MFMA fragments need 8 values
// that are strided in memory.  Tiles of 32 rows are staged row-major in LDS (coalesced
// 16-B global loads) and the fragments are gathered with gfx950's LDS transpose read
// (ds_read_b64_tr_b16: a 16-lane group reads a 4x16 bf16 block and every lane receives one
// COLUMN of it), two reads per fragment with k-slot (g,e) <-> row 4g+e / 16+4g+(e-4).
// fp32 (parity mode) gathers with scalar ds_read_b32; `use_tr = 0` keeps a scalar
// ds_read_u16 gather for bf16 as the cross-check of the transpose-read path.
//
// The output is tiny (e.g. 96 x 288) and the reduction long (50k-100k rows), so the row
// range is split across blocks and partial tiles are combined with fp32 atomics
// (hardware global_atomic_add_f32, compiled with -munsafe-fp-atomics) straight into the
// flat gradient buffer, in the reference's parameter layout (so_* strides).
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NT = 256;
constexpr int PADT = 16;  // row stride = 8 banks mod 64: the 8 rows of a half-wave transpose read land on distinct banks
constexpr int BR = 64;   // rows (reduction) per step = two 32-row MFMA k-blocks

// destination of output element (i, c) of tap `tap`: the reference parameter layout, or the second tensor past split_c
DEV float* tn_dst(const dist_gemm_tn_args& p, int ii, int c, int tap) {
    if (p.out2 && c >= p.split_c) return p.out2 + (long)ii * p.so_i2 + (c - p.split_c);
    return p.out + (long)ii * p.so_i + (long)tap * p.so_tap + (long)(c / p.inner) * p.so_outer + (c % p.inner);
}

template <typename T, bool TR>
DEV void gather_frag(Frag<T>& f, const T* tile, int ld, int col0, int li, int lg);

template <>
DEV void gather_frag<float, false>(Frag<float>& f, const float* tile, int ld, int col0, int li, int lg) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = (e < 4) ? 4 * lg + e : 16 + 4 * lg + (e - 4);
        f.v[e] = tile[r * ld + col0 + li];
    }
}
template <>
DEV void gather_frag<float, true>(Frag<float>& f, const float* tile, int ld, int col0, int li, int lg) {
    gather_frag<float, false>(f, tile, ld, col0, li, lg);
}
template <>
DEV void gather_frag<bf16_t, false>(Frag<bf16_t>& f, const bf16_t* tile, int ld, int col0, int li, int lg) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int r = (e < 4) ? 4 * lg + e : 16 + 4 * lg + (e - 4);
        f.v[e] = tile[r * ld + col0 + li];
    }
}
template <>
DEV void gather_frag<bf16_t, true>(Frag<bf16_t>& f, const bf16_t* tile, int ld, int col0, int li, int lg) {
    // lane li of the 16-lane group supplies the address of 4 contiguous bf16 of block row li>>2;
    // it receives column li of the 4x16 block (rows 4g..4g+3, then 16+4g..16+4g+3).
    typedef s16x4 __attribute__((address_space(3))) * lds_v4;
    const bf16_t* p0 = tile + (4 * lg + (li >> 2)) * ld + col0 + (li & 3) * 4;
    const bf16_t* p1 = p0 + 16 * ld;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p1));
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = a; u.s[1] = b;
    f.v = u.v;
}

// MODES: the row maps of A and B as compile-time constants, amap.mode * 8 + bmap.mode (0 = both plain), or -1 = read them from the
// arguments.  With runtime modes hipcc turns the small per-mode switches of rowmap_src2 / rowmap_step into select chains - every
// mode evaluated for every cell in every step: the incremental loader was 20-25 % SLOWER than re-dividing per load until the
// combinations the engine uses were instantiated with constant modes (profiles/r02_tn_incremental_rowmaps.md).
template <typename T, int BI, int BJ, int WI, int WJ, bool TR, int MODES>
__global__ __launch_bounds__(WI * WJ * 64) void gemm_tn_kernel(const dist_gemm_tn_args p, int chunk, int tiles_i, int tiles_c) {
    constexpr bool PLAIN = MODES == 0;
    dist_rowmap amap = p.amap, bmap = p.bmap;
    if constexpr (MODES >= 0) { amap.mode = MODES / 8; bmap.mode = MODES % 8; }
    constexpr int NT = WI * WJ * 64;             // 4, 6 or 8 waves (shadows the namespace constant used by tn_reduce_kernel)
    constexpr int LDI = BI + PADT, LDJ = BJ + PADT;
    constexpr int WTI = BI / WI, WTJ = BJ / WJ;
    constexpr int FI = WTI / 16, FJ = WTJ / 16;
    constexpr int VI = BI / 8, VJ = BJ / 8;
    constexpr int I_IT = (BR * VI + NT - 1) / NT, J_IT = (BR * VJ + NT - 1) / NT;
    static_assert(WTI % 16 == 0 && WTJ % 16 == 0, "whole 16x16 fragments per wave");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ys = reinterpret_cast<T*>(smem);          // [2][BR][LDI]
    T* Xs = Ys + 2 * BR * LDI;                   // [2][BR][LDJ]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = wid / WJ, wj = wid % WJ;
    const int li = lane & 15, lg = lane >> 4;

    // XCD-aware, bijective remap: the output tiles (and taps) of ONE row chunk get consecutive ids on the same XCD,
    // so the chunk's rows of dY and X are fetched from HBM once and re-read by the other tiles from that XCD's L2
    const int tiles_ij = tiles_i * tiles_c * p.taps;
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q = nblk / 8, r = nblk % 8, x = bid % 8, y = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int ms = bid / tiles_ij;
    int t = bid % tiles_ij;
    const int ti = t % tiles_i; t /= tiles_i;
    const int tc = t % tiles_c; const int tap = t / tiles_c;
    const int i0 = ti * BI, c0 = tc * BJ;
    const int M = (int)p.M;
    const int mbeg = ms * chunk, mend = min(M, mbeg + chunk);
    if (mbeg >= mend) return;

    const T* __restrict__ A = static_cast<const T*>(p.A);
    const T* __restrict__ B = static_cast<const T*>(p.B);

    const int kvec_last = (p.K + 7) / 8 * 8 - 8;           // last 8-wide column vector that stays inside a row of B
    // fused bias gradient: the blocks of the first column tile also sum the columns of their A rows
    const bool do_colsum = p.colsum != nullptr && tc == 0 && tap == 0;
    float csum[I_IT][8];
#pragma unroll
    for (int i = 0; i < I_IT; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[i][e] = 0.f;

    // Two register sets: the global loads of step s+2 are issued as soon as set (s & 1) has been written to LDS, so two
    // 64-row tiles are in flight per block (the reduction is cold-HBM latency-bound: one tile of prefetch left the waves
    // parked on s_waitcnt for most of a step).
    // Loads are UNCONDITIONAL (addresses clamped into the matrices, validity kept as a flag and applied when the tile
    // is written to LDS): with a branch around each load hipcc loses its load counting and waits vmcnt(0) before the LDS
    // writes, which drains the set that was just requested and defeats the 2-deep prefetch.
    // (thread, i) cells past the tile (BR * VI not a multiple of the block size) load a clamped address and are never stored
    // The row maps (conv taps, cls-row skips) are NOT re-evaluated per load: every (thread, cell) walks rows that advance by exactly
    // BR per step, so its row decomposition is computed once (rowmap_prep: the modulo / divides) and stepped without dividing
    // (rowmap_step); the source row of the block's tap is a few adds and compares (rowmap_src2).  With rowmap_src per load the
    // row-mapped instantiations ran ~1 370 instructions per two steps against ~270 for the plain ones (24 MFMAs either way), and
    // with one 6-8 wave block per CU that serial instruction stream was the kernel's time (profiles/r02_tn_incremental_rowmaps.md).
    // A cell stops advancing when its next row would leave the chunk: its address stays valid, `ld_m` flags it as padding.
    struct Regs { Frag<T> a[I_IT], b[J_IT]; unsigned oka, okb; };
    int a_mm[I_IT], b_mm[J_IT];                    // clamped row (address side) of each cell
    RowPrep a_q[I_IT], b_q[J_IT];
    // (constant-mode instantiations are only launched when the maps can be stepped - the launcher checks: with `inc` a runtime
    // flag hipcc evaluated BOTH arms of `inc ? rowmap_step : rowmap_prep` every step, i.e. the modulo / divides the stepping replaced)
    const bool a_inc = PLAIN || MODES > 0 || rowmap_inc_ok(amap, BR), b_inc = PLAIN || MODES > 0 || rowmap_inc_ok(bmap, BR);
#pragma unroll
    for (int i = 0; i < I_IT; ++i) {
        const int v = min(tid + i * NT, BR * VI - 1);
        a_mm[i] = min(mbeg + v / VI, mend - 1);
        a_q[i] = PLAIN ? RowPrep{0, 0} : rowmap_prep(amap, a_mm[i]);
    }
#pragma unroll
    for (int i = 0; i < J_IT; ++i) {
        const int v = min(tid + i * NT, BR * VJ - 1);
        b_mm[i] = min(mbeg + v / VJ, mend - 1);
        b_q[i] = PLAIN ? RowPrep{0, 0} : rowmap_prep(bmap, b_mm[i]);
    }
    int ld_m = mbeg;                               // first row of the tile the next gload fetches (wave-uniform)
    // Plain maps, a tile inside both matrices and a whole number of step PAIRS: every cell of every step is valid (the launcher rounds
    // the row chunks to 2 * BR for these instantiations, so only the split that holds row M - 1 can differ).  The validity flags and
    // the zeroing of invalid cells - 56 v_cndmask + the flag arithmetic per two steps, in a loop that issues ~190 VALU for 32 MFMAs -
    // are skipped on a block-uniform branch.
    const bool all_ok = PLAIN && (mend - mbeg) % (2 * BR) == 0 && i0 + BI <= p.NI && c0 + BJ <= p.K && (BR * VI) % NT == 0 && (BR * VJ) % NT == 0;
    f32x4 acc[FI][FJ];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsteps = (mend - mbeg + BR - 1) / BR;
    // (a generic lambda instantiated for both values: as a runtime flag hipcc folds it into the selects instead of branching around them)
    auto run = [&](auto okc) __attribute__((always_inline)) {
    constexpr bool OK = decltype(okc)::value;
    auto gload = [&](Regs& R) __attribute__((always_inline)) {
        R.oka = 0; R.okb = 0;
#pragma unroll
        for (int i = 0; i < I_IT; ++i) {
            const int v = min(tid + i * NT, BR * VI - 1);
            const int m = ld_m + v / VI, col = i0 + (v % VI) * 8;
            const int cc = min(col, p.NI - 8);
            const int src = PLAIN ? a_mm[i] : rowmap_src2(amap, a_mm[i], a_q[i], 0, 1);
            if (!OK && m < mend && col < p.NI && src >= 0) R.oka |= 1u << i;
            frag_load(R.a[i], A + (long)max(src, 0) * p.lda + cc);
            if (m + BR < mend) {                                             // the cell's next row is still inside the chunk
                a_mm[i] = m + BR;
                if (!PLAIN) a_q[i] = a_inc ? rowmap_step(amap, a_q[i], BR) : rowmap_prep(amap, a_mm[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < J_IT; ++i) {
            const int v = min(tid + i * NT, BR * VJ - 1);
            const int m = ld_m + v / VJ, col = c0 + (v % VJ) * 8;
            const int cc = min(col, kvec_last);
            const int src = PLAIN ? b_mm[i] : rowmap_src2(bmap, b_mm[i], b_q[i], tap, p.taps);
            if (!OK && m < mend && col < p.K && src >= 0) R.okb |= 1u << i;
            frag_load(R.b[i], B + (long)max(src, 0) * p.ldb + cc);
            if (m + BR < mend) {
                b_mm[i] = m + BR;
                if (!PLAIN) b_q[i] = b_inc ? rowmap_step(bmap, b_q[i], BR) : rowmap_prep(bmap, b_mm[i]);
            }
        }
        ld_m += BR;
    };
    auto sstore = [&](Regs& R, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < I_IT; ++i) {
            const int v = tid + i * NT;
            if (!OK && !((R.oka >> i) & 1u)) frag_zero(R.a[i]);
            if ((BR * VI) % NT != 0 && v >= BR * VI) continue;
            frag_store(R.a[i], Ys + (buf * BR + v / VI) * LDI + (v % VI) * 8);
            if (do_colsum) {
#pragma unroll
                for (int e = 0; e < 8; ++e) csum[i][e] += frag_get(R.a[i], e);
            }
        }
#pragma unroll
        for (int i = 0; i < J_IT; ++i) {
            const int v = tid + i * NT;
            if (!OK && !((R.okb >> i) & 1u)) frag_zero(R.b[i]);
            if ((BR * VJ) % NT != 0 && v >= BR * VJ) continue;
            frag_store(R.b[i], Xs + (buf * BR + v / VJ) * LDJ + (v % VJ) * 8);
        }
    };

    auto body = [&](Regs& R, int st) __attribute__((always_inline)) {
        const int buf = st & 1;
        sstore(R, buf);                                   // waits for this set's loads only
        __syncthreads();                                  // tile st visible; buffer (st+1)&1 no longer read by anyone
        // UNCONDITIONAL as well (past the end every row is flagged invalid and the addresses are clamped): with
        // `if (st + 2 < nsteps)` around it the waitcnt pass merges the two paths conservatively and the sstore above
        // waits vmcnt(7..0) instead of vmcnt(15..8), i.e. for BOTH sets - the measured step time was then
        // load time + compute time instead of their maximum
        gload(R);
        const T* ys = Ys + buf * BR * LDI;
        const T* xs = Xs + buf * BR * LDJ;
#pragma unroll
        for (int kb = 0; kb < BR / 32; ++kb) {
            Frag<T> fa[FI], fb[FJ];
#pragma unroll
            for (int i = 0; i < FI; ++i) gather_frag<T, TR>(fa[i], ys + kb * 32 * LDI, LDI, wi * WTI + i * 16, li, lg);
#pragma unroll
            for (int j = 0; j < FJ; ++j) gather_frag<T, TR>(fb[j], xs + kb * 32 * LDJ, LDJ, wj * WTJ + j * 16, li, lg);
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) mma16(fa[i], fb[j], acc[i][j]);     // D[i][c]: row = 4*lg + r, col = li
        }
    };
    Regs r0, r1;
    gload(r0);
    gload(r1);
    for (int st = 0; st < nsteps; st += 2) {              // an odd count runs one all-zero step at the end
        body(r0, st);
        body(r1, st + 1);
    }
    };
    if (all_ok) run(std::true_type{}); else run(std::false_type{});
    __syncthreads();                                      // the tiles are dead: LDS is reused by the bias-gradient reduction

    if (do_colsum) {
        // every (thread, i) owns one (tile row, column vector) cell: park the per-thread sums in a BR x BI fp32 matrix (plain
        // stores, no two threads share a cell), then thread c adds up column c.  (LDS atomics on BI addresses serialised
        // 8192 updates per block here and cost more than a separate column-sum pass over A.)
        float* red = reinterpret_cast<float*>(smem);          // the tiles are dead after the last barrier
        static_assert(BR * BI * 4 <= 2 * BR * (LDI + LDJ) * (int)sizeof(T), "bias-gradient scratch fits in the operand buffers");
#pragma unroll
        for (int i = 0; i < I_IT; ++i) {
            const int v = tid + i * NT;
            if ((BR * VI) % NT != 0 && v >= BR * VI) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) red[(v / VI) * BI + (v % VI) * 8 + e] = csum[i][e];
        }
        __syncthreads();
        if (tid < BI && i0 + tid < p.NI) {
            float a0 = 0.f, a1 = 0.f;
#pragma unroll 8
            for (int r = 0; r < BR; r += 2) { a0 += red[r * BI + tid]; a1 += red[(r + 1) * BI + tid]; }
            if (p.partial) p.partial[(long)gridDim.x * (BI * BJ) + ((long)ms * tiles_i + ti) * BI + tid] = a0 + a1;   // after the partial tiles
            else { atomicAdd(p.colsum + i0 + tid, a0 + a1); if (p.colsum2) atomicAdd(p.colsum2 + i0 + tid, a0 + a1); }
        }
    }

    // ---- epilogue: fp32 atomics into the parameter-layout gradient.  The MFMA layout gives a lane 4 rows x 1 column per
    // fragment (a wave instruction would touch 4 x 64-byte pieces); each wave stages its WTI x WTJ sub-tile in LDS and
    // issues the atomics row by row instead, 64 consecutive columns per instruction.
    __syncthreads();                                      // operand tiles / bias scratch are dead
    constexpr bool STAGE_FITS = WI * WJ * WTI * (WTJ + 1) * 4 <= 2 * BR * (LDI + LDJ) * (int)sizeof(T);
    if constexpr (!STAGE_FITS) {
        // the large tiles (192 x 256: 196 KB of fp32 per block against 122 KB of LDS): straight from the accumulators, a wave instruction
        // writes 4 rows x 16 consecutive floats - the tile is stored once per block, after ~40 steps of the loop
        float* pt = p.partial ? p.partial + ((long)ms * tiles_ij + (bid % tiles_ij)) * (BI * BJ) : nullptr;
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wi * WTI + i * 16 + lg * 4 + r, col = wj * WTJ + j * 16 + li;
                    if (pt) pt[row * BJ + col] = acc[i][j][r];
                    else if (i0 + row < p.NI && c0 + col < p.K) atomicAdd(tn_dst(p, i0 + row, c0 + col, tap), acc[i][j][r]);
                }
    } else {
        constexpr int SLD = WTJ + 1;                      // +1 float: conflict-free column writes
        float* st = reinterpret_cast<float*>(smem) + wid * (WTI * SLD);
#pragma unroll
        for (int i = 0; i < FI; ++i)
#pragma unroll
            for (int j = 0; j < FJ; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(i * 16 + lg * 4 + r) * SLD + j * 16 + li] = acc[i][j][r];
        // wave-local: the LDS unit executes one wave's accesses in order, no barrier needed
        if (p.partial) {
            // two-phase reduction: plain coalesced stores of this block's partial tile; tn_reduce_kernel sums the row splits
            float* pt = p.partial + ((long)ms * tiles_ij + (bid % tiles_ij)) * (BI * BJ);
            for (int v = lane; v < WTI * WTJ; v += 64) {
                const int row = v / WTJ, col = v - row * WTJ;
                pt[(wi * WTI + row) * BJ + wj * WTJ + col] = st[row * SLD + col];
            }
        } else {
            for (int v = lane; v < WTI * WTJ; v += 64) {
                const int row = v / WTJ, col = v - row * WTJ;
                const int ii = i0 + wi * WTI + row, c = c0 + wj * WTJ + col;
                if (ii < p.NI && c < p.K) atomicAdd(tn_dst(p, ii, c, tap), st[row * SLD + col]);
            }
        }
    }
}

// sums the row-split partial tiles and adds the result into the parameter-layout gradient.  A block = four groups of splits (tid >> 6) x 64 float4
// pieces: every group walks its quarter of the splits in index order with four loads in flight (small tiles are split dozens of ways: one thread walking
// all of them serially was latency-bound), the four group sums meet in LDS in a FIXED order, and each element receives exactly ONE atomic add per launch -
// the gradient is bit-repeatable whatever the block schedule (round 6; until then the groups were blockIdx.y and met in atomics: the 3-tap temporal
// gradients differed in the last bits from run to run).  Trailing blocks: the bias-gradient partials, combined the same way.
template <int BI, int BJ>
__global__ __launch_bounds__(NT) void tn_reduce_kernel(const dist_gemm_tn_args p, int msplit, int tiles_i, int tiles_c, int nb_main) {
    static_assert(NT == 256 && BJ % 4 == 0, "four split groups of 64 lanes; float4 pieces inside a tile row");
    const int tiles_ij = tiles_i * tiles_c * p.taps;
    const long total = (long)tiles_ij * BI * BJ, total4 = total / 4;
    const int tid = threadIdx.x, gq = tid >> 6, ln = tid & 63;
    const int per = (msplit + 3) / 4, s0 = gq * per, s1 = min(msplit, s0 + per);
    __shared__ f32x4 red[3][64];
    if ((int)blockIdx.x >= nb_main) {                     // the bias-gradient partials (tiles_i * BI per split)
        const long c = (long)((int)blockIdx.x - nb_main) * 64 + ln;
        const bool in = p.colsum != nullptr && c < p.NI;
        float a = 0.f;
        if (in) {
            const float* __restrict__ cp = p.partial + (long)msplit * total + c;
            for (int s = s0; s < s1; ++s) a += cp[(long)s * tiles_i * BI];
        }
        if (gq > 0) red[gq - 1][ln] = f32x4{a, 0.f, 0.f, 0.f};
        __syncthreads();
        if (gq != 0 || !in) return;
        a = ((a + red[0][ln][0]) + red[1][ln][0]) + red[2][ln][0];
        atomicAdd(p.colsum + c, a);
        if (p.colsum2) atomicAdd(p.colsum2 + c, a);
        return;
    }
    const long e4 = (long)blockIdx.x * 64 + ln;
    const bool in = e4 < total4;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (in) {
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(p.partial) + e4;
        int s = s0;
        for (; s + 3 < s1; s += 4) {
            a0 += src[(long)s * total4];
            a1 += src[(long)(s + 1) * total4];
            a2 += src[(long)(s + 2) * total4];
            a3 += src[(long)(s + 3) * total4];
        }
        for (; s < s1; ++s) a0 += src[(long)s * total4];
    }
    f32x4 v = (a0 + a1) + (a2 + a3);
    if (gq > 0) red[gq - 1][ln] = v;
    __syncthreads();
    if (gq != 0 || !in) return;
    v = ((v + red[0][ln]) + red[1][ln]) + red[2][ln];
    const long e = e4 * 4;
    const int t = (int)(e / (BI * BJ)), r = (int)(e % (BI * BJ));
    const int row = r / BJ, col = r - row * BJ;
    const int ti = t % tiles_i, tc = (t / tiles_i) % tiles_c, tap = t / (tiles_i * tiles_c);
    const int ii = ti * BI + row, c = tc * BJ + col;
    if (ii >= p.NI) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (c + q < p.K) atomicAdd(tn_dst(p, ii, c + q, tap), v[q]);
}

template <typename T, int BI, int BJ, int WI, int WJ, bool TR, int MODES>
int launch(const dist_gemm_tn_args& a, hipStream_t s) {
    constexpr size_t smem = (size_t)2 * BR * (BI + BJ + 2 * PADT) * sizeof(T);
    static DistSmemOnce attr;
    auto kern = gemm_tn_kernel<T, BI, BJ, WI, WJ, TR, MODES>;
    RUN_(dist_max_smem(attr, reinterpret_cast<const void*>(kern), smem));
    const int tiles_i = (a.NI + BI - 1) / BI, tiles_c = (a.K + BJ - 1) / BJ;
    const long tiles = (long)tiles_i * tiles_c * a.taps;
    // split the reduction so that at most 256 blocks exist (one per CU); at least 512 rows per block (every block ends with a
    // tile-size partial store).  512 blocks (two co-resident per CU) are 5-10 % faster per launch ALONE, but every split adds a
    // partial tile that is written and read again (33 MB each way for the 384x768 gradient at 28 splits, against 116 MB of
    // operands): in the step, where the other streams fill the CUs anyway, 256 blocks are 0.25 ms faster (22.77 / 22.79 ->
    // 22.52 / 22.53 ms; 384: 22.69, 128: 22.85).
    static const int max_blocks = dist_knob("DIST_AMD_TN_BLOCKS", 96);   // (round 4: 96 - the weight-gradient streams only have to keep up with the data-gradient chain,
                                                                       //  and every block they do not hold a CU with is one the chain gets: 17.93 ms with 256, 17.79 with 128, 17.57 with 96, 18.14 with 64)
    long msplit = (a.max_blocks > 0 ? a.max_blocks : max_blocks) / tiles;
    const long max_split = (a.M + 511) / 512;
    if (msplit > max_split) msplit = max_split;
    if (msplit < 1) msplit = 1;
    int chunk = (int)((a.M + msplit - 1) / msplit);
    static const bool pairs = (DIST_AB_KNOB("DIST_AMD_TN_PAIRS", 1) != 0);   // A/B: 0 = no fast path
    const int CH = (MODES == 0 && pairs) ? 2 * BR : BR;   // plain maps: whole step pairs per block (the kernel's all-valid fast path)
    chunk = (chunk + CH - 1) / CH * CH;
    msplit = (a.M + chunk - 1) / chunk;
    dist_gemm_tn_args b = a;
    const bool two_phase = a.partial != nullptr && msplit > 1 && tiles * msplit * (long)(BI * BJ) + msplit * (long)tiles_i * BI <= a.partial_elems;
    if (!two_phase) b.partial = nullptr;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * msplit)), dim3(WI * WJ * 64), smem, s, b, chunk, tiles_i, tiles_c);
    HIP_CHECK_RET(hipGetLastError());
    static const bool skip_reduce = (dist_measure_knob("DIST_AMD_TN_SKIP_REDUCE", 0) != 0);   // measurement knob (results WRONG): what the second phase holds of the step
    if (two_phase && !skip_reduce) {
        const long total4 = tiles * (long)(BI * BJ) / 4;
        const long nb_main = (total4 + 63) / 64, nb_cs = a.colsum ? ((long)tiles_i * BI + 63) / 64 : 0;
        hipLaunchKernelGGL((tn_reduce_kernel<BI, BJ>), dim3((unsigned)(nb_main + nb_cs)), dim3(NT), 0, s, b, (int)msplit, tiles_i, tiles_c, (int)nb_main);
        HIP_CHECK_RET(hipGetLastError());
    }
    return DIST_OK;
}

template <typename T, bool TR, int MODES>
int dispatch2(const dist_gemm_tn_args& a, hipStream_t s) {
    const bool i96 = (a.NI % 96 == 0) && (a.NI % 128 != 0);
    const bool j96 = (a.K % 96 == 0) && (a.K % 128 != 0);
    // 8 (6 for 96 x 96) waves per tile: half the accumulators / fragments / staging registers per lane of the 4-wave shapes
    // (216-280 registers: 1-2 blocks of 4 waves per CU; now 119-152: 8 waves per CU in one block).  10-20 % faster per launch
    // alone (conv3x3 dW 95.5 -> 77.4 us, 384x768 dW 71.9 -> 62.0 us); capping the registers at 128 for two 8-wave blocks spills
    // and loses (92 us).  DIST_AMD_TN_W8=0: the 4-wave shapes (measurement knob; they exist for plain and runtime modes only).
    static const int w8 = DIST_AB_KNOB("DIST_AMD_TN_W8", 1);
    // (the large plain gradients - both extents >= 192, >= 8192 rows - run on gemm_tn8p_kernel, gemm_tn8p.hip: dist_op_gemm_tn tries it first;
    //  192 x 256 / 192 x 192 register-staged tiles and an LDS-DMA ring in THIS kernel's lock-step schedule were measured and lost, profiles/r03_tn_big_tiles.md)
    if (w8 || MODES > 0) {
        if (i96 && j96) return launch<T, 96, 96, 3, 2, TR, MODES>(a, s);
        if (i96) return launch<T, 96, 128, 2, 4, TR, MODES>(a, s);
        if (j96) return launch<T, 128, 96, 4, 2, TR, MODES>(a, s);
        return launch<T, 128, 128, 2, 4, TR, MODES>(a, s);
    }
    if constexpr (DIST_AB && MODES <= 0) {                  // the 4-wave shapes: timing-only library
        if (i96 && j96) return launch<T, 96, 96, 2, 2, TR, MODES>(a, s);
        if (i96) return launch<T, 96, 128, 2, 2, TR, MODES>(a, s);
        if (j96) return launch<T, 128, 96, 2, 2, TR, MODES>(a, s);
        return launch<T, 128, 128, 2, 2, TR, MODES>(a, s);
    }
    return DIST_ERR_ARG;
}
template <typename T, bool TR>
int dispatch(const dist_gemm_tn_args& a, hipStream_t s) {
    // plain row maps on both operands (every Linear's dW): no per-row index arithmetic in the loader
    if (a.amap.mode == DIST_RM_PLAIN && a.bmap.mode == DIST_RM_PLAIN && a.taps == 1) return dispatch2<T, TR, 0>(a, s);
    // the combinations the engine uses (conv_t / temporal-ffn / stem, conv3x3, I2T, T2I weight gradients) with constant modes
    if constexpr (std::is_same<T, bf16_t>::value && TR) {
        static const int w8 = DIST_AB_KNOB("DIST_AMD_TN_W8", 1);
        static const int spec = DIST_AB_KNOB("DIST_AMD_TN_MODES", 1);   // A/B: 0 = runtime modes for the engine's combinations too
        const int modes = a.amap.mode * 8 + a.bmap.mode;
        auto steppable = [](const dist_rowmap& rm) {       // rowmap_inc_ok for a step of BR rows
            switch (rm.mode) {
                case DIST_RM_SHIFT: return BR <= rm.p0;
                case DIST_RM_SPATIAL: return rm.p0 > 0 && BR / rm.p0 + 1 <= rm.p0;
                case DIST_RM_STRIDED: return BR <= rm.p1;
                case DIST_RM_SKIPCLS: return BR <= rm.p0;
                default: return true;
            }
        };
        if (w8 && spec && steppable(a.amap) && steppable(a.bmap)) switch (modes) {
            case DIST_RM_SHIFT: return dispatch2<T, TR, DIST_RM_SHIFT>(a, s);
            case DIST_RM_SPATIAL: return dispatch2<T, TR, DIST_RM_SPATIAL>(a, s);
            case DIST_RM_SKIPCLS: return dispatch2<T, TR, DIST_RM_SKIPCLS>(a, s);
            case DIST_RM_SKIPCLS * 8 + DIST_RM_STRIDED: return dispatch2<T, TR, DIST_RM_SKIPCLS * 8 + DIST_RM_STRIDED>(a, s);
            default: break;
        }
    }
    return dispatch2<T, TR, -1>(a, s);
}

}  // namespace

extern "C" int dist_op_gemm_tn(const dist_gemm_tn_args* a, void* stream) {
    if (!a || !a->A || !a->B || !a->out || a->M <= 0 || a->NI <= 0 || a->K <= 0 || a->taps <= 0) return DIST_ERR_ARG;
    if (a->NI % 8 || a->lda % 8 || a->ldb % 8 || a->inner <= 0 || (a->K + 7) / 8 * 8 > a->ldb) return DIST_ERR_ARG;
    if (a->M > (1 << 30)) return DIST_ERR_ARG;
    if (a->out2 && (a->taps != 1 || a->inner != 1 || a->so_outer != 1 || a->split_c <= 0 || a->split_c >= a->K)) return DIST_ERR_ARG;
    if (a->colsum2 && !a->colsum) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    {   // the large plain gradients: two-group LDS-DMA kernel (gemm_tn8p.hip)
        const int rc = dist_k_gemm_tn8p(a, s);
        if (rc != 0) return rc < 0 ? rc : DIST_OK;
    }
    {   // the 3x3 frame convolution's gradient: all nine taps from one LDS-resident frame (conv_dw.hip)
        const int rc = dist_k_conv3x3_dw(a, s);
        if (rc != 0) return rc < 0 ? rc : DIST_OK;
    }
    {   // the temporal convolutions' gradients (3 / 5 frame taps, 96 output channels): a ring of frame slots in LDS (conv_t_dw.hip)
        const int rc = dist_k_conv_t_dw(a, s);
        if (rc != 0) return rc < 0 ? rc : DIST_OK;
    }
    if (a->dtype == DIST_BF16) return a->use_tr ? dispatch<bf16_t, true>(*a, s) : dispatch<bf16_t, false>(*a, s);
    return dispatch<float, false>(*a, s);
}
