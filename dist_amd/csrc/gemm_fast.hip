// Fast path of dist_op_gemm_nt for the frozen-ViT shapes (bf16; single-tap row maps, plain / insert-cls outputs):
//   C[m][n] = epi( sum_k A[m][k] * B[n][k] ),  256 x 256 x 32 tiles, 8 waves (2 x 4).
//
// CDNA4 structure (cdna_hip_programming.md §5):
//  * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no
//    ds_write pass) into a 4-deep ring of 32 KB stages, three K-tiles ahead of the MFMAs;
//  * ONE raw s_barrier per K-tile with a COUNTED s_waitcnt vmcnt (never 0 in steady state);
//  * the fragments of tile k+1 are read from LDS into a second register set while the 32 MFMAs
//    of tile k issue, so the two waves of a SIMD do not serialise "all read, then all multiply";
//  * the LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle sits on the
//    per-lane SOURCE address and again on the fragment reads: 16-byte chunk c of row r lives at
//    chunk c ^ f((r>>2)&3), f = {0,3,2,1}: every 16-lane ds_read_b128 service group then hits 16
//    distinct slots of the 256-B bank row (rows are 64 B);
//  * each wave owns a 128 x 64 output sub-tile (8 x 4 fragments, 128 accumulator VGPRs, one wave
//    pair per SIMD); the MFMA is issued swapped so a lane holds 4 consecutive output columns;
//  * epilogue through LDS: the residual tile is fetched and the result tile is written as full
//    128-byte rows (16 B per lane), instead of 8-byte pieces at a row stride.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

// Two shapes of the same kernel (NW = waves per block, every wave owns a 128 x 64 output sub-tile):
//   NW = 8: 256 x 256 tile, 4-stage ring (128 KB), one block per CU;
//   NW = 4: 256 x 128 tile, 3-stage ring (72 KB), TWO blocks per CU.  The two blocks run out of phase: one block's
//           prologue (first DMA latency) and epilogue (residual fetch, 64 KB of stores) overlap the other's MFMA
//           loop, and the barrier of one block no longer idles the SIMDs.  With one block per CU the fixed part of a
//           block (13-18 us) was as long as its whole K = 768 loop.
constexpr int BM = 256, BK = 32;
constexpr int A_BYTES = BM * BK * 2;                      // 16 KB
constexpr int EPI_BYTES = 128 * 64 * 2;                   // per-wave staging region: 128 rows x 128 B
template <int NW> struct Shape {
    static constexpr int WN = NW / 2;                     // waves along n (2 along m)
    static constexpr int BN = 64 * WN;
    static constexpr int NT = 64 * NW;
    static constexpr int STAGES = NW == 8 ? 4 : 3, AHEAD = STAGES - 1;
    static constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    static constexpr int PA = 16 / NW, PB = 2, NP = PA + PB;   // 1-KB LDS-DMA pieces per wave per stage (A rows, B rows)
    static constexpr int MINB = NW == 8 ? 1 : 2;
};

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

DEV int swz(int q) { return (4 - q) & 3; }               // f = {0,3,2,1}

struct FragSet { bf16x8 a[8]; bf16x8 b[4]; };

// Epilogue shared by the two main loops: accumulators -> (LayerNorm fold, bias, residual, QuickGELU) -> bf16 -> per-wave LDS
// staging -> full 128-byte rows.  acc[i][j][r] is output row mw + i*16 + (lane & 15), column nw + j*16 + (lane >> 4)*4 + r.
// EARLY_OK: the primary output may leave row block by row block behind its conversion (the bf16 two-group kernel; in the e4m3 instantiation the
// longer epilogue made hipcc spill inside the K loop, so it keeps the old order)
// AUX (with EARLY_OK, the same kernel): bias / column sums / row statistics of the tile were prefetched into LDS by the kernel's prologue
// (`aux`: bias[256] | colsum[256] | mean[256] | rstd[256] floats), so the epilogue starts without a global-load latency
// CF >= 0: the epilogue's shape is known at compile time (low 16 bits = p.flags, bit 16 = head-major output, bit 17 = C is NULL; plain or
// head-major output map) - the ViT's three epilogues run straight-line code instead of the generic one's flag branches (round 5)
constexpr int CF_HEADS = 1 << 16, CF_NOC = 1 << 17;
// qgelu_t<bf16_t> on two values with its multiplies and the add as packed fp32 instructions (same operations, same roundings: 4 packed + 4
// transcendental instructions per pair instead of 6 + 4)
DEV void qgelu2_bf16(float& a, float& b) {
#pragma clang fp contract(off)
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t x = {a, b};
    const f32x2_t u = (x * -1.702f) * 1.44269504088896340736f;
    const f32x2_t d = f32x2_t{__builtin_amdgcn_exp2f(u[0]), __builtin_amdgcn_exp2f(u[1])} + 1.f;
    const f32x2_t y = x * f32x2_t{__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    a = y[0]; b = y[1];
}
template <bool EARLY_OK = false, int CF = -1>
DEV void fast_epilogue(const dist_gemm_args& p, f32x4 (&acc)[8][4], char* smem, const int wid, const int lane,
                       const int m0, const int n0, const int wm, const int wn, const bool late_flush = false, const char* aux = nullptr, const int dbg = 0, const int respf_par = 0) {
    const int li = lane & 15, lg = lane >> 4;
    const int M = (int)p.M, N = (DIST_AB && (dbg & 1)) ? 0 : p.N;      // (timing-only library, dbg bit 0: no output stores; bit 1: no epilogue at all)
    if (DIST_AB && (dbg & 2)) {
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (sacc == 12345.678f) static_cast<bf16_t*>(p.C ? p.C : p.C2)[lane] = (bf16_t)sacc;
        return;
    }
    // ---- epilogue through LDS: per-wave region of 128 rows x 128 B, 16-B chunk c of row r at chunk c ^ (r & 7) ----
    __syncthreads();                                      // every wave is done reading the operand ring
    // RESPF (the residual epilogue of the bf16 two-group kernel): the kernel's last K-tile already moved rows 0 ... 63 of this wave's residual tile
    // by LDS-DMA into the ring buffer that K-tile no longer reads (`respf_par`: its parity) - the staging region of a wave is then the same 8 KB
    // (slot wid / 2, half wid % 2) of BOTH parities: rows 0 ... 63 in the free one, rows 64 ... 127 in the other
    constexpr bool RESPF = EARLY_OK && CF >= 0 && (CF & DIST_EPI_RES) != 0;
    char* const ew = RESPF ? smem + (wid >> 1) * (2 * EPI_BYTES) + respf_par * EPI_BYTES + (wid & 1) * (EPI_BYTES / 2) : smem + wid * EPI_BYTES;
    char* const ew_hi = RESPF ? smem + (wid >> 1) * (2 * EPI_BYTES) + (respf_par ^ 1) * EPI_BYTES + (wid & 1) * (EPI_BYTES / 2) - 64 * 128 : ew;
    auto erow = [&](const int r) __attribute__((always_inline)) -> char* { return (RESPF && r >= 64 ? ew_hi : ew) + r * 128; };
    bf16_t* __restrict__ C = (CF >= 0 && (CF & CF_NOC)) ? nullptr : static_cast<bf16_t*>(p.C);
    bf16_t* __restrict__ C2 = static_cast<bf16_t*>(p.C2);
    const bf16_t* __restrict__ R = static_cast<const bf16_t*>(p.res);
    const int flags = CF >= 0 ? (CF & 0xffff) : p.flags;
    const int mw = m0 + wm * 128, nw = n0 + wn * 64;
    const int crow = lane >> 3, cchunk = lane & 7;        // coalesced pass: 8 lanes cover one 128-B row
    const int icls = CF >= 0 ? 0 : (p.omap.mode == DIST_OM_INSERTCLS ? p.omap.p0 : 0);
    auto dest_row = [&](int m) -> long { return icls ? (long)(m / icls) * (icls + 1) + 1 + m % icls : (long)m; };
    // DIST_OM_HEADS: this wave's 64 columns are exactly one (q|k|v, head) slice; row (frame, token) goes to
    // [frame][head][part][token][64], so the 8 rows of a store instruction are 1 KB contiguous
    const bool heads_om = CF >= 0 ? (CF & CF_HEADS) != 0 : p.omap.mode == DIST_OM_HEADS;
    const int hp_part = (nw >> 6) / max(p.omap.p1, 1), hp_head = (nw >> 6) - hp_part * max(p.omap.p1, 1);

    // SPEC (compile-time epilogue): plain row maps, N % 64 == 0 and tensors below 4 GB (the launcher checks) - rows are addressed through
    // buffer descriptors that end behind row M - 1 (a wave whose columns lie beyond N gets an empty one), so the pieces of a ragged last row
    // tile are dropped / read as zero by the bounds check: no 64-bit address arithmetic, no exec masking per piece
    constexpr bool SPEC = CF >= 0;
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    uint4 rv[16];
    if (flags & DIST_EPI_RES) {
        // all 16 row-pieces of the residual tile are requested back to back (the operand fragment registers are
        // dead by now), so ONE memory latency is exposed instead of one per batch
        if constexpr (SPEC) {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(R), 0, nw < N ? M * p.ldres * 2 : 0, 0x00020000);
            const unsigned offr = (unsigned)(mw + crow) * (unsigned)p.ldres * 2u + (unsigned)(nw + cchunk * 8) * 2u;
            const int stepr = 8 * p.ldres * 2;
#pragma unroll
            for (int it = RESPF ? 8 : 0; it < 16; ++it) {   // (RESPF: rows 0 ... 63 are in the staging region already)
                const v4u_t v = __builtin_amdgcn_raw_buffer_load_b128(rr, offr, it * stepr, 0);
                rv[it] = make_uint4(v[0], v[1], v[2], v[3]);
            }
        } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 8 + crow;
            const int m = mw + r, n = nw + cchunk * 8;
            rv[it] = make_uint4(0, 0, 0, 0);
            if (m < M && n < N) rv[it] = *reinterpret_cast<const uint4*>(R + dest_row(m) * p.ldres + n);
        }
        }
        if constexpr (!RESPF) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 8 + crow;
            *reinterpret_cast<uint4*>(erow(r) + ((cchunk ^ (r & 7)) << 4)) = rv[it];
        }
        }
    }
    const bool from_lds = EARLY_OK && (!DIST_AB || aux != nullptr);      // (product: always in the bf16 two-group kernel)
    float bias4[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (from_lds) {
            const f32x4 b = (flags & DIST_EPI_BIAS) ? *reinterpret_cast<const f32x4*>(aux + (wn * 64 + j * 16 + lg * 4) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) bias4[j][r] = b[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nw + j * 16 + lg * 4 + r;
                bias4[j][r] = ((flags & DIST_EPI_BIAS) && n < N) ? p.bias[n] : 0.f;
            }
        }
    }
    const bool lnf = (flags & DIST_EPI_LNFOLD) != 0;
    if (lnf) {
        // LayerNorm folded into this GEMM: A held the raw rows, B = W diag(gamma); v = rstd[m] * (acc - mean[m] * colsum[n]) + bias'
        // (lnfold_bias: explicit roundings, the bias included - the conversion loop below then adds nothing)
        const float* __restrict__ st = static_cast<const float*>(p.aux);     // [2][M]: mean, rstd
        float cs4[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (from_lds) {
                const f32x4 c = *reinterpret_cast<const f32x4*>(aux + 1024 + (wn * 64 + j * 16 + lg * 4) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) cs4[j][r] = c[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = nw + j * 16 + lg * 4 + r;
                    cs4[j][r] = n < N ? p.bias2[n] : 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = min(mw + i * 16 + li, M - 1);
            float mean, rstd;
            if (from_lds) {
                mean = *reinterpret_cast<const float*>(aux + 2048 + (wm * 128 + i * 16 + li) * 4);
                rstd = *reinterpret_cast<const float*>(aux + 3072 + (wm * 128 + i * 16 + li) * 4);
            } else { mean = st[m]; rstd = st[(long)M + m]; }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = lnfold_bias(acc[i][j][r], mean, rstd, cs4[j][r], bias4[j][r]);
        }
    }
    const bool act_only = (flags & DIST_EPI_ACT2) && !C;
    // The primary output (C, or C2 when only the activated tensor is kept) leaves row block by row block, right behind its conversion:
    // the two 8-row pieces of block i are stored while blocks i + 1 ... are still being converted, instead of all 16 behind the last
    // conversion (round 5: the stores of a tile start ~1 us earlier; same values, same addresses).
    // measured (tools/r05_ef_check.sh, two alternations): in_proj (LN fold + head-major) +2...4 %, c_fc 0...+1 %; with a residual tile in the
    // staging region (out_proj / c_proj) -1...+0.5 %: those keep the old order.  (DIST_AMD_FAST_EARLY_FLUSH=0 in the timing-only library: the old order)
    const bool early = EARLY_OK && (!DIST_AB || !late_flush) && (SPEC || !(flags & DIST_EPI_RES));
    bf16_t* __restrict__ const dst1 = act_only ? C2 : C;
    const int ld1 = act_only ? p.ldc2 : p.ldc;
    int hfr1 = 0, htok1 = 0;
    if (heads_om) { hfr1 = (mw + crow) / p.omap.p0; htok1 = (mw + crow) - hfr1 * p.omap.p0; }
    auto flush_piece = [&](const int it) __attribute__((always_inline)) {
        const int r = it * 8 + crow;
        const int m = mw + r, n = nw + cchunk * 8;
        const uint4 v = *reinterpret_cast<const uint4*>(erow(r) + ((cchunk ^ (r & 7)) << 4));
        if (m < M && n < N) {
            if (heads_om) store16_nt(dst1 + ((((long)hfr1 * p.omap.p1 + hp_head) * 3 + hp_part) * p.omap.p0 + htok1) * 64 + cchunk * 8, v);
            else store16_nt(dst1 + dest_row(m) * ld1 + n, v);
        }
        if (heads_om) { htok1 += 8; if (htok1 >= p.omap.p0) { htok1 -= p.omap.p0; ++hfr1; } }      // (the launcher checks p0 >= 16)
    };
    // SPEC: the primary output through a buffer descriptor; a piece is READ from the staging region right behind its row block's conversion and
    // STORED behind the next block's (the LDS round trip hides behind that block's arithmetic instead of stalling every piece)
    __amdgpu_buffer_rsrc_t rd1 = __builtin_amdgcn_make_buffer_rsrc(dst1, 0, 0, 0x00020000);
    unsigned off1 = 0, fs_adj = 0;
    int step1 = 0;
    if constexpr (SPEC) {
        if (heads_om) {
            const unsigned fs = (unsigned)p.omap.p1 * 3u * (unsigned)p.omap.p0 * 128u;      // bytes per frame: [head][q|k|v][token][64]
            rd1 = __builtin_amdgcn_make_buffer_rsrc(dst1, 0, nw < N ? (int)((unsigned)(M / p.omap.p0) * fs) : 0, 0x00020000);
            off1 = (unsigned)hfr1 * fs + (unsigned)((hp_head * 3 + hp_part) * p.omap.p0 + htok1) * 128u + (unsigned)cchunk * 16u;
            fs_adj = fs - (unsigned)p.omap.p0 * 128u;
        } else {
            rd1 = __builtin_amdgcn_make_buffer_rsrc(dst1, 0, nw < N ? M * ld1 * 2 : 0, 0x00020000);
            off1 = (unsigned)(mw + crow) * (unsigned)ld1 * 2u + (unsigned)(nw + cchunk * 8) * 2u;
            step1 = 8 * ld1 * 2;
        }
    }
    auto ld_piece = [&](const int it) __attribute__((always_inline)) {
        const int r = it * 8 + crow;
        return *reinterpret_cast<const v4u_t*>(erow(r) + ((cchunk ^ (r & 7)) << 4));
    };
    auto st_piece = [&](const int it, const v4u_t& v) __attribute__((always_inline)) {      // (pieces in order 0 ... 15: the head-major offset is stepped)
        if (heads_om) {
            if (DIST_AB && (dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, rd1, off1, 0, 0);      // (timing-only library: without the streaming hint)
            else __builtin_amdgcn_raw_buffer_store_b128(v, rd1, off1, 0, 2);
            htok1 += 8; off1 += 1024u;
            if (htok1 >= p.omap.p0) { htok1 -= p.omap.p0; off1 += fs_adj; }                 // (the launcher checks p0 >= 16)
        } else if (DIST_AB && (dbg & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, rd1, off1, it * step1, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(v, rd1, off1, it * step1, 2);
    };
    v4u_t pend0 = {0, 0, 0, 0}, pend1 = {0, 0, 0, 0};
    // the residual values of row block i + 1 are read from the staging region BEFORE block i's results are written to it (the compiler keeps LDS
    // reads behind earlier LDS writes it cannot tell apart: read inside the j loop, every slot cost an exposed LDS round trip)
    float xr[2][4][4];
    auto ld_res = [&](const int i, float (&x)[4][4]) __attribute__((always_inline)) {
        const int r = i * 16 + li;
#pragma unroll
        for (int j = 0; j < 4; ++j) load4(reinterpret_cast<const bf16_t*>(erow(r) + ((((j << 1) | (lg >> 1)) ^ (r & 7)) << 4) + ((lg & 1) << 3)), x[j]);
    };
    constexpr bool RES_AHEAD = EARLY_OK;                  // (the bf16 two-group kernel; the e4m3 instantiation and the lock-step kernel keep the reads in the j loop:
                                                          //  with 32 more live registers here hipcc spilled in front of the e4m3 kernel's tail K-tile)
    if (RES_AHEAD && (flags & DIST_EPI_RES)) ld_res(0, xr[0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + li;
        if (RES_AHEAD && (flags & DIST_EPI_RES) && i < 7) {
            if constexpr (RESPF) {
                if (i == 3) {                             // rows 64 ... 127 of the residual tile (requested at the top) have had row blocks 0 ... 2 to arrive
#pragma unroll
                    for (int it = 8; it < 16; ++it) {
                        const int r2 = it * 8 + crow;
                        *reinterpret_cast<uint4*>(erow(r2) + ((cchunk ^ (r2 & 7)) << 4)) = rv[it];
                    }
                }
            }
            ld_res(i + 1, xr[(i + 1) & 1]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16_t* slot = reinterpret_cast<bf16_t*>(erow(r) + ((((j << 1) | (lg >> 1)) ^ (r & 7)) << 4) + ((lg & 1) << 3));
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = lnf ? acc[i][j][q] : acc[i][j][q] + bias4[j][q];
            if (flags & DIST_EPI_RES) {
                if constexpr (!RES_AHEAD) load4(slot, xr[0][j]);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += xr[RES_AHEAD ? (i & 1) : 0][j][q];
            }
            if (act_only) {
                if constexpr (SPEC) { qgelu2_bf16(v[0], v[1]); qgelu2_bf16(v[2], v[3]); }
                else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = qgelu_t<bf16_t>(v[q]);
                }
            }
            store4(slot, v);
        }
        if constexpr (SPEC) {
            if (dst1 && early) {
                if (i > 0) { st_piece(2 * i - 2, pend0); st_piece(2 * i - 1, pend1); }
                pend0 = ld_piece(2 * i); pend1 = ld_piece(2 * i + 1);
            }
        } else if (dst1 && early) { flush_piece(2 * i); flush_piece(2 * i + 1); }
    }
    if constexpr (SPEC) {
        if (dst1 && early) { st_piece(14, pend0); st_piece(15, pend1); }
        else if (dst1) {                                  // all 16 pieces behind the last conversion: 16 reads, then 16 stores
            v4u_t pv[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) pv[it] = ld_piece(it);
#pragma unroll
            for (int it = 0; it < 16; ++it) st_piece(it, pv[it]);
        }
    } else if (dst1 && !early) {                          // all 16 pieces behind the last conversion
#pragma unroll
        for (int it = 0; it < 16; ++it) flush_piece(it);
    }
    // C tiles are written once and read by a later kernel: streaming stores keep them from evicting the A / B panels that
    // the other column tiles on this XCD are still re-reading from L2 (QKV: 208 -> 189 us, FETCH_SIZE 350 -> 264 MB)
    auto flush = [&](bf16_t* __restrict__ dst, int ld) {
        // head-major rows: (frame, token) of the lane's first row by ONE division, then stepped by 8 rows per piece (16 divisions per
        // lane and tile were ~400 VALU of the QKV epilogue)
        int hfr = 0, htok = 0;
        if (heads_om) { hfr = (mw + crow) / p.omap.p0; htok = (mw + crow) - hfr * p.omap.p0; }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 8 + crow;
            const int m = mw + r, n = nw + cchunk * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(erow(r) + ((cchunk ^ (r & 7)) << 4));
            if (m < M && n < N) {
                if (heads_om) store16_nt(dst + ((((long)hfr * p.omap.p1 + hp_head) * 3 + hp_part) * p.omap.p0 + htok) * 64 + cchunk * 8, v);
                else store16_nt(dst + dest_row(m) * ld + n, v);
            }
            if (heads_om) { htok += 8; if (htok >= p.omap.p0) { htok -= p.omap.p0; ++hfr; } }      // (the launcher checks p0 >= 16)
        }
    };
    // DIST_EPI_OUT8: the staged bf16 tile (what C, or C2 with the activation, holds) leaves a second time as e4m3 with the caller's per-tensor
    // scale.  Row block i: all four slots of a lane are read, then its four packed dwords land in the first 64 bytes of the same lines
    // (a wave runs in lock-step: every read of a row block is issued before any of its writes), then 64-byte rows go out 16 bytes per lane.
    auto out8_pass = [&]() {
        unsigned char* __restrict__ C8 = static_cast<unsigned char*>(p.C8);
        const float inv = 1.0f / p.out8_scale[0];
        float amax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = i * 16 + li;
            float x[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                load4(reinterpret_cast<const bf16_t*>(erow(r) + ((((j << 1) | (lg >> 1)) ^ (r & 7)) << 4) + ((lg & 1) << 3)), x[j]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    amax = fmaxf(amax, fabsf(x[j][q]));
                    y[q] = fminf(fmaxf(x[j][q] * inv, -448.f), 448.f);
                }
                int w = 0;
                w = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], w, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], w, true);
                *reinterpret_cast<int*>(erow(r) + (((j ^ (r & 3)) + ((r >> 2) & 1) * 4) << 4) + lg * 4) = w;   // piece j of row r: spread over the banks
            }
        }
        int hfr = 0, htok = 0;
        if (heads_om) { hfr = (mw + (lane >> 2)) / p.omap.p0; htok = (mw + (lane >> 2)) - hfr * p.omap.p0; }
        if constexpr (SPEC) {                             // (compile-time epilogue: the image's rows through a buffer descriptor, as the primary output's)
            const int pc = lane & 3;
            __amdgpu_buffer_rsrc_t r8;
            unsigned off8, adj8 = 0;
            int step8 = 0;
            if (heads_om) {
                const unsigned fs = (unsigned)p.omap.p1 * 3u * (unsigned)p.omap.p0 * 64u;   // bytes per frame
                r8 = __builtin_amdgcn_make_buffer_rsrc(C8, 0, nw < N ? (int)((unsigned)(M / p.omap.p0) * fs) : 0, 0x00020000);
                off8 = (unsigned)hfr * fs + (unsigned)((hp_head * 3 + hp_part) * p.omap.p0 + htok) * 64u + (unsigned)pc * 16u;
                adj8 = fs - (unsigned)p.omap.p0 * 64u;
            } else {
                r8 = __builtin_amdgcn_make_buffer_rsrc(C8, 0, nw < N ? M * p.ldc8 : 0, 0x00020000);
                off8 = (unsigned)(mw + (lane >> 2)) * (unsigned)p.ldc8 + (unsigned)(nw + pc * 16);
                step8 = 16 * p.ldc8;
            }
            v4u_t pv8[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int r = it * 16 + (lane >> 2);
                pv8[it] = *reinterpret_cast<const v4u_t*>(erow(r) + (((pc ^ (r & 3)) + ((r >> 2) & 1) * 4) << 4));
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (heads_om) {
                    __builtin_amdgcn_raw_buffer_store_b128(pv8[it], r8, off8, 0, 2);
                    htok += 16; off8 += 1024u;
                    if (htok >= p.omap.p0) { htok -= p.omap.p0; off8 += adj8; }
                } else __builtin_amdgcn_raw_buffer_store_b128(pv8[it], r8, off8, it * step8, 2);
            }
        } else
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 16 + (lane >> 2), pc = lane & 3;
            const int m = mw + r, n = nw + pc * 16;
            const uint4 v = *reinterpret_cast<const uint4*>(erow(r) + (((pc ^ (r & 3)) + ((r >> 2) & 1) * 4) << 4));
            if (m < M && n < N) {
                // [frame][head][q|k|v][token][64] bytes: this wave's 64 columns are one (part, head) slice
                if (heads_om) store16_nt(C8 + ((((long)hfr * p.omap.p1 + hp_head) * 3 + hp_part) * p.omap.p0 + htok) * 64 + pc * 16, v);
                else store16_nt(C8 + (long)m * p.ldc8 + n, v);
            }
            if (heads_om) { htok += 16; if (htok >= p.omap.p0) { htok -= p.omap.p0; ++hfr; } }
        }
        if (p.out8_amax) {
            amax = wave_max(amax, 64);
            // >= 0: bit order = value order.  Read first: the running maximum only grows, so a (possibly stale) value that already covers this
            // tile makes the atomic unnecessary - a million same-address atomics per ViT pass cost more than the quantiser passes they replace
            // (a plain load: served by the CU's own L1 - staleness only costs a redundant atomic)
            if (lane == 0 && mw < M && nw < N && amax > *static_cast<const volatile float*>(p.out8_amax))
                atomicMax(reinterpret_cast<unsigned*>(p.out8_amax), __float_as_uint(amax));
        }
    };
    if (act_only) {                                       // (C2 left with the conversion loop above)
        if (flags & DIST_EPI_OUT8) out8_pass();
    } else {                                              // (C left with the conversion loop above; NULL with DIST_EPI_OUT8: the e4m3 image is the only output)
        if ((flags & DIST_EPI_ROWSTATS) && nw < N) {
            // DIST_EPI_ROWSTATS: sum and sum of squares of the wave's 64 stored columns (one slice; N % 64 == 0), read back from the
            // staged bf16 tile while its stores drain - the accumulators are dead here (inside the conversion loop above the same
            // sums cost 60-70 spilled registers).  Lane l owns rows l and l + 64; chunks and elements in a fixed order.
#pragma unroll 1
            for (int h2 = 0; h2 < 2; ++h2) {
                const int r = h2 * 64 + lane;
                float rs = 0.f, rq = 0.f;
                bf16x8 v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = *reinterpret_cast<const bf16x8*>(erow(r) + ((c ^ (r & 7)) << 4));
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float x = (float)v[c][e]; rs += x; rq += x * x; }
                if (mw + r < M) *reinterpret_cast<float2*>(p.rowstats + ((long)(nw >> 6) * M + mw + r) * 2) = make_float2(rs, rq);
            }
        }
        if (flags & DIST_EPI_OUT8) out8_pass();           // (never together with a second activated output: the launcher checks)
        if (flags & DIST_EPI_ACT2) {                      // second output = quickgelu(stored value)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = i * 16 + li;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16_t* slot = reinterpret_cast<bf16_t*>(erow(r) + ((((j << 1) | (lg >> 1)) ^ (r & 7)) << 4) + ((lg & 1) << 3));
                    float x[4];
                    load4(slot, x);
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = qgelu_t<bf16_t>(x[q]);
                    store4(slot, x);
                }
            }
            flush(C2, p.ldc2);
        }
    }
}

// block id -> (row tile, column tile): bijective XCD remap (consecutive ids share an XCD and its L2), then column tiles in
// `ngroups` groups, all row tiles of a group before the next group, the group's column tiles fastest.  With the contiguous
// XCD ranges an XCD works on ONE group for (nearly) its whole share: the group's weight rows (<= ~2.4 MB) stay in its 4 MB
// L2 while the activation panels stream through, instead of all of W (3.5 - 4.7 MB for the QKV / MLP GEMMs) being evicted
// and re-fetched for every batch of row tiles.
DEV void fast_tile(const dist_gemm_args& p, const int ngroups_flags, const int BN, int& tm, int& tn) {
    const int ngroups = ngroups_flags & 0xffff;           // (bit 16: the late-flush A/B switch of the timing-only library)
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (int)((p.M + BM - 1) / BM);
    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, x = bid % 8, y = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    if (ngroups <= 1) { tm = bid / tiles_n; tn = bid % tiles_n; }
    else {
        const int gq = tiles_n / ngroups, gr = tiles_n % ngroups;      // the first gr groups hold gq + 1 column tiles
        const int big = tiles_m * (gq + 1);
        int g, idg, gsz, g0;
        if (bid < gr * big) { g = bid / big; idg = bid - g * big; gsz = gq + 1; g0 = g * (gq + 1); }
        else { const int b2 = bid - gr * big; g = b2 / (tiles_m * gq); idg = b2 - g * (tiles_m * gq); gsz = gq; g0 = gr * (gq + 1) + g * gq; }
        tm = idg / gsz; tn = g0 + idg % gsz;
    }
}

// fast_tile with its run-time divisions done on the host (the two-group kernel: five reciprocal sequences of ~20 instructions sat in front of a
// tile's first LDS-DMA request): q = mulhi(n, ceil(2^32 / d)) is exact while n * d < 2^32 - block ids and tile counts are far below that.
struct TileMap {
    int q8, r8, tiles_n, ngroups, gq, gr, big, mid, grbig;
    unsigned mg_tn, mg_big, mg_mid, mg_gq1, mg_gq;
};
static TileMap make_tile_map(const long M, const int N, const int BN, const int ngroups) {
    TileMap t;
    const int tiles_n = (N + BN - 1) / BN, tiles_m = (int)((M + BM - 1) / BM), nblk = tiles_m * tiles_n;
    auto magic = [](const int d) -> unsigned { return d > 1 ? (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d) : 0u; };   // (d = 1: handled as n itself)
    t.q8 = nblk / 8; t.r8 = nblk % 8; t.tiles_n = tiles_n; t.ngroups = ngroups;
    t.gq = tiles_n / (ngroups > 0 ? ngroups : 1); t.gr = tiles_n % (ngroups > 0 ? ngroups : 1);
    t.big = tiles_m * (t.gq + 1); t.mid = tiles_m * t.gq; t.grbig = t.gr * t.big;
    t.mg_tn = magic(tiles_n); t.mg_big = magic(t.big); t.mg_mid = magic(t.mid); t.mg_gq1 = magic(t.gq + 1); t.mg_gq = magic(t.gq);
    return t;
}
DEV int tm_div(const int n, const int d, const unsigned mg) { return d > 1 ? (int)__umulhi((unsigned)n, mg) : n; }
DEV void fast_tile_mapped(const TileMap& t, int& tm, int& tn) {
    int bid = blockIdx.x;
    {
        const int x = bid & 7, y = bid >> 3;
        bid = (x < t.r8 ? x * (t.q8 + 1) : t.r8 * (t.q8 + 1) + (x - t.r8) * t.q8) + y;
    }
    if (t.ngroups <= 1) { tm = tm_div(bid, t.tiles_n, t.mg_tn); tn = bid - tm * t.tiles_n; }
    else if (bid < t.grbig) {
        const int g = tm_div(bid, t.big, t.mg_big), idg = bid - g * t.big;
        tm = tm_div(idg, t.gq + 1, t.mg_gq1); tn = g * (t.gq + 1) + idg - tm * (t.gq + 1);
    } else {
        const int b2 = bid - t.grbig, g = tm_div(b2, t.mid, t.mg_mid), idg = b2 - g * t.mid;
        tm = tm_div(idg, t.gq, t.mg_gq); tn = t.gr * (t.gq + 1) + g * t.gq + idg - tm * t.gq;
    }
}

template <int NW>
__global__ __launch_bounds__(Shape<NW>::NT, Shape<NW>::MINB) void gemm_fast_kernel(const dist_gemm_args p, const int ngroups) {
    using S = Shape<NW>;
    constexpr int BN = S::BN, STAGES = S::STAGES, AHEAD = S::AHEAD, STAGE_BYTES = S::STAGE_BYTES, PA = S::PA, PB = S::PB, NP = S::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / S::WN, wn = wid % S::WN;         // 2 x WN waves
    const int li = lane & 15, lg = lane >> 4;

    int tm, tn;
    fast_tile(p, ngroups, BN, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = (int)p.M, N = p.N, K = p.K;
    const bf16_t* __restrict__ A = static_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ B = static_cast<const bf16_t*>(p.B);

    // ---- LDS-DMA source addresses: each wave moves PA x 1 KB of A and PB x 1 KB of B per stage.
    // 1 KB = 16 rows x 64 B; lane l lands at row (l>>2), physical chunk (l&3) and therefore fetches
    // logical chunk (l&3) ^ f((l>>4)&3) of that row.
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ swz((lane >> 4) & 3);
    // 32-bit byte offsets from the (wave-uniform) matrix bases: half the address registers of 64-bit pointers, and the
    // loads take the scalar-base + vector-offset form (the launcher checks that both matrices are < 2 GB)
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* Bb = reinterpret_cast<const char*>(B);
    unsigned ga[PA], gb[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int r = (wid * PA + j) * 16 + lrow;
        ga[j] = ((unsigned)rowmap_src(p.amap, min(m0 + r, M - 1), 0, 1) * (unsigned)p.lda + lchunk * 8) * 2u;   // plain / strided / skip-cls rows
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int r = (wid * PB + j) * 16 + lrow;
        gb[j] = ((unsigned)min(n0 + r, N - 1) * (unsigned)p.ldb + lchunk * 8) * 2u;
    }
    // piece q of tile kt: q < PA -> this wave's A row groups, then its B row groups
    auto dma_piece = [&](int kt, int q) __attribute__((always_inline)) {
        char* sb = smem + (kt % STAGES) * STAGE_BYTES;
        const int k0 = kt * BK;
        if (q < PA) __builtin_amdgcn_global_load_lds((gbl_ptr)(Ab + (ga[q] + (unsigned)k0 * 2u)), (lds_ptr)(sb + (wid * PA + q) * 1024), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gbl_ptr)(Bb + (gb[q - PA] + (unsigned)k0 * 2u)), (lds_ptr)(sb + A_BYTES + (wid * PB + (q - PA)) * 1024), 16, 0, 0);
    };
    auto stage_issue = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NP; ++q) dma_piece(kt, q);
    };

    // fragment read offsets (bytes within a stage): row r, logical chunk lg
    int a_off[8], b_off[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = wm * 128 + i * 16 + li;
        a_off[i] = r * 64 + ((lg ^ swz((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + li;
        b_off[j] = A_BYTES + r * 64 + ((lg ^ swz((r >> 2) & 3)) << 4);
    }
    auto frag_read = [&](FragSet& f, int kt) {
        const char* sb = smem + (kt % STAGES) * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) f.a[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[i]);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / BK;                                // >= 4 (checked by the launcher)
    // one pipeline step: `cur` holds tile kt (already in registers); land tile kt+1, start reading it into
    // `nxt`, and multiply tile kt in four groups of 8 MFMAs with ONE LDS-DMA piece of tile kt+3 issued
    // behind each group: the DMA issue cost (~60-180 cycles per piece) hides under the matrix pipe instead
    // of sitting between the barrier and the first MFMA.
    auto step = [&](FragSet& cur, FragSet& nxt, int kt) __attribute__((always_inline)) {
        const bool more = kt + 1 < nk, refill = kt + AHEAD < nk;
        if (more) {
            // DMA groups outstanding here: tiles kt+1 .. kt+AHEAD-1 (NP pieces each); kt+1 must be complete
            if (AHEAD == 3 && kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // tile kt+1 landed for every wave; tile kt-1 no longer read by anyone
            frag_read(nxt, kt + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // 32 MFMAs in NP groups with ONE LDS-DMA piece of tile kt+AHEAD issued behind each group
        // group g multiplies a-fragments [lo(g), lo(g+1)): 2,2,2,2 (NP = 4) or 2,2,1,1,1,1 (NP = 6)
        auto lo = [](int g) { return NP == 4 ? 2 * g : (g < 2 ? 2 * g : g + 2); };
#pragma unroll
        for (int g = 0; g < NP; ++g) {
#pragma unroll
            for (int i = lo(g); i < lo(g + 1); ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.b[j], cur.a[i], acc[i][j], 0, 0, 0);   // swapped: D[n][m]
            __builtin_amdgcn_sched_barrier(0);
            if (refill) dma_piece(kt + AHEAD, g);         // overwrites the stage of tile kt-1
            __builtin_amdgcn_sched_barrier(0);
        }
        // the LDS reads of tile kt+1 are waited for HERE, behind the MFMAs of tile kt (the empty asm consumes the
        // registers, so hipcc places its lgkmcnt wait at this point and treats them as ready afterwards; without
        // it the loop-carried reads make it wait lgkmcnt(0) in front of the MFMAs, serialising read and multiply)
        asm volatile("" : "+v"(nxt.a[0]), "+v"(nxt.a[1]), "+v"(nxt.a[2]), "+v"(nxt.a[3]), "+v"(nxt.a[4]), "+v"(nxt.a[5]),
                          "+v"(nxt.a[6]), "+v"(nxt.a[7]), "+v"(nxt.b[0]), "+v"(nxt.b[1]), "+v"(nxt.b[2]), "+v"(nxt.b[3]));
    };

#pragma unroll
    for (int st = 0; st < AHEAD; ++st) stage_issue(st);
    if (AHEAD == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // tile 0 landed (this wave's share)
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    static_assert((AHEAD == 3 && NP == 4) || (AHEAD == 2 && NP == 6), "the counted waits above are written for these two shapes");
    __builtin_amdgcn_s_barrier();
    FragSet f0, f1;
    frag_read(f0, 0);
    asm volatile("" : "+v"(f0.a[0]), "+v"(f0.a[1]), "+v"(f0.a[2]), "+v"(f0.a[3]), "+v"(f0.a[4]), "+v"(f0.a[5]),
                      "+v"(f0.a[6]), "+v"(f0.a[7]), "+v"(f0.b[0]), "+v"(f0.b[1]), "+v"(f0.b[2]), "+v"(f0.b[3]));
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        step(f0, f1, kt);
        step(f1, f0, kt + 1);
    }
    if (kt < nk) step(f0, f1, kt);

    fast_epilogue(p, acc, smem, wid, lane, m0, n0, wm, wn);
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 main loop, two wave groups half a phase apart (cdna_hip_programming.md §5 "8-phase" schedule), fragments
// read ONE PHASE AHEAD of the MFMAs that consume them.
//
// In the kernel above the two waves of a SIMD run the same instruction stream in step, so both sit in their LDS-DMA issue
// and fragment reads at the same time and the matrix pipe idles (profiles/r01_gemm_ablation.md: the loop ran at ~57 % MFMA
// issue).  Here waves 0-3 (group 0) and waves 4-7 (group 1, the SIMD partners of 0-3) are offset by ONE barrier: every
// barrier interval has one group in a "load" section (fragment ds_reads + 2 LDS-DMA pieces + a counted DMA wait) and the
// other in a "multiply" section (16 MFMAs on one quadrant of its 128 x 64 sub-tile).  The fragments a multiply section
// uses were read in the load section BEFORE the previous multiply section, so their LDS latency lies under 16 MFMAs
// instead of between the barrier and the first MFMA (first version of this loop: reads in the same phase, 380 cycles per
// interval for 256 cycles of MFMA; profiles/r02_fast_gemm_two_group.md).
//
// LDS: two K-tile buffers of 64 KB, i.e. four half-tiles of 128 rows x 128 B each (the two buffers of one half-tile are
// neighbours: A-h0[0], A-h0[1], A-h1[0], A-h1[1] in the first 64 KB, the B half-tiles in the second 64 KB, so ONE base
// register per k-step reaches both buffers of an operand through the 16-bit ds_read immediate):
//   A-h0 / A-h1: for wave row wr, rows wr*128 + q*64 + [0, 64)  (local row wr*64 + rr)  -> quadrant q of every wave
//   B-h0 / B-h1: for wave col wc, cols wc*64 + q*32 + [0, 32)   (local row wc*32 + rr)
// so a half-tile is read in exactly ONE phase of a K-tile and is free for re-staging right after it.  K-tile t, buffer t & 1:
//   phase 0: MFMA A0.B0   load section: read B-h1(t)   -> SB[~t&1]   stage A-h0(t+2)
//   phase 1: MFMA A0.B1                 read A-h1(t)   -> FA[1]      stage B-h0(t+2)
//   phase 2: MFMA A1.B1                 read A-h0(t+1) -> FA[0]      stage B-h1(t+2)
//   phase 3: MFMA A1.B0                 read B-h0(t+1) -> SB[~t&1]   stage A-h1(t+2)
// (B0 of tile t lives in register set SB[t & 1], B1 in the other one: the set B1 leaves after phase 2 receives the next
// tile's B0.)  16-byte chunk c of local row r lives at chunk c ^ ((r >> 1) & 7): a 16-lane ds_read_b128 service group then
// touches 16 distinct slots of the 256-B bank row (rows are 128 B); the LDS-DMA image is lane-linear, so the same
// involution sits on the per-lane SOURCE offset.
//
// Ordering rules (L(g) / M(g) = load / multiply section of global phase g; group 0 runs L(g) in barrier interval 2g and
// M(g) in 2g+1, group 1 one interval later):
//   RAW: a half-tile is read in L(g) only if every wave waited for its own pieces of it (counted vmcnt) at the end of
//        L(g-1) or earlier - that wait is followed by a barrier both groups pass before any read of L(g);
//   WAR: every wave retires the reads of L(g) (lgkmcnt(0)) at the END of M(g), in front of the barrier that closes it, so
//        a half-tile last read in L(g) may be re-staged in L(g+2) by either group.
// Per-wave LDS-DMA issue order is A-h0, B-h0, B-h1, A-h1 of a tile, two pieces each, tile t+2 during tile t: `vmcnt(10)`
// at the end of every load section leaves five half-tiles in flight and retires exactly the one the NEXT load section reads.
//
// The pieces are buffer loads (resource descriptor + per-lane offset VGPR + scalar offset): the K-tile and quadrant offsets
// are scalar, rows beyond N read as zero by the descriptor's bounds check, and the load sections contain no VALU.
constexpr int P8_BK = 64;
constexpr int P8_HALF = 128 * P8_BK * 2;                  // 16 KB
constexpr int P8_BUF = P8_HALF;                           // second buffer of a half-tile: right behind the first
constexpr int P8_A0 = 0, P8_A1 = 2 * P8_HALF, P8_B0 = 4 * P8_HALF, P8_B1 = 6 * P8_HALF;
constexpr int P8_LDS = 8 * P8_HALF;                       // 128 KB
constexpr int P8_AUX = P8_LDS;                            // + 4 KB behind the ring in the bf16 kernel: the epilogue's vectors

template <int N> DEV void wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}
// the compiler may treat these registers as ready from here on (the hand-placed lgkmcnt(0) in front retired their ds_reads;
// without this it re-waits lgkmcnt(0) in front of the consuming MFMAs, behind the NEXT load section's reads)
DEV void regs_ready(bf16x8 (&f)[4][2]) {
    asm volatile("" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1]));
}
DEV void regs_ready(bf16x8 (&f)[2][2]) { asm volatile("" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1])); }
// fp8: a fragment is ONE 32-byte MFMA operand (8 consecutive VGPRs, filled by two 16-byte ds_reads)
typedef int v8i32_t __attribute__((ext_vector_type(8)));
typedef int v4i32_t __attribute__((ext_vector_type(4)));
DEV void regs_ready(v8i32_t (&f)[4][1]) { asm volatile("" : "+v"(f[0][0]), "+v"(f[1][0]), "+v"(f[2][0]), "+v"(f[3][0])); }
DEV void regs_ready(v8i32_t (&f)[2][1]) { asm volatile("" : "+v"(f[0][0]), "+v"(f[1][0])); }

// FP8 (DIST_EPI_FP8): the same bytes move the same way - a K-tile is still 128 BYTES per row, now 128 e4m3 elements - and a lane's
// two 16-byte fragment reads of a row (chunks lg and 4 + lg) are the 32 k-values of ONE block-scaled MFMA
// (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales: twice the bf16 rate): 8 MFMAs per phase instead of 16, each K-tile twice as
// deep.  Which k a (lane, byte) stands for is free as long as A and B agree, and both come through the same swizzle.  The fp32
// per-row scales of A and B multiply the accumulators in front of the shared epilogue.
DEV v8i32_t fp8_operand(const v4i32_t& lo, const v4i32_t& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

template <bool FP8, int CF = -1>
__global__ __launch_bounds__(512, 1) void gemm_fast8p_kernel(const dist_gemm_args p, const int ngroups, const TileMap tmap) {
    constexpr int BN = 256;
    constexpr unsigned ES = FP8 ? 1u : 2u;                // bytes per operand element
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;                // 2 x 4 waves; wr is also the wave group
    const int li = lane & 15, lg = lane >> 4;

    int tm, tn;
    if (DIST_AB && (ngroups & 0x200000)) fast_tile(p, ngroups, BN, tm, tn);      // (timing-only library: the divisions in the kernel)
    else fast_tile_mapped(tmap, tm, tn);
    if (DIST_AB && ((ngroups >> 22) & 63) && blockIdx.x < 256) {
        // timing-only library (DIST_AMD_FAST_SKEW, round 6): the column tiles of one row panel start together, so ALL of them see the fabric latency of the panel's
        // A pieces (the L2 merges their requests, it does not shorten them).  Column tile c of its group starts c x skew x 64 cycles late in the first round -
        // later rounds inherit the order - so that the followers find the leader's lines in L2.
        const int sk = (ngroups >> 22) & 63;
        const int col = tmap.ngroups <= 1 ? tn : (tn < tmap.gr * (tmap.gq + 1) ? tn % (tmap.gq + 1) : (tn - tmap.gr * (tmap.gq + 1)) % max(tmap.gq, 1));
        for (int i = 0; i < col * sk; ++i) __builtin_amdgcn_s_sleep(1);
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = (int)p.M, N = p.N, K = p.K;

    // ---- LDS-DMA sources.  Per half-tile this wave moves local rows 16*wid + 8*j + (lane >> 3), j = 0, 1 (1 KB each).
    // Per-lane byte offsets for piece j of quadrant 0 (plain row maps only: the launcher checks); quadrant 1 and the K-tile
    // are scalar offsets; both descriptors end behind the last row, so the rows of a ragged last row tile / a half-empty
    // column tile read as zero instead of being clamped.
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, M * p.lda * (int)ES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, N * p.ldb * (int)ES, 0x00020000);
    unsigned ga[2], gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int lr = 16 * wid + 8 * j + (lane >> 3);
        const int lc = (lane & 7) ^ ((lr >> 1) & 7);
        ga[j] = (unsigned)(m0 + (lr >> 6) * 128 + (lr & 63)) * (unsigned)p.lda * ES + lc * 16;
        gb[j] = (unsigned)(n0 + (lr >> 5) * 64 + (lr & 31)) * (unsigned)p.ldb * ES + lc * 16;
    }
    const int aq1 = 64 * p.lda * (int)ES, bq1 = 32 * p.ldb * (int)ES;  // quadrant 1: 64 rows of A / 32 rows of B further
    // stage half-tile `slot` (P8_A0 ...) of K-tile kt into buffer kt & 1
    auto stage = [&](const int slot, const int kt) __attribute__((always_inline)) {
        char* sb = smem + (kt & 1) * P8_BUF + slot + wid * 2048;
        const int k2 = kt * (P8_BK * 2);
        if (slot == P8_A0 || slot == P8_A1) {
            const int so = k2 + (slot == P8_A1 ? aq1 : 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)sb, 16, ga[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)(sb + 1024), 16, ga[1], so, 0, 0);
        } else {
            const int so = k2 + (slot == P8_B1 ? bq1 : 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)sb, 16, gb[0], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)(sb + 1024), 16, gb[1], so, 0, 0);
        }
    };

    // ---- fragment read offsets inside a half-tile: local row (wr*64 | wc*32) + f*16 + li, logical chunk kk*4 + lg
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int pc = ((kk * 4 + lg) ^ ((li >> 1) & 7)) << 4;
        a_rd[kk] = (wr * 64 + li) * 128 + pc;
        b_rd[kk] = P8_B0 + (wc * 32 + li) * 128 + pc;
    }
    using frag_t = std::conditional_t<FP8, v8i32_t, bf16x8>;
    constexpr int KK = FP8 ? 1 : 2;                       // fragments per row block and K-tile
    frag_t fa[2][4][KK], sb_[2][2][KK];
    auto read_a = [&](frag_t (&f)[4][KK], const int buf, const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* q = smem + (slot + buf * P8_BUF + i * 2048);
            if constexpr (FP8) f[i][0] = fp8_operand(*reinterpret_cast<const v4i32_t*>(q + a_rd[0]), *reinterpret_cast<const v4i32_t*>(q + a_rd[1]));
            else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) f[i][kk] = *reinterpret_cast<const bf16x8*>(q + a_rd[kk]);
            }
        }
    };
    auto read_b = [&](frag_t (&f)[2][KK], const int buf, const int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const char* q = smem + (slot - P8_B0 + buf * P8_BUF + j * 2048);
            if constexpr (FP8) f[j][0] = fp8_operand(*reinterpret_cast<const v4i32_t*>(q + b_rd[0]), *reinterpret_cast<const v4i32_t*>(q + b_rd[1]));
            else {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) f[j][kk] = *reinterpret_cast<const bf16x8*>(q + b_rd[kk]);
            }
        }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // multiply section: barrier (this wave's counted DMA wait is behind it), 16 MFMAs on fragments read a phase ago,
    // retire the reads of the load section in front (they have had the 16 MFMAs to complete), barrier
    auto mma16q = [&](const int qa, const int qb, const frag_t (&a)[4][KK], const frag_t (&b)[2][KK]) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FP8) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)                  // e4m3 x e4m3, block scales 2^0 (E8M0 127 in every byte); swapped: D[n][m]
                    acc[qa * 4 + i][qb * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        b[j][0], a[i][0], acc[qa * 4 + i][qb * 2 + j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            // pin the eight results to this phase: without it hipcc sinks all 64 scaled MFMAs of the loop body behind its last
            // barrier (they have no side effects and their results are only used a K-tile later) and spills the accumulators
            asm volatile("" : "+v"(acc[qa * 4][qb * 2]), "+v"(acc[qa * 4][qb * 2 + 1]), "+v"(acc[qa * 4 + 1][qb * 2]), "+v"(acc[qa * 4 + 1][qb * 2 + 1]),
                              "+v"(acc[qa * 4 + 2][qb * 2]), "+v"(acc[qa * 4 + 2][qb * 2 + 1]), "+v"(acc[qa * 4 + 3][qb * 2]), "+v"(acc[qa * 4 + 3][qb * 2 + 1]));
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[qa * 4 + i][qb * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][kk], a[i][kk], acc[qa * 4 + i][qb * 2 + j], 0, 0, 0);   // swapped: D[n][m]
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto close_phase = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // one K-tile = four phases; the tail is selected by wave-uniform scalars (s1 = tile kt+1 exists, s2 = tile kt+2 exists)
    // instead of peeled copies of the body: peeled copies made hipcc rename accumulators across their joins and spill
    // INSIDE the loop - and a scratch reload is a vmcnt(0).
    const int nk = K / (FP8 ? 2 * P8_BK : P8_BK);         // >= 2 (checked by the launcher); a K-tile is 128 bytes per row
    constexpr bool RESPF = !FP8 && CF >= 0 && (CF & DIST_EPI_RES) != 0;     // residual epilogue: half of the residual tile is prefetched by the last K-tile
    auto lane_now = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto ktile = [&](auto buf_c, const int kt) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, NB = BUF ^ 1;
        const bool s1 = kt + 1 < nk, s2 = kt + 2 < nk;
        // phase 0
        read_b(sb_[NB], BUF, P8_B1);
        __builtin_amdgcn_sched_barrier(0);
        if (s2) { stage(P8_A0, kt + 2); wait_vm<10>(); } else if (s1) wait_vm<8>(); else {
            wait_vm<0>();
            if constexpr (RESPF) {
                // the LAST K-tile: buffer NB is read no more - rows 0 ... 63 of this wave's residual tile go there now (8 pieces of 8 rows x 128 B, the
                // staging layout of fast_epilogue: 16-byte chunk c of row r at chunk c ^ (r & 7)), a K-tile ahead of the epilogue that adds them
                const int l = lane_now();
                const int pcrow = l >> 3, pc = (l & 7) ^ pcrow;
                const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.res), 0, n0 + wc * 64 < N ? M * p.ldres * 2 : 0, 0x00020000);
                const unsigned offr = (unsigned)(m0 + wr * 128 + pcrow) * (unsigned)p.ldres * 2u + (unsigned)(n0 + wc * 64 + pc * 8) * 2u;
                const int stepr = 8 * p.ldres * 2;
                char* d = smem + (wid >> 1) * (2 * P8_HALF) + NB * P8_BUF + (wid & 1) * (P8_HALF / 2);
#pragma unroll
                for (int k = 0; k < 8; ++k) __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr)(d + k * 1024), 16, offr, k * stepr, 0, 0);
            }
        }
        mma16q(0, 0, fa[0], sb_[BUF]);
        regs_ready(sb_[NB]);
        close_phase();
        // phase 1
        read_a(fa[1], BUF, P8_A1);
        __builtin_amdgcn_sched_barrier(0);
        if (s2) { stage(P8_B0, kt + 2); wait_vm<10>(); } else if (s1) wait_vm<6>();
        mma16q(0, 1, fa[0], sb_[NB]);
        regs_ready(fa[1]);
        close_phase();
        // phase 2
        if (s1 || FP8) read_a(fa[0], NB, P8_A0);          // (fp8: unconditional - a stale tile is read behind the last one and never used;
                                                          //  conditional definitions of 8-register fragments made hipcc spill inside the loop)
        __builtin_amdgcn_sched_barrier(0);
        if (s2) { stage(P8_B1, kt + 2); wait_vm<10>(); } else if (s1) wait_vm<4>();
        mma16q(1, 1, fa[1], sb_[NB]);
        regs_ready(fa[0]);
        close_phase();
        // phase 3
        if (s1 || FP8) read_b(sb_[NB], NB, P8_B0);
        __builtin_amdgcn_sched_barrier(0);
        if (s2) { stage(P8_A1, kt + 2); wait_vm<10>(); } else if (s1) wait_vm<2>();
        mma16q(1, 0, fa[1], sb_[BUF]);
        regs_ready(sb_[NB]);
        close_phase();
    };

    // the epilogue's per-column / per-row vectors of this tile go to LDS first (4-byte LDS-DMA, 16 x 256 B over the 8 waves: wave w moves
    // floats (w & 1) * 128 ... + 128 of array w >> 1 = bias | column sums | mean | rstd); they are older than every operand piece, so the
    // first counted wait retires them and the prologue's barrier publishes them.  Entries beyond N / M read as zero (never stored).
    const bool aux_lds = !FP8 && (!DIST_AB || !(ngroups & 0x20000));
    if (aux_lds) {
        const int arr = wid >> 1;
        const int flags = CF >= 0 ? (CF & 0xffff) : p.flags;
        const bool lnf = (flags & DIST_EPI_LNFOLD) != 0;
        if (arr == 0 ? (flags & DIST_EPI_BIAS) != 0 : lnf) {
            const float* st = static_cast<const float*>(p.aux);
            const float* src = arr == 0 ? p.bias : arr == 1 ? p.bias2 : arr == 2 ? st : st + M;
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (arr < 2 ? N : M) * 4, 0x00020000);
            const unsigned v = (unsigned)((arr < 2 ? n0 : m0) + (wid & 1) * 128 + lane) * 4;
            char* d = smem + P8_AUX + arr * 1024 + (wid & 1) * 512;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)d, 4, v, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(d + 256), 4, v, 256, 0, 0);
        }
    }
    // prologue: tiles 0 and 1 in the steady-state issue order; A-h0, B-h0, B-h1 of tile 0 must have landed (five half-tiles behind)
    stage(P8_A0, 0); stage(P8_B0, 0); stage(P8_B1, 0); stage(P8_A1, 0);
    stage(P8_A0, 1); stage(P8_B0, 1); stage(P8_B1, 1); stage(P8_A1, 1);
    wait_vm<10>();
    __builtin_amdgcn_s_barrier();
    read_a(fa[0], 0, P8_A0);
    read_b(sb_[0], 0, P8_B0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    regs_ready(fa[0]); regs_ready(sb_[0]);
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one barrier interval behind group 0
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    int kt = 0;
#pragma unroll 1
    for (; kt + 1 < nk; kt += 2) {
        ktile(I0{}, kt);
        ktile(I1{}, kt + 1);
    }
    if (kt < nk) ktile(I0{}, kt);
    if (wr == 0) __builtin_amdgcn_s_barrier();            // pairs with group 1's extra barrier
    if constexpr (FP8) {                                  // acc[i][j][r]: row mw + i*16 + li, column nw + j*16 + lg*4 + r
        const int mw = m0 + wr * 128, nw = n0 + wc * 64;
        float sb4[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nw + j * 16 + lg * 4 + r;
                sb4[j][r] = n < N ? p.b_scale[n] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sa = p.a_scale[((CF >= 0 ? CF : p.flags) & DIST_EPI_FP8_ASCALAR) ? 0 : min(mw + i * 16 + li, M - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= sa * sb4[j][r];
        }
    }
    fast_epilogue<!FP8, CF>(p, acc, smem, wid, lane, m0, n0, wr, wc, (ngroups & 0x10000) != 0, aux_lds ? smem + P8_AUX : nullptr, (ngroups >> 18) & 7, nk & 1);
}


#ifdef DIST_AMD_MEASURE
#include "measure/gemm_fast8q.inl"      // several tiles per block (measured and rejected, profiles/r05_gemm_pingpong.md)
#endif

}  // namespace

static bool fast_common_ok(const dist_gemm_args* a) {
    if (a->dtype != DIST_BF16 || a->taps != 1) return false;
    if (a->amap.mode != DIST_RM_PLAIN && a->amap.mode != DIST_RM_STRIDED && a->amap.mode != DIST_RM_SKIPCLS) return false;
    if (a->omap.mode != DIST_OM_PLAIN && a->omap.mode != DIST_OM_INSERTCLS && a->omap.mode != DIST_OM_HEADS) return false;
    if (a->flags & DIST_EPI_MULG) return false;
    if (a->flags & DIST_EPI_LNFOLD) {
        if (!a->aux || !a->bias2 || a->amap.mode != DIST_RM_PLAIN) return false;
    } else if (a->bias2) return false;
    if (a->flags & DIST_EPI_ROWSTATS) {
        if (!a->rowstats || !a->C || a->omap.mode != DIST_OM_PLAIN || (a->flags & DIST_EPI_ACT2)) return false;
    }
    if (a->K % BK || a->K < 4 * BK || a->N % 64 || a->M < 1024) return false;
    if (a->lda % 8 || a->ldb % 8 || a->ldc % 8 || a->ldc2 % 8 || a->ldres % 8) return false;
    if (a->flags & DIST_EPI_OUT8) {                       // e4m3 image of the output: plain map, 16-byte rows, not beside a second activated output
        if (!a->C8 || !a->out8_scale || a->ldc8 % 16 || (a->omap.mode != DIST_OM_PLAIN && a->omap.mode != DIST_OM_HEADS) || ((a->flags & DIST_EPI_ACT2) && a->C)) return false;
        if (a->omap.mode == DIST_OM_HEADS && a->ldc8 != 64) return false;
        if (!a->C && (a->flags & DIST_EPI_ROWSTATS)) return false;
    }
    if (a->flags & DIST_EPI_FP8) {                        // e4m3 operands: 16-byte row alignment, whole 128-deep K-tiles, plain row map
        if (!a->a_scale || !a->b_scale || a->amap.mode != DIST_RM_PLAIN || a->K % 128 || a->K < 256 || a->lda % 16 || a->ldb % 16) return false;
        if (a->N < 256 || (a->N % 256 > 0 && a->N % 256 < 128)) return false;
    }
    if (a->omap.mode == DIST_OM_HEADS && ((a->flags & (DIST_EPI_RES | DIST_EPI_ACT2)) || a->omap.p0 < 16)) return false;   // (rows stepped by 8 / 16 tokens)
    const long a_rows = a->amap.mode == DIST_RM_PLAIN ? a->M : 2 * a->M + a->M / 64 + 64;   // generous bound for the strided / skip-cls images
    if (a_rows * a->lda >= (1l << 30) || (long)a->N * a->ldb >= (1l << 30)) return false;     // 32-bit byte offsets
    return true;
}

#ifdef DIST_AMD_MEASURE
#include "measure/gemm_fast8q_launch.inl"
#else
static int try_fast_q(const dist_gemm_args*, int, hipStream_t) { return 0; }
#endif

// 0 = not eligible, 4 / 8 = waves of the variant that takes the shape
static int fast_variant(const dist_gemm_args* a) {
    if (!fast_common_ok(a)) return 0;
    static const int forced = DIST_AB_KNOB("DIST_AMD_FAST_NW", 0);   // A/B: 4 = the two-blocks-per-CU shape (timing-only library)
    // (a half-empty last column tile only pays with a deep K loop - bf16 has the branch-GEMM kernels for the rest; fp8 has no other kernel)
    static const int kmin = DIST_AB_KNOB("DIST_AMD_FAST_KMIN", 768);
    const bool ok8 = a->N >= 256 && !(a->N % 256 > 0 && (a->N % 256 < 128 || (a->K < kmin && !(a->flags & DIST_EPI_FP8))));
    const bool ok4 = a->N >= 128 && a->N % 128 == 0;
    // measured (tools/bench_fastk.py, profiles/r01_fast_gemm_shapes.md): the two-block shape hides ~40 % of the fixed
    // per-block cost but its K loop is ~30 % slower (prefetch depth 2, 1.5x the LDS-DMA pieces per FLOP) - it loses on
    // every ViT shape, so it only runs when asked for
    if (forced == 4 && ok4 && !(a->flags & DIST_EPI_FP8)) return 4;     // (the e4m3 mode exists in the 8-wave kernel only)
    return ok8 ? 8 : 0;
}

bool dist_k_gemm_fast_eligible(const dist_gemm_args* a) { return fast_variant(a) != 0; }

template <int NW>
static int launch_fast(const dist_gemm_args* a, hipStream_t s) {
    using S = Shape<NW>;
    constexpr size_t smem = (size_t)S::STAGES * S::STAGE_BYTES;
    static_assert(NW * EPI_BYTES <= S::STAGES * S::STAGE_BYTES, "epilogue staging fits in the operand ring");
    static DistSmemOnce attr;
    RUN_(dist_max_smem(attr, reinterpret_cast<const void*>(gemm_fast_kernel<NW>), smem));
    const long tiles = ((a->M + BM - 1) / BM) * ((a->N + S::BN - 1) / S::BN);
    // column-tile groups: as few groups as keep one group's weight rows under ~2.5 MB (DIST_AMD_FAST_NG forces a count; 1 = off)
    static const int forced_ng = DIST_AB_KNOB("DIST_AMD_FAST_NG", 0);
    const int tiles_n = (a->N + S::BN - 1) / S::BN;
    int ng = forced_ng > 0 ? forced_ng : 1;
    if (forced_ng == 0) {
        // fabric bytes of the launch ~ ng x (activation bytes) + 8 XCDs x (weight bytes) when a group's weight rows stay in L2,
        // and ~7 re-fetches of W per XCD when they do not (QKV measured: 276 MB = 77 + 8 x 7 x 3.5, profiles/r01_pmc_fast_gemm.md)
        const double es = (a->flags & DIST_EPI_FP8) ? 1.0 : 2.0;
        const double A = (double)a->M * a->K * es, W = (double)a->N * a->K * es;
        double best = A + 8.0 * W * (W <= 2.5e6 ? 1.0 : 7.0);
        for (int g = 2; g <= 4 && g <= tiles_n; ++g) {
            const double wg = (double)((tiles_n + g - 1) / g) * S::BN * a->K * es;
            if (wg > 2.5e6) continue;
            const double cost = g * A + 8.0 * W;
            if (cost < 0.85 * best) { best = cost; ng = g; }
        }
    }
    if (ng > tiles_n) ng = tiles_n;
    if (ng < 1) ng = 1;
    // two-group 256x256x64 main loop (gemm_fast8p_kernel) when K is a multiple of 64; DIST_AMD_FAST_8P=0 keeps the
    // lock-step 256x256x32 loop (measurement knob, and the A/B reference of tools/bench_fast8p.py)
    static const bool use_8p = (dist_knob("DIST_AMD_FAST_8P", 1) != 0);
    const TileMap tmap = make_tile_map(a->M, a->N, 256, ng);
    constexpr size_t smem8 = (size_t)P8_LDS;
    static_assert(8 * EPI_BYTES <= P8_LDS, "epilogue staging fits in the operand buffers");
    if (a->flags & DIST_EPI_FP8) {                        // fast_common_ok checked the shape; only the two-group loop has the fp8 MFMAs
        if (NW != 8) return DIST_ERR_ARG;
#ifdef DIST_AMD_MEASURE
        if (DIST_AB_KNOB("DIST_AMD_FAST_DBG", 0) & 4) {          // timing-only library: which epilogue shapes a workload launches
            static int seen[64]; static int nseen = 0;
            const int k2 = a->flags ^ (int)(a->N * 131 + a->K) ^ (a->omap.mode << 20) ^ (a->C ? 1 << 24 : 0) ^ (a->C2 ? 1 << 25 : 0);
            bool found = false;
            for (int i = 0; i < nseen; ++i) found |= seen[i] == k2;
            if (!found && nseen < 64) { seen[nseen++] = k2; fprintf(stderr, "[fast8p fp8] M=%ld N=%d K=%d flags=0x%x omap=%d C=%d C2=%d C8=%d\n", (long)a->M, a->N, a->K, a->flags, a->omap.mode, a->C != nullptr, a->C2 != nullptr, a->C8 != nullptr); }
        }
#endif
        // the e4m3 ViT's steady-state epilogues (BASELINE config 5: every GEMM of a block on e4m3 images with one scale per tensor) as straight-line
        // instantiations, like the bf16 kernel's below
        static const bool spec8 = DIST_AB_KNOB("DIST_AMD_FAST_SPEC", 1) != 0;
        const bool heads8 = a->omap.mode == DIST_OM_HEADS;
        const long ldo8 = a->C ? a->ldc : a->ldc2;
        const bool ok8s = (a->omap.mode == DIST_OM_PLAIN || heads8) && a->N % 64 == 0 && (a->flags & DIST_EPI_OUT8) && a->M * (long)a->ldc8 < (1L << 31) &&
                          (heads8 ? (a->M % a->omap.p0 == 0 && a->M * (long)a->omap.p1 * 384 < (1L << 31)) : ((!a->C && !a->C2) || a->M * ldo8 * 2 < (1L << 31))) &&
                          (!(a->flags & DIST_EPI_RES) || a->M * (long)a->ldres * 2 < (1L << 31));
        const int key8 = (spec8 && ok8s) ? ((a->flags & 0xffff) | (heads8 ? CF_HEADS : 0) | (a->C ? 0 : CF_NOC)) : -1;
        constexpr int F8 = DIST_EPI_FP8 | DIST_EPI_FP8_ASCALAR | DIST_EPI_OUT8 | DIST_EPI_BIAS;
        constexpr int K8_INPROJ = F8 | DIST_EPI_LNFOLD | CF_HEADS | CF_NOC, K8_FC = F8 | DIST_EPI_LNFOLD | DIST_EPI_ACT2 | CF_NOC,
                      K8_PROJ = F8 | DIST_EPI_RES | DIST_EPI_ROWSTATS;
#define LAUNCH8F_(CFV) do { static DistSmemOnce attr_; RUN_(dist_max_smem(attr_, reinterpret_cast<const void*>(gemm_fast8p_kernel<true, CFV>), smem8)); \
                            hipLaunchKernelGGL((gemm_fast8p_kernel<true, CFV>), dim3((unsigned)tiles), dim3(512), smem8, s, *a, ng, tmap); } while (0)
        switch (key8) {
            case K8_INPROJ: LAUNCH8F_(K8_INPROJ); break;
            case K8_FC: LAUNCH8F_(K8_FC); break;
            case K8_PROJ: LAUNCH8F_(K8_PROJ); break;
            default: LAUNCH8F_(-1);
        }
#undef LAUNCH8F_
        HIP_CHECK_RET(hipGetLastError());
        return 1;
    }
    if (NW == 8 && use_8p && a->K % P8_BK == 0 && a->K >= 2 * P8_BK && a->amap.mode == DIST_RM_PLAIN) {
        // the ping-pong persistent kernel (gemm_pp.hip) takes the frozen-ViT shapes: N % 256 == 0, K >= 768, the ViT's four epilogues
#ifdef DIST_AMD_MEASURE
        const int pp = dist_k_gemm_pp(a, ng, s);          // (timing-only library: csrc/measure/gemm_pp.hip)
        if (pp != 0) return pp;
#endif
        // several tiles per block with the operand stream running through the tile seams (the ViT's in_proj / c_fc epilogues)
        const int q = try_fast_q(a, ng, s);
        if (q != 0) return q;
        static const int late = DIST_AB_KNOB("DIST_AMD_FAST_EARLY_FLUSH", 1) == 0 ? 0x10000 : 0;      // A/B: the primary output's stores behind the last conversion
        static const int gaux = DIST_AB_KNOB("DIST_AMD_FAST_AUX_LDS", 1) == 0 ? 0x20000 : 0;          // A/B: the epilogue's vectors by global loads
        static const int dbg8 = (DIST_AB_KNOB("DIST_AMD_FAST_DBG", 0) & 3) << 18 | (DIST_AB_KNOB("DIST_AMD_FAST_PLAIN_ST", 0) == 1 ? 4 << 18 : 0);                     // timing only: 1 no output stores, 2 no epilogue
        // the ViT's three epilogues as straight-line instantiations (same arithmetic, same order: bit-identical to the generic one)
        static const bool spec = DIST_AB_KNOB("DIST_AMD_FAST_SPEC", 1) != 0;
        // (they address rows through buffer descriptors: N % 64 == 0, every tensor below 2 GB, whole frames in the head-major map)
        const bool heads = a->omap.mode == DIST_OM_HEADS;
        const long ld_out = a->C ? a->ldc : a->ldc2;
        const bool spec_ok = (a->omap.mode == DIST_OM_PLAIN || heads) && a->N % 64 == 0 &&
                             (heads ? (a->M % a->omap.p0 == 0 && a->M * (long)a->omap.p1 * 384 < (1L << 31)) : a->M * ld_out * 2 < (1L << 31)) &&
                             (!(a->flags & DIST_EPI_RES) || a->M * (long)a->ldres * 2 < (1L << 31));
        const int key = spec_ok ? ((a->flags & 0xffff) | (heads ? CF_HEADS : 0) | (a->C ? 0 : CF_NOC)) : -1;
        constexpr int K_INPROJ = DIST_EPI_BIAS | DIST_EPI_LNFOLD | CF_HEADS, K_FC = DIST_EPI_BIAS | DIST_EPI_LNFOLD | DIST_EPI_ACT2 | CF_NOC,
                      K_PROJ = DIST_EPI_BIAS | DIST_EPI_RES | DIST_EPI_ROWSTATS;       // the frozen ViT's in_proj, c_fc, out_proj / c_proj
        constexpr int K_NONE = 0, K_BIAS = DIST_EPI_BIAS, K_BIASRES = DIST_EPI_BIAS | DIST_EPI_RES;   // the branch's plain Linears (input_linear, data gradients)
        static const int kdiv = DIST_AB_KNOB("DIST_AMD_FAST_TILEMAP", 1) == 0 ? 0x200000 : 0;       // A/B: the tile map's divisions in the kernel
        static const int plain_st = DIST_AB_KNOB("DIST_AMD_FAST_PLAIN_ST", 0);     // 1: every output without the streaming hint; 2 / 3 / 4: only in_proj's / c_fc's / the residual GEMMs'
        static const int skew = (DIST_AB_KNOB("DIST_AMD_FAST_SKEW", 0) & 63) << 22;                 // A/B: column tiles of a row panel start skew x 64 cycles apart (first round)
        const int na = ng | late | gaux | dbg8 | kdiv | skew | (((plain_st == 2 && key == K_INPROJ) || (plain_st == 3 && key == K_FC) || (plain_st == 4 && key == K_PROJ)) ? 4 << 18 : 0);
#ifdef DIST_AMD_MEASURE
        if (DIST_AB_KNOB("DIST_AMD_FAST_DBG", 0) & 4) {          // timing-only library: which epilogue shapes a workload launches
            static int seen[64]; static int nseen = 0;
            const int k2 = key ^ (int)(a->N * 131 + a->K);
            bool found = false;
            for (int i = 0; i < nseen; ++i) found |= seen[i] == k2;
            if (!found && nseen < 64) { seen[nseen++] = k2; fprintf(stderr, "[fast8p] M=%ld N=%d K=%d flags=0x%x omap=%d C=%d C2=%d key=%d\n", (long)a->M, a->N, a->K, a->flags, a->omap.mode, a->C != nullptr, a->C2 != nullptr, key); }
        }
#endif
#define LAUNCH8_(CFV) do { static DistSmemOnce attr_; RUN_(dist_max_smem(attr_, reinterpret_cast<const void*>(gemm_fast8p_kernel<false, CFV>), smem8 + 4096)); \
                           hipLaunchKernelGGL((gemm_fast8p_kernel<false, CFV>), dim3((unsigned)tiles), dim3(512), smem8 + 4096, s, *a, na, tmap); } while (0)
        if (!spec) LAUNCH8_(-1);
        else switch (key) {
            case K_INPROJ: LAUNCH8_(K_INPROJ); break;
            case K_FC: LAUNCH8_(K_FC); break;
            case K_PROJ: LAUNCH8_(K_PROJ); break;
            case K_NONE: LAUNCH8_(K_NONE); break;
            case K_BIAS: LAUNCH8_(K_BIAS); break;
            case K_BIASRES: LAUNCH8_(K_BIASRES); break;
            default: LAUNCH8_(-1);
        }
#undef LAUNCH8_
        HIP_CHECK_RET(hipGetLastError());
        return 1;
    }
    hipLaunchKernelGGL(gemm_fast_kernel<NW>, dim3((unsigned)tiles), dim3(S::NT), smem, s, *a, ng);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}

// returns 1 if handled, 0 if the shape does not qualify (caller falls through), <0 on error
int dist_k_gemm_fast(const dist_gemm_args* a, hipStream_t s) {
    const int v = fast_variant(a);
    if (v == 0) return 0;
#ifdef DIST_AMD_MEASURE
    if (v == 4) return launch_fast<4>(a, s);              // (measured and rejected: profiles/r01_fast_gemm_shapes.md, r04_step_knobs.md)
#endif
    return launch_fast<8>(a, s);
}
