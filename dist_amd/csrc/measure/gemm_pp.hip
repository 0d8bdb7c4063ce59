// Ping-pong persistent GEMM for the frozen-ViT shapes (bf16, plain row map, N % 256 == 0, K % 128 == 0, K >= 768):
//   C[m][n] = epi( sum_k A[m][k] * B[n][k] )      reference call sites: clip.py:150-178 (in_proj / out_proj / c_fc / c_proj)
//
// Why: in gemm_fast8p_kernel (one 256 x 256 tile per block, one block per CU) a K = 768 tile is ~13.8 us of K loop and ~10 us of
// prologue latency + store-bound epilogue during which the CU's matrix pipes idle (profiles/r02_fast_gemm_two_group.md): a third of
// every tile is fixed cost.  Here the two 4-wave groups of a block (the SIMD partners) own DIFFERENT half-tiles and swap roles:
//
//   slot s:   group (s & 1)      COMPUTE   K loop of half-tile s (128 rows x 256 columns; wave w: 128 x 64), fragments by
//                                          ds_read_b128 interleaved with its own 64 MFMAs per K-tile, nothing else
//             group (s & 1) ^ 1  SERVICE   (a) every LDS-DMA piece of the compute group's operand stream (3 per wave and phase,
//                                              two K-tiles ahead, counted vmcnt), (b) the epilogue of ITS half-tile s - 1 cut into
//                                              per-phase chunks (residual tile by LDS-DMA, conversion of one accumulator fragment
//                                              per phase through a per-wave 16 KB staging region, 16-byte row stores), (c) in the
//                                              last K-tile the first fragments of its own next half-tile
//
// so the accumulators of one half of the register file drain while the other half fills, the block walks a static list of tiles
// (persistent: the operand stream never stops at a tile seam, so there is no prologue per tile either), and the matrix pipe of a
// SIMD always has one wave in its K loop.  Group 0 owns the top halves (rows 0..127 of a 256-row tile), group 1 the bottom halves:
// every output element sees its K-tiles in the same order as in gemm_fast8p_kernel, so the results are bit-identical to it.
//
// LDS (160 KB): operand ring 96 KB = two buffers x {A-q0, A-q1: 64 rows x 128 B; B-h0, B-h1: 128 rows x 128 B}, buffers of a
// slot adjacent (one base register + 16-bit immediates reach both); staging 4 x 16 KB for the service waves.
//   A-q (quadrant q of the half-tile's 128 rows): local row rr        -> rows row0 + q*64 + rr          (all four waves read it)
//   B-h (half h of every wave's 64 columns):      local row w*32 + rr -> cols n0 + w*64 + h*32 + rr
// 16-byte chunk c of local row r at chunk c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 over 128-byte rows; the same involution
// on the LDS-DMA source offset), as in gemm_fast8p_kernel.
//
// One s_barrier per phase (four phases per K-tile, 16 MFMAs per compute wave each).  Compute K-tile u (buffer u & 1):
//   phase 0: MFMA A0.B0   reads B-h1(u)        service stages B-h0(u+2) pieces 0,1,2
//   phase 1: MFMA A0.B1   reads A-q1(u)                        B-h0(u+2) piece 3, B-h1(u+2) pieces 0,1
//   phase 2: MFMA A1.B1   reads A-q0(u+1)                      B-h1(u+2) pieces 2,3, A-q1(u+2) piece 0
//   phase 3: MFMA A1.B0   reads B-h0(u+1)                      A-q1(u+2) piece 1, A-q0(u+3) pieces 0,1
// Ordering rules:
//   WAR: a slot read in phase P (reads retired by lgkmcnt(0) in front of the barrier that ends P) is re-staged in phase P+1 or later;
//   RAW: a piece issued in phase P is complete at the END of phase P+5 (the issuing wave's `vmcnt(15)`: three pieces per phase, the
//        five newest phases may stay in flight; vmcnt retires in order and also counts the epilogue's stores, which only makes the
//        fixed count more conservative), and every read above sits at least one barrier behind that wait;
//   role swap: the pieces a service wave issued in its last five phases are waited for by the same wave in the first five phases of
//        its compute slot (vmcnt 12, 9, 6, 3, 0); the epilogue's stores are issued at least eight phases before the swap.
#include <stdlib.h>
#include <type_traits>
#include "../common.h"
#include "../kernels.h"

// Round 5 verdict on this design (profiles/r05_gemm_pingpong.md): bit-identical to gemm_fast8p_kernel on every epilogue, and 25-35 % SLOWER
// per launch.  One wave per SIMD issues v_mfma_f32_16x16x32_bf16 at ~27 cycles instead of 16-17 (compute role alone, no DMA, no epilogue:
// 1244-1322 TF against ~1850 TF-equivalent inside the two-waves-per-SIMD loop), and a 128 x 256 half-tile needs 48 KB of operands per
// 64 MFMAs per wave against 64 KB per 128: at the ~50 GB/s per CU that LDS-DMA sustains in these GEMMs (~64 KB in flight per CU over
// 1.2-1.4 us of latency, whatever the ring depth or the waits) the operand stream, not the matrix pipe, bounds the loop.
// The kernel is therefore NOT part of the product library: this file is a source of the timing-only build alone (dist_amd/build.py
// MEASURE_SOURCES; the A/B reference of tools/check_pp.py) and gemm_fast.hip calls dist_k_gemm_pp only under -DDIST_AMD_MEASURE.
#ifndef DIST_AMD_MEASURE
#error "measure/gemm_pp.hip belongs to the timing-only library (python -m dist_amd.build --measure)"
#else

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;

constexpr int PP_BM = 256, PP_BN = 256, PP_BK = 64;
constexpr int PP_AQ = 64 * 128;                           // 8 KB
constexpr int PP_BH = 128 * 128;                          // 16 KB
constexpr int PP_A0 = 0, PP_A1 = 2 * PP_AQ, PP_B0 = 4 * PP_AQ, PP_B1 = 4 * PP_AQ + 2 * PP_BH;
constexpr int PP_RING = 4 * PP_AQ + 4 * PP_BH;            // 96 KB
constexpr int PP_STG = 128 * 128;                         // 16 KB per service wave
constexpr int PP_LDS = PP_RING + 4 * PP_STG;              // 160 KB
constexpr int PP_MIN_NK = 12;                             // the paced epilogue takes the first 42 phases of a slot

enum { PPF_LNFOLD = 1, PPF_RES = 2, PPF_ACT = 4, PPF_HEADS = 8, PPF_ROWSTATS = 16 };

template <int N> DEV void pp_wait_vm() {
    static_assert(N >= 0 && N <= 15, "count");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
}
// loads the compiler must not know as loads (it would wait vmcnt(0) in front of their first use and drain the LDS-DMA pipeline):
// the values are used behind a hand-placed counted wait that also "produces" them (pp_pin)
DEV f32x4 pp_ld16(const float* p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
DEV float pp_ld4(const float* p) { float v; asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
DEV void pp_pin(f32x4 (&a)[4]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])); }
DEV void pp_pin(float (&a)[8]) { asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])); }
DEV void pp_ready(bf16x8 (&f)[4][2]) {
    asm volatile("" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1]));
}
DEV void pp_ready(bf16x8 (&f)[2][2]) { asm volatile("" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1])); }

template <int I, int N, typename Fn> DEV void pp_static_for(Fn&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); pp_static_for<I + 1, N>(f); }
}

// linear tile index (position inside the XCD-contiguous ranges of fast_tile) -> (row tile, column tile): column tiles in `ngroups`
// groups, all row tiles of a group before the next group, the group's column tiles fastest (gemm_fast.hip fast_tile)
DEV void pp_tile(const int tiles_m, const int tiles_n, const int ngroups, const int t, int& tm, int& tn) {
    if (ngroups <= 1) { tm = t / tiles_n; tn = t % tiles_n; return; }
    const int gq = tiles_n / ngroups, gr = tiles_n % ngroups;
    const int big = tiles_m * (gq + 1);
    int g, idg, gsz, g0;
    if (t < gr * big) { g = t / big; idg = t - g * big; gsz = gq + 1; g0 = g * (gq + 1); }
    else { const int b2 = t - gr * big; g = b2 / (tiles_m * gq); idg = b2 - g * (tiles_m * gq); gsz = gq; g0 = gr * (gq + 1) + g * gq; }
    tm = idg / gsz; tn = g0 + idg % gsz;
}

template <int F>
__global__ __launch_bounds__(512, 1) void gemm_pp_kernel(const dist_gemm_args p, const int ngroups, const int dbg) {
    // timing-only switches (WRONG results; -DDIST_AMD_MEASURE builds only): 1 no output stores, 2 no epilogue chunks, 4 no counted DMA waits, 8 no ring DMA
#ifdef DIST_AMD_MEASURE
#define PP_DBG(bit) ((dbg & (bit)) != 0)
#else
#define PP_DBG(bit) false
    (void)dbg;
#endif
    constexpr bool LNF = (F & PPF_LNFOLD) != 0, RES = (F & PPF_RES) != 0, ACT = (F & PPF_ACT) != 0, HEADS = (F & PPF_HEADS) != 0, RST = (F & PPF_ROWSTATS) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wid >> 2, w = wid & 3;                  // wave group (SIMD partners are w and w + 4), wave inside the group
    const int M = (int)p.M, N = p.N;
    const int nk = p.K / PP_BK;                           // even, >= PP_MIN_NK (launcher)

    // ---- this block's tiles: XCD x owns a contiguous range of the tile order, its blocks interleave inside it
    const int tiles_n = N / PP_BN, tiles_m = (M + PP_BM - 1) / PP_BM, nblk = tiles_m * tiles_n;
    const int G8 = (int)gridDim.x >> 3, xcd = (int)blockIdx.x & 7, yb = (int)blockIdx.x >> 3;
    const int q8 = nblk / 8, r8 = nblk % 8;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = xcd < r8 ? q8 + 1 : q8;
    const int ntiles = yb < cnt ? (cnt - yb + G8 - 1) / G8 : 0;
    if (ntiles == 0) return;
    const int H = 2 * ntiles, U = H * nk;                 // half-tiles, K-tiles of this block's operand stream
    auto half_pos = [&](const int h, int& row0, int& n0) __attribute__((always_inline)) {
        int tm, tn;
        pp_tile(tiles_m, tiles_n, ngroups, start + yb + (h >> 1) * G8, tm, tn);
        row0 = tm * PP_BM + (h & 1) * 128; n0 = tn * PP_BN;
    };

    // ---- LDS-DMA sources: per-lane offsets inside a half-tile (rows of a piece x swizzled chunk); half-tile, quadrant and K-tile are scalar
    const unsigned lda2 = (unsigned)p.lda * 2u, ldb2 = (unsigned)p.ldb * 2u;
    // (every lane-constant address below is derived from an OPAQUE copy of the lane id taken at the start of the slot that uses it: hoisted
    //  to the kernel entry they are spilled around the other role's code, and a scratch reload waits vmcnt(0) - the whole ring)
    unsigned va[2], vb[2];
    auto lane_now = [&]() __attribute__((always_inline)) { int l = lane; asm volatile("" : "+v"(l)); return l; };
    auto dma_setup = [&](const int ln) __attribute__((always_inline)) {
        const int lr = ln >> 3, pc = ln & 7;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int lc = pc ^ ((4 * jj + (lr >> 1)) & 7);
            va[jj] = (unsigned)(16 * w + 8 * jj + lr) * lda2 + lc * 16;
            vb[jj] = (unsigned)(64 * w + 8 * jj + lr) * ldb2 + lc * 16;
        }
    };
    const char* Ab = static_cast<const char*>(p.A);
    const char* Bb = static_cast<const char*>(p.B);
    auto rsrc_a = [&](const int row0) __attribute__((always_inline)) {
        const int left = max(M - row0, 0);
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ab + (size_t)row0 * lda2), 0, left * (int)lda2, 0x00020000);
    };
    auto rsrc_b = [&](const int n0) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Bb + (size_t)n0 * ldb2), 0, (N - n0) * (int)ldb2, 0x00020000);
    };
    auto dma_a = [&](const __amdgpu_buffer_rsrc_t r, const int slot, const int jj, const int soff) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(smem + slot + (2 * w + jj) * 1024), 16, va[jj], soff, 0, 0);
    };
    auto dma_b = [&](const __amdgpu_buffer_rsrc_t r, const int slot, const int jj, const int soff) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(smem + slot + (4 * w + jj) * 1024), 16, vb[jj & 1], soff + (jj >> 1) * 16 * (int)ldb2, 0, 0);
    };
    const int aq1 = 64 * (int)lda2, bh1 = 32 * (int)ldb2;

    // ---- fragment read offsets (compute role)
    int a_rd[2], b_rd[2];
    auto frag_setup = [&](const int ln) __attribute__((always_inline)) {
        const int fi = ln & 15, fg = ln >> 4;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int pcs = ((kk * 4 + fg) ^ ((fi >> 1) & 7)) << 4;
            a_rd[kk] = fi * 128 + pcs;
            b_rd[kk] = PP_B0 + (w * 32 + fi) * 128 + pcs;
        }
    };
    bf16x8 fa[2][4][2], sb_[2][2][2];
    auto read_a = [&](bf16x8 (&f)[4][2], const int buf, const int q) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) f[i][kk] = *reinterpret_cast<const bf16x8*>(smem + ((q ? PP_A1 : PP_A0) + buf * PP_AQ + i * 2048) + a_rd[kk]);
    };
    auto read_b = [&](bf16x8 (&f)[2][2], const int buf, const int h) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) f[j][kk] = *reinterpret_cast<const bf16x8*>(smem + ((h ? 2 * PP_BH : 0) + buf * PP_BH + j * 2048) + b_rd[kk]);
    };

    f32x4 acc[8][4];

    auto barrier = [&]() __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // (timing experiment: 16 = a barrier only behind phases 1 and 3, 32 = only behind phase 3)
    auto barrier_ph = [&](const int ph) __attribute__((always_inline)) {
        if (PP_DBG(16) && !(ph & 1)) return;
        if (PP_DBG(32) && ph != 3) return;
        barrier();
    };
    // 16 MFMAs on quadrant (qa, qb); ZERO: the first k-step starts the accumulators (first K-tile of a half-tile)
    auto mma16q = [&](auto zero_c, const int qa, const int qb, const bf16x8 (&a)[4][2], const bf16x8 (&b)[2][2]) __attribute__((always_inline)) {
        constexpr bool ZERO = decltype(zero_c)::value;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 c = (ZERO && kk == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[qa * 4 + i][qb * 2 + j];
                    acc[qa * 4 + i][qb * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j][kk], a[i][kk], c, 0, 0, 0);   // swapped: D[n][m]
                }
    };
    // one MFMA, one LDS read, ... : the reads of a phase go out between its first MFMAs instead of in front of them
    auto interleave = [&](auto nreads_c) __attribute__((always_inline)) {
        constexpr int NR = decltype(nreads_c)::value;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 16 - NR, 0);
    };
    using T_ = std::true_type; using F_ = std::false_type;
    using I0 = std::integral_constant<int, 0>; using I4 = std::integral_constant<int, 4>; using I8 = std::integral_constant<int, 8>;

    // ------------------------------------------------------------------------------------------------ compute role: one K-tile
    // `first` (wave-uniform): the first two K-tiles of a slot wait for the pieces this wave issued in the last five phases of its service
    // slot (12 / 9 / 6 / 3 / 0 may stay in flight).  ONE loop body serves every K-tile: peeled head / tail copies made hipcc rename the
    // accumulators across their joins and spill inside the K loop (a scratch reload is a vmcnt(0)).  The last K-tile of a slot reads
    // "the next K-tile's" A-q0 / B-h0 like every other one: those slots hold the first K-tile of the OTHER group's half-tile, the values
    // are never used (this wave's own next fragments come from its service slot), and reading them is harmless.
    auto ktile = [&](auto buf_c, const bool first) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value, NB = BUF ^ 1;
        using Z = std::false_type;
        // phase 0
        __builtin_amdgcn_sched_barrier(0);
        read_b(sb_[NB], BUF, 1);
        mma16q(Z{}, 0, 0, fa[0], sb_[BUF]);
        interleave(I4{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pp_ready(sb_[NB]);
        if (first) { if constexpr (BUF == 0) pp_wait_vm<12>(); else pp_wait_vm<0>(); }
        barrier_ph(0);
        // phase 1
        read_a(fa[1], BUF, 1);
        mma16q(Z{}, 0, 1, fa[0], sb_[NB]);
        interleave(I8{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pp_ready(fa[1]);
        if constexpr (BUF == 0) { if (first) pp_wait_vm<9>(); }
        barrier_ph(1);
        // phase 2
        read_a(fa[0], NB, 0);
        mma16q(Z{}, 1, 1, fa[1], sb_[NB]);
        interleave(I8{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pp_ready(fa[0]);
        if constexpr (BUF == 0) { if (first) pp_wait_vm<6>(); }
        barrier_ph(2);
        // phase 3
        read_b(sb_[NB], NB, 0);
        mma16q(Z{}, 1, 0, fa[1], sb_[BUF]);
        interleave(I4{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pp_ready(sb_[NB]);
        if constexpr (BUF == 0) { if (first) pp_wait_vm<3>(); }
        barrier_ph(3);
    };

    // ------------------------------------------------------------------------------------------------ service role: the epilogue in chunks
    // (acc[i][j][r]: output row mw + i*16 + li, column nw + j*16 + lg*4 + r; staging: 128 rows x 128 B per wave, 16-byte chunk c of row r at
    //  chunk c ^ (r & 7) - gemm_fast.hip fast_epilogue, same arithmetic in the same order)
    char* const ew = smem + PP_RING + w * PP_STG;
    bf16_t* __restrict__ const Cout = static_cast<bf16_t*>(ACT ? p.C2 : p.C);
    const int ldo = ACT ? p.ldc2 : p.ldc;
    const bf16_t* __restrict__ const Rres = static_cast<const bf16_t*>(p.res);
    int e_mw = 0, e_nw = 0;                               // the half-tile being drained: first row, this wave's first column
    int e_part = 0, e_head = 0;                           // head-major output: this wave's 64 columns are one (q | k | v, head) slice
    f32x4 bias4[4], cs4[4];
    float mean8[8], rstd8[8];
    int hfr = 0, htok = 0;                                // head-major output: (frame, token) of the lane's next row piece
    unsigned vres = 0;                                    // residual tile by LDS-DMA: per-lane source offset inside the wave's 128 x 64 sub-tile
    int e_li = 0, e_lg = 0, e_crow = 0, e_cchunk = 0;     // this slot's lane decomposition (opaque copy)
    int e_cbase[4];                                       // staging offset of fragment (0, j): row li, the lane's 8 bytes of column block j
    int e_fbase = 0;                                      // staging offset of the lane's 16 bytes of row piece 0 (piece `it`: + it * 1024)
    auto epi_begin = [&](const int h) __attribute__((always_inline)) {
        int row0, n0;
        half_pos(h, row0, n0);
        e_mw = row0; e_nw = n0 + w * 64;
        const int ln = lane_now();
        dma_setup(ln);
        e_li = ln & 15; e_lg = ln >> 4; e_crow = ln >> 3; e_cchunk = ln & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) e_cbase[j] = e_li * 128 + ((((j << 1) | (e_lg >> 1)) ^ (e_li & 7)) << 4) + ((e_lg & 1) << 3);
        e_fbase = e_crow * 128 + ((e_cchunk ^ e_crow) << 4);
        if constexpr (RES) vres = (unsigned)e_crow * (unsigned)p.ldres * 2u + (unsigned)((e_cchunk ^ e_crow) * 16);
        if constexpr (HEADS) { e_part = (e_nw >> 6) / p.omap.p1; e_head = (e_nw >> 6) - e_part * p.omap.p1; }
    };
    // aux vectors of the sub-tile: bias / column sums of this lane's 16 columns, statistics of its 8 rows
    auto epi_aux = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = e_nw + j * 16 + e_lg * 4;
            bias4[j] = p.bias ? pp_ld16(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (LNF) cs4[j] = pp_ld16(p.bias2 + n);
        }
        if constexpr (LNF) {
            const float* __restrict__ st = static_cast<const float*>(p.aux);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = min(e_mw + i * 16 + e_li, M - 1);
                mean8[i] = pp_ld4(st + m); rstd8[i] = pp_ld4(st + (long)M + m);
            }
        }
    };
    auto epi_aux_pin = [&]() __attribute__((always_inline)) {
        pp_pin(bias4);
        if constexpr (LNF) { pp_pin(cs4); pp_pin(mean8); pp_pin(rstd8); }
    };
    // residual rows 8*it .. 8*it + 7 of the sub-tile -> staging (rows behind M read as zero: descriptor bounds)
    auto epi_res = [&](const int it0, const int n) __attribute__((always_inline)) {
        if constexpr (RES) {
            const int left = max(M - e_mw, 0);
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(Rres) + ((size_t)e_mw * p.ldres + e_nw) * 2), 0, left * p.ldres * 2, 0x00020000);
#pragma unroll
            for (int it = it0; it < it0 + n; ++it)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_ptr)(ew + it * 1024), 16, vres, it * 8 * p.ldres * 2, 0, 0);
        }
    };
    auto epi_conv = [&](auto i_c, auto j_c) __attribute__((always_inline)) {
        constexpr int i = decltype(i_c)::value, j = decltype(j_c)::value;
        bf16_t* slot = reinterpret_cast<bf16_t*>(ew + i * 2048 + e_cbase[j]);      // row i*16 + li: (row & 7) == (li & 7)
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (LNF) v[q] = lnfold_bias(acc[i][j][q], mean8[i], rstd8[i], cs4[j][q], bias4[j][q]);
            else v[q] = acc[i][j][q] + bias4[j][q];
        }
        if constexpr (RES) {
            float x[4];
            load4(slot, x);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += x[q];
        }
        if constexpr (ACT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = qgelu_t<bf16_t>(v[q]);
        }
        store4(slot, v);
    };
    auto epi_flush = [&](auto it_c) __attribute__((always_inline)) {
        constexpr int it = decltype(it_c)::value;
        const int crow = e_crow, cchunk = e_cchunk;
        const int r = it * 8 + crow;
        const int m = e_mw + r, n = e_nw + cchunk * 8;
        if constexpr (HEADS && it == 0) { hfr = (e_mw + crow) / p.omap.p0; htok = (e_mw + crow) - hfr * p.omap.p0; }
        const uint4 v = *reinterpret_cast<const uint4*>(ew + it * 1024 + e_fbase);          // row it*8 + crow: (row & 7) == crow
        if (m < M && !PP_DBG(1)) {
            if constexpr (HEADS) {
                store16_nt(Cout + ((((long)hfr * p.omap.p1 + e_head) * 3 + e_part) * p.omap.p0 + htok) * 64 + cchunk * 8, v);
            } else store16_nt(Cout + (long)m * ldo + n, v);
        }
        if constexpr (HEADS) { htok += 8; if (htok >= p.omap.p0) { htok -= p.omap.p0; ++hfr; } }
    };
    auto epi_rowstats = [&](const int h2) __attribute__((always_inline)) {
        if constexpr (RST) {
            const int r = h2 * 64 + e_crow * 8 + e_cchunk;
            float rs = 0.f, rq = 0.f;
#pragma unroll
            for (int c0 = 0; c0 < 8; c0 += 4) {                // (chunks and elements in the order of gemm_fast.hip: same sums)
                bf16x8 v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const bf16x8*>(ew + r * 128 + (((c0 + c) ^ (r & 7)) << 4));
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float x = (float)v[c][e]; rs += x; rq += x * x; }
            }
            if (e_mw + r < M) *reinterpret_cast<float2*>(p.rowstats + ((long)(e_nw >> 6) * M + e_mw + r) * 2) = make_float2(rs, rq);
        }
    };
    // chunk of slot phase P (compile time): 0..3 aux + residual, 6..37 one accumulator fragment each, row pieces of a row block behind
    // its conversion (10, 11, 14, 15, ... 38, 39), row statistics 40, 41
    auto epi_chunk = [&](auto P_c) __attribute__((always_inline)) {
        constexpr int P = decltype(P_c)::value;
        if (PP_DBG(2)) return;
        if constexpr (P == 0) epi_aux();
        if constexpr (P < 4) epi_res(P * 4, 4);
        if constexpr (P == 6) epi_aux_pin();
        if constexpr (P >= 6 && P < 38) epi_conv(std::integral_constant<int, ((P - 6) >> 2)>{}, std::integral_constant<int, ((P - 6) & 3)>{});
        if constexpr (P >= 10 && P < 40 && ((P - 10) & 3) < 2) epi_flush(std::integral_constant<int, 2 * ((P - 10) >> 2) + ((P - 10) & 3)>{});
        if constexpr (P == 40) epi_rowstats(0);
        if constexpr (P == 41) epi_rowstats(1);
    };

    // ------------------------------------------------------------------------------------------------ service role: operand stream
    // stream position x2 = u + 2 (B-h0, B-h1, A-q1) and x3 = u + 3 (A-q0): half-tile descriptors + K-tile byte offsets
    int u = 0;                                            // compute K-tile of the current step (global over the block's half-tiles)
    int kt2 = 2, kt3 = 3, h2 = 0, h3 = 0;                 // (nk >= 12: both start inside half-tile 0)
    __amdgpu_buffer_rsrc_t ra2, rb2, ra3;
    {
        int row0, n0;
        half_pos(0, row0, n0);
        ra2 = rsrc_a(row0); rb2 = rsrc_b(n0); ra3 = ra2;
    }
    // behind the end of the stream the pieces are still issued - through descriptors of zero records (they move no data and land zeros in
    // slots nobody reads) - so that every phase carries exactly three ring pieces per wave and the counted waits are constants
    auto svc_wait = [&](const int) __attribute__((always_inline)) { if (!PP_DBG(4)) pp_wait_vm<15>(); };
    auto svc_dma = [&](auto ph_c) __attribute__((always_inline)) {
        constexpr int ph = decltype(ph_c)::value;
        const int b2 = u & 1, b3 = b2 ^ 1;
        const int k2 = kt2 * (PP_BK * 2), k3 = kt3 * (PP_BK * 2);
        if (PP_DBG(8)) return;
        if constexpr (ph == 0) { dma_b(rb2, PP_B0 + b2 * PP_BH, 0, k2); dma_b(rb2, PP_B0 + b2 * PP_BH, 1, k2); dma_b(rb2, PP_B0 + b2 * PP_BH, 2, k2); }
        if constexpr (ph == 1) { dma_b(rb2, PP_B0 + b2 * PP_BH, 3, k2); dma_b(rb2, PP_B1 + b2 * PP_BH, 0, k2 + bh1); dma_b(rb2, PP_B1 + b2 * PP_BH, 1, k2 + bh1); }
        if constexpr (ph == 2) { dma_b(rb2, PP_B1 + b2 * PP_BH, 2, k2 + bh1); dma_b(rb2, PP_B1 + b2 * PP_BH, 3, k2 + bh1); dma_a(ra2, PP_A1 + b2 * PP_AQ, 0, k2 + aq1); }
        if constexpr (ph == 3) { dma_a(ra2, PP_A1 + b2 * PP_AQ, 1, k2 + aq1); dma_a(ra3, PP_A0 + b3 * PP_AQ, 0, k3); dma_a(ra3, PP_A0 + b3 * PP_AQ, 1, k3); }
    };
    auto stream_src2 = [&]() __attribute__((always_inline)) {
        int row0 = M, n0 = N;                              // behind the last half-tile: zero records
        if (h2 < H) half_pos(h2, row0, n0);
        ra2 = rsrc_a(row0); rb2 = rsrc_b(n0);
    };
    auto stream_src3 = [&]() __attribute__((always_inline)) {
        int row0 = M, n0 = N;
        if (h3 < H) half_pos(h3, row0, n0);
        ra3 = rsrc_a(row0);
    };
    // end of a step: advance both stream positions (a wrap enters the next half-tile: new descriptors)
    auto svc_advance = [&]() __attribute__((always_inline)) {
        ++u;
        if (++kt2 == nk) {
            kt2 = 0; ++h2;
            stream_src2();
        }
        if (++kt3 == nk) {
            kt3 = 0; ++h3;
            stream_src3();
        }
    };
    // the compute role does not issue pieces but must keep the same stream state for its next service slot
    auto stream_skip = [&](const int steps) __attribute__((always_inline)) {
        u += steps;
        // positions after `steps` K-tiles: x2 = u + 2, x3 = u + 3
        const int x2 = u + 2, x3 = u + 3;
        h2 = x2 / nk; kt2 = x2 - h2 * nk; h3 = x3 / nk; kt3 = x3 - h3 * nk;
        stream_src2(); stream_src3();
    };

    // one service K-tile; KT < 12: the epilogue chunks of phases 4*KT .. 4*KT + 3 (compile time), else none
    auto svc_ktile = [&](auto kt_c, auto epi_c, auto last_c) __attribute__((always_inline)) {
        constexpr int KT = decltype(kt_c)::value;          // 12 = no chunks
        constexpr bool has_epi = decltype(epi_c)::value, own_next = decltype(last_c)::value;
        // (first K-tile: the aux loads / residual pieces of a phase go out IN FRONT of its ring pieces, so that the count of phase 5 holds)
        // phase 0
        if constexpr (KT == 0) { if constexpr (has_epi) epi_chunk(std::integral_constant<int, 0>{}); }
        svc_dma(I0{});
        if constexpr (KT > 0 && KT < 12) { if constexpr (has_epi) epi_chunk(std::integral_constant<int, 4 * KT + 0>{}); }
        svc_wait(0);
        barrier_ph(0);
        // phase 1
        if constexpr (KT == 0) { if constexpr (has_epi) epi_chunk(std::integral_constant<int, 1>{}); }
        svc_dma(std::integral_constant<int, 1>{});
        if constexpr (KT > 0 && KT < 12) { if constexpr (has_epi) epi_chunk(std::integral_constant<int, 4 * KT + 1>{}); }
        if constexpr (KT == 1) {
            // phase 5 of the slot: everything up to the last residual piece (issued at the head of phase 3) has landed - behind it in the
            // queue: the ring pieces of phases 3, 4, 5
            if (!PP_DBG(4)) pp_wait_vm<9>();
        } else svc_wait(1);
        barrier_ph(1);
        // phase 2 (last K-tile of the slot: the first A fragments of this group's next half-tile)
        if constexpr (KT == 0) { if constexpr (has_epi) epi_chunk(std::integral_constant<int, 2>{}); }
        svc_dma(std::integral_constant<int, 2>{});
        if constexpr (KT > 0 && KT < 12) { if constexpr (has_epi) epi_chunk(std::integral_constant<int, 4 * KT + 2>{}); }
        if constexpr (own_next) { frag_setup(lane_now()); read_a(fa[0], 0, 0); }
        svc_wait(2);
        if constexpr (own_next) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pp_ready(fa[0]); }
        barrier_ph(2);
        // phase 3
        if constexpr (KT == 0) { if constexpr (has_epi) epi_chunk(std::integral_constant<int, 3>{}); }
        svc_dma(std::integral_constant<int, 3>{});
        if constexpr (KT > 0 && KT < 12) { if constexpr (has_epi) epi_chunk(std::integral_constant<int, 4 * KT + 3>{}); }
        if constexpr (own_next) { read_b(sb_[0], 0, 0); }
        svc_wait(3);
        if constexpr (own_next) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); pp_ready(sb_[0]); }
        barrier_ph(3);
        svc_advance();
    };

    // one service slot (EPI: with the chunks of this group's previous half-tile; own_nxt: this group computes the next half-tile)
    // (`last`: the slot's last K-tile also reads the first fragments of this group's next half-tile - unconditionally, so that the
    //  fragment registers are defined on every path out of a service slot and not live across it; behind the block's last half-tile the
    //  values are never used)
    auto svc_slot = [&](auto epi_c) __attribute__((always_inline)) {
        if constexpr (decltype(epi_c)::value) {
            svc_ktile(std::integral_constant<int, 0>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 1>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 2>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 3>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 4>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 5>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 6>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 7>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 8>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 9>{}, epi_c, F_{});
            svc_ktile(std::integral_constant<int, 10>{}, epi_c, F_{});
            if (nk == 12) svc_ktile(std::integral_constant<int, 11>{}, epi_c, T_{});
            else {
                svc_ktile(std::integral_constant<int, 11>{}, epi_c, F_{});
#pragma unroll 1
                for (int kt = 12; kt + 1 < nk; ++kt) svc_ktile(std::integral_constant<int, 12>{}, F_{}, F_{});
                svc_ktile(std::integral_constant<int, 12>{}, F_{}, T_{});
            }
        } else {
#pragma unroll 1
            for (int kt = 0; kt + 1 < nk; ++kt) svc_ktile(std::integral_constant<int, 12>{}, F_{}, F_{});
            svc_ktile(std::integral_constant<int, 12>{}, F_{}, T_{});
        }
    };

    // ------------------------------------------------------------------------------------------------ prologue
    // group 1 issues K-tiles 0 and 1 in the steady order; group 0 reads its first fragments; then A-q0(2) (its slot held A-q0(0))
    if (g == 1) {
        dma_setup(lane_now());
        const int k1 = PP_BK * 2;
        dma_a(ra2, PP_A0, 0, 0); dma_a(ra2, PP_A0, 1, 0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dma_b(rb2, PP_B0, jj, 0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dma_b(rb2, PP_B1, jj, bh1);
        dma_a(ra2, PP_A1, 0, aq1); dma_a(ra2, PP_A1, 1, aq1);
        dma_a(ra2, PP_A0 + PP_AQ, 0, k1); dma_a(ra2, PP_A0 + PP_AQ, 1, k1);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dma_b(rb2, PP_B0 + PP_BH, jj, k1);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) dma_b(rb2, PP_B1 + PP_BH, jj, k1 + bh1);
        dma_a(ra2, PP_A1 + PP_AQ, 0, k1 + aq1); dma_a(ra2, PP_A1 + PP_AQ, 1, k1 + aq1);
        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");     // A-q0(0), B-h0(0) landed (this wave's pieces)
    }
    barrier();
    if (g == 0) {
        frag_setup(lane_now());
        read_a(fa[0], 0, 0);
        read_b(sb_[0], 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        pp_ready(fa[0]); pp_ready(sb_[0]);
    }
    barrier();
    if (g == 1) {                                            // (va / vb: still the values of the first dma_setup)
        dma_a(ra2, PP_A0, 0, 2 * PP_BK * 2); dma_a(ra2, PP_A0, 1, 2 * PP_BK * 2);      // A-q0(2)
        pp_wait_vm<15>();                                      // B-h1(0) landed (16 pieces behind it)
    }
    barrier();

    // ------------------------------------------------------------------------------------------------ slots
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
#pragma unroll 1
    for (int s = 0; s < H; ++s) {
        if ((s & 1) == g) {
            frag_setup(lane_now());
            {   // opaque zero: folded into the first MFMAs (C = 0) the accumulators get renamed in the head K-tile and hipcc spills there
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+v"(acc[i][j])); }
            }
            __builtin_amdgcn_s_setprio(2);
#pragma unroll 1
            for (int kt = 0; kt < nk; kt += 2) {
                ktile(K0{}, kt == 0);
                ktile(K1{}, kt == 0);
            }
            __builtin_amdgcn_s_setprio(0);
            stream_skip(nk);
        } else if (s == 0) {
            dma_setup(lane_now());
            svc_slot(F_{});
        } else {
            epi_begin(s - 1);
            svc_slot(T_{});
        }
    }
    // ------------------------------------------------------------------------------------------------ drain: the last half-tile's epilogue
    if (((H - 1) & 1) == g) {
        epi_begin(H - 1);
        epi_aux();
        epi_res(0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        epi_aux_pin();
        pp_static_for<0, 32>([&](auto f_c) { constexpr int f = decltype(f_c)::value; epi_conv(std::integral_constant<int, (f >> 2)>{}, std::integral_constant<int, (f & 3)>{}); });
        pp_static_for<0, 16>([&](auto it_c) { epi_flush(it_c); });
        epi_rowstats(0);
        epi_rowstats(1);
    }
}

}  // namespace

// flag set of the call -> instantiation (0 = not this kernel's call)
static int pp_variant(const dist_gemm_args* a) {
    if (a->dtype != DIST_BF16 || a->taps != 1 || a->amap.mode != DIST_RM_PLAIN) return -1;
    if (a->flags & (DIST_EPI_MULG | DIST_EPI_MULG_POST | DIST_EPI_FP8 | DIST_EPI_OUT8 | DIST_EPI_FP8_ASCALAR)) return -1;
    if (a->omap.mode != DIST_OM_PLAIN && a->omap.mode != DIST_OM_HEADS) return -1;
    int f = 0;
    if (a->flags & DIST_EPI_LNFOLD) { if (!a->aux || !a->bias2) return -1; f |= PPF_LNFOLD; } else if (a->bias2) return -1;
    if (a->flags & DIST_EPI_RES) { if (!a->res) return -1; f |= PPF_RES; }
    if (a->flags & DIST_EPI_ACT2) { if (a->C || !a->C2) return -1; f |= PPF_ACT; } else if (!a->C) return -1;
    if (a->flags & DIST_EPI_ROWSTATS) { if (!a->rowstats || a->omap.mode != DIST_OM_PLAIN) return -1; f |= PPF_ROWSTATS; }
    if (a->omap.mode == DIST_OM_HEADS) {
        if ((f & (PPF_RES | PPF_ACT | PPF_ROWSTATS)) || a->omap.p0 < 16 || a->omap.p1 < 1 || a->ldc != 64) return -1;
        f |= PPF_HEADS;
    }
    if (a->N % PP_BN || a->K % (2 * PP_BK) || a->K / PP_BK < PP_MIN_NK || a->M < 2048) return -1;
    if (a->lda % 8 || a->ldb % 8 || a->ldc % 8 || a->ldc2 % 8 || a->ldres % 8) return -1;
    if ((long)a->M * a->lda >= (1l << 30) || (long)a->N * a->ldb >= (1l << 30) || (long)a->M * a->ldres >= (1l << 30)) return -1;
    return f;
}

template <int F>
static int launch_pp(const dist_gemm_args* a, int ng, int grid, hipStream_t s) {
    static const int dbg = dist_measure_knob("DIST_AMD_PP_DBG", 0);
    static DistSmemOnce attr;
    RUN_(dist_max_smem(attr, reinterpret_cast<const void*>(gemm_pp_kernel<F>), (size_t)PP_LDS));
    hipLaunchKernelGGL(gemm_pp_kernel<F>, dim3((unsigned)grid), dim3(512), (size_t)PP_LDS, s, *a, ng, dbg);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}

// returns 1 if handled, 0 if the call is not this kernel's (the caller falls through to gemm_fast8p_kernel), < 0 on error
int dist_k_gemm_pp(const dist_gemm_args* a, int ngroups, hipStream_t s) {
    static const int on = DIST_AB_KNOB("DIST_AMD_FAST_PP", 0);
    if (!on) return 0;
    const int f = pp_variant(a);
    if (f < 0) return 0;
    const long tiles = ((a->M + PP_BM - 1) / PP_BM) * (a->N / PP_BN);
    static const int cap = DIST_AB_KNOB("DIST_AMD_PP_GRID", 256);
    int grid = cap < 8 ? 8 : (cap > 256 ? 256 : cap & ~7);
    if (tiles < 2 * grid) return 0;                       // fewer than two tiles per block: nothing to ping-pong with
    switch (f) {
        case 0: return launch_pp<0>(a, ngroups, grid, s);
#ifndef PP_DEV_ONE
        case PPF_LNFOLD: return launch_pp<PPF_LNFOLD>(a, ngroups, grid, s);
        case PPF_LNFOLD | PPF_HEADS: return launch_pp<PPF_LNFOLD | PPF_HEADS>(a, ngroups, grid, s);
        case PPF_HEADS: return launch_pp<PPF_HEADS>(a, ngroups, grid, s);
        case PPF_LNFOLD | PPF_ACT: return launch_pp<PPF_LNFOLD | PPF_ACT>(a, ngroups, grid, s);
        case PPF_ACT: return launch_pp<PPF_ACT>(a, ngroups, grid, s);
        case PPF_RES: return launch_pp<PPF_RES>(a, ngroups, grid, s);
        case PPF_RES | PPF_ROWSTATS: return launch_pp<PPF_RES | PPF_ROWSTATS>(a, ngroups, grid, s);
#endif
        default: return 0;
    }
}
#endif  // DIST_AMD_MEASURE
