// C[omap(m)][n] = epi( sum_tap sum_k A[amap(m,tap)][k] * B[n][tap*K + k] )
//
// One MFMA GEMM family for every Linear / Conv on the DiST hot path (both operands
// K-contiguous: torch Linear weights are used as stored).  Convolutions are row-mapped
// GEMMs: the A loader resolves (row, tap) -> source row (or the zero padding) on the fly,
// so no im2col buffer ever exists in HBM.
//
// Tile: BM x BN x BK, WM x WN = 4 or 8 waves (64-wide), each wave owns a (BM/WM) x (BN/WN)
// sub-tile as 16x16 MFMA fragments.  Global -> registers -> LDS double buffer, one barrier
// per K tile.  bf16 LDS rows are unpadded (64 or 128 bytes) and XOR-swizzled by 16-byte chunk:
// chunk c of row r lives at c ^ f((r >> 2) & 3), f = {0,3,2,1} (64-byte rows) or c ^ ((r >> 1) & 7)
// (128-byte rows), which is conflict-free for the REAL ds_read_b128 service groups (lanes {0-3,12-15,20-27},
// {4-11,16-19,28-31}, ...: MI355X_MICROARCH.md §LDS).  The first layout - rows padded by 16 bytes, conflict-free
// only if a service group were lanes 0-15 - spent half of its LDS cycles in bank conflicts
// (SQ_LDS_BANK_CONFLICT 4.5 M of SQ_LDS_IDX_ACTIVE 9.1 M per launch, profiles/r02_nt_lds_swizzle.md); the
// fp32 parity instantiations keep it.
// The MFMA is issued "swapped" (B fragment as the A operand), so each lane ends up with
// 4 CONSECUTIVE output columns of one output row: bias / residual / activation epilogues
// and the stores are 4-wide vectors.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

constexpr int PAD = 8;

// Epilogue shared by the two loaders: bias / activation-derivative / residual / QuickGELU on the accumulators, then out through
// LDS so that residual / derivative tiles are fetched and result tiles stored as full 16-byte-per-lane rows.
//
// FREESLOT (the LDS-DMA loader): each wave's staging area is `region`, unpadded rows in ring slots that the last K-tile does not
// read, so no block barrier precedes the epilogue; with `pre` the first tile the epilogue would fetch (the activation-derivative
// tile, else the residual tile) is already on its way there by LDS-DMA (requested in one go by nt_prefetch_tile).
template <typename T, int BM, int BN, int WM, int WN, bool GENERIC, bool FREESLOT = false>
DEV void nt_epilogue(const dist_gemm_args& p, f32x4 (&acc)[BM / WM / 16][BN / WN / 16], char* smem, const int m0, const int n0,
                     char* region = nullptr, bool pre = false) {
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 15, lg = lane >> 4;
    const int M = (int)p.M, N = p.N;
    // ---- epilogue through LDS.  A lane holds, per fragment, row ..+li and 4 columns ..+4*lg+{0..3}; writing
    // those 8-byte pieces straight to HBM at a row stride (and fetching the residual the same way) ran at a
    // fraction of the write bandwidth.  Each wave stages its WTM x WTN sub-tile in LDS instead, so the residual /
    // activation-derivative tiles are fetched and the result tiles are stored as full 16-byte-per-lane rows.
    if (!FREESLOT) __syncthreads();                         // operand tiles are dead
    constexpr int ES = (int)sizeof(T);
    constexpr int ROWB = WTN * ES + (FREESLOT ? 0 : 16);    // staging row (padded for bank spread where LDS allows)
    constexpr int VPR = WTN * ES / 16, EPV = 16 / ES;       // 16-byte vectors per row, elements per vector
    char* ew = FREESLOT ? region : smem + wid * (WTM * ROWB);
    T* __restrict__ C = static_cast<T*>(p.C);
    T* __restrict__ C2 = static_cast<T*>(p.C2);
    const T* __restrict__ R = static_cast<const T*>(p.res);
    const T* __restrict__ X = static_cast<const T*>(p.aux);
    const int flags = p.flags;
    const int om = GENERIC ? p.omap.mode : (int)DIST_OM_PLAIN;
    const int op0 = p.omap.p0, op1 = p.omap.p1, op2 = p.omap.p2;
    const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;
    const int reps = om == DIST_OM_DUP ? op0 : 1;

    // destination (row, column) of logical element (m, n) for repetition `a`
    auto dest_of = [&](int m, int n, int a, int& ncol) -> long {
        ncol = n;
        if (om == DIST_OM_PLAIN) return m;
        if (om == DIST_OM_INSERTCLS) return (long)(m / op0) * (op0 + 1) + 1 + m % op0;
        if (om == DIST_OM_HEADS) {                           // [frame][head][q|k|v][token][64]; the leading dimension is 64
            const int hd = n >> 6, part = hd / op1, hh = hd - part * op1;
            ncol = n & 63;
            return ((long)((m / op0) * op1 + hh) * 3 + part) * op0 + m % op0;
        }
        const int bj = m / op1, nn = m % op1;
        if (om == DIST_OM_SPLITCOLS) { a = n / op2; ncol = n - a * op2; }
        return ((long)bj * op0 + a) * op1 + nn;
    };
    auto stage_in = [&](const T* __restrict__ src, int ld, int a) {
        if (FREESLOT && pre) {                              // already requested: wait for this wave's LDS-DMA pieces
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pre = false;
            return;
        }
        for (int v = lane; v < WTM * VPR; v += 64) {
            const int row = v / VPR, vec = v - row * VPR;
            const int m = mw + row, n = nw + vec * EPV;
            uint4 val = make_uint4(0, 0, 0, 0);
            if (m < M && n < N) { int nc; const long d = dest_of(m, n, a, nc); val = *reinterpret_cast<const uint4*>(src + d * ld + nc); }
            *reinterpret_cast<uint4*>(ew + row * ROWB + vec * 16) = val;
        }
    };
    auto flush = [&](T* __restrict__ dst, int ld, int a) {
        for (int v = lane; v < WTM * VPR; v += 64) {
            const int row = v / VPR, vec = v - row * VPR;
            const int m = mw + row, n = nw + vec * EPV;
            if (m < M && n < N) {
                int nc; const long d = dest_of(m, n, a, nc);
                store16_nt(dst + d * ld + nc, *reinterpret_cast<const uint4*>(ew + row * ROWB + vec * 16));
            }
        }
    };
    auto slot = [&](int i, int j) -> T* { return reinterpret_cast<T*>(ew + (i * 16 + li) * ROWB + (j * 16 + lg * 4) * ES); };

    if (flags & DIST_EPI_BIAS) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int n = nw + j * 16 + lg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float bv = (n + r < N) ? p.bias[n + r] + (p.bias2 ? p.bias2[n + r] : 0.f) : 0.f;
#pragma unroll
                for (int i = 0; i < FM; ++i) acc[i][j][r] += bv;
            }
        }
    }
    if ((flags & DIST_EPI_MULG) && !(flags & DIST_EPI_MULG_POST)) {   // v *= quickgelu'(aux[dest])   (never combined with DUP)
        stage_in(X, p.ldaux, 0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                float x[4];
                load4(slot(i, j), x);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] *= qgelu_grad_t<T>(x[r]);
            }
    }
    const bool act_only = (flags & DIST_EPI_ACT2) && !C;
    for (int a = 0; a < reps; ++a) {
        if (flags & DIST_EPI_RES) stage_in(R, p.ldres, a);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (flags & DIST_EPI_RES) {
                    float x[4];
                    load4(slot(i, j), x);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += x[r];
                }
                if (flags & DIST_EPI_MULG_POST) {            // keep the sum in the accumulators: the factor tile is staged next
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = v[r];
                    continue;
                }
                if (act_only) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = qgelu_t<T>(v[r]);
                }
                store4(slot(i, j), v);
            }
        if (flags & DIST_EPI_MULG_POST) {                   // v = (acc + bias + res) * quickgelu'(aux[dest])
            stage_in(X, p.ldaux, a);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    float x[4], v[4];
                    load4(slot(i, j), x);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * qgelu_grad_t<T>(x[r]);
                    store4(slot(i, j), v);
                }
        }
        if (act_only) {
            flush(C2, p.ldc2, a);
        } else {
            flush(C, p.ldc, a);
            if (flags & DIST_EPI_ACT2) {                    // second output = quickgelu(stored value)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) {
                        float x[4];
                        load4(slot(i, j), x);
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[r] = qgelu_t<T>(x[r]);
                        store4(slot(i, j), x);
                    }
                flush(C2, p.ldc2, a);
            }
        }
    }
}

template <typename T, int BM, int BN, int BK, int WM, int WN, bool GENERIC, int MINW = 1>
__global__ __launch_bounds__(WM * WN * 64, MINW) void gemm_nt_kernel(const dist_gemm_args p) {
    constexpr int NT = WM * WN * 64;                    // 4 or 8 waves
    constexpr bool SWZ = sizeof(T) == 2 && (BK == 32 || BK == 64);   // bf16: unpadded, chunk-swizzled rows
    constexpr int LD = SWZ ? BK : BK + PAD;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    constexpr int KV = BK / 8;
    constexpr int A_IT = (BM * KV + NT - 1) / NT, B_IT = (BN * KV + NT - 1) / NT;
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);                 // [2][BM][LD]
    T* Bs = As + 2 * BM * LD;                           // [2][BN][LD]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 15, lg = lane >> 4;

    // XCD-aware, bijective block remap: consecutive remapped ids share one XCD's L2, and the
    // n-tile index runs fastest so the blocks that re-read one A row panel are neighbours.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (int)((p.M + BM - 1) / BM);
    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, x = bid % 8, y = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = (int)p.M, N = p.N, K = p.K, taps = p.taps;

    const T* __restrict__ A = static_cast<const T*>(p.A);
    const T* __restrict__ B = static_cast<const T*>(p.B);

    const int ktp = (K + BK - 1) / BK;          // K tiles per tap
    const int total = ktp * taps;

    // Two register sets: the loads of K-tile t+2 are issued as soon as set (t & 1) has been written to LDS, so two tiles
    // are in flight per block (with one, every K-tile cost one full memory latency: the MFMAs of a tile take ~0.2 us).
    // The K-tiles are walked tap-major with INCREMENTAL state: the row decomposition of a thread's A rows happens once
    // (rowmap_prep), the source row / padding flag only when the tap changes, the B pointer advances by BK per tile.  The first
    // version derived everything from the flat tile index in every iteration (tile / tiles-per-tap, the row map with its modulo
    // and divide, masks bit by bit, clamps): 155 instructions per K-tile for 6 MFMAs per wave, 8.2 M VALU + 6.4 M SALU
    // instructions per conv launch against 0.68 M MFMAs - the kernel was bound by its own index arithmetic
    // (profiles/r02_nt_instruction_mix.md).  The loads stay UNCONDITIONAL with clamped addresses - also past the last tile -
    // and the zero padding is applied when the tile is written to LDS: a branch around a load makes hipcc lose its load counting
    // (vmcnt(0) everywhere, which drains both sets).
    struct Regs { Frag<T> a[A_IT], b[B_IT]; unsigned oka, okb; };
    int am[A_IT], akk[A_IT], bkk[B_IT];
    RowPrep aq[A_IT];
    // 32-bit BYTE offsets from the (wave-uniform) matrix bases: half the registers of 64-bit pointers, no 64-bit VALU per load, and
    // the loads take the scalar-base + vector-offset form (dist_op_gemm_nt checks that both matrices are < 2 GB)
    const char* Ab = reinterpret_cast<const char*>(A);
    const char* Bb = reinterpret_cast<const char*>(B);
    unsigned pa[A_IT];                          // source row of the current tap (row 0 when the tap falls into the padding)
    unsigned pb[B_IT];                          // weight row n, start of the current tap's K range is added per tile
    unsigned arow_ok = 0, brow_ok = 0, atap_ok = 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int v = tid + i * NT;
        const int row = min(v / KV, BM - 1);
        akk[i] = (v % KV) * 8;
        am[i] = min(m0 + row, M - 1);
        if (GENERIC) aq[i] = rowmap_prep(p.amap, am[i]);
        if (v < BM * KV && m0 + row < M) arow_ok |= 1u << i;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int v = tid + i * NT;
        const int row = min(v / KV, BN - 1);
        bkk[i] = (v % KV) * 8;
        pb[i] = (unsigned)min(n0 + row, N - 1) * (unsigned)p.ldb * (unsigned)sizeof(T);
        if (v < BN * KV && n0 + row < N) brow_ok |= 1u << i;
    }
    auto retap = [&](const int tap) __attribute__((always_inline)) {       // wave-uniform: the row images of this tap
        atap_ok = 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int src = GENERIC ? rowmap_src2(p.amap, am[i], aq[i], tap, taps) : am[i];
            if (src >= 0) atap_ok |= 1u << i;
            pa[i] = (unsigned)max(src, 0) * (unsigned)p.lda * (unsigned)sizeof(T);
        }
        atap_ok &= arow_ok;
    };
    int ld_k0 = 0, ld_tap = 0, ld_boff = 0;     // position of the next tile to load: K offset inside the tap, tap, tap * K + k0
    retap(0);
    auto gload = [&](Regs& R) __attribute__((always_inline)) {
        R.oka = 0; R.okb = 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int kk = ld_k0 + akk[i];
            if (((atap_ok >> i) & 1u) && kk < K) R.oka |= 1u << i;
            frag_load(R.a[i], reinterpret_cast<const T*>(Ab + (pa[i] + (unsigned)min(kk, K - 8) * (unsigned)sizeof(T))));
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int kk = ld_k0 + bkk[i];
            if (((brow_ok >> i) & 1u) && kk < K) R.okb |= 1u << i;
            frag_load(R.b[i], reinterpret_cast<const T*>(Bb + (pb[i] + (unsigned)((ld_boff - ld_k0) + min(kk, K - 8)) * (unsigned)sizeof(T))));
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {                  // to the next tile, tap-major
        ld_k0 += BK; ld_boff += BK;
        if (ld_k0 >= K) {
            ld_boff += K - ld_k0;                                           // the next tap starts at tap * K (K need not be a multiple of BK)
            ld_k0 = 0; ++ld_tap;
            retap(ld_tap);
        }
    };
    // physical 16-byte chunk of logical chunk c in row r (identity for the padded fp32 layout)
    auto pchunk = [](int r, int c) -> int {
        if (!SWZ) return c;
        return BK == 32 ? (c ^ ((4 - ((r >> 2) & 3)) & 3)) : (c ^ ((r >> 1) & 7));
    };
    auto sstore = [&](Regs& R, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int v = tid + i * NT;
            if (!((R.oka >> i) & 1u)) frag_zero(R.a[i]);
            if (v < BM * KV) frag_store(R.a[i], As + (buf * BM + v / KV) * LD + pchunk(v / KV, v % KV) * 8);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int v = tid + i * NT;
            if (!((R.okb >> i) & 1u)) frag_zero(R.b[i]);
            if (v < BN * KV) frag_store(R.b[i], Bs + (buf * BN + v / KV) * LD + pchunk(v / KV, v % KV) * 8);
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto body = [&](Regs& R, int tt) __attribute__((always_inline)) {
        const int buf = tt & 1;
        sstore(R, buf);                                     // waits for this set's loads only
        __syncthreads();                                    // tile tt visible; buffer (tt+1)&1 no longer read by anyone
        if (tt + 2 < total) advance();                      // (past the end the last tile is simply loaded again and never stored)
        gload(R);
        // fragment rows are (sub-tile base, a multiple of 16) + f * 16 + li: the swizzle term depends on li only
        const T* as = As + (buf * BM + wm * WTM + li) * LD;
        const T* bs = Bs + (buf * BN + wn * WTN + li) * LD;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            Frag<T> fa[FM], fb[FN];
            const int co = pchunk(li, kk * 4 + lg) * 8;
#pragma unroll
            for (int i = 0; i < FM; ++i) frag_load(fa[i], as + i * 16 * LD + co);
#pragma unroll
            for (int j = 0; j < FN; ++j) frag_load(fb[j], bs + j * 16 * LD + co);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) mma16(fb[j], fa[i], acc[i][j]);   // swapped: D[n][m]
        }
    };
    Regs r0, r1;
    gload(r0);
    if (total > 1) advance();
    gload(r1);
    int tt = 0;
    for (; tt + 1 < total; tt += 2) {
        body(r0, tt);
        body(r1, tt + 1);
    }
    if (tt < total) body(r0, tt);

    nt_epilogue<T, BM, BN, WM, WN, GENERIC>(p, acc, smem, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same GEMM with an LDS-DMA loader (bf16, K % 32 == 0): operand tiles go L2 / HBM -> LDS by `buffer_load ... lds` into a ring
// of three 32-deep stages, two K-tiles ahead of the MFMAs, with no staging registers and no ds_write pass.  The register-staged
// loader above keeps two K-tiles in flight per block and its time follows the bytes a CU ingests (conv3x3: 303 MB of mostly
// L2-resident re-reads per launch = 1.2 MB per CU, 51 us); here a stage in flight costs LDS only.
//  * per-lane SOURCE offsets carry everything the row map does: the source row of the current tap (rowmap_prep once, rowmap_src2
//    per tap), the chunk swizzle of the LDS image (the DMA writes lane-linear 1 KB pieces = 16 rows x 64 B; chunk c of row r
//    lives at c ^ f((r >> 2) & 3)), and the zero padding: a padded row's offset lies beyond the descriptor, so the piece reads
//    zeros; K offset and tap * K are scalar offsets;
//  * every wave moves BM / 128 pieces of A and one piece of B per stage (B is staged as 128 rows: rows >= N read as zero);
//  * one raw s_barrier per K-tile behind a counted vmcnt: stage t has landed for every wave, stage t - 1 is no longer read.
typedef __attribute__((address_space(3))) void* nt_lds_ptr;
template <int N> DEV void nt_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else static_assert(N < 0, "unsupported count");
}
// byte offset of 16-byte vector v (row-major over a wave's sub-tile, VPR vectors per row) of a plain bf16 matrix; beyond it: reads zero
template <int VPR> DEV unsigned nt_pre_off(const int v, const int mw, const int nw, const int M, const int N, const int ld) {
    const int row = v / VPR, vec = v - row * VPR;
    const int m = mw + row, n = nw + vec * 8;
    return (m < M && n < N) ? ((unsigned)m * (unsigned)ld + (unsigned)n) * 2u : 0x80000000u;
}
// OGEN = false: the output map is known to be plain at compile time (the epilogue's row-piece loops then carry no code for the other maps -
// their divides - at all; as a runtime mode hipcc keeps them in the loop bodies behind branches or selects)
template <int BM, int BN, int WM, int WN, bool GENERIC, int MINW, bool OGEN = true>
__global__ __launch_bounds__(512, MINW) void gemm_nt_dma_kernel(const dist_gemm_args p, const int rotate) {
    using T = bf16_t;
    constexpr int BK = 32, STAGES = 3, BNP = 128;
    constexpr int PA = BM / 128, NP = PA + 1;             // 1 KB pieces per wave per stage
    constexpr int A_BYTES = BM * 64, STAGE_BYTES = (BM + BNP) * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN, FM = WTM / 16, FN = WTN / 16;
    static_assert(WM * WN == 8 && (BM == 128 || BM == 256) && BN <= BNP, "8 waves, one B piece per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int li = lane & 15, lg = lane >> 4;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (int)((p.M + BM - 1) / BM);
    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nblk / 8, r = nblk % 8, x = bid % 8, y = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = (int)p.M, N = p.N, K = p.K, taps = p.taps;
    const int total = (K / BK) * taps;

    constexpr unsigned OOB = 0x80000000u;                 // beyond either descriptor: the lane's 16 bytes read as zero
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, 0x7ffffff0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.B), 0, N * p.ldb * 2, 0x00020000);
    const int lrow = lane >> 2;
    const int lch = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    // (scalars, not arrays: with `ga[]` captured by reference, written in retap and read in stage, hipcc silently emits no host
    // stub for the kernel - the library then fails to load with an undefined symbol)
    int am[PA]; RowPrep aq[PA];
    unsigned ga0 = 0, ga1 = 0;
    bool arow_ok[PA];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = (wid * PA + j) * 16 + lrow;
        am[j] = min(m0 + row, M - 1);
        arow_ok[j] = m0 + row < M;
        if (GENERIC) aq[j] = rowmap_prep(p.amap, am[j]);
    }
    const unsigned gb = ((unsigned)(n0 + wid * 16 + lrow) * (unsigned)p.ldb + lch * 8) * 2u;   // rows >= N: beyond rb
    auto retap = [&](const int tap) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int src = GENERIC ? rowmap_src2(p.amap, am[j], aq[j], tap, taps) : am[j];
            const unsigned g = (arow_ok[j] && src >= 0) ? ((unsigned)src * (unsigned)p.lda + lch * 8) * 2u : OOB;
            if (j == 0) ga0 = g; else ga1 = g;
        }
    };
    // `rotate` bit 0 (measurement knob, off): block b starts at K-tile (b mod total) and wraps around - all blocks walk the SAME small
    // weight matrix, and in step they all read the same 8 KB of it at the same moment.  Measured: not a limiter (conv3x3 47.5 vs
    // 47.8 us), and the summation order of a row would depend on the block that holds it, i.e. on the batch it is part of - the
    // engine's results are bit-identical for a clip whatever it is batched with, and stay so.
    const int ktp = K / BK;
    const int t0 = (rotate & 1) ? (int)(blockIdx.x % (unsigned)total) : 0;
    int ld_tap = t0 / ktp, ld_k0 = (t0 - ld_tap * ktp) * BK;
    retap(ld_tap);
    auto stage = [&](const int t) __attribute__((always_inline)) {        // K-tile t (tap-major) into ring slot t % 3
        char* sb = smem + (t % STAGES) * STAGE_BYTES;
        // (written out: with the builtin inside a loop over the template-dependent PA, hipcc silently emits no host stub for the kernel)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (nt_lds_ptr)(sb + (wid * PA) * 1024), 16, ga0, ld_k0 * 2, 0, 0);
        if constexpr (PA == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (nt_lds_ptr)(sb + (wid * PA + 1) * 1024), 16, ga1, ld_k0 * 2, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (nt_lds_ptr)(sb + A_BYTES + wid * 1024), 16, gb, (ld_tap * K + ld_k0) * 2, 0, 0);
        ld_k0 += BK;
        if (ld_k0 >= K) { ld_k0 = 0; ++ld_tap; if (ld_tap >= taps) ld_tap = 0; retap(ld_tap); }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int co = (lg ^ ((4 - ((li >> 2) & 3)) & 3)) * 16;               // fragment chunk of this lane, bytes (depends on li only)
    stage(0);
    if (total > 1) stage(1);
    for (int t = 0; t < total; ++t) {
        if (t + 1 < total) nt_wait_vm<NP>(); else nt_wait_vm<0>();       // stage t landed (this wave's pieces); t + 1 may be in flight
        __builtin_amdgcn_s_barrier();                                     // ... for every wave; nobody reads stage t - 1 any more
        if (t + 2 < total) stage(t + 2);
        __builtin_amdgcn_sched_barrier(0);
        const char* as = smem + (t % STAGES) * STAGE_BYTES + (wm * WTM + li) * 64 + co;
        const char* bs = smem + (t % STAGES) * STAGE_BYTES + A_BYTES + (wn * WTN + li) * 64 + co;
        Frag<T> fa[FM], fb[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) fa[i].v = *reinterpret_cast<const bf16x8*>(as + i * 16 * 64);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb[j].v = *reinterpret_cast<const bf16x8*>(bs + j * 16 * 64);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) mma16(fb[j], fa[i], acc[i][j]);   // swapped: D[n][m]
    }
    // ---- epilogue.  The last K-tile reads ring slot (total - 1) % 3 only: the other two slots are free for every wave that is past
    // that tile's barrier, and hold exactly one BM x BN bf16 tile - waves 0-3 stage in one, waves 4-7 in the other, without a block
    // barrier.  The tile the epilogue fetches first (activation derivative, else residual) is requested here in ONE go by LDS-DMA
    // (no registers, no dependent round trips; rows / columns beyond the matrix read as zero through the descriptor's bound): fetched
    // piece by piece inside the epilogue it was half of conv3x3's time (profiles/r02_nt_instruction_mix.md).
    constexpr int WTILE = WTM * WTN * 2, NPRE = WTILE / 1024, VPR = WTN / 8;
    static_assert(WTILE % 1024 == 0 && NPRE <= 6 && 4 * WTILE <= STAGE_BYTES, "per-wave staging tile: whole 1 KB pieces, four per ring slot");
    char* region = smem + ((total + (wid >> 2)) % STAGES) * STAGE_BYTES + (wid & 3) * WTILE;
    bool pre = false;
    {
        const int flags = p.flags;
        const bool first_aux = (flags & DIST_EPI_MULG) && !(flags & DIST_EPI_MULG_POST);
        const void* src = first_aux ? p.aux : ((flags & DIST_EPI_RES) ? p.res : nullptr);
        const int ld = first_aux ? p.ldaux : p.ldres;
        const int om = OGEN ? p.omap.mode : (int)DIST_OM_PLAIN;
        if (src && om == DIST_OM_PLAIN && (long)M * ld < (1l << 30) && !(rotate & 2)) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(src), 0, (int)((long)M * ld * 2), 0x00020000);
            const int mw = m0 + wm * WTM, nw = n0 + wn * WTN;
            const unsigned o0 = nt_pre_off<VPR>(lane, mw, nw, M, N, ld), o1 = nt_pre_off<VPR>(lane + 64, mw, nw, M, N, ld);
            const unsigned o2 = nt_pre_off<VPR>(lane + 128, mw, nw, M, N, ld), o3 = nt_pre_off<VPR>(lane + 192, mw, nw, M, N, ld);
            const unsigned o4 = nt_pre_off<VPR>(lane + 256, mw, nw, M, N, ld), o5 = nt_pre_off<VPR>(lane + 320, mw, nw, M, N, ld);
            // (written out, as in stage(): the builtin inside a loop over a template-dependent count, or with a dependent call among
            // its arguments, loses the kernel's host stub)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (nt_lds_ptr)(region), 16, o0, 0, 0, 0);
            if constexpr (NPRE > 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (nt_lds_ptr)(region + 1024), 16, o1, 0, 0, 0);
            if constexpr (NPRE > 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (nt_lds_ptr)(region + 2048), 16, o2, 0, 0, 0);
            if constexpr (NPRE > 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (nt_lds_ptr)(region + 3072), 16, o3, 0, 0, 0);
            if constexpr (NPRE > 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (nt_lds_ptr)(region + 4096), 16, o4, 0, 0, 0);
            if constexpr (NPRE > 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (nt_lds_ptr)(region + 5120), 16, o5, 0, 0, 0);
            pre = true;
        }
    }
    nt_epilogue<T, BM, BN, WM, WN, OGEN, true>(p, acc, smem, m0, n0, region, pre);
}

template <int BM, int BN, int WM, int WN, bool GENERIC, int MINW, bool OGEN = true>
int launch_dma(const dist_gemm_args& a, hipStream_t s) {
    constexpr size_t ring = (size_t)3 * (BM + 128) * 64;
    constexpr size_t stag = (size_t)WM * WN * (BM / WM) * ((BN / WN) * 2 + 16);
    constexpr size_t smem = ring > stag ? ring : stag;
    static DistSmemOnce attr;
    auto kern = gemm_nt_dma_kernel<BM, BN, WM, WN, GENERIC, MINW, OGEN>;
    RUN_(dist_max_smem(attr, reinterpret_cast<const void*>(kern), smem));
    const long tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    static const int rotate = DIST_AB_KNOB("DIST_AMD_NT_ROTATE", 0);   // measurement knob: 1 = rotated K order, 2 = no tile prefetch
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), smem, s, a, rotate);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

template <typename T, int BM, int BN, int BK, int WM, int WN, bool GENERIC, int MINW = 1>
int launch(const dist_gemm_args& a, hipStream_t s) {
    // operand double buffer (bf16: unpadded swizzled rows) or the per-wave epilogue staging (padded rows), whichever is larger
    constexpr size_t ld = (sizeof(T) == 2 && (BK == 32 || BK == 64)) ? BK : BK + PAD;
    constexpr size_t opnd = (size_t)2 * (BM + BN) * ld * sizeof(T);
    constexpr size_t stag = (size_t)WM * WN * (BM / WM) * ((BN / WN) * sizeof(T) + 16);
    constexpr size_t smem = opnd > stag ? opnd : stag;
    static DistSmemOnce attr;
    auto kern = gemm_nt_kernel<T, BM, BN, BK, WM, WN, GENERIC, MINW>;
    RUN_(dist_max_smem(attr, reinterpret_cast<const void*>(kern), smem));
    const long tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(WM * WN * 64), smem, s, a);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

template <typename T>
int dispatch(const dist_gemm_args& a, hipStream_t s) {
    const bool plain = a.amap.mode == DIST_RM_PLAIN && a.omap.mode == DIST_OM_PLAIN && a.taps == 1;
    const bool n96 = (a.N % 96 == 0) && (a.N % 128 != 0);
    // 8 waves per block (the wave grid is 4x2 / 2x4): half the accumulators, fragments and staging registers per lane, so two
    // 8-wave blocks (16 waves) share a CU where the 4-wave shapes had 2-3 blocks of 4 (conv3x3 69.6 -> 59.3 us, 384x384 Linear
    // 48.6 -> 44.8 us, 384->96 Linear 18.3 -> 16.7 us alone; profiles/r01_nt_8wave.md).  DIST_AMD_NT_W8=0 restores the 4-wave shapes
    // (measurement knob: bit 0 = N % 96 shapes, bit 1 = plain K % 64, bit 2 = generic).
    static const int w8 = DIST_AB_KNOB("DIST_AMD_NT_W8", 7);
    // DIST_AMD_NT_OCC=0 (measurement knob): the N % 96 shape compiled for 4 instead of 6 waves per SIMD (two 8-wave blocks per CU
    // instead of three).  Measured (profiles/r02_nt_occupancy.md): conv3x3 59.6 -> 55.8 us, step -0.15 ms with three blocks; the
    // generic 128x128x32 shape needs spills at 80 registers and gains nothing, so only N % 96 has the variant.
    static const int occ = DIST_AB_KNOB("DIST_AMD_NT_OCC", 1);
    if constexpr (std::is_same<T, bf16_t>::value) {
        // LDS-DMA loader (three stages in flight, no staging registers).  DIST_AMD_NT_DMA=0: the register-staged loader.
        static const int dma = dist_knob("DIST_AMD_NT_DMA", 1);
        if (dma && a.K % 32 == 0) {
            static const bool oplain_on = (DIST_AB_KNOB("DIST_AMD_NT_OPLAIN", 1) != 0);   // A/B: 0 = the generic out-map instantiation for plain outputs too
            const bool op = (!DIST_AB || oplain_on) && a.omap.mode == DIST_OM_PLAIN;
            if (n96 && a.N == 96 && a.M > 768l * 128)                                                      // 392 blocks: one round of 2 per CU
                return op ? launch_dma<256, 96, 8, 1, true, 4, false>(a, s) : launch_dma<256, 96, 8, 1, true, 4>(a, s);
            if (n96) return op ? launch_dma<128, 96, 4, 2, true, 6, false>(a, s) : launch_dma<128, 96, 4, 2, true, 6>(a, s);
            return op ? launch_dma<128, 128, 2, 4, true, 6, false>(a, s) : launch_dma<128, 128, 2, 4, true, 6>(a, s);
        }
        // A block is a serial chain of latency-bound K-tile steps, so a launch takes (rounds of resident blocks) x (one block's
        // time): 100 352 rows as 128-row tiles are 784 blocks on 768 resident slots (256 CUs x 3) - TWO rounds for 16 blocks.
        // 256-row tiles (8 x 1 waves, 32 x 96 per wave) make the same work 392 blocks = one round.  DIST_AMD_NT_BM256=0: off.
        static const int bm256 = DIST_AB_KNOB("DIST_AMD_NT_BM256", 1);
        if (n96 && a.N == 96 && bm256 && a.M > 768l * 128) return launch<T, 256, 96, 32, 8, 1, true, 4>(a, s);
        if (n96 && (occ & 1)) return launch<T, 128, 96, 32, 4, 2, true, 6>(a, s);
    }
    // (the rejected shapes - 4-wave blocks, the 4-waves-per-SIMD build of the N % 96 shape - exist in the timing-only library only)
    if constexpr (DIST_AB) {
        if (n96 && !(w8 & 1)) return launch<T, 128, 96, 32, 4, 1, true>(a, s);
        if (!n96 && plain && a.K % 64 == 0 && !(w8 & 2)) return launch<T, 128, 128, 64, 2, 2, false>(a, s);
        if (!n96 && !(plain && a.K % 64 == 0) && !(w8 & 4)) return launch<T, 128, 128, 32, 2, 2, true>(a, s);
    }
    if constexpr (!std::is_same<T, bf16_t>::value || DIST_AB) {        // (bf16 N % 96 shapes all returned above: three blocks per CU)
        if (n96) return launch<T, 128, 96, 32, 4, 2, true>(a, s);
    }
    if (plain && a.K % 64 == 0) return launch<T, 128, 128, 64, 2, 4, false>(a, s);
    return launch<T, 128, 128, 32, 2, 4, true>(a, s);
}

}  // namespace

extern "C" int dist_op_gemm_nt(const dist_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->B || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->taps <= 0) return DIST_ERR_ARG;
    if (a->K % 8 || a->lda % 8 || a->ldb % 8) return DIST_ERR_ARG;
    {   // the epilogue moves 16-byte vectors: rows of every output / residual operand must keep that alignment
        const int epv = a->dtype == DIST_BF16 ? 8 : 4;
        if (a->N % epv) return DIST_ERR_ARG;
        if ((a->C && a->ldc % epv) || ((a->flags & DIST_EPI_ACT2) && a->ldc2 % epv) || ((a->flags & DIST_EPI_RES) && a->ldres % epv) ||
            ((a->flags & DIST_EPI_MULG) && a->ldaux % epv)) return DIST_ERR_ARG;
        if (a->omap.mode == DIST_OM_SPLITCOLS && a->omap.p2 % epv) return DIST_ERR_ARG;
        if (a->omap.mode == DIST_OM_DUP && (a->flags & DIST_EPI_MULG)) return DIST_ERR_ARG;
        if ((a->flags & DIST_EPI_MULG_POST) && (!(a->flags & DIST_EPI_MULG) || (a->flags & DIST_EPI_ACT2))) return DIST_ERR_ARG;
    }
    if (a->M > (1 << 30)) return DIST_ERR_ARG;
    {   // the loaders address both operands with 32-bit byte offsets (generous bound for the strided / skip-cls row images of A)
        const int64_t es = a->dtype == DIST_BF16 ? 2 : 4;
        const int64_t a_rows = a->amap.mode == DIST_RM_PLAIN ? a->M : 2 * a->M + a->M / 64 + 64;
        if (a_rows * a->lda * es >= (1ll << 31) || (int64_t)a->N * a->ldb * es >= (1ll << 31)) return DIST_ERR_ARG;
    }
    if (!a->C && !(a->flags & (DIST_EPI_ACT2 | DIST_EPI_OUT8))) return DIST_ERR_ARG;   // (an e4m3 image may be the only output)
    if ((a->flags & DIST_EPI_ACT2) && !a->C2 && !(a->flags & DIST_EPI_OUT8)) return DIST_ERR_ARG;   // (e4m3-only activated output: C8)
    if ((a->flags & DIST_EPI_BIAS) && !a->bias) return DIST_ERR_ARG;
    if ((a->flags & DIST_EPI_RES) && !a->res) return DIST_ERR_ARG;
    if ((a->flags & DIST_EPI_MULG) && !a->aux) return DIST_ERR_ARG;
    if (a->omap.mode == DIST_OM_SPLITCOLS && (a->omap.p2 % 4 || a->N != a->omap.p0 * a->omap.p2)) return DIST_ERR_ARG;
    if (a->omap.mode == DIST_OM_HEADS && (a->omap.p0 <= 0 || a->omap.p1 <= 0 || a->N != 3 * 64 * a->omap.p1 || (a->C && a->ldc != 64) ||
                                          (a->flags & (DIST_EPI_RES | DIST_EPI_MULG | DIST_EPI_ACT2)) || (!a->C && !(a->flags & DIST_EPI_OUT8)))) return DIST_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((a->flags & DIST_EPI_OUT8) && !(a->flags & DIST_EPI_FP8)) {   // e4m3 image of the output: only the 256x256 LDS-DMA kernels write it
        if (a->dtype != DIST_BF16 || (a->flags & DIST_EPI_MULG)) return DIST_ERR_ARG;
        const int fast = dist_k_gemm_fast(a, s);
        return fast > 0 ? DIST_OK : (fast < 0 ? fast : DIST_ERR_ARG);
    }
    if (a->flags & DIST_EPI_FP8) {                             // e4m3 operands: only the two-group 256x256 LDS-DMA kernel has the fp8 MFMAs
        if (a->dtype != DIST_BF16 || !a->a_scale || !a->b_scale || (a->flags & DIST_EPI_MULG)) return DIST_ERR_ARG;
        if ((a->flags & DIST_EPI_LNFOLD) && (!a->aux || !a->bias2)) return DIST_ERR_ARG;
        if ((a->flags & DIST_EPI_ROWSTATS) && !a->rowstats) return DIST_ERR_ARG;
        const int fast = dist_k_gemm_fast(a, s);
        return fast > 0 ? DIST_OK : (fast < 0 ? fast : DIST_ERR_ARG);
    }
    if (a->flags & (DIST_EPI_LNFOLD | DIST_EPI_ROWSTATS)) {   // only the 256x256 LDS-DMA kernels implement the folded LayerNorm / row statistics
        if ((a->flags & DIST_EPI_LNFOLD) && ((a->flags & DIST_EPI_MULG) || !a->aux || !a->bias2)) return DIST_ERR_ARG;
        if ((a->flags & DIST_EPI_ROWSTATS) && (!a->rowstats || (a->flags & DIST_EPI_MULG))) return DIST_ERR_ARG;
        const int fast = dist_k_gemm_fast(a, s);
        return fast > 0 ? DIST_OK : (fast < 0 ? fast : DIST_ERR_ARG);
    }
    const int small = dist_k_gemm_small(a, s);            // a few hundred rows (ada-pooling, head): split-K, no LDS staging
    if (small != 0) return small < 0 ? small : DIST_OK;
    const int fast = dist_k_gemm_fast(a, s);              // large plain bf16 GEMMs (frozen ViT): LDS-DMA 256x256 kernel
    if (fast != 0) return fast < 0 ? fast : DIST_OK;
    return a->dtype == DIST_BF16 ? dispatch<bf16_t>(*a, s) : dispatch<float>(*a, s);
}
