// out[i][c] += sum_m A[m][i] * B[m][c]   -   the LARGE plain weight gradients (input_linear 384 x 768, the c_proj pair 384 x 480,
// [ffn.c_fc | temporal_ffn.c_fc1] 480 x 384: 67 of the 103 GF of weight gradients per layer), round 4.
//
// The two-wave-group schedule of the dominant NT GEMM (gemm_fast8p_kernel, gemm_fast.hip) on operands that are reduced over their
// ROW index.  gemm_tn_kernel (gemm_tn.hip) is a serial chain per 64-row step - global loads -> registers -> ds_write_b128 (13 LDS
// cycles each) -> block barrier -> 24 transpose reads -> 16 MFMAs, every wave of the block in the same stage at the same time: 20 %
// MFMA busy, 25 % LDS busy, 38 % of the wave cycles parked (profiles/r03_tn_big_tiles.md).  Here
//   * rows travel L2 / HBM -> LDS by `buffer_load ... lds` (no staging registers, no ds_write pass) into a RING of S = 20 half-tile slots of
//     [32 rows][128 column slots] bf16 (8 KB = one 1 KB piece per wave; rows of 256 B = one pass over the 64 banks), in staging order
//         seq = 4*t + {A-h0: 0, B-h0: 1, B-h1: 2, A-h1: 3}   ->   slot seq mod S,
//     phase r reads seq r + 2 and stages seq r + S into the slot phase r - 2 read (the WAR rule of gemm_fast.hip); S - 3 = 17 half-tiles = 136 KB
//     stay in flight behind the counted vmcnt;
//   * waves 0-3 and 4-7 (the SIMD partners) run ONE barrier apart: in every barrier interval one group is in its load section
//     (transpose reads for the multiply section AFTER the next, one LDS-DMA piece, the counted wait) and the other multiplies
//     (2*FA MFMAs on one quadrant of its (32*FA) x 64 sub-tile) - the phase table of gemm_fast.hip;
//   * fragments by ds_read_b64_tr_b16 (2 LDS cycles per wave instruction), issued as INLINE ASSEMBLY: behind the builtin hipcc waits vmcnt(0)
//     in front of every transpose read (it cannot separate the read from the LDS-DMA pieces in flight), which serialises the ring; the counted
//     vmcnt + barrier one phase earlier is the ordering.  Lane (li, lg) addresses 4 consecutive slots of row 4*lg + (li >> 2) (and 16 rows
//     further) and receives column li - k-slot (lg, e) <-> row 4*lg + e / 16 + 4*lg + (e - 4), the same assignment for both operands;
//   * 32-byte piece c32 of row r lives at piece c32 ^ (r & 7): the eight rows a 32-lane service group touches cover the 64 banks
//     exactly once (SQ_LDS_BANK_CONFLICT = 0); the LDS-DMA image is lane-linear, so the same involution sits on the per-lane SOURCE offset;
//   * rows behind the block's row range read ZERO through the buffer descriptor (base = first row of the range, size = the range) and are
//     requested like any other piece: no row clamping, no validity flags, no tail logic - the loop body is lcm(4, S) phases (x 2: the two B
//     register sets swap roles every K-tile) with compile-time slots, and the launcher makes a row range a whole number of bodies.  Column slots
//     beyond NI / K read whatever follows in the row - output element (i, c) depends on column i of A and column c of B only, and the second
//     phase never stores those outputs;
//   * the bias gradient (column sums of A) rides along: wave (wr, wc) reads fragment wc of its wave row a second time and sums its
//     packed pairs with v_dot2_f32_bf16 against (1, 1) - 2 transpose reads and 4 VALU per quadrant and K-tile, no branch
//     (CSB: the operands are SWAPPED - the kernel's rows are B's columns - and the sums run over fragment wr of the wave column);
//   * the row-split partial tile leaves the accumulators in FRAGMENT order (one 1 KB piece per wave instruction, 16 bytes per lane);
//     tn8p_reduce_kernel sums the splits in index order (bit-repeatable) into the parameter layout.
// Tile = 192 x 256 (FA = 3 fragments per quadrant).  Every large gradient of the path has a 384-wide side: that side becomes the tile rows
// (2 x 192), if necessary by swapping the operands (480 x 384 runs as 384 x 480: 94 % of the MFMAs useful instead of 70 %).
// Measured (profiles/r04_tn8p.md): main kernel 53 -> 43 us on the input_linear gradient at 3.7-3.9 TB/s whatever the ring depth (S = 16, 20, or two
// 64-row K-tile buffers: 42.4 / 43.0 / 44.8 us) - the gradients are memory-bound (operands + partial tiles), not schedule-bound.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef s16x4 __attribute__((address_space(3))) * lds_v4;
typedef __attribute__((ext_vector_type(2))) __bf16 t8_bf16x2;

constexpr int T8_BJ = 256;

DEV float* tn8_dst(const dist_gemm_tn_args& p, int ii, int c) {
    if (p.out2 && c >= p.split_c) return p.out2 + (long)ii * p.so_i2 + (c - p.split_c);
    return p.out + (long)ii * p.so_i + (long)(c / p.inner) * p.so_outer + (c % p.inner);
}

template <int V> using IC = std::integral_constant<int, V>;
// 8 k-slots of one column for each of 16 columns: two transpose reads 16 rows apart (row pitch 256 B)
template <int OFF> DEV bf16x8 t8_tr8(const unsigned addr) {
    static_assert(OFF >= 0 && OFF + 16 * 256 < 65536, "16-bit ds offset");
    s16x4 a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b) : "v"(addr), "n"(OFF + 16 * 256) : "memory");
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = a; u.s[1] = b;
    return u.v;
}

template <int G, int N, class F> DEV void t8_unroll(F& f) {
    if constexpr (G < N) { f(IC<G>{}); t8_unroll<G + 1, N>(f); }
}
template <int N> DEV void t8_wait_vm_n() {
    static_assert(N >= 0 && N < 64, "6-bit vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
template <int OFF> DEV bf16x8 t8_tr8w(const unsigned (&ad)[3]) {   // OFF: byte offset of the slot inside the ring; ad[w] = lane address + w * 64 KB
    return t8_tr8<OFF & 0xffff>(ad[OFF >> 16]);
}

template <int S, bool CSB>
__global__ __launch_bounds__(512, 1) void gemm_tn8r_kernel(const dist_gemm_tn_args p, const int chunk, const int tiles_i, const int tiles_c) {
    constexpr int FA = 3, BI = 64 * FA, BK = 32, HT = BK * 256;              // 8 KB per half-tile slot
    constexpr int PER0 = (S % 4 == 0) ? S : (S % 2 == 0 ? 2 * S : 4 * S);   // lcm(4, S)
    constexpr int PER = ((PER0 / 4) % 2 == 0) ? PER0 : 2 * PER0;            // phases per loop body: an even number of K-tiles
    static_assert(S >= 8 && S * HT <= 160 * 1024 && PER % S == 0 && PER % 8 == 0, "ring shape");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int li = lane & 15, lg = lane >> 4;

    const int tiles_ij = tiles_i * tiles_c;
    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x;
        const int q = nblk / 8, r = nblk % 8, x = bid % 8, y = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int ms = bid / tiles_ij, t = bid % tiles_ij;
    const int ti = t % tiles_i, tc = t / tiles_i;
    const int i0 = ti * BI, c0 = tc * T8_BJ;
    const int M = (int)p.M;
    const int mbeg = ms * chunk, rows = min(M, mbeg + chunk) - mbeg;
    if (rows <= 0) return;
    const int nbody = (rows + BK * (PER / 4) - 1) / (BK * (PER / 4));

    // ---- LDS-DMA sources: this wave moves rows 4*wid + (lane >> 4) of every half-tile (one 1 KB piece)
    const bf16_t* Ab = static_cast<const bf16_t*>(p.A) + ((long)mbeg * p.lda + i0);
    const bf16_t* Bb = static_cast<const bf16_t*>(p.B) + ((long)mbeg * p.ldb + c0);
    // The descriptors end behind the LAST COLUMN THE VIEW OWNS in the block's last row, (rows - 1) * ld + (width - first column): A / B may be
    // column-offset views of a wider buffer (ld > width), and a range of rows * ld elements would reach `offset` elements past that buffer's end
    // for the last row block.  Every row behind the block's range still starts at >= rows * ld - first column > the bound and reads as zero.
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Ab), 0, (int)(((long)(rows - 1) * p.lda + (p.NI - i0)) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Bb), 0, (int)(((long)(rows - 1) * p.ldb + (p.K - c0)) * 2), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned ga, gb;
    {
        const int r = 4 * wid + (lane >> 4);
        const int pc = lane & 15;
        const int slot = (((pc >> 1) ^ (r & 7)) << 4) + (pc & 1) * 8;
        ga = (slot & 63) < 16 * FA ? (unsigned)(r * p.lda + (slot >> 6) * (BI / 2) + (slot & 63)) * 2u : OOB;
        gb = (unsigned)(r * p.ldb + (slot >> 5) * 64 + (slot & 31)) * 2u;
    }
    const int atile = BK * p.lda * 2, btile = BK * p.ldb * 2;
    // stage half-tile KIND of K-tile kt into ring slot SLOT
    auto stage = [&](auto slot_c, auto kind_c, const int kt) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value, KIND = decltype(kind_c)::value;
        char* sb = smem + SLOT * HT + wid * 1024;
        if constexpr (KIND == 0 || KIND == 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr)sb, 16, ga, kt * atile + (KIND == 3 ? 16 * FA * 2 : 0), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr)sb, 16, gb, kt * btile + (KIND == 2 ? 64 : 0), 0, 0);
    };

    const int rx = 4 * (lg & 1) + (li >> 2);
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((lds_ptr)smem);
    const unsigned rbase = lds0 + (4 * lg + (li >> 2)) * 256 + (li & 3) * 8;
    unsigned a_ad[FA][3], b_ad[2][3], c_ad[3];
#pragma unroll
    for (int w = 0; w < 3; ++w) {
#pragma unroll
        for (int f = 0; f < FA; ++f) a_ad[f][w] = rbase + (((wr * 4 + f) ^ rx) << 5) + w * 65536;
#pragma unroll
        for (int j = 0; j < 2; ++j) b_ad[j][w] = rbase + (((wc * 2 + j) ^ rx) << 5) + w * 65536;
        c_ad[w] = rbase + ((CSB ? ((wc * 2 + wr) ^ rx) : ((wr * 4 + wc) ^ rx)) << 5) + w * 65536;
    }
    bf16x8 fa[2][FA], sb_[2][2], fc;
    auto read_a = [&](bf16x8 (&f)[FA], auto slot_c) __attribute__((always_inline)) {
        constexpr int OFF = decltype(slot_c)::value * HT;
#pragma unroll
        for (int i = 0; i < FA; ++i) f[i] = t8_tr8w<OFF>(a_ad[i]);
        if constexpr (!CSB) fc = t8_tr8w<OFF>(c_ad);
    };
    auto read_b = [&](bf16x8 (&f)[2], auto slot_c) __attribute__((always_inline)) {
        constexpr int OFF = decltype(slot_c)::value * HT;
#pragma unroll
        for (int j = 0; j < 2; ++j) f[j] = t8_tr8w<OFF>(b_ad[j]);
        if constexpr (CSB) fc = t8_tr8w<OFF>(c_ad);
    };

    f32x4 acc[2 * FA][4];
#pragma unroll
    for (int i = 0; i < 2 * FA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float cs[2] = {0.f, 0.f};
    auto colsum_q = [&](const int q) __attribute__((always_inline)) {
        asm volatile("" : "+v"(fc));
        const t8_bf16x2 one = {(bf16_t)1.0f, (bf16_t)1.0f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const t8_bf16x2 pr = {fc[2 * e], fc[2 * e + 1]};
            cs[q] = __builtin_amdgcn_fdot2_f32_bf16(pr, one, cs[q], false);
        }
    };

    // phase G of the loop body; tb = first K-tile of the body
    auto phase = [&](auto gc, const int tb) __attribute__((always_inline)) {
        constexpr int G = decltype(gc)::value;
        constexpr int PH = G % 4, BUF = (G / 4) % 2, NB = BUF ^ 1;
        constexpr int RSLOT = (G + 2) % S;                                  // the half-tile read in this phase: seq r + 2
        constexpr int SSLOT = G % S, SKIND = (G + S) % 4;                   // the half-tile staged in this phase: seq r + S
        // load section
        if constexpr (PH == 0) read_b(sb_[NB], IC<RSLOT>{});                // B-h1(t)
        else if constexpr (PH == 1) read_a(fa[1], IC<RSLOT>{});             // A-h1(t)
        else if constexpr (PH == 2) read_a(fa[0], IC<RSLOT>{});             // A-h0(t+1)
        else read_b(sb_[NB], IC<RSLOT>{});                                  // B-h0(t+1)
        __builtin_amdgcn_sched_barrier(0);
        stage(IC<SSLOT>{}, IC<SKIND>{}, tb + (G + S) / 4);
        t8_wait_vm_n<S - 3>();
        // multiply section
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        constexpr int QA = (PH >= 2) ? 1 : 0, QB = (PH == 1 || PH == 2) ? 1 : 0;
        const bf16x8 (&a)[FA] = fa[QA];
        const bf16x8 (&b)[2] = (PH == 0 || PH == 3) ? sb_[BUF] : sb_[NB];
#pragma unroll
        for (int i = 0; i < FA; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[QA * FA + i][QB * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[QA * FA + i][QB * 2 + j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (PH == 0 || PH == 3) asm volatile("" : "+v"(sb_[NB][0]), "+v"(sb_[NB][1]));
        else if constexpr (PH == 1) asm volatile("" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[1][2]));
        else asm volatile("" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]));
        if constexpr (CSB) { if constexpr (PH == 0) colsum_q(1); else if constexpr (PH == 3) colsum_q(0); }
        else { if constexpr (PH == 1) colsum_q(1); else if constexpr (PH == 2) colsum_q(0); }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: the whole ring is requested (seq 0 .. S-1); A-h0, B-h0, B-h1 of K-tile 0 must have landed
    {
        auto pro = [&](auto qc) __attribute__((always_inline)) { constexpr int Q = decltype(qc)::value; stage(IC<Q>{}, IC<Q % 4>{}, Q / 4); };
        t8_unroll<0, S>(pro);
    }
    t8_wait_vm_n<S - 3>();
    __builtin_amdgcn_s_barrier();
    read_a(fa[0], IC<0>{});
    read_b(sb_[0], IC<1>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(sb_[0][0]), "+v"(sb_[0][1]));
    colsum_q(0);
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int body = 0; body < nbody; ++body) {
        const int tb = body * (PER / 4);
        auto ph = [&](auto gc) __attribute__((always_inline)) { phase(gc, tb); };
        t8_unroll<0, PER>(ph);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the pieces requested behind the row range)

    float* pt = p.partial + ((long)ms * tiles_ij + t) * (BI * T8_BJ) + (long)wid * (2 * FA * 4 * 256) + lane * 4;
#pragma unroll
    for (int i = 0; i < 2 * FA; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(pt + (i * 4 + j) * 256) = acc[i][j];
    if (p.colsum != nullptr && (CSB ? ti : tc) == 0) {
        float* cp = p.partial + (long)gridDim.x * (BI * T8_BJ) + (CSB ? ((long)ms * tiles_c + tc) * T8_BJ : ((long)ms * tiles_i + ti) * BI);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v = cs[q];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if constexpr (CSB) { if (lg == 0) cp[wc * 64 + q * 32 + wr * 16 + li] = v; }
            else if (lg == 0 && wc < FA) cp[wr * (BI / 2) + q * 16 * FA + wc * 16 + li] = v;
        }
    }
}

// second phase: the row splits of one 16-byte piece of the output tile are summed by FOUR threads (a quarter of the splits each, four loads in
// flight per thread: one thread per piece left 4.5 waves per CU with 4 KB in flight each - 17 us for 50 MB), combined through LDS in a fixed
// order (bit-repeatable) and added into the parameter-layout gradient; trailing blocks: the bias-gradient partials.
// SWAP: the kernel ran on swapped operands (its rows are the output's columns): p.NI / p.K are the SWAPPED extents, the destination
// strides the caller's.
template <int FA, bool SWAP>
__global__ __launch_bounds__(256) void tn8p_reduce_kernel(const dist_gemm_tn_args p, const int msplit, const int tiles_i, const int tiles_c, const int nblk) {
    constexpr int BI = 64 * FA, TILE = BI * T8_BJ;
    const int tiles_ij = tiles_i * tiles_c;
    const long total4 = (long)tiles_ij * (TILE / 4);
    const int tid = threadIdx.x, gq = tid >> 6;
    const long e = (long)blockIdx.x * 64 + (tid & 63);
    const int per = (msplit + 3) / 4, s0 = gq * per, s1 = min(msplit, s0 + per);
    __shared__ f32x4 red[3][64];
    if (e >= total4) {
        const long c = e - total4;
        if (gq != 0 || !p.colsum || c >= (SWAP ? p.K : p.NI)) return;      // (whole trailing blocks: no barrier below is skipped by a part of a block)
        const float* __restrict__ cp = p.partial + (long)nblk * TILE + c;
        float a = 0.f;
        for (int s = 0; s < msplit; ++s) a += cp[(long)s * (SWAP ? tiles_c * T8_BJ : tiles_i * BI)];
        atomicAdd(p.colsum + c, a);
        if (p.colsum2) atomicAdd(p.colsum2 + c, a);
        return;
    }
    const int t = (int)(e / (TILE / 4)), r = (int)(e % (TILE / 4));
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(p.partial + (long)t * TILE) + r;
    const long stride4 = (long)tiles_ij * (TILE / 4);
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int s = s0;
    for (; s + 3 < s1; s += 4) {
        a0 += src[(long)s * stride4];
        a1 += src[(long)(s + 1) * stride4];
        a2 += src[(long)(s + 2) * stride4];
        a3 += src[(long)(s + 3) * stride4];
    }
    for (; s < s1; ++s) a0 += src[(long)s * stride4];
    f32x4 v = (a0 + a1) + (a2 + a3);
    if (gq > 0) red[gq - 1][tid & 63] = v;
    __syncthreads();
    if (gq != 0) return;
    v = ((v + red[0][tid]) + red[1][tid]) + red[2][tid];
    const int ti = t % tiles_i, tc = t / tiles_i;
    // piece r = ((wave * 2*FA + i) * 4 + j) * 64 + lane
    const int lane = r & 63, j = (r >> 6) & 3, iw = r >> 8, i = iw % (2 * FA), wid = iw / (2 * FA);
    const int wr = wid >> 2, wc = wid & 3, li = lane & 15, lg = lane >> 4;
    const int row = ti * BI + wr * (BI / 2) + (i / FA) * (16 * FA) + (i % FA) * 16 + li;
    const int col = tc * T8_BJ + wc * 64 + (j >> 1) * 32 + (j & 1) * 16 + lg * 4;
    if (row >= p.NI || col >= p.K) return;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (col + q < p.K) atomicAdd(SWAP ? tn8_dst(p, col + q, row) : tn8_dst(p, row, col + q), v[q]);
}

// S = slots of the ring (20: 160 KB of LDS, 136 KB in flight)
template <int S, bool SWAP>
int launch_tn8p(const dist_gemm_tn_args& a, hipStream_t s, const int max_blocks) {
    constexpr int FA = 3, BI = 64 * FA, TILE = BI * T8_BJ;
    constexpr int PER0 = (S % 4 == 0) ? S : 2 * S;
    constexpr int GRAIN = 32 * ((PER0 / 4) % 2 == 0 ? PER0 / 4 : PER0 / 2);     // rows per loop body: a row range is a whole number of them
    constexpr size_t LDS = (size_t)S * 32 * 256;
    const int tiles_i = (a.NI + BI - 1) / BI, tiles_c = (a.K + T8_BJ - 1) / T8_BJ;
    const long tiles = (long)tiles_i * tiles_c;
    long msplit = max_blocks / tiles;
    const long max_split = (a.M + 511) / 512;             // at least 512 rows per block
    if (msplit > max_split) msplit = max_split;
    if (msplit < 1) msplit = 1;
    int chunk = (int)((a.M + msplit - 1) / msplit);
    chunk = (chunk + GRAIN - 1) / GRAIN * GRAIN;
    msplit = (a.M + chunk - 1) / chunk;
    const long nblk = tiles * msplit;
    if (nblk * TILE + msplit * (SWAP ? (long)tiles_c * T8_BJ : (long)tiles_i * BI) > a.partial_elems) return 0;      // not taken: gemm_tn_kernel
    static DistSmemOnce attr;
    auto kern = gemm_tn8r_kernel<S, SWAP>;
    RUN_(dist_max_smem(attr, reinterpret_cast<const void*>(kern), LDS));
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), LDS, s, a, chunk, tiles_i, tiles_c);
    HIP_CHECK_RET(hipGetLastError());
    const long pieces = tiles * (TILE / 4);
    const long blocks = (pieces + 63) / 64 + (a.colsum ? ((SWAP ? (long)tiles_c * T8_BJ : (long)tiles_i * BI) + 63) / 64 : 0);
    hipLaunchKernelGGL((tn8p_reduce_kernel<FA, SWAP>), dim3((unsigned)blocks), dim3(256), 0, s, a, (int)msplit, tiles_i, tiles_c, (int)nblk);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}

}  // namespace

// 1 = launched, 0 = not this kernel's shape (the caller falls back to gemm_tn_kernel), < 0 = error
int dist_k_gemm_tn8p(const dist_gemm_tn_args* a, hipStream_t s) {
    if (a->dtype != DIST_BF16 || !a->use_tr || a->taps != 1) return 0;
    if (a->amap.mode != DIST_RM_PLAIN || a->bmap.mode != DIST_RM_PLAIN) return 0;
    if (!a->partial) return 0;
    if (a->NI < 192 || a->K < 192 || a->M < 8192) return 0;
    if (a->lda % 8 || a->ldb % 8 || ((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15)) return 0;
    // 32-bit byte offsets inside a row range; the scalar offsets of K-tiles staged behind a block's last row reach up to ~480 rows further
    if ((long)(a->M + 512) * a->lda >= (1l << 30) || (long)(a->M + 512) * a->ldb >= (1l << 30)) return 0;
    static const int mode = dist_knob("DIST_AMD_TN8P", 1);        // 0: gemm_tn_kernel for everything (the A/B reference)
    if (!mode) return 0;
    static const int max_blocks = dist_knob("DIST_AMD_TN8P_BLOCKS", 96);    // partial bytes = blocks x 196 KB, written and read again, and a block holds its CU for the whole
                                                                           // launch: 96 blocks leave the other CUs to the data-gradient chain the backward pass waits for
                                                                           // (step 17.90-17.98 ms with 256, 17.80-17.86 with 128, 17.50-17.55 with 96 here and in gemm_tn.hip;
                                                                           // alone 256 is the fastest: profiles/r04_tn8p.md)
    // the orientation with the smaller padded tile area; swapped: the kernel's A is the caller's B (the destination strides stay the caller's)
    auto pad = [](int x, int q) { return (long)((x + q - 1) / q * q); };
    const int mb = a->max_blocks > 0 ? a->max_blocks : max_blocks;
    if (pad(a->NI, 192) * pad(a->K, 256) <= pad(a->K, 192) * pad(a->NI, 256)) return launch_tn8p<20, false>(*a, s, mb);
    dist_gemm_tn_args b = *a;
    b.A = a->B; b.B = a->A; b.lda = a->ldb; b.ldb = a->lda; b.NI = a->K; b.K = a->NI;
    return launch_tn8p<20, true>(b, s, mb);
}
