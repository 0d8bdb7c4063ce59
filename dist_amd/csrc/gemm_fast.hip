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
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 256, BN = 256, BK = 32, NT = 512;
constexpr int STAGES = 4, AHEAD = 3;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;          // 32 KB
constexpr int A_BYTES = BM * BK * 2;                      // 16 KB
constexpr int EPI_BYTES = 128 * 64 * 2;                   // per-wave staging region: 128 rows x 128 B

typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

DEV int swz(int q) { return (4 - q) & 3; }               // f = {0,3,2,1}

struct FragSet { bf16x8 a[8]; bf16x8 b[4]; };

__global__ __launch_bounds__(NT) void gemm_fast_kernel(const dist_gemm_args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;                // 2 x 4 waves
    const int li = lane & 15, lg = lane >> 4;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (int)((p.M + BM - 1) / BM);
    const int nblk = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {   // bijective XCD remap: consecutive ids share an XCD (and its L2); n-tiles of one A panel adjacent
        const int q = nblk / 8, r = nblk % 8, x = bid % 8, y = bid / 8;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = (int)p.M, N = p.N, K = p.K;
    const bf16_t* __restrict__ A = static_cast<const bf16_t*>(p.A);
    const bf16_t* __restrict__ B = static_cast<const bf16_t*>(p.B);

    // ---- LDS-DMA source addresses: each wave moves 2 x 1 KB of A and 2 x 1 KB of B per stage.
    // 1 KB = 16 rows x 64 B; lane l lands at row (l>>2), physical chunk (l&3) and therefore fetches
    // logical chunk (l&3) ^ f((l>>4)&3) of that row.
    const int lrow = lane >> 2;
    const int lchunk = (lane & 3) ^ swz((lane >> 4) & 3);
    const bf16_t* ga[2];
    const bf16_t* gb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (wid * 2 + j) * 16 + lrow;
        ga[j] = A + (long)rowmap_src(p.amap, min(m0 + r, M - 1), 0, 1) * p.lda + lchunk * 8;   // plain / strided / skip-cls rows
        gb[j] = B + (long)min(n0 + r, N - 1) * p.ldb + lchunk * 8;
    }
    // piece q of tile kt: q = 0,1 -> A rows of this wave's two 16-row groups, q = 2,3 -> B
    auto dma_piece = [&](int kt, int q) {
        char* sb = smem + (kt % STAGES) * STAGE_BYTES;
        const int k0 = kt * BK, j = q & 1;
        const int off = (wid * 2 + j) * 1024;
        if (q < 2) __builtin_amdgcn_global_load_lds((gbl_ptr)(ga[j] + k0), (lds_ptr)(sb + off), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gbl_ptr)(gb[j] + k0), (lds_ptr)(sb + A_BYTES + off), 16, 0, 0);
    };
    auto stage_issue = [&](int kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dma_piece(kt, q);
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
    auto step = [&](FragSet& cur, FragSet& nxt, int kt) {
        const bool more = kt + 1 < nk, refill = kt + AHEAD < nk;
        if (more) {
            // DMA groups outstanding here: tiles kt+1 and kt+2 (4 pieces each); kt+1 must be complete
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                 // tile kt+1 landed for every wave; tile kt-1 no longer read by anyone
            frag_read(nxt, kt + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 2 * g; i < 2 * g + 2; ++i)
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

    stage_issue(0);
    stage_issue(1);
    stage_issue(2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // tile 0 landed (this wave's share)
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

    // ---- epilogue through LDS: per-wave region of 128 rows x 128 B, 16-B chunk c of row r at chunk c ^ (r & 7) ----
    __syncthreads();                                      // every wave is done reading the operand ring
    char* ew = smem + wid * EPI_BYTES;
    bf16_t* __restrict__ C = static_cast<bf16_t*>(p.C);
    bf16_t* __restrict__ C2 = static_cast<bf16_t*>(p.C2);
    const bf16_t* __restrict__ R = static_cast<const bf16_t*>(p.res);
    const int flags = p.flags;
    const int mw = m0 + wm * 128, nw = n0 + wn * 64;
    const int crow = lane >> 3, cchunk = lane & 7;        // coalesced pass: 8 lanes cover one 128-B row
    const int icls = p.omap.mode == DIST_OM_INSERTCLS ? p.omap.p0 : 0;
    auto dest_row = [&](int m) -> long { return icls ? (long)(m / icls) * (icls + 1) + 1 + m % icls : (long)m; };

    if (flags & DIST_EPI_RES) {
        // all 16 row-pieces of the residual tile are requested back to back (the operand fragment registers are
        // dead by now), so ONE memory latency is exposed instead of one per batch
        uint4 rv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 8 + crow;
            const int m = mw + r, n = nw + cchunk * 8;
            rv[it] = make_uint4(0, 0, 0, 0);
            if (m < M && n < N) rv[it] = *reinterpret_cast<const uint4*>(R + dest_row(m) * p.ldres + n);
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 8 + crow;
            *reinterpret_cast<uint4*>(ew + r * 128 + ((cchunk ^ (r & 7)) << 4)) = rv[it];
        }
    }
    float bias4[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = nw + j * 16 + lg * 4 + r;
            bias4[j][r] = ((flags & DIST_EPI_BIAS) && n < N) ? p.bias[n] : 0.f;
        }
    const bool act_only = (flags & DIST_EPI_ACT2) && !C;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int r = i * 16 + li;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16_t* slot = reinterpret_cast<bf16_t*>(ew + r * 128 + ((((j << 1) | (lg >> 1)) ^ (r & 7)) << 4) + ((lg & 1) << 3));
            float v[4] = {acc[i][j][0] + bias4[j][0], acc[i][j][1] + bias4[j][1], acc[i][j][2] + bias4[j][2], acc[i][j][3] + bias4[j][3]};
            if (flags & DIST_EPI_RES) {
                float x[4];
                load4(slot, x);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += x[q];
            }
            if (act_only) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = qgelu(v[q]);
            }
            store4(slot, v);
        }
    }
    auto flush = [&](bf16_t* __restrict__ dst, int ld) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int r = it * 8 + crow;
            const int m = mw + r, n = nw + cchunk * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(ew + r * 128 + ((cchunk ^ (r & 7)) << 4));
            if (m < M && n < N) *reinterpret_cast<uint4*>(dst + dest_row(m) * ld + n) = v;
        }
    };
    if (act_only) {
        flush(C2, p.ldc2);
    } else {
        flush(C, p.ldc);
        if (flags & DIST_EPI_ACT2) {                      // second output = quickgelu(stored value)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = i * 16 + li;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bf16_t* slot = reinterpret_cast<bf16_t*>(ew + r * 128 + ((((j << 1) | (lg >> 1)) ^ (r & 7)) << 4) + ((lg & 1) << 3));
                    float x[4];
                    load4(slot, x);
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = qgelu(x[q]);
                    store4(slot, x);
                }
            }
            flush(C2, p.ldc2);
        }
    }
}

}  // namespace

bool dist_k_gemm_fast_eligible(const dist_gemm_args* a) {
    if (a->dtype != DIST_BF16 || a->taps != 1) return false;
    if (a->amap.mode != DIST_RM_PLAIN && a->amap.mode != DIST_RM_STRIDED && a->amap.mode != DIST_RM_SKIPCLS) return false;
    if (a->omap.mode != DIST_OM_PLAIN && a->omap.mode != DIST_OM_INSERTCLS) return false;
    if (a->flags & DIST_EPI_MULG) return false;
    if (a->K % BK || a->K < 4 * BK || a->N % 64 || a->N < 256 || a->M < 1024) return false;
    if (a->N % BN > 0 && (a->N % BN < 128 || a->K < 768)) return false;   // a half-empty last column tile only pays for long K
    if (a->lda % 8 || a->ldb % 8 || a->ldc % 8 || a->ldc2 % 8 || a->ldres % 8) return false;
    return true;
}

// returns 1 if handled, 0 if the shape does not qualify (caller falls through), <0 on error
int dist_k_gemm_fast(const dist_gemm_args* a, hipStream_t s) {
    if (!dist_k_gemm_fast_eligible(a)) return 0;
    constexpr size_t smem = (size_t)STAGES * STAGE_BYTES;
    static_assert(8 * EPI_BYTES <= STAGES * STAGE_BYTES, "epilogue staging fits in the operand ring");
    static bool attr_done = false;
    if (!attr_done) {
        HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_fast_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_done = true;
    }
    const long tiles = ((a->M + BM - 1) / BM) * ((a->N + BN - 1) / BN);
    hipLaunchKernelGGL(gemm_fast_kernel, dim3((unsigned)tiles), dim3(NT), smem, s, *a);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}
