// Fused TemporalNet forward (reference models/module_zoo/branches/dist.py:48-65):
//
//     X' = g( X + conv_{1x3x3}( g( conv_{3x1x1}( LN_C(X) ) ) ) )          g = QuickGELU
//
// on the channels-last temporal map X[b, k, n, c] (rows (b*T + k)*N + n, c contiguous).  ONE launch per layer instead of
// LayerNorm + two row-mapped GEMMs: the normalised tensor U and the activated tensor V exist only in LDS (they are written out
// only when the caller asks for them), the nine spatial taps re-read one frame's V tile from LDS instead of L2, and the stores
// are whole contiguous 37 KB frame blocks.
//
// Work decomposition: one workgroup (8 waves) per (clip, frame k); the frames of a clip go to the same XCD (they share the
// two neighbour frames through that XCD's L2).  Two workgroups share a CU (<= 80 KB LDS, <= 128 registers per lane), so
// one's load / LayerNorm / store phases lie under the other's MFMA phases.
//
//   temporal taps d = -1, 0, +1 (frames outside [0, T) contribute nothing - Conv3d zero padding - and are skipped):
//       X[k+d] -> registers (4 lanes per row) -> LayerNorm -> bf16 U tile in LDS -> z += U . W1[d]^T   (MFMA, A = U tile)
//   z + b1 -> bf16 -> LDS; streamed out as whole rows (z is what backward needs) while V = g(z) replaces it in LDS
//   spatial taps (dy, dx): p += V[(y+dy, x+dx)] . W2[dy,dx]^T, the accumulators start at X[k] + b2; a lane whose tap leaves
//       the G x G plane reads the tile's all-zero slot
//   p -> bf16 -> LDS -> streamed out together with X' = g(p)
//
// Weights: the packed forward layout W[n][tap*Ct + c] (bf16) travels L2 -> LDS by `buffer_load ... lds` one tap (Ct/32 k-blocks
// of [Ct][32]) per barrier interval, double buffered: the tap after the one being multiplied is in flight.
//
// LDS tile: slot s (a position of the plane, or the zero slot N) holds Ct bf16 = Ct/8 16-byte chunks; chunk c of slot s lives at
// chunk (c & ~3) | ((c & 3) ^ (2 * ((s >> 2) & 1))): conflict-free ds_read_b128 fragment reads for ANY alignment of the 16
// consecutive slots a fragment covers (the spatial taps shift them by dy*G + dx) given the real service groups of the
// instruction (lanes {0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS).
//
// Rounding points (the bf16 oracle, oracle/dist_oracle.py `temporal_net`, has the same): U, z, V = g(bf16 z), p = X + conv + b2, X' = g(bf16 p).
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((address_space(3))) void* tn_lds_ptr;
typedef __attribute__((ext_vector_type(2))) __bf16 tn_bf16x2;

struct TnArgs {
    const bf16_t* X; const bf16_t* W1; const bf16_t* W2;
    const float *b1, *b2, *lnw, *lnb;
    bf16_t *z, *p, *Xp, *U, *V;
    float *mean, *rstd;
    int clips, T, G, N, tk; float eps;
    int stagger;  // DIST_AMD_TNET_STAGGER: the second workgroup of a CU starts this many s_sleep(127) (3.4 us each) late
    int dbg;      // measurement knob DIST_AMD_TNET_DBG (results are WRONG with any bit set): 1 = no weight DMA, 2 = no MFMAs, 4 = no row loads, 8 = no stores, 16 = no LayerNorm math / U tile, 32 = no stream-out, 64 = no accumulator -> tile, 128 = no block barriers
};

#define TN_SYNC() do { if (!(p.dbg & 128)) __syncthreads(); } while (0)
// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15); every lane of the row receives it.  Four v_add_f32_dpp
DEV float tn_row_sum16(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));   // row_mirror
    return v;
}
DEV int tn_pchunk(const int slot, const int c) { return (c & ~3) | ((c & 3) ^ (((slot >> 2) & 1) << 1)); }

template <int CT, bool SAVE_UV>
__global__ __launch_bounds__(512, 4) void tnet_fwd_kernel(const TnArgs p) {
    constexpr int KBT = CT / 32;                 // k-blocks per tap
    constexpr int NTL = CT / 16;                 // 16-wide column tiles
    constexpr int ROWB = CT * 2;                 // bytes per slot
    constexpr int CPR = CT / 8;                  // 16-byte chunks per slot
    constexpr int LNC = CPR / 4;                 // chunks per lane in the LayerNorm layout (4 lanes per row)
    constexpr int SLOTB = CT * 64;               // one k-block of weights: [CT][32] bf16
    constexpr int NPIECE = SLOTB / 1024;         // 1 KB LDS-DMA pieces per k-block
    constexpr int GPIECES = KBT * NPIECE;        // pieces per tap
    static_assert(CT % 32 == 0 && CT <= 128 && GPIECES <= 24, "channel count");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int N = p.N, G = p.G, T = p.T;
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int clip = xcd + 8 * (qb / T), k = qb - (qb / T) * T;
    if (clip >= p.clips) return;

    const int act_bytes = ((N + 1) * ROWB + 1023) & ~1023;
    char* act = smem;
    char* ring = smem + act_bytes;                                   // [2][KBT][CT][32] bf16
    float* par = reinterpret_cast<float*>(ring + 2 * KBT * SLOTB);  // gamma, beta, b1, b2
    const int MT = (N + 15) >> 4;
    // The two workgroups of a CU would walk the same phases in step (LayerNorm VALU phases together, MFMA phases together).  Workgroups
    // are dispatched to an XCD's 32 CUs in order, so the 32 that follow the first 32 of an XCD are the second tenants: they start late.
    if (p.stagger > 0 && ((qb >> 5) & 1)) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // ---- parameters and the zero slot
    for (int i = tid; i < 4 * CT; i += 512) {
        const int w = i / CT, c = i - w * CT;
        par[i] = w == 0 ? p.lnw[c] : (w == 1 ? p.lnb[c] : (w == 2 ? p.b1[c] : p.b2[c]));
    }
    for (int i = tid; i < ROWB / 4; i += 512) reinterpret_cast<unsigned*>(act + N * ROWB)[i] = 0u;

    // ---- weight stream: tap groups gg = 0 .. n1 + 9 - 1 (valid temporal taps first), group gg in ring half gg & 1
    const int tk = p.tk, thalf = tk >> 1;
    const int d_first = max(-thalf, -k), d_last = min(thalf, T - 1 - k);          // temporal offsets with a frame behind them
    const int n1 = d_last - d_first + 1, ngroups = n1 + 9;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W1), 0, CT * tk * CT * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W2), 0, CT * 9 * CT * 2, 0x00020000);
    const int lrow = lane >> 2;
    const int lch = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);       // source-side chunk swizzle of a [16 rows][64 B] piece
    const unsigned wv1 = ((unsigned)lrow * (unsigned)(tk * CT) + lch * 8) * 2u;
    const unsigned wv2 = ((unsigned)lrow * (unsigned)(9 * CT) + lch * 8) * 2u;
    auto issue_piece = [&](const int gg, const int idx) __attribute__((always_inline)) {   // piece idx of group gg (wave-uniform)
        const int kb = idx / NPIECE, j = idx - kb * NPIECE;
        char* dst = ring + ((gg & 1) * KBT + kb) * SLOTB + j * 1024;
        if (gg < n1) {
            const int tap = d_first + thalf + gg;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (tn_lds_ptr)dst, 16, wv1, (j * 16 * tk * CT + tap * CT + kb * 32) * 2, 0, 0);
        } else {
            const int tap = gg - n1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (tn_lds_ptr)dst, 16, wv2, (j * 16 * 9 * CT + tap * CT + kb * 32) * 2, 0, 0);
        }
    };
    auto issue_group = [&](const int gg) __attribute__((always_inline)) {
        if (p.dbg & 1) return;
        if (wid < GPIECES) issue_piece(gg, wid);
        if constexpr (GPIECES > 8) { if (wid + 8 < GPIECES) issue_piece(gg, wid + 8); }
        if constexpr (GPIECES > 16) { if (wid + 16 < GPIECES) issue_piece(gg, wid + 16); }
    };

    // ---- LayerNorm layout: rows wid*32 + pp*16 + (lane >> 2), chunks (lane & 3) + 4 e
    const int lq = lane & 3;
    const long frame_rows = (long)(clip * T + k) * N;                 // first row of frame k
    auto ln_load = [&](const int d, bf16x8 (&raw)[2][LNC]) __attribute__((always_inline)) {
        const bf16_t* src = p.X + (frame_rows + (long)d * N) * CT;
        if (p.dbg & 4) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int e = 0; e < LNC; ++e) raw[pp][e] = bf16x8{(bf16_t)(float)lane, 0, 0, 0, 0, 0, 0, 0};
            return;
        }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int row = min(wid * 32 + pp * 16 + (lane >> 2), N - 1);
#pragma unroll
            for (int e = 0; e < LNC; ++e) raw[pp][e] = *reinterpret_cast<const bf16x8*>(src + (long)row * CT + (lq + 4 * e) * 8);
        }
    };
    // (statistics straight from the raw vectors, then one chunk at a time: the accumulators of the temporal taps stay live across this)
    auto ln_apply = [&](const int d, const bf16x8 (&raw)[2][LNC]) __attribute__((always_inline)) {
        constexpr float invC = 1.f / (float)CT;
        if (p.dbg & 16) return;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int row = wid * 32 + pp * 16 + (lane >> 2);
            // one pass over the packed bf16 pairs: v_dot2c_f32_bf16 with (1, 1) sums them, with themselves sums their squares (the products
            // of two bf16 values are exact in fp32; the fp32 sums over 96 channels carry ~1e-6 relative error, so E[x^2] - mean^2 is good to
            // ~1e-6 (1 + mean^2 / var) - far below the bf16 rounding of U for any realistic row).  The two-pass form costs 5 VALU operations
            // per element against 1 here, and this workgroup normalises three frames (profiles/r03_tnet_fused.md).
            float s = 0.f, q = 0.f;
            const tn_bf16x2 one = {(bf16_t)1.0f, (bf16_t)1.0f};
#pragma unroll
            for (int e = 0; e < LNC; ++e)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const tn_bf16x2 pr = {raw[pp][e][2 * i], raw[pp][e][2 * i + 1]};
                    s = __builtin_amdgcn_fdot2_f32_bf16(pr, one, s, false);
                    q = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, q, false);
                }
            const float mean = wave_sum(s, 4) * invC;
            const float rstd = rsqrtf(fmaxf(wave_sum(q, 4) * invC - mean * mean, 0.f) + p.eps);
            const float nmr = -mean * rstd;
            if (row < N) {
                if (d == 0 && lq == 0) { p.mean[frame_rows + row] = mean; p.rstd[frame_rows + row] = rstd; }
#pragma unroll
                for (int e = 0; e < LNC; ++e) {
                    const int c = lq + 4 * e;
                    bf16x8 o;
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const float4 gv = *reinterpret_cast<const float4*>(par + c * 8 + hh * 4);
                        const float4 bv = *reinterpret_cast<const float4*>(par + CT + c * 8 + hh * 4);
                        o[hh * 4 + 0] = (bf16_t)__builtin_fmaf(__builtin_fmaf((float)raw[pp][e][hh * 4 + 0], rstd, nmr), gv.x, bv.x);
                        o[hh * 4 + 1] = (bf16_t)__builtin_fmaf(__builtin_fmaf((float)raw[pp][e][hh * 4 + 1], rstd, nmr), gv.y, bv.y);
                        o[hh * 4 + 2] = (bf16_t)__builtin_fmaf(__builtin_fmaf((float)raw[pp][e][hh * 4 + 2], rstd, nmr), gv.z, bv.z);
                        o[hh * 4 + 3] = (bf16_t)__builtin_fmaf(__builtin_fmaf((float)raw[pp][e][hh * 4 + 3], rstd, nmr), gv.w, bv.w);
                    }
                    *reinterpret_cast<bf16x8*>(act + row * ROWB + tn_pchunk(row, c) * 16) = o;
                    if (SAVE_UV && d == 0) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p.U + (frame_rows + row) * CT + c * 8));
                }
            }
        }
    };

    // ---- this wave's output tiles: m-tiles wid and wid + 8, all NTL column tiles
    // (a wave whose second tile lies beyond the plane multiplies the zero slot: waves 5-7 of a 13-tile plane; the SIMD that carries
    // waves 0 and 4 has four real tiles, so the launch is not longer for it, and the loop has no branches)
    f32x4 acc[2][NTL];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int cob = (lg ^ ((4 - ((li >> 2) & 3)) & 3)) * 16;          // B-fragment chunk of this lane inside a [16][64 B] piece
    // multiply one tap: A fragments of m-tile i at byte offset ab[i] (+ 64 per k-block) of the act tile
    auto mma_tap = [&](const int gg, const unsigned (&ab)[2]) __attribute__((always_inline)) {
        const char* bs0 = ring + (gg & 1) * KBT * SLOTB + li * 64 + cob;
        if (p.dbg & 2) return;
        constexpr int NH = NTL / 2;                                     // B fragments in two batches: fewer live registers
#pragma unroll
        for (int kb = 0; kb < KBT; ++kb) {
            bf16x8 fa[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(act + ab[i] + kb * 64);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8 fb[NH];
#pragma unroll
                for (int j = 0; j < NH; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bs0 + kb * SLOTB + (h * NH + j) * 1024);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NH; ++j)
                        acc[i][h * NH + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][h * NH + j], 0, 0, 0);   // D[n][m]
            }
        }
    };
    // accumulators (+ bias) -> bf16 -> act tile.  Lane (li, lg) holds position m = tile*16 + li, channels j*16 + 4 lg .. + 3
    auto acc_to_tile = [&](const float* bias) __attribute__((always_inline)) {
        if (p.dbg & 64) return;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = (wid + 8 * i) * 16 + li;
            if (m < N) {
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + j * 16 + lg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    bf16x4 o = {(bf16_t)(acc[i][j][0] + bv.x), (bf16_t)(acc[i][j][1] + bv.y), (bf16_t)(acc[i][j][2] + bv.z), (bf16_t)(acc[i][j][3] + bv.w)};
                    *reinterpret_cast<bf16x4*>(act + m * ROWB + tn_pchunk(m, 2 * j + (lg >> 1)) * 16 + (lg & 1) * 8) = o;
                }
            }
        }
    };
    // the act tile leaves as whole rows: `a` = the tile as it is, `b` (optional) = g(tile); with `keep_act` g(tile) also replaces the tile
    auto stream_out = [&](bf16_t* a, bf16_t* b, const bool keep_act) __attribute__((always_inline)) {
        bf16_t* ga = a + frame_rows * CT;
        bf16_t* gb = b ? b + frame_rows * CT : nullptr;
        if (p.dbg & 32) return;
        int slot = tid / CPR, c = tid - slot * CPR;
        constexpr int DS = 512 / CPR, DC = 512 - DS * CPR;
        for (int idx = tid; idx < N * CPR; idx += 512) {
            char* lp = act + slot * ROWB + tn_pchunk(slot, c) * 16;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(lp);
            if (!(p.dbg & 8)) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(ga + (long)idx * 8));
            bf16x8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (bf16_t)qgelu_t<bf16_t>((float)v[i]);
            if (gb && !(p.dbg & 8)) __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(gb + (long)idx * 8));
            if (keep_act) *reinterpret_cast<bf16x8*>(lp) = o;
            slot += DS; c += DC;
            if (c >= CPR) { c -= CPR; ++slot; }
        }
    };

    // =========================================== temporal convolution ===========================================
    issue_group(0);
    unsigned ab[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min((wid + 8 * i) * 16 + li, N);                 // rows beyond N read the zero slot
        ab[i] = (unsigned)(m * ROWB + ((lg ^ (((m >> 2) & 1) << 1)) << 4));
    }
    TN_SYNC();                                                   // parameters + zero slot visible
    for (int g = 0; g < n1; ++g) {
        {   // (the rows of the next frame are NOT requested ahead, under the MFMAs of this one: 24 more live registers per lane spill at
            // the 128 the two-workgroups-per-CU shape allows; the other workgroup of the CU covers the latency)
            bf16x8 raw[2][LNC];
            ln_load(d_first + g, raw);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // those rows and this wave's weight pieces of group g
            ln_apply(d_first + g, raw);
        }
        TN_SYNC();                                               // U tile + weights of group g complete; ring half (g+1)&1 free
        issue_group(g + 1);
        mma_tap(g, ab);
        TN_SYNC();                                               // U tile no longer read
    }
    // =========================================== z -> V =========================================================
    acc_to_tile(par + 2 * CT);
    // residual X[k] in the accumulator layout (8-byte pieces; L2 hits - this workgroup has just read the frame)
    bf16x4 res[2][NTL];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min((wid + 8 * i) * 16 + li, N - 1);
#pragma unroll
        for (int j = 0; j < NTL; ++j) res[i][j] = *reinterpret_cast<const bf16x4*>(p.X + (frame_rows + m) * CT + j * 16 + lg * 4);
    }
    TN_SYNC();
    stream_out(p.z, SAVE_UV ? p.V : nullptr, true);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const float4 bv = *reinterpret_cast<const float4*>(par + 3 * CT + j * 16 + lg * 4);
            acc[i][j] = f32x4{(float)res[i][j][0] + bv.x, (float)res[i][j][1] + bv.y, (float)res[i][j][2] + bv.z, (float)res[i][j][3] + bv.w};
        }
    // =========================================== spatial convolution ============================================
    int my[2], mx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = (wid + 8 * i) * 16 + li;
        my[i] = m < N ? m / G : -4; mx[i] = m - (m / G) * G;           // rows beyond N: every tap invalid
    }
    for (int tap = 0; tap < 9; ++tap) {
        const int g = n1 + tap;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TN_SYNC();                                               // V tile (first tap) + weights of this tap complete; other ring half free
        if (g + 1 < ngroups) issue_group(g + 1);
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        unsigned at[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = my[i] + dy, x = mx[i] + dx;
            const bool ok = y >= 0 && y < G && x >= 0 && x < G;
            const int s = ok ? y * G + x : N;
            at[i] = (unsigned)(s * ROWB + ((lg ^ (((s >> 2) & 1) << 1)) << 4));
        }
        mma_tap(g, at);
    }
    TN_SYNC();                                                   // V tile no longer read
    acc_to_tile(nullptr);
    TN_SYNC();
    stream_out(p.p, p.Xp, false);
}

template <int CT, bool SAVE_UV>
int launch_tnet_fwd2(const TnArgs& a, hipStream_t s) {
    const int act_bytes = ((a.N + 1) * CT * 2 + 1023) & ~1023;
    const size_t smem = (size_t)act_bytes + 2 * (CT / 32) * CT * 64 + 4 * CT * sizeof(float);
    if (smem > 160 * 1024) return DIST_ERR_ARG;
    static DistSmemOnce attr;
    RUN_(dist_max_smem(attr, reinterpret_cast<const void*>(tnet_fwd_kernel<CT, SAVE_UV>), smem));
    const int groups = (a.clips + 7) / 8;
    hipLaunchKernelGGL((tnet_fwd_kernel<CT, SAVE_UV>), dim3((unsigned)(groups * 8 * a.T)), dim3(512), smem, s, a);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
template <int CT>
int launch_tnet_fwd(const TnArgs& a, hipStream_t s) { return (a.U && a.V) ? launch_tnet_fwd2<CT, true>(a, s) : launch_tnet_fwd2<CT, false>(a, s); }


// =====================================================================================================================================
// Backward of the same block, two launches (the temporal taps of dU need dz of the NEIGHBOUR frames, which other workgroups produce):
//
//   tnet_bwd_spatial_kernel   dz = conv_{1x3x3}^T(dp) * g'(z)                 one workgroup per (clip, frame): the dp frame in LDS, nine
//                             flipped taps over it with the data-gradient weights W2b[ci][tap*Ct + co], g'(z) on the accumulators
//   tnet_bwd_temporal_kernel  dX = dp + LN'( conv_{3x1x1}^T(dz) )             the dz frames k+1, k, k-1 through LDS one after the other,
//                             LayerNorm backward on the accumulators (row sums by two lane exchanges), parameter gradients as per-workgroup
//                             partial rows (plain stores) that tnet_dgb_reduce_kernel adds up: no atomics anywhere
//
// dp = dL/dp (already multiplied by g'(p) in the T2I data-gradient epilogue), z / X / mean / rstd as the forward saved them.  The weight
// gradients dW2 = dp^T V(shifted), dW1 = dz^T U(shifted) stay with the row-split weight-gradient GEMM (gemm_tn.hip).
// Shared machinery: the LDS tile layout, the weight stream and the tap multiply of the forward kernel, as device functions.
template <int CT> struct TnK {
    static constexpr int KBT = CT / 32, NTL = CT / 16, ROWB = CT * 2, CPR = CT / 8, SLOTB = CT * 64, NPIECE = SLOTB / 1024, GPIECES = KBT * NPIECE;
};
// one tap (KBT k-blocks of [CT][32]) of a packed weight [CT][ntaps*CT] into ring half `half`; wave-uniform piece distribution
template <int CT> DEV void tn_issue_tap(char* ring, const int half, const __amdgpu_buffer_rsrc_t r, const unsigned wv, const int wcols, const int tap, const int wid) {
    using K = TnK<CT>;
#pragma unroll
    for (int rr = 0; rr < (K::GPIECES + 7) / 8; ++rr) {
        const int idx = wid + 8 * rr;
        if (idx < K::GPIECES) {
            const int kb = idx / K::NPIECE, j = idx - kb * K::NPIECE;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (tn_lds_ptr)(ring + (half * K::KBT + kb) * K::SLOTB + j * 1024), 16, wv,
                                                     (j * 16 * wcols + tap * CT + kb * 32) * 2, 0, 0);
        }
    }
}
template <int CT> DEV void tn_mma_tap(f32x4 (&acc)[2][CT / 16], const char* act, const char* ring, const int half, const unsigned (&ab)[2], const int li, const int cob) {
    using K = TnK<CT>;
    const char* bs0 = ring + half * K::KBT * K::SLOTB + li * 64 + cob;
    constexpr int NH = K::NTL / 2;
#pragma unroll
    for (int kb = 0; kb < K::KBT; ++kb) {
        bf16x8 fa[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(act + ab[i] + kb * 64);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 fb[NH];
#pragma unroll
            for (int j = 0; j < NH; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bs0 + kb * K::SLOTB + (h * NH + j) * 1024);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NH; ++j)
                    acc[i][h * NH + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][h * NH + j], 0, 0, 0);
        }
    }
}
// a whole frame [N][CT] bf16 (contiguous in memory) into the swizzled LDS tile by LDS-DMA: lane l of 1 KB piece j lands at tile byte
// j*1024 + 16 l, i.e. slot s / physical chunk q, and fetches the logical chunk q ^ swizzle(s) of slot s (the swizzle is an involution);
// bytes beyond the frame read as zero through the descriptor's bound (they land in the zero slot / the tile's padding)
template <int CT> DEV void tn_tile_dma(char* act, const bf16_t* frame, const int N, const int wid, const int lane) {
    using K = TnK<CT>;
    const int bytes = N * K::ROWB;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(frame), 0, bytes, 0x00020000);
    const int pieces = (bytes + 1023) >> 10;
    for (int j = wid; j < pieces; j += 8) {
        const int off = j * 1024 + lane * 16;
        const int slot = off / K::ROWB, q = (off - slot * K::ROWB) >> 4;
        const unsigned src = off < bytes ? (unsigned)(slot * K::ROWB + tn_pchunk(slot, q) * 16) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (tn_lds_ptr)(act + j * 1024), 16, src, 0, 0, 0);
    }
}
template <int CT> DEV void tn_acc_to_tile(const f32x4 (&acc)[2][CT / 16], char* act, const int N, const int wid, const int li, const int lg) {
    using K = TnK<CT>;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = (wid + 8 * i) * 16 + li;
        if (m < N) {
#pragma unroll
            for (int j = 0; j < K::NTL; ++j) {
                bf16x4 o = {(bf16_t)acc[i][j][0], (bf16_t)acc[i][j][1], (bf16_t)acc[i][j][2], (bf16_t)acc[i][j][3]};
                *reinterpret_cast<bf16x4*>(act + m * K::ROWB + tn_pchunk(m, 2 * j + (lg >> 1)) * 16 + (lg & 1) * 8) = o;
            }
        }
    }
}
template <int CT> DEV void tn_tile_out(const char* act, bf16_t* dst, const int N, const int tid) {
    using K = TnK<CT>;
    int slot = tid / K::CPR, c = tid - slot * K::CPR;
    constexpr int DS = 512 / K::CPR, DC = 512 - DS * K::CPR;
    for (int idx = tid; idx < N * K::CPR; idx += 512) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(act + slot * K::ROWB + tn_pchunk(slot, c) * 16);
        __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(dst + (long)idx * 8));
        slot += DS; c += DC;
        if (c >= K::CPR) { c -= K::CPR; ++slot; }
    }
}

struct TnBwdArgs {
    const bf16_t *dp, *z, *dz_in, *X, *W2b, *W1b;
    const float *mean, *rstd, *lnw;
    bf16_t *dz, *dX;
    float* dgb;              // [2 * Ct][workgroups] partial (dgamma rows, then dbeta rows)
    int clips, T, G, N, tk, nwg;
};

template <int CT>
__global__ __launch_bounds__(512, 4) void tnet_bwd_spatial_kernel(const TnBwdArgs p) {
    using K = TnK<CT>;
    constexpr int NTL = K::NTL, ROWB = K::ROWB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int N = p.N, G = p.G, T = p.T;
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int clip = xcd + 8 * (qb / T), k = qb - (qb / T) * T;
    if (clip >= p.clips) return;
    const int act_bytes = ((N + 1) * ROWB + 1023) & ~1023;
    char* act = smem;
    char* ring = smem + act_bytes;
    const long frame_rows = (long)(clip * T + k) * N;

    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W2b), 0, CT * 9 * CT * 2, 0x00020000);
    const int lrow = lane >> 2, lch = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    const unsigned wv2 = ((unsigned)lrow * (unsigned)(9 * CT) + lch * 8) * 2u;
    const int cob = (lg ^ ((4 - ((li >> 2) & 3)) & 3)) * 16;

    tn_issue_tap<CT>(ring, 0, r2, wv2, 9 * CT, 0, wid);
    tn_tile_dma<CT>(act, p.dp + frame_rows * CT, N, wid, lane);      // (its tail zeroes the zero slot)
    f32x4 acc[2][NTL];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int my[2], mx[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = (wid + 8 * i) * 16 + li;
        my[i] = m < N ? m / G : -4; mx[i] = m - (m / G) * G;
    }
    // the zero slot (the DMA's last piece zero-fills only what lies inside that piece; zeros over zeros where both write)
    for (int i = tid; i < ROWB / 4; i += 512) reinterpret_cast<unsigned*>(act + N * ROWB)[i] = 0u;
    for (int tap = 0; tap < 9; ++tap) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tap + 1 < 9) tn_issue_tap<CT>(ring, (tap + 1) & 1, r2, wv2, 9 * CT, tap + 1, wid);
        // dz[m] += dp[m - (dy, dx)] . W2b[tap]: the forward tap (dy, dx) seen from the other side
        const int dy = -(tap / 3 - 1), dx = -(tap - (tap / 3) * 3 - 1);
        unsigned at[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int y = my[i] + dy, x = mx[i] + dx;
            const bool ok = y >= 0 && y < G && x >= 0 && x < G;
            const int s = ok ? y * G + x : N;
            at[i] = (unsigned)(s * ROWB + ((lg ^ (((s >> 2) & 1) << 1)) << 4));
        }
        tn_mma_tap<CT>(acc, act, ring, tap & 1, at, li, cob);
    }
    // dz = acc * g'(z): z in the accumulator layout (8-byte pieces straight from memory)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min((wid + 8 * i) * 16 + li, N - 1);
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const bf16x4 zv = *reinterpret_cast<const bf16x4*>(p.z + (frame_rows + m) * CT + j * 16 + lg * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] *= qgelu_grad_t<bf16_t>((float)zv[r]);
        }
    }
    __syncthreads();                                                   // dp tile no longer read
    tn_acc_to_tile<CT>(acc, act, N, wid, li, lg);
    __syncthreads();
    tn_tile_out<CT>(act, p.dz + frame_rows * CT, N, tid);
}

template <int CT>
__global__ __launch_bounds__(512, 4) void tnet_bwd_temporal_kernel(const TnBwdArgs p) {
    using K = TnK<CT>;
    constexpr int NTL = K::NTL, ROWB = K::ROWB;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int N = p.N, T = p.T;
    const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
    const int clip = xcd + 8 * (qb / T), k = qb - (qb / T) * T;
    const int wg = (xcd + 8 * (qb / T)) * T + k;                       // row of the parameter-gradient partials (also for idle workgroups)
    const int act_bytes = ((N + 1) * ROWB + 1023) & ~1023;
    char* act = smem;
    char* ring = smem + act_bytes;
    float* gsum = reinterpret_cast<float*>(ring + 2 * K::KBT * K::SLOTB);   // [2 * CT] dgamma | dbeta of this workgroup
    if (clip >= p.clips) {                                              // idle slot of the XCD-aligned grid: its partial row is zero
        for (int i = tid; i < 2 * CT; i += 512) p.dgb[(long)i * p.nwg + wg] = 0.f;
        return;
    }
    const long frame_rows = (long)(clip * T + k) * N;
    const int tk = p.tk, thalf = tk >> 1;
    // dU[k] = sum_t dz[k - (t - thalf)] . W1b[t]: taps whose source frame exists
    const int t_first = max(0, k - (T - 1) + thalf), t_last = min(tk - 1, k + thalf);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W1b), 0, CT * tk * CT * 2, 0x00020000);
    const int lrow = lane >> 2, lch = (lane & 3) ^ ((4 - ((lane >> 4) & 3)) & 3);
    const unsigned wv1 = ((unsigned)lrow * (unsigned)(tk * CT) + lch * 8) * 2u;
    const int cob = (lg ^ ((4 - ((li >> 2) & 3)) & 3)) * 16;
    for (int i = tid; i < 3 * CT; i += 512) gsum[i] = i < 2 * CT ? 0.f : p.lnw[i - 2 * CT];
    static_assert(16 * 2 * CT * 4 <= 2 * K::KBT * K::SLOTB, "the (wave, tile) partial rows fit in the weight ring");
    for (int i = tid; i < ROWB / 4; i += 512) reinterpret_cast<unsigned*>(act + N * ROWB)[i] = 0u;     // the zero slot
    f32x4 acc[2][NTL];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned ab[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min((wid + 8 * i) * 16 + li, N);
        ab[i] = (unsigned)(m * ROWB + ((lg ^ (((m >> 2) & 1) << 1)) << 4));
    }
    for (int t = t_first; t <= t_last; ++t) {
        const int g = t - t_first;
        tn_issue_tap<CT>(ring, g & 1, r1, wv1, tk * CT, t, wid);       // (one tap per interval: the weights arrive beside the frame)
        tn_tile_dma<CT>(act, p.dz_in + (frame_rows - (long)(t - thalf) * N) * CT, N, wid, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // frame + tap complete
        tn_mma_tap<CT>(acc, act, ring, g & 1, ab, li, cob);
        __syncthreads();                                               // frame no longer read
    }
    // ---- LayerNorm backward on the accumulators: lane (li, lg) holds row m = tile*16 + li, channels j*16 + 4 lg .. + 3 (24 of the 96);
    // the three other lanes with the same li hold the rest of the row
    // X[k] comes through the (now free) tile as a whole frame by LDS-DMA and is read from LDS in the accumulator layout (8-byte cells,
    // twice: row sums, then the result) - gathered straight from memory those reads cost 36 partial-line requests per tile per wave.
    // LN'(dU) replaces x in its cell (bf16); dp is added on the way out, where both are whole coalesced rows.
    const float* gam = gsum + 2 * CT;
    constexpr float invC = 1.f / (float)CT;
    tn_tile_dma<CT>(act, p.X + frame_rows * CT, N, wid, lane);
    float rmean[2], rrstd[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min((wid + 8 * i) * 16 + li, N - 1);
        rmean[i] = p.mean[frame_rows + m]; rrstd[i] = p.rstd[frame_rows + m];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int mrow = (wid + 8 * i) * 16 + li;
        const bool live = mrow < N;
        const int m = min(mrow, N);                                    // rows beyond the plane: the zero slot
        char* row = act + m * ROWB + (lg & 1) * 8;
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const bf16x4 xv = *reinterpret_cast<const bf16x4*>(row + tn_pchunk(m, 2 * j + (lg >> 1)) * 16);
            const float4 gv = *reinterpret_cast<const float4*>(gam + j * 16 + lg * 4);
            const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float du = live ? acc[i][j][r] : 0.f;
                acc[i][j][r] = du;
                const float t = g4[r] * du;
                a1 += t;
                a2 += t * (((float)xv[r] - rmean[i]) * rrstd[i]);
            }
        }
        a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
        a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
        const float s1 = a1 * invC, s2 = a2 * invC;
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            char* cell = row + tn_pchunk(m, 2 * j + (lg >> 1)) * 16;
            const bf16x4 xv = *reinterpret_cast<const bf16x4*>(cell);
            const float4 gv = *reinterpret_cast<const float4*>(gam + j * 16 + lg * 4);
            const float g4[4] = {gv.x, gv.y, gv.z, gv.w};
            float dgm[4], dbt[4], dx[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float du = acc[i][j][r];
                const float xh = ((float)xv[r] - rmean[i]) * rrstd[i];
                dgm[r] = du * xh; dbt[r] = du;
                dx[r] = rrstd[i] * (g4[r] * du - s1 - xh * s2);
            }
            if (live) {                                                // in place: this lane is the only reader and writer of the cell
                bf16x4 o = {(bf16_t)dx[0], (bf16_t)dx[1], (bf16_t)dx[2], (bf16_t)dx[3]};
                *reinterpret_cast<bf16x4*>(cell) = o;
            }
            // parameter gradients: sum over the 16 rows of the tile (the DPP row li = 0..15 of this lg group)
#pragma unroll
            for (int r = 0; r < 4; ++r) { dgm[r] = tn_row_sum16(dgm[r]); dbt[r] = tn_row_sum16(dbt[r]); }
            if (li == 0) {                                             // one partial row per (wave, tile) in the dead weight ring: plain stores
                float* prow = reinterpret_cast<float*>(ring) + (wid * 2 + i) * 2 * CT;
                *reinterpret_cast<float4*>(prow + j * 16 + lg * 4) = make_float4(dgm[0], dgm[1], dgm[2], dgm[3]);
                *reinterpret_cast<float4*>(prow + CT + j * 16 + lg * 4) = make_float4(dbt[0], dbt[1], dbt[2], dbt[3]);
            }
        }
    }
    __syncthreads();
    {   // dX = dp + LN'(dU): whole rows, 16 bytes per lane
        const bf16_t* dpf = p.dp + frame_rows * CT;
        bf16_t* dst = p.dX + frame_rows * CT;
        int slot = tid / K::CPR, c = tid - slot * K::CPR;
        constexpr int DS = 512 / K::CPR, DC = 512 - DS * K::CPR;
        for (int idx = tid; idx < N * K::CPR; idx += 512) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(act + slot * ROWB + tn_pchunk(slot, c) * 16);
            const bf16x8 d = *reinterpret_cast<const bf16x8*>(dpf + (long)idx * 8);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)((float)v[e] + (float)d[e]);
            __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(dst + (long)idx * 8));
            slot += DS; c += DC;
            if (c >= K::CPR) { c -= K::CPR; ++slot; }
        }
    }
    for (int i = tid; i < 2 * CT; i += 512) {                           // the 16 (wave, tile) rows in a fixed order: bit-repeatable
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) a += reinterpret_cast<const float*>(ring)[r * 2 * CT + i];
        p.dgb[(long)i * p.nwg + wg] = a;                                 // [2 Ct][workgroups]: a channel's partials are one contiguous row
    }
}

// dgamma[c] += sum over workgroups of partial[c][wg], dbeta likewise: one wave per channel, fixed order (no atomics, bit-repeatable)
__global__ __launch_bounds__(64) void tnet_dgb_reduce_kernel(const float* part, int nwg, int CT, float* dgamma, float* dbeta) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float* row = part + (long)c * nwg;
    float a = 0.f;
    for (int w = lane; w < nwg; w += 64) a += row[w];
    a = wave_sum(a, 64);
    if (lane == 0) { if (c < CT) dgamma[c] += a; else dbeta[c - CT] += a; }
}

// the same reduction for SEVERAL layers in one launch (the engine defers the per-layer reductions to the end of backward: every launch on the
// data-gradient chain costs the step ~10 us): blockIdx.y = layer, its partial table at part + y * stride, its destinations from the table
struct TnDgbMulti { float* dgamma[32]; float* dbeta[32]; };
__global__ __launch_bounds__(64) void tnet_dgb_reduce_multi_kernel(const float* part, long stride, int nwg, int CT, TnDgbMulti dst) {
    const int c = blockIdx.x, lane = threadIdx.x, layer = blockIdx.y;
    const float* row = part + (long)layer * stride + (long)c * nwg;
    float a = 0.f;
    for (int w = lane; w < nwg; w += 64) a += row[w];
    a = wave_sum(a, 64);
    if (lane == 0) { if (c < CT) dst.dgamma[layer][c] += a; else dst.dbeta[layer][c - CT] += a; }
}

template <int CT>
int launch_tnet_bwd(const TnBwdArgs& aa, float* dgamma, float* dbeta, int phase, hipStream_t s) {
    const int act_bytes = ((aa.N + 1) * CT * 2 + 1023) & ~1023;
    const size_t smem = (size_t)act_bytes + 2 * (CT / 32) * CT * 64 + 3 * CT * sizeof(float);
    if (smem > 160 * 1024) return DIST_ERR_ARG;
    static DistSmemOnce attr_s, attr_t;
    RUN_(dist_max_smem(attr_s, reinterpret_cast<const void*>(tnet_bwd_spatial_kernel<CT>), smem));
    RUN_(dist_max_smem(attr_t, reinterpret_cast<const void*>(tnet_bwd_temporal_kernel<CT>), smem));
    const int groups = (aa.clips + 7) / 8, nwg = groups * 8 * aa.T;
    TnBwdArgs a = aa;
    a.nwg = nwg;
    if (phase == 0 || phase == 1) hipLaunchKernelGGL(tnet_bwd_spatial_kernel<CT>, dim3((unsigned)nwg), dim3(512), smem, s, a);
    if (phase != 1) {
        hipLaunchKernelGGL(tnet_bwd_temporal_kernel<CT>, dim3((unsigned)nwg), dim3(512), smem, s, a);
        if (phase != 3) hipLaunchKernelGGL(tnet_dgb_reduce_kernel, dim3((unsigned)(2 * CT)), dim3(64), 0, s, a.dgb, nwg, CT, dgamma, dbeta);
    }
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

}  // namespace

bool dist_k_tnet_fwd_eligible(int dtype, int Ct, int G, int tk) {
    static const bool on = (dist_knob("DIST_AMD_TNET_FUSED", 1) != 0);   // measurement knob: 0 = LayerNorm + two GEMMs
    return on && dtype == DIST_BF16 && (Ct == 32 || Ct == 64 || Ct == 96) && G >= 1 && G * G <= 256 && tk % 2 == 1 && tk <= 5;
}

extern "C" int dist_op_temporal_net_fwd(const dist_tnet_args* a, void* stream) {
    if (!a || !a->X || !a->W1 || !a->W2 || !a->b1 || !a->b2 || !a->ln_w || !a->ln_b || !a->z || !a->p || !a->Xp || !a->mean || !a->rstd) return DIST_ERR_ARG;
    if ((a->U == nullptr) != (a->V == nullptr)) return DIST_ERR_ARG;                  // both or neither
    if (a->clips <= 0 || a->T <= 0 || a->G <= 0) return DIST_ERR_ARG;
    if (!dist_k_tnet_fwd_eligible(a->dtype, a->Ct, a->G, a->tk)) return DIST_ERR_ARG;
    if ((long)a->clips * a->T * a->G * a->G * a->Ct >= (1l << 30)) return DIST_ERR_ARG;
    TnArgs k;
    k.X = static_cast<const bf16_t*>(a->X); k.W1 = static_cast<const bf16_t*>(a->W1); k.W2 = static_cast<const bf16_t*>(a->W2);
    k.b1 = a->b1; k.b2 = a->b2; k.lnw = a->ln_w; k.lnb = a->ln_b;
    k.z = static_cast<bf16_t*>(a->z); k.p = static_cast<bf16_t*>(a->p); k.Xp = static_cast<bf16_t*>(a->Xp);
    k.U = static_cast<bf16_t*>(a->U); k.V = static_cast<bf16_t*>(a->V);
    k.mean = a->mean; k.rstd = a->rstd;
    static const int dbg = dist_measure_knob("DIST_AMD_TNET_DBG", 0);
    static const int stagger = DIST_AB_KNOB("DIST_AMD_TNET_STAGGER", 0);     // A/B (profiles/r03_tnet_fused.md: flat to +10 %)
    k.dbg = dbg; k.stagger = stagger;
    k.clips = a->clips; k.T = a->T; k.G = a->G; k.N = a->G * a->G; k.tk = a->tk; k.eps = a->eps;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (a->Ct) {
        case 32: return launch_tnet_fwd<32>(k, s);
        case 64: return launch_tnet_fwd<64>(k, s);
        case 96: return launch_tnet_fwd<96>(k, s);
        default: return DIST_ERR_ARG;
    }
}

// rows of fp32 scratch dist_op_temporal_net_bwd needs for its parameter-gradient partials
extern "C" int64_t dist_op_temporal_net_bwd_scratch(int clips, int T, int Ct) { return (int64_t)((clips + 7) / 8) * 8 * T * 2 * Ct; }

extern "C" int dist_op_temporal_net_bwd(const dist_tnet_bwd_args* a, void* stream) {
    if (!a || !a->dp || !a->z || !a->X || !a->mean || !a->rstd || !a->ln_w || !a->W1b || !a->W2b || !a->dz || !a->dX || !a->dgamma || !a->dbeta || !a->scratch)
        return DIST_ERR_ARG;
    if (a->clips <= 0 || a->T <= 0 || a->G <= 0 || a->phase < 0 || a->phase > 3) return DIST_ERR_ARG;
    if (!dist_k_tnet_fwd_eligible(a->dtype, a->Ct, a->G, a->tk)) return DIST_ERR_ARG;
    if ((long)a->clips * a->T * a->G * a->G * a->Ct >= (1l << 30)) return DIST_ERR_ARG;
    if (a->scratch_elems < dist_op_temporal_net_bwd_scratch(a->clips, a->T, a->Ct)) return DIST_ERR_WORKSPACE;
    TnBwdArgs k;
    k.dp = static_cast<const bf16_t*>(a->dp); k.z = static_cast<const bf16_t*>(a->z); k.X = static_cast<const bf16_t*>(a->X);
    k.W1b = static_cast<const bf16_t*>(a->W1b); k.W2b = static_cast<const bf16_t*>(a->W2b);
    k.mean = a->mean; k.rstd = a->rstd; k.lnw = a->ln_w;
    k.dz = static_cast<bf16_t*>(a->dz); k.dz_in = k.dz; k.dX = static_cast<bf16_t*>(a->dX); k.dgb = a->scratch;
    k.clips = a->clips; k.T = a->T; k.G = a->G; k.N = a->G * a->G; k.tk = a->tk;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (a->Ct) {
        case 32: return launch_tnet_bwd<32>(k, a->dgamma, a->dbeta, a->phase, s);
        case 64: return launch_tnet_bwd<64>(k, a->dgamma, a->dbeta, a->phase, s);
        case 96: return launch_tnet_bwd<96>(k, a->dgamma, a->dbeta, a->phase, s);
        default: return DIST_ERR_ARG;
    }
}

extern "C" int dist_op_temporal_net_bwd_reduce(const float* scratch, int64_t layer_stride, int layers, int clips, int T, int Ct,
                                               float* const* dgamma, float* const* dbeta, void* stream) {
    if (!scratch || !dgamma || !dbeta || layers <= 0 || layers > 32 || clips <= 0 || T <= 0 || Ct <= 0) return DIST_ERR_ARG;
    if (layer_stride < dist_op_temporal_net_bwd_scratch(clips, T, Ct)) return DIST_ERR_ARG;
    TnDgbMulti d;
    for (int i = 0; i < 32; ++i) { d.dgamma[i] = i < layers ? dgamma[i] : nullptr; d.dbeta[i] = i < layers ? dbeta[i] : nullptr; }
    for (int i = 0; i < layers; ++i) if (!d.dgamma[i] || !d.dbeta[i]) return DIST_ERR_ARG;
    const int nwg = (clips + 7) / 8 * 8 * T;
    hipLaunchKernelGGL(tnet_dgb_reduce_multi_kernel, dim3((unsigned)(2 * Ct), (unsigned)layers), dim3(64), 0, static_cast<hipStream_t>(stream),
                       scratch, (long)layer_stride, nwg, Ct, d);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
