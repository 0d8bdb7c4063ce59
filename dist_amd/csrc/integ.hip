// Fused IntegrationNetwork forward (reference models/module_zoo/branches/dist.py:16-45):
//
//     R = ffn.c_proj( g( ffn.c_fc( ln(M') ) ) )  +  temporal_ffn.c_proj( g( conv_{3x1x1}( temporal_ffn.c_fc1( ln_temporal(M') ) ) ) )
//
// on the token rows M'[(clip*t + frame)*L + token][Ci].  ONE launch per layer instead of LayerNorm (two outputs) + four GEMMs: the
// normalised rows, the hidden activations hf / g2 and the 96-wide temporal intermediate h1 are read back from LDS, not from HBM
// (the unfused sequence reads 174 MB per layer at the bench size, this kernel 39 MB; what backward needs is still written).
//
// Work decomposition: one workgroup (8 waves) per (clip, group of TOK = BM / t tokens): its BM rows are ALL t frames of those tokens, so
// the temporal taps of temporal_ffn.c_fc2 (rows +-L apart in memory) are rows +-TOK of the same tile.
//
//   stage -1 (template T2I; dist.py:68-86) M' = M + [cls_token ; X' Wt^T + bt] formed here instead of read: the X' rows of the tile's tokens in regions B / C,
//            M' on the accumulators, LayerNorm statistics from row sums across lanes and waves, xhat straight into region A - replaces stage 0
//   stage 0  rows -> registers (8 lanes per row) -> statistics -> xhat = (x - mean) rstd as bf16 in LDS (region A, [BM][Ci]);
//            training: xhat itself (MODE 2: the weight gradients are taken against it and unfolded, dist_op_integration_unfold) or the two affine
//            outputs Na / Nb (MODE 1), and mean / rstd go to HBM
//   stage 1  [zf | h1] = xhat [Wa diag(ga) ; Wb diag(gb)]^T + [ba + Wa beta_a ; bb + Wb beta_b]      (LayerNorm folded into the weights,
//            dist_op_integration_pack) - N = Ci + C4 columns split over the waves by PAIRS of 16-column blocks
//            hf = g(zf) replaces xhat in region A, h1 goes to region B
//   stage 2  h2 = sum_d h1[row + d TOK] W2[d]^T + b2 (frames outside the clip contribute nothing), g2 = g(h2) -> region C   (6 waves)
//   stage 3  R = [hf | g2] [Wp | Wt]^T + bp + bt
//
// MFMA orientation: the WEIGHTS are the A operand (16 output columns x 32 k) and the activations the B operand (32 k x 16 tile rows),
// so a lane of the result holds four consecutive output COLUMNS of one tile row.  Each pair of MFMAs works on a permuted set of 32
// weight rows (n = 32 p + 8 (m / 4) + 4 q + m % 4 for row m of MFMA q) so that the two results of a lane are 8 consecutive columns:
// one 16-byte store to HBM / LDS per (pair, 16-row block), no transposition through LDS.  The weights come in exactly this
// fragment order (1 KB per MFMA operand, dist_op_integration_pack) straight from L2 into registers - no LDS, no barrier in the K loops.
//
// LDS: region A rows of Ci bf16, 16-byte chunk c of row r at chunk c ^ (r & 15) (768-byte rows alias to the same banks: the XOR makes
// the 16 lanes of every ds_read_b128 service group hit 16 different chunks); regions B / C rows of C4 bf16 with the chunk map of
// tnet.hip.  BM = 128: 154 KB, one workgroup per CU; BM = 64: 78 KB, two.
//
// Rounding points: xhat, zf, hf = g(bf16 zf), h1, h2, g2 = g(bf16 h2), R are rounded to bf16 (the unfused sequence rounds Na / Nb instead
// of xhat and multiplies by the unfolded weights: same precision, different rounding - see tests/test_integ_gpu.py for the measured gap).
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

#include "integ_common.h"

namespace {

template <int CI, int C4, int BM, int MODE, int DBG, bool T2I>
__global__ __launch_bounds__(512, (BM == 128 ? 2 : 4)) void integ_fwd_kernel(const IgArgs p) {
    constexpr bool TRAIN = MODE != 0;            // MODE 1: the two affine LayerNorm outputs Na / Nb are written, MODE 2: xhat itself (one tensor)
    constexpr int CC = CI + C4;
    constexpr int CPR = CI / 8;                  // 16-byte chunks per region-A row
    constexpr int CPB = C4 / 8;                  // ... per region-B / C row
    constexpr int LCH = CPR / 8;                 // chunks per lane in stage 0 (8 lanes per row)
    constexpr int RPW = BM / 8;                  // rows per wave in stage 0
    constexpr int NPASS = RPW / 8;
    constexpr int RB = BM / 16, RBH = RB / 2;    // 16-row blocks (the N dimension of the MFMAs)
    constexpr int KS1 = CI / 32, NP1 = CC / 32;
    constexpr int KT = C4 / 32, KS2 = 3 * KT, NP2 = C4 / 32;
    constexpr int KS3 = CC / 32, NP3 = CI / 32;
    static_assert(CPR % 16 == 0 && C4 % 32 == 0 && NP1 <= 16 && NP2 * 2 <= 8 && NP3 == 12 && (BM == 64 || BM == 128), "geometry");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const regA = smem;                               // [BM][CI]  xhat, then hf
    char* const regB = smem + BM * CI * 2;                 // [BM][C4]  h1
    char* const regC = regB + BM * C4 * 2;                 // [BM][C4]  g2

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int clip = blockIdx.x / p.groups, grp = blockIdx.x - clip * p.groups;
    const int tsh = p.tokshift, TOK = 1 << tsh, tokmask = TOK - 1;
    const int L = p.L, t = p.t;
    // tile row r = frame * TOK + token slot; global row of it, or -1 for a token slot beyond L (last group of a clip)
    auto grow_of = [&](const int r) -> int {
        const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
        return j < L ? (clip * t + f) * L + j : -1;
    };

    if constexpr (T2I) {
        // ---------------- stage -1 (T2I, dist.py:68-86): M' = M + [cls_token_f ; X'(frames 2f, 2f+1; position j-1) Wt^T + bt], then the LayerNorm
        // statistics of M' across the waves.  X' rows of the two frames go to regions B (a = 0) and C (a = 1): K = 2 C4, six k-steps.
        static_assert(!T2I || (BM == 128 && MODE != 1), "T2I in front: 128-row tiles, xhat form");
        constexpr int KST = 2 * C4 / 32;
        float* const red = reinterpret_cast<float*>(regC + BM * C4 * 2);   // [8][BM][2] partial row sums, then [BM][2] (mean, rstd)
        float* const tot = red + 8 * BM * 2;
        const int N = L - 1, T = 2 * t;
        {
            const int lq4 = lane & 3, lrow = lane >> 2;
            const int r = wid * 16 + lrow, f = r >> tsh, j = (grp << tsh) + (r & tokmask);
            const bool ok = j >= 1 && j < L;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bf16_t* src = p.Xp + ((long)(clip * T + 2 * f + a) * N + (ok ? j - 1 : 0)) * C4;
                char* dst = (a ? regC : regB) + r * (C4 * 2);
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    const int c = lq4 + 4 * m;
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c * 8);
                    if (!ok) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    *reinterpret_cast<bf16x8*>(dst + (ig_pchunk(r, c) << 4)) = v;
                }
            }
        }
        const int m3 = wid >> 1, odd = wid & 1;
        const int pF = 3 * m3 + (odd ? 2 : 0), pH = 3 * m3 + 1;
        // M at this lane's (row, 8 columns) positions (rows beyond L are clamped: finite numbers, never stored)
        // ... straight into region A, row-major: the A operand of the I2T product below, and where each lane finds its pieces again (M, then M', then xhat
        // share one slot per (row, pair): 48 registers less across the products and barriers - the kernel spilled with them live)
        auto slotF = [&](const int r) __attribute__((always_inline)) { return reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pF * 4 + lg) ^ li) << 4)); };
        auto slotH = [&](const int r) __attribute__((always_inline)) { return reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pH * 4 + lg) ^ li) << 4)); };
        int gro[RB];
        {
            bf16x8 mF[RB], mH[RBH];
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                const int r = (((rr + odd * RBH) & (RB - 1)) * 16) + li;
                const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
                gro[rr] = j < L ? (clip * t + f) * L + j : -1;
                const long base = (long)((clip * t + f) * L + min(j, L - 1)) * CI;
                mF[rr] = *reinterpret_cast<const bf16x8*>(p.M + base + pF * 32 + lg * 8);
                if (rr < RBH) mH[rr] = *reinterpret_cast<const bf16x8*>(p.M + base + pH * 32 + lg * 8);
            }
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                const int r = (((rr + odd * RBH) & (RB - 1)) * 16) + li;
                *slotF(r) = mF[rr];
                if (rr < RBH) *slotH(r) = mH[rr];
            }
        }
        const bool i2t = p.Xn != nullptr;                  // (kernel argument: the same for every workgroup)
        __syncthreads();
        if (i2t && wid < NP2 * 2) {
            // ---------------- I2T (dist.py:90-105): Linear on the patch rows of M (C4 columns: item = (column pair, half of the row blocks) as in stage 2), the
            // result added to BOTH frames 2f, 2f+1 of X' (nearest upsampling in time) -> X of the next layer.  Reads region A (M) and regions B / C (X').
            const int pi = wid % NP2, hi = wid / NP2;
            f32x4 ai[2][RBH];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int rb = 0; rb < RBH; ++rb) ai[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 wi[2][2];
            wi[0][0] = IG_LDW(p.Wi, ((long)pi * KS1) * 2); wi[0][1] = IG_LDW(p.Wi, ((long)pi * KS1) * 2 + 1);
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                if (ks + 1 < KS1) { wi[(ks + 1) & 1][0] = IG_LDW(p.Wi, ((long)pi * KS1 + ks + 1) * 2); wi[(ks + 1) & 1][1] = IG_LDW(p.Wi, ((long)pi * KS1 + ks + 1) * 2 + 1); }
#pragma unroll
                for (int rbh = 0; rbh < RBH; ++rbh) {
                    const int r = (hi * RBH + rbh) * 16 + li;
                    const bf16x8 a = IG_LDS(regA + r * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
                    ai[0][rbh] = IG_MMA(wi[ks & 1][0], a, ai[0][rbh]);
                    ai[1][rbh] = IG_MMA(wi[ks & 1][1], a, ai[1][rbh]);
                }
            }
            const int n0 = pi * 32 + lg * 8;
            float bv[8];
            ig_load8(p.bi + n0, bv);
#pragma unroll
            for (int rbh = 0; rbh < RBH; ++rbh) {
                const int r = (hi * RBH + rbh) * 16 + li;
                const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
                if (j < 1 || j >= L) continue;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = ai[0][rbh][e] + bv[e]; v[4 + e] = ai[1][rbh][e] + bv[4 + e]; }     // (one rounding, of the sum with X')
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const bf16x8 xp = *reinterpret_cast<const bf16x8*>((a ? regC : regB) + r * (C4 * 2) + (ig_pchunk(r, pi * 4 + lg) << 4));
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(v[e] + (float)xp[e]);
                    IG_ST(o, p.Xn + ((long)(clip * T + 2 * f + a) * N + (j - 1)) * C4 + n0);
                }
            }
        }
        f32x4 aF[2][RB], aH[2][RBH];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) aF[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < RBH; ++rb) aH[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        {
            bf16x8 wb[2][4];
            auto ldw = [&](bf16x8 (&w)[4], const int ks) __attribute__((always_inline)) {
                w[0] = IG_LDW(p.Wt, ((long)pF * KST + ks) * 2); w[1] = IG_LDW(p.Wt, ((long)pF * KST + ks) * 2 + 1);
                w[2] = IG_LDW(p.Wt, ((long)pH * KST + ks) * 2); w[3] = IG_LDW(p.Wt, ((long)pH * KST + ks) * 2 + 1);
            };
            ldw(wb[0], 0);
            ldw(wb[1], 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                bf16x8 (&w)[4] = wb[ks & 1];
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    const int r = (((rr + odd * RBH) & (RB - 1)) * 16) + li;
                    const bf16x8 a = IG_LDS((ks < KST / 2 ? regB : regC) + r * (C4 * 2) + (ig_pchunk(r, (ks % (KST / 2)) * 4 + lg) << 4));
                    aF[0][rr] = IG_MMA(w[0], a, aF[0][rr]);
                    aF[1][rr] = IG_MMA(w[1], a, aF[1][rr]);
                    if (rr < RBH) { aH[0][rr] = IG_MMA(w[2], a, aH[0][rr]); aH[1][rr] = IG_MMA(w[3], a, aH[1][rr]); }
                }
                if (ks + 2 < KST) ldw(w, ks + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                   // every wave has read M for the I2T product: the slots may change
        float bF[8], bH[8];
        ig_load8(p.bt + pF * 32 + lg * 8, bF);
        ig_load8(p.bt + pH * 32 + lg * 8, bH);
        // M' (rounded to bf16 as the unfused GEMM stores it) replaces M in the registers; partial row sums of M' and M'^2
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int r = (((rr + odd * RBH) & (RB - 1)) * 16) + li;
            const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
            const bool is_cls = j == 0;
            float s1 = 0.f, s2 = 0.f;
            {
                float cl[8];
                if (is_cls) ig_load8(p.cls + (long)f * CI + pF * 32 + lg * 8, cl);
                bf16x8 mv = *slotF(r);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float add = is_cls ? cl[e] : (e < 4 ? aF[0][rr][e] : aF[1][rr][e - 4]) + bF[e];
                    const bf16_t v = (bf16_t)((float)mv[e] + add);
                    mv[e] = v;
                    s1 += (float)v; s2 += (float)v * (float)v;
                }
                *slotF(r) = mv;
            }
            if (rr < RBH) {
                float cl[8];
                if (is_cls) ig_load8(p.cls + (long)f * CI + pH * 32 + lg * 8, cl);
                bf16x8 mv = *slotH(r);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float add = is_cls ? cl[e] : (e < 4 ? aH[0][rr][e] : aH[1][rr][e - 4]) + bH[e];
                    const bf16_t v = (bf16_t)((float)mv[e] + add);
                    mv[e] = v;
                    s1 += (float)v; s2 += (float)v * (float)v;
                }
                *slotH(r) = mv;
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lg == 0) reinterpret_cast<float2*>(red)[wid * BM + r] = make_float2(s1, s2);
        }
        __syncthreads();
        if (tid < BM) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) { const float2 v = reinterpret_cast<const float2*>(red)[w8 * BM + tid]; a1 += v.x; a2 += v.y; }
            constexpr float invC = 1.f / (float)CI;
            const float mu = a1 * invC;
            const float rs = rsqrtf(fmaxf(a2 * invC - mu * mu, 0.f) + p.eps);
            reinterpret_cast<float2*>(tot)[tid] = make_float2(mu, rs);
            const int gr = grow_of(tid);
            if (TRAIN && gr >= 0) { p.mean[gr] = mu; p.rstd[gr] = rs; }
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int r = (((rr + odd * RBH) & (RB - 1)) * 16) + li;
            const float2 st = reinterpret_cast<const float2*>(tot)[r];
            bf16x8 o;
            const bf16x8 mpF = *slotF(r);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(((float)mpF[e] - st.x) * st.y);
            *slotF(r) = o;
            if (gro[rr] >= 0) {
                if (MODE == 2) IG_ST(o, p.Xh + (long)gro[rr] * CI + pF * 32 + lg * 8);
                if (p.Mpo) IG_ST(mpF, p.Mpo + (long)gro[rr] * CI + pF * 32 + lg * 8);
            }
            if (rr < RBH) {
                const bf16x8 mpH = *slotH(r);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(((float)mpH[e] - st.x) * st.y);
                *slotH(r) = o;
                if (gro[rr] >= 0) {
                    if (MODE == 2) IG_ST(o, p.Xh + (long)gro[rr] * CI + pH * 32 + lg * 8);
                    if (p.Mpo) IG_ST(mpH, p.Mpo + (long)gro[rr] * CI + pH * 32 + lg * 8);
                }
            }
        }
    } else
    // ---------------- stage 0: rows -> LayerNorm statistics -> xhat tile (+ Na, Nb, mean, rstd)
    {
        const int lq = lane & 7, lrow = lane >> 3;
        bf16x8 raw[NPASS][LCH];
        int grow0[NPASS];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = wid * RPW + ps * 8 + lrow;
            const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
            grow0[ps] = j < L ? (clip * t + f) * L + j : -1;
            const bf16_t* src = p.Mp + (long)((clip * t + f) * L + min(j, L - 1)) * CI;       // (clamped: a slot beyond L computes on real numbers)
#pragma unroll
            for (int m = 0; m < LCH; ++m) raw[ps][m] = (DBG & 32) ? bf16x8{(bf16_t)(float)(lane + m), 1, 0, 2, 0, 0, 0, 0} : *reinterpret_cast<const bf16x8*>(src + (lq + 8 * m) * 8);
        }
        float mean[NPASS], rstd[NPASS];
        constexpr float invC = 1.f / (float)CI;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < LCH; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)raw[ps][m][e];
            mean[ps] = ig_sum8(s) * invC;
            float q = 0.f;
#pragma unroll
            for (int m = 0; m < LCH; ++m)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)raw[ps][m][e] - mean[ps]; q += d * d; }
            rstd[ps] = rsqrtf(ig_sum8(q) * invC + p.eps);
            if (TRAIN && lq == 0 && grow0[ps] >= 0) { p.mean[grow0[ps]] = mean[ps]; p.rstd[grow0[ps]] = rstd[ps]; }
        }
#pragma unroll
        for (int m = 0; m < LCH; ++m) {
            const int c = lq + 8 * m;
            float wa[8], ba[8], wb[8], bb[8];
            if (MODE == 1) { ig_load8(p.ga + c * 8, wa); ig_load8(p.ba + c * 8, ba); ig_load8(p.gb + c * 8, wb); ig_load8(p.bb + c * 8, bb); }
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int r = wid * RPW + ps * 8 + lrow;
                float xh[8];
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) { xh[e] = ((float)raw[ps][m][e] - mean[ps]) * rstd[ps]; o[e] = (bf16_t)xh[e]; }
                *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + ((c ^ (r & 15)) << 4)) = o;
                if (MODE == 2 && grow0[ps] >= 0) IG_ST(o, p.Xh + (long)grow0[ps] * CI + c * 8);
                if (MODE == 1 && grow0[ps] >= 0) {
                    bf16x8 na, nb;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { na[e] = (bf16_t)(xh[e] * wa[e] + ba[e]); nb[e] = (bf16_t)(xh[e] * wb[e] + bb[e]); }
                    IG_ST(na, p.Na + (long)grow0[ps] * CI + c * 8);
                    IG_ST(nb, p.Nb + (long)grow0[ps] * CI + c * 8);
                }
            }
        }
    }
    int grow[RB];                                          // global row of tile row rb * 16 + li
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) grow[rb] = grow_of(rb * 16 + li);
    __syncthreads();

    // ---------------- stage 1: [zf | h1] = xhat W1^T, pairs 2 wid and 2 wid + 1 (the last wave has one pair when NP1 is odd)
    {
        // (NP1 odd: the last wave's second pair repeats its first one and is dropped in the epilogue - no branch inside the K loop)
        const int pA = 2 * wid, pB = min(2 * wid + 1, NP1 - 1);
        f32x4 acc[4][RB];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[f][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 wb[2][4];
        auto ldw = [&](bf16x8 (&w)[4], const int ks) __attribute__((always_inline)) {
            w[0] = IG_LDW(p.W1, ((long)pA * KS1 + ks) * 2); w[1] = IG_LDW(p.W1, ((long)pA * KS1 + ks) * 2 + 1);
            w[2] = IG_LDW(p.W1, ((long)pB * KS1 + ks) * 2); w[3] = IG_LDW(p.W1, ((long)pB * KS1 + ks) * 2 + 1);
        };
        ldw(wb[0], 0);
        if (KS1 > 1) ldw(wb[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            bf16x8 (&w)[4] = wb[ks & 1];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bf16x8 a = IG_LDS(regA + (rb * 16 + li) * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[f][rb] = IG_MMA(w[f], a, acc[f][rb]);
            }
            if (ks + 2 < KS1) ldw(w, ks + 2);
            __builtin_amdgcn_sched_barrier(0);             // (the loads stay HERE, a whole k-step ahead of their use: the scheduler would sink them to it)
        }
        __syncthreads();                                   // every wave has read xhat: region A becomes hf
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int pr = pp ? 2 * wid + 1 : pA;
            if (pr >= NP1) continue;
            const int n0 = pr * 32 + lg * 8;
            float bv[8];
            ig_load8(p.b1 + n0, bv);
            const bool is_zf = pr < NP3;                   // (Ci / 32 pairs of zf, then the h1 pairs)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = rb * 16 + li;
                bf16x8 zb;
#pragma unroll
                for (int e = 0; e < 4; ++e) { zb[e] = (bf16_t)(acc[pp * 2][rb][e] + bv[e]); zb[4 + e] = (bf16_t)(acc[pp * 2 + 1][rb][e] + bv[4 + e]); }
                if (is_zf) {
                    bf16x8 hf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hf[e] = (bf16_t)IG_GELU((float)zb[e]);
                    *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pr * 4 + lg) ^ li) << 4)) = hf;
                    if (TRAIN && grow[rb] >= 0) {
                        IG_ST(zb, p.zfh2 + (long)grow[rb] * CC + n0);
                        IG_ST(hf, p.hfg2 + (long)grow[rb] * CC + n0);
                    }
                } else {
                    *reinterpret_cast<bf16x8*>(regB + r * (C4 * 2) + (ig_pchunk(r, (pr - NP3) * 4 + lg) << 4)) = zb;
                    if (TRAIN && grow[rb] >= 0) IG_ST(zb, p.h1 + (long)grow[rb] * C4 + (n0 - CI));
                }
            }
        }
    }
    // the temporal weights of this wave's stage-2 item travel while the workgroup meets at the barrier
    const bool s2 = wid < NP2 * 2;
    const int p2 = s2 ? wid % NP2 : 0, half2 = s2 ? wid / NP2 : 0;
    bf16x8 w2[KS2][2];
    if (s2) {
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) { w2[ks][0] = IG_LDW(p.W2, ((long)p2 * KS2 + ks) * 2); w2[ks][1] = IG_LDW(p.W2, ((long)p2 * KS2 + ks) * 2 + 1); }
    }
    __syncthreads();                                       // hf in region A, h1 in region B

    // ---------------- stage 2: h2 = conv_t(h1) + b2, g2 = g(h2) -> region C; item = (pair p2, half of the 16-row blocks)
    if (s2) {
        f32x4 a2[2][RBH];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int rb = 0; rb < RBH; ++rb) a2[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
            for (int rbh = 0; rbh < RBH; ++rbh) {
                const int r = (half2 * RBH + rbh) * 16 + li;
                const int f = (r >> tsh) + tap - 1;
                const bool ok = (unsigned)f < (unsigned)t;
                const int rs = ok ? r + (tap - 1) * TOK : r;
#pragma unroll
                for (int k3 = 0; k3 < KT; ++k3) {
                    bf16x8 a = IG_LDS(regB + rs * (C4 * 2) + (ig_pchunk(rs, k3 * 4 + lg) << 4));
                    if (!ok) a = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    a2[0][rbh] = IG_MMA(w2[tap * KT + k3][0], a, a2[0][rbh]);
                    a2[1][rbh] = IG_MMA(w2[tap * KT + k3][1], a, a2[1][rbh]);
                }
            }
        }
        const int n0 = p2 * 32 + lg * 8;
        float bv[8];
        ig_load8(p.b2 + n0, bv);
#pragma unroll
        for (int rbh = 0; rbh < RBH; ++rbh) {
            const int r = (half2 * RBH + rbh) * 16 + li;
            const int gr = grow_of(r);
            bf16x8 hb, gb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hb[e] = (bf16_t)(a2[0][rbh][e] + bv[e]); hb[4 + e] = (bf16_t)(a2[1][rbh][e] + bv[4 + e]); }
#pragma unroll
            for (int e = 0; e < 8; ++e) gb[e] = (bf16_t)IG_GELU((float)hb[e]);
            *reinterpret_cast<bf16x8*>(regC + r * (C4 * 2) + (ig_pchunk(r, p2 * 4 + lg) << 4)) = gb;
            if (TRAIN && gr >= 0) {
                IG_ST(hb, p.zfh2 + (long)gr * CC + CI + n0);
                IG_ST(gb, p.hfg2 + (long)gr * CC + CI + n0);
            }
        }
    }

    // ---------------- stage 3: R = [hf | g2] W3^T + b3.  Waves 2m, 2m+1 share pairs 3m .. 3m+2: one full pair each and half the
    // 16-row blocks of the middle one.  The blocks are walked from this wave's half on, so the first RBH of them are the shared pair's.
    {
        const int m3 = wid >> 1, odd = wid & 1;
        const int pF = 3 * m3 + (odd ? 2 : 0), pH = 3 * m3 + 1;
        f32x4 aF[2][RB], aH[2][RBH];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) aF[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < RBH; ++rb) aH[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        bf16x8 wb[2][4];
        auto ldw = [&](bf16x8 (&w)[4], const int ks) __attribute__((always_inline)) {
            w[0] = IG_LDW(p.W3, ((long)pF * KS3 + ks) * 2); w[1] = IG_LDW(p.W3, ((long)pF * KS3 + ks) * 2 + 1);
            w[2] = IG_LDW(p.W3, ((long)pH * KS3 + ks) * 2); w[3] = IG_LDW(p.W3, ((long)pH * KS3 + ks) * 2 + 1);
        };
        ldw(wb[0], 0);
        ldw(wb[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) {
            if (ks == KS1) __syncthreads();                // g2 is complete in region C (the stage-2 waves pass here after writing it)
            bf16x8 (&w)[4] = wb[ks & 1];
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                const int rb = (rr + odd * RBH) & (RB - 1);
                const int r = rb * 16 + li;
                bf16x8 a;
                if (ks < KS1) a = IG_LDS(regA + r * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
                else a = IG_LDS(regC + r * (C4 * 2) + (ig_pchunk(r, (ks - KS1) * 4 + lg) << 4));
                aF[0][rr] = IG_MMA(w[0], a, aF[0][rr]);
                aF[1][rr] = IG_MMA(w[1], a, aF[1][rr]);
                if (rr < RBH) { aH[0][rr] = IG_MMA(w[2], a, aH[0][rr]); aH[1][rr] = IG_MMA(w[3], a, aH[1][rr]); }
            }
            if (ks + 2 < KS3) ldw(w, ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        float bF[8], bH[8];
        ig_load8(p.b3 + pF * 32 + lg * 8, bF);
        ig_load8(p.b3 + pH * 32 + lg * 8, bH);
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int rb = (rr + odd * RBH) & (RB - 1);
            const int gr = grow_of(rb * 16 + li);
            if (gr < 0) continue;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)(aF[0][rr][e] + bF[e]); o[4 + e] = (bf16_t)(aF[1][rr][e] + bF[4 + e]); }
            IG_ST(o, p.R + (long)gr * CI + pF * 32 + lg * 8);
            if (rr < RBH) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)(aH[0][rr][e] + bH[e]); o[4 + e] = (bf16_t)(aH[1][rr][e] + bH[4 + e]); }
                IG_ST(o, p.R + (long)gr * CI + pH * 32 + lg * 8);
            }
        }
    }
}

// ---- fused data-gradient backward -----------------------------------------------------------------------------------------------------
// dM' = LN'( dzf Wa' + dh1 Wb' ),  dh1 = conv_t^T(dh2),  [dzf | dh2] = (dR [Wp | Wt]) * g'([zf | h2])          (dist.py:40-45 through autograd)
// The same three GEMM shapes as the forward kernel on the same tile (t frames x TOK tokens), with the transposed weights (B1 / B2 / B3 of
// dist_op_integration_pack): stage 1 N = Ci + C4 over dR, stage 2 the temporal taps with the opposite shift, stage 3 K = Ci + C4 over
// [dzf | dh1] with the LayerNorm weights folded in, so its accumulators ARE dxhat and the LayerNorm backward
//     dx = rstd (dxhat - mean_k dxhat - xhat mean_k(dxhat xhat))
// runs on them in registers: row sums over a lane's columns, over the 4 lanes of a row (ds_bpermute), over the 8 waves through LDS.
// Written: [dzf | dh2] and dh1 (the weight-gradient GEMMs read them), dM' (+ a second copy, + dR for the last layer).  The LayerNorm
// parameter gradients come from dist_op_integration_unfold, not from here.

template <int CI, int C4, int BM>
__global__ __launch_bounds__(512, (BM == 128 ? 2 : 4)) void integ_bwd_kernel(const IgBwdArgs p) {
#ifdef DIST_INTEG_BWD_DBG
    constexpr int DBG = DIST_INTEG_BWD_DBG;      // timing-only ablation (tools/integ_ablate.sh), same bits as the forward kernel
#else
    constexpr int DBG = 0;
#endif
    constexpr int CC = CI + C4;
    constexpr int CPR = CI / 8, LCH = CPR / 8, RPW = BM / 8, NPASS = RPW / 8;
    constexpr int RB = BM / 16, RBH = RB / 2;
    constexpr int KS1 = CI / 32, NP1 = CC / 32;
    constexpr int KT = C4 / 32, KS2 = 3 * KT, NP2 = C4 / 32;
    constexpr int KS3 = CC / 32, NP3 = CI / 32;
    static_assert(CPR % 16 == 0 && C4 % 32 == 0 && NP1 <= 16 && NP2 * 2 <= 8 && NP3 == 12 && (BM == 64 || BM == 128), "geometry");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const regA = smem;                               // [BM][CI]  dR, then dzf
    char* const regB = smem + BM * CI * 2;                 // [BM][C4]  dh2
    char* const regC = regB + BM * C4 * 2;                 // [BM][C4]  dh1
    float* const red = reinterpret_cast<float*>(regC + BM * C4 * 2);   // [8 waves][BM][2] partial row sums, then [BM][2] totals
    float* const tot = red + 8 * BM * 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int clip = blockIdx.x / p.groups, grp = blockIdx.x - clip * p.groups;
    const int tsh = p.tokshift, TOK = 1 << tsh, tokmask = TOK - 1;
    const int L = p.L, t = p.t;
    auto grow_of = [&](const int r) -> int {
        const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
        return j < L ? (clip * t + f) * L + j : -1;
    };

    // ---------------- stage 0: dR rows -> region A
    {
        const int lq = lane & 7, lrow = lane >> 3;
        bf16x8 raw[NPASS][LCH];
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = wid * RPW + ps * 8 + lrow;
            const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
            const bf16_t* src = p.dR + (long)((clip * t + f) * L + min(j, L - 1)) * CI;
#pragma unroll
            for (int m = 0; m < LCH; ++m) raw[ps][m] = *reinterpret_cast<const bf16x8*>(src + (lq + 8 * m) * 8);
        }
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int r = wid * RPW + ps * 8 + lrow;
#pragma unroll
            for (int m = 0; m < LCH; ++m) *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((lq + 8 * m) ^ (r & 15)) << 4)) = raw[ps][m];
        }
    }
    int grow[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) grow[rb] = grow_of(rb * 16 + li);
    __syncthreads();

    // ---------------- stage 1: [dzf | dh2] = (dR B1^T) * g'([zf | h2])
    {
        const int pA = 2 * wid, pB = min(2 * wid + 1, NP1 - 1);
        f32x4 acc[4][RB];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[f][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 wb[2][4];
        auto ldw = [&](bf16x8 (&w)[4], const int ks) __attribute__((always_inline)) {
            w[0] = IG_LDW(p.W1, ((long)pA * KS1 + ks) * 2); w[1] = IG_LDW(p.W1, ((long)pA * KS1 + ks) * 2 + 1);
            w[2] = IG_LDW(p.W1, ((long)pB * KS1 + ks) * 2); w[3] = IG_LDW(p.W1, ((long)pB * KS1 + ks) * 2 + 1);
        };
        ldw(wb[0], 0);
        ldw(wb[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            bf16x8 (&w)[4] = wb[ks & 1];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bf16x8 a = IG_LDS(regA + (rb * 16 + li) * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
#pragma unroll
                for (int f = 0; f < 4; ++f) acc[f][rb] = IG_MMA(w[f], a, acc[f][rb]);
            }
            if (ks + 2 < KS1) ldw(w, ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // every wave has read dR: region A becomes dzf
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int pr = pp ? 2 * wid + 1 : pA;
            if (pr >= NP1) continue;
            const int n0 = pr * 32 + lg * 8;
            bf16x8 zv[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                zv[rb] = grow[rb] >= 0 ? *reinterpret_cast<const bf16x8*>(p.zfh2 + (long)grow[rb] * CC + n0) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = rb * 16 + li;
                bf16x8 dv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dv[e] = (bf16_t)(acc[pp * 2][rb][e] * qgelu_grad_t<bf16_t>((float)zv[rb][e]));
                    dv[4 + e] = (bf16_t)(acc[pp * 2 + 1][rb][e] * qgelu_grad_t<bf16_t>((float)zv[rb][4 + e]));
                }
                if (pr < NP3) *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pr * 4 + lg) ^ li) << 4)) = dv;
                else *reinterpret_cast<bf16x8*>(regB + r * (C4 * 2) + (ig_pchunk(r, (pr - NP3) * 4 + lg) << 4)) = dv;
                if (grow[rb] >= 0) {
                    if (pr < NP3) IG_ST(dv, p.dzfh2 + (long)grow[rb] * p.ldz + n0);
                    else IG_ST(dv, p.dh2 + (long)grow[rb] * p.ld2 + (n0 - CI));
                }
            }
        }
    }
    const bool s2 = wid < NP2 * 2;
    const int p2 = s2 ? wid % NP2 : 0, half2 = s2 ? wid / NP2 : 0;
    bf16x8 w2[KS2][2];
    if (s2) {
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) { w2[ks][0] = IG_LDW(p.W2, ((long)p2 * KS2 + ks) * 2); w2[ks][1] = IG_LDW(p.W2, ((long)p2 * KS2 + ks) * 2 + 1); }
    }
    __syncthreads();                                       // dzf in region A, dh2 in region B

    // ---------------- stage 2: dh1[r] = sum_tap dh2[r - (tap - 1) TOK] W2[tap] -> region C
    if (s2) {
        f32x4 a2[2][RBH];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int rb = 0; rb < RBH; ++rb) a2[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
            for (int rbh = 0; rbh < RBH; ++rbh) {
                const int r = (half2 * RBH + rbh) * 16 + li;
                const int f = (r >> tsh) - (tap - 1);
                const bool ok = (unsigned)f < (unsigned)t;
                const int rs = ok ? r - (tap - 1) * TOK : r;
#pragma unroll
                for (int k3 = 0; k3 < KT; ++k3) {
                    bf16x8 a = IG_LDS(regB + rs * (C4 * 2) + (ig_pchunk(rs, k3 * 4 + lg) << 4));
                    if (!ok) a = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    a2[0][rbh] = IG_MMA(w2[tap * KT + k3][0], a, a2[0][rbh]);
                    a2[1][rbh] = IG_MMA(w2[tap * KT + k3][1], a, a2[1][rbh]);
                }
            }
        }
        const int n0 = p2 * 32 + lg * 8;
#pragma unroll
        for (int rbh = 0; rbh < RBH; ++rbh) {
            const int r = (half2 * RBH + rbh) * 16 + li;
            const int gr = grow_of(r);
            bf16x8 hb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hb[e] = (bf16_t)a2[0][rbh][e]; hb[4 + e] = (bf16_t)a2[1][rbh][e]; }
            *reinterpret_cast<bf16x8*>(regC + r * (C4 * 2) + (ig_pchunk(r, p2 * 4 + lg) << 4)) = hb;
            if (gr >= 0) IG_ST(hb, p.dh1 + (long)gr * p.ld1 + n0);
        }
    }

    // ---------------- stage 3: dxhat = [dzf | dh1] B3^T, then the LayerNorm backward on the accumulators
    {
        const int m3 = wid >> 1, odd = wid & 1;
        const int pF = 3 * m3 + (odd ? 2 : 0), pH = 3 * m3 + 1;
        f32x4 aF[2][RB], aH[2][RBH];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) aF[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int rb = 0; rb < RBH; ++rb) aH[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        bf16x8 wb[2][4];
        auto ldw = [&](bf16x8 (&w)[4], const int ks) __attribute__((always_inline)) {
            w[0] = IG_LDW(p.W3, ((long)pF * KS3 + ks) * 2); w[1] = IG_LDW(p.W3, ((long)pF * KS3 + ks) * 2 + 1);
            w[2] = IG_LDW(p.W3, ((long)pH * KS3 + ks) * 2); w[3] = IG_LDW(p.W3, ((long)pH * KS3 + ks) * 2 + 1);
        };
        ldw(wb[0], 0);
        ldw(wb[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) {
            if (ks == KS1) __syncthreads();                // dh1 is complete in region C
            bf16x8 (&w)[4] = wb[ks & 1];
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                const int rb = (rr + odd * RBH) & (RB - 1);
                const int r = rb * 16 + li;
                bf16x8 a;
                if (ks < KS1) a = IG_LDS(regA + r * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
                else a = IG_LDS(regC + r * (C4 * 2) + (ig_pchunk(r, (ks - KS1) * 4 + lg) << 4));
                aF[0][rr] = IG_MMA(w[0], a, aF[0][rr]);
                aF[1][rr] = IG_MMA(w[1], a, aF[1][rr]);
                if (rr < RBH) { aH[0][rr] = IG_MMA(w[2], a, aH[0][rr]); aH[1][rr] = IG_MMA(w[3], a, aH[1][rr]); }
            }
            if (ks + 2 < KS3) ldw(w, ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        const bool i2tb = p.dY != nullptr, t2ib = p.dp != nullptr;      // (kernel arguments: the same for every workgroup)
        if (i2tb) {
            // dY rows of this tile -> region B (free since the barrier inside the K loop: stage 2 was its last reader) and to memory; visible after the
            // two barriers of the LayerNorm backward below
            const int lq4 = lane & 3, lrow = lane >> 2;
            const int r = wid * 16 + lrow, f = r >> tsh, j = (grp << tsh) + (r & tokmask);
            const bool ok = j >= 1 && j < L;
            const int N = L - 1, T = 2 * t;
            const bf16_t* s0 = p.dXn + ((long)(clip * T + 2 * f) * N + (ok ? j - 1 : 0)) * C4;
            const bf16_t* s1 = s0 + (long)N * C4;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const int c = lq4 + 4 * m;
                const bf16x8 u = *reinterpret_cast<const bf16x8*>(s0 + c * 8), v = *reinterpret_cast<const bf16x8*>(s1 + c * 8);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = ok ? (bf16_t)((float)u[e] + (float)v[e]) : (bf16_t)0.f;
                *reinterpret_cast<bf16x8*>(regB + r * (C4 * 2) + (ig_pchunk(r, c) << 4)) = o;
                if (ok) IG_ST(o, p.dY + ((long)(clip * t + f) * N + (j - 1)) * C4 + c * 8);
            }
        }
        // xhat at this lane's (row, 8 columns) positions; partial row sums S1 = sum dxhat, S2 = sum dxhat xhat.  The xhat values are needed again behind
        // the two barriers of the row sums: they wait in region A (dead once every wave has left the K loop - hence the barrier here), each lane's in the
        // slot its dM' piece of the same (row, pair) will take, instead of in 48 registers (the kernel spilled 76 bytes per lane with them live)
        __syncthreads();
        int gro[RB];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            gro[rr] = grow_of((((rr + odd * RBH) & (RB - 1)) * 16) + li);
            float s1 = 0.f, s2v = 0.f;
            const long base = (long)max(gro[rr], 0) * CI;
            const int r = (((rr + odd * RBH) & (RB - 1)) * 16) + li;
            const bf16x8 xF1 = *reinterpret_cast<const bf16x8*>(p.Xh + base + pF * 32 + lg * 8);
            *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pF * 4 + lg) ^ li) << 4)) = xF1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1 += aF[0][rr][e] + aF[1][rr][e];
                s2v += aF[0][rr][e] * (float)xF1[e] + aF[1][rr][e] * (float)xF1[4 + e];
            }
            if (rr < RBH) {
                const bf16x8 xH1 = *reinterpret_cast<const bf16x8*>(p.Xh + base + pH * 32 + lg * 8);
                *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pH * 4 + lg) ^ li) << 4)) = xH1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1 += aH[0][rr][e] + aH[1][rr][e];
                    s2v += aH[0][rr][e] * (float)xH1[e] + aH[1][rr][e] * (float)xH1[4 + e];
                }
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2v += __shfl_xor(s2v, 16, 64); s2v += __shfl_xor(s2v, 32, 64);
            if (lg == 0) {
                reinterpret_cast<float2*>(red)[wid * BM + r] = make_float2(s1, s2v);
            }
        }
        __syncthreads();
        if (tid < BM) {
            float a1 = 0.f, a2s = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) { const float2 v = reinterpret_cast<const float2*>(red)[w8 * BM + tid]; a1 += v.x; a2s += v.y; }
            reinterpret_cast<float2*>(tot)[tid] = make_float2(a1, a2s);
        }
        __syncthreads();
        constexpr float invC = 1.f / (float)CI;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int gr = gro[rr];
            if (gr < 0) continue;
            const int r = (((rr + odd * RBH) & (RB - 1)) * 16) + li;
            const float2 cs = reinterpret_cast<const float2*>(tot)[r];
            const float rs = p.rstd[gr], m1 = cs.x * invC, m2 = cs.y * invC;
            bf16x8 o;
            float v[8];
            const bf16x8 xFr = *reinterpret_cast<const bf16x8*>(regA + r * (CI * 2) + (((pF * 4 + lg) ^ li) << 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = rs * (aF[0][rr][e] - m1 - (float)xFr[e] * m2);
                v[4 + e] = rs * (aF[1][rr][e] - m1 - (float)xFr[4 + e] * m2);
            }
            if (p.add_dR) {
                const bf16x8 d = *reinterpret_cast<const bf16x8*>(p.dR + (long)gr * CI + pF * 32 + lg * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)d[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
            IG_ST(o, p.dMp + (long)gr * CI + pF * 32 + lg * 8);
            const bool to_dm = p.dM && !i2tb && (!p.dm_cls || ((grp << tsh) + (r & tokmask)) == 0);
            if (to_dm) IG_ST(o, p.dM + (long)gr * CI + pF * 32 + lg * 8);
            if (t2ib) *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pF * 4 + lg) ^ li) << 4)) = o;     // dM' tile: the A operand of the T2I product at the end
            const bool cls_row = p.dcls && grp == 0 && (r & tokmask) == 0;
            if (cls_row) {
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(p.dcls + (long)(r >> tsh) * CI + pF * 32 + lg * 8 + e, (float)o[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { aF[0][rr][e] = (float)o[e]; aF[1][rr][e] = (float)o[4 + e]; }       // (the I2T term below accumulates on the stored dM')
            if (rr < RBH) {
                const bf16x8 xHr = *reinterpret_cast<const bf16x8*>(regA + r * (CI * 2) + (((pH * 4 + lg) ^ li) << 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = rs * (aH[0][rr][e] - m1 - (float)xHr[e] * m2);
                    v[4 + e] = rs * (aH[1][rr][e] - m1 - (float)xHr[4 + e] * m2);
                }
                if (p.add_dR) {
                    const bf16x8 d = *reinterpret_cast<const bf16x8*>(p.dR + (long)gr * CI + pH * 32 + lg * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)d[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                IG_ST(o, p.dMp + (long)gr * CI + pH * 32 + lg * 8);
                if (to_dm) IG_ST(o, p.dM + (long)gr * CI + pH * 32 + lg * 8);
                if (t2ib) *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pH * 4 + lg) ^ li) << 4)) = o;
                if (cls_row) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) atomicAdd(p.dcls + (long)(r >> tsh) * CI + pH * 32 + lg * 8 + e, (float)o[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { aH[0][rr][e] = (float)o[e]; aH[1][rr][e] = (float)o[4 + e]; }
            }
        }
        if (i2tb) {
            // dM = dM' + dY W4^T (K = C4): the accumulators hold dM' as stored; cls rows and slots beyond L have all-zero dY rows
            constexpr int KS4 = C4 / 32;
#pragma unroll
            for (int ks = 0; ks < KS4; ++ks) {
                const bf16x8 w0 = IG_LDW(p.W4, ((long)pF * KS4 + ks) * 2), w1 = IG_LDW(p.W4, ((long)pF * KS4 + ks) * 2 + 1);
                const bf16x8 w2 = IG_LDW(p.W4, ((long)pH * KS4 + ks) * 2), w3 = IG_LDW(p.W4, ((long)pH * KS4 + ks) * 2 + 1);
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    const int r = (((rr + odd * RBH) & (RB - 1)) * 16) + li;
                    const bf16x8 a = IG_LDS(regB + r * (C4 * 2) + (ig_pchunk(r, ks * 4 + lg) << 4));
                    aF[0][rr] = IG_MMA(w0, a, aF[0][rr]);
                    aF[1][rr] = IG_MMA(w1, a, aF[1][rr]);
                    if (rr < RBH) { aH[0][rr] = IG_MMA(w2, a, aH[0][rr]); aH[1][rr] = IG_MMA(w3, a, aH[1][rr]); }
                }
            }
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                const int gr = gro[rr];
                if (gr < 0) continue;
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)aF[0][rr][e]; o[4 + e] = (bf16_t)aF[1][rr][e]; }
                IG_ST(o, p.dM + (long)gr * CI + pF * 32 + lg * 8);
                if (rr < RBH) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)aH[0][rr][e]; o[4 + e] = (bf16_t)aH[1][rr][e]; }
                    IG_ST(o, p.dM + (long)gr * CI + pH * 32 + lg * 8);
                }
            }
        }
        if (t2ib) {
            // ---------------- T2I backward: dp = (dX_next + dM' W5^T) g'(p) on the two temporal frames of every patch row of the tile.  N = 2 C4 columns = 6 pairs,
            // 24 items (pair, quarter of the 16-row blocks), three per wave; A = the dM' tile in region A.
            __syncthreads();
            constexpr int NP5 = 2 * C4 / 32, RBQ = RB / 4;
            const int N = L - 1, T = 2 * t;
#pragma unroll 1
            for (int it = 0; it < 3; ++it) {
                const int id = wid * 3 + it, p5 = id >> 2, qr = id & 3;
                f32x4 a5[2][RBQ];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int rb = 0; rb < RBQ; ++rb) a5[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
                bf16x8 w5[2][2];
                w5[0][0] = IG_LDW(p.W5, ((long)p5 * KS1) * 2); w5[0][1] = IG_LDW(p.W5, ((long)p5 * KS1) * 2 + 1);
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    if (ks + 1 < KS1) { w5[(ks + 1) & 1][0] = IG_LDW(p.W5, ((long)p5 * KS1 + ks + 1) * 2); w5[(ks + 1) & 1][1] = IG_LDW(p.W5, ((long)p5 * KS1 + ks + 1) * 2 + 1); }
#pragma unroll
                    for (int rb = 0; rb < RBQ; ++rb) {
                        const int r = (qr * RBQ + rb) * 16 + li;
                        const bf16x8 a = IG_LDS(regA + r * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
                        a5[0][rb] = IG_MMA(w5[ks & 1][0], a, a5[0][rb]);
                        a5[1][rb] = IG_MMA(w5[ks & 1][1], a, a5[1][rb]);
                    }
                }
                const int fa = p5 / (NP5 / 2), c0 = (p5 % (NP5 / 2)) * 32 + lg * 8;       // temporal frame 2f + fa, columns c0 .. c0 + 7
#pragma unroll
                for (int rb = 0; rb < RBQ; ++rb) {
                    const int r = (qr * RBQ + rb) * 16 + li;
                    const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
                    if (j < 1 || j >= L) continue;
                    const long row = ((long)(clip * T + 2 * f + fa) * N + (j - 1)) * C4 + c0;
                    const bf16x8 pv = *reinterpret_cast<const bf16x8*>(p.pact + row);
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = a5[0][rb][e]; v[4 + e] = a5[1][rb][e]; }
                    if (p.dXn) {
                        const bf16x8 dx = *reinterpret_cast<const bf16x8*>(p.dXn + row);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)dx[e];
                    }
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(v[e] * qgelu_grad_t<bf16_t>((float)pv[e]));
                    IG_ST(o, p.dp + row);
                }
            }
        }
    }
}

// ---- weights into MFMA-operand order -----------------------------------------------------------------------------------------------
// matrix [N][K] (N, K multiples of 32) -> fragments ((p * K/32 + ks) * 2 + q) of 64 lanes x 8 bf16: lane l holds row
// n = 32 p + 8 ((l & 15) / 4) + 4 q + (l & 3), columns k = 32 ks + 8 (l / 16) .. + 7
struct IgPack {
    const float *Wa, *ba, *ga, *bta;         // ffn.c_fc [Ci][Ci], bias, ln weight / bias
    const float *Wb, *bb, *gb, *btb;         // temporal_ffn.c_fc1 [C4][Ci], bias, ln_temporal weight / bias
    const float *W2, *b2;                    // temporal_ffn.c_fc2 [C4][C4][3], bias
    const float *Wp, *bp, *Wt, *bt;          // ffn.c_proj [Ci][Ci], temporal_ffn.c_proj [Ci][C4], biases
    bf16_t *W1o, *W2o, *W3o; float *b1o, *b2o, *b3o;
    bf16_t *B1o, *B2o, *B3o;                 // optional: the data-gradient operands of integ_bwd_kernel (same shapes, transposed weights)
    const float* Wt2i; bf16_t* Wto;          // optional: temporal2integration linear_fuse [Ci][C4][2] -> [Ci][2 C4] in fragment order (T2I in front of the forward)
    const float* Wi2t; bf16_t* Wio;          // optional: integration2temporal linear_fuse [C4][Ci] in fragment order (I2T behind it)
    bf16_t* W4o;                             // optional: its transpose [Ci][C4] (I2T backward behind the fused backward)
    bf16_t* W5o;                             // optional: the T2I weight as [2 C4][Ci] (row a C4 + c = W[:, c, a]): T2I backward behind that
};

template <int CI, int C4>
__global__ __launch_bounds__(256) void integ_pack_kernel(const IgPack* __restrict__ descs, const IgPack one) {
    constexpr int CC = CI + C4;
    constexpr int KS1 = CI / 32, NP1 = CC / 32, KS2 = 3 * C4 / 32, NP2 = C4 / 32, KS3 = CC / 32, NP3 = CI / 32;
    constexpr int F1 = NP1 * KS1 * 2, F2 = NP2 * KS2 * 2, F3 = NP3 * KS3 * 2;      // fragments per matrix
    constexpr int PIECES = (F1 + F2 + F3) * 64;
    constexpr int PBLK = (PIECES + 255) / 256;
    constexpr int NB = CC + C4 + CI;                                              // bias outputs, one wave each
    const IgPack& d = descs ? descs[blockIdx.y] : one;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < 2 * PBLK) {
        const bool bwd = (int)blockIdx.x >= PBLK;          // second half of the piece blocks: B1 [CC][CI], B2 [C4][3 C4], B3 [CI][CC]
        if (bwd && !d.B1o) return;
        int piece = ((int)blockIdx.x - (bwd ? PBLK : 0)) * 256 + tid;
        if (piece >= PIECES) return;
        int which = 0;
        if (piece >= F1 * 64) { piece -= F1 * 64; which = 1; if (piece >= F2 * 64) { piece -= F2 * 64; which = 2; } }
        const int KS = which == 0 ? KS1 : (which == 1 ? KS2 : KS3);
        const int l = piece & 63, frag = piece >> 6, q = frag & 1, pk = frag >> 1, ks = pk % KS, pr = pk / KS;
        const int n = 32 * pr + 8 * ((l & 15) >> 2) + 4 * q + (l & 3), k0 = 32 * ks + 8 * (l >> 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e;
            float v;
            if (!bwd) {
                if (which == 0) v = n < CI ? d.Wa[(long)n * CI + k] * d.ga[k] : d.Wb[(long)(n - CI) * CI + k] * d.gb[k];
                else if (which == 1) { const int tap = k / C4, ci = k - tap * C4; v = d.W2[((long)n * C4 + ci) * 3 + tap]; }
                else v = k < CI ? d.Wp[(long)n * CI + k] : d.Wt[(long)n * C4 + (k - CI)];
            } else {
                if (which == 0) v = n < CI ? d.Wp[(long)k * CI + n] : d.Wt[(long)k * C4 + (n - CI)];                       // B1[n][k] = [Wp | Wt][k][n]
                else if (which == 1) { const int tap = k / C4, co = k - tap * C4; v = d.W2[((long)co * C4 + n) * 3 + tap]; }  // B2[ci][tap C4 + co] = W2[co][ci][tap]
                else v = k < CI ? d.Wa[(long)k * CI + n] * d.ga[n] : d.Wb[(long)(k - CI) * CI + n] * d.gb[n];              // B3[k'][n'] = [Wa' ; Wb'][n'][k']
            }
            o[e] = (bf16_t)v;
        }
        bf16_t* dst = bwd ? (which == 0 ? d.B1o : (which == 1 ? d.B2o : d.B3o)) : (which == 0 ? d.W1o : (which == 1 ? d.W2o : d.W3o));
        *reinterpret_cast<bf16x8*>(dst + (long)piece * 8) = o;
        return;
    }
    constexpr int KST = 2 * C4 / 32, KS4 = C4 / 32, FT = NP3 * KST * 2, FI = NP2 * KS1 * 2, F4 = NP3 * KS4 * 2, F5 = (2 * C4 / 32) * KS1 * 2, TBLK = ((FT + FI + F4 + F5) * 64 + 255) / 256;
    if ((int)blockIdx.x < 2 * PBLK + TBLK) {               // T2I weight: Wt[n][a C4 + c] = W[n][c][a]; the I2T weight [C4][Ci] as it is; its transpose [Ci][C4]
        int piece = ((int)blockIdx.x - 2 * PBLK) * 256 + tid;
        if (piece >= (FT + FI + F4 + F5) * 64) return;
        int which = 0;
        if (piece >= FT * 64) { piece -= FT * 64; which = 1; if (piece >= FI * 64) { piece -= FI * 64; which = 2; if (piece >= F4 * 64) { piece -= F4 * 64; which = 3; } } }
        bf16_t* dst = which == 0 ? d.Wto : (which == 1 ? d.Wio : (which == 2 ? d.W4o : d.W5o));
        if (!dst) return;
        const int KS = which == 0 ? KST : (which == 2 ? KS4 : KS1);
        const int l = piece & 63, frag = piece >> 6, q = frag & 1, pk = frag >> 1, ks = pk % KS, pr = pk / KS;
        const int n = 32 * pr + 8 * ((l & 15) >> 2) + 4 * q + (l & 3), k0 = 32 * ks + 8 * (l >> 4);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + e, a = k / C4, c = k - a * C4;
            float v;
            if (which == 0) v = d.Wt2i[((long)n * C4 + c) * 2 + a];
            else if (which == 1) v = d.Wi2t[(long)n * CI + k];
            else if (which == 2) v = d.Wi2t[(long)k * CI + n];
            else v = d.Wt2i[((long)k * C4 + (n % C4)) * 2 + n / C4];          // W5[a C4 + c][k] = W[k][c][a]
            o[e] = (bf16_t)v;
        }
        *reinterpret_cast<bf16x8*>(dst + (long)piece * 8) = o;
        return;
    }
    // biases: b1 = [ba + Wa beta_a ; bb + Wb beta_b] (the LayerNorm shift through the weights), b2, b3 = bp + bt
    const int w = ((int)blockIdx.x - 2 * PBLK - TBLK) * 4 + (tid >> 6), lane = tid & 63;
    if (w >= NB) return;
    if (w < CC) {
        const float* row = w < CI ? d.Wa + (long)w * CI : d.Wb + (long)(w - CI) * CI;
        const float* beta = w < CI ? d.bta : d.btb;
        float s = 0.f;
        for (int k = lane; k < CI; k += 64) s += row[k] * beta[k];
        s = wave_sum(s, 64);
        if (lane == 0) d.b1o[w] = s + (w < CI ? d.ba[w] : d.bb[w - CI]);
    } else if (w < CC + C4) {
        if (lane == 0) d.b2o[w - CC] = d.b2[w - CC];
    } else {
        if (lane == 0) d.b3o[w - CC - C4] = d.bp[w - CC - C4] + d.bt[w - CC - C4];
    }
}

// DBG: timing-only ablation (results WRONG with any bit set): 1 = no weight loads, 2 = no MFMAs, 4 = no QuickGELU, 8 = no global stores,
// 16 = no LDS fragment reads, 32 = no row loads.  Only DBG = 0 is compiled unless the file is built with -DDIST_INTEG_ABLATE
// (tools/integ_ablate.sh), which adds the variants DIST_AMD_INTEG_DBG can select.
// ---- LayerNorm fold, backward side ----------------------------------------------------------------------------------------------------
// The weight-gradient GEMM of a folded Linear z = W (xhat gamma + beta) + b reads xhat, so it leaves G' = dz^T xhat in W's gradient slot.
// In place:  dW = G' diag(gamma) + db beta^T,  and  dgamma[k] += sum_n W[n][k] G'[n][k],  dbeta[k] += sum_n W[n][k] db[n].
// One block per 64 columns k (the lanes of a wave walk consecutive k: coalesced), its 16 waves split the rows n and keep 8 rows of loads in
// flight (12 blocks of 4 waves walking 96 dependent rows each took 69 us on the weight-gradient stream); fixed summation order.
struct IgUnfold { const float *W, *gamma, *beta; float *G; float* db; float *dgamma, *dbeta; int N, K; const float *Gs, *dbs; };
__global__ __launch_bounds__(1024) void integ_unfold_kernel(const IgUnfold a, const IgUnfold b) {
    const int nba = a.K / 64;
    const IgUnfold& d = (int)blockIdx.x < nba ? a : b;
    const int kl = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int kb = (int)blockIdx.x < nba ? blockIdx.x : blockIdx.x - nba;
    const int k = kb * 64 + kl;
    __shared__ float red[2][16][64];
    const float g = d.gamma[k], be = d.beta[k];
    float sg = 0.f, sb = 0.f;
    const int per = (d.N + 15) / 16, n0 = wv * per, n1 = min(n0 + per, d.N);
    // accumulating form: this pass's G' / db come from scratch (Gs / dbs), the slots d.G / d.db hold earlier passes' gradients and are added to
    const bool acc = d.Gs != nullptr;
    const float* Gin = acc ? d.Gs : d.G;                 // (no __restrict__: in the plain form they ARE d.G / d.db, which this loop writes - each thread's
    const float* dbin = acc ? d.dbs : d.db;             //  store follows its own load of the same element, which is all the ordering the kernel needs)
    for (int n = n0; n < n1; n += 8) {
        float w[8], gp[8], dbn[8], old[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int nn = min(n + u, n1 - 1);
            w[u] = d.W[(long)nn * d.K + k]; gp[u] = Gin[(long)nn * d.K + k]; dbn[u] = dbin[nn];
            old[u] = acc ? d.G[(long)nn * d.K + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (n + u < n1) {
                sg += w[u] * gp[u]; sb += w[u] * dbn[u];
                d.G[(long)(n + u) * d.K + k] = old[u] + (gp[u] * g + dbn[u] * be);
                if (acc && kb == 0 && kl == 0) d.db[n + u] += dbn[u];
            }
        }
    }
    red[0][wv][kl] = sg; red[1][wv][kl] = sb;
    __syncthreads();
    if (wv == 0) {
        float tg = 0.f, tb = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) { tg += red[0][u][kl]; tb += red[1][u][kl]; }
        d.dgamma[k] += tg;
        d.dbeta[k] += tb;
    }
}

template <int CI, int C4, int BM, int MODE, int DBG, bool T2I = false>
int launch_integ_v(const IgArgs& a, hipStream_t s) {
    const int smem = BM * CI * 2 + 2 * BM * C4 * 2 + (T2I ? 9 * BM * 2 * 4 : 0);
    static DistSmemOnce attr;
    RUN_(dist_max_smem(attr, (const void*)integ_fwd_kernel<CI, C4, BM, MODE, DBG, T2I>, (size_t)smem));
    hipLaunchKernelGGL((integ_fwd_kernel<CI, C4, BM, MODE, DBG, T2I>), dim3((unsigned)(a.clips * a.groups)), dim3(512), smem, s, a);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}
template <int CI, int C4, int BM>
int launch_integ(const IgArgs& a, const int mode, hipStream_t s) {
#ifdef DIST_INTEG_ABLATE
    static const int dbg = dist_measure_knob("DIST_AMD_INTEG_DBG", 0);
#define IG_VARIANT(D) if (dbg == D) return mode ? launch_integ_v<CI, C4, BM, 1, D>(a, s) : launch_integ_v<CI, C4, BM, 0, D>(a, s);
    IG_VARIANT(1) IG_VARIANT(2) IG_VARIANT(4) IG_VARIANT(8) IG_VARIANT(16) IG_VARIANT(32) IG_VARIANT(3) IG_VARIANT(19) IG_VARIANT(23) IG_VARIANT(63) IG_VARIANT(59)
#undef IG_VARIANT
#endif
    if constexpr (BM == 128) {
        if (a.Xp && mode == 2) return launch_integ_v<CI, C4, BM, 2, 0, true>(a, s);
        if (a.Xp && mode == 0) return launch_integ_v<CI, C4, BM, 0, 0, true>(a, s);
    }
    if (a.Xp) return DIST_ERR_ARG;
    return mode == 2 ? launch_integ_v<CI, C4, BM, 2, 0>(a, s) : (mode == 1 ? launch_integ_v<CI, C4, BM, 1, 0>(a, s) : launch_integ_v<CI, C4, BM, 0, 0>(a, s));
}

}  // namespace

bool dist_k_integ_eligible(int dtype, int Ci, int C4, int t, int tk) {
    return dtype == DIST_BF16 && Ci == 384 && C4 == 96 && tk == 3 && (t == 4 || t == 8 || t == 16 || t == 32);
}

extern "C" int64_t dist_op_integration_pack_elems(int Ci, int C4, int which) {
    const int64_t CC = Ci + C4;
    switch (which) {
        case 0: return CC * Ci;                 // W1 (bf16 elements)
        case 1: return (int64_t)C4 * 3 * C4;    // W2
        case 2: return (int64_t)Ci * CC;        // W3
        case 3: return CC;                      // b1 (floats)
        case 4: return C4;                      // b2
        case 5: return Ci;                      // b3
        case 6: return (int64_t)Ci * 2 * C4;    // Wt (bf16 elements): the T2I weight in front of the forward
        case 7: return (int64_t)C4 * Ci;        // Wi: the I2T weight behind it
        case 8: return (int64_t)Ci * C4;        // W4: its transpose, the I2T backward behind the fused backward
        case 9: return (int64_t)2 * C4 * Ci;    // W5: the T2I weight transposed, the T2I backward behind that
        default: return -1;
    }
}

static IgPack ig_pack_of(const dist_integ_pack_args& a) {
    IgPack d;
    d.Wa = a.ffn_fc_w; d.ba = a.ffn_fc_b; d.ga = a.ln_w; d.bta = a.ln_b;
    d.Wb = a.tf_fc1_w; d.bb = a.tf_fc1_b; d.gb = a.ln_t_w; d.btb = a.ln_t_b;
    d.W2 = a.tf_fc2_w; d.b2 = a.tf_fc2_b;
    d.Wp = a.ffn_proj_w; d.bp = a.ffn_proj_b; d.Wt = a.tf_proj_w; d.bt = a.tf_proj_b;
    d.W1o = static_cast<bf16_t*>(a.W1); d.W2o = static_cast<bf16_t*>(a.W2); d.W3o = static_cast<bf16_t*>(a.W3);
    d.b1o = a.b1; d.b2o = a.b2; d.b3o = a.b3;
    d.B1o = static_cast<bf16_t*>(a.B1); d.B2o = static_cast<bf16_t*>(a.B2); d.B3o = static_cast<bf16_t*>(a.B3);
    d.Wt2i = a.t2i_w; d.Wto = static_cast<bf16_t*>(a.Wt);
    d.Wi2t = a.i2t_w; d.Wio = static_cast<bf16_t*>(a.Wi); d.W4o = static_cast<bf16_t*>(a.W4); d.W5o = static_cast<bf16_t*>(a.W5);
    return d;
}

int64_t dist_k_integ_pack_desc_bytes() { return (int64_t)sizeof(IgPack); }
void dist_k_integ_pack_desc(const dist_integ_pack_args* a, void* out) { *static_cast<IgPack*>(out) = ig_pack_of(*a); }

// `descs_dev`: n descriptors (dist_k_integ_pack_desc) in device memory, or nullptr with n == 1 and `one` passed by value
int dist_k_integ_pack(const void* descs_dev, const dist_integ_pack_args* one, int n, int Ci, int C4, hipStream_t s) {
    if (Ci != 384 || C4 != 96 || n <= 0) return DIST_ERR_ARG;
    constexpr int CI = 384, C4c = 96, CC = CI + C4c;
    constexpr int PIECES = ((CC / 32) * (CI / 32) * 2 + (C4c / 32) * (3 * C4c / 32) * 2 + (CI / 32) * (CC / 32) * 2) * 64;
    const int pblk = (PIECES + 255) / 256, bblk = (CC + C4c + CI + 3) / 4, tblk = (((CI / 32) * (2 * C4c / 32) * 2 + (C4c / 32) * (CI / 32) * 2 + (CI / 32) * (C4c / 32) * 2 + (2 * C4c / 32) * (CI / 32) * 2) * 64 + 255) / 256;
    IgPack d{};
    if (!descs_dev) { if (!one || n != 1) return DIST_ERR_ARG; d = ig_pack_of(*one); }
    hipLaunchKernelGGL((integ_pack_kernel<CI, C4c>), dim3((unsigned)(2 * pblk + tblk + bblk), (unsigned)n), dim3(256), 0, s, static_cast<const IgPack*>(descs_dev), d);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_integration_pack(const dist_integ_pack_args* a, void* stream) {
    if (!a || !a->ffn_fc_w || !a->ffn_fc_b || !a->ln_w || !a->ln_b || !a->tf_fc1_w || !a->tf_fc1_b || !a->ln_t_w || !a->ln_t_b || !a->tf_fc2_w ||
        !a->tf_fc2_b || !a->ffn_proj_w || !a->ffn_proj_b || !a->tf_proj_w || !a->tf_proj_b || !a->W1 || !a->W2 || !a->W3 || !a->b1 || !a->b2 || !a->b3)
        return DIST_ERR_ARG;
    if (!dist_k_integ_eligible(DIST_BF16, a->Ci, a->C4, 8, 3)) return DIST_ERR_ARG;
    return dist_k_integ_pack(nullptr, a, 1, a->Ci, a->C4, static_cast<hipStream_t>(stream));
}

extern "C" int dist_op_integration_bwd(const dist_integ_bwd_args* a, void* stream) {
    if (!a || !a->dR || !a->zf_h2 || !a->Xhat || !a->rstd || !a->B1 || !a->B2 || !a->B3 || !a->dzf_dh2 || !a->dh1 || !a->dMp) return DIST_ERR_ARG;
    if (a->clips <= 0 || a->t <= 0 || a->L <= 0) return DIST_ERR_ARG;
    if (!dist_k_integ_eligible(a->dtype, a->Ci, a->C4, a->t, a->tk)) return DIST_ERR_ARG;
    // (timing-only library: 64-row tiles walked by four waves, two workgroups per CU - measure/integ4.hip, measured and rejected in round 6)
    static const int w4_env = DIST_AB_KNOB("DIST_AMD_INTEG_W4", 0);
    const bool w4 = w4_env != 0 && a->t <= 64 && 64 % a->t == 0;
    const int BM = w4 ? 64 : 128;
    if (BM % a->t) return DIST_ERR_ARG;
    IgBwdArgs k;
    k.dR = static_cast<const bf16_t*>(a->dR); k.zfh2 = static_cast<const bf16_t*>(a->zf_h2); k.Xh = static_cast<const bf16_t*>(a->Xhat); k.rstd = a->rstd;
    k.W1 = static_cast<const bf16_t*>(a->B1); k.W2 = static_cast<const bf16_t*>(a->B2); k.W3 = static_cast<const bf16_t*>(a->B3);
    k.dzfh2 = static_cast<bf16_t*>(a->dzf_dh2); k.dh1 = static_cast<bf16_t*>(a->dh1);
    k.ldz = a->ld_dzf > 0 ? a->ld_dzf : a->Ci + a->C4;
    k.dh2 = a->dh2 ? static_cast<bf16_t*>(a->dh2) : k.dzfh2 + a->Ci;
    k.ld2 = a->dh2 ? (a->ld_dh2 > 0 ? a->ld_dh2 : a->C4) : k.ldz;
    k.ld1 = a->ld_dh1 > 0 ? a->ld_dh1 : a->C4;
    if (k.ldz % 8 || k.ld2 % 8 || k.ld1 % 8) return DIST_ERR_ARG;
    k.dMp = static_cast<bf16_t*>(a->dMp); k.dM = static_cast<bf16_t*>(a->dM_copy);
    k.add_dR = a->add_dR ? 1 : 0;
    k.dm_cls = a->dM_cls_only ? 1 : 0;
    k.dXn = static_cast<const bf16_t*>(a->i2t_dXnext); k.W4 = static_cast<const bf16_t*>(a->i2t_B); k.dY = static_cast<bf16_t*>(a->i2t_dY);
    if ((k.dY != nullptr) != (k.dXn != nullptr && k.W4 != nullptr) || (k.dY && !k.dM)) return DIST_ERR_ARG;   // the I2T term: dX_next, its operand, the dY output and dM_copy (which then receives dM) - all or none
    k.dcls = a->t2i_dcls;
    k.W5 = static_cast<const bf16_t*>(a->t2i_B); k.pact = static_cast<const bf16_t*>(a->t2i_p); k.dp = static_cast<bf16_t*>(a->t2i_dp);
    if (k.dp && !(k.W5 && k.pact)) return DIST_ERR_ARG;
    k.clips = a->clips; k.t = a->t; k.L = a->L;
    const int TOK = BM / a->t;
    int sh = 0;
    while ((1 << sh) < TOK) ++sh;
    k.tokshift = sh;
    k.groups = (a->L + TOK - 1) / TOK;
#ifdef DIST_AMD_MEASURE
    if (w4) { const int rc = integ_bwd4_launch(k, static_cast<hipStream_t>(stream)); return rc < 0 ? rc : DIST_OK; }
#endif
    const int smem = 128 * 384 * 2 + 2 * 128 * 96 * 2 + 9 * 128 * 2 * 4;
    static DistSmemOnce attr;
    RUN_(dist_max_smem(attr, (const void*)integ_bwd_kernel<384, 96, 128>, (size_t)smem));
    hipLaunchKernelGGL((integ_bwd_kernel<384, 96, 128>), dim3((unsigned)(k.clips * k.groups)), dim3(512), smem, static_cast<hipStream_t>(stream), k);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_integration_unfold(const dist_integ_unfold_args* a, void* stream) {
    if (!a || !a->ffn_fc_w || !a->ln_w || !a->ln_b || !a->d_ffn_fc_w || !a->d_ffn_fc_b || !a->d_ln_w || !a->d_ln_b || !a->tf_fc1_w || !a->ln_t_w ||
        !a->ln_t_b || !a->d_tf_fc1_w || !a->d_tf_fc1_b || !a->d_ln_t_w || !a->d_ln_t_b) return DIST_ERR_ARG;
    if (a->Ci <= 0 || a->Ci % 64 || a->C4 <= 0) return DIST_ERR_ARG;
    const int ng = (a->g_ffn_fc_w != nullptr) + (a->g_ffn_fc_b != nullptr) + (a->g_tf_fc1_w != nullptr) + (a->g_tf_fc1_b != nullptr);
    if (ng != 0 && ng != 4) return DIST_ERR_ARG;
    IgUnfold x{a->ffn_fc_w, a->ln_w, a->ln_b, a->d_ffn_fc_w, a->d_ffn_fc_b, a->d_ln_w, a->d_ln_b, a->Ci, a->Ci, a->g_ffn_fc_w, a->g_ffn_fc_b};
    IgUnfold y{a->tf_fc1_w, a->ln_t_w, a->ln_t_b, a->d_tf_fc1_w, a->d_tf_fc1_b, a->d_ln_t_w, a->d_ln_t_b, a->C4, a->Ci, a->g_tf_fc1_w, a->g_tf_fc1_b};
    hipLaunchKernelGGL(integ_unfold_kernel, dim3((unsigned)(2 * a->Ci / 64)), dim3(1024), 0, static_cast<hipStream_t>(stream), x, y);
    HIP_CHECK_RET(hipGetLastError());
    return DIST_OK;
}

extern "C" int dist_op_integration_fwd(const dist_integ_args* a, void* stream) {
    if (!a || !a->W1 || !a->W2 || !a->W3 || !a->b1 || !a->b2 || !a->b3 || !a->R) return DIST_ERR_ARG;
    const bool t2i = a->t2i_Xp != nullptr;              // M' is formed in the kernel: M, X', the packed T2I weight, its bias, the cls tokens instead of Mp
    if (t2i ? !(a->t2i_M && a->t2i_W && a->t2i_bias && a->t2i_cls) : !a->Mp) return DIST_ERR_ARG;
    if (a->clips <= 0 || a->t <= 0 || a->L <= 0) return DIST_ERR_ARG;
    if (!dist_k_integ_eligible(a->dtype, a->Ci, a->C4, a->t, a->tk)) return DIST_ERR_ARG;
    const bool train = a->Na || a->Nb || a->Xhat || a->zf_h2 || a->hf_g2 || a->h1 || a->mean || a->rstd;
    int mode = 0;
    if (train) {                                           // the tensors backward reads come all or none; the normalised rows either as xhat or as Na + Nb
        if (!(a->zf_h2 && a->hf_g2 && a->h1 && a->mean && a->rstd)) return DIST_ERR_ARG;
        if (a->Xhat) { if (a->Na || a->Nb) return DIST_ERR_ARG; mode = 2; }
        else { if (!(a->Na && a->Nb && a->ln_w && a->ln_b && a->ln_t_w && a->ln_t_b)) return DIST_ERR_ARG; mode = 1; }
    }
    static const int bm_env = DIST_AB_KNOB("DIST_AMD_INTEG_BM", 0);     // A/B: 64-row tiles, two workgroups per CU (timing-only library)
    int BM = bm_env == 64 || bm_env == 128 ? bm_env : 128;
    // round 6 (timing-only library): the T2I-in-front forms on 64-row tiles walked by FOUR waves, two independent workgroups per CU at 256 registers each
    // (measure/integ4.hip; measured and rejected, profiles/r06_integ_w4.md)
    static const int w4_env = DIST_AB_KNOB("DIST_AMD_INTEG_W4", 0);
    const bool w4 = w4_env != 0 && t2i && (mode == 0 || mode == 2) && a->t <= 64 && 64 % a->t == 0 && bm_env == 0;
    if (w4) BM = 64;
    if (BM / a->t < 1 || (BM % a->t)) return DIST_ERR_ARG;
    IgArgs k;
    k.Mp = static_cast<const bf16_t*>(a->Mp);
    k.W1 = static_cast<const bf16_t*>(a->W1); k.W2 = static_cast<const bf16_t*>(a->W2); k.W3 = static_cast<const bf16_t*>(a->W3);
    k.b1 = a->b1; k.b2 = a->b2; k.b3 = a->b3;
    k.ga = a->ln_w; k.ba = a->ln_b; k.gb = a->ln_t_w; k.bb = a->ln_t_b;
    k.R = static_cast<bf16_t*>(a->R); k.Na = static_cast<bf16_t*>(a->Na); k.Nb = static_cast<bf16_t*>(a->Nb); k.Xh = static_cast<bf16_t*>(a->Xhat);
    k.zfh2 = static_cast<bf16_t*>(a->zf_h2); k.hfg2 = static_cast<bf16_t*>(a->hf_g2); k.h1 = static_cast<bf16_t*>(a->h1);
    k.mean = a->mean; k.rstd = a->rstd;
    k.clips = a->clips; k.t = a->t; k.L = a->L;
    k.M = static_cast<const bf16_t*>(a->t2i_M); k.Xp = static_cast<const bf16_t*>(a->t2i_Xp); k.Wt = static_cast<const bf16_t*>(a->t2i_W);
    k.bt = a->t2i_bias; k.cls = a->t2i_cls; k.Mpo = static_cast<bf16_t*>(a->Mp_out);
    k.Wi = static_cast<const bf16_t*>(a->i2t_W); k.bi = a->i2t_bias; k.Xn = static_cast<bf16_t*>(a->i2t_Xnext);
    if (k.Xn && !(t2i && k.Wi && k.bi)) return DIST_ERR_ARG;           // I2T behind needs T2I in front (its X' tile) and its own operands
    if (t2i && ((BM != 128 && !w4) || a->Na)) return DIST_ERR_ARG;
    const int TOK = BM / a->t;
    int sh = 0;
    while ((1 << sh) < TOK) ++sh;
    k.tokshift = sh;
    k.groups = (a->L + TOK - 1) / TOK;
    k.eps = a->eps > 0.f ? a->eps : 1e-5f;
    hipStream_t s = static_cast<hipStream_t>(stream);
#ifdef DIST_AMD_MEASURE
    if (w4) { const int rc = integ_fwd4_launch(k, mode, s); return rc < 0 ? rc : (rc == 1 ? DIST_OK : DIST_ERR_ARG); }
    if (BM == 64) return launch_integ<384, 96, 64>(k, mode, s);        // (measured, profiles/r03_integ_fused.md: faster in inference, slower in training)
#endif
    return launch_integ<384, 96, 128>(k, mode, s);
}
