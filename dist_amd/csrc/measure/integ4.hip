// Fused IntegrationNetwork kernels, 64-row form: FOUR waves per workgroup, TWO workgroups per CU (round 6).
//
// Same mathematics, same operand packs, same rounding points and the same K order of every product as integ.hip's kernels (the LayerNorm row sums
// combine four waves' partials instead of eight: last-bit differences in mean / rstd only; reference
// models/module_zoo/branches/dist.py:16-45, :68-105) - what changes is who shares a CU.  integ.hip walks one 128-row tile per CU through its stages
// (T2I / I2T, LayerNorm, three GEMM stages, their stores) with nothing else resident: 154 KB of LDS, 512 threads at 256 registers.  Its MFMA
// phases, its row loads and its 0.5 MB of stores per tile therefore run one after the other (profiles/r03_integ_fused.md: 43 us of MFMA phases
// for 17 us of MFMA work; rounds 4 / 5 VERDICT item "overlap inside the fused IntegrationNetwork kernels").  The 64-row tile of round 3
// (DIST_AMD_INTEG_BM=64) kept EIGHT waves per workgroup, i.e. four waves per SIMD at 128 registers: it spilled 340-396 bytes per lane inside the
// K loops and lost.  Here a tile is 64 rows = all t frames of TOK = 64 / t tokens, walked by 4 waves (one per SIMD) that keep 256 registers:
// two workgroups of 75 KB share a CU, each a complete, independent instance of the stage sequence - while one waits for its row loads, its
// store acknowledgements or a barrier, the other one's MFMAs have the matrix pipes.  No code couples the two: the hardware interleaves them.
//
// Work split (NW = 4, RB = 4 sixteen-row blocks): a wave owns whole COLUMN PAIRS (32 output columns) over all four row blocks -
//   T2I, stage 3: pairs 3w .. 3w + 2 (of 12);  stage 1: pairs 4w .. 4w + 3 (of 15: twelve of zf, three of h1);  I2T, stage 2: pair w (waves 0 .. 2).
// A weight fragment (1 KB, straight from L2 in MFMA operand order) feeds 4 MFMAs instead of 8: the weight stream per row doubles (2 MB per
// 128 rows and CU, from the XCD's L2), which is what the second resident workgroup was meant to hide.
//
// MEASURED (round 6, profiles/r06_integ_w4.md): parity-green on every test of tests/test_integ_gpu.py and through the engine, and NOT faster: forward 115-118 us
// against 114-118 (inference form 98.5 against 94.5-101), backward 154 against 147, step 16.50-16.54 against 16.48-16.51 ms.  The counters say why: in both forms
// the texture-data unit sits stalled on L2 data for 40 % of the launch (TD_TC_STALL 121 k of 299 k cycles per CU) and the MFMA pipes are busy for 17 % - the K loops
// are bound by the WEIGHT STREAM from L2 (1 MB per 128-row tile, every fragment used by exactly one wave), and two workgroups per CU read it 1.7 x as often
// (TCP_TCC_READ_REQ 6.9 M against 4.1 M).  What the second workgroup overlaps it pays for in weight traffic.  Timing-only library only (DIST_AMD_INTEG_W4=1).
#ifndef DIST_AMD_MEASURE
#error "measure/integ4.hip belongs to the timing-only library (python -m dist_amd.build --measure)"
#endif
#include <stdlib.h>
#include "../common.h"
#include "../kernels.h"
#include "../integ_common.h"

namespace {

constexpr int W4_BM = 64, W4_NW = 4, W4_RB = 4;

template <int CI, int C4, int MODE>
__global__ __launch_bounds__(256, 2) void integ_fwd4_kernel(const IgArgs p) {
    constexpr int DBG = 0;
    constexpr bool TRAIN = MODE != 0;            // MODE 2: xhat, [zf | h2], [hf | g2], h1, mean, rstd are written for the backward pass; MODE 0: inference
    constexpr int BM = W4_BM, NW = W4_NW, RB = W4_RB;
    constexpr int CC = CI + C4;
    constexpr int KS1 = CI / 32, NP1 = CC / 32;                  // stage 1: K = Ci, N = Ci + C4
    constexpr int KT = C4 / 32, KS2 = 3 * KT, NP2 = C4 / 32;     // stage 2: three temporal taps of K = C4, N = C4
    constexpr int KS3 = CC / 32, NP3 = CI / 32;                  // stage 3: K = Ci + C4, N = Ci
    constexpr int KST = 2 * C4 / 32;                             // T2I: K = 2 C4 (two frames), N = Ci
    constexpr int PW3 = NP3 / NW;                                // pairs per wave of the N = Ci products (3)
    constexpr int PW1 = (NP1 + NW - 1) / NW;                     // ... of stage 1 (4; the last wave has NP1 - 3 * PW1 = 3)
    static_assert(NP3 % NW == 0 && NP2 <= NW - 1 + 1 && NP2 <= NW && PW1 * (NW - 1) < NP1 && C4 % 32 == 0 && CI % 128 == 0, "geometry");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const regA = smem;                               // [BM][CI]  M, M', xhat, then hf
    char* const regB = smem + BM * CI * 2;                 // [BM][C4]  X' (frame 2f), then h1
    char* const regC = regB + BM * C4 * 2;                 // [BM][C4]  X' (frame 2f + 1), then g2
    float* const red = reinterpret_cast<float*>(regC + BM * C4 * 2);   // [NW][BM][2] partial row sums
    float* const tot = red + NW * BM * 2;                  // [BM][2] (mean, rstd)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int clip = blockIdx.x / p.groups, grp = blockIdx.x - clip * p.groups;
    const int tsh = p.tokshift, TOK = 1 << tsh, tokmask = TOK - 1;
    const int L = p.L, t = p.t;
    const int N = L - 1, T = 2 * t;
    // tile row r = frame * TOK + token slot; global row of it, or -1 for a token slot beyond L (last group of a clip)
    auto grow_of = [&](const int r) -> int {
        const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
        return j < L ? (clip * t + f) * L + j : -1;
    };
    auto slot = [&](const int r, const int pr) __attribute__((always_inline)) { return reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pr * 4 + lg) ^ li) << 4)); };

    // ---------------- stage -1 (T2I, dist.py:68-86): M' = M + [cls_token_f ; X'(frames 2f, 2f+1; position j-1) Wt^T + bt], LayerNorm statistics, xhat
    {
        const int lq4 = lane & 3, lrow = lane >> 2;
        const int r = wid * 16 + lrow, f = r >> tsh, j = (grp << tsh) + (r & tokmask);
        const bool ok = j >= 1 && j < L;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const bf16_t* src = p.Xp + ((long)(clip * T + 2 * f + a) * N + (ok ? j - 1 : 0)) * C4;
            char* dst = (a ? regC : regB) + r * (C4 * 2);
#pragma unroll
            for (int m = 0; m < C4 / 32; ++m) {
                const int c = lq4 + 4 * m;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(src + c * 8);
                if (!ok) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<bf16x8*>(dst + (ig_pchunk(r, c) << 4)) = v;
            }
        }
    }
    int gro[RB];
    {
        // M at this lane's (row, 8 columns) positions of the wave's three pairs (rows beyond L are clamped: finite numbers, never stored) -> region A,
        // row-major: the B operand of the I2T product, and where each lane finds its pieces again (M, then M', then xhat share one slot)
        bf16x8 mv[PW3][RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = rb * 16 + li;
            const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
            gro[rb] = j < L ? (clip * t + f) * L + j : -1;
            const long base = (long)((clip * t + f) * L + min(j, L - 1)) * CI;
#pragma unroll
            for (int q = 0; q < PW3; ++q) mv[q][rb] = *reinterpret_cast<const bf16x8*>(p.M + base + (PW3 * wid + q) * 32 + lg * 8);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int q = 0; q < PW3; ++q) *slot(rb * 16 + li, PW3 * wid + q) = mv[q][rb];
    }
    const bool i2t = p.Xn != nullptr;                      // (kernel argument: the same for every workgroup)
    __syncthreads();
    if (i2t && wid < NP2) {
        // ---------------- I2T (dist.py:90-105): Linear on the patch rows of M (pair `wid` of the C4 columns), the result added to BOTH frames 2f, 2f+1 of
        // X' (nearest upsampling in time) -> X of the next layer.  Reads region A (M) and regions B / C (X').
        const int pi = wid;
        f32x4 ai[2][RB];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) ai[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 wi[3][2];
        wi[0][0] = IG_LDW(p.Wi, ((long)pi * KS1) * 2); wi[0][1] = IG_LDW(p.Wi, ((long)pi * KS1) * 2 + 1);
        wi[1][0] = IG_LDW(p.Wi, ((long)pi * KS1 + 1) * 2); wi[1][1] = IG_LDW(p.Wi, ((long)pi * KS1 + 1) * 2 + 1);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            if (ks + 2 < KS1) { wi[(ks + 2) % 3][0] = IG_LDW(p.Wi, ((long)pi * KS1 + ks + 2) * 2); wi[(ks + 2) % 3][1] = IG_LDW(p.Wi, ((long)pi * KS1 + ks + 2) * 2 + 1); }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = rb * 16 + li;
                const bf16x8 a = IG_LDS(regA + r * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
                ai[0][rb] = IG_MMA(wi[ks % 3][0], a, ai[0][rb]);
                ai[1][rb] = IG_MMA(wi[ks % 3][1], a, ai[1][rb]);
            }
        }
        const int n0 = pi * 32 + lg * 8;
        float bv[8];
        ig_load8(p.bi + n0, bv);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = rb * 16 + li;
            const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
            if (j < 1 || j >= L) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = ai[0][rb][e] + bv[e]; v[4 + e] = ai[1][rb][e] + bv[4 + e]; }     // (one rounding, of the sum with X')
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
    {
        f32x4 at[PW3][2][RB];
#pragma unroll
        for (int q = 0; q < PW3; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) at[q][h][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            bf16x8 wb[2][PW3 * 2];
            auto ldw = [&](bf16x8 (&w)[PW3 * 2], const int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < PW3; ++q) {
                    w[2 * q] = IG_LDW(p.Wt, ((long)(PW3 * wid + q) * KST + ks) * 2);
                    w[2 * q + 1] = IG_LDW(p.Wt, ((long)(PW3 * wid + q) * KST + ks) * 2 + 1);
                }
            };
            ldw(wb[0], 0);
            ldw(wb[1], 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KST; ++ks) {
                bf16x8 (&w)[PW3 * 2] = wb[ks & 1];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int r = rb * 16 + li;
                    const bf16x8 a = IG_LDS((ks < KST / 2 ? regB : regC) + r * (C4 * 2) + (ig_pchunk(r, (ks % (KST / 2)) * 4 + lg) << 4));
#pragma unroll
                    for (int q = 0; q < PW3; ++q) {
                        at[q][0][rb] = IG_MMA(w[2 * q], a, at[q][0][rb]);
                        at[q][1][rb] = IG_MMA(w[2 * q + 1], a, at[q][1][rb]);
                    }
                }
                if (ks + 2 < KST) ldw(w, ks + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                   // every wave has read M for the I2T product: the slots may change
        // M' (rounded to bf16 as the unfused GEMM stores it) replaces M in its slot; partial row sums of M' and M'^2 over this wave's 96 columns
        float bt[PW3][8];
#pragma unroll
        for (int q = 0; q < PW3; ++q) ig_load8(p.bt + (PW3 * wid + q) * 32 + lg * 8, bt[q]);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = rb * 16 + li;
            const int f = r >> tsh, j = (grp << tsh) + (r & tokmask);
            const bool is_cls = j == 0;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int q = 0; q < PW3; ++q) {
                float cl[8];
                if (is_cls) ig_load8(p.cls + (long)f * CI + (PW3 * wid + q) * 32 + lg * 8, cl);
                bf16x8 mv = *slot(r, PW3 * wid + q);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float add = is_cls ? cl[e] : (e < 4 ? at[q][0][rb][e] : at[q][1][rb][e - 4]) + bt[q][e];
                    const bf16_t v = (bf16_t)((float)mv[e] + add);
                    mv[e] = v;
                    s1 += (float)v; s2 += (float)v * (float)v;
                }
                *slot(r, PW3 * wid + q) = mv;
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lg == 0) reinterpret_cast<float2*>(red)[wid * BM + r] = make_float2(s1, s2);
        }
    }
    __syncthreads();
    if (tid < BM) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { const float2 v = reinterpret_cast<const float2*>(red)[w * BM + tid]; a1 += v.x; a2 += v.y; }
        constexpr float invC = 1.f / (float)CI;
        const float mu = a1 * invC;
        const float rs = rsqrtf(fmaxf(a2 * invC - mu * mu, 0.f) + p.eps);
        reinterpret_cast<float2*>(tot)[tid] = make_float2(mu, rs);
        const int gr = grow_of(tid);
        if (TRAIN && gr >= 0) { p.mean[gr] = mu; p.rstd[gr] = rs; }
    }
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int r = rb * 16 + li;
        const float2 st = reinterpret_cast<const float2*>(tot)[r];
#pragma unroll
        for (int q = 0; q < PW3; ++q) {
            const bf16x8 mp = *slot(r, PW3 * wid + q);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(((float)mp[e] - st.x) * st.y);
            *slot(r, PW3 * wid + q) = o;
            if (gro[rb] >= 0) {
                if (MODE == 2) IG_ST(o, p.Xh + (long)gro[rb] * CI + (PW3 * wid + q) * 32 + lg * 8);
                if (p.Mpo) IG_ST(mp, p.Mpo + (long)gro[rb] * CI + (PW3 * wid + q) * 32 + lg * 8);
            }
        }
    }
    __syncthreads();

    // ---------------- stage 1: [zf | h1] = xhat W1^T + b1, pairs 4 wid .. 4 wid + 3 (the last wave repeats its last pair and drops the copy: no branch in the K loop)
    {
        int pr1[PW1];
#pragma unroll
        for (int q = 0; q < PW1; ++q) pr1[q] = min(PW1 * wid + q, NP1 - 1);
        f32x4 acc[PW1 * 2][RB];
#pragma unroll
        for (int f = 0; f < PW1 * 2; ++f)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[f][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 wb[2][PW1 * 2];
        auto ldw = [&](bf16x8 (&w)[PW1 * 2], const int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < PW1; ++q) {
                w[2 * q] = IG_LDW(p.W1, ((long)pr1[q] * KS1 + ks) * 2);
                w[2 * q + 1] = IG_LDW(p.W1, ((long)pr1[q] * KS1 + ks) * 2 + 1);
            }
        };
        ldw(wb[0], 0);
        ldw(wb[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            bf16x8 (&w)[PW1 * 2] = wb[ks & 1];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bf16x8 a = IG_LDS(regA + (rb * 16 + li) * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
#pragma unroll
                for (int f = 0; f < PW1 * 2; ++f) acc[f][rb] = IG_MMA(w[f], a, acc[f][rb]);
            }
            if (ks + 2 < KS1) ldw(w, ks + 2);
            __builtin_amdgcn_sched_barrier(0);             // (the loads stay HERE, a whole k-step ahead of their use)
        }
        __syncthreads();                                   // every wave has read xhat: region A becomes hf (X' in region B was last read before the statistics barriers)
#pragma unroll
        for (int q = 0; q < PW1; ++q) {
            const int pr = PW1 * wid + q;
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
                for (int e = 0; e < 4; ++e) { zb[e] = (bf16_t)(acc[q * 2][rb][e] + bv[e]); zb[4 + e] = (bf16_t)(acc[q * 2 + 1][rb][e] + bv[4 + e]); }
                if (is_zf) {
                    bf16x8 hf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) hf[e] = (bf16_t)IG_GELU((float)zb[e]);
                    *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pr * 4 + lg) ^ li) << 4)) = hf;
                    if (TRAIN && gro[rb] >= 0) {
                        IG_ST(zb, p.zfh2 + (long)gro[rb] * CC + n0);
                        IG_ST(hf, p.hfg2 + (long)gro[rb] * CC + n0);
                    }
                } else {
                    *reinterpret_cast<bf16x8*>(regB + r * (C4 * 2) + (ig_pchunk(r, (pr - NP3) * 4 + lg) << 4)) = zb;
                    if (TRAIN && gro[rb] >= 0) IG_ST(zb, p.h1 + (long)gro[rb] * C4 + (n0 - CI));
                }
            }
        }
    }
    // the temporal weights of this wave's stage-2 pair travel while the workgroup meets at the barrier
    const bool s2 = wid < NP2;
    const int p2 = s2 ? wid : 0;
    bf16x8 w2[KS2][2];
    if (s2) {
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) { w2[ks][0] = IG_LDW(p.W2, ((long)p2 * KS2 + ks) * 2); w2[ks][1] = IG_LDW(p.W2, ((long)p2 * KS2 + ks) * 2 + 1); }
    }
    __syncthreads();                                       // hf in region A, h1 in region B

    // ---------------- stage 2: h2 = conv_t(h1) + b2, g2 = g(h2) -> region C; wave w < 3: pair w over the four row blocks
    if (s2) {
        f32x4 a2[2][RB];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a2[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = rb * 16 + li;
                const int f = (r >> tsh) + tap - 1;
                const bool ok = (unsigned)f < (unsigned)t;
                const int rs = ok ? r + (tap - 1) * TOK : r;
#pragma unroll
                for (int k3 = 0; k3 < KT; ++k3) {
                    bf16x8 a = IG_LDS(regB + rs * (C4 * 2) + (ig_pchunk(rs, k3 * 4 + lg) << 4));
                    if (!ok) a = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    a2[0][rb] = IG_MMA(w2[tap * KT + k3][0], a, a2[0][rb]);
                    a2[1][rb] = IG_MMA(w2[tap * KT + k3][1], a, a2[1][rb]);
                }
            }
        }
        const int n0 = p2 * 32 + lg * 8;
        float bv[8];
        ig_load8(p.b2 + n0, bv);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = rb * 16 + li;
            bf16x8 hb, gb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hb[e] = (bf16_t)(a2[0][rb][e] + bv[e]); hb[4 + e] = (bf16_t)(a2[1][rb][e] + bv[4 + e]); }
#pragma unroll
            for (int e = 0; e < 8; ++e) gb[e] = (bf16_t)IG_GELU((float)hb[e]);
            *reinterpret_cast<bf16x8*>(regC + r * (C4 * 2) + (ig_pchunk(r, p2 * 4 + lg) << 4)) = gb;
            if (TRAIN && gro[rb] >= 0) {
                IG_ST(hb, p.zfh2 + (long)gro[rb] * CC + CI + n0);
                IG_ST(gb, p.hfg2 + (long)gro[rb] * CC + CI + n0);
            }
        }
    }

    // ---------------- stage 3: R = [hf | g2] W3^T + b3, pairs 3 wid .. 3 wid + 2
    {
        f32x4 a3[PW3][2][RB];
#pragma unroll
        for (int q = 0; q < PW3; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a3[q][h][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 wb[2][PW3 * 2];
        auto ldw = [&](bf16x8 (&w)[PW3 * 2], const int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < PW3; ++q) {
                w[2 * q] = IG_LDW(p.W3, ((long)(PW3 * wid + q) * KS3 + ks) * 2);
                w[2 * q + 1] = IG_LDW(p.W3, ((long)(PW3 * wid + q) * KS3 + ks) * 2 + 1);
            }
        };
        ldw(wb[0], 0);
        ldw(wb[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS3; ++ks) {
            if (ks == KS1) __syncthreads();                // g2 is complete in region C (the stage-2 waves pass here after writing it)
            bf16x8 (&w)[PW3 * 2] = wb[ks & 1];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = rb * 16 + li;
                bf16x8 a;
                if (ks < KS1) a = IG_LDS(regA + r * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
                else a = IG_LDS(regC + r * (C4 * 2) + (ig_pchunk(r, (ks - KS1) * 4 + lg) << 4));
#pragma unroll
                for (int q = 0; q < PW3; ++q) {
                    a3[q][0][rb] = IG_MMA(w[2 * q], a, a3[q][0][rb]);
                    a3[q][1][rb] = IG_MMA(w[2 * q + 1], a, a3[q][1][rb]);
                }
            }
            if (ks + 2 < KS3) ldw(w, ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int q = 0; q < PW3; ++q) {
            float bv[8];
            ig_load8(p.b3 + (PW3 * wid + q) * 32 + lg * 8, bv);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                if (gro[rb] < 0) continue;
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)(a3[q][0][rb][e] + bv[e]); o[4 + e] = (bf16_t)(a3[q][1][rb][e] + bv[4 + e]); }
                IG_ST(o, p.R + (long)gro[rb] * CI + (PW3 * wid + q) * 32 + lg * 8);
            }
        }
    }
}


// ---- fused data-gradient backward, 64-row form (integ.hip's integ_bwd_kernel: same stages, same operand packs, same rounding points) ---------------------
//   stage 0  dR rows -> region A
//   stage 1  [dzf | dh2] = (dR B1^T) * g'([zf | h2])            pairs 4w .. 4w + 3
//   stage 2  dh1 = conv_t^T(dh2)                                 pair w (waves 0 .. 2)
//   stage 3  dxhat = [dzf | dh1] B3^T, LayerNorm backward on the accumulators -> dM' ; + I2T backward (dM = dM' + dY W4^T) ; + T2I backward (dp)
template <int CI, int C4>
__global__ __launch_bounds__(256, 2) void integ_bwd4_kernel(const IgBwdArgs p) {
    constexpr int DBG = 0;
    constexpr int BM = W4_BM, NW = W4_NW, RB = W4_RB;
    constexpr int CC = CI + C4;
    constexpr int CPR = CI / 8, LCH = CPR / 8, RPW = BM / NW, NPASS = RPW / 8;
    constexpr int KS1 = CI / 32, NP1 = CC / 32;
    constexpr int KT = C4 / 32, KS2 = 3 * KT, NP2 = C4 / 32;
    constexpr int KS3 = CC / 32, NP3 = CI / 32;
    constexpr int PW3 = NP3 / NW, PW1 = (NP1 + NW - 1) / NW;
    static_assert(NP3 % NW == 0 && NP2 <= NW && PW1 * (NW - 1) < NP1 && C4 % 32 == 0 && CI % 128 == 0 && RPW % 8 == 0, "geometry");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const regA = smem;                               // [BM][CI]  dR, then dzf, then xhat / dM'
    char* const regB = smem + BM * CI * 2;                 // [BM][C4]  dh2, then dY
    char* const regC = regB + BM * C4 * 2;                 // [BM][C4]  dh1
    float* const red = reinterpret_cast<float*>(regC + BM * C4 * 2);   // [NW][BM][2] partial row sums
    float* const tot = red + NW * BM * 2;                  // [BM][2] totals

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
    auto slot = [&](const int r, const int pr) __attribute__((always_inline)) { return reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pr * 4 + lg) ^ li) << 4)); };

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
    int gro[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) gro[rb] = grow_of(rb * 16 + li);
    __syncthreads();

    // ---------------- stage 1: [dzf | dh2] = (dR B1^T) * g'([zf | h2])
    {
        int pr1[PW1];
#pragma unroll
        for (int q = 0; q < PW1; ++q) pr1[q] = min(PW1 * wid + q, NP1 - 1);
        f32x4 acc[PW1 * 2][RB];
#pragma unroll
        for (int f = 0; f < PW1 * 2; ++f)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[f][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 wb[2][PW1 * 2];
        auto ldw = [&](bf16x8 (&w)[PW1 * 2], const int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < PW1; ++q) {
                w[2 * q] = IG_LDW(p.W1, ((long)pr1[q] * KS1 + ks) * 2);
                w[2 * q + 1] = IG_LDW(p.W1, ((long)pr1[q] * KS1 + ks) * 2 + 1);
            }
        };
        ldw(wb[0], 0);
        ldw(wb[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            bf16x8 (&w)[PW1 * 2] = wb[ks & 1];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bf16x8 a = IG_LDS(regA + (rb * 16 + li) * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
#pragma unroll
                for (int f = 0; f < PW1 * 2; ++f) acc[f][rb] = IG_MMA(w[f], a, acc[f][rb]);
            }
            if (ks + 2 < KS1) ldw(w, ks + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();                                   // every wave has read dR: region A becomes dzf
#pragma unroll
        for (int q = 0; q < PW1; ++q) {
            const int pr = PW1 * wid + q;
            if (pr >= NP1) continue;
            const int n0 = pr * 32 + lg * 8;
            bf16x8 zv[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
                zv[rb] = gro[rb] >= 0 ? *reinterpret_cast<const bf16x8*>(p.zfh2 + (long)gro[rb] * CC + n0) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = rb * 16 + li;
                bf16x8 dv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    dv[e] = (bf16_t)(acc[q * 2][rb][e] * qgelu_grad_t<bf16_t>((float)zv[rb][e]));
                    dv[4 + e] = (bf16_t)(acc[q * 2 + 1][rb][e] * qgelu_grad_t<bf16_t>((float)zv[rb][4 + e]));
                }
                if (pr < NP3) *reinterpret_cast<bf16x8*>(regA + r * (CI * 2) + (((pr * 4 + lg) ^ li) << 4)) = dv;
                else *reinterpret_cast<bf16x8*>(regB + r * (C4 * 2) + (ig_pchunk(r, (pr - NP3) * 4 + lg) << 4)) = dv;
                if (gro[rb] >= 0) {
                    if (pr < NP3) IG_ST(dv, p.dzfh2 + (long)gro[rb] * p.ldz + n0);
                    else IG_ST(dv, p.dh2 + (long)gro[rb] * p.ld2 + (n0 - CI));
                }
            }
        }
    }
    const bool s2 = wid < NP2;
    const int p2 = s2 ? wid : 0;
    bf16x8 w2[KS2][2];
    if (s2) {
#pragma unroll
        for (int ks = 0; ks < KS2; ++ks) { w2[ks][0] = IG_LDW(p.W2, ((long)p2 * KS2 + ks) * 2); w2[ks][1] = IG_LDW(p.W2, ((long)p2 * KS2 + ks) * 2 + 1); }
    }
    __syncthreads();                                       // dzf in region A, dh2 in region B

    // ---------------- stage 2: dh1[r] = sum_tap dh2[r - (tap - 1) TOK] W2[tap] -> region C
    if (s2) {
        f32x4 a2[2][RB];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) a2[q][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int r = rb * 16 + li;
                const int f = (r >> tsh) - (tap - 1);
                const bool ok = (unsigned)f < (unsigned)t;
                const int rs = ok ? r - (tap - 1) * TOK : r;
#pragma unroll
                for (int k3 = 0; k3 < KT; ++k3) {
                    bf16x8 a = IG_LDS(regB + rs * (C4 * 2) + (ig_pchunk(rs, k3 * 4 + lg) << 4));
                    if (!ok) a = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
                    a2[0][rb] = IG_MMA(w2[tap * KT + k3][0], a, a2[0][rb]);
                    a2[1][rb] = IG_MMA(w2[tap * KT + k3][1], a, a2[1][rb]);
                }
            }
        }
        const int n0 = p2 * 32 + lg * 8;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = rb * 16 + li;
            bf16x8 hb;
#pragma unroll
            for (int e = 0; e < 4; ++e) { hb[e] = (bf16_t)a2[0][rb][e]; hb[4 + e] = (bf16_t)a2[1][rb][e]; }
            *reinterpret_cast<bf16x8*>(regC + r * (C4 * 2) + (ig_pchunk(r, p2 * 4 + lg) << 4)) = hb;
            if (gro[rb] >= 0) IG_ST(hb, p.dh1 + (long)gro[rb] * p.ld1 + n0);
        }
    }

    // ---------------- stage 3: dxhat = [dzf | dh1] B3^T (pairs 3w .. 3w + 2), then the LayerNorm backward on the accumulators
    {
        f32x4 a3[PW3][2][RB];
#pragma unroll
        for (int q = 0; q < PW3; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) a3[q][h][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            bf16x8 wb[2][PW3 * 2];
            auto ldw = [&](bf16x8 (&w)[PW3 * 2], const int ks) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < PW3; ++q) {
                    w[2 * q] = IG_LDW(p.W3, ((long)(PW3 * wid + q) * KS3 + ks) * 2);
                    w[2 * q + 1] = IG_LDW(p.W3, ((long)(PW3 * wid + q) * KS3 + ks) * 2 + 1);
                }
            };
            ldw(wb[0], 0);
            ldw(wb[1], 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS3; ++ks) {
                if (ks == KS1) __syncthreads();            // dh1 is complete in region C
                bf16x8 (&w)[PW3 * 2] = wb[ks & 1];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int r = rb * 16 + li;
                    bf16x8 a;
                    if (ks < KS1) a = IG_LDS(regA + r * (CI * 2) + (((ks * 4 + lg) ^ li) << 4));
                    else a = IG_LDS(regC + r * (C4 * 2) + (ig_pchunk(r, (ks - KS1) * 4 + lg) << 4));
#pragma unroll
                    for (int q = 0; q < PW3; ++q) {
                        a3[q][0][rb] = IG_MMA(w[2 * q], a, a3[q][0][rb]);
                        a3[q][1][rb] = IG_MMA(w[2 * q + 1], a, a3[q][1][rb]);
                    }
                }
                if (ks + 2 < KS3) ldw(w, ks + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const bool i2tb = p.dY != nullptr, t2ib = p.dp != nullptr;      // (kernel arguments: the same for every workgroup)
        if (i2tb) {
            // dY rows of this tile -> region B (free since the barrier inside the K loop: stage 2 was its last reader) and to memory; visible after the
            // barriers of the LayerNorm backward below
            const int lq4 = lane & 3, lrow = lane >> 2;
            const int r = wid * 16 + lrow, f = r >> tsh, j = (grp << tsh) + (r & tokmask);
            const bool ok = j >= 1 && j < L;
            const int N = L - 1, T = 2 * t;
            const bf16_t* s0 = p.dXn + ((long)(clip * T + 2 * f) * N + (ok ? j - 1 : 0)) * C4;
            const bf16_t* s1 = s0 + (long)N * C4;
#pragma unroll
            for (int m = 0; m < C4 / 32; ++m) {
                const int c = lq4 + 4 * m;
                const bf16x8 u = *reinterpret_cast<const bf16x8*>(s0 + c * 8), v = *reinterpret_cast<const bf16x8*>(s1 + c * 8);
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = ok ? (bf16_t)((float)u[e] + (float)v[e]) : (bf16_t)0.f;
                *reinterpret_cast<bf16x8*>(regB + r * (C4 * 2) + (ig_pchunk(r, c) << 4)) = o;
                if (ok) IG_ST(o, p.dY + ((long)(clip * t + f) * N + (j - 1)) * C4 + c * 8);
            }
        }
        // xhat at this lane's (row, 8 columns) positions; partial row sums S1 = sum dxhat, S2 = sum dxhat xhat.  The xhat values wait in region A (dead once every
        // wave has left the K loop - hence the barrier), each lane's in the slot its dM' piece of the same (row, pair) takes afterwards
        __syncthreads();
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int r = rb * 16 + li;
            float s1 = 0.f, s2v = 0.f;
            const long base = (long)max(gro[rb], 0) * CI;
#pragma unroll
            for (int q = 0; q < PW3; ++q) {
                const bf16x8 xh = *reinterpret_cast<const bf16x8*>(p.Xh + base + (PW3 * wid + q) * 32 + lg * 8);
                *slot(r, PW3 * wid + q) = xh;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1 += a3[q][0][rb][e] + a3[q][1][rb][e];
                    s2v += a3[q][0][rb][e] * (float)xh[e] + a3[q][1][rb][e] * (float)xh[4 + e];
                }
            }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2v += __shfl_xor(s2v, 16, 64); s2v += __shfl_xor(s2v, 32, 64);
            if (lg == 0) reinterpret_cast<float2*>(red)[wid * BM + r] = make_float2(s1, s2v);
        }
        __syncthreads();
        if (tid < BM) {
            float a1 = 0.f, a2s = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { const float2 v = reinterpret_cast<const float2*>(red)[w * BM + tid]; a1 += v.x; a2s += v.y; }
            reinterpret_cast<float2*>(tot)[tid] = make_float2(a1, a2s);
        }
        __syncthreads();
        constexpr float invC = 1.f / (float)CI;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int gr = gro[rb];
            if (gr < 0) continue;
            const int r = rb * 16 + li;
            const float2 cs = reinterpret_cast<const float2*>(tot)[r];
            const float rs = p.rstd[gr], m1 = cs.x * invC, m2 = cs.y * invC;
            const bool to_dm = p.dM && !i2tb && (!p.dm_cls || ((grp << tsh) + (r & tokmask)) == 0);
            const bool cls_row = p.dcls && grp == 0 && (r & tokmask) == 0;
#pragma unroll
            for (int q = 0; q < PW3; ++q) {
                const int c0 = (PW3 * wid + q) * 32 + lg * 8;
                const bf16x8 xr = *slot(r, PW3 * wid + q);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = rs * (a3[q][0][rb][e] - m1 - (float)xr[e] * m2);
                    v[4 + e] = rs * (a3[q][1][rb][e] - m1 - (float)xr[4 + e] * m2);
                }
                if (p.add_dR) {
                    const bf16x8 d = *reinterpret_cast<const bf16x8*>(p.dR + (long)gr * CI + c0);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += (float)d[e];
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16_t)v[e];
                IG_ST(o, p.dMp + (long)gr * CI + c0);
                if (to_dm) IG_ST(o, p.dM + (long)gr * CI + c0);
                if (t2ib) *slot(r, PW3 * wid + q) = o;    // dM' tile: the B operand of the T2I product at the end
                if (cls_row) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) atomicAdd(p.dcls + (long)(r >> tsh) * CI + c0 + e, (float)o[e]);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { a3[q][0][rb][e] = (float)o[e]; a3[q][1][rb][e] = (float)o[4 + e]; }       // (the I2T term below accumulates on the stored dM')
            }
        }
        if (i2tb) {
            // dM = dM' + dY W4^T (K = C4): the accumulators hold dM' as stored; cls rows and slots beyond L have all-zero dY rows
            constexpr int KS4 = C4 / 32;
#pragma unroll
            for (int ks = 0; ks < KS4; ++ks) {
                bf16x8 w4[PW3 * 2];
#pragma unroll
                for (int q = 0; q < PW3; ++q) {
                    w4[2 * q] = IG_LDW(p.W4, ((long)(PW3 * wid + q) * KS4 + ks) * 2);
                    w4[2 * q + 1] = IG_LDW(p.W4, ((long)(PW3 * wid + q) * KS4 + ks) * 2 + 1);
                }
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const int r = rb * 16 + li;
                    const bf16x8 a = IG_LDS(regB + r * (C4 * 2) + (ig_pchunk(r, ks * 4 + lg) << 4));
#pragma unroll
                    for (int q = 0; q < PW3; ++q) {
                        a3[q][0][rb] = IG_MMA(w4[2 * q], a, a3[q][0][rb]);
                        a3[q][1][rb] = IG_MMA(w4[2 * q + 1], a, a3[q][1][rb]);
                    }
                }
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const int gr = gro[rb];
                if (gr < 0) continue;
#pragma unroll
                for (int q = 0; q < PW3; ++q) {
                    bf16x8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = (bf16_t)a3[q][0][rb][e]; o[4 + e] = (bf16_t)a3[q][1][rb][e]; }
                    IG_ST(o, p.dM + (long)gr * CI + (PW3 * wid + q) * 32 + lg * 8);
                }
            }
        }
        if (t2ib) {
            // ---------------- T2I backward: dp = (dX_next + dM' W5^T) g'(p) on the two temporal frames of every patch row of the tile.  N = 2 C4 columns = 6 pairs,
            // 12 items (pair, half of the 16-row blocks), three per wave; B operand = the dM' tile in region A.
            __syncthreads();
            constexpr int NP5 = 2 * C4 / 32, RBQ = RB / 2;
            static_assert(NP5 * 2 == 3 * NW, "three (pair, half) items per wave");
            const int N = L - 1, T = 2 * t;
#pragma unroll 1
            for (int it = 0; it < 3; ++it) {
                const int id = wid * 3 + it, p5 = id >> 1, qr = id & 1;
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

template <int MODE>
int fwd4_launch_v(const IgArgs& a, hipStream_t s) {
    constexpr int CI = 384, C4 = 96;
    const int smem = W4_BM * CI * 2 + 2 * W4_BM * C4 * 2 + (W4_NW + 1) * W4_BM * 2 * 4;
    static DistSmemOnce attr;
    RUN_(dist_max_smem(attr, (const void*)integ_fwd4_kernel<CI, C4, MODE>, (size_t)smem));
    hipLaunchKernelGGL((integ_fwd4_kernel<CI, C4, MODE>), dim3((unsigned)(a.clips * a.groups)), dim3(256), smem, s, a);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}

}  // namespace

namespace dist_integ {

// the T2I-in-front forms (training with xhat, inference) at Ci = 384, C4 = 96 and t | 64; `a.groups` / `a.tokshift` are already those of 64-row tiles
int integ_fwd4_launch(const IgArgs& a, const int mode, hipStream_t s) {
    if (!a.Xp || (mode != 0 && mode != 2)) return 0;
    return mode == 2 ? fwd4_launch_v<2>(a, s) : fwd4_launch_v<0>(a, s);
}

// `a.groups` / `a.tokshift` are those of 64-row tiles
int integ_bwd4_launch(const IgBwdArgs& a, hipStream_t s) {
    constexpr int CI = 384, C4 = 96;
    const int smem = W4_BM * CI * 2 + 2 * W4_BM * C4 * 2 + (W4_NW + 1) * W4_BM * 2 * 4;
    static DistSmemOnce attr;
    RUN_(dist_max_smem(attr, (const void*)integ_bwd4_kernel<CI, C4>, (size_t)smem));
    hipLaunchKernelGGL((integ_bwd4_kernel<CI, C4>), dim3((unsigned)(a.clips * a.groups)), dim3(256), smem, s, a);
    HIP_CHECK_RET(hipGetLastError());
    return 1;
}

}  // namespace dist_integ
