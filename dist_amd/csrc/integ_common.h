// Shared by the fused IntegrationNetwork kernels (integ.hip: 8 waves per 128-row tile, one workgroup per CU; measure/integ4.hip, timing-only library:
// 4 waves per 64-row tile, two workgroups per CU): kernel argument blocks, fragment helpers, the timing-only ablation macros.  Not part of the C ABI.
#pragma once
#include "common.h"

namespace dist_integ {


struct IgArgs {
    const bf16_t* Mp;
    const bf16_t *W1, *W2, *W3;
    const float *b1, *b2, *b3;
    const float *ga, *ba, *gb, *bb;
    bf16_t *R, *Na, *Nb, *Xh, *zfh2, *hfg2, *h1;
    float *mean, *rstd;
    int clips, t, L, groups, tokshift;
    float eps;
    // T2I in front (template T2I): M' = M + [cls_token ; conv_strided(X')] is formed here instead of being read
    const bf16_t *M, *Xp, *Wt; const float *bt, *cls; bf16_t* Mpo;
    // ... and I2T behind it (dist.py:90-105): X_next[frames 2f, 2f+1; position j-1] = X' + (M[f, j] Wi^T + bi), when Xn is given
    const bf16_t* Wi; const float* bi; bf16_t* Xn;
};

DEV int ig_pchunk(const int row, const int c) { return (c & ~3) | ((c & 3) ^ (((row >> 2) & 1) << 1)); }
// sum over the 8 lanes that share a row in stage 0 (lanes 8 k .. 8 k + 7)
DEV float ig_sum8(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));   // row_half_mirror
    return v;
}
DEV f32x4 ig_mma(const bf16x8& w, const bf16x8& x, const f32x4& acc) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0); }
DEV bf16x8 ig_ldw(const bf16_t* base, const long frag, const int lane) { return *reinterpret_cast<const bf16x8*>(base + (frag * 64 + lane) * 8); }
#define IG_LDW(base, frag) ((DBG & 1) ? bf16x8{(bf16_t)(float)lane, 0, 0, 0, 0, 0, 0, 0} : ig_ldw(base, frag, lane))
#define IG_MMA(w, x, acc) ((DBG & 2) ? (acc) : ig_mma(w, x, acc))
#define IG_GELU(x) ((DBG & 4) ? (x) : qgelu_t<bf16_t>(x))
#define IG_LDS(ptr) ((DBG & 16) ? bf16x8{(bf16_t)(float)li, 0, 0, 0, 0, 0, 0, 0} : *reinterpret_cast<const bf16x8*>(ptr))
#define IG_ST(v, ptr) do { if (!(DBG & 8)) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(ptr)); } while (0)
DEV void ig_load8(const float* p, float (&o)[8]) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}


struct IgBwdArgs {
    const bf16_t *dR, *zfh2, *Xh; const float* rstd;
    const bf16_t *W1, *W2, *W3;
    bf16_t *dzfh2, *dh2, *dh1, *dMp, *dM;
    int ldz, ld2, ld1;       // row pitches of dzf, dh2 (a pointer to ITS first column), dh1
    int add_dR, dm_cls;      // dm_cls: the second copy of dM' only receives the cls rows (token 0)
    // I2T backward behind the LayerNorm backward (dist.py:100-105 through autograd): dY = dX_next[2f] + dX_next[2f+1] (written out: the I2T weight gradient reads it),
    // dM = dM' + dY Wi on the patch rows - dM then holds the WHOLE gradient w.r.t. M, not a copy of dM'
    const bf16_t *dXn, *W4; bf16_t* dY;
    // T2I backward behind that (dist.py:81-86 through autograd, and through X' = g(p)): dp[2f+a][j-1] = (dX_next[2f+a][j-1] + dM'[f][j] W5_a^T) g'(p[2f+a][j-1])
    const bf16_t *W5, *pact; bf16_t* dp;
    float* dcls;             // optional: the T2I cls-token gradient [t][Ci] += dM' of the cls rows (token 0 of every frame), fp32 atomics
    int clips, t, L, groups, tokshift;
};

// measure/integ4.hip (timing-only library): the 64-row / 4-wave forms (1 = launched, 0 = not this form's call, < 0 error)
int integ_fwd4_launch(const IgArgs& a, int mode, hipStream_t s);
int integ_bwd4_launch(const IgBwdArgs& a, hipStream_t s);

}  // namespace dist_integ
using namespace dist_integ;
