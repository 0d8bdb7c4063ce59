/* dist_amd.h — C ABI of the MI355X-native DiST training hot path.
 *
 * The reference (alibaba-mmai-research/DiST) is pure Python/PyTorch and has no FFI
 * (SURVEY.md §8(b)); its operator interface for this path is
 *   models/base/backbone.py:228-251   ClipVisionTextTransformer.forward
 *   models/base/clip.py:263-300       VisionTransformer.forward   (frozen ViT)
 *   models/base/clip.py:482-533       CLIP.forward_with_text      (cosine logits)
 *   models/module_zoo/branches/dist.py:222-247  DiSTNetwork.forward
 *   models/utils/losses.py:20-31      SoftTargetCrossEntropy
 *   models/utils/optimizer.py:67-73,138-186     AdamW over the dist_net groups
 * Each entry point below names the reference interface it replaces.  INTEGRATION.md
 * shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success or a negative error code
 * (dist_strerror()); no C++ exception crosses the boundary; all device buffers are
 * caller-allocated and caller-owned (the library never allocates device memory);
 * every call is asynchronous on the hipStream_t passed as `void* stream`; one
 * handle per (device, stream), not thread-safe.  Device pointers only, no torch types.
 */
#ifndef DIST_AMD_H
#define DIST_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { DIST_F32 = 0, DIST_BF16 = 1 };

enum {
    DIST_OK = 0,
    DIST_ERR_ARG = -1,        /* invalid argument / unsupported shape */
    DIST_ERR_STATE = -2,      /* call order (e.g. backward before forward) */
    DIST_ERR_WORKSPACE = -3,  /* workspace too small / not bound */
    DIST_ERR_UNBOUND = -4     /* a weight or buffer was not bound */
    /* <= -1000: -(hipError_t) - 1000 */
};

const char* dist_strerror(int code);
/* 9 for this header (8: dist_ln_bwd_args.partial / partial_elems; 9: the dist_integ_args / dist_integ_bwd_args fields of the fused T2I / I2T stages, which round 3
 * added under 8, and dist_measure_build).  Bumped on EVERY layout change of a struct below; a binding compares it, and dist_abi_sizeof() of each
 * struct it mirrors, before the first dist_create (dist_amd/lib.py does). */
#define DIST_ABI_VERSION 9
int dist_abi_version(void);
/* 1 when the library was compiled with -DDIST_AMD_MEASURE: only then do the timing-only environment knobs whose results are wrong
 * (DIST_AMD_SKIP, DIST_AMD_DUMMY*, DIST_AMD_TN_SKIP_REDUCE, DIST_AMD_*_DBG, DIST_AMD_TNET_BWD_NOREDUCE) exist.  The shipped build returns 0. */
int dist_measure_build(void);
/* sizeof() of an argument struct of this header by name ("dist_gemm_args", "dist_gemm_tn_args", "dist_ln_args",
 * "dist_ln_bwd_args", "dist_adamw_seg", "dist_config", "dist_rowmap", "dist_outmap", "dist_tnet_args", "dist_tnet_bwd_args", "dist_integ_args", "dist_integ_pack_args", "dist_integ_unfold_args", "dist_integ_bwd_args"); -1 for an unknown name.
 * Lets a foreign-language binding verify its mirror of the layout without a GPU. */
int dist_abi_sizeof(const char* struct_name);

/* ---------------------------------------------------------------------------------------
 * Operator level (also what the unit parity tests drive).
 * ------------------------------------------------------------------------------------- */

/* logical-row -> source-row maps: conv taps and token re-layouts without im2col */
enum { DIST_RM_PLAIN = 0, DIST_RM_SHIFT = 1, DIST_RM_SPATIAL = 2, DIST_RM_STRIDED = 3, DIST_RM_SKIPCLS = 4 };
typedef struct dist_rowmap { int mode, p0, p1, sign; } dist_rowmap;
enum { DIST_OM_PLAIN = 0, DIST_OM_DUP = 1, DIST_OM_INSERTCLS = 2, DIST_OM_SPLITCOLS = 3, DIST_OM_HEADS = 4 };
typedef struct dist_outmap { int mode, p0, p1, p2; } dist_outmap;

enum {
    DIST_EPI_BIAS = 1,   /* v += bias[n] */
    DIST_EPI_MULG = 2,   /* v *= quickgelu'(aux[dest][n])   (backward through an activation) */
    DIST_EPI_RES = 4,    /* v += res[dest][n] */
    DIST_EPI_ACT2 = 8,   /* C2[dest][n] = quickgelu(v)  (C, if non-null, keeps the pre-activation) */
    DIST_EPI_MULG_POST = 16, /* with MULG: the derivative factor is applied AFTER bias and residual: v = (acc + bias + res) * quickgelu'(aux) */
    DIST_EPI_LNFOLD = 32,    /* the GEMM consumes the RAW rows x of a LayerNorm-then-Linear pair (clip.py:160-176: ln_1 -> attn
                              * in_proj, ln_2 -> mlp.c_fc) and normalises in the epilogue:
                              *   v = rstd[m] * (acc - mean[m] * colsum[n]) + bias[n]
                              * with B = W * diag(gamma) (bf16), colsum[n] = sum_k B[n][k], bias[n] = b[n] + sum_k W[n][k] beta[k]
                              * (dist_op_ln_fold prepares all three).  `aux` carries the row statistics as fp32 [2][M] (mean, then
                              * rstd: dist_op_layernorm with y = NULL), `bias2` carries colsum.  Large plain bf16 GEMMs only
                              * (the 256x256 LDS-DMA kernel); not combinable with MULG. */
    DIST_EPI_ROWSTATS = 64,  /* producer side of the same fold: besides C, the GEMM stores the row sums of its STORED (rounded)
                              * values and of their squares over every 64-column slice, rowstats[slice][m] = (S, Q) fp32, slice =
                              * column / 64 - plain stores in a fixed order (no atomics: results do not depend on the batch a row
                              * sits in).  dist_op_ln_stats_from_partials turns them into the [2][M] mean / rstd a DIST_EPI_LNFOLD
                              * GEMM consumes, so the residual stream is never re-read for its LayerNorm statistics
                              * (clip.py:160-176: x -> ln_2 -> c_fc, x -> ln_1 -> in_proj of the next block).  Large plain bf16
                              * GEMMs with a plain output map, C != NULL. */
    DIST_EPI_FP8 = 128,      /* BASELINE config 5 (fp8 frozen spatial branch): A [M][K] and B [N][K] hold OCP e4m3 bytes (lda / ldb in
                              * elements = bytes, multiples of 16; K a multiple of 128, >= 256), a_scale[m] / b_scale[n] are their fp32
                              * per-row scales (dist_op_quant_rows_fp8), the product runs on the block-scaled fp8 MFMA of CDNA4
                              * (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales) at twice the bf16 rate, and
                              *   acc[m][n] = a_scale[m] * b_scale[n] * sum_k A[m][k] * B[n][k]   (fp32)
                              * enters the same epilogue as the bf16 kernel (bias, LayerNorm fold, residual, QuickGELU, head-major /
                              * insert-cls outputs, row statistics); C / C2 / res are bf16 (dtype = DIST_BF16).  Large plain GEMMs
                              * only (the 256x256 LDS-DMA kernel; M >= 1024, N % 64 == 0). */
    DIST_EPI_FP8_ASCALAR = 256, /* with DIST_EPI_FP8: A carries ONE scale for all rows, a_scale[0] (an operand a producer wrote with DIST_EPI_OUT8) */
    DIST_EPI_OUT8 = 512      /* the producer side of the fp8 operands: besides C / C2 (either may then be NULL), the stored value v (after bias,
                              * residual and - when C is NULL with DIST_EPI_ACT2 - the QuickGELU) also leaves as OCP e4m3,
                              *   C8[m][n] = e4m3_rne(clamp(float(bf16(v)) / out8_scale[0], -448, 448)),   C8 bytes [M][ldc8], ldc8 % 16 == 0,
                              * with a per-TENSOR scale the caller read from an earlier pass (e4m3 is a floating format: 14 binades of normal
                              * range make a per-tensor scale with a margin as precise as a per-row one), and *out8_amax (optional, fp32 >= 0 as
                              * its bit pattern) = max(*out8_amax, max |float(bf16(v))|) for the next pass's scale (dist_op_fp8_scale_update).
                              * Plain output map, or DIST_OM_HEADS (ldc8 = 64: the q | k | v image dist_op_attention_fp8 reads); 256x256 LDS-DMA
                              * kernel only. */
};

/* C[omap(m)][n] = epi( sum_tap sum_k A[amap(m,tap)][k] * B[n][tap*K + k] )
 * replaces nn.Linear / nn.Conv3d (kernel (kt,1,1), (1,3,3), strided (a,1,1)) / Conv2d patch
 * embedding call sites: clip.py:116-123,155-161,232; dist.py:23-36,54-60,75,97,178-198. */
typedef struct dist_gemm_args {
    const void* A; const void* B; void* C; void* C2;
    const float* bias; const void* res; const void* aux;
    int64_t M; int N; int K; int taps;
    int lda, ldb, ldc, ldc2, ldres, ldaux;
    dist_rowmap amap; dist_outmap omap;
    int flags; int dtype;
    const float* bias2;   /* optional second bias, added with `bias` (two Linears evaluated as one GEMM over side-by-side inputs) */
    float* rowstats;      /* DIST_EPI_ROWSTATS: fp32 [N / 64][M][2] */
    const float* a_scale; /* DIST_EPI_FP8: fp32 [M] ([1] with DIST_EPI_FP8_ASCALAR) */
    const float* b_scale; /* DIST_EPI_FP8: fp32 [N] */
    void* C8; int ldc8;   /* DIST_EPI_OUT8: e4m3 image of the output */
    const float* out8_scale; float* out8_amax;
} dist_gemm_args;
int dist_op_gemm_nt(const dist_gemm_args* a, void* stream);
/* per-row symmetric quantisation to OCP e4m3 (the operands of a DIST_EPI_FP8 GEMM; torch: (x.float() * (448 / amax)).to(float8_e4m3fn)):
 *   amax = max_k |x[r][k]| (fp32), scale[r] = amax / 448 (1 when the row is all zero), q[r][k] = e4m3_rne(float(x[r][k]) * (448 / amax)).
 * x [rows][ld] of `dtype` (bf16 / fp32), K % 8 == 0, K <= 8192, ld % 8 == 0; q [rows][ldq] bytes, ldq % 8 == 0 (% 16 for a GEMM operand).  Used once per frozen weight at
 * pack time (per output channel) and per GEMM input row at run time (per token). */
int dist_op_quant_rows_fp8(const void* x, int dtype, int64_t rows, int K, int ld, void* q, int ldq, float* scale, void* stream);
/* per-tensor scales for DIST_EPI_OUT8 from collected maxima: for i < n: scale[i] = 2^ceil(log2(max(amax[i], 2^-24) * margin / 448)),
 * then amax[i] = 0 (ready to collect the next pass).  dist_op_amax: *amax = max(*amax, max |x|) over a bf16 / fp32 tensor (the
 * calibration pass of a tensor no DIST_EPI_OUT8 producer has written yet). */
int dist_op_fp8_scale_update(float* amax, float* scale, int n, float margin, void* stream);
int dist_op_amax(const void* x, int dtype, int64_t n, float* amax, void* stream);
/* out[r] = scale[r] * sum_k e4m3(q[r][k]) in fp32 (fixed order): the column sums a DIST_EPI_LNFOLD GEMM on e4m3 weights needs */
int dist_op_fp8_rowsum(const void* q, const float* scale, int64_t rows, int K, int ldq, float* out, void* stream);
/* prepares a LayerNorm-then-Linear pair for DIST_EPI_LNFOLD: Wp[n][k] = bf16(W[n][k] * gamma[k]) (overwrites the packed
 * forward-layout copy of W), colsum[n] = sum_k float(Wp[n][k]), bias_out[n] = bias[n] + sum_k W[n][k] * beta[k] */
int dist_op_ln_fold(const float* W, const float* bias, const float* gamma, const float* beta, void* Wp, float* colsum, float* bias_out,
                    int N, int K, void* stream);

/* out[i*so_i + tap*so_tap + (c/inner)*so_outer + c%inner] += sum_m A[amap(m)][i] * B[bmap(m,tap)][c]
 * (fp32 atomics; weight gradients of every Linear / Conv3d on the path, autograd in the reference) */
typedef struct dist_gemm_tn_args {
    const void* A; const void* B; float* out;
    int64_t M; int NI; int K; int taps;
    int lda, ldb;
    dist_rowmap amap; dist_rowmap bmap;
    int64_t so_i, so_tap, so_outer; int inner;
    int dtype; int use_tr;   /* use_tr: bf16 LDS transpose reads (ds_read_b64_tr_b16) */
    float* colsum;           /* optional: colsum[i] += sum_m A[amap(m)][i] (the bias gradient), fused into the same pass */
    float* partial;          /* optional scratch (fp32): when large enough, row-split partial tiles are stored here with plain
                                coalesced stores and summed by a second small kernel instead of per-block atomics */
    int64_t partial_elems;
    /* optional second destination (two weights that share the activation-gradient operand, e.g. a Linear pair whose inputs
     * are stored side by side): columns c >= split_c go to out2[i*so_i2 + (c - split_c)], colsum is also added to colsum2.
     * Plain layouts only (taps == 1, inner == 1, so_outer == 1). */
    int split_c; float* out2; int64_t so_i2; float* colsum2;
    /* ABI 9: upper bound of the workgroups of the main kernel (0 = the library's default, 96: a launch that runs BESIDE a critical chain should not hold
     * every CU while it waits for HBM; a caller with nothing else in flight passes 256: one block per CU is the fastest form of a lone launch) */
    int max_blocks;
} dist_gemm_tn_args;
int dist_op_gemm_tn(const dist_gemm_tn_args* a, void* stream);

/* LayerNorm over the last axis (clip.py:181-187), optional periodic addend before the
 * norm (cls/positional embedding, clip.py:274-276) and optional second affine output
 * that shares the statistics (IntegrationNetwork.ln / ln_temporal, dist.py:35-36,43-45). */
typedef struct dist_ln_args {
    const void* x; void* y; void* y2;
    const float* w; const float* b; const float* w2; const float* b2;
    const float* addend; int addend_period;        /* x[r] += addend[r % period] (fp32 table) */
    float* mean; float* rstd;                       /* optional, saved for backward */
    int64_t rows; int C; int dtype; float eps;
} dist_ln_args;
int dist_op_layernorm(const dist_ln_args* a, void* stream);
/* mean[m] = S / C, rstd[m] = rsqrt(Q / C - mean^2 + eps) with (S, Q) = sum over `slices` of part[slice][m] (fp32 [slices][rows][2],
 * left by a DIST_EPI_ROWSTATS GEMM over a C = 64 * slices wide tensor): the statistics-only form of dist_op_layernorm (y = NULL)
 * without reading the tensor.  Slices are added in index order. */
int dist_op_ln_stats_from_partials(const float* part, int slices, int64_t rows, int C, float eps, float* mean, float* rstd, void* stream);

/* dx (+)= LN'(dy [, dy2]); dw/db (+= atomics), all optional except x/mean/rstd */
typedef struct dist_ln_bwd_args {
    const void* x; const float* mean; const float* rstd;
    const void* dy; const float* w; const void* dy2; const float* w2;
    void* dx; int accumulate_dx;
    float* dw; float* db; float* dw2; float* db2;
    int64_t rows; int C; int dtype;
    const void* dx_add;      /* optional: dx = dx_add + LN'(...) (out-of-place accumulate; overrides accumulate_dx) */
    void* dx_copy;           /* optional: second copy of dx */
    float* partial;          /* optional fp32 scratch (ABI 8): when it holds >= dist_op_layernorm_bwd_scratch(rows, C) elements the per-block parameter-
                              * gradient sums are stored there with plain stores and added up by a second small launch in a fixed order (bit-repeatable,
                              * no same-address atomics at the end of every block); NULL / too small: fp32 atomics as before */
    int64_t partial_elems;
} dist_ln_bwd_args;
int64_t dist_op_layernorm_bwd_scratch(int64_t rows, int C);
int dist_op_layernorm_bwd(const dist_ln_bwd_args* a, void* stream);

/* per-frame multi-head self-attention -> [frames*L, d]
 * (nn.MultiheadAttention inside ResidualAttentionBlockMid, clip.py:155,166-168).
 * qkv_layout DIST_QKV_ROWS:  packed rows [frames*L, 3*d] (the in_proj output as torch lays it out);
 *            DIST_QKV_HEADS: [frame][head][q|k|v][L][64], what dist_op_gemm_nt writes with omap DIST_OM_HEADS - every
 *            (frame, head) operand is one contiguous 25 KB block instead of 128-byte pieces at a 3*d row stride. */
enum { DIST_QKV_ROWS = 0, DIST_QKV_HEADS = 1 };
int dist_op_attention(const void* qkv, void* out, int frames, int L, int heads, int qkv_layout, int dtype, void* stream);
/* the same attention (bf16 qkv) with its output written as OCP e4m3 INSTEAD of bf16 - the A operand of a DIST_EPI_FP8 out-projection with
 * DIST_EPI_FP8_ASCALAR: out8[row][c] = e4m3_rne(clamp(float(bf16(o)) / out8_scale[0], -448, 448)), out8 bytes [frames*L][heads*64];
 * *out8_amax (optional) = running maximum of |bf16(o)| for the next pass's scale (see DIST_EPI_OUT8). */
int dist_op_attention_out8(const void* qkv, void* out8, const float* out8_scale, float* out8_amax, int frames, int L, int heads, int qkv_layout,
                           void* stream);
/* ... and with q | k | v arriving as e4m3 too: qkv8 = the head-major DIST_EPI_OUT8 image of the in_proj output with ONE scale in_scale[0]
 * (bytes [frame][head][q|k|v][L][64]).  The bytes are widened to bf16 on their way into LDS (exact), the products stay bf16 MFMAs, the
 * scale enters the scores as in_scale^2 and the output as in_scale.  Exactly one of out (bf16 [frames*L][heads*64]) / out8 (as above). */
int dist_op_attention_fp8(const void* qkv8, const float* in_scale, void* out, void* out8, const float* out8_scale, float* out8_amax,
                          int frames, int L, int heads, void* stream);

/* Fused TemporalNet forward (dist.py:48-65): X' = g(X + conv_{1x3x3}(g(conv_{3x1x1}(LN_C(X))))) on the channels-last temporal map
 * X [clips*T*G*G][Ct] (rows (b*T + k)*G*G + n) in ONE launch - LayerNorm, the temporal taps, QuickGELU, the nine spatial taps, the
 * residual and the second QuickGELU; the normalised and the activated tensor live in LDS only.  bf16; Ct in {32, 64, 96}; G*G <= 256;
 * tk in {1, 3, 5}.  W1 = c_fc1.weight packed as [Ct][tk*Ct] (W1[n][tap*Ct + c] = weight[n][c][tap]), W2 = c_fc2.weight packed as
 * [Ct][9*Ct] (tap = 3*ky + kx) - the forward layouts dist_pack_weights produces.  Outputs (same shape as X):
 *   z  = bf16(conv_t(LN(X)) + b1)            (pre-activation of the first QuickGELU; backward needs it)
 *   p  = bf16(X + conv_s(g(z)) + b2)         (pre-activation of the second)
 *   Xp = bf16(g(p))                          = TemporalNet(X)
 *   U  = bf16(LN(X)), V = bf16(g(z))         optional (both or neither; NULL: they never reach memory)
 *   mean, rstd: fp32 [rows] LayerNorm statistics.
 * Rounding points are those of the unfused sequence dist_op_layernorm -> dist_op_gemm_nt (ACT2) -> dist_op_gemm_nt (RES | ACT2). */
typedef struct dist_tnet_args {
    const void* X; const void* W1; const void* W2;
    const float* b1; const float* b2; const float* ln_w; const float* ln_b;
    void* z; void* p; void* Xp; void* U; void* V;
    float* mean; float* rstd;
    int clips, T, G, Ct, tk; int dtype; float eps;
} dist_tnet_args;
int dist_op_temporal_net_fwd(const dist_tnet_args* a, void* stream);
/* Data-gradient backward of the same block (autograd in the reference, runs/train.py:110), two fused launches + a small reduction:
 *   dz = conv_{1x3x3}^T(dp) * g'(z)                (dp = dL/dp: the incoming gradient already multiplied by g'(p); nine flipped taps over the
 *                                                   frame in LDS, W2b = c_fc2.weight in the data-gradient layout [Ct][9*Ct], W2b[ci][tap*Ct + co])
 *   dX = dp + bf16(LN'(conv_{3x1x1}^T(dz)))        (W1b [Ct][tk*Ct] likewise; LayerNorm backward on the accumulators, dp added as whole rows)
 *   dgamma += sum_rows dU * xhat, dbeta += sum_rows dU   (per-workgroup partial rows in `scratch`, summed in a fixed order: no atomics)
 * dz is an output (the weight-gradient GEMM dW1 = dz^T U needs it) and is re-read by the second launch.  The weight gradients themselves
 * (dW1, dW2, db1 = colsum dz, db2 = colsum dp) stay with dist_op_gemm_tn.  scratch: fp32, >= dist_op_temporal_net_bwd_scratch() elements. */
typedef struct dist_tnet_bwd_args {
    const void* dp; const void* z; const void* X; const float* mean; const float* rstd; const float* ln_w;
    const void* W1b; const void* W2b;
    void* dz; void* dX; float* dgamma; float* dbeta;
    float* scratch; int64_t scratch_elems;
    int clips, T, G, Ct, tk; int dtype;
    int phase;               /* 0 = everything; 1 = only dz (first launch); 2 = only dX, dgamma, dbeta from the dz of an earlier phase-1 call
                              * (lets the caller start the weight-gradient GEMMs that read dz on another stream in between); 3 = as 2, but the
                              * parameter-gradient partials stay in `scratch` for a later dist_op_temporal_net_bwd_reduce over several layers */
} dist_tnet_bwd_args;
int64_t dist_op_temporal_net_bwd_scratch(int clips, int T, int Ct);
int dist_op_temporal_net_bwd(const dist_tnet_bwd_args* a, void* stream);
/* dgamma[l][c] += / dbeta[l][c] += the partial rows that phase-3 calls of `layers` (<= 32) layers left at scratch + l * layer_stride: ONE launch
 * for all of them (every launch on a serial chain costs ~10 us of step; same fixed summation order as the per-layer form) */
int dist_op_temporal_net_bwd_reduce(const float* scratch, int64_t layer_stride, int layers, int clips, int T, int Ct,
                                    float* const* dgamma, float* const* dbeta, void* stream);

/* Fused IntegrationNetwork forward (reference models/module_zoo/branches/dist.py:16-45; integ.hip), bf16, Ci = 384, C4 = 96, 3 temporal taps,
 * t in {4, 8, 16, 32} - DIST_ERR_ARG otherwise (dist_op_layernorm + four dist_op_gemm_nt calls do the same):
 *   R = ffn.c_proj(g(ffn.c_fc(ln(M')))) + temporal_ffn.c_proj(g(conv_{3x1x1}(temporal_ffn.c_fc1(ln_temporal(M')))))
 * on M'[(clip*t + frame)*L + token][Ci].  W1 / W2 / W3 / b1 / b2 / b3 come from dist_op_integration_pack (the two LayerNorms folded into the
 * first pair of weights; MFMA-operand order).  The tensors backward reads are written when given (all or none): Na / Nb (the two affine
 * LayerNorm outputs) or Xhat, mean / rstd, zf_h2 = [ffn.c_fc output | temporal conv output] and hf_g2 = their activations (rows of Ci + C4), h1. */
typedef struct dist_integ_args {
    const void* Mp;
    const void* W1; const void* W2; const void* W3;
    const float* b1; const float* b2; const float* b3;
    const float* ln_w; const float* ln_b; const float* ln_t_w; const float* ln_t_b;   /* only read when Na / Nb are written */
    void* R;
    void* Na; void* Nb; float* mean; float* rstd; void* zf_h2; void* hf_g2; void* h1;
    int clips, t, L, Ci, C4, tk; int dtype; float eps;
    /* T2I in front (dist.py:68-86, alpha = 2, temporal width = C4): t2i_Xp != NULL makes the kernel form M' = M + [cls_token ; conv_strided(X')] itself
     * instead of reading Mp - M [rows][Ci], X' [clips*2t*(L-1)][C4], t2i_W from dist_op_integration_pack (Wt), bias [Ci], cls tokens [t][Ci] (fp32);
     * Mp_out (optional) receives M'.  Xhat / inference forms only. */
    const void* t2i_M; const void* t2i_Xp; const void* t2i_W; const float* t2i_bias; const float* t2i_cls; void* Mp_out;
    /* ... and I2T behind it (dist.py:90-105; needs the T2I operands): i2t_Xnext [clips*2t*(L-1)][C4] = X' + upsample_t(M[:, 1:] i2t_W^T + i2t_bias), the temporal
     * map of the NEXT layer; i2t_W from dist_op_integration_pack (Wi) */
    const void* i2t_W; const float* i2t_bias; void* i2t_Xnext;
    void* Xhat;              /* instead of Na / Nb: the normalised rows (x - mean) rstd themselves, ONE tensor - for a backward pass whose weight-gradient
                              * GEMMs read xhat and whose results dist_op_integration_unfold turns into the gradients of W, gamma and beta */
} dist_integ_args;
int dist_op_integration_fwd(const dist_integ_args* a, void* stream);
/* fp32 master weights of one IntegrationNetwork (torch layouts: Linear [out][in], Conv3d [out][in][3][1][1]) -> the operands above.
 * Sizes: dist_op_integration_pack_elems(Ci, C4, which) with which = 0..5 for W1, W2, W3 (bf16 elements), b1, b2, b3 (floats). */
typedef struct dist_integ_pack_args {
    const float* ffn_fc_w; const float* ffn_fc_b; const float* ln_w; const float* ln_b;
    const float* tf_fc1_w; const float* tf_fc1_b; const float* ln_t_w; const float* ln_t_b;
    const float* tf_fc2_w; const float* tf_fc2_b;
    const float* ffn_proj_w; const float* ffn_proj_b; const float* tf_proj_w; const float* tf_proj_b;
    void* W1; void* W2; void* W3; float* b1; float* b2; float* b3;
    int Ci, C4;
    void* B1; void* B2; void* B3;   /* optional (all or none): the operands of dist_op_integration_bwd, sized like W1 / W2 / W3 */
    const float* t2i_w; void* Wt;   /* optional: temporal2integration_nets.i.linear_fuse.weight [Ci][C4][2][1][1] -> the t2i_W operand (which = 6 elements) */
    const float* i2t_w; void* Wi;   /* optional: integration2temporal_nets.i.linear_fuse.weight [C4][Ci] -> the i2t_W operand (which = 7) */
    void* W4;                       /* optional (with i2t_w): its transpose, the i2t_B operand of dist_op_integration_bwd (which = 8) */
    void* W5;                       /* optional (with t2i_w): the T2I weight as [2 C4][Ci], the t2i_B operand of dist_op_integration_bwd (which = 9) */
} dist_integ_pack_args;
int64_t dist_op_integration_pack_elems(int Ci, int C4, int which);
/* Fused data-gradient backward of the IntegrationNetwork (integ.hip), same geometries as dist_op_integration_fwd:
 *   [dzf | dh2] = (dR [Wp | Wt]) * g'([zf | h2]);  dh1 = conv_t^T(dh2);  dM' = LN'(dzf Wa' + dh1 Wb') (+ dR when add_dR: the last layer's residual)
 * dzf_dh2 (rows of Ci + C4) and dh1 are outputs because the weight-gradient GEMMs read them; dM_copy (optional) receives a second copy of dM'.
 * Xhat / rstd / zf_h2: what dist_op_integration_fwd saved.  The parameter gradients of the two LayerNorms come from dist_op_integration_unfold. */
typedef struct dist_integ_bwd_args {
    const void* dR; const void* zf_h2; const void* Xhat; const float* rstd;
    const void* B1; const void* B2; const void* B3;
    void* dzf_dh2; void* dh1; void* dMp; void* dM_copy;
    int add_dR;
    int clips, t, L, Ci, C4, tk; int dtype;
    /* optional layout of the three gradient outputs (0 / NULL = the defaults above): row pitch of dzf_dh2, a separate place and pitch for its dh2 columns,
     * row pitch of dh1 - e.g. one buffer with rows [dzf | dh1 | dh2] so that ONE weight-gradient GEMM over [dzf | dh1] serves both Linears that read xhat */
    int ld_dzf; void* dh2; int ld_dh2; int ld_dh1;
    /* I2T backward behind it (dist.py:100-105 through autograd; alpha = 2, temporal width = C4): with i2t_dXnext [clips*2t*(L-1)][C4] (the gradient w.r.t. the NEXT
     * layer's temporal map) the kernel forms dY = dX_next[2f] + dX_next[2f+1] (written to i2t_dY [clips*t*(L-1)][C4]: the I2T weight gradient reads it) and
     * dM_copy = dM' + [0 ; dY i2t_B^T] - the whole gradient w.r.t. M.  i2t_B from dist_op_integration_pack (W4). */
    const void* i2t_dXnext; const void* i2t_B; void* i2t_dY;
    /* T2I backward behind that (dist.py:81-86 and the activation X' = g(p) through autograd): t2i_dp [clips*2t*(L-1)][C4] = (dX_next + conv_strided^T(dM'[:, 1:])) * g'(t2i_p)
     * with t2i_p the TemporalNet pre-activation of this layer; dX_next = i2t_dXnext (NULL for the last layer: no such term); t2i_B from dist_op_integration_pack (W5). */
    const void* t2i_B; const void* t2i_p; void* t2i_dp;
    float* t2i_dcls;         /* optional: gradient of the T2I cls tokens [t][Ci] (fp32) += dM' of the cls rows, summed over the clips (atomics) */
    int dM_cls_only;         /* dM_copy only receives the cls rows (token 0 of every frame): for a caller whose next GEMM writes the other rows of that tensor anyway */
} dist_integ_bwd_args;
int dist_op_integration_bwd(const dist_integ_bwd_args* a, void* stream);
/* backward side of the LayerNorm fold.  On entry d_ffn_fc_w / d_tf_fc1_w hold G' = dz^T xhat (dist_op_gemm_tn with B = Xhat) and d_*_b the bias
 * gradients; on exit they hold the gradients of the unfolded weights, dW = G' diag(gamma) + db beta^T, and
 * d_ln_w[k] += sum_n W[n][k] G'[n][k], d_ln_b[k] += sum_n W[n][k] db[n] (likewise ln_temporal through temporal_ffn.c_fc1).  fp32, fixed order. */
typedef struct dist_integ_unfold_args {
    const float* ffn_fc_w; const float* ln_w; const float* ln_b; float* d_ffn_fc_w; float* d_ffn_fc_b; float* d_ln_w; float* d_ln_b;
    const float* tf_fc1_w; const float* ln_t_w; const float* ln_t_b; float* d_tf_fc1_w; float* d_tf_fc1_b; float* d_ln_t_w; float* d_ln_t_b;
    int Ci, C4;
    /* ACCUMULATING form (ABI 9; all four or none): G' and the bias gradients of THIS backward pass were written to scratch instead of the gradient
     * slots (which already hold earlier passes' unfolded gradients): d_*_w += G' diag(gamma) + db beta^T, d_*_b += db, d_ln_* += ... as above. */
    const float* g_ffn_fc_w; const float* g_ffn_fc_b; const float* g_tf_fc1_w; const float* g_tf_fc1_b;
} dist_integ_unfold_args;
int dist_op_integration_unfold(const dist_integ_unfold_args* a, void* stream);
int dist_op_integration_pack(const dist_integ_pack_args* a, void* stream);

/* one-query cross attention (CrossAttentionBlockGenral, clip.py:139-147; dist.py:144,158):
 * q [B, C], kv [B*S, 2C] -> o [B, C], probs [B, H, S] (fp32, saved for backward) */
int dist_op_xattn1q(const void* q, const void* kv, void* o, float* probs, int B, int S, int C, int dtype, void* stream);
int dist_op_xattn1q_bwd(const void* q, const void* kv, const float* probs, const void* d_o,
                        void* dq, void* dkv, int B, int S, int C, int dtype, void* stream);

/* [b,3,T,H,W] fp32 frames -> patch rows [b*T*N, 3*P*P] (c,py,px order) in `dtype` */
int dist_op_patchify(const float* video, void* patches, int b, int T, int H, int W, int P, int dtype, void* stream);

/* misc elementwise (dtype-generic) */
int dist_op_add(const void* a, const void* b, void* out, int64_t n, int dtype, void* stream);
int dist_op_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int dtype, void* stream);
int dist_op_colsum(const void* x, float* out, int64_t rows, int C, int ld, dist_rowmap map, int dtype, void* stream);

/* cosine logits + SoftTargetCrossEntropy forward and backward in one pass
 * (clip.py:509-518, base_blocks.py:579-585, losses.py:29-31).
 * v [b,E] (dtype), text [K,E] fp32 unit or not; outputs fp32. */
int dist_op_logits_loss(const void* v, const float* text, const float* logit_scale, const float* soft_target,
                        float* logits, float* vid_norm, float* loss, void* dv, float* dlogit_scale,
                        const float* dlogits_in, int b, int E, int K, int dtype, void* stream);

/* fused multi-tensor AdamW over a flat fp32 buffer split into segments
 * (torch.optim.AdamW as constructed in models/utils/optimizer.py:67-73) */
typedef struct dist_adamw_seg { int64_t begin, end; float lr, weight_decay; } dist_adamw_seg;
int dist_op_adamw(float* param, const float* grad, float* m, float* v, const dist_adamw_seg* segs_dev, int nseg,
                  int64_t n, float beta1, float beta2, float eps, int step, float grad_scale, void* stream);

/* Batch-mode Mixup / CutMix of one rank's clips, in place on fp32 frames [b][3][T][H][W] (= [b][per_clip]), and the soft
 * target; replaces Mixup._mix_batch and mixup_target of the reference (dataset/utils/mixup.py:212-223, :18-23; call site
 * runs/train.py:92-93).  Clip i is paired with clip b-1-i (x.flip(0)), so ranks exchange nothing.  The host draws lam / the
 * box exactly as the reference does (np.random, dataset/utils/mixup.py:43-64,89-100,160-176) and passes the fp32 values
 * torch would use: lam = float(lam), one_minus_lam = float(1. - lam) (double subtraction first), on / off as mixup_target
 * computes them.  Results are bit-identical to the reference's torch ops.
 *   dist_op_mixup:        x[i] = x[i]*lam + x[b-1-i]*(1-lam) for every i (three roundings per element, as torch)
 *   dist_op_cutmix:       swaps the box [yl,yh) x [xl,xh) of every (channel, frame) plane between clip i and clip b-1-i
 *   dist_op_mixup_target: soft[i][k] = y1*lam + y2*(1-lam), y1 / y2 = smoothed one-hot of labels[i] / labels[b-1-i] */
int dist_op_mixup(float* video, int b, int64_t per_clip, float lam, float one_minus_lam, void* stream);
int dist_op_cutmix(float* video, int b, int planes, int H, int W, int yl, int yh, int xl, int xh, void* stream);
int dist_op_mixup_target(const int64_t* labels, int b, int K, float lam, float one_minus_lam, float on_value, float off_value, float* soft, void* stream);
/* The patch rows of the MIXED batch without the round trip of the mixed frames (Mixup fused into the patch-row gather): exactly what
 * dist_op_patchify returns after dist_op_mixup (kind 1) / dist_op_cutmix (kind 2) - the same three fp32 roundings per element, then the rounding to `dtype` -
 * while `video` is only READ (it keeps the unmixed clips).  kind 0 = dist_op_patchify.  One pass over the frames instead of three (308 MB read + 154 MB
 * written at b = 32 instead of 924 + 154).  Replaces, together with dist_op_mixup_target, the video side of reference dataset/utils/mixup.py:212-223 + clip.py:266. */
int dist_op_patchify_mixed(const float* video, void* patches, int b, int T, int H, int W, int P, int dtype, int kind, float lam, float one_minus_lam,
                           int yl, int yh, int xl, int xh, void* stream);

/* Evaluation side: the operators that consume the predictions right behind the forward pass, so that the
 * multi-view test loop (runs/test.py:24-178) and the per-iteration training metrics (runs/train.py:165-178) need no host round trip.
 *   dist_op_softmax_rows:    y[r][:] = softmax(x[r][:]) in fp32 - the head's eval activation (models/base/base_blocks.py:573-585,
 *                            nn.Softmax(dim=-1) on logits_per_image.mean(dim=1)); x, y fp32 [rows][K], may alias.
 *   dist_op_topk_correct:    correct[i] = number of rows whose label is among the ks[i] highest scores (utils/metrics.py:100-129
 *                            topks_correct; fp32 counts like the reference's .float().sum()).  nk <= 4.  Exactly equal scores rank
 *                            by class index (torch.topk leaves that order unspecified); labels outside [0, K) never count.
 *   dist_op_ensemble_update: TestMeter.update_stats (utils/meters.py:82-112): for every clip i in order, vid = clip_ids[i] / num_clips,
 *                            video_preds[vid] += preds[i] (DIST_ENSEMBLE_SUM, same fp32 addition order as the reference's loop) or
 *                            = max(video_preds[vid], preds[i]) (DIST_ENSEMBLE_MAX); video_labels[vid] = labels[i]; clip_count[vid] += 1.
 *                            *err |= 1 when two views of a video carry different labels (the reference's assert, which like the
 *                            reference only fires once the stored label is > 0), |= 2 for a clip id outside [0, num_videos * num_clips)
 *                            (the reference raises IndexError); the caller zeroes *err and reads it when it finalises. */
enum { DIST_ENSEMBLE_SUM = 0, DIST_ENSEMBLE_MAX = 1 };
int dist_op_softmax_rows(const float* x, int rows, int K, float* y, void* stream);
int dist_op_topk_correct(const float* preds, const int64_t* labels, int n, int K, const int* ks, int nk, float* correct, void* stream);
int dist_op_ensemble_update(float* video_preds, int64_t* video_labels, int64_t* clip_count, const float* preds, const int64_t* labels,
                            const int64_t* clip_ids, int n, int K, int64_t num_videos, int num_clips, int method, int* err, void* stream);

/* ---------------------------------------------------------------------------------------
 * Engine level: the whole hot path behind one handle.
 * ------------------------------------------------------------------------------------- */
typedef struct dist_config {
    int dtype;            /* DIST_F32 | DIST_BF16: activation / working-weight storage */
    int batch;            /* clips per call (max) */
    int frames;           /* DATA.NUM_INPUT_FRAMES (T) */
    int alpha;            /* DATA.SPARSE_SAMPLE_ALPHA */
    int resolution;       /* frame H = W */
    int patch;            /* ViT patch = DIST.S_PATCH_SIZE */
    int width;            /* ViT width d */
    int layers;           /* ViT layers (<= 32) */
    int integration_dim;  /* DIST.INTEGRATION_DIM */
    int temporal_dim;     /* DIST.TEMPORAL_DIM */
    int temporal_kernel;  /* DIST.TEMPORAL_KERNEL_SIZE */
    int temporal_patch;   /* DIST.T_PATCH_SIZE */
    int int_temporal_div; /* 1 / DIST.INTEGRATION_TEMPORAL_MLP_RATIO (4) */
    int ada_layers;       /* DIST.ADA_POOLING_LAYERS */
    int num_classes;      /* VIDEO.HEAD.NUM_CLASSES */
    int embed_dim;        /* CLIP embed dim E */
    int use_tr;           /* bf16 dW GEMMs use LDS transpose reads */
    int vit_fp8;          /* BASELINE config 5 (fp8 frozen spatial branch; bf16 engines only): bit mask of the frozen-ViT GEMMs that run on
                           * e4m3 operands (DIST_EPI_FP8; weights quantised per output channel at pack time, inputs per token in front of
                           * the GEMM): 1 = attn.in_proj, 2 = attn.out_proj, 4 = mlp.c_fc, 8 = mlp.c_proj; 0 = bf16 everywhere.
                           * 16 (with all of 1 | 2 | 4 | 8): the producing epilogues write the e4m3 images themselves (DIST_EPI_OUT8, per-tensor
                           * power-of-two scales from the previous pass's maxima; the first pass after a pack calibrates with the per-token
                           * quantisers) - the hidden tensor of the MLP, the q | k | v tensor and the attention output then exist as e4m3
                           * only */
    int temporal_hidden;  /* ABI 9: hidden width of a TemporalNet = int(TEMPORAL_DIM * DIST.TEMPORAL_CONV_MLP_RATIO) (dist.py:51-58); 0 = TEMPORAL_DIM (ratio 1: every released yaml) */
    int integration_hidden; /* ABI 9: hidden width of IntegrationNetwork.ffn = int(INTEGRATION_DIM * DIST.INTEGRATION_MLP_RATIO) (dist.py:20-25); 0 = INTEGRATION_DIM.  Other widths than the
                           * default run the unfused kernel sequence (the fused TemporalNet / IntegrationNetwork kernels are built for ratio 1). */
    int selected_mask;    /* ABI 9: DIST.SELECTED_LAYERS as a bit mask - bit i set = the output of ViT block i feeds one DiST layer (reference dist.py:170-190, 226:
                           * `for idx, layer_id in enumerate(self.selected_layers)`); 0 = every block (what all released yamls select).  The dist_net.* tables then
                           * hold popcount(mask) layers, numbered 0.. in block order; the frozen ViT always runs all `layers` blocks. */
} dist_config;

typedef struct dist_handle dist_handle;

int dist_create(const dist_config* cfg, dist_handle** out);
void dist_destroy(dist_handle* h);
const char* dist_last_error(const dist_handle* h);

/* parameter tables (reference state-dict names and shapes).
 * kind 0: dist_net.* trainable tensors, laid out back to back in ONE flat fp32 buffer
 *         (offsets in elements); kind 1: frozen visual.* tensors, flat fp32 as well. */
int dist_param_count(const dist_handle* h, int kind);
const char* dist_param_name(const dist_handle* h, int kind, int i);
int dist_param_ndim(const dist_handle* h, int kind, int i);
int64_t dist_param_dim(const dist_handle* h, int kind, int i, int d);
int64_t dist_param_offset(const dist_handle* h, int kind, int i);
int64_t dist_param_total(const dist_handle* h, int kind);
int dist_param_group(const dist_handle* h, int i);   /* DiST optimizer group 0..4 (optimizer.py:138-186 as intended) */

/* bytes the caller must provide */
size_t dist_workspace_bytes(const dist_handle* h);   /* activations saved for backward + scratch */
size_t dist_packed_bytes(const dist_handle* h);      /* working copies of all weights in cfg.dtype */

/* borrowed device pointers; the caller keeps them alive while the handle uses them */
int dist_bind(dist_handle* h, float* theta /*dist_net flat fp32*/, float* grads /*same size*/,
              const float* visual /*flat fp32*/, float* logit_scale, float* dlogit_scale,
              void* packed, void* workspace);

/* refresh the cfg.dtype working copies (GEMM layouts, transposes) from theta / visual.
 * what: 1 = visual, 2 = dist_net, 3 = both.  Call after load_state_dict / optimizer.step. */
int dist_pack_weights(dist_handle* h, int what, void* stream);

/* VisionTransformer.forward under eval()+no_grad (clip.py:263-300,454-458): video [b,3,T,H,W] fp32
 * -> mid_feat kept inside the workspace ([layers][b,t,L,d], read back through dist_debug_tensor("feat.<i>")). */
int dist_vit_forward(dist_handle* h, const float* video, int b, void* stream);
/* One-shot: the NEXT ViT pass that starts (dist_vit_forward, dist_vit_prefetch[_layers] with video != NULL) gathers its patch rows through
 * dist_op_patchify_mixed(kind, lam, ...) - the batch-mode Mixup / CutMix of reference runs/train.py:92-93 applied while the frames are read, the frames
 * themselves untouched.  kind 0 cancels.  Host only. */
int dist_vit_mix_next(dist_handle* h, int kind, float lam, float one_minus_lam, int yl, int yh, int xl, int xh);
/* Software pipelining over batches.  The ViT is frozen (clip.py:454-458: eval() + no_grad), so its forward for batch n+1
 * does not depend on the optimizer step of batch n.  The workspace holds TWO feature slots (patch rows + mid_feat):
 * dist_vit_prefetch runs the ViT of the NEXT batch into the spare slot on `stream` (NULL = the handle's own lowest-priority
 * prefetch stream) after everything already queued on `after` (the stream carrying the step - NULL is the null stream, as
 * for every `stream` argument of this ABI: the backward that last read that slot and the weight re-pack are there);
 * dist_vit_adopt (host only) makes
 * the prefetched slot the current one, i.e. the state after dist_vit_forward of that batch.  The branch forward that
 * follows waits on the slot's per-layer events, whatever stream produced them. */
int dist_vit_prefetch(dist_handle* h, const float* video, int b, void* stream, void* after);
/* The same pass issued in parts, so the caller can spread it over its step (e.g. half beside the branch forward, the rest
 * beside the backward): video != NULL starts a new pass (patch rows + ViT layers [0, layer_end)); video == NULL continues
 * the pass in flight up to layer_end, again behind what is queued on `after` at the time of the call.  The pass is
 * adoptable once layer_end == cfg.layers was reached. */
int dist_vit_prefetch_layers(dist_handle* h, const float* video, int b, int layer_end, void* stream, void* after);
int dist_vit_adopt(dist_handle* h);
/* The CALLER's frozen-ViT features instead of a dist_vit_forward pass - what the reference's DiSTNetwork.forward consumes
 * (models/module_zoo/branches/dist.py:222-247: input['mid_feat']['img'][layer_id] and input['images']): mid_feat[i], i < layers, is the device
 * pointer of ViT block i's output (NULL allowed for a block that is not in dist_config.selected_mask) in the reference's layout [L][b*t][width] (sequence first, clip.py:282-300; src_dtype DIST_F32 or DIST_BF16), video the
 * frames [b][3][T][H][W] fp32 (the temporal stem's input).  Copied into the current feature slot (converted, token-major); dist_branch_forward(b)
 * follows as after dist_vit_forward. */
int dist_features_import(dist_handle* h, const void* const* mid_feat, int src_dtype, const float* video, int b, void* stream);
/* DiSTNetwork.forward + cosine logits (dist.py:222-247, clip.py:509-518): -> logits [b,K] fp32,
 * vid_logits [b,E] fp32 (L2-normalised video embedding) */
int dist_branch_forward(dist_handle* h, const float* text_features, int b, float* logits, float* vid_logits, void* stream);
/* Inference mode (the reference's `torch.no_grad()` evaluation loops, runs/train.py:205-260, runs/test.py:24-178): while on, dist_branch_forward
 * writes nothing that only a backward pass would read (the TemporalNet's normalised / activated tensors, the pre-activations of the
 * IntegrationNetwork's two QuickGELUs: 87 MB per layer at the bench geometry), and dist_branch_backward refuses to run behind such a
 * forward (DIST_ERR_STATE).  Those two QuickGELUs then act on the fp32 accumulators instead of the stored bf16 pre-activations: the logits
 * agree with a training-mode forward to bf16 rounding (0.005 on a +-6 range), not bit for bit. */
int dist_set_inference(dist_handle* h, int on);
/* backward of the branch given dlogits [b,K] fp32 (autograd in the reference, runs/train.py:110);
 * accumulates into the bound flat grads buffer (zeroed first when zero_grads != 0). */
int dist_branch_backward(dist_handle* h, const float* dlogits, int b, int zero_grads, void* stream);
/* Data-parallel overlap hook: during dist_branch_backward, `fn(user, begin, end)` is called on the host right
 * after the launches that complete the gradient range [begin, end) of the flat dist_net buffer have been
 * enqueued (ada-pooling + head first, then layer L-1 ... 0, then the stem), so the caller can record an event
 * on the compute stream and start that slice's all-reduce on a side stream while the rest of backward runs
 * (replaces DDP's bucket hooks, reference models/base/builder.py:72-74).  fn == NULL removes the hook. */
typedef void (*dist_grad_ready_fn)(void* user, int64_t begin, int64_t end);
int dist_set_grad_ready_hook(dist_handle* h, dist_grad_ready_fn fn, void* user);
/* SoftTargetCrossEntropy value and dlogits for the logits of the last branch_forward */
int dist_loss(dist_handle* h, const float* soft_target, int b, float* loss, float* dlogits, void* stream);
/* Phase marks: device-side timestamps of the last step (HIP events on the streams the work runs on), for reading the real
 * schedule of the pipelined step without a profiler.  dist_marks_read synchronises the device and returns, per mark, the
 * milliseconds since DIST_MARK_VIT_BEGIN of the same step (NaN for a mark that was not recorded). */
enum { DIST_MARK_VIT_BEGIN = 0, DIST_MARK_VIT_END = 1, DIST_MARK_FWD_BEGIN = 2, DIST_MARK_FWD_MID = 3, DIST_MARK_FWD_END = 4,
       DIST_MARK_BWD_BEGIN = 5, DIST_MARK_BWD_END = 6, DIST_MARK_STEP_END = 7, DIST_NMARKS = 8 };
int dist_marks_enable(dist_handle* h, int on);
int dist_marks_read(dist_handle* h, float* ms, int n);
/* measurement hook for bench.py: between begin and end every launch of the dominant kernel (the 256x256x64
 * LDS-DMA MFMA GEMM of the frozen ViT, gemm_fast.hip) is bracketed by HIP events on its own stream; end() synchronises
 * those events and returns the summed duration, the summed algorithmic FLOPs (2*M*N*K) and the launch count. */
int dist_profile_begin(dist_handle* h);
int dist_profile_end(dist_handle* h, double* ms_total, double* flops_total, int* launches);
/* read back an intermediate for tests: name in {"feat.<i>","stem","tn_out.<i>","int_out.<i>","x_temporal.<i>","mid.<i>"} */
int dist_debug_tensor(dist_handle* h, const char* name, const void** ptr, int64_t* rows, int* cols);

#ifdef __cplusplus
}
#endif
#endif
