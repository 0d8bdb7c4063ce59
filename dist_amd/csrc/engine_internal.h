// Engine internals shared by the engine's translation units (engine.hip: handle, tables, workspace, packing; engine_vit.hip: the frozen ViT pass;
// engine_fwd.hip: the branch forward + loss; engine_bwd.hip: the branch backward).  Not part of the C ABI.
#pragma once
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace dist_engine {

struct Param {
    std::string name;
    int ndim = 0;
    int64_t dim[5] = {0, 0, 0, 0, 0};
    int64_t offset = 0;
    int64_t numel = 0;
    int group = -1;
};

struct PW { long f = -1, b = -1; };          // packed working copies (element offsets): forward / data-gradient layout

struct Lin {                                 // y = x W^T + bias, W [N][K] (optionally conv taps)
    long w = -1, bias = -1;                  // offsets into theta / visual (fp32 master)
    PW pk;
    int N = 0, K = 0, taps = 1;
};
struct LNp { long w = -1, b = -1; int C = 0; };

struct VitLayer {
    LNp ln1, ln2; Lin qkv, out, fc, proj;
    // LayerNorm fold (DIST_EPI_LNFOLD): W diag(gamma) in bf16 (packed-buffer element offsets), column sums and folded biases
    long pk_fold_qkv = -1, pk_fold_fc = -1;
    float *cs_qkv = nullptr, *b_qkv = nullptr, *cs_fc = nullptr, *b_fc = nullptr;
    // fp8 frozen spatial branch (dist_config.vit_fp8, BASELINE config 5): e4m3 copies [N][K] of the four GEMM weights (the folded ones for
    // qkv / fc) with per-output-channel scales, and the column sums of the DEQUANTISED folded weights (what the fold's mean term must use)
    struct Fp8W { unsigned char* q = nullptr; float* s = nullptr; };
    Fp8W q_qkv, q_out, q_fc, q_proj;
    float *cs8_qkv = nullptr, *cs8_fc = nullptr;
};
struct DistLayer {
    LNp tn_ln; Lin tn_fc1, tn_fc2;           // TemporalNet
    Lin in_lin, i2t, t2i; long cls_token = -1;
    LNp in_ln, in_ln_t; Lin ffn_fc, ffn_proj, tf_fc1, tf_fc2, tf_proj;   // IntegrationNetwork
    long pk_proj_f = -1, pk_proj_b = -1;     // the two c_proj weights side by side: [Ci][Ci+C4] forward, [Ci+C4][Ci] data-gradient
    // fused IntegrationNetwork forward (integ.hip): MFMA-operand-ordered weights with the two LayerNorms folded in (workspace pointers)
    void *ig_W1 = nullptr, *ig_W2 = nullptr, *ig_W3 = nullptr; float *ig_b1 = nullptr, *ig_b2 = nullptr, *ig_b3 = nullptr;
    void *ig_B1 = nullptr, *ig_B2 = nullptr, *ig_B3 = nullptr;      // ... and the data-gradient side (fused backward)
    void* ig_Wt = nullptr;                                          // ... and the T2I weight (T2I formed in front of the fused forward)
    void* ig_Wi = nullptr;                                          // ... and the I2T weight (I2T behind it)
    void* ig_W4 = nullptr;                                          // ... and its transpose (I2T backward behind the fused backward)
    void* ig_W5 = nullptr;                                          // ... and the T2I weight transposed (T2I backward behind that)
};
struct XAttn { LNp ln1; Lin q, kv, out; };   // CrossAttentionBlockGenral (in_proj split into q / kv rows)
struct AdaLayer { long pos = -1; XAttn sp, tm; LNp ln_sp, ln_tm; Lin sp_fc, sp_proj, tm_fc, tm_proj; };

template <typename T> struct Buf { T* p = nullptr; };

struct Arena {
    char* base = nullptr;
    size_t off = 0;
    void* take(size_t bytes) {
        off = (off + 255) & ~(size_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        return p;
    }
};

struct DistLayerWs {
    void *X, *U, *z, *V, *p, *Xp, *M, *Mp, *Na, *Nb, *zf, *hf, *h1, *h2, *g2, *R;
    float *tn_mean, *tn_rstd, *in_mean, *in_rstd;
};
struct AdaWs {
    void *kn, *kv, *qn, *q, *o, *s1, *sn, *zs, *hs, *c, *kn2, *kv2, *qn2, *q2, *o2, *u1, *un, *zu, *hu;
    float *kn_mean, *kn_rstd, *qn_mean, *qn_rstd, *s1_mean, *s1_rstd, *kn2_mean, *kn2_rstd, *qn2_mean, *qn2_rstd, *u1_mean, *u1_rstd;
    float *probs, *probs2;
};

}  // namespace dist_engine
using namespace dist_engine;

struct dist_handle {
    dist_config cfg;
    int es = 2;                                  // element size of cfg.dtype
    // derived geometry
    int G = 0, N = 0, L = 0, t = 0, heads = 0, C4 = 0, Kp = 0, PP3 = 0, iheads = 0;
    std::vector<Param> params[2];
    std::unordered_map<std::string, int> index[2];
    int64_t total[2] = {0, 0};
    // model tables
    Lin conv1; long class_emb = -1, pos_emb = -1; LNp ln_pre;
    std::vector<VitLayer> vit;
    Lin stem; std::vector<DistLayer> dl; std::vector<AdaLayer> ada;
    Lin cls_proj; LNp ln_post; Lin proj; long agg_cls = -1, agg_sp_cls = -1;
    // packing
    std::vector<PackDesc> descs; std::vector<int> blk_desc, blk_first;
    int nblk_visual = 0;                         // blocks [0, nblk_visual) pack visual.*, the rest dist_net.*
    size_t packed_hdr = 0, packed_total = 0; long packed_elems = 0;
    // bound buffers
    float *theta = nullptr, *grads = nullptr, *logit_scale = nullptr, *dlogit_scale = nullptr;
    const float* visual = nullptr;
    char *packed = nullptr, *ws = nullptr;
    size_t ws_bytes = 0;
    // workspace
    void *patches, *x0, *xa, *hbuf, *qkv, *att, *mlp;
    float *lnstats2 = nullptr, *lnstats3 = nullptr; int dummy = 0, dummy_reps = 1;   // perturbation experiment (DIST_AMD_DUMMY)
    float* lnstats = nullptr;                    // [2][rowsS] mean / rstd of the LayerNorm folded into the next ViT GEMM
    float* lnpart = nullptr;                     // [width / 64][rowsS][2] partial (sum, sum of squares) of the residual stream, left by the GEMM that wrote it
    unsigned char* aq = nullptr; float* sa = nullptr;   // vit_fp8: e4m3 image [rowsS][<= 4 width] + per-row scales of the GEMM input being consumed
    // vit_fp8 & 16: the producing epilogues write the e4m3 images themselves (DIST_EPI_OUT8) with per-tensor scales of the PREVIOUS pass:
    // per block four tensors - 0 = attention-block output (c_fc input), 1 = hidden (c_proj input), 2 = block output (next in_proj input),
    // 3 = attention output (out_proj input; written as e4m3 by the attention kernel itself), 4 = q | k | v (in_proj output, head-major e4m3 only)
    unsigned char *x8 = nullptr, *xa8 = nullptr; float *f8_amax = nullptr, *f8_scale = nullptr;
    long f8_passes = 0; int x8_layer = -1;
    bool vit_fold = false;                       // ln_1 -> in_proj and ln_2 -> c_fc folded (bf16, shapes the LDS-DMA GEMM takes)
    std::vector<void*> feat;
    // Two feature slots (patch rows + the 12 mid_feat tensors + their events): the frozen ViT of the NEXT batch can fill the
    // spare slot (dist_vit_prefetch) while the branch forward / backward of the current batch read the other one.
    // `patches`, `feat`, `ev_feat`, `ev_pre` above / below always alias slot[cur].
    struct FeatSlot {
        void* patches = nullptr; std::vector<void*> feat; std::vector<hipEvent_t> ev_feat; hipEvent_t ev_pre = nullptr;
        int b = 0; bool prefetched = false;
        int next_layer = 0, pending_b = 0;       // a prefetch pass issued in parts (dist_vit_prefetch_layers)
        std::vector<char> valid;                 // feat[i] holds block i of the clip this slot was last filled with (dist_features_import only writes the blocks it is given)
    } slot[2];
    int cur = 0;
    hipEvent_t ev_vit_done = nullptr, ev_after = nullptr, ev_bpre = nullptr; bool vit_ran = false;
    void use_slot(int k) { cur = k; patches = slot[k].patches; feat = slot[k].feat; ev_feat = slot[k].ev_feat; ev_pre = slot[k].ev_pre; }
    std::vector<DistLayerWs> lw; std::vector<AdaWs> aw;
    void* Xlast;
    std::vector<void*> sbuf, ubuf;
    void *Fz, *mean_cls, *ysum, *zpost, *v;
    float *y_mean, *y_rstd, *logits, *dlogits, *loss;
    // backward scratch
    // layer-loop scratch, double-buffered by layer parity (the weight-gradient stream lags the data-gradient chain)
    struct BwdSet { void *dMp, *dM, *dXp, *dp, *dXo, *dz, *dU, *dY, *dh2, *dh1, *dzf, *dNa, *dNb, *dcat; } bs[2];   // dcat: [dzf | dh1 | dh2] rows of Ci + 2 C4 (fused IntegrationNetwork backward)
    void *dR, *dkv, *dkn;
    float* ln_partial = nullptr; long ln_partial_elems = 0;     // per-block parameter-gradient sums of the LayerNorm backward (two-phase, no atomics)
    float* tnb_scratch = nullptr; long tnb_scratch_elems = 0;   // parameter-gradient partial rows of the fused TemporalNet backward
    bool ig_on = false, ig_xhat = false, ig_bwd = false, ig_t2i = false, ig_i2t = false, ig_i2tb = false, ig_t2ib = false, keep_mid = false; void* ig_descs = nullptr;   // ig_xhat: the forward keeps xhat (in the Na buffer) instead of Na / Nb; the weight gradients of the two folded Linears are unfolded afterwards
                  // fused IntegrationNetwork forward: taken for this geometry; its pack descriptors (device)
    float* tn_partial[3] = {nullptr, nullptr, nullptr};   // two-phase dW reduction scratch, one per stream that launches dW GEMMs
    long tn_partial_elems = 0;
    float* ig_gscratch = nullptr;              // [layers][(Ci + C4) * Ci + (Ci + C4)]: G' = dz^T xhat and db of the two folded Linears when dist_branch_backward ACCUMULATES (zero_grads = 0)
    long ig_gscratch_elems = 0;                // per layer
    std::vector<int> sel;                      // DIST.SELECTED_LAYERS: ViT block of DiST layer i (dist_config.selected_mask)
    int nsel = 0;                              // number of DiST layers
    int Ch = 0, Cf = 0;                        // hidden widths: TemporalNet (Ct * TEMPORAL_CONV_MLP_RATIO), IntegrationNetwork.ffn (Ci * INTEGRATION_MLP_RATIO)
    int wgrad_blocks = 0;                      // dist_gemm_tn_args.max_blocks of the engine's weight gradients (0 = the library's default, 96; one block per CU for the last
                                               // layers of the pass - whose gradients finish behind the chain - was measured: 17.85 -> 17.90 ms, not kept)
    bool bwd_accumulate = false;               // the running dist_branch_backward was called with zero_grads = 0
    // weight-gradient side stream (created once per handle; host-side objects only)
    hipStream_t side = nullptr, side2 = nullptr, pf = nullptr;   // pf: the handle's own ViT prefetch stream
    struct { int kind = 0; float lam = 1.f, oml = 0.f; int yl = 0, yh = 0, xl = 0, xh = 0; } mix;   // dist_vit_mix_next: consumed by the next pass that gathers patch rows
    hipStream_t chain2 = nullptr;                                // backward: the temporal data-gradient chain (T2I data gradient + TemporalNet backward), beside the integration chain
    std::vector<hipEvent_t> ev_dmp, ev_dx;                       // chain -> chain2: dM'_i written; chain2 -> chain: dX_i written
    hipEvent_t ev_c2 = nullptr;
    int skip = 0;                              // DIST_AMD_SKIP (measurement knob, results WRONG): 1 = no weight-gradient GEMMs, 2 = no TemporalNet backward data-gradient kernels, 4 = no TemporalNet forward, 8 = no IntegrationNetwork forward GEMMs, 16 = no large-wgrad (in_lin / proj pair / ffn_fc) only, 32 = no ViT attention, 64 = no ViT MLP (fc + proj GEMMs)
    int serial = 0;                            // DIST_AMD_SERIAL (measurement knob): bit 0 = branch forward, bit 1 = backward on the caller's stream only
    std::vector<hipEvent_t> ev_a;              // chain -> side: "buffer produced"
    std::vector<hipEvent_t> ev_b_dr, ev_b_done; // side -> chain: per layer "dR consumed", "all weight gradients of the layer issued and done"
    hipEvent_t ev_join = nullptr, ev_pre = nullptr, ev_b2 = nullptr;
    std::vector<hipEvent_t> ev_feat;           // chain -> side: ViT layer i output (mid_feat[i]) is complete
    void *dv, *dzp, *dy, *du, *ds, *dc, *dzu, *dun, *do2, *dq2, *dkv2, *dqn2, *dkn2, *dzs, *dsn, *do_, *dq, *dqn;
    int fwd_b = 0, branch_b = 0;
    bool inference = false, branch_infer = false;   // dist_set_inference: the next branch forwards keep nothing for a backward pass
    const float* text = nullptr;               // borrowed: text features of the last branch_forward
    // gradient-ready hook + the slices it reports
    dist_grad_ready_fn grad_hook = nullptr; void* grad_hook_user = nullptr;
    std::vector<int64_t> layer_begin, layer_end; int64_t tail_begin = 0;
    // phase marks (dist_marks_enable / dist_marks_read): device-side timestamps of the last step on the streams the work runs
    // on, taken WITHOUT a profiler (rocprofv3 makes the ~700 launches of a step host-bound and shows a schedule that the
    // un-profiled run does not have)
    bool marks_on = false;
    hipEvent_t mark_ev[DIST_NMARKS] = {};
    // measurement hook (dist_profile_begin/end)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;           // pairs (start, stop)
    int prof_n = 0;
    double prof_flops = 0.0;
    char err[512] = {0};
};

namespace dist_engine {

inline int fail(dist_handle* h, int rc, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(h->err, sizeof(h->err), fmt, ap);
    va_end(ap);
    return rc;
}

inline void mark(dist_handle* h, int which, hipStream_t st) {
    if (h->marks_on && h->mark_ev[which]) hipEventRecord(h->mark_ev[which], st);
}
#define RUN(call)                                                        \
    do {                                                                 \
        int rc_ = (call);                                                \
        if (rc_ != DIST_OK) return fail(h, rc_, "%s failed (%d) at %s:%d", #call, rc_, __FILE__, __LINE__); \
    } while (0)

inline std::string fmt(const char* f, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return buf;
}

// ---- launch helpers ---------------------------------------------------------------------------------------
struct Ctx {
    dist_handle* h;
    hipStream_t s;
    int dtype;
    const char* pk(long off) const { return h->packed + h->packed_hdr + (size_t)off * h->es; }
    const float* th(long off) const { return h->theta + off; }
    const float* vs(long off) const { return h->visual + off; }
    float* gr(long off) const { return h->grads + off; }
};

inline dist_rowmap RM(int mode = DIST_RM_PLAIN, int p0 = 0, int p1 = 0, int sign = 1) { return dist_rowmap{mode, p0, p1, sign}; }
inline dist_outmap OM(int mode = DIST_OM_PLAIN, int p0 = 0, int p1 = 0, int p2 = 0) { return dist_outmap{mode, p0, p1, p2}; }

// C (and/or C2) = epi(A[amap] . W^T): thin positional wrapper over dist_op_gemm_nt
inline int gemm(const Ctx& c, const void* A, int lda, const void* W, long M, int N, int K, int taps, void* C, int ldc,
         const float* bias, const void* res, const void* aux, void* C2, dist_rowmap am = RM(), dist_outmap om = OM(), int extra_flags = 0, const float* bias2 = nullptr,
         float* rowstats = nullptr) {
    dist_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = W; g.C = C; g.C2 = C2; g.bias = bias; g.bias2 = bias2; g.res = res; g.aux = aux;
    if (rowstats) { g.rowstats = rowstats; extra_flags |= DIST_EPI_ROWSTATS; }
    g.M = M; g.N = N; g.K = K; g.taps = taps;
    g.lda = lda; g.ldb = taps * K; g.ldc = ldc; g.ldc2 = ldc; g.ldres = ldc; g.ldaux = ldc;
    g.amap = am; g.omap = om;
    g.flags = (bias ? DIST_EPI_BIAS : 0) | (res ? DIST_EPI_RES : 0) | (aux ? DIST_EPI_MULG : 0) | (C2 ? DIST_EPI_ACT2 : 0) | extra_flags;
    g.dtype = c.dtype;
    dist_handle* h = c.h;
    const bool dominant = h->prof_on && dist_k_gemm_fast_eligible(&g);
    if (!dominant) return dist_op_gemm_nt(&g, c.s);
    if (h->prof_n + 2 > (int)h->prof_ev.size()) {
        for (int i = 0; i < 256; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return DIST_ERR_STATE; h->prof_ev.push_back(e); }
    }
    hipEventRecord(h->prof_ev[h->prof_n], c.s);
    const int rc = dist_op_gemm_nt(&g, c.s);
    hipEventRecord(h->prof_ev[h->prof_n + 1], c.s);
    h->prof_n += 2;
    h->prof_flops += 2.0 * (double)M * N * K;
    return rc;
}

// weight gradient of a Lin into the flat grads buffer, in the reference parameter layout
// C = epi(LN(A) . W^T) with the LayerNorm folded into the GEMM (DIST_EPI_LNFOLD): A holds the RAW rows, Wf = W diag(gamma),
// stats = [2][M] mean / rstd, colsum / biasf from dist_op_ln_fold.  Returns 1 when the LDS-DMA kernel took it, 0 when the shape
// is not eligible (the caller runs LayerNorm + GEMM instead), < 0 on error.
inline int gemm_lnfold(const Ctx& c, const void* A, int lda, const void* Wf, long M, int N, int K, void* C, int ldc, const float* biasf,
                const float* stats, const float* colsum, void* C2, dist_outmap om = OM()) {
    dist_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.B = Wf; g.C = C; g.C2 = C2; g.bias = biasf; g.bias2 = colsum; g.aux = stats;
    g.M = M; g.N = N; g.K = K; g.taps = 1;
    g.lda = lda; g.ldb = K; g.ldc = ldc; g.ldc2 = ldc; g.ldres = ldc; g.ldaux = ldc;
    g.amap = RM(); g.omap = om;
    g.flags = DIST_EPI_BIAS | DIST_EPI_LNFOLD | (C2 ? DIST_EPI_ACT2 : 0);
    g.dtype = c.dtype;
    if (!dist_k_gemm_fast_eligible(&g)) return 0;
    dist_handle* h = c.h;
    if (h->prof_on) {
        if (h->prof_n + 2 > (int)h->prof_ev.size()) {
            for (int i = 0; i < 256; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return DIST_ERR_STATE; h->prof_ev.push_back(e); }
        }
        hipEventRecord(h->prof_ev[h->prof_n], c.s);
    }
    const int rc = dist_op_gemm_nt(&g, c.s);
    if (h->prof_on) {
        hipEventRecord(h->prof_ev[h->prof_n + 1], c.s);
        h->prof_n += 2;
        h->prof_flops += 2.0 * (double)M * N * K;
    }
    return rc < 0 ? rc : 1;
}
// C (or C2 = quickgelu) = epi(Aq . Wq^T) on e4m3 operands (DIST_EPI_FP8): Aq / sa from dist_op_quant_rows_fp8 over the bf16 input, Wq / its
// scales from the pack.  `stats` != nullptr: LayerNorm fold (bias = folded bias, colsum of the dequantised weights).  Returns 1 when
// launched, 0 when the shape is not eligible (the caller runs the bf16 GEMM), < 0 on error.
struct Out8 { unsigned char* img = nullptr; const float* scale = nullptr; float* amax = nullptr; bool act = false; };   // DIST_EPI_OUT8 (act: QuickGELU'd, e4m3 only)
inline int gemm_fp8(const Ctx& c, const unsigned char* Aq, const float* sa, bool sa_scalar, const VitLayer::Fp8W& W, long M, int N, int K, void* C, int ldc,
             const float* bias, const void* res, void* C2, const float* stats, const float* colsum, float* rowstats, dist_outmap om = OM(), Out8 o8 = Out8()) {
    dist_gemm_args g;
    memset(&g, 0, sizeof(g));
    g.A = Aq; g.B = W.q; g.C = C; g.C2 = C2; g.bias = bias; g.bias2 = colsum; g.aux = stats; g.res = res;
    g.a_scale = sa; g.b_scale = W.s; g.rowstats = rowstats;
    g.M = M; g.N = N; g.K = K; g.taps = 1;
    g.lda = K; g.ldb = K; g.ldc = ldc; g.ldc2 = ldc; g.ldres = ldc; g.ldaux = ldc;
    g.amap = RM(); g.omap = om;
    g.flags = DIST_EPI_FP8 | (sa_scalar ? DIST_EPI_FP8_ASCALAR : 0) | (bias ? DIST_EPI_BIAS : 0) | (res ? DIST_EPI_RES : 0) | (stats ? DIST_EPI_LNFOLD : 0) |
              ((C2 || o8.act) ? DIST_EPI_ACT2 : 0) | (rowstats ? DIST_EPI_ROWSTATS : 0);
    if (o8.img) { g.C8 = o8.img; g.ldc8 = om.mode == DIST_OM_HEADS ? 64 : N; g.out8_scale = o8.scale; g.out8_amax = o8.amax; g.flags |= DIST_EPI_OUT8; }
    g.dtype = DIST_BF16;
    if (!dist_k_gemm_fast_eligible(&g)) return 0;
    const int rc = dist_op_gemm_nt(&g, c.s);
    return rc < 0 ? rc : 1;
}
// does the e4m3 mode of the LDS-DMA kernel take C [M][N] = A [M][K] W^T ?
inline bool fp8_shape_ok(const Ctx& c, long M, int N, int K) {
    dist_gemm_args g;
    memset(&g, 0, sizeof(g));
    void* nz = reinterpret_cast<void*>(16);
    g.A = nz; g.B = nz; g.C = nz; g.a_scale = static_cast<const float*>(nz); g.b_scale = static_cast<const float*>(nz);
    g.M = M; g.N = N; g.K = K; g.taps = 1; g.lda = K; g.ldb = K; g.ldc = g.ldc2 = g.ldres = g.ldaux = N;
    g.amap = RM(); g.omap = OM(); g.flags = DIST_EPI_FP8; g.dtype = DIST_BF16;
    return dist_k_gemm_fast_eligible(&g);
}
// can C = A W^T + bias + res (plain maps) leave DIST_EPI_ROWSTATS partials, i.e. does the LDS-DMA kernel take this shape?
inline bool rowstats_ok(const Ctx& c, long M, int N, int K) {
    dist_gemm_args g;
    memset(&g, 0, sizeof(g));
    void* nz = reinterpret_cast<void*>(16);
    g.A = nz; g.B = nz; g.C = nz; g.res = nz; g.bias = static_cast<const float*>(nz); g.rowstats = static_cast<float*>(nz);
    g.M = M; g.N = N; g.K = K; g.taps = 1; g.lda = K; g.ldb = K; g.ldc = g.ldc2 = g.ldres = g.ldaux = N;
    g.amap = RM(); g.omap = OM(); g.flags = DIST_EPI_BIAS | DIST_EPI_RES | DIST_EPI_ROWSTATS; g.dtype = c.dtype;
    return N % 64 == 0 && dist_k_gemm_fast_eligible(&g);
}
inline int wgrad(const Ctx& c, const Lin& l, const void* dY, int ld_dy, const void* X, int ldx, long M,
          dist_rowmap am = RM(), dist_rowmap bm = RM(), int style = 0, bool with_bias = false, float* out_w = nullptr, float* out_b = nullptr) {
    dist_gemm_tn_args g;
    memset(&g, 0, sizeof(g));
    g.A = dY; g.B = X; g.out = c.gr(l.w);
    g.M = M; g.NI = l.N; g.K = l.K; g.taps = l.taps; g.lda = ld_dy; g.ldb = ldx; g.amap = am; g.bmap = bm;
    if (style == 0) { g.so_i = l.K; g.so_tap = 0; g.so_outer = 1; g.inner = 1; }
    else if (style == 1 || style == 2) { g.so_i = (long)l.K * l.taps; g.so_tap = 1; g.so_outer = l.taps; g.inner = 1; }
    else if (style == 4) { g.NI = l.K; g.K = l.N; g.so_i = l.N; g.so_tap = 0; g.so_outer = 1; g.inner = 1; }   // [K][N] matrix used as x @ W
    else { const int PP3 = c.h->PP3, PP = PP3 / 3; g.K = PP3; g.so_i = (long)PP3 * l.taps; g.so_tap = PP; g.so_outer = (long)PP * l.taps; g.inner = PP; }
    g.dtype = c.dtype; g.use_tr = c.h->cfg.use_tr; g.max_blocks = c.h->wgrad_blocks;
    if (c.h->skip & 1) return DIST_OK;
    if ((c.h->skip & 16) && l.N >= 384 && l.K >= 384) return DIST_OK;
    g.colsum = (with_bias && l.bias >= 0) ? c.gr(l.bias) : nullptr;       // db fused into the same pass over dY
    if (out_w) { g.out = out_w; if (g.colsum) g.colsum = out_b; }         // (the accumulating backward: G' of a folded Linear goes to scratch)
    {   // two-phase reduction scratch of the stream this launch goes to
        dist_handle* h = c.h;
        const int k = c.s == h->side ? 1 : (c.s == h->side2 ? 2 : 0);
        g.partial = h->tn_partial[k]; g.partial_elems = h->tn_partial_elems;
    }
    return dist_op_gemm_tn(&g, c.s);
}
// weight (+ bias) gradients of two Linears that consume the SAME dY and whose inputs are stored side by side
// ([X1 | X2], ldx = K1 + K2): one pass over dY, columns < K1 go to l1's weight, the rest to l2's; both biases get colsum(dY)
inline int wgrad_pair(const Ctx& c, const Lin& l1, const Lin& l2, const void* dY, int ld_dy, const void* X12, int ldx, long M) {
    dist_gemm_tn_args g;
    memset(&g, 0, sizeof(g));
    g.A = dY; g.B = X12; g.out = c.gr(l1.w);
    g.M = M; g.NI = l1.N; g.K = l1.K + l2.K; g.taps = 1; g.lda = ld_dy; g.ldb = ldx; g.amap = RM(); g.bmap = RM();
    g.so_i = l1.K; g.so_tap = 0; g.so_outer = 1; g.inner = 1;
    g.split_c = l1.K; g.out2 = c.gr(l2.w); g.so_i2 = l2.K;
    if (c.h->skip & 17) return DIST_OK;
    g.dtype = c.dtype; g.use_tr = c.h->cfg.use_tr; g.max_blocks = c.h->wgrad_blocks;
    g.colsum = c.gr(l1.bias); g.colsum2 = c.gr(l2.bias);
    dist_handle* h = c.h;
    const int k = c.s == h->side ? 1 : (c.s == h->side2 ? 2 : 0);
    g.partial = h->tn_partial[k]; g.partial_elems = h->tn_partial_elems;
    return dist_op_gemm_tn(&g, c.s);
}
inline int bgrad(const Ctx& c, long bias_off, const void* dY, long rows, int C, dist_rowmap m = RM()) {
    return dist_op_colsum(dY, c.gr(bias_off), rows, C, C, m, c.dtype, c.s);
}
inline int ln_fwd(const Ctx& c, const float* wbase, const LNp& l, const void* x, void* y, long rows, float* mean, float* rstd,
           const LNp* l2 = nullptr, void* y2 = nullptr, const float* addend = nullptr, int period = 0) {
    dist_ln_args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.y2 = y2; a.w = wbase + l.w; a.b = wbase + l.b;
    if (l2) { a.w2 = wbase + l2->w; a.b2 = wbase + l2->b; }
    a.addend = addend; a.addend_period = period; a.mean = mean; a.rstd = rstd;
    a.rows = rows; a.C = l.C; a.dtype = c.dtype; a.eps = 1e-5f;
    return dist_op_layernorm(&a, c.s);
}
inline int ln_bwd(const Ctx& c, const LNp& l, const void* x, const float* mean, const float* rstd, const void* dy, void* dx, bool accumulate,
           long rows, const LNp* l2 = nullptr, const void* dy2 = nullptr, const void* dx_add = nullptr, void* dx_copy = nullptr, bool param_grads = true) {
    dist_ln_bwd_args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.mean = mean; a.rstd = rstd; a.dy = dy; a.w = c.th(l.w); a.dx = dx; a.accumulate_dx = accumulate ? 1 : 0;
    if (param_grads) { a.dw = c.gr(l.w); a.db = c.gr(l.b); }
    if (l2) { a.dy2 = dy2; a.w2 = c.th(l2->w); if (param_grads) { a.dw2 = c.gr(l2->w); a.db2 = c.gr(l2->b); } }
    a.rows = rows; a.C = l.C; a.dtype = c.dtype;
    a.dx_add = dx_add; a.dx_copy = dx_copy;
    // Two-phase parameter gradients (dist_ln_bwd_args.partial): measured in the step and NOT the default - 20.03 -> 20.20 ms with the same grid
    // caps, 20.2 with 512 blocks (three alternations): the second launch sits on the data-gradient chain and costs more than the same-line
    // atomics it removes.  DIST_AMD_LN_TWO_PHASE=1 turns it on (the data-gradient chain owns the scratch: its launches are serial).
    static const bool two_phase = DIST_AB_KNOB("DIST_AMD_LN_TWO_PHASE", 0) == 1;
    if (two_phase && c.s != c.h->side && c.s != c.h->side2) { a.partial = c.h->ln_partial; a.partial_elems = c.h->ln_partial_elems; }
    return dist_op_layernorm_bwd(&a, c.s);
}

}  // namespace dist_engine
