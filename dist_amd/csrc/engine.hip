// Engine: the whole DiST hot path behind one handle (include/dist_amd.h, "Engine level").
//
//   dist_vit_forward     frozen CLIP ViT (reference clip.py:263-300, eval + no_grad)
//   dist_branch_forward  DiSTNetwork.forward + cosine logits (dist.py:222-247, clip.py:509-518)
//   dist_loss            SoftTargetCrossEntropy (losses.py:29-31)
//   dist_branch_backward hand-derived backward of the branch (autograd in the reference)
//
// Token layouts: integration / ViT tokens S[b, j, l, c] as rows
// (b*t + j)*L + l, temporal map X[b, k, n, c] as rows (b*T + k)*N + n.  Every Linear and
// Conv3d is one launch of the row-mapped MFMA GEMM (gemm_nt.hip) with fused epilogues; no
// im2col, permute or upsample tensor is ever materialised.  The host side only sequences
// launches on the caller's stream; it never allocates device memory and never syncs.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace {

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

}  // namespace

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

namespace {

int fail(dist_handle* h, int rc, const char* fmt, ...) {
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

std::string fmt(const char* f, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return buf;
}

long add_param(dist_handle* h, int kind, const std::string& name, std::initializer_list<int64_t> dims) {
    Param p;
    p.name = name;
    p.ndim = (int)dims.size();
    p.numel = 1;
    int i = 0;
    for (int64_t d : dims) { p.dim[i++] = d; p.numel *= d; }
    p.offset = h->total[kind];
    h->total[kind] += p.numel;
    if (kind == 0) {
        // DiST optimizer groups as intended by models/utils/optimizer.py:138-186
        const bool one_d = name.find("bias") != std::string::npos || p.ndim == 1;
        auto ends_with = [&](const char* s) { const size_t n = strlen(s); return name.size() >= n && name.compare(name.size() - n, n, s) == 0; };
        if (ends_with("cls_token") || ends_with("positional_embedding")) p.group = 0;
        else if (name.find("adapooling_nets") != std::string::npos) p.group = one_d ? 2 : 1;
        else p.group = one_d ? 4 : 3;
    }
    h->index[kind][name] = (int)h->params[kind].size();
    h->params[kind].push_back(p);
    return p.offset;
}

// which fused IntegrationNetwork kernels this handle uses (geometry + measurement knobs): decided BEFORE the tables are built, because the packed GEMM
// operands of the launches they replace are then neither allocated nor re-packed every step
void set_fused_flags(dist_handle* h) {
    const dist_config& c = h->cfg;
    const int Ci = c.integration_dim, Ct = c.temporal_dim, C4 = h->C4;
    h->ig_on = dist_k_integ_eligible(c.dtype, Ci, C4, h->t, c.temporal_kernel) && h->Cf == Ci && (dist_knob("DIST_AMD_INTEG_FUSED", 1) != 0);
    h->ig_xhat = h->ig_on && (DIST_AB_KNOB("DIST_AMD_INTEG_XHAT", 1) != 0);
    h->ig_bwd = h->ig_xhat && (dist_knob("DIST_AMD_INTEG_BWD_FUSED", 1) != 0);
    // T2I (dist.py:68-86) formed in front of the fused forward instead of a GEMM + a cls-row kernel + a round trip of M' (alpha = 2, temporal width = C4)
    h->ig_t2i = h->ig_xhat && c.alpha == 2 && Ct == C4 && (DIST_AB_KNOB("DIST_AMD_INTEG_T2I", 1) != 0);
    // ... and I2T (dist.py:90-105) behind it: the next layer's temporal map leaves the same launch (the I2T GEMM and its second pass over M are gone)
    h->ig_i2t = h->ig_t2i && (DIST_AB_KNOB("DIST_AMD_INTEG_I2T", 1) != 0);
    // ... and the I2T backward (pair sums of dX_next, dM = dM' + dY Wi) behind the fused backward: the pair-sum kernel and the I2T data-gradient GEMM are gone
    h->ig_i2tb = h->ig_bwd && c.alpha == 2 && Ct == C4 && (DIST_AB_KNOB("DIST_AMD_INTEG_I2T_BWD", 1) != 0);
    // ... and the T2I backward (dp = (dX_next + conv^T(dM')) g'(p)) behind that: the T2I data-gradient GEMM leaves the chain as well
    h->ig_t2ib = h->ig_bwd && c.alpha == 2 && Ct == C4 && (DIST_AB_KNOB("DIST_AMD_INTEG_T2I_BWD", 1) != 0);
    h->keep_mid = (dist_knob("DIST_AMD_KEEP_MID", 0) != 0);     // debugging: M' of every layer stays readable (dist_debug_tensor "mid.i")
}

long pk_alloc(dist_handle* h, long elems) {
    h->packed_elems = (h->packed_elems + 127) & ~127L;
    const long o = h->packed_elems;
    h->packed_elems += elems;
    return o;
}

// registers the pack descriptors of one GEMM weight.  `style`: 0 linear [N][K]; 1 conv [Co][Ci][taps] with
// shift/spatial taps (data-gradient = per-tap transposes); 2 strided conv (data-gradient = plain transpose);
// 3 patch conv [Co][3][tp][P][P] (forward only, K padded to Kp); 4 [K][N] projection matrix used as x @ W.
void add_pack(dist_handle* h, Lin& l, int kind, int style, bool need_bwd, bool need_fwd = true) {
    PackDesc d;
    memset(&d, 0, sizeof(d));
    d.src_off = l.w; d.src_kind = kind; d.co = l.N; d.kin = l.K; d.kpad = l.K; d.inner = 1;
    if (style == 0) { d.s_co = l.K; d.s_tap = 0; d.s_outer = 1; }
    else if (style == 1 || style == 2) { d.s_co = (long)l.K * l.taps; d.s_tap = 1; d.s_outer = l.taps; }
    else if (style == 3) { d.kin = h->PP3; d.kpad = h->Kp; d.inner = h->PP3 / 3; d.s_co = (long)h->PP3 * l.taps; d.s_tap = h->PP3 / 3; d.s_outer = (long)(h->PP3 / 3) * l.taps; }
    else { d.s_co = 1; d.s_tap = 0; d.s_outer = l.N; }
    // forward layout [N][taps*kpad]  (need_fwd / need_bwd false: a fused kernel with its own operand order replaces the GEMM that read this copy)
    if (need_fwd) {
        d.layout = PACK_F; d.rows = l.N; d.cols = l.taps * d.kpad;
        l.pk.f = d.dst_off = pk_alloc(h, (long)d.rows * d.cols);
        h->descs.push_back(d);
    }
    if (!need_bwd) return;
    if (style == 1) { d.layout = PACK_B; d.rows = l.K; d.cols = l.taps * l.N; }
    else { d.layout = PACK_FT; d.rows = l.taps * d.kpad; d.cols = l.N; }
    l.pk.b = d.dst_off = pk_alloc(h, (long)d.rows * d.cols);
    h->descs.push_back(d);
}

Lin make_lin(dist_handle* h, int kind, const std::string& prefix, int N, int K, int taps, int style, bool need_bwd,
             std::initializer_list<int64_t> wdims, bool has_bias = true, const char* wname = "weight") {
    Lin l;
    l.N = N; l.K = K; l.taps = taps;
    l.w = add_param(h, kind, prefix + wname, wdims);
    if (has_bias) l.bias = add_param(h, kind, prefix + "bias", {N});
    add_pack(h, l, kind, style, need_bwd);
    return l;
}
LNp make_ln(dist_handle* h, int kind, const std::string& prefix, int C) {
    LNp l;
    l.C = C;
    l.w = add_param(h, kind, prefix + "weight", {C});
    l.b = add_param(h, kind, prefix + "bias", {C});
    return l;
}

// nn.MultiheadAttention parameters: in_proj_weight [3C, C] split into the q rows and the k,v rows
XAttn make_xattn(dist_handle* h, const std::string& prefix, int C) {
    XAttn x;
    const long w = add_param(h, 0, prefix + "attn.in_proj_weight", {3 * C, C});
    const long b = add_param(h, 0, prefix + "attn.in_proj_bias", {3 * C});
    x.q.N = C; x.q.K = C; x.q.w = w; x.q.bias = b;
    x.kv.N = 2 * C; x.kv.K = C; x.kv.w = w + (long)C * C; x.kv.bias = b + C;
    add_pack(h, x.q, 0, 0, true);
    add_pack(h, x.kv, 0, 0, true);
    x.out = make_lin(h, 0, prefix + "attn.out_proj.", C, C, 1, 0, true, {C, C});
    x.ln1 = make_ln(h, 0, prefix + "ln_1.", C);
    return x;
}

void build_tables(dist_handle* h) {
    const dist_config& c = h->cfg;
    const int d = c.width, Ci = c.integration_dim, Ct = c.temporal_dim, P = c.patch, C4 = h->C4, t = h->t, Ch = h->Ch, Cf = h->Cf;
    // ---- frozen visual.* (kind 1), OpenAI-CLIP names (reference clip.py:218-247) ----
    h->conv1.N = d; h->conv1.K = h->Kp; h->conv1.taps = 1;
    h->conv1.w = add_param(h, 1, "visual.conv1.weight", {d, 3, P, P});
    add_pack(h, h->conv1, 1, 3, false);
    h->class_emb = add_param(h, 1, "visual.class_embedding", {d});
    h->pos_emb = add_param(h, 1, "visual.positional_embedding", {h->L, d});
    h->ln_pre = make_ln(h, 1, "visual.ln_pre.", d);
    h->vit.resize(c.layers);
    for (int i = 0; i < c.layers; ++i) {
        VitLayer& v = h->vit[i];
        const std::string p = fmt("visual.transformer.resblocks.%d.", i);
        v.qkv = make_lin(h, 1, p + "attn.in_proj_", 3 * d, d, 1, 0, false, {3 * d, d});
        v.out = make_lin(h, 1, p + "attn.out_proj.", d, d, 1, 0, false, {d, d});
        v.ln1 = make_ln(h, 1, p + "ln_1.", d);
        v.ln2 = make_ln(h, 1, p + "ln_2.", d);
        v.fc = make_lin(h, 1, p + "mlp.c_fc.", 4 * d, d, 1, 0, false, {4 * d, d});
        v.proj = make_lin(h, 1, p + "mlp.c_proj.", d, 4 * d, 1, 0, false, {d, 4 * d});
        if (c.dtype == DIST_BF16) {
            v.pk_fold_qkv = pk_alloc(h, (long)3 * d * d);
            v.pk_fold_fc = pk_alloc(h, (long)4 * d * d);
        }
    }
    add_param(h, 1, "visual.ln_post.weight", {d});
    add_param(h, 1, "visual.ln_post.bias", {d});
    add_param(h, 1, "visual.proj", {d, c.embed_dim});
    const int ndesc_visual = (int)h->descs.size();

    // ---- trainable dist_net.* (kind 0) (reference dist.py:165-202) ----
    h->stem.N = Ct; h->stem.K = h->Kp; h->stem.taps = c.temporal_patch;
    h->stem.w = add_param(h, 0, "dist_net.temporal_stem.weight", {Ct, 3, c.temporal_patch, P, P});
    h->stem.bias = add_param(h, 0, "dist_net.temporal_stem.bias", {Ct});
    add_pack(h, h->stem, 0, 3, false);
    h->dl.resize(h->nsel);
    h->layer_begin.resize(h->nsel); h->layer_end.resize(h->nsel);
    for (int i = 0; i < h->nsel; ++i) {
        DistLayer& l = h->dl[i];
        h->layer_begin[i] = h->total[0];
        l.in_lin = make_lin(h, 0, fmt("dist_net.input_linears.%d.", i), Ci, d, 1, 0, false, {Ci, d});
        l.i2t = make_lin(h, 0, fmt("dist_net.integration2temporal_nets.%d.linear_fuse.", i), Ct, Ci, 1, 0, true, {Ct, Ci});
        l.cls_token = add_param(h, 0, fmt("dist_net.temporal2integration_nets.%d.cls_token", i), {1, 1, t, Ci});
        l.t2i.N = Ci; l.t2i.K = Ct; l.t2i.taps = c.alpha;
        l.t2i.w = add_param(h, 0, fmt("dist_net.temporal2integration_nets.%d.linear_fuse.weight", i), {Ci, Ct, c.alpha, 1, 1});
        l.t2i.bias = add_param(h, 0, fmt("dist_net.temporal2integration_nets.%d.linear_fuse.bias", i), {Ci});
        add_pack(h, l.t2i, 0, 2, true, !h->ig_t2i);
        std::string p = fmt("dist_net.temporal_nets.%d.", i);
        l.tn_fc1 = make_lin(h, 0, p + "temporal_net.c_fc1.", Ch, Ct, c.temporal_kernel, 1, true, {Ch, Ct, c.temporal_kernel, 1, 1});
        l.tn_fc2 = make_lin(h, 0, p + "temporal_net.c_fc2.", Ct, Ch, 9, 1, true, {Ct, Ch, 1, 3, 3});
        l.tn_ln = make_ln(h, 0, p + "ln.", Ct);
        p = fmt("dist_net.integration_nets.%d.", i);
        // ffn.c_fc and temporal_ffn.c_fc1 read the same (normalised) rows: their weights, and their biases, sit side by side in the flat buffers so that ONE
        // weight-gradient GEMM over [dzf | dh1] writes both gradients as a [Ci + C4][Ci] matrix (fused IntegrationNetwork backward)
        l.ffn_fc.N = Cf; l.ffn_fc.K = Ci; l.ffn_fc.taps = 1;
        l.tf_fc1.N = C4; l.tf_fc1.K = Ci; l.tf_fc1.taps = 1;
        l.ffn_fc.w = add_param(h, 0, p + "ffn.c_fc.weight", {Cf, Ci});
        l.tf_fc1.w = add_param(h, 0, p + "temporal_ffn.c_fc1.weight", {C4, Ci, 1, 1, 1});
        l.ffn_fc.bias = add_param(h, 0, p + "ffn.c_fc.bias", {Cf});
        l.tf_fc1.bias = add_param(h, 0, p + "temporal_ffn.c_fc1.bias", {C4});
        add_pack(h, l.ffn_fc, 0, 0, !h->ig_bwd, !h->ig_on);
        add_pack(h, l.tf_fc1, 0, 0, !h->ig_bwd, !h->ig_on);
        // (ffn.c_proj / temporal_ffn.c_proj are only ever multiplied as the side-by-side pair packed below: no copies of their own)
        l.ffn_proj.N = Ci; l.ffn_proj.K = Cf; l.ffn_proj.taps = 1;
        l.ffn_proj.w = add_param(h, 0, p + "ffn.c_proj.weight", {Ci, Cf}); l.ffn_proj.bias = add_param(h, 0, p + "ffn.c_proj.bias", {Ci});
        l.tf_fc2.N = C4; l.tf_fc2.K = C4; l.tf_fc2.taps = c.temporal_kernel;
        l.tf_fc2.w = add_param(h, 0, p + "temporal_ffn.c_fc2.weight", {C4, C4, c.temporal_kernel, 1, 1}); l.tf_fc2.bias = add_param(h, 0, p + "temporal_ffn.c_fc2.bias", {C4});
        add_pack(h, l.tf_fc2, 0, 1, !h->ig_bwd, !h->ig_on);
        l.tf_proj.N = Ci; l.tf_proj.K = C4; l.tf_proj.taps = 1;
        l.tf_proj.w = add_param(h, 0, p + "temporal_ffn.c_proj.weight", {Ci, C4, 1, 1, 1}); l.tf_proj.bias = add_param(h, 0, p + "temporal_ffn.c_proj.bias", {Ci});
        {   // R = [gelu(zf) | gelu(h2)] [W_ffn | W_tf]^T: the two projections are ONE GEMM over activations stored side by side
            // (forward), one data-gradient GEMM and one weight-gradient GEMM (backward)
            const int Cc = Cf + C4;
            PackDesc d;
            auto desc_of = [&](const Lin& lin, int layout) {
                memset(&d, 0, sizeof(d));
                d.src_off = lin.w; d.src_kind = 0; d.co = lin.N; d.kin = lin.K; d.kpad = lin.K; d.inner = 1;
                d.s_co = lin.K; d.s_tap = 0; d.s_outer = 1; d.layout = layout;
                if (layout == PACK_F) { d.rows = lin.N; d.cols = lin.K; } else { d.rows = lin.K; d.cols = lin.N; }
            };
            if (!h->ig_on) {
                l.pk_proj_f = pk_alloc(h, (long)Ci * Cc);
                desc_of(l.ffn_proj, PACK_F); d.dst_off = l.pk_proj_f; d.dpitch = Cc; h->descs.push_back(d);
                desc_of(l.tf_proj, PACK_F); d.dst_off = l.pk_proj_f + Cf; d.dpitch = Cc; h->descs.push_back(d);
            }
            if (!h->ig_bwd) {
                l.pk_proj_b = pk_alloc(h, (long)Cc * Ci);
                desc_of(l.ffn_proj, PACK_FT); d.dst_off = l.pk_proj_b; h->descs.push_back(d);
                desc_of(l.tf_proj, PACK_FT); d.dst_off = l.pk_proj_b + (long)Cf * Ci; h->descs.push_back(d);
            }
        }
        l.in_ln = make_ln(h, 0, p + "ln.", Ci);
        l.in_ln_t = make_ln(h, 0, p + "ln_temporal.", Ci);
        h->layer_end[i] = h->total[0];
    }
    h->tail_begin = h->total[0];
    h->ada.resize(c.ada_layers);
    for (int a = 0; a < c.ada_layers; ++a) {
        AdaLayer& A = h->ada[a];
        const std::string p = fmt("dist_net.adapooling_nets.%d.", a);
        A.pos = add_param(h, 0, p + "positional_embedding", {1, t, Ci});
        A.tm = make_xattn(h, p + "temporal_transformer.", Ci);
        A.sp = make_xattn(h, p + "spatial_transformer.", Ci);
        A.tm_fc = make_lin(h, 0, p + "output_map_cls_token.c_fc.", 4 * Ci, Ci, 1, 0, true, {4 * Ci, Ci});
        A.tm_proj = make_lin(h, 0, p + "output_map_cls_token.c_proj.", Ci, 4 * Ci, 1, 0, true, {Ci, 4 * Ci});
        A.sp_fc = make_lin(h, 0, p + "output_map_spatial_cls_token.c_fc.", 4 * Ci, Ci, 1, 0, true, {4 * Ci, Ci});
        A.sp_proj = make_lin(h, 0, p + "output_map_spatial_cls_token.c_proj.", Ci, 4 * Ci, 1, 0, true, {Ci, 4 * Ci});
        A.ln_tm = make_ln(h, 0, p + "ln_out_temp_cls_token.", Ci);
        A.ln_sp = make_ln(h, 0, p + "ln_out_spat_cls_token.", Ci);
    }
    h->cls_proj = make_lin(h, 0, "dist_net.proj_spatial_cls_token.", Ci, d, 1, 0, false, {Ci, d});
    h->ln_post = make_ln(h, 0, "dist_net.ln_post.", Ci);
    h->proj.N = c.embed_dim; h->proj.K = Ci;
    h->proj.w = add_param(h, 0, "dist_net.proj", {Ci, c.embed_dim});
    add_pack(h, h->proj, 0, 4, true);
    h->agg_cls = add_param(h, 0, "dist_net.aggregated_cls_token", {1, 1, Ci});
    h->agg_sp_cls = add_param(h, 0, "dist_net.aggregated_spatial_cls_token", {1, 1, Ci});

    // ---- pack launch table: one block per PACK_PER_BLOCK destination elements ----
    for (size_t di = 0; di < h->descs.size(); ++di) {
        if ((int)di == ndesc_visual) h->nblk_visual = (int)h->blk_desc.size();
        const PackDesc& pd = h->descs[di];
        const long total = (long)pd.rows * pd.cols;
        int nb = (int)((total + PACK_PER_BLOCK - 1) / PACK_PER_BLOCK);
        if (pd.layout == PACK_B) { const int rpb = pack_b_rows(pd.cols); nb = (pd.rows + rpb - 1) / rpb; }   // whole rows per block
        h->blk_first.push_back((int)h->blk_desc.size());
        for (int b = 0; b < nb; ++b) h->blk_desc.push_back((int)di);
    }
    size_t hdr = h->descs.size() * sizeof(PackDesc);
    hdr = (hdr + 255) & ~(size_t)255;
    hdr += h->blk_desc.size() * sizeof(int);
    hdr = (hdr + 255) & ~(size_t)255;
    hdr += h->blk_first.size() * sizeof(int);
    hdr = (hdr + 255) & ~(size_t)255;
    h->packed_hdr = hdr;
    h->packed_total = hdr + (size_t)(h->packed_elems + 128) * h->es;
}

// workspace carve-up; run with base == nullptr to size it
size_t layout_ws(dist_handle* h, char* base) {
    const dist_config& c = h->cfg;
    Arena a;
    a.base = base;
    const size_t es = h->es;
    const long b = c.batch;
    const long rowsX = b * c.frames * h->N, rowsS = b * h->t * h->L, rowsQ = b * h->t * h->N, bt = b * h->t;
    const int d = c.width, Ci = c.integration_dim, Ct = c.temporal_dim, C4 = h->C4, H = h->iheads, Ch = h->Ch, Cf = h->Cf;
    auto T_ = [&](long rows, long cols) { return a.take((size_t)rows * cols * es); };
    auto F_ = [&](long n) { return static_cast<float*>(a.take((size_t)n * sizeof(float))); };
    for (int k = 0; k < 2; ++k) h->slot[k].patches = T_(rowsX, h->Kp);
    h->x0 = T_(rowsS, d); h->xa = T_(rowsS, d); h->hbuf = T_(rowsS, d);
    h->qkv = T_(rowsS, 3 * d); h->att = T_(rowsS, d); h->mlp = T_(rowsS, 4 * d);
    h->lnstats = F_(2 * rowsS); h->lnstats2 = F_(2 * rowsS); h->lnstats3 = F_(2 * rowsS);
    h->lnpart = F_(2 * rowsS * ((c.width + 63) / 64));
    for (int i = 0; i < c.layers; ++i) {
        VitLayer& v = h->vit[i];
        v.cs_qkv = F_(3 * d); v.b_qkv = F_(3 * d); v.cs_fc = F_(4 * d); v.b_fc = F_(4 * d);
    }
    if (c.vit_fp8 && c.dtype == DIST_BF16) {
        auto B_ = [&](long n) { return static_cast<unsigned char*>(a.take((size_t)n)); };
        h->aq = B_(rowsS * 4 * d); h->sa = F_(rowsS);
        if (c.vit_fp8 & 16) { h->x8 = B_(rowsS * d); h->xa8 = B_(rowsS * d); h->f8_amax = F_(5 * c.layers); h->f8_scale = F_(5 * c.layers); }
        for (int i = 0; i < c.layers; ++i) {
            VitLayer& v = h->vit[i];
            v.q_qkv.q = B_((long)3 * d * d); v.q_qkv.s = F_(3 * d); v.cs8_qkv = F_(3 * d);
            v.q_out.q = B_((long)d * d); v.q_out.s = F_(d);
            v.q_fc.q = B_((long)4 * d * d); v.q_fc.s = F_(4 * d); v.cs8_fc = F_(4 * d);
            v.q_proj.q = B_((long)4 * d * d); v.q_proj.s = F_(d);
        }
    }
    for (int k = 0; k < 2; ++k) {
        h->slot[k].feat.resize(c.layers);
        for (int i = 0; i < c.layers; ++i) h->slot[k].feat[i] = T_(rowsS, d);
        h->slot[k].valid.assign(c.layers, 0);
    }
    h->patches = h->slot[h->cur].patches; h->feat = h->slot[h->cur].feat;
    h->lw.resize(h->nsel);
    for (int i = 0; i < h->nsel; ++i) {
        DistLayerWs& w = h->lw[i];
        w.X = T_(rowsX, Ct); w.U = T_(rowsX, Ct); w.z = T_(rowsX, Ch); w.V = T_(rowsX, Ch); w.p = T_(rowsX, Ct); w.Xp = T_(rowsX, Ct);
        w.M = T_(rowsS, Ci); w.Mp = T_(rowsS, Ci); w.Na = T_(rowsS, Ci); w.Nb = T_(rowsS, Ci);
        // [zf | h2] and [hf | g2] = their activations live side by side in rows of Ci + C4 elements (see pk_proj_f)
        w.zf = T_(rowsS, Cf + C4); w.hf = T_(rowsS, Cf + C4);
        w.h2 = w.zf ? static_cast<char*>(w.zf) + (size_t)Cf * es : nullptr;
        w.g2 = w.hf ? static_cast<char*>(w.hf) + (size_t)Cf * es : nullptr;
        w.h1 = T_(rowsS, C4); w.R = T_(rowsS, Ci);
        w.tn_mean = F_(rowsX); w.tn_rstd = F_(rowsX); w.in_mean = F_(rowsS); w.in_rstd = F_(rowsS);
    }
    h->Xlast = T_(rowsX, Ct);
    h->aw.resize(c.ada_layers);
    h->sbuf.resize(c.ada_layers + 1); h->ubuf.resize(c.ada_layers + 1);
    for (int k = 0; k <= c.ada_layers; ++k) { h->sbuf[k] = T_(bt, Ci); h->ubuf[k] = T_(b, Ci); }
    for (int k = 0; k < c.ada_layers; ++k) {
        AdaWs& w = h->aw[k];
        w.kn = T_(rowsS, Ci); w.kv = T_(rowsS, 2 * Ci); w.qn = T_(bt, Ci); w.q = T_(bt, Ci); w.o = T_(bt, Ci); w.s1 = T_(bt, Ci);
        w.sn = T_(bt, Ci); w.zs = T_(bt, 4 * Ci); w.hs = T_(bt, 4 * Ci); w.c = T_(bt, Ci); w.kn2 = T_(bt, Ci); w.kv2 = T_(bt, 2 * Ci);
        w.qn2 = T_(b, Ci); w.q2 = T_(b, Ci); w.o2 = T_(b, Ci); w.u1 = T_(b, Ci); w.un = T_(b, Ci); w.zu = T_(b, 4 * Ci); w.hu = T_(b, 4 * Ci);
        w.kn_mean = F_(rowsS); w.kn_rstd = F_(rowsS); w.qn_mean = F_(bt); w.qn_rstd = F_(bt); w.s1_mean = F_(bt); w.s1_rstd = F_(bt);
        w.kn2_mean = F_(bt); w.kn2_rstd = F_(bt); w.qn2_mean = F_(b); w.qn2_rstd = F_(b); w.u1_mean = F_(b); w.u1_rstd = F_(b);
        w.probs = F_(bt * H * h->L); w.probs2 = F_(b * H * h->t);
    }
    h->Fz = T_(rowsS, Ci); h->mean_cls = T_(b, d); h->ysum = T_(b, Ci); h->zpost = T_(b, Ci); h->v = T_(b, c.embed_dim);
    h->y_mean = F_(b); h->y_rstd = F_(b); h->logits = F_(b * c.num_classes); h->dlogits = F_(b * c.num_classes); h->loss = F_(4);
    // backward scratch
    h->dR = T_(rowsS, Ci); h->dkv = T_(rowsS, 2 * Ci); h->dkn = T_(rowsS, Ci);
    h->ln_partial_elems = dist_op_layernorm_bwd_scratch(rowsS > rowsX ? rowsS : rowsX, Ci > Ct ? Ci : Ct);
    h->ln_partial = F_(h->ln_partial_elems);
    h->tnb_scratch_elems = dist_op_temporal_net_bwd_scratch((int)b, c.frames, Ct);
    h->tnb_scratch = F_(h->tnb_scratch_elems * h->nsel);         // one partial table per DiST layer
    if (h->ig_on) {
        for (int i = 0; i < h->nsel; ++i) {
            DistLayer& l = h->dl[i];
            l.ig_W1 = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 0) * 2); l.ig_W2 = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 1) * 2);
            l.ig_W3 = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 2) * 2);
            if (h->ig_t2i) l.ig_Wt = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 6) * 2);
            if (h->ig_i2t) l.ig_Wi = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 7) * 2);
            if (h->ig_i2tb) l.ig_W4 = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 8) * 2);
            if (h->ig_t2ib) l.ig_W5 = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 9) * 2);
            if (h->ig_bwd) {
                l.ig_B1 = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 0) * 2); l.ig_B2 = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 1) * 2);
                l.ig_B3 = a.take((size_t)dist_op_integration_pack_elems(Ci, C4, 2) * 2);
            }
            l.ig_b1 = F_(dist_op_integration_pack_elems(Ci, C4, 3)); l.ig_b2 = F_(dist_op_integration_pack_elems(Ci, C4, 4)); l.ig_b3 = F_(dist_op_integration_pack_elems(Ci, C4, 5));
        }
        h->ig_descs = a.take((size_t)dist_k_integ_pack_desc_bytes() * h->nsel);
    }
    h->tn_partial_elems = 16l << 20;                         // 64 MB each: 256 partial tiles of 192 x 256 (the LDS-DMA weight-gradient kernel) + slack
    for (int k = 0; k < 3; ++k) h->tn_partial[k] = F_(h->tn_partial_elems);
    if (h->ig_xhat) { h->ig_gscratch_elems = (long)(Ci + C4) * Ci + (Ci + C4); h->ig_gscratch = F_(h->ig_gscratch_elems * h->nsel); }
    for (int k = 0; k < 2; ++k) {
        dist_handle::BwdSet& q = h->bs[k];
        q.dMp = T_(rowsS, Ci); q.dM = T_(rowsS, Ci); q.dXp = T_(rowsX, Ct); q.dp = T_(rowsX, Ct); q.dXo = T_(rowsX, Ct);
        q.dz = T_(rowsX, Ch); q.dU = T_(rowsX, Ct); q.dY = T_(rowsQ, Ct); q.dh1 = T_(rowsS, C4);
        q.dzf = T_(rowsS, Cf + C4);                      // [dzf | dh2], same side-by-side rows
        q.dh2 = q.dzf ? static_cast<char*>(q.dzf) + (size_t)Cf * es : nullptr;
        q.dNa = T_(rowsS, Ci); q.dNb = T_(rowsS, Ci);
        q.dcat = T_(rowsS, Ci + 2 * C4);
    }
    h->dv = T_(b, c.embed_dim); h->dzp = T_(b, Ci); h->dy = T_(b, Ci); h->du = T_(b, Ci); h->ds = T_(bt, Ci); h->dc = T_(bt, Ci);
    h->dzu = T_(bt, 4 * Ci); h->dun = T_(bt, Ci); h->do2 = T_(b, Ci); h->dq2 = T_(b, Ci); h->dkv2 = T_(bt, 2 * Ci); h->dqn2 = T_(b, Ci);
    h->dkn2 = T_(bt, Ci); h->dzs = T_(bt, 4 * Ci); h->dsn = T_(bt, Ci); h->do_ = T_(bt, Ci); h->dq = T_(bt, Ci); h->dqn = T_(bt, Ci);
    return a.off + 256;
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

dist_rowmap RM(int mode = DIST_RM_PLAIN, int p0 = 0, int p1 = 0, int sign = 1) { return dist_rowmap{mode, p0, p1, sign}; }
dist_outmap OM(int mode = DIST_OM_PLAIN, int p0 = 0, int p1 = 0, int p2 = 0) { return dist_outmap{mode, p0, p1, p2}; }

// C (and/or C2) = epi(A[amap] . W^T): thin positional wrapper over dist_op_gemm_nt
int gemm(const Ctx& c, const void* A, int lda, const void* W, long M, int N, int K, int taps, void* C, int ldc,
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
int gemm_lnfold(const Ctx& c, const void* A, int lda, const void* Wf, long M, int N, int K, void* C, int ldc, const float* biasf,
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
int gemm_fp8(const Ctx& c, const unsigned char* Aq, const float* sa, bool sa_scalar, const VitLayer::Fp8W& W, long M, int N, int K, void* C, int ldc,
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
bool fp8_shape_ok(const Ctx& c, long M, int N, int K) {
    dist_gemm_args g;
    memset(&g, 0, sizeof(g));
    void* nz = reinterpret_cast<void*>(16);
    g.A = nz; g.B = nz; g.C = nz; g.a_scale = static_cast<const float*>(nz); g.b_scale = static_cast<const float*>(nz);
    g.M = M; g.N = N; g.K = K; g.taps = 1; g.lda = K; g.ldb = K; g.ldc = g.ldc2 = g.ldres = g.ldaux = N;
    g.amap = RM(); g.omap = OM(); g.flags = DIST_EPI_FP8; g.dtype = DIST_BF16;
    return dist_k_gemm_fast_eligible(&g);
}
// can C = A W^T + bias + res (plain maps) leave DIST_EPI_ROWSTATS partials, i.e. does the LDS-DMA kernel take this shape?
bool rowstats_ok(const Ctx& c, long M, int N, int K) {
    dist_gemm_args g;
    memset(&g, 0, sizeof(g));
    void* nz = reinterpret_cast<void*>(16);
    g.A = nz; g.B = nz; g.C = nz; g.res = nz; g.bias = static_cast<const float*>(nz); g.rowstats = static_cast<float*>(nz);
    g.M = M; g.N = N; g.K = K; g.taps = 1; g.lda = K; g.ldb = K; g.ldc = g.ldc2 = g.ldres = g.ldaux = N;
    g.amap = RM(); g.omap = OM(); g.flags = DIST_EPI_BIAS | DIST_EPI_RES | DIST_EPI_ROWSTATS; g.dtype = c.dtype;
    return N % 64 == 0 && dist_k_gemm_fast_eligible(&g);
}
int wgrad(const Ctx& c, const Lin& l, const void* dY, int ld_dy, const void* X, int ldx, long M,
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
int wgrad_pair(const Ctx& c, const Lin& l1, const Lin& l2, const void* dY, int ld_dy, const void* X12, int ldx, long M) {
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
int bgrad(const Ctx& c, long bias_off, const void* dY, long rows, int C, dist_rowmap m = RM()) {
    return dist_op_colsum(dY, c.gr(bias_off), rows, C, C, m, c.dtype, c.s);
}
int ln_fwd(const Ctx& c, const float* wbase, const LNp& l, const void* x, void* y, long rows, float* mean, float* rstd,
           const LNp* l2 = nullptr, void* y2 = nullptr, const float* addend = nullptr, int period = 0) {
    dist_ln_args a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.y = y; a.y2 = y2; a.w = wbase + l.w; a.b = wbase + l.b;
    if (l2) { a.w2 = wbase + l2->w; a.b2 = wbase + l2->b; }
    a.addend = addend; a.addend_period = period; a.mean = mean; a.rstd = rstd;
    a.rows = rows; a.C = l.C; a.dtype = c.dtype; a.eps = 1e-5f;
    return dist_op_layernorm(&a, c.s);
}
int ln_bwd(const Ctx& c, const LNp& l, const void* x, const float* mean, const float* rstd, const void* dy, void* dx, bool accumulate,
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

}  // namespace

// =============================================================================================================
extern "C" const char* dist_strerror(int code) {
    switch (code) {
        case DIST_OK: return "ok";
        case DIST_ERR_ARG: return "invalid argument or unsupported shape";
        case DIST_ERR_STATE: return "invalid call order";
        case DIST_ERR_WORKSPACE: return "workspace too small or not bound";
        case DIST_ERR_UNBOUND: return "weights/buffers not bound";
        default: return code <= -1000 ? hipGetErrorString((hipError_t)(-code - 1000)) : "unknown error";
    }
}
// 5: dist_gemm_args.rowstats, marks, mixup, evaluation side; 6: fp8 operands (a_scale / b_scale, DIST_EPI_FP8);
// 7: dist_config.vit_fp8 and dist_gemm_args.C8 / ldc8 / out8_scale / out8_amax (both structs grew after 6 without a bump), vit_fp8 range-checked.
// Rule: EVERY change of a public struct's layout bumps this number (dist_amd/lib.py and tests/test_abi_and_host.py pin it).
// 8: dist_ln_bwd_args.partial / partial_elems (two-phase LayerNorm parameter gradients), dist_tnet_args / dist_tnet_bwd_args, dist_set_inference.
extern "C" int dist_abi_version(void) { return DIST_ABI_VERSION; }
extern "C" int dist_measure_build(void) {
#ifdef DIST_AMD_MEASURE
    return 1;
#else
    return 0;
#endif
}
extern "C" int dist_abi_sizeof(const char* n) {
    if (!n) return -1;
#define DIST_SZ(T) if (!strcmp(n, #T)) return (int)sizeof(T)
    DIST_SZ(dist_gemm_args); DIST_SZ(dist_gemm_tn_args); DIST_SZ(dist_ln_args); DIST_SZ(dist_ln_bwd_args);
    DIST_SZ(dist_adamw_seg); DIST_SZ(dist_config); DIST_SZ(dist_rowmap); DIST_SZ(dist_outmap); DIST_SZ(dist_tnet_args); DIST_SZ(dist_tnet_bwd_args); DIST_SZ(dist_integ_args); DIST_SZ(dist_integ_pack_args); DIST_SZ(dist_integ_unfold_args); DIST_SZ(dist_integ_bwd_args);
#undef DIST_SZ
    return -1;
}

extern "C" void dist_destroy(dist_handle* h);
extern "C" int dist_create(const dist_config* cfg, dist_handle** out) {
    if (!cfg || !out) return DIST_ERR_ARG;
    const dist_config& c = *cfg;
    if (c.dtype != DIST_F32 && c.dtype != DIST_BF16) return DIST_ERR_ARG;
    if (c.batch <= 0 || c.frames <= 0 || c.alpha <= 0 || c.frames % c.alpha || c.patch <= 0 || c.resolution % c.patch) return DIST_ERR_ARG;
    if (c.width % 64 || c.integration_dim % 64 || c.temporal_dim % 8 || c.layers <= 0 || c.ada_layers < 0) return DIST_ERR_ARG;
    if (c.int_temporal_div <= 0 || c.integration_dim % c.int_temporal_div || (c.integration_dim / c.int_temporal_div) % 8) return DIST_ERR_ARG;
    if (c.temporal_kernel % 2 == 0 || c.temporal_patch % 2 == 0 || c.num_classes <= 0 || c.embed_dim % 8 || c.embed_dim > 1024) return DIST_ERR_ARG;
    if (c.width > 1024 || c.integration_dim > 1024) return DIST_ERR_ARG;
    if (c.temporal_hidden < 0 || c.temporal_hidden % 8 || c.temporal_hidden > 1024 || c.integration_hidden < 0 || c.integration_hidden % 8 || c.integration_hidden > 4096) return DIST_ERR_ARG;
    if (c.layers > 32 || (c.layers < 32 && ((unsigned)c.selected_mask >> c.layers) != 0)) return DIST_ERR_ARG;      // SELECTED_LAYERS beyond the ViT's blocks
    // vit_fp8: bits 1 | 2 | 4 | 8 select GEMMs, 16 needs all four; bf16 engines only (a binding built against the 17-field struct of
    // ABI <= 6 hands over 4 bytes of garbage here: refuse instead of silently switching the frozen ViT to e4m3)
    if ((c.vit_fp8 & ~31) || (c.vit_fp8 && c.dtype != DIST_BF16) || ((c.vit_fp8 & 16) && (c.vit_fp8 & 15) != 15)) return DIST_ERR_ARG;
    dist_handle* h = new (std::nothrow) dist_handle();
    if (!h) return DIST_ERR_ARG;
    h->cfg = c;
    for (int i = 0; i < c.layers; ++i) if (!c.selected_mask || ((unsigned)c.selected_mask >> i) & 1u) h->sel.push_back(i);
    h->nsel = (int)h->sel.size();
    h->Ch = c.temporal_hidden > 0 ? c.temporal_hidden : c.temporal_dim;
    h->Cf = c.integration_hidden > 0 ? c.integration_hidden : c.integration_dim;
    h->es = c.dtype == DIST_BF16 ? 2 : 4;
    h->G = c.resolution / c.patch; h->N = h->G * h->G; h->L = h->N + 1; h->t = c.frames / c.alpha;
    h->heads = c.width / 64; h->iheads = c.integration_dim / 64; h->C4 = c.integration_dim / c.int_temporal_div;
    h->PP3 = 3 * c.patch * c.patch; h->Kp = (h->PP3 + 7) / 8 * 8;
    set_fused_flags(h);
    build_tables(h);
    h->ws_bytes = layout_ws(h, nullptr);
    h->serial = dist_knob("DIST_AMD_SERIAL", h->serial);
    h->skip = dist_measure_knob("DIST_AMD_SKIP", h->skip);
    h->dummy = dist_measure_knob("DIST_AMD_DUMMY", h->dummy);
    h->dummy_reps = dist_measure_knob("DIST_AMD_DUMMY_REPS", h->dummy_reps);
    *out = h;
    return DIST_OK;
}
// side streams + events of the multi-stream forward / backward: created by dist_bind (the first call that needs a
// device), so dist_create and the parameter tables stay host-only
static int ensure_streams(dist_handle* h) {
    if (h->side) return DIST_OK;
    const dist_config& c = h->cfg;
    // the side streams carry work that is off the critical path (branch forward beside the ViT, weight gradients beside the
    // data-gradient chain): lowest priority, so the caller's stream gets freed CUs first (DIST_AMD_SIDE_PRIO=0: default priority)
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    const int prio = DIST_AB_KNOB("DIST_AMD_SIDE_PRIO", 1) == 0 ? 0 : least;
    // the frozen ViT's stream (DIST_AMD_PF_PRIO in the timing-only library: 0 = default priority, 2 = the highest)
    const int pf_knob = DIST_AB_KNOB("DIST_AMD_PF_PRIO", 1);
    const int pf_prio = pf_knob == 0 ? 0 : (pf_knob == 2 ? greatest : least);
    bool ok = hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, prio) == hipSuccess &&
              hipStreamCreateWithPriority(&h->side2, hipStreamNonBlocking, prio) == hipSuccess &&
              hipStreamCreateWithPriority(&h->pf, hipStreamNonBlocking, pf_prio) == hipSuccess;
    // the optional second data-gradient chain of the backward (DIST_AMD_BWD_TCHAIN=1, a measurement): created only when asked for - an extra
    // stream that merely EXISTS beside the gradient reducer's cost 4.7 ms per step (tests/test_rccl_gpu.py: 23.2 vs 18.5 ms)
    if (DIST_AB_KNOB("DIST_AMD_BWD_TCHAIN", 0) == 1)
        ok = ok && hipStreamCreateWithPriority(&h->chain2, hipStreamNonBlocking, 0) == hipSuccess;
    auto mk = [&](hipEvent_t& e) { ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; };
    h->ev_a.resize(10 * c.layers + 4); h->ev_b_dr.resize(c.layers); h->ev_b_done.resize(c.layers);
    for (auto& e : h->ev_a) mk(e);
    for (auto& e : h->ev_b_dr) mk(e);
    for (auto& e : h->ev_b_done) mk(e);
    h->ev_dmp.resize(c.layers); h->ev_dx.resize(c.layers);
    for (auto& e : h->ev_dmp) mk(e);
    for (auto& e : h->ev_dx) mk(e);
    mk(h->ev_c2);
    for (int k = 0; k < 2; ++k) {
        h->slot[k].ev_feat.resize(c.layers + 1);
        for (auto& e : h->slot[k].ev_feat) mk(e);
        mk(h->slot[k].ev_pre);
    }
    h->ev_feat = h->slot[h->cur].ev_feat; h->ev_pre = h->slot[h->cur].ev_pre;
    mk(h->ev_join); mk(h->ev_b2); mk(h->ev_vit_done); mk(h->ev_after); mk(h->ev_bpre);
    return ok ? DIST_OK : DIST_ERR_STATE;
}

extern "C" void dist_destroy(dist_handle* h) {
    if (!h) return;
    for (hipEvent_t e : h->prof_ev) hipEventDestroy(e);
    for (hipEvent_t& e : h->mark_ev) if (e) { hipEventDestroy(e); e = nullptr; }
    for (hipEvent_t e : h->ev_a) if (e) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_b_dr) if (e) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_b_done) if (e) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_dmp) if (e) hipEventDestroy(e);
    for (hipEvent_t e : h->ev_dx) if (e) hipEventDestroy(e);
    if (h->ev_c2) hipEventDestroy(h->ev_c2);
    if (h->chain2) hipStreamDestroy(h->chain2);
    for (int k = 0; k < 2; ++k) {
        for (hipEvent_t e : h->slot[k].ev_feat) if (e) hipEventDestroy(e);
        if (h->slot[k].ev_pre) hipEventDestroy(h->slot[k].ev_pre);
    }
    for (hipEvent_t e : {h->ev_join, h->ev_b2, h->ev_vit_done, h->ev_after, h->ev_bpre}) if (e) hipEventDestroy(e);
    if (h->side) hipStreamDestroy(h->side);
    if (h->side2) hipStreamDestroy(h->side2);
    if (h->pf) hipStreamDestroy(h->pf);
    delete h;
}
extern "C" const char* dist_last_error(const dist_handle* h) { return h ? h->err : ""; }

extern "C" int dist_param_count(const dist_handle* h, int kind) { return (h && (kind == 0 || kind == 1)) ? (int)h->params[kind].size() : 0; }
static const Param* get_param(const dist_handle* h, int kind, int i) {
    if (!h || (kind != 0 && kind != 1) || i < 0 || i >= (int)h->params[kind].size()) return nullptr;
    return &h->params[kind][i];
}
extern "C" const char* dist_param_name(const dist_handle* h, int kind, int i) { const Param* p = get_param(h, kind, i); return p ? p->name.c_str() : ""; }
extern "C" int dist_param_ndim(const dist_handle* h, int kind, int i) { const Param* p = get_param(h, kind, i); return p ? p->ndim : 0; }
extern "C" int64_t dist_param_dim(const dist_handle* h, int kind, int i, int d) { const Param* p = get_param(h, kind, i); return (p && d >= 0 && d < p->ndim) ? p->dim[d] : 0; }
extern "C" int64_t dist_param_offset(const dist_handle* h, int kind, int i) { const Param* p = get_param(h, kind, i); return p ? p->offset : -1; }
extern "C" int64_t dist_param_total(const dist_handle* h, int kind) { return (h && (kind == 0 || kind == 1)) ? h->total[kind] : 0; }
extern "C" int dist_param_group(const dist_handle* h, int i) { const Param* p = get_param(h, 0, i); return p ? p->group : -1; }
extern "C" size_t dist_workspace_bytes(const dist_handle* h) { return h ? h->ws_bytes : 0; }
extern "C" size_t dist_packed_bytes(const dist_handle* h) { return h ? h->packed_total : 0; }

extern "C" int dist_bind(dist_handle* h, float* theta, float* grads, const float* visual, float* logit_scale, float* dlogit_scale,
                         void* packed, void* workspace) {
    if (!h) return DIST_ERR_ARG;
    if (!theta || !visual || !logit_scale || !packed || !workspace) return fail(h, DIST_ERR_UNBOUND, "dist_bind: null buffer");
    if (ensure_streams(h) != DIST_OK) return fail(h, DIST_ERR_STATE, "dist_bind: cannot create the side streams / events (no HIP device?)");
    h->theta = theta; h->grads = grads; h->visual = visual; h->logit_scale = logit_scale; h->dlogit_scale = dlogit_scale;
    h->packed = static_cast<char*>(packed); h->ws = static_cast<char*>(workspace);
    layout_ws(h, h->ws);
    // pack tables -> header of the packed buffer (pageable host memory: the copies complete before returning)
    char* p = h->packed;
    HIP_CHECK_RET(hipMemcpy(p, h->descs.data(), h->descs.size() * sizeof(PackDesc), hipMemcpyHostToDevice));
    size_t off = (h->descs.size() * sizeof(PackDesc) + 255) & ~(size_t)255;
    HIP_CHECK_RET(hipMemcpy(p + off, h->blk_desc.data(), h->blk_desc.size() * sizeof(int), hipMemcpyHostToDevice));
    off = (off + h->blk_desc.size() * sizeof(int) + 255) & ~(size_t)255;
    HIP_CHECK_RET(hipMemcpy(p + off, h->blk_first.data(), h->blk_first.size() * sizeof(int), hipMemcpyHostToDevice));
    if (h->ig_on) {                                        // the all-layers pack of the fused IntegrationNetwork forward reads its descriptors from the workspace
        const size_t db = (size_t)dist_k_integ_pack_desc_bytes();
        std::vector<char> host(db * h->dl.size());
        for (size_t i = 0; i < h->dl.size(); ++i) {
            const DistLayer& l = h->dl[i];
            dist_integ_pack_args a;
            memset(&a, 0, sizeof(a));
            a.ffn_fc_w = theta + l.ffn_fc.w; a.ffn_fc_b = theta + l.ffn_fc.bias; a.ln_w = theta + l.in_ln.w; a.ln_b = theta + l.in_ln.b;
            a.tf_fc1_w = theta + l.tf_fc1.w; a.tf_fc1_b = theta + l.tf_fc1.bias; a.ln_t_w = theta + l.in_ln_t.w; a.ln_t_b = theta + l.in_ln_t.b;
            a.tf_fc2_w = theta + l.tf_fc2.w; a.tf_fc2_b = theta + l.tf_fc2.bias;
            a.ffn_proj_w = theta + l.ffn_proj.w; a.ffn_proj_b = theta + l.ffn_proj.bias; a.tf_proj_w = theta + l.tf_proj.w; a.tf_proj_b = theta + l.tf_proj.bias;
            a.W1 = l.ig_W1; a.W2 = l.ig_W2; a.W3 = l.ig_W3; a.b1 = l.ig_b1; a.b2 = l.ig_b2; a.b3 = l.ig_b3;
            a.B1 = l.ig_B1; a.B2 = l.ig_B2; a.B3 = l.ig_B3;
            if (h->ig_t2i || h->ig_t2ib) { a.t2i_w = theta + l.t2i.w; a.Wt = l.ig_Wt; a.W5 = l.ig_W5; }
            if (h->ig_i2t || h->ig_i2tb) { a.i2t_w = theta + l.i2t.w; a.Wi = l.ig_Wi; a.W4 = l.ig_W4; }
            a.Ci = h->cfg.integration_dim; a.C4 = h->C4;
            dist_k_integ_pack_desc(&a, host.data() + i * db);
        }
        HIP_CHECK_RET(hipMemcpy(h->ig_descs, host.data(), host.size(), hipMemcpyHostToDevice));
    }
    h->fwd_b = h->branch_b = 0;
    h->slot[0].b = h->slot[1].b = 0; h->slot[0].prefetched = h->slot[1].prefetched = false;
    h->use_slot(0);
    return DIST_OK;
}

extern "C" int dist_pack_weights(dist_handle* h, int what, void* stream) {
    if (!h) return DIST_ERR_ARG;
    if (!h->packed) return fail(h, DIST_ERR_UNBOUND, "dist_pack_weights before dist_bind");
    const PackDesc* descs = reinterpret_cast<const PackDesc*>(h->packed);
    size_t off = (h->descs.size() * sizeof(PackDesc) + 255) & ~(size_t)255;
    const int* blk_desc = reinterpret_cast<const int*>(h->packed + off);
    off = (off + h->blk_desc.size() * sizeof(int) + 255) & ~(size_t)255;
    const int* blk_first = reinterpret_cast<const int*>(h->packed + off);
    const int nblk = (int)h->blk_desc.size();
    hipStream_t s = static_cast<hipStream_t>(stream);
    void* dst = h->packed + h->packed_hdr;
    int first = 0, count = nblk;
    if (what == 1) count = h->nblk_visual;
    else if (what == 2) { first = h->nblk_visual; count = nblk - h->nblk_visual; }
    else if (what != 3) return fail(h, DIST_ERR_ARG, "dist_pack_weights: what must be 1, 2 or 3");
    RUN(dist_k_pack(descs, blk_desc, blk_first, first, count, h->theta, h->visual, dst, h->cfg.dtype, s));
    if ((what & 2) && h->ig_on) RUN(dist_k_integ_pack(h->ig_descs, nullptr, h->nsel, h->cfg.integration_dim, h->C4, s));
    // frozen ViT: ln_1 -> attn.in_proj and ln_2 -> mlp.c_fc folded (W diag(gamma), column sums, folded biases); DIST_AMD_LNFOLD=0: off
    static const bool fold_on = (dist_knob("DIST_AMD_LNFOLD", 1) != 0);
    // (only a pack of the frozen weights touches the fold: the per-step re-pack of the trainable weights, what = 2, used to reset
    // vit_fold and every ViT pass after the first optimizer step ran the unfolded LayerNorm + GEMM)
    if (what & 1) h->vit_fold = false;
    if ((what & 1) && h->cfg.dtype == DIST_BF16 && fold_on && h->vit.size() && h->vit[0].cs_qkv) {
        const int d = h->cfg.width;
        for (VitLayer& v : h->vit) {
            char* base = h->packed + h->packed_hdr;
            RUN(dist_op_ln_fold(h->visual + v.qkv.w, h->visual + v.qkv.bias, h->visual + v.ln1.w, h->visual + v.ln1.b,
                                base + (size_t)v.pk_fold_qkv * h->es, v.cs_qkv, v.b_qkv, 3 * d, d, s));
            RUN(dist_op_ln_fold(h->visual + v.fc.w, h->visual + v.fc.bias, h->visual + v.ln2.w, h->visual + v.ln2.b,
                                base + (size_t)v.pk_fold_fc * h->es, v.cs_fc, v.b_fc, 4 * d, d, s));
        }
        h->vit_fold = true;
        if (h->cfg.vit_fp8 && h->aq) {
            h->f8_passes = 0;                              // new weights: the next pass calibrates the per-tensor scales again
            if (h->f8_amax) HIP_CHECK_RET(hipMemsetAsync(h->f8_amax, 0, sizeof(float) * 5 * h->cfg.layers, s));
            // fp8 frozen spatial branch: per-output-channel e4m3 copies of the (folded) GEMM weights; the fold's mean term uses the column
            // sums of the weights the MFMA really multiplies by
            for (VitLayer& v : h->vit) {
                char* base = h->packed + h->packed_hdr;
                RUN(dist_op_quant_rows_fp8(base + (size_t)v.pk_fold_qkv * h->es, DIST_BF16, 3 * d, d, d, v.q_qkv.q, d, v.q_qkv.s, s));
                RUN(dist_op_fp8_rowsum(v.q_qkv.q, v.q_qkv.s, 3 * d, d, d, v.cs8_qkv, s));
                RUN(dist_op_quant_rows_fp8(base + (size_t)v.out.pk.f * h->es, DIST_BF16, d, d, d, v.q_out.q, d, v.q_out.s, s));
                RUN(dist_op_quant_rows_fp8(base + (size_t)v.pk_fold_fc * h->es, DIST_BF16, 4 * d, d, d, v.q_fc.q, d, v.q_fc.s, s));
                RUN(dist_op_fp8_rowsum(v.q_fc.q, v.q_fc.s, 4 * d, d, d, v.cs8_fc, s));
                RUN(dist_op_quant_rows_fp8(base + (size_t)v.proj.pk.f * h->es, DIST_BF16, d, 4 * d, 4 * d, v.q_proj.q, 4 * d, v.q_proj.s, s));
            }
        }
    }
    if (what == 2) mark(h, DIST_MARK_STEP_END, s);
    return DIST_OK;
}

// -------------------------------------------------------------------------------------------------------------
// The frozen ViT of one batch into feature slot `k` on `stream`.  `after` (when ordered_after): a stream whose already queued work must
// finish first (pipelined use: the backward of the batch that last used this slot, the weight re-pack).
static int vit_forward_slot(dist_handle* h, const float* video, int b, int k, void* stream, bool ordered_after, void* after, int l0, int l1) {
    const dist_config& c = h->cfg;
    Ctx x{h, static_cast<hipStream_t>(stream), c.dtype};
    dist_handle::FeatSlot& S = h->slot[k];
    const int d = c.width, N = h->N, L = h->L;
    const long rowsS = (long)b * h->t * L, rowsQ = (long)b * h->t * N;

    if (ordered_after && after != stream) {
        HIP_CHECK_RET(hipEventRecord(h->ev_after, static_cast<hipStream_t>(after)));
        HIP_CHECK_RET(hipStreamWaitEvent(x.s, h->ev_after, 0));
    }
    const void* xin = h->x0;
    if (l0 == 0) {
        // the ViT-internal scratch (x0, xa, hbuf, qkv, att, mlp) exists once: consecutive ViT passes are ordered, whatever their streams
        if (h->vit_ran) HIP_CHECK_RET(hipStreamWaitEvent(x.s, h->ev_vit_done, 0));
        HIP_CHECK_RET(hipEventRecord(S.ev_pre, x.s));                   // everything queued before this pass (re-pack, previous step)
        mark(h, DIST_MARK_VIT_BEGIN, x.s);
        RUN(dist_op_patchify(video, S.patches, b, c.frames, c.resolution, c.resolution, c.patch, c.dtype, stream));
        HIP_CHECK_RET(hipEventRecord(S.ev_feat[c.layers], x.s));       // patch rows ready (temporal stem input)
        // patch embedding of the frames k = alpha*j only (the reference embeds all T frames and drops the
        // others at clip.py:284); rows land behind their frame's cls row
        RUN(gemm(x, S.patches, h->Kp, x.pk(h->conv1.pk.f), rowsQ, d, h->Kp, 1, h->xa, d, nullptr, nullptr, nullptr, nullptr,
                 RM(DIST_RM_STRIDED, c.alpha, N), OM(DIST_OM_INSERTCLS, N)));
        RUN(dist_k_cls_rows(h->xa, nullptr, x.vs(h->class_emb), b * h->t, L, d, 1, c.dtype, x.s));
        RUN(ln_fwd(x, h->visual, h->ln_pre, h->xa, h->x0, rowsS, nullptr, nullptr, nullptr, nullptr, x.vs(h->pos_emb), L));
    } else {
        xin = S.feat[l0 - 1];
    }
    // Row statistics of the residual stream from the GEMM that writes it (DIST_EPI_ROWSTATS): `out` leaves the partials ln_2 / c_fc
    // need, `proj` those of the next block's ln_1 / in_proj; dist_op_ln_stats_from_partials (5 MB in, 0.4 MB out) replaces the
    // statistics pass over the 77 MB tensor (23 of 24 per ViT pass).  The first block of a call still runs the statistics pass (its input comes from ln_pre, or from an
    // earlier partial pass).  DIST_AMD_ROWSTATS=0: off (measurement knob).
    static const bool rs_env = (dist_knob("DIST_AMD_ROWSTATS", 1) != 0);
    const bool rs = rs_env && h->vit_fold && rowstats_ok(x, rowsS, d, d) && rowstats_ok(x, rowsS, d, 4 * d);
    bool part_of_xin = false;                          // lnpart holds the partials of `xin`
    // dist_config.vit_fp8 (BASELINE config 5): which of the four GEMMs of a block run on e4m3 operands - bit 0 in_proj, 1 out_proj, 2 c_fc,
    // 3 c_proj.  Needs the LayerNorm fold (bf16 engine); a GEMM whose shape the fp8 kernel does not take runs in bf16.
    const int f8 = (h->aq && h->vit_fold) ? c.vit_fp8 : 0;
    // producers write the e4m3 images (bit 16): needs all four GEMMs on e4m3, scales from an earlier pass, shapes the fp8 kernel takes
    const bool img_mode = (f8 & 31) == 31 && h->x8 && fp8_shape_ok(x, rowsS, 3 * d, d) && fp8_shape_ok(x, rowsS, d, d) && fp8_shape_ok(x, rowsS, 4 * d, d) &&
                          fp8_shape_ok(x, rowsS, d, 4 * d);
    if (img_mode && l0 == 0) {
        if (h->f8_passes > 0) RUN(dist_op_fp8_scale_update(h->f8_amax, h->f8_scale, 5 * c.layers, 4.0f, stream));   // last pass's maxima -> this pass's scales
        h->x8_layer = -1;
    }
    const bool fused = img_mode && h->f8_passes > 0;      // the first pass after a pack calibrates: per-token quantisers + dist_op_amax
    for (int i = l0; i < l1; ++i) {
        const VitLayer& v = h->vit[i];
        // the QKV GEMM writes [frame][head][q|k|v][L][64] (DIST_OM_HEADS, leading dimension 64): every (frame, head) operand of
        // the attention kernel is one contiguous block instead of 128-byte pieces at a 3d row stride.
        // LayerNorm fold: ln_1 only computes the row statistics (reads 77 MB, writes 0.4 MB); the GEMM consumes the raw rows with
        // W diag(gamma) and normalises in its epilogue - the normalised tensor is never written or read back.
        int folded = 0;
        if (h->vit_fold) {
            if (part_of_xin) { if (!(h->skip & 128)) RUN(dist_op_ln_stats_from_partials(h->lnpart, d / 64, rowsS, d, 1e-5f, h->lnstats, h->lnstats + rowsS, stream)); }   // (skip 128, timing only: stale statistics)
            else RUN(ln_fwd(x, h->visual, v.ln1, xin, nullptr, rowsS, h->lnstats, h->lnstats + rowsS));
            if (f8 & 1) {                                  // e4m3 image of the raw rows, then the folded GEMM on the block-scaled fp8 MFMA
                Out8 q8o;                                  // image mode: q | k | v leave as e4m3 ONLY (head-major bytes in h->qkv)
                if (fused) { q8o.img = static_cast<unsigned char*>(h->qkv); q8o.scale = h->f8_scale + 5 * i + 4; q8o.amax = h->f8_amax + 5 * i + 4; }
                void* qkv16 = fused ? nullptr : h->qkv;
                if (fused && i > 0 && h->x8_layer == i - 1) {     // the previous block's c_proj left the image
                    folded = gemm_fp8(x, h->x8, h->f8_scale + 5 * (i - 1) + 2, true, v.q_qkv, rowsS, 3 * d, d, qkv16, 64, v.b_qkv, nullptr, nullptr, h->lnstats, v.cs8_qkv, nullptr,
                                      OM(DIST_OM_HEADS, L, h->heads), q8o);
                } else {
                    RUN(dist_op_quant_rows_fp8(xin, DIST_BF16, rowsS, d, d, h->aq, d, h->sa, stream));
                    folded = gemm_fp8(x, h->aq, h->sa, false, v.q_qkv, rowsS, 3 * d, d, qkv16, 64, v.b_qkv, nullptr, nullptr, h->lnstats, v.cs8_qkv, nullptr, OM(DIST_OM_HEADS, L, h->heads), q8o);
                }
                if (img_mode && !fused && folded > 0) RUN(dist_op_amax(h->qkv, DIST_BF16, rowsS * 3 * d, h->f8_amax + 5 * i + 4, stream));
                if (folded < 0) return fail(h, folded, "fp8 QKV GEMM failed");
            }
            if (!folded) folded = gemm_lnfold(x, xin, d, x.pk(v.pk_fold_qkv), rowsS, 3 * d, d, h->qkv, 64, v.b_qkv, h->lnstats, v.cs_qkv, nullptr, OM(DIST_OM_HEADS, L, h->heads));
            if (folded < 0) return fail(h, folded, "folded QKV GEMM failed");
        }
        if (!folded) {
            RUN(ln_fwd(x, h->visual, v.ln1, xin, h->hbuf, rowsS, nullptr, nullptr));
            RUN(gemm(x, h->hbuf, d, x.pk(v.qkv.pk.f), rowsS, 3 * d, d, 1, h->qkv, 64, x.vs(v.qkv.bias), nullptr, nullptr, nullptr, RM(), OM(DIST_OM_HEADS, L, h->heads)));
        }
        if (fused) RUN(dist_op_attention_fp8(h->qkv, h->f8_scale + 5 * i + 4, nullptr, h->aq, h->f8_scale + 5 * i + 3, h->f8_amax + 5 * i + 3, b * h->t, L, h->heads, stream));
        else if (!(h->skip & 32)) RUN(dist_op_attention(h->qkv, h->att, b * h->t, L, h->heads, DIST_QKV_HEADS, c.dtype, stream));
        int done8 = 0;
        if (f8 & 2) {
            if (fused) {                                   // the attention kernel left the e4m3 image in h->aq
                Out8 o8;
                o8.img = h->xa8; o8.scale = h->f8_scale + 5 * i; o8.amax = h->f8_amax + 5 * i;
                done8 = gemm_fp8(x, h->aq, h->f8_scale + 5 * i + 3, true, v.q_out, rowsS, d, d, h->xa, d, x.vs(v.out.bias), xin, nullptr, nullptr, nullptr, rs ? h->lnpart : nullptr, OM(), o8);
            } else {
                RUN(dist_op_quant_rows_fp8(h->att, DIST_BF16, rowsS, d, d, h->aq, d, h->sa, stream));
                done8 = gemm_fp8(x, h->aq, h->sa, false, v.q_out, rowsS, d, d, h->xa, d, x.vs(v.out.bias), xin, nullptr, nullptr, nullptr, rs ? h->lnpart : nullptr);
                if (img_mode && done8 > 0) {
                    RUN(dist_op_amax(h->xa, DIST_BF16, rowsS * d, h->f8_amax + 5 * i, stream));
                    RUN(dist_op_amax(h->att, DIST_BF16, rowsS * d, h->f8_amax + 5 * i + 3, stream));
                }
            }
            if (done8 < 0) return fail(h, done8, "fp8 out-projection GEMM failed");
        }
        if (!done8) RUN(gemm(x, h->att, d, x.pk(v.out.pk.f), rowsS, d, d, 1, h->xa, d, x.vs(v.out.bias), xin, nullptr, nullptr, RM(), OM(), 0, nullptr, rs ? h->lnpart : nullptr));
        folded = 0;
        if (h->vit_fold) {
            if (rs) { if (!(h->skip & 128)) RUN(dist_op_ln_stats_from_partials(h->lnpart, d / 64, rowsS, d, 1e-5f, h->lnstats, h->lnstats + rowsS, stream)); }
            else RUN(ln_fwd(x, h->visual, v.ln2, h->xa, nullptr, rowsS, h->lnstats, h->lnstats + rowsS));
            if (f8 & 4) {
                if (fused) {                               // input: the image out_proj left; output: the QuickGELU'd hidden tensor as e4m3 ONLY (h->aq)
                    Out8 o8;
                    o8.img = h->aq; o8.scale = h->f8_scale + 5 * i + 1; o8.amax = h->f8_amax + 5 * i + 1; o8.act = true;
                    folded = gemm_fp8(x, h->xa8, h->f8_scale + 5 * i, true, v.q_fc, rowsS, 4 * d, d, nullptr, 4 * d, v.b_fc, nullptr, nullptr, h->lnstats, v.cs8_fc, nullptr, OM(), o8);
                } else {
                    RUN(dist_op_quant_rows_fp8(h->xa, DIST_BF16, rowsS, d, d, h->aq, d, h->sa, stream));
                    folded = gemm_fp8(x, h->aq, h->sa, false, v.q_fc, rowsS, 4 * d, d, nullptr, 4 * d, v.b_fc, nullptr, h->mlp, h->lnstats, v.cs8_fc, nullptr);
                    if (img_mode && folded > 0) RUN(dist_op_amax(h->mlp, DIST_BF16, rowsS * 4 * d, h->f8_amax + 5 * i + 1, stream));
                }
                if (folded < 0) return fail(h, folded, "fp8 MLP GEMM failed");
            }
            if (!folded && !(h->skip & 64)) folded = gemm_lnfold(x, h->xa, d, x.pk(v.pk_fold_fc), rowsS, 4 * d, d, nullptr, 4 * d, v.b_fc, h->lnstats, v.cs_fc, h->mlp);
            if (h->skip & 64) folded = 1;
            if (folded < 0) return fail(h, folded, "folded MLP GEMM failed");
        }
        if (!folded) {
            RUN(ln_fwd(x, h->visual, v.ln2, h->xa, h->hbuf, rowsS, nullptr, nullptr));
            RUN(gemm(x, h->hbuf, d, x.pk(v.fc.pk.f), rowsS, 4 * d, d, 1, nullptr, 4 * d, x.vs(v.fc.bias), nullptr, nullptr, h->mlp));
        }
        done8 = 0;
        if (f8 & 8) {
            if (fused) {
                Out8 o8;
                o8.img = h->x8; o8.scale = h->f8_scale + 5 * i + 2; o8.amax = h->f8_amax + 5 * i + 2;
                done8 = gemm_fp8(x, h->aq, h->f8_scale + 5 * i + 1, true, v.q_proj, rowsS, d, 4 * d, S.feat[i], d, x.vs(v.proj.bias), h->xa, nullptr, nullptr, nullptr,
                                 rs ? h->lnpart : nullptr, OM(), o8);
                if (done8 > 0) h->x8_layer = i;
            } else {
                RUN(dist_op_quant_rows_fp8(h->mlp, DIST_BF16, rowsS, 4 * d, 4 * d, h->aq, 4 * d, h->sa, stream));
                done8 = gemm_fp8(x, h->aq, h->sa, false, v.q_proj, rowsS, d, 4 * d, S.feat[i], d, x.vs(v.proj.bias), h->xa, nullptr, nullptr, nullptr, rs ? h->lnpart : nullptr);
                if (img_mode && done8 > 0) RUN(dist_op_amax(S.feat[i], DIST_BF16, rowsS * d, h->f8_amax + 5 * i + 2, stream));
            }
            if (done8 < 0) return fail(h, done8, "fp8 MLP projection GEMM failed");
        }
        if (!done8) RUN(gemm(x, h->mlp, 4 * d, x.pk(v.proj.pk.f), rowsS, d, 4 * d, 1, S.feat[i], d, x.vs(v.proj.bias), h->xa, nullptr, nullptr, RM(), OM(), 0, nullptr, rs ? h->lnpart : nullptr));
        part_of_xin = rs;
        HIP_CHECK_RET(hipEventRecord(S.ev_feat[i], x.s));               // mid_feat[i] complete: the branch may consume it
        S.valid[i] = 1;
        if (h->dummy & 1) for (int r = 0; r < h->dummy_reps; ++r) RUN(ln_fwd(x, h->visual, v.ln1, S.feat[i], nullptr, rowsS, h->lnstats2, h->lnstats2 + rowsS));
        xin = S.feat[i];
    }
    S.next_layer = l1;
    S.pending_b = b;
    if (l1 == c.layers && img_mode) ++h->f8_passes;
    if (l1 == c.layers) {
        HIP_CHECK_RET(hipEventRecord(h->ev_vit_done, x.s));
        mark(h, DIST_MARK_VIT_END, x.s);
        h->vit_ran = true;
        S.b = b;
    }
    return DIST_OK;
}

static int vit_args_ok(dist_handle* h, const float* video, int b, const char* who) {
    if (!h || !video) return DIST_ERR_ARG;
    if (!h->ws) return fail(h, DIST_ERR_UNBOUND, "%s before dist_bind", who);
    if (b <= 0 || b > h->cfg.batch) return fail(h, DIST_ERR_ARG, "batch %d outside (0, %d]", b, h->cfg.batch);
    return DIST_OK;
}

extern "C" int dist_vit_forward(dist_handle* h, const float* video, int b, void* stream) {
    RUN(vit_args_ok(h, video, b, "dist_vit_forward"));
    h->slot[h->cur].prefetched = false;
    RUN(vit_forward_slot(h, video, b, h->cur, stream, false, nullptr, 0, h->cfg.layers));
    h->fwd_b = b;
    h->branch_b = 0;
    return DIST_OK;
}

// The caller's own frozen-ViT features instead of a dist_vit_forward pass (reference DiSTNetwork.forward reads input['mid_feat']['img'][layer_id]
// and input['images'], dist.py:222-247): copied into the current feature slot, converted to the engine's storage type and token-major rows.
extern "C" int dist_features_import(dist_handle* h, const void* const* mid_feat, int src_dtype, const float* video, int b, void* stream) {
    RUN(vit_args_ok(h, video, b, "dist_features_import"));
    if (!mid_feat || (src_dtype != DIST_F32 && src_dtype != DIST_BF16)) return fail(h, DIST_ERR_ARG, "dist_features_import: mid_feat / src_dtype");
    const dist_config& c = h->cfg;
    for (int i : h->sel)
        if (!mid_feat[i]) return fail(h, DIST_ERR_ARG, "dist_features_import: mid_feat[%d] is NULL (every selected block is needed; the others may be NULL)", i);
    dist_handle::FeatSlot& S = h->slot[h->cur];
    hipStream_t s = static_cast<hipStream_t>(stream);
    S.prefetched = false;
    if (h->vit_ran) HIP_CHECK_RET(hipStreamWaitEvent(s, h->ev_vit_done, 0));      // a ViT pass still in flight may be writing this slot
    HIP_CHECK_RET(hipEventRecord(S.ev_pre, s));
    RUN(dist_op_patchify(video, S.patches, b, c.frames, c.resolution, c.resolution, c.patch, c.dtype, stream));
    HIP_CHECK_RET(hipEventRecord(S.ev_feat[c.layers], s));
    for (int i = 0; i < c.layers; ++i) {
        if (mid_feat[i]) RUN(dist_k_import_feat(mid_feat[i], src_dtype, S.feat[i], c.dtype, b * h->t, h->L, c.width, s));
        S.valid[i] = mid_feat[i] ? 1 : 0;                               // a block the caller did not supply still holds an OLDER clip: not readable
        HIP_CHECK_RET(hipEventRecord(S.ev_feat[i], s));
    }
    S.next_layer = c.layers; S.pending_b = b; S.b = b;
    h->fwd_b = b;
    h->branch_b = 0;
    return DIST_OK;
}

// Software pipelining over batches: the ViT is frozen, so its forward for batch n+1 does not depend on the optimizer step of
// batch n.  dist_vit_prefetch runs it into the spare feature slot on its own (low-priority) stream while the branch
// forward / backward / AdamW of batch n run on the caller's stream; dist_vit_adopt makes that slot the current one.
extern "C" int dist_vit_prefetch_layers(dist_handle* h, const float* video, int b, int layer_end, void* stream, void* after) {
    if (!h) return DIST_ERR_ARG;
    const int k = h->cur ^ 1;
    dist_handle::FeatSlot& S = h->slot[k];
    int l0 = 0;
    if (video) {                                       // a new pass into the spare slot
        RUN(vit_args_ok(h, video, b, "dist_vit_prefetch"));
        S.prefetched = true;
        S.b = 0;
    } else {                                           // continue the pass in flight
        if (!h->ws) return fail(h, DIST_ERR_UNBOUND, "dist_vit_prefetch_layers before dist_bind");
        if (!S.prefetched || S.b != 0 || S.pending_b <= 0) return fail(h, DIST_ERR_STATE, "dist_vit_prefetch_layers(video = NULL): no prefetch pass in flight");
        l0 = S.next_layer;
        b = S.pending_b;
    }
    if (layer_end < l0 || layer_end > h->cfg.layers) return fail(h, DIST_ERR_ARG, "layer_end %d outside [%d, %d]", layer_end, l0, h->cfg.layers);
    if (!video && layer_end == l0) return DIST_OK;
    return vit_forward_slot(h, video, b, k, stream ? stream : h->pf, true, after, l0, layer_end);
}
extern "C" int dist_vit_prefetch(dist_handle* h, const float* video, int b, void* stream, void* after) {
    if (!h || !video) return DIST_ERR_ARG;
    return dist_vit_prefetch_layers(h, video, b, h->cfg.layers, stream, after);
}
extern "C" int dist_vit_adopt(dist_handle* h) {
    if (!h) return DIST_ERR_ARG;
    if (!h->ws) return fail(h, DIST_ERR_UNBOUND, "dist_vit_adopt before dist_bind");
    const int k = h->cur ^ 1;
    if (!h->slot[k].prefetched || h->slot[k].b <= 0) return fail(h, DIST_ERR_STATE, "dist_vit_adopt needs dist_vit_prefetch first");
    h->use_slot(k);
    h->fwd_b = h->slot[k].b;
    h->branch_b = 0;
    return DIST_OK;
}

// one-query cross attention block: s_out = s_in + out_proj(attn(q = W_q LN(s_in), kv = W_kv LN(keys))).
// The key side (LayerNorm + K/V projection of all keys: the only large kernels of the ada-pooling tail) does not depend
// on the query, so it is a separate call: for the spatial blocks it runs on the second side stream for all ada layers at
// once, beside the chain of small query-side kernels.
static int xattn_keys(dist_handle* h, const Ctx& x, const XAttn& A, const void* keys, long nkeys_total, void* kn, float* kn_mean, float* kn_rstd, void* kv) {
    const int Ci = h->cfg.integration_dim;
    RUN(ln_fwd(x, h->theta, A.ln1, keys, kn, nkeys_total, kn_mean, kn_rstd));
    RUN(gemm(x, kn, Ci, x.pk(A.kv.pk.f), nkeys_total, 2 * Ci, Ci, 1, kv, 2 * Ci, x.th(A.kv.bias), nullptr, nullptr, nullptr));
    return DIST_OK;
}
static int xattn_query(dist_handle* h, const Ctx& x, const XAttn& A, const void* s_in, long nq, int S, const void* kv, hipEvent_t kv_ready,
                       void* qn, float* qn_mean, float* qn_rstd, void* q, void* o, float* probs, void* s_out) {
    const int Ci = h->cfg.integration_dim;
    RUN(ln_fwd(x, h->theta, A.ln1, s_in, qn, nq, qn_mean, qn_rstd));
    RUN(gemm(x, qn, Ci, x.pk(A.q.pk.f), nq, Ci, Ci, 1, q, Ci, x.th(A.q.bias), nullptr, nullptr, nullptr));
    if (kv_ready) HIP_CHECK_RET(hipStreamWaitEvent(x.s, kv_ready, 0));
    RUN(dist_op_xattn1q(q, kv, o, probs, (int)nq, S, Ci, x.dtype, x.s));
    RUN(gemm(x, o, Ci, x.pk(A.out.pk.f), nq, Ci, Ci, 1, s_out, Ci, x.th(A.out.bias), s_in, nullptr, nullptr));
    return DIST_OK;
}

extern "C" int dist_branch_forward(dist_handle* h, const float* text_features, int b, float* logits, float* vid_logits, void* stream) {
    if (!h || !text_features) return DIST_ERR_ARG;
    if (!h->ws) return fail(h, DIST_ERR_UNBOUND, "dist_branch_forward before dist_bind");
    if (h->fwd_b != b) return fail(h, DIST_ERR_STATE, "dist_branch_forward(b=%d) needs dist_vit_forward with the same batch first (have %d)", b, h->fwd_b);
    const dist_config& c = h->cfg;
    // The branch runs on the handle's side stream: layer i only needs mid_feat[i], so it executes underneath the
    // remaining frozen-ViT layers still queued on the caller's stream (their LayerNorm / attention phases and the
    // tails of the GEMM rounds leave CUs idle); the caller's stream joins at the end.
    hipStream_t A = static_cast<hipStream_t>(stream);
    hipStream_t S1 = (h->serial & 1) ? A : h->side, S2 = (h->serial & 1) ? A : h->side2;
    Ctx x{h, S1, c.dtype};
    const int d = c.width, Ci = c.integration_dim, Ct = c.temporal_dim, C4 = h->C4, N = h->N, L = h->L, T = c.frames, t = h->t, al = c.alpha;
    const long rowsX = (long)b * T * N, rowsS = (long)b * t * L, rowsQ = (long)b * t * N, bt = (long)b * t;
    const int Ch = h->Ch, Cf = h->Cf;             // hidden widths (== Ct / Ci for MLP ratio 1)
    const int nl = h->nsel, nv = c.layers;       // DiST layers (one per SELECTED ViT block, dist.py:226) / ViT blocks
    stream = S1;
    HIP_CHECK_RET(hipStreamWaitEvent(x.s, h->ev_pre, 0));               // what preceded dist_vit_forward on the caller's stream
    HIP_CHECK_RET(hipStreamWaitEvent(x.s, h->ev_feat[nv], 0));          // patch rows
    const bool pref = h->slot[h->cur].prefetched && !(h->serial & 1);
    if (pref) {                                                         // the features come from another stream: the side streams
        HIP_CHECK_RET(hipEventRecord(h->ev_bpre, A));                   // must also follow the caller's stream (re-pack of the last step)
        HIP_CHECK_RET(hipStreamWaitEvent(S1, h->ev_bpre, 0));
        HIP_CHECK_RET(hipStreamWaitEvent(S2, h->ev_bpre, 0));
    }

    // Two streams inside the branch: the temporal chain (TemporalNet_i, I2T_i) on `xt`, the integration chain
    // (input_linear_i, T2I_i, IntegrationNetwork_i) on `x`.  TemporalNet_{i+1} only needs X_{i+1} = X'_i + I2T(M_i),
    // not R_i, so it runs underneath IntegrationNetwork_i; the two chains meet at M_i (-> I2T) and X'_i (-> T2I).
    Ctx xt{h, S2, c.dtype};
    HIP_CHECK_RET(hipStreamWaitEvent(xt.s, h->ev_pre, 0));
    HIP_CHECK_RET(hipStreamWaitEvent(xt.s, h->ev_feat[nv], 0));
    auto ev_xp = [&](int i) { return h->ev_a[i]; };
    auto ev_m = [&](int i) { return h->ev_a[nl + i]; };

    // temporal stem: Conv3d k=(tp,P,P) as a 5-tap row-shifted GEMM over the shared patch rows (dist.py:178-181,225)
    mark(h, DIST_MARK_FWD_BEGIN, xt.s);
    RUN(gemm(xt, h->patches, h->Kp, x.pk(h->stem.pk.f), rowsX, Ct, h->Kp, h->stem.taps, h->lw[0].X, Ct, x.th(h->stem.bias), nullptr, nullptr, nullptr,
             RM(DIST_RM_SHIFT, T * N, N, 1)));
    for (int i = 0; i < nl; ++i) {
        const DistLayer& l = h->dl[i];
        DistLayerWs& w = h->lw[i];
        void* Xnext = (i + 1 < nl) ? h->lw[i + 1].X : h->Xlast;
        // ---- temporal chain: TemporalNet (dist.py:48-65): one fused launch (tnet.hip) where the geometry allows, else LayerNorm + two GEMMs
        if (h->skip & 4) {
        } else if (Ch == Ct && dist_k_tnet_fwd_eligible(c.dtype, Ct, h->G, l.tn_fc1.taps)) {
            dist_tnet_args ta;
            memset(&ta, 0, sizeof(ta));
            ta.X = w.X; ta.W1 = x.pk(l.tn_fc1.pk.f); ta.W2 = x.pk(l.tn_fc2.pk.f);
            ta.b1 = x.th(l.tn_fc1.bias); ta.b2 = x.th(l.tn_fc2.bias); ta.ln_w = x.th(l.tn_ln.w); ta.ln_b = x.th(l.tn_ln.b);
            ta.z = w.z; ta.p = w.p; ta.Xp = w.Xp;
            if (!h->inference) { ta.U = w.U; ta.V = w.V; }                      // (U, V: the weight-gradient GEMMs of backward read them)
            ta.mean = w.tn_mean; ta.rstd = w.tn_rstd;
            ta.clips = b; ta.T = T; ta.G = h->G; ta.Ct = Ct; ta.tk = l.tn_fc1.taps; ta.dtype = c.dtype; ta.eps = 1e-5f;
            RUN(dist_op_temporal_net_fwd(&ta, xt.s));
        } else {
            RUN(ln_fwd(xt, h->theta, l.tn_ln, w.X, w.U, rowsX, w.tn_mean, w.tn_rstd));
            RUN(gemm(xt, w.U, Ct, x.pk(l.tn_fc1.pk.f), rowsX, Ch, Ct, l.tn_fc1.taps, w.z, Ch, x.th(l.tn_fc1.bias), nullptr, nullptr, w.V,
                     RM(DIST_RM_SHIFT, T * N, N, 1)));
            RUN(gemm(xt, w.V, Ch, x.pk(l.tn_fc2.pk.f), rowsX, Ct, Ch, 9, w.p, Ct, x.th(l.tn_fc2.bias), w.X, nullptr, w.Xp, RM(DIST_RM_SPATIAL, h->G, 0, 1)));
        }
        HIP_CHECK_RET(hipEventRecord(ev_xp(i), xt.s));
        // ---- integration chain: mid_feat = input_linear(F_i) + res_feat (dist.py:229)
        HIP_CHECK_RET(hipStreamWaitEvent(x.s, h->ev_feat[h->sel[i]], 0));
        RUN(gemm(x, h->feat[h->sel[i]], d, x.pk(l.in_lin.pk.f), rowsS, Ci, d, 1, w.M, Ci, x.th(l.in_lin.bias), i ? h->lw[i - 1].R : nullptr, nullptr, nullptr));
        // I2T (dist.py:90-105,231): Linear on the non-cls rows, nearest-upsampled x alpha in T, + x_temporal - inside the fused IntegrationNetwork launch below
        // (behind its T2I stage), or as a GEMM on the temporal chain.  (the last layer's result is discarded by the reference, dist.py:235 -> skipped)
        const bool i2t_fused = h->ig_i2t && i + 1 < nl && !(h->skip & 8);
        if (!i2t_fused) HIP_CHECK_RET(hipEventRecord(ev_m(i), x.s));
        if (i + 1 < nl && !i2t_fused) {
            HIP_CHECK_RET(hipStreamWaitEvent(xt.s, ev_m(i), 0));
            RUN(gemm(xt, w.M, Ci, x.pk(l.i2t.pk.f), rowsQ, Ct, Ci, 1, Xnext, Ct, x.th(l.i2t.bias), w.Xp, nullptr, nullptr,
                     RM(DIST_RM_SKIPCLS, N), OM(DIST_OM_DUP, al, N)));
        }
        // T2I (dist.py:68-86,232): strided temporal conv into the patch rows of M', learnable cls row
        HIP_CHECK_RET(hipStreamWaitEvent(x.s, ev_xp(i), 0));
        const bool t2i_in_front = h->ig_t2i;                // (with DIST_AMD_SKIP & 8 nothing forms M': the knob's results are wrong by design)
        if (!t2i_in_front) {
        RUN(gemm(x, w.Xp, Ct, x.pk(l.t2i.pk.f), rowsQ, Ci, Ct, al, w.Mp, Ci, x.th(l.t2i.bias), w.M, nullptr, nullptr,
                 RM(DIST_RM_STRIDED, al, N), OM(DIST_OM_INSERTCLS, N)));
        RUN(dist_k_cls_rows(w.Mp, w.M, x.th(l.cls_token), (int)bt, L, Ci, t, c.dtype, x.s));
        }
        // IntegrationNetwork (dist.py:16-45): one fused launch (integ.hip) where the geometry allows, else LayerNorm + four GEMMs
        if (h->ig_on) {
            if (!(h->skip & 8)) {
                dist_integ_args ia;
                memset(&ia, 0, sizeof(ia));
                if (t2i_in_front) {       // M' is formed in the kernel; it is written out only where something else reads it (the last layer's residual, the unfused backward)
                    ia.t2i_M = w.M; ia.t2i_Xp = w.Xp; ia.t2i_W = l.ig_Wt; ia.t2i_bias = x.th(l.t2i.bias); ia.t2i_cls = x.th(l.cls_token);
                    if (i == nl - 1 || !h->ig_bwd || h->keep_mid) ia.Mp_out = w.Mp;
                    if (i2t_fused) { ia.i2t_W = l.ig_Wi; ia.i2t_bias = x.th(l.i2t.bias); ia.i2t_Xnext = Xnext; }
                } else ia.Mp = w.Mp;
                ia.W1 = l.ig_W1; ia.W2 = l.ig_W2; ia.W3 = l.ig_W3; ia.b1 = l.ig_b1; ia.b2 = l.ig_b2; ia.b3 = l.ig_b3;
                ia.R = w.R;
                if (!h->inference) {                                            // (what backward reads)
                    if (h->ig_xhat) ia.Xhat = w.Na;
                    else {
                        ia.ln_w = x.th(l.in_ln.w); ia.ln_b = x.th(l.in_ln.b); ia.ln_t_w = x.th(l.in_ln_t.w); ia.ln_t_b = x.th(l.in_ln_t.b);
                        ia.Na = w.Na; ia.Nb = w.Nb;
                    }
                    ia.mean = w.in_mean; ia.rstd = w.in_rstd; ia.zf_h2 = w.zf; ia.hf_g2 = w.hf; ia.h1 = w.h1;
                }
                ia.clips = (int)b; ia.t = t; ia.L = L; ia.Ci = Ci; ia.C4 = C4; ia.tk = l.tf_fc2.taps; ia.dtype = c.dtype; ia.eps = 1e-5f;
                RUN(dist_op_integration_fwd(&ia, x.s));
                if (i2t_fused) {                                                // X of the next layer is written by this launch: the temporal chain continues behind it
                    HIP_CHECK_RET(hipEventRecord(ev_m(i), x.s));
                    HIP_CHECK_RET(hipStreamWaitEvent(xt.s, ev_m(i), 0));
                }
            }
        } else {
        RUN(ln_fwd(x, h->theta, l.in_ln, w.Mp, w.Na, rowsS, w.in_mean, w.in_rstd, &l.in_ln_t, w.Nb));
        if (!(h->skip & 8)) {
        // (inference: the pre-activations zf / h2 are what backward needs - only the activated tensors are written)
        RUN(gemm(x, w.Na, Ci, x.pk(l.ffn_fc.pk.f), rowsS, Cf, Ci, 1, h->inference ? nullptr : w.zf, Cf + C4, x.th(l.ffn_fc.bias), nullptr, nullptr, w.hf));
        RUN(gemm(x, w.Nb, Ci, x.pk(l.tf_fc1.pk.f), rowsS, C4, Ci, 1, w.h1, C4, x.th(l.tf_fc1.bias), nullptr, nullptr, nullptr));
        RUN(gemm(x, w.h1, C4, x.pk(l.tf_fc2.pk.f), rowsS, C4, C4, l.tf_fc2.taps, h->inference ? nullptr : w.h2, Cf + C4, x.th(l.tf_fc2.bias), nullptr, nullptr, w.g2,
                 RM(DIST_RM_SHIFT, t * L, L, 1)));
        // R = ffn.c_proj(hf) + temporal_ffn.c_proj(g2): one GEMM over [hf | g2] (K = Ci + C4) with the two weights side by side
        RUN(gemm(x, w.hf, Cf + C4, x.pk(l.pk_proj_f), rowsS, Ci, Cf + C4, 1, w.R, Ci, x.th(l.ffn_proj.bias), nullptr, nullptr, nullptr,
                 RM(), OM(), 0, x.th(l.tf_proj.bias)));
        }
        }
        if (i == nl / 2 - 1) mark(h, DIST_MARK_FWD_MID, x.s);
    }
    // current_layer_feat = res_feat + updated_mid_feat (dist.py:239)
    RUN(dist_op_add(h->lw[nl - 1].R, h->lw[nl - 1].Mp, h->Fz, rowsS * Ci, c.dtype, stream));
    // key side of every spatial ada block on the (now idle) temporal stream: needs Fz only
    hipEvent_t ev_fz = h->ev_a[2 * nl];
    HIP_CHECK_RET(hipEventRecord(ev_fz, x.s));
    HIP_CHECK_RET(hipStreamWaitEvent(xt.s, ev_fz, 0));
    for (int a = 0; a < c.ada_layers; ++a) {
        AdaWs& w = h->aw[a];
        RUN(xattn_keys(h, xt, h->ada[a].sp, h->Fz, rowsS, w.kn, w.kn_mean, w.kn_rstd, w.kv));
        HIP_CHECK_RET(hipEventRecord(h->ev_a[2 * nl + 1 + a], xt.s));
    }
    RUN(dist_k_bcast_rows(x.th(h->agg_sp_cls), h->sbuf[0], bt, Ci, c.dtype, x.s));
    RUN(dist_k_bcast_rows(x.th(h->agg_cls), h->ubuf[0], b, Ci, c.dtype, x.s));
    for (int a = 0; a < c.ada_layers; ++a) {
        const AdaLayer& A = h->ada[a];
        AdaWs& w = h->aw[a];
        // spatial: per-frame cls query over the L tokens of its frame (dist.py:144-146)
        RUN(xattn_query(h, x, A.sp, h->sbuf[a], bt, L, w.kv, h->ev_a[2 * nl + 1 + a], w.qn, w.qn_mean, w.qn_rstd, w.q, w.o, w.probs, w.s1));
        RUN(ln_fwd(x, h->theta, A.ln_sp, w.s1, w.sn, bt, w.s1_mean, w.s1_rstd));
        RUN(gemm(x, w.sn, Ci, x.pk(A.sp_fc.pk.f), bt, 4 * Ci, Ci, 1, w.zs, 4 * Ci, x.th(A.sp_fc.bias), nullptr, nullptr, w.hs));
        RUN(gemm(x, w.hs, 4 * Ci, x.pk(A.sp_proj.pk.f), bt, Ci, 4 * Ci, 1, h->sbuf[a + 1], Ci, x.th(A.sp_proj.bias), w.s1, nullptr, nullptr));
        // temporal: per-clip cls query over its t frame tokens (+ positional embedding) (dist.py:153-160)
        RUN(dist_k_add_table(h->sbuf[a + 1], x.th(A.pos), w.c, bt, Ci, t, c.dtype, x.s));
        RUN(xattn_keys(h, x, A.tm, w.c, bt, w.kn2, w.kn2_mean, w.kn2_rstd, w.kv2));
        RUN(xattn_query(h, x, A.tm, h->ubuf[a], b, t, w.kv2, nullptr, w.qn2, w.qn2_mean, w.qn2_rstd, w.q2, w.o2, w.probs2, w.u1));
        RUN(ln_fwd(x, h->theta, A.ln_tm, w.u1, w.un, b, w.u1_mean, w.u1_rstd));
        RUN(gemm(x, w.un, Ci, x.pk(A.tm_fc.pk.f), b, 4 * Ci, Ci, 1, w.zu, 4 * Ci, x.th(A.tm_fc.bias), nullptr, nullptr, w.hu));
        RUN(gemm(x, w.hu, 4 * Ci, x.pk(A.tm_proj.pk.f), b, Ci, 4 * Ci, 1, h->ubuf[a + 1], Ci, x.th(A.tm_proj.bias), w.u1, nullptr, nullptr));
    }
    // x_logits = ln_post(top_cls + proj_spatial_cls_token(mean_t vit_cls)); cls_x = x_logits @ proj (dist.py:242-246)
    RUN(dist_k_mean_cls(h->feat[h->sel[nl - 1]], h->mean_cls, b, t, L, d, c.dtype, x.s));
    RUN(gemm(x, h->mean_cls, d, x.pk(h->cls_proj.pk.f), b, Ci, d, 1, h->ysum, Ci, x.th(h->cls_proj.bias), h->ubuf[c.ada_layers], nullptr, nullptr));
    RUN(ln_fwd(x, h->theta, h->ln_post, h->ysum, h->zpost, b, h->y_mean, h->y_rstd));
    RUN(gemm(x, h->zpost, Ci, x.pk(h->proj.pk.f), b, c.embed_dim, Ci, 1, h->v, c.embed_dim, nullptr, nullptr, nullptr, nullptr));
    // join: the caller's stream continues after the branch; the logits kernel runs there (it reads caller-produced text features)
    HIP_CHECK_RET(hipEventRecord(h->ev_join, x.s));
    mark(h, DIST_MARK_FWD_END, x.s);
    HIP_CHECK_RET(hipStreamWaitEvent(A, h->ev_join, 0));
    // cosine logits (clip.py:509-518)
    RUN(dist_k_logits_loss(h->v, text_features, h->logit_scale, nullptr, h->logits, vid_logits, nullptr, nullptr, nullptr, nullptr, nullptr,
                           b, c.embed_dim, c.num_classes, c.dtype, A));
    if (logits) HIP_CHECK_RET(hipMemcpyAsync(logits, h->logits, (size_t)b * c.num_classes * sizeof(float), hipMemcpyDeviceToDevice, A));
    h->branch_b = b;
    h->branch_infer = h->inference;
    h->text = text_features;
    return DIST_OK;
}

extern "C" int dist_loss(dist_handle* h, const float* soft_target, int b, float* loss, float* dlogits, void* stream) {
    if (!h || !soft_target) return DIST_ERR_ARG;
    if (h->branch_b != b || !h->text) return fail(h, DIST_ERR_STATE, "dist_loss(b=%d) needs dist_branch_forward with the same batch first", b);
    const dist_config& c = h->cfg;
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_CHECK_RET(hipMemsetAsync(h->loss, 0, sizeof(float), s));
    RUN(dist_k_logits_loss(h->v, h->text, h->logit_scale, soft_target, nullptr, nullptr, h->loss, nullptr, nullptr, nullptr, h->dlogits,
                           b, c.embed_dim, c.num_classes, c.dtype, stream));
    if (loss) HIP_CHECK_RET(hipMemcpyAsync(loss, h->loss, sizeof(float), hipMemcpyDeviceToDevice, s));
    if (dlogits) HIP_CHECK_RET(hipMemcpyAsync(dlogits, h->dlogits, (size_t)b * c.num_classes * sizeof(float), hipMemcpyDeviceToDevice, s));
    return DIST_OK;
}

// -------------------------------------------------------------------------------------------------------------
// backward helpers: y = act?(x W^T + b) given dY (already multiplied by act' where needed)
namespace {

// bias + weight gradients of a plain Linear: db += colsum(dY), dW += dY^T X
int lin_wb(dist_handle* h, const Ctx& x, const Lin& l, const void* dY, const void* X, long rows) {
    RUN(wgrad(x, l, dY, l.N, X, l.K, rows, RM(), RM(), 0, true));
    return DIST_OK;
}
// dX = dY W (optionally * gelu'(aux), optionally accumulated through `res`)
int lin_dx(dist_handle* h, const Ctx& x, const Lin& l, const void* dY, long rows, void* dX, const void* aux = nullptr, const void* res = nullptr, int ld_dy = 0) {
    RUN(gemm(x, dY, ld_dy ? ld_dy : l.N, x.pk(l.pk.b), rows, l.K, l.N, 1, dX, l.K, nullptr, res, aux, nullptr));
    return DIST_OK;
}

// backward of s_out = s_in + MLP(LN(s_in)) with MLP = c_proj(gelu(c_fc(.))); d (rows x Ci) holds dL/ds_out on entry
// and dL/ds_in on exit
int mlp_block_bwd(dist_handle* h, const Ctx& x, const Lin& fc, const Lin& proj, const LNp& ln, const void* s_in, const float* mean, const float* rstd,
                  const void* sn, const void* zs, const void* hs, void* d, long rows, void* dzs, void* dsn) {
    RUN(lin_wb(h, x, proj, d, hs, rows));
    RUN(lin_dx(h, x, proj, d, rows, dzs, zs));
    RUN(lin_wb(h, x, fc, dzs, sn, rows));
    RUN(lin_dx(h, x, fc, dzs, rows, dsn));
    RUN(ln_bwd(x, ln, s_in, mean, rstd, dsn, d, true, rows));
    return DIST_OK;
}

// backward of s_out = s_in + out_proj(attn1q(W_q LN(s_in), W_kv LN(keys))); d holds dL/ds_out -> dL/ds_in;
// dkeys receives (or accumulates) the gradient w.r.t. the key/value source rows.
int xattn_bwd(dist_handle* h, const Ctx& x, const XAttn& A, const void* s_in, long nq, const void* keys, long nkeys_total, int S,
              const void* kn, const float* kn_mean, const float* kn_rstd, const void* kv, const void* qn, const float* qn_mean, const float* qn_rstd,
              const void* q, const void* o, const float* probs, void* d, void* dkeys, bool acc_keys,
              void* d_o, void* dq, void* dkv, void* dqn, void* dkn) {
    const int Ci = h->cfg.integration_dim;
    RUN(lin_wb(h, x, A.out, d, o, nq));
    RUN(lin_dx(h, x, A.out, d, nq, d_o));
    RUN(dist_op_xattn1q_bwd(q, kv, probs, d_o, dq, dkv, (int)nq, S, Ci, x.dtype, x.s));
    RUN(lin_wb(h, x, A.q, dq, qn, nq));
    RUN(lin_wb(h, x, A.kv, dkv, kn, nkeys_total));
    RUN(lin_dx(h, x, A.q, dq, nq, dqn));
    RUN(lin_dx(h, x, A.kv, dkv, nkeys_total, dkn));
    RUN(ln_bwd(x, A.ln1, s_in, qn_mean, qn_rstd, dqn, d, true, nq));
    RUN(ln_bwd(x, A.ln1, keys, kn_mean, kn_rstd, dkn, dkeys, acc_keys, nkeys_total));
    return DIST_OK;
}

}  // namespace

extern "C" int dist_branch_backward(dist_handle* h, const float* dlogits, int b, int zero_grads, void* stream) {
    if (!h || !dlogits) return DIST_ERR_ARG;
    if (!h->grads || !h->dlogit_scale) return fail(h, DIST_ERR_UNBOUND, "dist_branch_backward needs grads and dlogit_scale bound");
    if (h->branch_b != b || !h->text) return fail(h, DIST_ERR_STATE, "dist_branch_backward(b=%d) needs dist_branch_forward with the same batch first", b);
    if (h->branch_infer) return fail(h, DIST_ERR_STATE, "dist_branch_backward after an inference-mode forward (dist_set_inference): nothing was kept for it");
    const dist_config& c = h->cfg;
    Ctx x{h, static_cast<hipStream_t>(stream), c.dtype};
    const int d = c.width, Ci = c.integration_dim, Ct = c.temporal_dim, C4 = h->C4, N = h->N, L = h->L, T = c.frames, t = h->t, al = c.alpha;
    const long rowsX = (long)b * T * N, rowsS = (long)b * t * L, rowsQ = (long)b * t * N, bt = (long)b * t;
    const int nl = h->nsel, na = c.ada_layers, Ch = h->Ch, Cf = h->Cf;
    const size_t es = h->es;

    mark(h, DIST_MARK_BWD_BEGIN, x.s);
    if (zero_grads) {
        HIP_CHECK_RET(hipMemsetAsync(h->grads, 0, (size_t)h->total[0] * sizeof(float), x.s));
        HIP_CHECK_RET(hipMemsetAsync(h->dlogit_scale, 0, sizeof(float), x.s));
    }
    // Accumulating backward with the LayerNorm fold: the weight-gradient GEMMs of ffn.c_fc / temporal_ffn.c_fc1 leave G' = dz^T xhat, which the
    // unfold turns into dW - in place only when the slots were zero.  With earlier gradients in them G' goes to per-layer scratch and the unfold adds.
    h->bwd_accumulate = !zero_grads && h->ig_xhat;
    if (h->bwd_accumulate) HIP_CHECK_RET(hipMemsetAsync(h->ig_gscratch, 0, (size_t)h->ig_gscratch_elems * nl * sizeof(float), x.s));
    // logits -> v (cosine normalisation backward, clip.py:511-517); logit_scale gets its (never applied) gradient
    RUN(dist_k_logits_loss(h->v, h->text, h->logit_scale, nullptr, nullptr, nullptr, nullptr, h->dv, h->dlogit_scale, dlogits, nullptr,
                           b, c.embed_dim, c.num_classes, c.dtype, stream));
    // cls_x = z @ proj ; z = ln_post(u + cls_proj(mean_cls))
    RUN(wgrad(x, h->proj, h->zpost, Ci, h->dv, c.embed_dim, b, RM(), RM(), 4));          // dProj[ci][e] = sum_b z[b][ci] dv[b][e]
    RUN(gemm(x, h->dv, c.embed_dim, x.pk(h->proj.pk.b), b, Ci, c.embed_dim, 1, h->dzp, Ci, nullptr, nullptr, nullptr, nullptr));
    RUN(ln_bwd(x, h->ln_post, h->ysum, h->y_mean, h->y_rstd, h->dzp, h->du, false, b));
    RUN(lin_wb(h, x, h->cls_proj, h->du, h->mean_cls, b));
    // ada-pooling layers in reverse (dist.py:139-162)
    for (int a = na - 1; a >= 0; --a) {
        const AdaLayer& A = h->ada[a];
        AdaWs& w = h->aw[a];
        const bool first = (a == na - 1);
        // temporal MLP + temporal cross attention (du: dL/du_{a+1} -> dL/du_a)
        RUN(mlp_block_bwd(h, x, A.tm_fc, A.tm_proj, A.ln_tm, w.u1, w.u1_mean, w.u1_rstd, w.un, w.zu, w.hu, h->du, b, h->dzu, h->dun));
        RUN(xattn_bwd(h, x, A.tm, h->ubuf[a], b, w.c, bt, t, w.kn2, w.kn2_mean, w.kn2_rstd, w.kv2, w.qn2, w.qn2_mean, w.qn2_rstd, w.q2, w.o2, w.probs2,
                      h->du, h->dc, false, h->do2, h->dq2, h->dkv2, h->dqn2, h->dkn2));
        // c = s_{a+1} + pos: dpos[j] = sum_b dc[b,j]; ds_{a+1} (+)= dc
        RUN(dist_k_cls_rows_bwd(h->dc, x.gr(A.pos), (int)bt, 1, Ci, t, c.dtype, x.s));
        if (first) HIP_CHECK_RET(hipMemcpyAsync(h->ds, h->dc, (size_t)bt * Ci * es, hipMemcpyDeviceToDevice, x.s));
        else RUN(dist_op_add(h->ds, h->dc, h->ds, bt * Ci, c.dtype, stream));
        // spatial MLP + spatial cross attention (ds: dL/ds_{a+1} -> dL/ds_a; dFz accumulates)
        RUN(mlp_block_bwd(h, x, A.sp_fc, A.sp_proj, A.ln_sp, w.s1, w.s1_mean, w.s1_rstd, w.sn, w.zs, w.hs, h->ds, bt, h->dzs, h->dsn));
        RUN(xattn_bwd(h, x, A.sp, h->sbuf[a], bt, h->Fz, rowsS, L, w.kn, w.kn_mean, w.kn_rstd, w.kv, w.qn, w.qn_mean, w.qn_rstd, w.q, w.o, w.probs,
                      h->ds, h->dR, !first, h->do_, h->dq, h->dkv, h->dqn, h->dkn));
    }
    if (na > 0) {
        RUN(bgrad(x, h->agg_cls, h->du, b, Ci));
        RUN(bgrad(x, h->agg_sp_cls, h->ds, bt, Ci));
    } else {
        HIP_CHECK_RET(hipMemsetAsync(h->dR, 0, (size_t)rowsS * Ci * es, x.s));
        RUN(bgrad(x, h->agg_cls, h->du, b, Ci));
    }
    // ---- layer loop on two streams -------------------------------------------------------------------------
    // The data-gradient chain (dX GEMMs, LayerNorm / activation backward) is the critical path and stays on the
    // caller's stream `A`.  Every weight / bias gradient GEMM only CONSUMES chain buffers, so it runs on the handle's
    // side stream `B` behind an event of its producer, up to ~2 layers behind the chain: two latency-bound kernel
    // sequences share the CUs instead of one.  Hazards: (1) chain scratch is double-buffered by layer parity and A
    // waits for B's "layer i+2 done" event before reusing a set; (2) the two in-place updates of the single-stream
    // version are out-of-place here (dM = copy of dM' from the LayerNorm backward, dX_out = dp + LN'(dU)).
    hipStream_t A = x.s, B = (h->serial & 2) ? A : h->side, B2 = (h->serial & 2) ? A : h->side2;
    Ctx xb{h, B, c.dtype}, xb2{h, B2, c.dtype};          // two weight-gradient streams: independent dW GEMMs also overlap each other
    // Two data-gradient chains (round 3): the integration chain (fused IntegrationNetwork backward, I2T data gradient) stays on A, the temporal
    // chain (T2I data gradient + TemporalNet backward of the SAME layer) runs on `Tc` one layer behind it: layer i's temporal chain needs dM'_i
    // and dX_{i+1}, the integration chain of layer i-1 needs dM_i = dM'_i + I2T^T(dX_{i+1}) - not dX_i.  DIST_AMD_BWD_TCHAIN=0: one chain.
    // Measured: 18.78 -> 20.23 ms with the fifth stream (any fifth ACTIVE stream costs that much on this system, profiles/r01_streams_and_queues.md),
    // so the default is one chain.  DIST_AMD_BWD_TCHAIN=1: own stream; 2: the second weight-gradient stream carries the temporal chain instead.
    static const int tchain_env = DIST_AB_KNOB("DIST_AMD_BWD_TCHAIN", 0);
    const bool tchain = tchain_env > 0 && !(h->serial & 2) && (h->chain2 || tchain_env == 2);
    hipStream_t Tc = tchain ? (tchain_env == 2 ? h->side2 : h->chain2) : A;
    if (tchain && tchain_env == 2) { B2 = B; xb2.s = B; }
    Ctx xt{h, Tc, c.dtype};
    int evn = 0;
    auto fork_t = [&]() -> int {         // B and B2 wait for everything enqueued on the temporal chain so far
        if (!tchain) return DIST_OK;
        if (hipEventRecord(h->ev_c2, Tc) != hipSuccess || hipStreamWaitEvent(B, h->ev_c2, 0) != hipSuccess || hipStreamWaitEvent(B2, h->ev_c2, 0) != hipSuccess)
            return DIST_ERR_STATE;
        return DIST_OK;
    };
    auto fork = [&]() -> int {           // B and B2 wait for everything enqueued on A so far
        hipEvent_t e = h->ev_a[evn++];
        if (hipEventRecord(e, A) != hipSuccess || hipStreamWaitEvent(B, e, 0) != hipSuccess || hipStreamWaitEvent(B2, e, 0) != hipSuccess)
            return DIST_ERR_STATE;
        return DIST_OK;
    };
    auto merge_b2 = [&]() -> int {       // fold B2 into B so one event on B covers both
        if (hipEventRecord(h->ev_b2, B2) != hipSuccess || hipStreamWaitEvent(B, h->ev_b2, 0) != hipSuccess) return DIST_ERR_STATE;
        return DIST_OK;
    };
    // the ada / head part above ran on A and produced dFz in h->dR; B must also see the zeroed gradient buffer
    RUN(fork());
    if (h->grad_hook) h->grad_hook(h->grad_hook_user, h->tail_begin, h->total[0]);      // ada-pooling + head gradients are final on A

    const void* dR = h->dR;               // dL/dR_i (for the last layer: dFz)
    const void* dXn = nullptr;            // dL/dX_{i+1} (none for the last layer)
    int hook_next = nl - 1;               // next layer slice to report through the gradient-ready hook
    for (int i = nl - 1; i >= 0; --i) {
        const DistLayer& l = h->dl[i];
        DistLayerWs& w = h->lw[i];
        dist_handle::BwdSet& q = h->bs[i & 1];
        const bool last = (i == nl - 1);
        if (i + 2 < nl) {                 // set (i & 1) was last used by layer i+2: its weight gradients must be done
            HIP_CHECK_RET(hipStreamWaitEvent(A, h->ev_b_done[i + 2], 0));
            for (; hook_next >= i + 2; --hook_next)
                if (h->grad_hook) h->grad_hook(h->grad_hook_user, h->layer_begin[hook_next], h->layer_end[hook_next]);
        }
        if (i + 1 < nl) HIP_CHECK_RET(hipStreamWaitEvent(A, h->ev_b_dr[i + 1], 0));    // dR_{i+1} (set (i & 1) of layer i+2 ... see below)

        // ---- IntegrationNetwork backward (dist.py:40-45) ----
        // B: dW/db of ffn.c_proj and temporal_ffn.c_proj read dR (produced before this layer started)
        const int Cc = Cf + C4;            // [zf | h2], [hf | g2], [dzf | dh2] rows
        RUN(wgrad_pair(xb, l.ffn_proj, l.tf_proj, dR, Ci, w.hf, Cc, rowsS));
        RUN(merge_b2());
        HIP_CHECK_RET(hipEventRecord(h->ev_b_dr[i], B));
        if (h->ig_bwd) {
            // one fused launch (integ.hip): [dzf | dh2], dh1 and dM' = LN'(dzf Wa' + dh1 Wb') (+ dFz for the last layer; a second copy becomes dM)
            dist_integ_bwd_args ba;
            memset(&ba, 0, sizeof(ba));
            ba.dR = dR; ba.zf_h2 = w.zf; ba.Xhat = w.Na; ba.rstd = w.in_rstd; ba.B1 = l.ig_B1; ba.B2 = l.ig_B2; ba.B3 = l.ig_B3;
            // the three gradients leave in ONE buffer, rows [dzf | dh1 | dh2] of Ci + 2 C4: the weight gradients of ffn.c_fc and temporal_ffn.c_fc1 (both
            // against xhat, their parameters side by side in the flat buffers) are one GEMM over its first Ci + C4 columns
            const int Cd = Ci + 2 * C4;
            char* dcat = static_cast<char*>(q.dcat);
            void* d_h1 = dcat + (size_t)Ci * es; void* d_h2 = dcat + (size_t)(Ci + C4) * es;
            ba.dzf_dh2 = q.dcat; ba.ld_dzf = Cd; ba.dh2 = d_h2; ba.ld_dh2 = Cd; ba.dh1 = d_h1; ba.ld_dh1 = Cd;
            // (dM = dM' + I2T term: the I2T data-gradient GEMM below reads dM' as its residual and writes every patch row of dM; only the cls rows come from here)
            ba.dMp = q.dMp; ba.dM_copy = last ? nullptr : q.dM; ba.dM_cls_only = 1; ba.add_dR = last ? 1 : 0;
            ba.t2i_dcls = x.gr(l.cls_token);
            if (h->ig_i2tb && !last && !tchain) {             // I2T backward in the same launch: dY (for the I2T weight gradient) and dM = dM' + [0 ; dY Wi] leave it
                ba.i2t_dXnext = dXn; ba.i2t_B = l.ig_W4; ba.i2t_dY = q.dY; ba.dM_cls_only = 0;
            }
            if (h->ig_t2ib && !tchain) {                      // ... and the T2I backward: dp = (dX_next + conv^T(dM')) g'(p), what the TemporalNet backward starts from
                ba.t2i_B = l.ig_W5; ba.t2i_p = w.p; ba.t2i_dp = q.dp;
                if (!last) ba.i2t_dXnext = dXn;
            }
            ba.clips = (int)b; ba.t = t; ba.L = L; ba.Ci = Ci; ba.C4 = C4; ba.tk = l.tf_fc2.taps; ba.dtype = c.dtype;
            RUN(dist_op_integration_bwd(&ba, x.s));
            RUN(fork());
            Lin both = l.ffn_fc;                              // [Ci + C4][Ci]: ffn.c_fc.weight followed by temporal_ffn.c_fc1.weight (and the two biases)
            both.N = Ci + C4;
            static const bool merge_env = (DIST_AB_KNOB("DIST_AMD_INTEG_WG_MERGE", 1) != 0);
            // (accumulating backward: G' and db of the two folded Linears go to this layer's scratch, [Ci + C4][Ci] then [Ci + C4])
            float* gs_w = h->bwd_accumulate ? h->ig_gscratch + (long)i * h->ig_gscratch_elems : nullptr;
            float* gs_b = gs_w ? gs_w + (long)(Ci + C4) * Ci : nullptr;
            if (merge_env && l.tf_fc1.w == l.ffn_fc.w + (long)Ci * Ci && l.tf_fc1.bias == l.ffn_fc.bias + Ci) {
                RUN(wgrad(xb, both, q.dcat, Cd, w.Na, Ci, rowsS, RM(), RM(), 0, true, gs_w, gs_b));
            } else {
                RUN(wgrad(xb, l.ffn_fc, q.dcat, Cd, w.Na, Ci, rowsS, RM(), RM(), 0, true, gs_w, gs_b));
                RUN(wgrad(xb2, l.tf_fc1, d_h1, Cd, w.Na, Ci, rowsS, RM(), RM(), 0, true, gs_w ? gs_w + (long)Ci * Ci : nullptr, gs_b ? gs_b + Ci : nullptr));
            }
            RUN(wgrad(xb2, l.tf_fc2, d_h2, Cd, w.h1, C4, rowsS, RM(), RM(DIST_RM_SHIFT, t * L, L, 1), 1, true));
        } else {
        // [dzf | dh2] = (dR [Wp ; W3]) * g'([zf | h2]): one data-gradient GEMM for the two projections
        RUN(gemm(x, dR, Ci, x.pk(l.pk_proj_b), rowsS, Cc, Ci, 1, q.dzf, Cc, nullptr, nullptr, w.zf, nullptr));
        RUN(fork());
        float* gs_w = h->bwd_accumulate ? h->ig_gscratch + (long)i * h->ig_gscratch_elems : nullptr;
        float* gs_b = gs_w ? gs_w + (long)(Ci + C4) * Ci : nullptr;
        RUN(wgrad(xb, l.ffn_fc, q.dzf, Cc, w.Na, Ci, rowsS, RM(), RM(), 0, true, gs_w, gs_b));
        RUN(wgrad(xb2, l.tf_fc2, q.dh2, Cc, w.h1, C4, rowsS, RM(), RM(DIST_RM_SHIFT, t * L, L, 1), 1, true));
        RUN(gemm(x, q.dh2, Cc, x.pk(l.tf_fc2.pk.b), rowsS, C4, C4, l.tf_fc2.taps, q.dh1, C4, nullptr, nullptr, nullptr, nullptr,
                 RM(DIST_RM_SHIFT, t * L, L, -1)));
        RUN(fork());
        RUN(wgrad(xb2, l.tf_fc1, q.dh1, C4, h->ig_xhat ? w.Na : w.Nb, Ci, rowsS, RM(), RM(), 0, true,     // (ig_xhat: w.Na holds xhat, see the unfold below)
                  gs_w ? gs_w + (long)Ci * Ci : nullptr, gs_b ? gs_b + Ci : nullptr));
        RUN(lin_dx(h, x, l.ffn_fc, q.dzf, rowsS, q.dNa, nullptr, nullptr, Cc));
        RUN(lin_dx(h, x, l.tf_fc1, q.dh1, rowsS, q.dNb));
        // dM' = LN'(dNa, dNb) (+ dFz for the last layer); a second copy becomes dM (updated in place by the I2T term)
        RUN(ln_bwd(x, l.in_ln, w.Mp, w.in_mean, w.in_rstd, q.dNa, q.dMp, false, rowsS, &l.in_ln_t, q.dNb, last ? dR : nullptr, last ? nullptr : q.dM, !h->ig_xhat));
        }
        // ---- T2I backward (dist.py:81-86): M' = M + [cls_token ; conv_strided(X')] ----
        RUN(fork());
        if (!h->ig_bwd) RUN(dist_k_cls_rows_bwd(q.dMp, x.gr(l.cls_token), (int)bt, L, Ci, t, c.dtype, B));      // (the fused backward adds the cls rows of dM' itself)
        RUN(wgrad(xb2, l.t2i, q.dMp, Ci, w.Xp, Ct, rowsQ, RM(DIST_RM_SKIPCLS, N), RM(DIST_RM_STRIDED, al, N), 2, true));
        // dX' = dX_next (identity, absent for the last layer) + conv^T(dQ): column block a of row (bj,n) -> frame bj*alpha+a
        // ... and straight through X' = g(p): dp = (dX_next + conv^T(dQ)) * g'(p) in the same epilogue (no dX' tensor, no
        // separate activation-backward pass)
        if (tchain) {
            HIP_CHECK_RET(hipEventRecord(h->ev_dmp[i], A));
            HIP_CHECK_RET(hipStreamWaitEvent(Tc, h->ev_dmp[i], 0));
        }
        if (!(h->ig_bwd && h->ig_t2ib && !tchain))
        RUN(gemm(xt, q.dMp, Ci, x.pk(l.t2i.pk.b), rowsQ, al * Ct, Ci, 1, q.dp, Ct, nullptr, last ? nullptr : dXn, w.p, nullptr,
                 RM(DIST_RM_SKIPCLS, N), OM(DIST_OM_SPLITCOLS, al, N, Ct), DIST_EPI_MULG_POST));
        // ---- I2T backward (dist.py:100-105): X_next = X' + upsample(Linear(M[1:])) ----
        const void* dM = q.dMp;            // last layer: no I2T path, dM = dM'
        if (!last) {
            if (!(h->ig_i2tb && h->ig_bwd && !tchain)) {
                if (tchain) HIP_CHECK_RET(hipStreamWaitEvent(A, h->ev_dx[i + 1], 0));       // dX_{i+1} comes from the temporal chain
                RUN(dist_k_pair_sum(dXn, q.dY, bt, N * Ct, al, c.dtype, A));
                RUN(gemm(x, q.dY, Ct, x.pk(l.i2t.pk.b), rowsQ, Ci, Ct, 1, q.dM, Ci, nullptr, h->ig_bwd ? q.dMp : q.dM, nullptr, nullptr, RM(), OM(DIST_OM_INSERTCLS, N)));
            }
            dM = q.dM;
        }
        RUN(fork());
        if (!last) RUN(wgrad(xb2, l.i2t, q.dY, Ct, w.M, Ci, rowsQ, RM(), RM(DIST_RM_SKIPCLS, N), 0, true));
        // ---- mid_feat = input_linear(F_i) + R_{i-1}: no dF_i (frozen ViT) ----
        RUN(wgrad(xb, l.in_lin, dM, Ci, h->feat[h->sel[i]], d, rowsS, RM(), RM(), 0, true));
        // ---- TemporalNet backward (dist.py:63-65): X' = g(p), p = X + conv3x3(V) + b, V = g(z), z = conv_t(U), U = LN(X) ----
        // bf16: two fused launches (tnet.hip): dz = conv3x3^T(dp) * g'(z); dX = dp + LN'(conv_t^T(dz)) with the LayerNorm parameter
        // gradients as per-workgroup partial rows (no atomics); otherwise two row-mapped GEMMs + the LayerNorm backward kernel
        const bool tn_fused = Ch == Ct && dist_k_tnet_fwd_eligible(c.dtype, Ct, h->G, l.tn_fc1.taps) &&
                              (dist_knob("DIST_AMD_TNET_BWD_FUSED", 1) != 0);   // measurement knob
        if (h->skip & 2) {
        } else if (tn_fused) {
            dist_tnet_bwd_args ta;
            memset(&ta, 0, sizeof(ta));
            ta.dp = q.dp; ta.z = w.z; ta.X = w.X; ta.mean = w.tn_mean; ta.rstd = w.tn_rstd; ta.ln_w = x.th(l.tn_ln.w);
            ta.W1b = x.pk(l.tn_fc1.pk.b); ta.W2b = x.pk(l.tn_fc2.pk.b);
            ta.dz = q.dz; ta.dX = q.dXo; ta.dgamma = x.gr(l.tn_ln.w); ta.dbeta = x.gr(l.tn_ln.b);
            ta.scratch = h->tnb_scratch; ta.scratch_elems = h->tnb_scratch_elems;
            ta.clips = b; ta.T = T; ta.G = h->G; ta.Ct = Ct; ta.tk = l.tn_fc1.taps; ta.dtype = c.dtype;
            ta.phase = 1;                                              // dz first: the weight-gradient streams start on it
            RUN(dist_op_temporal_net_bwd(&ta, xt.s));
        } else {
            RUN(gemm(xt, q.dp, Ct, x.pk(l.tn_fc2.pk.b), rowsX, Ch, Ct, 9, q.dz, Ch, nullptr, nullptr, w.z, nullptr, RM(DIST_RM_SPATIAL, h->G, 0, -1)));
        }
        RUN(fork());
        RUN(fork_t());
        RUN(wgrad(xb, l.tn_fc2, q.dp, Ct, w.V, Ch, rowsX, RM(), RM(DIST_RM_SPATIAL, h->G, 0, 1), 1, true));
        RUN(wgrad(xb2, l.tn_fc1, q.dz, Ch, w.U, Ct, rowsX, RM(), RM(DIST_RM_SHIFT, T * N, N, 1), 1, true));
        RUN(merge_b2());
        if (h->ig_xhat && !(h->skip & 1)) {   // the weight gradients of ffn.c_fc / temporal_ffn.c_fc1 were taken against xhat: unfold them (+ the two LayerNorms' gradients)
            dist_integ_unfold_args ua;
            memset(&ua, 0, sizeof(ua));
            if (h->bwd_accumulate) {
                const float* gs_w = h->ig_gscratch + (long)i * h->ig_gscratch_elems; const float* gs_b = gs_w + (long)(Ci + C4) * Ci;
                ua.g_ffn_fc_w = gs_w; ua.g_ffn_fc_b = gs_b; ua.g_tf_fc1_w = gs_w + (long)Ci * Ci; ua.g_tf_fc1_b = gs_b + Ci;
            }
            ua.ffn_fc_w = x.th(l.ffn_fc.w); ua.ln_w = x.th(l.in_ln.w); ua.ln_b = x.th(l.in_ln.b);
            ua.d_ffn_fc_w = x.gr(l.ffn_fc.w); ua.d_ffn_fc_b = x.gr(l.ffn_fc.bias); ua.d_ln_w = x.gr(l.in_ln.w); ua.d_ln_b = x.gr(l.in_ln.b);
            ua.tf_fc1_w = x.th(l.tf_fc1.w); ua.ln_t_w = x.th(l.in_ln_t.w); ua.ln_t_b = x.th(l.in_ln_t.b);
            ua.d_tf_fc1_w = x.gr(l.tf_fc1.w); ua.d_tf_fc1_b = x.gr(l.tf_fc1.bias); ua.d_ln_t_w = x.gr(l.in_ln_t.w); ua.d_ln_t_b = x.gr(l.in_ln_t.b);
            ua.Ci = Ci; ua.C4 = C4;
            RUN(dist_op_integration_unfold(&ua, B));
        }
        HIP_CHECK_RET(hipEventRecord(h->ev_b_done[i], B));
        if (h->dummy & 2) for (int r = 0; r < h->dummy_reps; ++r) RUN(ln_fwd(x, h->visual, h->vit[0].ln1, h->feat[h->sel[i]], nullptr, rowsS, h->lnstats2, h->lnstats2 + rowsS));
        if (h->dummy & 4) for (int r = 0; r < h->dummy_reps; ++r) RUN(ln_fwd(xb, h->visual, h->vit[0].ln1, h->feat[h->sel[i]], nullptr, rowsS, h->lnstats3, h->lnstats3 + rowsS));
        if (!(h->skip & 2) && tn_fused) {
            dist_tnet_bwd_args ta;
            memset(&ta, 0, sizeof(ta));
            ta.dp = q.dp; ta.z = w.z; ta.X = w.X; ta.mean = w.tn_mean; ta.rstd = w.tn_rstd; ta.ln_w = x.th(l.tn_ln.w);
            ta.W1b = x.pk(l.tn_fc1.pk.b); ta.W2b = x.pk(l.tn_fc2.pk.b);
            ta.dz = q.dz; ta.dX = q.dXo; ta.dgamma = x.gr(l.tn_ln.w); ta.dbeta = x.gr(l.tn_ln.b);
            ta.scratch = h->tnb_scratch + (long)i * h->tnb_scratch_elems; ta.scratch_elems = h->tnb_scratch_elems;
            ta.clips = b; ta.T = T; ta.G = h->G; ta.Ct = Ct; ta.tk = l.tn_fc1.taps; ta.dtype = c.dtype;
            ta.phase = 2;
            // measurement knob: leave the LayerNorm parameter gradients unsummed (phase 3).  The step does not change (19.64 vs 19.70 ms),
            // so one multi-layer dist_op_temporal_net_bwd_reduce at the end of backward would buy nothing: the per-layer sum stays here
            static const bool no_reduce = (dist_measure_knob("DIST_AMD_TNET_BWD_NOREDUCE", 0) != 0);
            if (no_reduce) ta.phase = 3;
            RUN(dist_op_temporal_net_bwd(&ta, xt.s));
        }
        if (!(h->skip & 2) && !tn_fused) {
            RUN(gemm(xt, q.dz, Ch, x.pk(l.tn_fc1.pk.b), rowsX, Ct, Ch, l.tn_fc1.taps, q.dU, Ct, nullptr, nullptr, nullptr, nullptr,
                     RM(DIST_RM_SHIFT, T * N, N, -1)));
            RUN(ln_bwd(xt, l.tn_ln, w.X, w.tn_mean, w.tn_rstd, q.dU, q.dXo, false, rowsX, nullptr, nullptr, q.dp));   // dX_i = dp + LN'(dU)
        }
        if (tchain) HIP_CHECK_RET(hipEventRecord(h->ev_dx[i], Tc));
        dR = dM;                          // dL/dR_{i-1}
        dXn = q.dXo;
    }
    // temporal stem (dist.py:178-181): no input gradient
    RUN(fork());
    RUN(fork_t());
    RUN(wgrad(xb, h->stem, dXn, Ct, h->patches, h->Kp, rowsX, RM(), RM(DIST_RM_SHIFT, T * N, N, 1), 3, true));
    RUN(merge_b2());
    // join: the caller's stream continues only after every weight gradient is complete
    HIP_CHECK_RET(hipEventRecord(h->ev_join, B));
    HIP_CHECK_RET(hipStreamWaitEvent(A, h->ev_join, 0));
    mark(h, DIST_MARK_BWD_END, A);
    if (h->grad_hook) {
        for (; hook_next >= 0; --hook_next) h->grad_hook(h->grad_hook_user, h->layer_begin[hook_next], h->layer_end[hook_next]);
        h->grad_hook(h->grad_hook_user, 0, h->layer_begin[0]);
    }
    return DIST_OK;
}

extern "C" int dist_set_inference(dist_handle* h, int on) {
    if (!h) return DIST_ERR_ARG;
    h->inference = on != 0;
    return DIST_OK;
}

extern "C" int dist_set_grad_ready_hook(dist_handle* h, dist_grad_ready_fn fn, void* user) {
    if (!h) return DIST_ERR_ARG;
    h->grad_hook = fn; h->grad_hook_user = user;
    return DIST_OK;
}

extern "C" int dist_marks_enable(dist_handle* h, int on) {
    if (!h) return DIST_ERR_ARG;
    if (on) for (hipEvent_t& e : h->mark_ev) if (!e) HIP_CHECK_RET(hipEventCreate(&e));
    h->marks_on = on != 0;
    return DIST_OK;
}
// ms[i] = time of mark i of the LAST step relative to DIST_MARK_BWD_END of the step before it is not available from events of one
// step alone, so the reference point is this step's own DIST_MARK_VIT_BEGIN (the first thing a pipelined step queues);
// a mark that was never recorded reads as NaN.  Synchronises with the device.
extern "C" int dist_marks_read(dist_handle* h, float* ms, int n) {
    if (!h || !ms || n < DIST_NMARKS) return DIST_ERR_ARG;
    if (!h->mark_ev[DIST_MARK_VIT_BEGIN]) return fail(h, DIST_ERR_STATE, "dist_marks_read before dist_marks_enable");
    HIP_CHECK_RET(hipDeviceSynchronize());
    for (int i = 0; i < DIST_NMARKS; ++i) {
        float t = 0.f;
        ms[i] = hipEventElapsedTime(&t, h->mark_ev[DIST_MARK_VIT_BEGIN], h->mark_ev[i]) == hipSuccess ? t : __builtin_nanf("");
    }
    return DIST_OK;
}

extern "C" int dist_profile_begin(dist_handle* h) {
    if (!h) return DIST_ERR_ARG;
    h->prof_on = true; h->prof_n = 0; h->prof_flops = 0.0;
    return DIST_OK;
}
extern "C" int dist_profile_end(dist_handle* h, double* ms_total, double* flops_total, int* launches) {
    if (!h || !ms_total || !flops_total || !launches) return DIST_ERR_ARG;
    h->prof_on = false;
    double ms = 0.0;
    for (int i = 0; i + 1 < h->prof_n; i += 2) {
        HIP_CHECK_RET(hipEventSynchronize(h->prof_ev[i + 1]));
        float t = 0.f;
        HIP_CHECK_RET(hipEventElapsedTime(&t, h->prof_ev[i], h->prof_ev[i + 1]));
        ms += t;
    }
    *ms_total = ms; *flops_total = h->prof_flops; *launches = h->prof_n / 2;
    return DIST_OK;
}

extern "C" int dist_debug_tensor(dist_handle* h, const char* name, const void** ptr, int64_t* rows, int* cols) {
    if (!h || !name || !ptr || !rows || !cols) return DIST_ERR_ARG;
    if (!h->ws) return fail(h, DIST_ERR_UNBOUND, "dist_debug_tensor before dist_bind");
    const dist_config& c = h->cfg;
    const long b = h->fwd_b;
    const long rowsX = b * c.frames * h->N, rowsS = b * h->t * h->L;
    int i = -1;
    const char* dot = strchr(name, '.');
    if (dot) i = atoi(dot + 1);
    const std::string key(name, dot ? (size_t)(dot - name) : strlen(name));
    auto vit_ok = [&]() { return i >= 0 && i < c.layers; };            // "feat.i": ViT block i; every other name: DiST layer i (< number of selected blocks)
    auto layer_ok = [&]() { return i >= 0 && i < h->nsel; };
    if (key == "feat" && vit_ok()) {
        if (!h->slot[h->cur].valid[i])
            return fail(h, DIST_ERR_STATE, "feat.%d: the current feature slot was filled by dist_features_import without block %d (it still holds an older clip)", i, i);
        *ptr = h->feat[i]; *rows = rowsS; *cols = c.width; return DIST_OK;
    }
    if (key == "stem") { *ptr = h->lw[0].X; *rows = rowsX; *cols = c.temporal_dim; return DIST_OK; }
    if (key == "tn_out" && layer_ok()) { *ptr = h->lw[i].Xp; *rows = rowsX; *cols = c.temporal_dim; return DIST_OK; }
    if (key == "int_out" && layer_ok()) { *ptr = h->lw[i].R; *rows = rowsS; *cols = c.integration_dim; return DIST_OK; }
    if (key == "x_temporal" && layer_ok()) { *ptr = (i + 1 < h->nsel) ? h->lw[i + 1].X : h->Xlast; *rows = rowsX; *cols = c.temporal_dim; return DIST_OK; }
    if (key == "mid" && layer_ok()) {
        // with T2I formed inside the fused forward and the fused backward reading xhat, M' is only written for the last layer (DIST_AMD_KEEP_MID=1 keeps all)
        if (h->ig_t2i && h->ig_bwd && !h->keep_mid && i != h->nsel - 1)
            return fail(h, DIST_ERR_STATE, "mid.%d is not materialised by the fused IntegrationNetwork kernels (set DIST_AMD_KEEP_MID=1)", i);
        if (h->branch_infer || h->inference) return fail(h, DIST_ERR_STATE, "mid.%d is not written in inference mode", i);
        *ptr = h->lw[i].Mp; *rows = rowsS; *cols = c.integration_dim; return DIST_OK;
    }
    if (key == "patches") { *ptr = h->patches; *rows = rowsX; *cols = h->Kp; return DIST_OK; }
    if (key == "vit_ln_out") { *ptr = h->hbuf; *rows = rowsS; *cols = c.width; return DIST_OK; }   // LayerNorm output scratch of the ViT (unused when folded)
    return fail(h, DIST_ERR_ARG, "unknown debug tensor %s", name);
}
