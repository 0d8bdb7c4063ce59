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
#include "engine_internal.h"

namespace {

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

extern "C" int dist_set_inference(dist_handle* h, int on) {
    if (!h) return DIST_ERR_ARG;
    h->inference = on != 0;
    // (round 6: moving the prefetch stream to default priority for forward-only loops - where the ViT pass is the critical path - was worth 0.2 ms of 12.3 when the
    //  stream is CREATED that way (timing-only library, DIST_AMD_PF_PRIO=0: 12.26 / 12.30 -> 12.06 / 12.09 ms), and cost 1.7 ms when dist_set_inference re-created it:
    //  the new stream lands on a hardware queue another of the step's streams already uses (11.82 -> 13.58 ms).  The stream keeps the priority it was created with.)
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
