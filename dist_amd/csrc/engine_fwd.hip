// Engine, part 3: DiSTNetwork forward + cosine logits + soft-target cross entropy (dist_branch_forward, dist_loss).
#include "engine_internal.h"

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

