// Engine, part 2: the frozen CLIP ViT pass into a feature slot (dist_vit_forward / prefetch / adopt, dist_features_import).
#include "engine_internal.h"

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
        S.valid.assign(c.layers, 0);                                    // a new clip enters the slot: block i is readable again once ITS launch below is issued (a pass issued in parts)
        mark(h, DIST_MARK_VIT_BEGIN, x.s);
        // (dist_vit_mix_next: the batch-mode Mixup / CutMix of this batch is applied while its frames are gathered - one pass over them instead of three)
        RUN(dist_op_patchify_mixed(video, S.patches, b, c.frames, c.resolution, c.resolution, c.patch, c.dtype, h->mix.kind, h->mix.lam, h->mix.oml,
                                   h->mix.yl, h->mix.yh, h->mix.xl, h->mix.xh, stream));
        h->mix.kind = 0;
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

extern "C" int dist_vit_mix_next(dist_handle* h, int kind, float lam, float one_minus_lam, int yl, int yh, int xl, int xh) {
    if (!h || kind < 0 || kind > 2) return DIST_ERR_ARG;
    const int R = h->cfg.resolution;
    if (kind == 2 && (yl < 0 || yh > R || xl < 0 || xh > R || yl > yh || xl > xh)) return fail(h, DIST_ERR_ARG, "dist_vit_mix_next: box [%d,%d) x [%d,%d) outside the %d x %d frame", yl, yh, xl, xh, R, R);
    h->mix.kind = kind; h->mix.lam = lam; h->mix.oml = one_minus_lam; h->mix.yl = yl; h->mix.yh = yh; h->mix.xl = xl; h->mix.xh = xh;
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

