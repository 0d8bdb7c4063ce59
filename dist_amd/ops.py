"""Thin torch-tensor wrappers over the operator-level C ABI (include/dist_amd.h).

torch is only the device-memory container here: every function passes raw device
pointers + sizes to libdist_amd.so on torch's current HIP stream.  No fallback: a
missing library or a non-zero return code raises DistError.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L


def _dt(t):
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rowmap(mode=L.RM_PLAIN, p0=0, p1=0, sign=1):
    return L.RowMap(mode, p0, p1, sign)


def outmap(mode=L.OM_PLAIN, p0=0, p1=0, p2=0):
    return L.OutMap(mode, p0, p1, p2)


def gemm_nt(A, B, M, N, K, *, taps=1, bias=None, res=None, aux=None, amap=None, omap=None,
            C_out=None, C2_out=None, lda=None, ldc=None, ldc2=None, mulg_post=False, lnfold=None, rowstats=None, fp8=None, out8=None, act_only8=False):
    """C[omap(m)][n] = epi(sum_tap sum_k A[amap(m,tap)][k] B[n][tap*K+k]); returns None (writes C_out / C2_out)."""
    lib = L.load()
    a = L.GemmArgs()
    a.A, a.B, a.C, a.C2 = _p(A), _p(B), _p(C_out), _p(C2_out)
    a.bias, a.res, a.aux = _p(bias), _p(res), _p(aux)
    a.M, a.N, a.K, a.taps = M, N, K, taps
    a.lda = lda if lda is not None else A.shape[-1]
    a.ldb = B.shape[-1]
    ldo = ldc if ldc is not None else (C_out.shape[-1] if C_out is not None else (C2_out.shape[-1] if C2_out is not None else N))
    a.ldc = ldo
    a.ldc2 = ldc2 if ldc2 is not None else (C2_out.shape[-1] if C2_out is not None else ldo)
    a.ldres = res.shape[-1] if res is not None else ldo
    a.ldaux = aux.shape[-1] if aux is not None else ldo
    a.amap = amap or rowmap()
    a.omap = omap or outmap()
    a.flags = (L.EPI_BIAS if bias is not None else 0) | (L.EPI_RES if res is not None else 0) | \
              (L.EPI_MULG if aux is not None else 0) | (L.EPI_ACT2 if C2_out is not None else 0) | \
              (L.EPI_MULG_POST if mulg_post else 0)
    if lnfold is not None:          # (stats fp32 [2][M] = mean then rstd, colsum fp32 [N]): LayerNorm folded into the GEMM (EPI_LNFOLD)
        stats, colsum = lnfold
        a.aux, a.bias2 = _p(stats), _p(colsum)
        a.flags |= L.EPI_LNFOLD
    if rowstats is not None:        # fp32 [N/64][M][2]: (sum, sum of squares) of the stored values per 64-column slice (EPI_ROWSTATS)
        a.rowstats = _p(rowstats)
        a.flags |= L.EPI_ROWSTATS
    if out8 is not None:            # (C8 uint8 [M, N], scale fp32 [1], amax fp32 [1] or None): e4m3 image of the output (EPI_OUT8)
        c8, sc, am = out8
        a.C8, a.ldc8, a.out8_scale, a.out8_amax = _p(c8), c8.shape[-1], _p(sc), _p(am)
        a.flags |= L.EPI_OUT8 | (L.EPI_ACT2 if act_only8 else 0)      # act_only8: C8 = e4m3(quickgelu(v)) with no bf16 output at all
    if fp8 is not None:             # (a_scale fp32 [M] or [1], b_scale fp32 [N]): A / B hold e4m3 bytes (quant_rows_fp8), outputs are bf16 (EPI_FP8)
        assert A.dtype == torch.uint8 and B.dtype == torch.uint8
        a.a_scale, a.b_scale = _p(fp8[0]), _p(fp8[1])
        a.flags |= L.EPI_FP8 | (L.EPI_FP8_ASCALAR if fp8[0].numel() == 1 else 0)
        a.dtype = L.BF16
    else:
        a.dtype = _dt(A)
    L.check(lib.dist_op_gemm_nt(C.byref(a), _stream()))


def pack_conv_taps(w):
    """Conv3d weight [Co, Ci, kt, kh, kw] (fp32 master, reference layout) -> the packed forward layout [Co][tap*Ci + ci] in bf16,
    tap = (t*kh + y)*kw + x - what dist_pack_weights keeps for the row-mapped GEMMs and the fused TemporalNet."""
    co, ci = w.shape[0], w.shape[1]
    return w.reshape(co, ci, -1).permute(0, 2, 1).reshape(co, -1).to(torch.bfloat16).contiguous()


def temporal_net_fwd(X, W1p, b1, W2p, b2, ln_w, ln_b, clips, T, G, *, tk=3, save_uv=False, eps=1e-5):
    """Fused TemporalNet forward (dist_op_temporal_net_fwd): X bf16 [clips*T*G*G, Ct]; W1p / W2p from pack_conv_taps.
    Returns dict(z, p, Xp, mean, rstd[, U, V])."""
    lib = L.load()
    rows, Ct = X.shape
    assert rows == clips * T * G * G and X.dtype == torch.bfloat16 and X.is_contiguous()
    out = {k: torch.empty_like(X) for k in (("z", "p", "Xp", "U", "V") if save_uv else ("z", "p", "Xp"))}
    out["mean"] = torch.empty(rows, dtype=torch.float32, device=X.device)
    out["rstd"] = torch.empty(rows, dtype=torch.float32, device=X.device)
    a = L.TnetArgs()
    a.X, a.W1, a.W2, a.b1, a.b2, a.ln_w, a.ln_b = _p(X), _p(W1p), _p(W2p), _p(b1), _p(b2), _p(ln_w), _p(ln_b)
    a.z, a.p, a.Xp, a.U, a.V = _p(out["z"]), _p(out["p"]), _p(out["Xp"]), _p(out.get("U")), _p(out.get("V"))
    a.mean, a.rstd = _p(out["mean"]), _p(out["rstd"])
    a.clips, a.T, a.G, a.Ct, a.tk, a.dtype, a.eps = clips, T, G, Ct, tk, L.BF16, eps
    L.check(lib.dist_op_temporal_net_fwd(C.byref(a), _stream()))
    return out


def pack_conv_taps_dgrad(w):
    """Conv3d weight [Co, Ci, kt, kh, kw] -> the data-gradient layout [Ci][tap*Co + co] in bf16 (PACK_B of dist_pack_weights)."""
    co, ci = w.shape[0], w.shape[1]
    return w.reshape(co, ci, -1).permute(1, 2, 0).reshape(ci, -1).to(torch.bfloat16).contiguous()


def temporal_net_bwd(dp, z, X, mean, rstd, ln_w, W1b, W2b, clips, T, G, *, tk=3, dgamma=None, dbeta=None, scratch=None, phase=0, dz=None):
    """Fused TemporalNet data-gradient backward (dist_op_temporal_net_bwd).  Returns dict(dz, dX, dgamma, dbeta, scratch); dgamma / dbeta
    are ACCUMULATED into when given.  phase 0 = everything; 1 = dz only; 2 = dX + parameter gradients; 3 = dX with the parameter
    gradients left as partial rows in `scratch` for temporal_net_bwd_reduce.  `dz` (phases 2, 3): the dz an earlier phase-1 call returned."""
    lib = L.load()
    rows, Ct = X.shape
    out = {"dz": dz if dz is not None else torch.empty_like(X), "dX": torch.empty_like(X),
           "dgamma": dgamma if dgamma is not None else torch.zeros(Ct, dtype=torch.float32, device=X.device),
           "dbeta": dbeta if dbeta is not None else torch.zeros(Ct, dtype=torch.float32, device=X.device)}
    n = lib.dist_op_temporal_net_bwd_scratch(clips, T, Ct)
    if scratch is None:
        scratch = torch.empty(n, dtype=torch.float32, device=X.device)
    out["scratch"] = scratch
    a = L.TnetBwdArgs()
    a.dp, a.z, a.X, a.mean, a.rstd, a.ln_w, a.W1b, a.W2b = _p(dp), _p(z), _p(X), _p(mean), _p(rstd), _p(ln_w), _p(W1b), _p(W2b)
    a.dz, a.dX, a.dgamma, a.dbeta, a.scratch, a.scratch_elems = _p(out["dz"]), _p(out["dX"]), _p(out["dgamma"]), _p(out["dbeta"]), _p(scratch), n
    a.clips, a.T, a.G, a.Ct, a.tk, a.dtype, a.phase = clips, T, G, Ct, tk, L.BF16, phase
    L.check(lib.dist_op_temporal_net_bwd(C.byref(a), _stream()))
    return out


def temporal_net_bwd_reduce(scratch, layers, clips, T, Ct, dgammas, dbetas):
    """Sum the partial rows `layers` phase-3 launches left in `scratch` ([layers, dist_op_temporal_net_bwd_scratch] floats) INTO
    dgammas[l] / dbetas[l] (dist_op_temporal_net_bwd_reduce): one launch for every layer."""
    lib = L.load()
    n = lib.dist_op_temporal_net_bwd_scratch(clips, T, Ct)
    FP = C.POINTER(C.c_float)
    dg = (FP * layers)(*[C.cast(_p(t), FP) for t in dgammas])
    db = (FP * layers)(*[C.cast(_p(t), FP) for t in dbetas])
    L.check(lib.dist_op_temporal_net_bwd_reduce(_p(scratch), n, layers, clips, T, Ct, dg, db, _stream()))


def integration_pack(w, bwd=False, t2i_w=None, i2t_w=None):
    """fp32 master weights of one IntegrationNetwork (dict with the reference's parameter names below `integration_nets.i.`) -> the operands
    of integration_fwd (dist_op_integration_pack): dict(W1, W2, W3 bf16; b1, b2, b3 fp32 [; B1, B2, B3 bf16 for integration_bwd])."""
    lib = L.load()
    Ci, C4 = w["ffn.c_fc.weight"].shape[0], w["temporal_ffn.c_fc1.weight"].shape[0]
    dev = w["ffn.c_fc.weight"].device
    n = [lib.dist_op_integration_pack_elems(Ci, C4, k) for k in range(6)]
    out = {"W1": torch.empty(n[0], dtype=torch.bfloat16, device=dev), "W2": torch.empty(n[1], dtype=torch.bfloat16, device=dev),
           "W3": torch.empty(n[2], dtype=torch.bfloat16, device=dev), "b1": torch.empty(n[3], dtype=torch.float32, device=dev),
           "b2": torch.empty(n[4], dtype=torch.float32, device=dev), "b3": torch.empty(n[5], dtype=torch.float32, device=dev)}
    if bwd:
        out.update({f"B{k + 1}": torch.empty(n[k], dtype=torch.bfloat16, device=dev) for k in range(3)})
    keep = [w[k].float().contiguous() for k in ("ffn.c_fc.weight", "ffn.c_fc.bias", "ln.weight", "ln.bias", "temporal_ffn.c_fc1.weight", "temporal_ffn.c_fc1.bias",
                                                "ln_temporal.weight", "ln_temporal.bias", "temporal_ffn.c_fc2.weight", "temporal_ffn.c_fc2.bias",
                                                "ffn.c_proj.weight", "ffn.c_proj.bias", "temporal_ffn.c_proj.weight", "temporal_ffn.c_proj.bias")]
    a = L.IntegPackArgs()
    (a.ffn_fc_w, a.ffn_fc_b, a.ln_w, a.ln_b, a.tf_fc1_w, a.tf_fc1_b, a.ln_t_w, a.ln_t_b, a.tf_fc2_w, a.tf_fc2_b,
     a.ffn_proj_w, a.ffn_proj_b, a.tf_proj_w, a.tf_proj_b) = [_p(t) for t in keep]
    a.W1, a.W2, a.W3, a.b1, a.b2, a.b3 = _p(out["W1"]), _p(out["W2"]), _p(out["W3"]), _p(out["b1"]), _p(out["b2"]), _p(out["b3"])
    a.Ci, a.C4 = Ci, C4
    if bwd:
        a.B1, a.B2, a.B3 = _p(out["B1"]), _p(out["B2"]), _p(out["B3"])
    if t2i_w is not None:                               # temporal2integration linear_fuse.weight [Ci][C4][2][1][1] -> the T2I operand in front of the forward
        keep.append(t2i_w.float().contiguous())
        out["Wt"] = torch.empty(lib.dist_op_integration_pack_elems(Ci, C4, 6), dtype=torch.bfloat16, device=dev)
        a.t2i_w, a.Wt = _p(keep[-1]), _p(out["Wt"])
        if bwd:
            out["W5"] = torch.empty(lib.dist_op_integration_pack_elems(Ci, C4, 9), dtype=torch.bfloat16, device=dev)
            a.W5 = _p(out["W5"])
    if i2t_w is not None:                               # integration2temporal linear_fuse.weight [C4][Ci] -> the I2T operand behind it
        keep.append(i2t_w.float().contiguous())
        out["Wi"] = torch.empty(lib.dist_op_integration_pack_elems(Ci, C4, 7), dtype=torch.bfloat16, device=dev)
        a.i2t_w, a.Wi = _p(keep[-1]), _p(out["Wi"])
        if bwd:
            out["W4"] = torch.empty(lib.dist_op_integration_pack_elems(Ci, C4, 8), dtype=torch.bfloat16, device=dev)
            a.W4 = _p(out["W4"])
    L.check(lib.dist_op_integration_pack(C.byref(a), _stream()))
    torch.cuda.current_stream().synchronize()          # (the fp32 copies in `keep` must outlive the launch)
    return out


def integration_fwd(Mp, pk, clips, t, Ltok, *, ln=None, train=True, tk=3, eps=1e-5, out=None, xhat=False, t2i=None, i2t_bias=None):
    """Fused IntegrationNetwork forward (dist_op_integration_fwd) on Mp [clips*t*Ltok, Ci] (bf16).  `pk` from integration_pack; `ln` =
    (ln.weight, ln.bias, ln_temporal.weight, ln_temporal.bias) fp32, needed when train (the tensors backward reads are written) unless
    xhat=True (the normalised rows themselves are kept instead of the two affine outputs).
    Returns dict(R [, Na, Nb | Xhat, mean, rstd, zf_h2, hf_g2, h1])."""
    lib = L.load()
    rows, Ci = Mp.shape
    C4 = pk["b2"].numel()
    reuse = out is not None                           # (buffers of an earlier call with the same shapes)
    out = out if reuse else {"R": torch.empty_like(Mp)}
    a = L.IntegArgs()
    a.Mp, a.W1, a.W2, a.W3, a.b1, a.b2, a.b3 = _p(Mp), _p(pk["W1"]), _p(pk["W2"]), _p(pk["W3"]), _p(pk["b1"]), _p(pk["b2"]), _p(pk["b3"])
    a.R = _p(out["R"])
    if train:
        if not reuse:
            out.update(mean=torch.empty(rows, dtype=torch.float32, device=Mp.device),
                       rstd=torch.empty(rows, dtype=torch.float32, device=Mp.device),
                       zf_h2=torch.empty(rows, Ci + C4, dtype=Mp.dtype, device=Mp.device), hf_g2=torch.empty(rows, Ci + C4, dtype=Mp.dtype, device=Mp.device),
                       h1=torch.empty(rows, C4, dtype=Mp.dtype, device=Mp.device))
            out.update({"Xhat": torch.empty_like(Mp)} if xhat else {"Na": torch.empty_like(Mp), "Nb": torch.empty_like(Mp)})
        if xhat:
            a.Xhat = _p(out["Xhat"])
        else:
            a.ln_w, a.ln_b, a.ln_t_w, a.ln_t_b = [_p(v) for v in ln]
            a.Na, a.Nb = _p(out["Na"]), _p(out["Nb"])
        a.mean, a.rstd, a.zf_h2, a.hf_g2, a.h1 = [_p(out[k]) for k in ("mean", "rstd", "zf_h2", "hf_g2", "h1")]
    if t2i is not None:                                 # (Xp, bias, cls_tokens): `Mp` is M, and M' = M + [cls ; conv_strided(Xp)] is formed in the kernel (returned as "Mp")
        Xp, tb, cls = t2i
        if "Mp" not in out:
            out["Mp"] = torch.empty_like(Mp)
        a.Mp = None
        a.t2i_M, a.t2i_Xp, a.t2i_W, a.t2i_bias, a.t2i_cls, a.Mp_out = _p(Mp), _p(Xp), _p(pk["Wt"]), _p(tb), _p(cls), _p(out["Mp"])
        if i2t_bias is not None:                        # I2T behind it: X of the next layer = Xp + upsample_t(M[:, 1:] Wi^T + bias) (returned as "Xnext")
            if "Xnext" not in out:
                out["Xnext"] = torch.zeros_like(Xp)
            a.i2t_W, a.i2t_bias, a.i2t_Xnext = _p(pk["Wi"]), _p(i2t_bias), _p(out["Xnext"])
    a.clips, a.t, a.L, a.Ci, a.C4, a.tk, a.dtype, a.eps = clips, t, Ltok, Ci, C4, tk, L.BF16, eps
    L.check(lib.dist_op_integration_fwd(C.byref(a), _stream()))
    return out


def integration_bwd(dR, saved, pk, clips, t, Ltok, *, add_dR=False, copy=False, tk=3, i2t_dXnext=None, t2i_p=None, t2i_dXnext=None):
    """Fused IntegrationNetwork data-gradient backward (dist_op_integration_bwd).  `saved`: what integration_fwd(..., xhat=True) returned;
    `pk` from integration_pack(w, bwd=True).  Returns dict(dzf_dh2, dh1, dMp [, dM])."""
    rows, Ci = dR.shape
    C4 = saved["h1"].shape[1]
    out = {"dzf_dh2": torch.empty(rows, Ci + C4, dtype=dR.dtype, device=dR.device), "dh1": torch.empty(rows, C4, dtype=dR.dtype, device=dR.device),
           "dMp": torch.empty_like(dR)}
    if copy:
        out["dM"] = torch.empty_like(dR)
    a = L.IntegBwdArgs()
    a.dR, a.zf_h2, a.Xhat, a.rstd = _p(dR), _p(saved["zf_h2"]), _p(saved["Xhat"]), _p(saved["rstd"])
    a.B1, a.B2, a.B3 = _p(pk["B1"]), _p(pk["B2"]), _p(pk["B3"])
    a.dzf_dh2, a.dh1, a.dMp, a.dM_copy = _p(out["dzf_dh2"]), _p(out["dh1"]), _p(out["dMp"]), _p(out.get("dM"))
    if i2t_dXnext is not None:                          # I2T backward behind it: "dM" = dM' + [0 ; dY Wi], "dY" = the frame-pair sums of dX_next
        out["dM"] = torch.empty_like(dR)
        out["dY"] = torch.zeros(clips * t * (Ltok - 1), C4, dtype=dR.dtype, device=dR.device)
        a.dM_copy, a.i2t_dXnext, a.i2t_B, a.i2t_dY = _p(out["dM"]), _p(i2t_dXnext), _p(pk["W4"]), _p(out["dY"])
    if t2i_p is not None:                               # T2I backward behind that: "dp" = (dX_next + conv^T(dM'[:, 1:])) * g'(p); t2i_dXnext None = the last layer
        out["dp"] = torch.zeros_like(t2i_p)
        a.t2i_B, a.t2i_p, a.t2i_dp = _p(pk["W5"]), _p(t2i_p), _p(out["dp"])
        if i2t_dXnext is None and t2i_dXnext is not None:
            a.i2t_dXnext = _p(t2i_dXnext)
    a.add_dR, a.clips, a.t, a.L, a.Ci, a.C4, a.tk, a.dtype = int(add_dR), clips, t, Ltok, Ci, C4, tk, L.BF16
    L.check(L.load().dist_op_integration_bwd(C.byref(a), _stream()))
    return out


def integration_unfold(w, g, gs=None):
    """backward side of the LayerNorm fold (dist_op_integration_unfold), in place on the fp32 gradient dict `g` (same keys as the weights `w`):
    g["ffn.c_fc.weight"] / g["temporal_ffn.c_fc1.weight"] hold dz^T xhat on entry.  `gs` (accumulating form): this pass's dz^T xhat and bias
    gradients of the two Linears (same four keys); `g` then holds EARLIER passes' gradients and is added to."""
    a = L.IntegUnfoldArgs()
    if gs is not None:
        a.g_ffn_fc_w, a.g_ffn_fc_b = _p(gs["ffn.c_fc.weight"]), _p(gs["ffn.c_fc.bias"])
        a.g_tf_fc1_w, a.g_tf_fc1_b = _p(gs["temporal_ffn.c_fc1.weight"]), _p(gs["temporal_ffn.c_fc1.bias"])
    a.ffn_fc_w, a.ln_w, a.ln_b = _p(w["ffn.c_fc.weight"]), _p(w["ln.weight"]), _p(w["ln.bias"])
    a.d_ffn_fc_w, a.d_ffn_fc_b, a.d_ln_w, a.d_ln_b = _p(g["ffn.c_fc.weight"]), _p(g["ffn.c_fc.bias"]), _p(g["ln.weight"]), _p(g["ln.bias"])
    a.tf_fc1_w, a.ln_t_w, a.ln_t_b = _p(w["temporal_ffn.c_fc1.weight"]), _p(w["ln_temporal.weight"]), _p(w["ln_temporal.bias"])
    a.d_tf_fc1_w, a.d_tf_fc1_b, a.d_ln_t_w, a.d_ln_t_b = (_p(g["temporal_ffn.c_fc1.weight"]), _p(g["temporal_ffn.c_fc1.bias"]),
                                                          _p(g["ln_temporal.weight"]), _p(g["ln_temporal.bias"]))
    a.Ci, a.C4 = w["ffn.c_fc.weight"].shape[0], w["temporal_ffn.c_fc1.weight"].shape[0]
    L.check(L.load().dist_op_integration_unfold(C.byref(a), _stream()))


def ln_fold(W, bias, gamma, beta):
    """LayerNorm -> Linear fold (dist_op_ln_fold): returns (Wp bf16 [N,K] = W*gamma, colsum fp32 [N], bias' fp32 [N])."""
    lib = L.load()
    N, K = W.shape
    Wp = torch.empty(N, K, dtype=torch.bfloat16, device=W.device)
    colsum = torch.empty(N, dtype=torch.float32, device=W.device)
    bout = torch.empty(N, dtype=torch.float32, device=W.device)
    L.check(lib.dist_op_ln_fold(_p(W), _p(bias), _p(gamma), _p(beta), _p(Wp), _p(colsum), _p(bout), N, K, _stream()))
    return Wp, colsum, bout


def gemm_tn(A, B, out, M, NI, K, *, taps=1, amap=None, bmap=None, so_i=None, so_tap=None, so_outer=1, inner=1, use_tr=1, colsum=None, partial=None,
            out2=None, split_c=0, so_i2=0, colsum2=None, max_blocks=0, lda=None, ldb=None):
    lib = L.load()
    a = L.GemmTnArgs()
    a.A, a.B, a.out = _p(A), _p(B), _p(out)
    a.M, a.NI, a.K, a.taps = M, NI, K, taps
    a.lda, a.ldb = lda if lda is not None else A.shape[-1], ldb if ldb is not None else B.shape[-1]      # (lda / ldb: column-offset views of wider buffers)
    a.amap = amap or rowmap()
    a.bmap = bmap or rowmap()
    a.so_i = so_i if so_i is not None else taps * K
    a.so_tap = so_tap if so_tap is not None else K
    a.so_outer, a.inner = so_outer, inner
    a.dtype, a.use_tr = _dt(A), use_tr
    a.colsum = _p(colsum)
    a.partial = _p(partial)
    a.partial_elems = partial.numel() if partial is not None else 0
    a.max_blocks = max_blocks          # 0: the library's default (96); 256: one block per CU, the fastest form of a lone launch
    if out2 is not None:      # columns >= split_c go to a second tensor (two weights that share dY)
        a.out2, a.split_c, a.so_i2, a.colsum2 = _p(out2), split_c, so_i2, _p(colsum2)
    L.check(lib.dist_op_gemm_tn(C.byref(a), _stream()))


def layernorm(x, w, b, *, y=None, y2=None, w2=None, b2=None, addend=None, period=0, mean=None, rstd=None, eps=1e-5):
    lib = L.load()
    rows, Cc = x.numel() // x.shape[-1], x.shape[-1]
    stats_only = y is False         # y=False: statistics only (mean / rstd), no normalised output
    y = None if stats_only else (torch.empty_like(x) if y is None else y)
    a = L.LnArgs()
    a.x, a.y, a.y2 = _p(x), _p(y), _p(y2)
    a.w, a.b, a.w2, a.b2 = _p(w), _p(b), _p(w2), _p(b2)
    a.addend, a.addend_period = _p(addend), period
    a.mean, a.rstd = _p(mean), _p(rstd)
    a.rows, a.C, a.dtype, a.eps = rows, Cc, _dt(x), eps
    L.check(lib.dist_op_layernorm(C.byref(a), _stream()))
    return y


def ln_stats_from_partials(part, C_, eps=1e-5):
    """fp32 [2*M] = mean then rstd from the [C/64][M][2] partials an EPI_ROWSTATS GEMM left (dist_op_ln_stats_from_partials)."""
    slices, M = part.shape[0], part.shape[1]
    stats = torch.empty(2 * M, dtype=torch.float32, device=part.device)
    L.check(L.load().dist_op_ln_stats_from_partials(_p(part), slices, M, C_, eps, _p(stats[:M]), _p(stats[M:]), _stream()))
    return stats


def layernorm_bwd(x, mean, rstd, dy, w, *, dy2=None, w2=None, dx=None, accumulate=False, dw=None, db=None, dw2=None, db2=None,
                  dx_add=None, dx_copy=None, two_phase=False):
    lib = L.load()
    rows, Cc = x.numel() // x.shape[-1], x.shape[-1]
    a = L.LnBwdArgs()
    a.x, a.mean, a.rstd = _p(x), _p(mean), _p(rstd)
    a.dy, a.w, a.dy2, a.w2 = _p(dy), _p(w), _p(dy2), _p(w2)
    a.dx, a.accumulate_dx = _p(dx), int(accumulate)
    a.dw, a.db, a.dw2, a.db2 = _p(dw), _p(db), _p(dw2), _p(db2)
    a.rows, a.C, a.dtype = rows, Cc, _dt(x)
    a.dx_add, a.dx_copy = _p(dx_add), _p(dx_copy)
    if two_phase:                                   # parameter gradients through per-block partial sums + a fixed-order reduction (no atomics)
        n = lib.dist_op_layernorm_bwd_scratch(rows, Cc)
        scratch = torch.empty(n, dtype=torch.float32, device=x.device)
        a.partial, a.partial_elems = _p(scratch), n
    L.check(lib.dist_op_layernorm_bwd(C.byref(a), _stream()))


def attention(qkv, frames, Ltok, heads, layout=L.QKV_ROWS):
    """layout QKV_ROWS: qkv is [frames*L, 3d]; QKV_HEADS: qkv holds [frame][head][q|k|v][L][64] (any shape, same size)."""
    lib = L.load()
    out = torch.empty(frames * Ltok, heads * 64, dtype=qkv.dtype, device=qkv.device)
    L.check(lib.dist_op_attention(_p(qkv), _p(out), frames, Ltok, heads, layout, _dt(qkv), _stream()))
    return out


def xattn1q(q, kv, B, S, Cc):
    lib = L.load()
    o = torch.empty(B, Cc, dtype=q.dtype, device=q.device)
    probs = torch.empty(B, Cc // 64, S, dtype=torch.float32, device=q.device)
    L.check(lib.dist_op_xattn1q(_p(q), _p(kv), _p(o), _p(probs), B, S, Cc, _dt(q), _stream()))
    return o, probs


def xattn1q_bwd(q, kv, probs, d_o, B, S, Cc):
    lib = L.load()
    dq = torch.empty_like(q)
    dkv = torch.empty_like(kv)
    L.check(lib.dist_op_xattn1q_bwd(_p(q), _p(kv), _p(probs), _p(d_o), _p(dq), _p(dkv), B, S, Cc, _dt(q), _stream()))
    return dq, dkv


def patchify(video, P, dtype):
    lib = L.load()
    b, _, T, H, W = video.shape
    Kp = (3 * P * P + 7) // 8 * 8
    out = torch.empty(b * T * (H // P) * (W // P), Kp, dtype=dtype, device=video.device)
    L.check(lib.dist_op_patchify(_p(video), _p(out), b, T, H, W, P, _dt(out), _stream()))
    return out


def add(a, b):
    lib = L.load()
    out = torch.empty_like(a)
    L.check(lib.dist_op_add(_p(a), _p(b), _p(out), a.numel(), _dt(a), _stream()))
    return out


def gelu_bwd(dy, pre):
    lib = L.load()
    out = torch.empty_like(dy)
    L.check(lib.dist_op_gelu_bwd(_p(dy), _p(pre), _p(out), dy.numel(), _dt(dy), _stream()))
    return out


def colsum(x, out, rows, Cc, ld=None, rmap=None):
    lib = L.load()
    L.check(lib.dist_op_colsum(_p(x), _p(out), rows, Cc, ld or x.shape[-1], rmap or rowmap(), _dt(x), _stream()))


def logits_loss(v, text, logit_scale, soft_target=None, dlogits_in=None, want_grad=True):
    lib = L.load()
    b, E = v.shape
    K = text.shape[0]
    dev = v.device
    logits = torch.empty(b, K, dtype=torch.float32, device=dev)
    vid = torch.empty(b, E, dtype=torch.float32, device=dev)
    loss = torch.zeros((), dtype=torch.float32, device=dev)
    dls = torch.zeros((), dtype=torch.float32, device=dev)
    dv = torch.empty_like(v) if want_grad else None
    L.check(lib.dist_op_logits_loss(_p(v), _p(text), _p(logit_scale), _p(soft_target), _p(logits), _p(vid), _p(loss), _p(dv), _p(dls),
                                    _p(dlogits_in), b, E, K, _dt(v), _stream()))
    return logits, vid, loss, dv, dls


def adamw(param, grad, m, v, segs, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """segs: device uint8 tensor holding an array of dist_adamw_seg."""
    lib = L.load()
    nseg = segs.numel() // C.sizeof(L.AdamwSeg)
    L.check(lib.dist_op_adamw(_p(param), _p(grad), _p(m), _p(v), _p(segs), nseg, param.numel(), beta1, beta2, eps, step, grad_scale, _stream()))


def make_segs(entries, device):
    """entries: list of (begin, end, lr, wd) -> device byte tensor."""
    arr = (L.AdamwSeg * len(entries))(*[L.AdamwSeg(int(b), int(e), float(lr), float(wd)) for b, e, lr, wd in entries])
    buf = bytes(arr)
    return torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(device)


# ---- batch-mode Mixup / CutMix (reference dataset/utils/mixup.py:18-23,212-223) --------------------------------------
def _f32(v):
    """the fp32 value torch uses for a Python-float scalar operand of an fp32 tensor op"""
    return float(np.float32(v))


def mixup_(video, lam):
    """in place: x[i] = x[i]*lam + x[b-1-i]*(1-lam) over fp32 clips [b, ...] (Mixup._mix_batch, mixup branch)."""
    assert video.is_cuda and video.dtype == torch.float32 and video.is_contiguous()
    b = video.shape[0]
    L.check(L.load().dist_op_mixup(_p(video), b, video.numel() // b, _f32(lam), _f32(1.0 - lam), _stream()))
    return video


def cutmix_(video, yl, yh, xl, xh):
    """in place: the box [yl,yh) x [xl,xh) of every plane is swapped between clip i and clip b-1-i (cutmix branch)."""
    assert video.is_cuda and video.dtype == torch.float32 and video.is_contiguous() and video.dim() >= 3
    b, H, W = video.shape[0], video.shape[-2], video.shape[-1]
    L.check(L.load().dist_op_cutmix(_p(video), b, video.numel() // (b * H * W), H, W, int(yl), int(yh), int(xl), int(xh), _stream()))
    return video


MIX_KINDS = {"none": 0, "mixup": 1, "cutmix": 2}


def patchify_mixed(video, P, dtype, kind="none", lam=1.0, box=None):
    """patch rows of the batch AS IF mixed by mixup_ / cutmix_ first (dist_op_patchify_mixed): `video` is only read."""
    assert video.is_cuda and video.dtype == torch.float32 and video.is_contiguous()
    b, _, T, H, W = video.shape
    Kp = (3 * P * P + 7) // 8 * 8
    out = torch.empty(b * T * (H // P) * (W // P), Kp, dtype=dtype, device=video.device)
    yl, yh, xl, xh = box if box is not None else (0, 0, 0, 0)
    L.check(L.load().dist_op_patchify_mixed(_p(video), _p(out), b, T, H, W, P, _dt(out), MIX_KINDS[kind], _f32(lam), _f32(1.0 - lam),
                                            int(yl), int(yh), int(xl), int(xh), _stream()))
    return out


def mixup_target(labels, num_classes, lam=1.0, smoothing=0.0):
    """soft target [b, K] fp32 = y1*lam + y2*(1-lam) with smoothed one-hot rows (reference mixup_target)."""
    assert labels.is_cuda
    labels = labels.long().contiguous().view(-1)
    off = smoothing / num_classes
    on = 1.0 - smoothing + off
    soft = torch.empty(labels.numel(), num_classes, dtype=torch.float32, device=labels.device)
    L.check(L.load().dist_op_mixup_target(_p(labels), labels.numel(), num_classes, _f32(lam), _f32(1.0 - lam), _f32(on), _f32(off), _p(soft), _stream()))
    return soft


# ---- evaluation side (reference base_blocks.py:573-585, utils/metrics.py:100-129, utils/meters.py:82-112) ------------------
ENSEMBLE_SUM, ENSEMBLE_MAX = 0, 1


def softmax_rows(x, out=None):
    """fp32 softmax over the last axis of logits [rows, K] (the head's eval activation)."""
    assert x.is_cuda and x.dim() == 2
    x = x.float().contiguous()
    y = torch.empty_like(x) if out is None else out
    L.check(L.load().dist_op_softmax_rows(_p(x), x.shape[0], x.shape[1], _p(y), _stream()))
    return y


def topk_correct(preds, labels, ks):
    """fp32 tensor [len(ks)]: number of rows whose label is within the ks[i] best scores (reference topks_correct)."""
    assert preds.is_cuda and labels.is_cuda and preds.dim() == 2 and 0 < len(ks) <= 4
    assert preds.size(0) == labels.size(0), "Batch dim of predictions and labels must match"
    preds = preds.float().contiguous()
    labels = labels.long().contiguous().view(-1)
    out = torch.empty(len(ks), dtype=torch.float32, device=preds.device)
    kk = (C.c_int * len(ks))(*[int(k) for k in ks])
    L.check(L.load().dist_op_topk_correct(_p(preds), _p(labels), preds.shape[0], preds.shape[1], kk, len(ks), _p(out), _stream()))
    return out


def ensemble_update(video_preds, video_labels, clip_count, preds, labels, clip_ids, num_clips, method, err):
    """TestMeter.update_stats on device state (fp32 [V,K], int64 [V], int64 [V]); err: int32 [1] flag word."""
    assert all(t.is_cuda for t in (video_preds, video_labels, clip_count, preds, labels, clip_ids, err))
    assert video_preds.dtype == torch.float32 and video_labels.dtype == torch.int64 and clip_count.dtype == torch.int64 and err.dtype == torch.int32
    preds = preds.float().contiguous()
    labels = labels.long().contiguous().view(-1)
    clip_ids = clip_ids.long().contiguous().view(-1)
    assert preds.shape[0] == labels.numel() == clip_ids.numel() and preds.shape[1] == video_preds.shape[1]
    L.check(L.load().dist_op_ensemble_update(_p(video_preds), _p(video_labels), _p(clip_count), _p(preds), _p(labels), _p(clip_ids),
                                             preds.shape[0], preds.shape[1], video_preds.shape[0], int(num_clips), int(method), _p(err), _stream()))


# ---- fp8 operands of the frozen spatial branch (BASELINE config 5) -----------------------------------------------------------
def quant_rows_fp8(x, q=None, scale=None):
    """per-row e4m3 quantisation of x [rows, K] (bf16 / fp32): returns (q uint8 [rows, K] holding OCP e4m3 bytes, scale fp32 [rows])."""
    assert x.is_cuda and x.dim() == 2 and x.is_contiguous()
    rows, K = x.shape
    q = torch.empty(rows, K, dtype=torch.uint8, device=x.device) if q is None else q
    scale = torch.empty(rows, dtype=torch.float32, device=x.device) if scale is None else scale
    L.check(L.load().dist_op_quant_rows_fp8(_p(x), _dt(x), rows, K, x.stride(0), _p(q), q.stride(0), _p(scale), _stream()))
    return q, scale


def fp8_scale_update(amax, scale, margin=4.0):
    """scale[i] = smallest power of two >= amax[i] * margin / 448; amax[i] = 0 (per-tensor scales for EPI_OUT8 producers)."""
    assert amax.is_cuda and scale.is_cuda and amax.dtype == scale.dtype == torch.float32 and amax.numel() == scale.numel()
    L.check(L.load().dist_op_fp8_scale_update(_p(amax), _p(scale), amax.numel(), float(margin), _stream()))


def amax_(x, amax):
    """amax[0] = max(amax[0], max |x|) (calibration of a tensor no EPI_OUT8 producer has written yet)."""
    assert x.is_cuda and x.is_contiguous() and amax.dtype == torch.float32
    L.check(L.load().dist_op_amax(_p(x), _dt(x), x.numel(), _p(amax), _stream()))


def attention_out8(qkv, frames, Ltok, heads, scale, amax=None, layout=L.QKV_HEADS):
    """ViT attention with the output as e4m3 bytes [frames*Ltok, heads*64] (per-tensor scale fp32 [1]; amax fp32 [1] collects max |o|)."""
    assert qkv.is_cuda and qkv.dtype == torch.bfloat16 and scale.dtype == torch.float32
    out8 = torch.empty(frames * Ltok, heads * 64, dtype=torch.uint8, device=qkv.device)
    L.check(L.load().dist_op_attention_out8(_p(qkv), _p(out8), _p(scale), _p(amax), frames, Ltok, heads, layout, _stream()))
    return out8


def attention_fp8(qkv8, in_scale, frames, Ltok, heads, out8_scale=None, amax=None):
    """ViT attention on the head-major e4m3 q | k | v image (uint8, one scale fp32 [1]); returns bf16 rows, or e4m3 bytes when out8_scale is given."""
    assert qkv8.is_cuda and qkv8.dtype == torch.uint8 and in_scale.dtype == torch.float32
    if out8_scale is None:
        out = torch.empty(frames * Ltok, heads * 64, dtype=torch.bfloat16, device=qkv8.device)
        L.check(L.load().dist_op_attention_fp8(_p(qkv8), _p(in_scale), _p(out), None, None, None, frames, Ltok, heads, _stream()))
        return out
    out8 = torch.empty(frames * Ltok, heads * 64, dtype=torch.uint8, device=qkv8.device)
    L.check(L.load().dist_op_attention_fp8(_p(qkv8), _p(in_scale), None, _p(out8), _p(out8_scale), _p(amax), frames, Ltok, heads, _stream()))
    return out8
