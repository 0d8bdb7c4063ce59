"""CPU oracle for the DiST training hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product (dist_amd/) never does and fails loudly without its HIP
library.

This is a plain-torch (CPU, fp32 or fp64) restatement of the reference's
algorithm in the token layout the HIP kernels use:

    spatial / integration tokens  S[b, j, l, c]   j in [0,t) ViT frames, l=0 is cls
    temporal map                  X[b, k, n, c]   k in [0,T) frames, n row-major patch

The reference keeps S as [L, b*t, C] and X as [b, Ct, T, H, W]
(models/base/clip.py:284, models/module_zoo/branches/dist.py:43,83-85,103); the
arithmetic is the same.  Gradients come from autograd over this restatement, as
the reference's come from autograd over its own graph (runs/train.py:110).

PINNING: the reference ships no tests.  This oracle is pinned
against OUTPUTS OF THE REFERENCE ITSELF, generated in the build container by
oracle/make_golden.py (which imports /root/reference) and committed under
tests/golden/; tests/test_oracle_golden.py checks it on every run.

`rnd` hooks: when `bf16=True` every tensor that the HIP path stores in bf16
(kernel boundaries) is rounded to bf16 here too, so the bf16 kernels can be
checked against an oracle with identical rounding points.

`fused=True` (with bf16=True) moves the rounding points to where the FUSED kernels of
rounds 2-3 have them (the unfused launch sequences keep the `fused=False` points):
  * LayerNorm folded into the consumer GEMM (frozen ViT ln_1 -> in_proj, ln_2 -> c_fc;
    csrc/engine.hip gemm_lnfold): the GEMM multiplies the RAW bf16 rows with
    bf16(W diag(gamma)) and normalises behind it - no rounded LayerNorm output;
  * IntegrationNetwork (csrc/integ.hip:12-34): xhat = (M' - mean) rstd is the one rounded
    LayerNorm tensor (statistics of the UNROUNDED M' that the T2I stage leaves on the
    accumulators), both first Linears use bf16(W diag(gamma)) and b + W beta, the two
    c_proj products are ONE accumulation (r1 is not rounded);
  * gradients are rounded to bf16 where the backward kernels store them in bf16
    (dM, dM', [dzf | dh1 | dh2], dY, dp, dz, dX): `gr` hooks.
"""
import math

import torch
import torch.nn.functional as F


def qgelu(x):
    """QuickGELU, reference models/base/clip.py:199-201."""
    return x * torch.sigmoid(1.702 * x)


def layer_norm(x, w, b, eps=1e-5):
    """LayerNorm over the channel axis, fp32-internal (reference clip.py:181-187)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


class Oracle:
    def __init__(self, g, params, dtype=torch.float32, bf16=False, vit_fp8=0, fused=False):
        """g: dist_amd.synth.Geometry; params: name -> tensor (reference state-dict names).
        vit_fp8 (with bf16=True): bit mask of the frozen-ViT GEMMs evaluated on per-row e4m3 operands the way dist_config.vit_fp8 does
        (1 in_proj, 2 out_proj, 4 c_fc, 8 c_proj; oracle/fp8_oracle.py - no counterpart in the reference, parity unpinned)."""
        self.g = g
        self.dtype = dtype
        self.bf16 = bf16
        self.vit_fp8 = vit_fp8 if bf16 else 0
        self.fused = bool(fused and bf16)
        self.p = {k: torch.as_tensor(v).to(dtype) for k, v in params.items()}
        self.selected = list(getattr(g, "selected", None) or range(g.layers))       # DIST.SELECTED_LAYERS (reference dist.py:170-190,226)

    # ---- rounding points ---------------------------------------------------------------
    def rnd(self, x):
        if not self.bf16:
            return x
        return x.to(torch.bfloat16).to(self.dtype) if not x.requires_grad else _RoundBF16.apply(x)

    def gr(self, x):
        """identity whose GRADIENT is rounded to bf16 (fused mode: a gradient tensor the backward kernels store in bf16)"""
        return _RoundGradBF16.apply(x) if (self.fused and x.requires_grad) else x

    def ln_fold_lin(self, x, W, bias, gamma, beta):
        """LayerNorm folded into the consumer GEMM (csrc/engine.hip gemm_lnfold, dist_op_ln_fold): raw rows x bf16(W diag(gamma)), normalised
        behind the product: rstd * (x W'^T - mean * rowsum(W')) + (bias + W beta)"""
        mean = x.mean(dim=-1, keepdim=True)
        rstd = (x.var(dim=-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        Wf = self.rnd(W * gamma[None, :])
        return rstd * (x @ Wf.t() - mean * Wf.sum(dim=1)) + (bias + W @ beta)

    def w(self, name):
        """weight as the kernels see it (bf16-rounded working copy in bf16 mode)."""
        t = self.p[name]
        return self.rnd(t) if (self.bf16 and t.dim() >= 2) else t

    # ---- patches ----------------------------------------------------------------------
    def patchify(self, video):
        """[b,3,T,H,W] -> [b,T,N,3*P*P], patch vector ordered (c,py,px) like a flattened
        conv kernel (reference clip.py:232,271 Conv2d stride=patch; dist.py:178-181)."""
        g = self.g
        b = video.shape[0]
        P, G = g.patch, g.grid
        x = video.to(self.dtype).permute(0, 2, 1, 3, 4)            # b,T,3,H,W
        x = x.reshape(b, g.T, 3, G, P, G, P).permute(0, 1, 3, 5, 2, 4, 6)
        return self.rnd(x.reshape(b, g.T, g.N, 3 * P * P))

    # ---- frozen ViT (reference clip.py:263-300, 150-178) -------------------------------
    def vit(self, patches):
        g, p = self.g, self.p
        b = patches.shape[0]
        with torch.no_grad():
            Wc = self.w("visual.conv1.weight").reshape(g.d, -1)
            pe = self.rnd(patches[:, ::g.alpha] @ Wc.t())            # only frames k = alpha*j continue (clip.py:284)
            cls = self.rnd(p["visual.class_embedding"]).expand(b, g.t, 1, g.d)
            x = torch.cat([cls, pe], dim=2) + p["visual.positional_embedding"]
            x = self.rnd(layer_norm(x, p["visual.ln_pre.weight"], p["visual.ln_pre.bias"]))
            feats = []
            for i in range(g.layers):
                x = self.vit_block(x, i)
                feats.append(x)
        return feats

    def img_logits(self, feats):
        """`img_logits` of the reference's output dictionary: ln_post(cls token of the last block) @ visual.proj, [b*t, E], not normalised
        (reference clip.py:291-298 `cls_x`, returned as `img_cls_tokens_ori` at :503,532)."""
        p = self.p
        with torch.no_grad():
            cls = feats[-1][:, :, 0].reshape(-1, self.g.d)
            y = self.rnd(layer_norm(cls, p["visual.ln_post.weight"], p["visual.ln_post.bias"]))
            return self.rnd(y @ self.w("visual.proj"))

    # ---- e4m3 operands (dist_config.vit_fp8) ---------------------------------------------
    @staticmethod
    def _fp8_rows(a2, per_tensor):
        """e4m3 image of a [rows, K]: per-row amax / 448 scales, or (the producers' DIST_EPI_OUT8 images, vit_fp8 bit 16) ONE power-of-two
        scale for the tensor, the smallest >= amax * 4 / 448 (dist_op_fp8_scale_update), values clamped to +-448"""
        import fp8_oracle as fo
        if not per_tensor:
            return fo.quant_rows(a2)
        t = max(float(a2.abs().max()), 2.0 ** -24) * 4.0 / 448.0
        sc = 2.0 ** int(math.ceil(math.log2(t)))
        q = (a2.float() / sc).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        return q, torch.full((a2.shape[0],), sc, dtype=torch.float32)

    def _fp8_lin(self, a, W, bias, per_tensor=False):
        """a [..., K] (bf16-valued) x W [N, K] (bf16 working copy) on e4m3 operands: dequantised product + bias"""
        import fp8_oracle as fo
        qa, sa = self._fp8_rows(a.reshape(-1, a.shape[-1]), per_tensor)
        qw, sw = fo.quant_rows(W)
        return (fo.gemm(qa, sa, qw, sw).to(self.dtype) + bias).reshape(*a.shape[:-1], W.shape[0])

    def _fp8_ln_lin(self, x, W, bias, gamma, beta, per_tensor=False):
        """LayerNorm folded into the GEMM as the engine does it: raw rows and W diag(gamma) (bf16) in e4m3, the statistics applied behind:
        rstd * (deq(xq) deq(Wq)^T - mean * colsum(deq(Wq))) + (bias + W beta)"""
        import fp8_oracle as fo
        xr = x.reshape(-1, x.shape[-1])
        mean = xr.mean(dim=1, keepdim=True)
        rstd = (xr.var(dim=1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        Wf = self.rnd(W * gamma[None, :])
        qa, sa = self._fp8_rows(xr, per_tensor)
        qw, sw = fo.quant_rows(Wf)
        colsum = fo.dequant(qw, sw).sum(dim=1).to(self.dtype)
        y = rstd * (fo.gemm(qa, sa, qw, sw).to(self.dtype) - mean * colsum[None, :]) + (bias + W @ beta)
        return y.reshape(*x.shape[:-1], W.shape[0])

    def vit_block(self, x, i):
        g, p = self.g, self.p
        pre = f"visual.transformer.resblocks.{i}."
        b, t, L, d = x.shape
        f8 = self.vit_fp8
        img = bool(f8 & 16)                                  # producers' per-tensor images of every GEMM input
        if f8 & 1:
            qkv = self.rnd(self._fp8_ln_lin(x, p[pre + "attn.in_proj_weight"], p[pre + "attn.in_proj_bias"], p[pre + "ln_1.weight"], p[pre + "ln_1.bias"],
                                            per_tensor=img and i > 0))
        elif self.fused:
            qkv = self.rnd(self.ln_fold_lin(x, p[pre + "attn.in_proj_weight"], p[pre + "attn.in_proj_bias"], p[pre + "ln_1.weight"], p[pre + "ln_1.bias"]))
        else:
            h = self.rnd(layer_norm(x, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"]))
            qkv = self.rnd(h @ self.w(pre + "attn.in_proj_weight").t() + p[pre + "attn.in_proj_bias"])
        if f8 & 1 and img:                                   # image mode: q | k | v exist as e4m3 only (one power-of-two scale for the tensor)
            q8, s8 = self._fp8_rows(qkv.reshape(-1, qkv.shape[-1]), True)
            qkv = (q8.view(torch.float8_e4m3fn).to(self.dtype) * s8[:, None].to(self.dtype)).reshape(qkv.shape)
        q, k, v = qkv.reshape(b * t, L, 3, g.heads, 64).permute(2, 0, 3, 1, 4)   # [bt,h,L,64]
        att = torch.softmax((q @ k.transpose(-1, -2)) / 8.0, dim=-1)
        o = self.rnd((att @ v).permute(0, 2, 1, 3).reshape(b, t, L, d))
        if f8 & 2:
            x = self.rnd(x + self._fp8_lin(o, self.w(pre + "attn.out_proj.weight"), p[pre + "attn.out_proj.bias"], per_tensor=img))
        else:
            x = self.rnd(x + o @ self.w(pre + "attn.out_proj.weight").t() + p[pre + "attn.out_proj.bias"])
        if f8 & 4:
            h = self.rnd(qgelu(self._fp8_ln_lin(x, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"], p[pre + "ln_2.weight"], p[pre + "ln_2.bias"], per_tensor=img)))
        elif self.fused:
            h = self.rnd(qgelu(self.ln_fold_lin(x, p[pre + "mlp.c_fc.weight"], p[pre + "mlp.c_fc.bias"], p[pre + "ln_2.weight"], p[pre + "ln_2.bias"])))
        else:
            h = self.rnd(layer_norm(x, p[pre + "ln_2.weight"], p[pre + "ln_2.bias"]))
            h = self.rnd(qgelu(h @ self.w(pre + "mlp.c_fc.weight").t() + p[pre + "mlp.c_fc.bias"]))
        if f8 & 8:
            x = self.rnd(x + self._fp8_lin(h, self.w(pre + "mlp.c_proj.weight"), p[pre + "mlp.c_proj.bias"], per_tensor=img))
        else:
            x = self.rnd(x + h @ self.w(pre + "mlp.c_proj.weight").t() + p[pre + "mlp.c_proj.bias"])
        return x

    # ---- DiST branch (reference dist.py:222-247) ---------------------------------------
    def shift_t(self, x, delta):
        """x[b,k,...] -> x[b,k+delta,...] with zero fill outside [0,K) (Conv3d zero padding)."""
        if delta == 0:
            return x
        z = torch.zeros_like(x[:, :abs(delta)])
        return torch.cat([x[:, delta:], z], 1) if delta > 0 else torch.cat([z, x[:, :delta]], 1)

    def temporal_stem(self, patches):
        """Conv3d 3->Ct k=(tp,P,P) s=(1,P,P) pad=(tp//2,0,0) (dist.py:178-181,225)."""
        g = self.g
        W = self.w("dist_net.temporal_stem.weight")                  # [Ct,3,tp,P,P]
        out = self.p["dist_net.temporal_stem.bias"].expand(*patches.shape[:3], g.Ct)
        for dt in range(g.tpatch):
            Wd = W[:, :, dt].reshape(g.Ct, -1)
            out = out + self.shift_t(patches, dt - g.tpatch // 2) @ Wd.t()
        return self.rnd(out)

    def temporal_net(self, X, i, keep):
        """gelu(x + conv1x3x3(gelu(conv3x1x1(LN(x))))) (dist.py:48-65)."""
        g, p = self.g, self.p
        pre = f"dist_net.temporal_nets.{i}."
        b = X.shape[0]
        U = self.rnd(layer_norm(X, p[pre + "ln.weight"], p[pre + "ln.bias"]))
        keep[f"tn_U.{i}"] = U
        W1 = self.w(pre + "temporal_net.c_fc1.weight")               # [Co,Ct,tk,1,1]
        z = p[pre + "temporal_net.c_fc1.bias"].expand(*U.shape[:-1], W1.shape[0])     # hidden width = Ct * TEMPORAL_CONV_MLP_RATIO (dist.py:54)
        for dt in range(g.tk):
            z = z + self.shift_t(U, dt - g.tk // 2) @ W1[:, :, dt, 0, 0].t()
        zr = self.rnd(z)
        z = self.gr(zr)                                              # (dz is stored in bf16: tnet_bwd_spatial_kernel)
        V = self.rnd(qgelu(z))
        keep[f"tn_V.{i}"] = V
        W2 = self.w(pre + "temporal_net.c_fc2.weight")               # [Ct,Co,1,3,3]
        Vg = V.reshape(b, g.T, g.grid, g.grid, V.shape[-1])
        Vp = F.pad(Vg, (0, 0, 1, 1, 1, 1))
        acc = p[pre + "temporal_net.c_fc2.bias"].expand(b, g.T, g.grid, g.grid, g.Ct)
        for dy in range(3):
            for dx in range(3):
                acc = acc + Vp[:, :, dy:dy + g.grid, dx:dx + g.grid] @ W2[:, :, 0, dy, dx].t()
        pr = self.rnd(X + acc.reshape(X.shape))
        pre_act = self.gr(pr)                                        # (dp is stored in bf16: the fused T2I backward / the T2I data-gradient GEMM)
        Xp = self.rnd(qgelu(pre_act))
        keep[f"tn_p.{i}"] = pr
        keep[f"tn_z.{i}"] = zr
        keep[f"tn_out.{i}"] = Xp
        return Xp

    def integration_net(self, Mp, i, keep):
        """ffn(ln(x)) + temporal_ffn(ln_temporal(x)) (dist.py:16-45)."""
        g, p = self.g, self.p
        pre = f"dist_net.integration_nets.{i}."
        if self.fused:
            return self.integration_net_fused(Mp, i, keep)
        na = self.rnd(layer_norm(Mp, p[pre + "ln.weight"], p[pre + "ln.bias"]))
        nb = self.rnd(layer_norm(Mp, p[pre + "ln_temporal.weight"], p[pre + "ln_temporal.bias"]))
        zf = self.rnd(na @ self.w(pre + "ffn.c_fc.weight").t() + p[pre + "ffn.c_fc.bias"])
        hf = self.rnd(qgelu(zf))
        h1 = self.rnd(nb @ self.w(pre + "temporal_ffn.c_fc1.weight")[:, :, 0, 0, 0].t() + p[pre + "temporal_ffn.c_fc1.bias"])
        W2 = self.w(pre + "temporal_ffn.c_fc2.weight")               # [C4,C4,tk,1,1]
        h2 = p[pre + "temporal_ffn.c_fc2.bias"].expand_as(h1)
        for dt in range(g.tk):                                       # conv over the j (frame) axis per (b,l)
            h2 = h2 + self.shift_t(h1, dt - g.tk // 2) @ W2[:, :, dt, 0, 0].t()
        h2 = self.rnd(h2)
        g2 = self.rnd(qgelu(h2))
        r1 = self.rnd(hf @ self.w(pre + "ffn.c_proj.weight").t() + p[pre + "ffn.c_proj.bias"])
        R = self.rnd(r1 + g2 @ self.w(pre + "temporal_ffn.c_proj.weight")[:, :, 0, 0, 0].t() + p[pre + "temporal_ffn.c_proj.bias"])
        keep[f"int_out.{i}"] = R
        keep.update({f"int_na.{i}": na, f"int_nb.{i}": nb, f"int_zf.{i}": zf, f"int_hf.{i}": hf, f"int_h1.{i}": h1, f"int_h2.{i}": h2, f"int_g2.{i}": g2})
        return R

    def integration_net_fused(self, Mp, i, keep):
        """the same network with the rounding points of integ_fwd_kernel / integ_bwd_kernel (csrc/integ.hip:12-34): `Mp` is the UNROUNDED
        M' when the T2I stage runs in front of it (it stays on the accumulators)."""
        g, p = self.g, self.p
        pre = f"dist_net.integration_nets.{i}."
        mean = Mp.mean(dim=-1, keepdim=True)
        rstd = (Mp.var(dim=-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        xh = self.rnd((Mp - mean) * rstd)                            # the ONE LayerNorm tensor of the block
        Wa, Wb = p[pre + "ffn.c_fc.weight"], p[pre + "temporal_ffn.c_fc1.weight"][:, :, 0, 0, 0]
        Waf = self.rnd(Wa * p[pre + "ln.weight"][None, :])           # W diag(gamma) in bf16, re-derived from the fp32 masters every step
        Wbf = self.rnd(Wb * p[pre + "ln_temporal.weight"][None, :])
        zfr = self.rnd(xh @ Waf.t() + (p[pre + "ffn.c_fc.bias"] + Wa @ p[pre + "ln.bias"]))
        zf = self.gr(zfr)
        hf = self.rnd(qgelu(zf))
        h1r = self.rnd(xh @ Wbf.t() + (p[pre + "temporal_ffn.c_fc1.bias"] + Wb @ p[pre + "ln_temporal.bias"]))
        h1 = self.gr(h1r)
        W2 = self.w(pre + "temporal_ffn.c_fc2.weight")
        h2 = p[pre + "temporal_ffn.c_fc2.bias"].expand_as(h1)
        for dt in range(g.tk):
            h2 = h2 + self.shift_t(h1, dt - g.tk // 2) @ W2[:, :, dt, 0, 0].t()
        h2r = self.rnd(h2)
        h2 = self.gr(h2r)
        g2 = self.rnd(qgelu(h2))
        R = self.rnd(hf @ self.w(pre + "ffn.c_proj.weight").t() + p[pre + "ffn.c_proj.bias"]
                     + g2 @ self.w(pre + "temporal_ffn.c_proj.weight")[:, :, 0, 0, 0].t() + p[pre + "temporal_ffn.c_proj.bias"])
        keep[f"int_out.{i}"] = R
        keep.update({f"int_xhat.{i}": xh, f"int_zf.{i}": zfr, f"int_hf.{i}": hf, f"int_h1.{i}": h1r, f"int_h2.{i}": h2r, f"int_g2.{i}": g2})
        return R

    def mha_1q(self, pre, q_in, kv_in):
        """nn.MultiheadAttention with ONE query per batch element; the same ln_1 normalises
        q and k/v (reference clip.py:139-147).  q_in [B,C], kv_in [B,S,C] -> [B,C]."""
        g, p = self.g, self.p
        C, H = g.Ci, g.iheads
        Win, bin_ = self.w(pre + "attn.in_proj_weight"), p[pre + "attn.in_proj_bias"]
        qn = self.rnd(layer_norm(q_in, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"]))
        kn = self.rnd(layer_norm(kv_in, p[pre + "ln_1.weight"], p[pre + "ln_1.bias"]))
        q = self.rnd(qn @ Win[:C].t() + bin_[:C])
        kv = self.rnd(kn @ Win[C:].t() + bin_[C:])
        k, v = kv[..., :C], kv[..., C:]
        B, S = kv_in.shape[0], kv_in.shape[1]
        qh = q.reshape(B, H, 1, 64)
        kh = k.reshape(B, S, H, 64).permute(0, 2, 1, 3)
        vh = v.reshape(B, S, H, 64).permute(0, 2, 1, 3)
        att = torch.softmax((qh @ kh.transpose(-1, -2)) / 8.0, dim=-1)
        o = self.rnd((att @ vh).reshape(B, C))
        return o @ self.w(pre + "attn.out_proj.weight").t() + p[pre + "attn.out_proj.bias"]

    def mlp(self, pre, x):
        p = self.p
        h = self.rnd(qgelu(self.rnd(x @ self.w(pre + "c_fc.weight").t() + p[pre + "c_fc.bias"])))
        return h @ self.w(pre + "c_proj.weight").t() + p[pre + "c_proj.bias"]

    def branch(self, patches, feats, text_features):
        g, p = self.g, self.p
        b = patches.shape[0]
        keep = {}
        X = self.temporal_stem(patches)
        keep["stem"] = X
        R = None
        for idx, lid in enumerate(self.selected):
            Xp = self.temporal_net(X, idx, keep)                                           # dist.py:228
            M = feats[lid] @ self.w(f"dist_net.input_linears.{idx}.weight").t() + p[f"dist_net.input_linears.{idx}.bias"]
            if R is not None:
                M = M + R                                                                  # dist.py:229
            M = self.gr(self.rnd(M))                                                       # (dM = dM' + I2T term is stored in bf16)
            # I2T (dist.py:90-105): drop cls row, Linear Ci->Ct, nearest upsample x alpha in T
            pre = f"dist_net.integration2temporal_nets.{idx}.linear_fuse."
            Y = self.gr(M[:, :, 1:] @ self.w(pre + "weight").t() + p[pre + "bias"])         # (dY, the frame-pair sums of dX_next, is stored in bf16)
            X_next = self.gr(self.rnd(Xp + Y.repeat_interleave(g.alpha, dim=1)))           # dist.py:231 (dX of the next TemporalNet: bf16)
            # T2I (dist.py:68-86): Conv3d k=s=(alpha,1,1) + learnable per-frame cls row
            pre = f"dist_net.temporal2integration_nets.{idx}."
            Wt = self.w(pre + "linear_fuse.weight")                                        # [Ci,Ct,alpha,1,1]
            Xr = Xp.reshape(b, g.t, g.alpha, g.N, g.Ct)
            Q = p[pre + "linear_fuse.bias"].expand(b, g.t, g.N, g.Ci)
            for a in range(g.alpha):
                Q = Q + Xr[:, :, a] @ Wt[:, :, a, 0, 0].t()
            cls = p[pre + "cls_token"][0, 0].reshape(1, g.t, 1, g.Ci).expand(b, g.t, 1, g.Ci)
            Mp_raw = self.gr(M + torch.cat([cls, Q], dim=2))                               # dist.py:232 (dM' is stored in bf16)
            Mp = self.rnd(Mp_raw)
            # fused: T2I is formed in front of the fused IntegrationNetwork kernel where alpha = 2 and the temporal width equals Ci / 4
            # (engine.hip set_fused_flags: ig_t2i) - the LayerNorm then sees the unrounded M'
            t2i_fused = self.fused and g.alpha == 2 and g.Ct * 4 == g.Ci
            R = self.integration_net(Mp_raw if t2i_fused else Mp, idx, keep)               # dist.py:234
            X = X_next
            keep[f"x_temporal.{idx}"] = X
            keep[f"mid.{idx}"] = Mp
        Fz = self.rnd(R + Mp)                                                              # dist.py:239
        u = self.rnd(p["dist_net.aggregated_cls_token"].reshape(1, g.Ci)).expand(b, g.Ci)  # dist.py:237
        s = self.rnd(p["dist_net.aggregated_spatial_cls_token"].reshape(1, g.Ci)).expand(b * g.t, g.Ci)
        Fzf = Fz.reshape(b * g.t, g.L, g.Ci)
        for a in range(g.ada):                                                             # dist.py:139-162
            pre = f"dist_net.adapooling_nets.{a}."
            s = self.rnd(s + self.mha_1q(pre + "spatial_transformer.", s, Fzf))
            sn = self.rnd(layer_norm(s, p[pre + "ln_out_spat_cls_token.weight"], p[pre + "ln_out_spat_cls_token.bias"]))
            s = self.rnd(s + self.mlp(pre + "output_map_spatial_cls_token.", sn))
            c = self.rnd(s.reshape(b, g.t, g.Ci) + p[pre + "positional_embedding"])
            u = self.rnd(u + self.mha_1q(pre + "temporal_transformer.", u, c))
            un = self.rnd(layer_norm(u, p[pre + "ln_out_temp_cls_token.weight"], p[pre + "ln_out_temp_cls_token.bias"]))
            u = self.rnd(u + self.mlp(pre + "output_map_cls_token.", un))
        keep["top_cls"] = u
        mean_cls = self.rnd(feats[self.selected[-1]][:, :, 0].mean(dim=1))                 # dist.py:243
        zc = mean_cls @ self.w("dist_net.proj_spatial_cls_token.weight").t() + p["dist_net.proj_spatial_cls_token.bias"]
        z = self.rnd(layer_norm(self.rnd(u + zc), p["dist_net.ln_post.weight"], p["dist_net.ln_post.bias"]))
        v = self.rnd(z @ self.w("dist_net.proj"))                                          # dist.py:246
        # cosine logits (clip.py:509-518)
        vn = v / v.norm(dim=1, keepdim=True)
        tn = text_features / text_features.norm(dim=1, keepdim=True)
        logits = p["logit_scale"].exp() * vn @ tn.t()
        return logits, vn, keep

    def forward(self, video, text_features):
        """-> dict(logits [b,K] (= preds in train mode, base_blocks.py:579-585), vid_logits, keep)."""
        patches = self.patchify(torch.as_tensor(video))
        feats = self.vit(patches)
        logits, vn, keep = self.branch(patches, feats, torch.as_tensor(text_features).to(self.dtype))
        return {"logits": logits, "vid_logits": vn, "feats": feats, "keep": keep, "img_logits": self.img_logits(feats)}

    def loss(self, logits, soft_target):
        """SoftTargetCrossEntropy (reference models/utils/losses.py:29-31)."""
        return torch.sum(-torch.as_tensor(soft_target).to(logits.dtype) * F.log_softmax(logits, dim=-1), dim=-1).mean()

    def forward_backward(self, video, text_features, soft_target):
        """One reference train step up to the gradients (runs/train.py:101-110)."""
        for k, v in self.p.items():
            if k.startswith("dist_net.") or k == "logit_scale":
                v.requires_grad_(True)
                v.grad = None
        out = self.forward(video, text_features)
        loss = self.loss(out["logits"], soft_target)
        loss.backward()
        grads = {k: v.grad for k, v in self.p.items() if v.grad is not None}
        out["loss"] = loss.detach()
        out["grads"] = grads
        return out


class _RoundGradBF16(torch.autograd.Function):
    """identity in forward, bf16 rounding of the gradient in backward"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, gy):
        return gy.to(torch.bfloat16).to(gy.dtype)


class _RoundBF16(torch.autograd.Function):
    """Round to bf16 in forward, straight-through in backward (the kernels' dX is
    stored bf16 as well; gradient rounding is applied by the caller where it matters)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, gy):
        return gy


# --------------------------------------------------------------------------------------
# optimizer restatement (reference models/utils/optimizer.py:138-186 as INTENDED)
# --------------------------------------------------------------------------------------

def dist_param_groups(shapes):
    """The five DiST parameter groups the released constructor evidently intends
    (reference models/utils/optimizer.py:138-186; broken as shipped: misplaced brackets,
    SURVEY.md §0).  shapes: name -> shape.  Returns name -> group id:
    0 cls_token/positional_embedding (wd 0), 1 adapooling weights (wd NEW_NET_WEIGHT_DECAY),
    2 adapooling bias/1-D (wd 0), 3 other weights (wd NEW_NET_WEIGHT_DECAY), 4 other bias/1-D (wd 0)."""
    out = {}
    for n, shp in shapes.items():
        if "dist_net" not in n:
            continue
        one_d = ("bias" in n) or len(shp) == 1
        if n.endswith("cls_token") or n.endswith("positional_embedding"):
            out[n] = 0
        elif "adapooling_nets" in n:
            out[n] = 2 if one_d else 1
        else:
            out[n] = 4 if one_d else 3
    return out


def adamw_step(param, grad, m, v, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.AdamW single-tensor math (reference optimizer.py:67-73 uses torch's)."""
    param = param * (1.0 - lr * wd)
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)) + eps
    param = param - (lr / bc1) * m / denom
    return param, m, v


def lr_at(cur_epoch, base_lr, max_epoch, warmup_epochs, warmup_start_lr):
    """Cosine schedule with linear warm-up (reference models/utils/lr_policy.py:10-44)."""
    cos = lambda e: base_lr * (math.cos(math.pi * e / max_epoch) + 1.0) * 0.5
    lr = cos(cur_epoch)
    if cur_epoch < warmup_epochs:
        alpha = (cos(warmup_epochs) - warmup_start_lr) / warmup_epochs
        lr = cur_epoch * alpha + warmup_start_lr
    return lr
