"""The fused IntegrationNetwork launches at the bench geometry (32 clips x 8 frames x 197 tokens, Ci = 384) and the unfused sequences they replace, on
cold operands.  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes): bytes per launch of each kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from tests.test_integ_gpu import make, CI, C4
clips, t, Ltok = 32, 8, 197
rows = clips * t * Ltok
w, _ = make(1, 8, 16, seed=1)
wc = {k: v.cuda() for k, v in w.items()}
Wt = torch.randn(CI, C4, 2, 1, 1, device="cuda") * 0.07
bt2 = torch.randn(CI, device="cuda") * 0.1; cls = torch.randn(t, CI, device="cuda") * 0.5
pk = ops.integration_pack(wc, bwd=True, t2i_w=Wt)
bf = lambda v: v.to(torch.bfloat16).contiguous()
def cold():
    big = torch.empty(96 << 20, device="cuda"); big.zero_(); del big                      # push the operands out of the Infinity Cache
for rep in range(3):
    x = (torch.randn(rows, CI, device="cuda") * 1.5 + 0.3).to(torch.bfloat16)
    dR = (torch.randn(rows, CI, device="cuda") * 0.5).to(torch.bfloat16)
    cold()
    saved = ops.integration_fwd(x, pk, clips, t, Ltok, xhat=True)
    cold()
    ops.integration_fwd(x, pk, clips, t, Ltok, train=False)
    cold()
    Xp = (torch.randn(clips * 2 * t * (Ltok - 1), C4, device="cuda") * 0.8).to(torch.bfloat16)
    ops.integration_fwd(x, pk, clips, t, Ltok, xhat=True, t2i=(Xp, bt2, cls))        # T2I in front (what the engine launches; M' is also written here)
    cold()
    ops.integration_bwd(dR, saved, pk, clips, t, Ltok, copy=True)
    cold()
    # the unfused forward sequence (engine.hip, DIST_AMD_INTEG_FUSED=0)
    Na, Nb = torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    ops.layernorm(x, wc["ln.weight"], wc["ln.bias"], y=Na, y2=Nb, w2=wc["ln_temporal.weight"], b2=wc["ln_temporal.bias"], mean=mean, rstd=rstd)
    zf = torch.empty(rows, CI + C4, dtype=torch.bfloat16, device="cuda"); hf = torch.empty_like(zf)
    h1 = torch.empty(rows, C4, dtype=torch.bfloat16, device="cuda"); R = torch.empty_like(x)
    ops.gemm_nt(Na, bf(wc["ffn.c_fc.weight"]), rows, CI, CI, bias=wc["ffn.c_fc.bias"], C_out=zf, C2_out=hf)
    ops.gemm_nt(Nb, bf(wc["temporal_ffn.c_fc1.weight"].reshape(C4, CI)), rows, C4, CI, bias=wc["temporal_ffn.c_fc1.bias"], C_out=h1)
    W2 = bf(wc["temporal_ffn.c_fc2.weight"].reshape(C4, C4, 3).permute(0, 2, 1).reshape(C4, 3 * C4))
    ops.gemm_nt(h1, W2, rows, C4, C4, taps=3, bias=wc["temporal_ffn.c_fc2.bias"], amap=ops.rowmap(L.RM_SHIFT, t * Ltok, Ltok, 1),
                C_out=zf[:, CI:], C2_out=hf[:, CI:], ldc=CI + C4, ldc2=CI + C4)
    Wp = bf(torch.cat([wc["ffn.c_proj.weight"], wc["temporal_ffn.c_proj.weight"].reshape(CI, C4)], dim=1))
    ops.gemm_nt(hf, Wp, rows, CI, CI + C4, bias=wc["ffn.c_proj.bias"] + wc["temporal_ffn.c_proj.bias"], C_out=R)
    cold()
    # the unfused data-gradient sequence
    dz = torch.empty_like(zf); dh1 = torch.empty_like(h1); dNa = torch.empty_like(x); dNb = torch.empty_like(x); dMp = torch.empty_like(x); dM = torch.empty_like(x)
    ops.gemm_nt(dR, bf(Wp.t()), rows, CI + C4, CI, aux=zf, C_out=dz)
    W2b = bf(wc["temporal_ffn.c_fc2.weight"].reshape(C4, C4, 3).permute(1, 2, 0).reshape(C4, 3 * C4))
    ops.gemm_nt(dz[:, CI:], W2b, rows, C4, C4, taps=3, amap=ops.rowmap(L.RM_SHIFT, t * Ltok, Ltok, -1), C_out=dh1, lda=CI + C4)
    ops.gemm_nt(dz, bf(wc["ffn.c_fc.weight"].t()), rows, CI, CI, C_out=dNa, lda=CI + C4)
    ops.gemm_nt(dh1, bf(wc["temporal_ffn.c_fc1.weight"].reshape(C4, CI).t()), rows, CI, C4, C_out=dNb)
    dga, dba, dgb, dbb = (torch.zeros(CI, device="cuda") for _ in range(4))
    ops.layernorm_bwd(x, mean, rstd, dNa, wc["ln.weight"], dy2=dNb, w2=wc["ln_temporal.weight"], dx=dMp, dw=dga, db=dba, dw2=dgb, db2=dbb, dx_copy=dM)
    torch.cuda.synchronize()
print("done")
