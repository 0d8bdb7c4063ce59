"""The fused TemporalNet launches at the bench geometry (b=32, T=16, 14x14, Ct=96) and the unfused sequences they replace, 2 launches each on
rotating cold operands.  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes): bytes per launch of each kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
dt = torch.bfloat16
clips, T, G, Ct = 32, 16, 14, 96
N = G * G; rows = clips * T * N
W1 = torch.randn(Ct, Ct, 3, 1, 1, device="cuda") * 0.05; W2 = torch.randn(Ct, Ct, 1, 3, 3, device="cuda") * 0.03
W1p, W2p, W1b, W2b = ops.pack_conv_taps(W1), ops.pack_conv_taps(W2), ops.pack_conv_taps_dgrad(W1), ops.pack_conv_taps_dgrad(W2)
b1, b2, lb = (torch.randn(Ct, device="cuda") * 0.1 for _ in range(3)); lw = torch.randn(Ct, device="cuda") * 0.1 + 1
for rep in range(3):
    X = (torch.randn(rows, Ct, device="cuda") * 1.3).to(dt)
    dp = (torch.randn(rows, Ct, device="cuda") * 0.5).to(dt)
    big = torch.empty(96 << 20, device="cuda"); big.zero_(); del big                      # push the operands out of the Infinity Cache
    f = ops.temporal_net_fwd(X, W1p, b1, W2p, b2, lw, lb, clips, T, G, save_uv=True)
    g = ops.temporal_net_fwd(X, W1p, b1, W2p, b2, lw, lb, clips, T, G, save_uv=False)
    big = torch.empty(96 << 20, device="cuda"); big.zero_(); del big
    ops.temporal_net_bwd(dp, f["z"], X, f["mean"], f["rstd"], lw, W1b, W2b, clips, T, G)
    # the unfused sequences
    U = torch.empty_like(X); z = torch.empty_like(X); V = torch.empty_like(X); p = torch.empty_like(X); Xp = torch.empty_like(X)
    ops.layernorm(X, lw, lb, y=U)
    ops.gemm_nt(U, W1p, rows, Ct, Ct, taps=3, bias=b1, amap=ops.rowmap(L.RM_SHIFT, T * N, N, 1), C_out=z, C2_out=V)
    ops.gemm_nt(V, W2p, rows, Ct, Ct, taps=9, bias=b2, res=X, amap=ops.rowmap(L.RM_SPATIAL, G, 0, 1), C_out=p, C2_out=Xp)
    torch.cuda.synchronize()
print("done")
