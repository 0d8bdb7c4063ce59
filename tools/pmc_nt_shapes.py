"""The TemporalNet convolutions and two plain branch GEMMs through gemm_nt, 3 launches each on rotating cold operands
(run under rocprofv3 --pmc ... for stall analysis)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
dt = torch.bfloat16
def mk(*shape, n=3): return [torch.randn(*shape, device="cuda").to(dt) for _ in range(n)]
for (M, N, K, taps, mode, kw, tag, extra) in [(100352, 96, 96, 9, L.RM_SPATIAL, (14, 0), "conv3x3", "res+act"), (100352, 96, 96, 3, L.RM_SHIFT, (16*196, 196), "conv_t", "act"),
                                               (50432, 96, 384, 1, 0, (0, 0), "tf_fc1", ""), (50432, 384, 96, 1, 0, (0, 0), "tf_proj", "res")]:
    As = mk(M, K); W = (torch.randn(N, taps * K, device="cuda") * (taps * K) ** -0.5).to(dt)
    Cs, C2s, Rs = mk(M, N), mk(M, N) if "act" in extra else [None] * 3, mk(M, N) if "res" in extra else [None] * 3
    bias = torch.randn(N, device="cuda")
    for a, c, c2, r in zip(As, Cs, C2s, Rs):
        ops.gemm_nt(a, W, M, N, K, taps=taps, bias=bias, res=r, C_out=c, C2_out=c2, amap=ops.rowmap(mode, kw[0], kw[1]))
    torch.cuda.synchronize()
print("done")
