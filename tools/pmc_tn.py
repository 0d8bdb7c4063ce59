import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
dt = torch.bfloat16
M, NI, K = 50432, 384, 384
As = [torch.randn(M, NI, device="cuda").to(dt) for _ in range(8)]
Bs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(8)]
out = torch.zeros(NI, K, device="cuda")
for a, b in zip(As, Bs):
    ops.gemm_tn(a, b, out, M, NI, K)
torch.cuda.synchronize()
