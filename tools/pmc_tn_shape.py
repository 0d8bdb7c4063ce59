"""one weight-gradient shape for PMC passes: python tools/pmc_tn_shape.py [NI K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
M = 50432
NI, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (384, 768)
dt = torch.bfloat16
As = [torch.randn(M, NI, device="cuda").to(dt) for _ in range(6)]
Bs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(6)]
out = torch.zeros(NI, K, device="cuda"); cs = torch.zeros(NI, device="cuda"); part = torch.empty(16 << 20, device="cuda")
for r in range(2):
    for a, b in zip(As, Bs):
        ops.gemm_tn(a, b, out, M, NI, K, colsum=cs, partial=part)
torch.cuda.synchronize()
print("done")
