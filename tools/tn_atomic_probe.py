"""old gemm_tn_kernel with the atomic epilogue (partial=None) against the two-phase form: what do ~4 M fp32 global atomics cost? (run under rocprofv3 --kernel-trace --stats)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
M, NI, K = 50432, 384, 768
dt = torch.bfloat16
As = [torch.randn(M, NI, device="cuda").to(dt) for _ in range(6)]
Bs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(6)]
out = torch.zeros(NI, K, device="cuda"); cs = torch.zeros(NI, device="cuda"); part = torch.empty(16 << 20, device="cuda")
use_part = len(sys.argv) > 1 and sys.argv[1] == "partial"
for r in range(2):
    for a, b in zip(As, Bs):
        ops.gemm_tn(a, b, out, M, NI, K, colsum=cs, partial=part if use_part else None)
torch.cuda.synchronize()
print("done")
