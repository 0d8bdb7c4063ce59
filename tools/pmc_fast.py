"""The four frozen-ViT GEMM shapes of config 2 through the fast kernel, 3 launches each on rotating cold operands.
Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) for roofline.traffic."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
dt = torch.bfloat16
M = 50432
for (N, K, tag) in [(2304, 768, "qkv"), (768, 768, "out"), (3072, 768, "fc"), (768, 3072, "proj")]:
    As = [torch.randn(M, K, device="cuda").to(dt) for _ in range(3)]
    Cs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(3)]
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt); bias = torch.randn(N, device="cuda")
    for a, c in zip(As, Cs):
        ops.gemm_nt(a, W, M, N, K, bias=bias, C_out=c)
    torch.cuda.synchronize()
    del As, Cs
print("done")
