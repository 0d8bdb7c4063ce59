"""Plain-epilogue GEMM on the ViT shapes (the F = 0 instantiation of gemm_pp.hip, or gemm_fast8p with DIST_AMD_FAST_PP=0): time per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
from tools.check_pp import timeit_rot
dt = torch.bfloat16
for (M, N, K, tag) in [(50432, 2304, 768, "qkv"), (50432, 3072, 768, "fc"), (50432, 768, 3072, "proj"), (50432, 768, 768, "out")]:
    As = [torch.randn(M, K, device="cuda").to(dt) for _ in range(4)]
    Cs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(4)]
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt); bias = torch.randn(N, device="cuda")
    fns = [(lambda a=a, c=c: ops.gemm_nt(a, W, M, N, K, bias=bias, C_out=c)) for a, c in zip(As, Cs)]
    t = timeit_rot(fns)
    print(f"{tag:5s} {M}x{N}x{K}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:7.1f} TF", flush=True)
    del As, Cs
