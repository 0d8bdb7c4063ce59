"""GEMM shape sweep at ViT-B/16 b=32 shapes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
from tools.bench_ops import timeit
dt = torch.bfloat16
for (M, N, K, tag) in [(50432, 2304, 768, "qkv"), (50432, 768, 768, "out"), (50432, 3072, 768, "fc"), (50432, 768, 3072, "proj"), (8192, 8192, 8192, "8k")]:
    A = torch.randn(M, K, device="cuda").to(dt); B = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    C = torch.empty(M, N, device="cuda", dtype=dt); bias = torch.randn(N, device="cuda"); res = torch.randn(M, N, device="cuda").to(dt)
    t = timeit(lambda: ops.gemm_nt(A, B, M, N, K, bias=bias, C_out=C))
    t3 = timeit(lambda: ops.gemm_nt(A, B, M, N, K, bias=bias, res=res, C_out=C))
    t2 = timeit(lambda: torch.nn.functional.linear(A, B))
    print(f"{tag:6s} M={M} N={N} K={K}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF | +res {2*M*N*K/t3/1e12:7.1f} TF | hipBLASLt {2*M*N*K/t2/1e12:7.1f} TF", flush=True)
