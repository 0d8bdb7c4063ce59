"""IntegrationNetwork forward / backward GEMM shapes (M = 50432), cold operands: which kernel family takes them (DIST_AMD_FAST_KMIN)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from tools.bench_tnet import timeit_rot
dt = torch.bfloat16
NSET = 10
M = 50432
for (N, K, tag, extra) in [(384, 384, "ffn_fc", "act"), (384, 480, "proj pair", ""), (480, 384, "dzf dgrad", ""), (384, 384, "dNa dgrad", ""), (384, 768, "in_lin", "res"), (96, 384, "tf_fc1", "")]:
    As = [torch.randn(M, K, device="cuda").to(dt) for _ in range(NSET)]
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    Cs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(NSET)]
    C2s = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(NSET)] if "act" in extra else [None] * NSET
    Rs = [torch.randn(M, N, device="cuda").to(dt) for _ in range(NSET)] if "res" in extra else [None] * NSET
    bias = torch.randn(N, device="cuda")
    fns = [(lambda a=a, c=c, c2=c2, r=r: ops.gemm_nt(a, W, M, N, K, bias=bias, res=r, C_out=c, C2_out=c2)) for a, c, c2, r in zip(As, Cs, C2s, Rs)]
    t = timeit_rot(fns)
    print(f"gemm_nt {tag:10s} M={M} N={N} K={K} {extra:4s}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF", flush=True)
    del As, Cs, C2s, Rs
