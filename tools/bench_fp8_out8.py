"""Cost of the DIST_EPI_OUT8 epilogue pass on the ViT-L/14 shapes (cold operands): usage: python tools/bench_fp8_out8.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
from bench_fp8 import timeit
NSET = 6
dt = torch.bfloat16
M = 65792
def run(tag, N, K, mode):
    As = [torch.randn(M, K, device="cuda").to(dt) for _ in range(NSET)]
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    qs = [ops.quant_rows_fp8(a) for a in As]
    qw, sw = ops.quant_rows_fp8(W)
    bias = torch.randn(N, device="cuda")
    Cs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(NSET)]
    C8s = [torch.empty(M, N, device="cuda", dtype=torch.uint8) for _ in range(NSET)]
    Rs = [torch.randn(M, N, device="cuda").to(dt) for _ in range(NSET)]
    sc, am = torch.tensor([0.25], device="cuda"), torch.zeros(1, device="cuda")
    one = torch.tensor([0.01], device="cuda")
    fns = {
        "act bf16": [(lambda q=q, c=c: ops.gemm_nt(q[0], qw, M, N, K, bias=bias, C2_out=c, fp8=(q[1], sw))) for q, c in zip(qs, Cs)],
        "act e4m3 only": [(lambda q=q, c8=c8: ops.gemm_nt(q[0], qw, M, N, K, bias=bias, out8=(c8, sc, am), act_only8=True, fp8=(q[1], sw))) for q, c8 in zip(qs, C8s)],
        "act e4m3 only, no amax": [(lambda q=q, c8=c8: ops.gemm_nt(q[0], qw, M, N, K, bias=bias, out8=(c8, sc, None), act_only8=True, fp8=(q[1], sw))) for q, c8 in zip(qs, C8s)],
        "res bf16": [(lambda q=q, c=c, r=r: ops.gemm_nt(q[0], qw, M, N, K, bias=bias, res=r, C_out=c, fp8=(q[1], sw))) for q, c, r in zip(qs, Cs, Rs)],
        "res bf16 + e4m3 image": [(lambda q=q, c=c, r=r, c8=c8: ops.gemm_nt(q[0], qw, M, N, K, bias=bias, res=r, C_out=c, out8=(c8, sc, am), fp8=(q[1], sw))) for q, c, r, c8 in zip(qs, Cs, Rs, C8s)],
        "res bf16, scalar A scale": [(lambda q=q, c=c, r=r: ops.gemm_nt(q[0], qw, M, N, K, bias=bias, res=r, C_out=c, fp8=(one, sw))) for q, c, r in zip(qs, Cs, Rs)],
    }
    for k in mode:
        t = timeit(fns[k])
        print(f"{tag:10s} N={N} K={K} {k:28s}: {t*1e6:7.1f} us", flush=True)
run("L/14 fc", 4096, 1024, ["act bf16", "act e4m3 only", "act e4m3 only, no amax"])
run("L/14 out", 1024, 1024, ["res bf16", "res bf16 + e4m3 image"])
run("L/14 proj", 1024, 4096, ["res bf16", "res bf16 + e4m3 image", "res bf16, scalar A scale"])
