"""fp8 (block-scaled e4m3 MFMA) vs bf16 on the frozen-ViT GEMM shapes, cold operands (rotating buffer sets > Infinity Cache), plus the
row quantiser.  usage: python tools/bench_fp8.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops

def timeit(fns, reps=3):
    for f in fns: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for f in fns: f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * len(fns)) * 1e-3

NSET = 8
dt = torch.bfloat16
for (tag, M, N, K) in [("B/16 qkv", 50432, 2304, 768), ("B/16 out", 50432, 768, 768), ("B/16 fc", 50432, 3072, 768), ("B/16 proj", 50432, 768, 3072),
                       ("L/14 qkv", 65792, 3072, 1024), ("L/14 out", 65792, 1024, 1024), ("L/14 fc", 65792, 4096, 1024), ("L/14 proj", 65792, 1024, 4096),
                       ("8192^3", 8192, 8192, 8192)]:
    As = [torch.randn(M, K, device="cuda").to(dt) for _ in range(NSET)]
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt)
    Cs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(NSET)]
    bias = torch.randn(N, device="cuda")
    t16 = timeit([(lambda a=a, c=c: ops.gemm_nt(a, W, M, N, K, bias=bias, C_out=c)) for a, c in zip(As, Cs)])
    qs = [ops.quant_rows_fp8(a) for a in As]
    qw, sw = ops.quant_rows_fp8(W)
    t8 = timeit([(lambda q=q, c=c: ops.gemm_nt(q[0], qw, M, N, K, bias=bias, C_out=c, fp8=(q[1], sw))) for q, c in zip(qs, Cs)])
    tq = timeit([(lambda a=a, q=q: ops.quant_rows_fp8(a, q[0], q[1])) for a, q in zip(As, qs)])
    fl = 2.0 * M * N * K
    print(f"{tag:10s} M={M} N={N} K={K}: bf16 {t16*1e6:7.1f} us {fl/t16/1e12:7.1f} TF | fp8 {t8*1e6:7.1f} us {fl/t8/1e12:7.1f} TF ({t16/t8:.2f}x) | "
          f"quantise A {tq*1e6:6.1f} us ({M*K*3/tq/1e9:5.0f} GB/s) | fp8 + quantise {(t8+tq)*1e6:7.1f} us ({t16/(t8+tq):.2f}x)", flush=True)
    del As, Cs, qs
