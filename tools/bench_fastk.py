"""Fast GEMM kernel: time vs K at fixed M, N (cold operands) -> per-K-tile cost and fixed cost per launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
def timeit_rot(fns, reps=3):
    for f in fns: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for f in fns: f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * len(fns)) * 1e-3
dt = torch.bfloat16
M = 50432
for N in (768, 2304):
    for K in (128, 256, 512, 768, 1536, 3072):
        n = max(3, min(12, (3 << 30) // (M * (K + N) * 2)))
        As = [torch.randn(M, K, device="cuda").to(dt) for _ in range(n)]
        Cs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(n)]
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt); bias = torch.randn(N, device="cuda")
        t = timeit_rot([(lambda a=a, c=c: ops.gemm_nt(a, W, M, N, K, bias=bias, C_out=c)) for a, c in zip(As, Cs)])
        print(f"N={N:5d} K={K:5d}: {t*1e6:8.1f} us {2*M*N*K/t/1e12:7.1f} TF", flush=True)
        del As, Cs
