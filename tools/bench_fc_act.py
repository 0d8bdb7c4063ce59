import sys, os
sys.path.insert(0, os.getcwd())
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
dt = torch.bfloat16; M = 50432
for (N, K) in [(3072, 768), (2304, 768), (1536, 768), (768, 768)]:
    As = [torch.randn(M, K, device="cuda").to(dt) for _ in range(6)]
    Cs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in range(6)]
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt); bias = torch.randn(N, device="cuda")
    t1 = timeit_rot([(lambda a=a, c=c: ops.gemm_nt(a, W, M, N, K, bias=bias, C_out=c)) for a, c in zip(As, Cs)])
    t2 = timeit_rot([(lambda a=a, c=c: ops.gemm_nt(a, W, M, N, K, bias=bias, C2_out=c)) for a, c in zip(As, Cs)])
    tiles = 197 * (N // 256)
    print(f"N={N} K={K}: plain {t1*1e6:7.1f} us ({2*M*N*K/t1/1e12:6.1f} TF)  act-only {t2*1e6:7.1f} us ({2*M*N*K/t2/1e12:6.1f} TF)  tiles={tiles} rounds={tiles/256:.2f}", flush=True)
    del As, Cs
