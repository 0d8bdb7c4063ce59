import sys, os
sys.path.insert(0, os.getcwd())
import torch
from dist_amd import ops, lib as L
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
def mk(n, *shape): return [torch.randn(*shape, device="cuda").to(dt) for _ in range(n)]
for (M, NI, K) in [(201728, 384, 768), (50432, 384, 768), (50432, 384, 384), (50432, 384, 96), (200704, 96, 96), (50176, 96, 384)]:
    n = 4 if M > 100000 else 12
    As, Bs = mk(n, M, NI), mk(n, M, K)
    out = torch.zeros(NI, K, device="cuda"); part = torch.empty(9 << 20, device="cuda")
    cs = torch.zeros(NI, device="cuda")
    for tag, tr, c in [("full", 1, None), ("full+colsum", 1, cs), ("full", 1, None), ("full+colsum", 1, cs)]:
        t = timeit_rot([(lambda a=a, b=b: ops.gemm_tn(a, b, out, M, NI, K, partial=part, use_tr=tr, colsum=c)) for a, b in zip(As, Bs)])
        print(f"M={M} NI={NI} K={K} {tag:20s}: {t*1e6:8.1f} us {2*M*NI*K/t/1e12:7.1f} TF", flush=True)
    t = timeit_rot([(lambda a=a, b=b: torch.mm(a.t(), b)) for a, b in zip(As, Bs)])
    print(f"M={M} NI={NI} K={K} torch.mm(A.t(),B)    : {t*1e6:8.1f} us {2*M*NI*K/t/1e12:7.1f} TF", flush=True)
    del As, Bs
