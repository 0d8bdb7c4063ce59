"""Fast GEMM on the four frozen-ViT shapes + the branch's large Linears (cold operands, rotating buffer sets > MALL).
Run once per main loop in the same gpurun call for the A/B:  DIST_AMD_FAST_8P=0 python tools/bench_fast8p.py ; python tools/bench_fast8p.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
def timeit_rot(fns, reps=4):
    for f in fns: f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            for f in fns: f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / (reps * len(fns)) * 1e-3)
    return best
dt = torch.bfloat16; M = 50432
tag8 = "lockstep-32" if os.environ.get("DIST_AMD_FAST_8P") == "0" else "two-group-64 var=%s stagger=%s" % (os.environ.get("DIST_AMD_FAST_VAR", "0"), os.environ.get("DIST_AMD_FAST_STAGGER", "0"))
tot = 0.0
for (N, K, tag, act, cnt) in [(3072, 768, "fc", True, 12), (2304, 768, "qkv", False, 12), (768, 3072, "proj", False, 12), (768, 768, "out", False, 12),
                              (384, 768, "in_lin", False, 12), (8192, 8192, "8192^3", False, 0)]:
    Mm = 8192 if tag == "8192^3" else M
    n = 6 if tag != "8192^3" else 3
    As = [torch.randn(Mm, K, device="cuda").to(dt) for _ in range(n)]
    Cs = [torch.empty(Mm, N, device="cuda", dtype=dt) for _ in range(n)]
    W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(dt); bias = torch.randn(N, device="cuda")
    if act: fns = [(lambda a=a, c=c: ops.gemm_nt(a, W, Mm, N, K, bias=bias, C2_out=c)) for a, c in zip(As, Cs)]
    else: fns = [(lambda a=a, c=c: ops.gemm_nt(a, W, Mm, N, K, bias=bias, C_out=c)) for a, c in zip(As, Cs)]
    t = timeit_rot(fns)
    tot += t * cnt
    print(f"[{tag8}] {tag:7s} M={Mm} N={N} K={K}: {t*1e6:7.1f} us {2*Mm*N*K/t/1e12:7.1f} TF", flush=True)
    del As, Cs
print(f"[{tag8}] ViT + input_linear GEMMs per step (12 layers): {tot*1e3:.2f} ms", flush=True)
