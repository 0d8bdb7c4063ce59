"""Per-step vs fixed cost of the generic GEMM kernels: same shape, growing reduction length (cold operands)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from tools.bench_cold import timeit_rot

dt = torch.bfloat16
def mk(n, *shape): return [torch.randn(*shape, device="cuda").to(dt) for _ in range(n)]

print("== TN: dWi tile shape (NI=384, K=768), M grows; blocks stay ~396 ==")
for M in (6304, 12608, 25216, 50432, 100864, 201728):
    n = max(2, min(12, (1 << 30) // (M * (384 + 768) * 2)))
    As, Bs = mk(n, M, 384), mk(n, M, 768)
    out = torch.zeros(384, 768, device="cuda"); part = torch.empty(9 << 20, device="cuda")
    t = timeit_rot([(lambda a=a, b=b: ops.gemm_tn(a, b, out, M, 384, 768, partial=part)) for a, b in zip(As, Bs)])
    print(f"M={M:7d}: {t*1e6:8.1f} us {2*M*384*768/t/1e12:7.1f} TF", flush=True)
    del As, Bs
print("== TN: NI=96,K=96 taps=1 plain ==")
for M in (12544, 25088, 50176, 100352, 200704):
    n = 12
    As, Bs = mk(n, M, 96), mk(n, M, 96)
    out = torch.zeros(96, 96, device="cuda"); part = torch.empty(9 << 20, device="cuda")
    t = timeit_rot([(lambda a=a, b=b: ops.gemm_tn(a, b, out, M, 96, 96, partial=part)) for a, b in zip(As, Bs)])
    print(f"M={M:7d}: {t*1e6:8.1f} us {2*M*96*96/t/1e12:7.1f} TF", flush=True)
    del As, Bs
print("== NT: M=50432 N=384, K grows (128x128x64 kernel; K=768+ goes to the fast kernel) ==")
for K in (64, 128, 192, 384, 640):
    As = mk(12, 50432, K); W = (torch.randn(384, K, device="cuda") * K ** -0.5).to(dt); Cs = mk(12, 50432, 384)
    bias = torch.randn(384, device="cuda")
    t = timeit_rot([(lambda a=a, c=c: ops.gemm_nt(a, W, 50432, 384, K, bias=bias, C_out=c)) for a, c in zip(As, Cs)])
    print(f"K={K:5d}: {t*1e6:8.1f} us {2*50432*384*K/t/1e12:7.1f} TF", flush=True)
    del As, Cs
print("== NT: M=100352 N=96 K=96, taps grow (temporal shift map, 128x96x32 kernel) ==")
for taps in (1, 3, 5, 9):
    As = mk(12, 100352, 96); W = (torch.randn(96, 96 * taps, device="cuda") * (96 * taps) ** -0.5).to(dt); Cs = mk(12, 100352, 96)
    bias = torch.randn(96, device="cuda")
    t = timeit_rot([(lambda a=a, c=c: ops.gemm_nt(a, W, 100352, 96, 96, taps=taps, bias=bias, C_out=c, amap=ops.rowmap(L.RM_SHIFT, 16 * 196, 196))) for a, c in zip(As, Cs)])
    print(f"taps={taps}: {t*1e6:8.1f} us {2*100352*96*96*taps/t/1e12:7.1f} TF", flush=True)
    del As, Cs
