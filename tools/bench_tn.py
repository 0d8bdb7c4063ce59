import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from tools.bench_ops import timeit
dt = torch.bfloat16
for (M, NI, K, taps, mode, kw, tag) in [(100352, 96, 96, 9, L.RM_SPATIAL, (14, 0), "tn_fc2"), (100352, 96, 96, 3, L.RM_SHIFT, (16*196, 196), "tn_fc1"),
                                        (50432, 384, 768, 1, 0, (0, 0), "dWi"), (50432, 384, 384, 1, 0, (0, 0), "dWffn"), (50432, 384, 96, 1, 0, (0,0), "dW3"),
                                        (100352, 96, 768, 5, L.RM_SHIFT, (16*196, 196), "stem")]:
    A = torch.randn(M, NI, device="cuda").to(dt); B = torch.randn(M, K, device="cuda").to(dt)
    out = torch.zeros(NI, K * taps, device="cuda")
    t = timeit(lambda: ops.gemm_tn(A, B, out, M, NI, K, taps=taps, bmap=ops.rowmap(mode, kw[0], kw[1])))
    print(f"gemm_tn {tag:8s} M={M} NI={NI} K={K} taps={taps}: {t*1e6:8.1f} us  {2*M*NI*K*taps/t/1e12:7.1f} TF", flush=True)
