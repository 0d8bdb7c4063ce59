import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
from bench_ln import timeit_rot
dt = torch.bfloat16
for rows, C in [(50432, 768), (50432, 384), (100352, 96)]:
    xs = [torch.randn(rows, C, device="cuda").to(dt) for _ in range(12)]; ys = [torch.empty_like(x) for x in xs]
    w = torch.ones(C, device="cuda")
    mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda")
    t = timeit_rot([(lambda x=x, y=y: ops.layernorm_bwd(x, mean, rstd, y, w, dx=y)) for x, y in zip(xs, ys)])
    print(f"ln_bwd without parameter gradients {rows}x{C}: {t*1e6:7.1f} us {3*rows*C*2/t/1e9:7.0f} GB/s")
