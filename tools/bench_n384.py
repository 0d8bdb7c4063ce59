"""input_linear (dist.py:229: 768 -> 384 on 50 432 token rows, bias + residual) on the ViT GEMM kernel: N = 384 is one and a half 256-column tiles, so a quarter of the
launch's MFMAs multiply zeros.  Timing-only library: DIST_AMD_FAST_NW=4 selects the 256 x 128 / 4-wave / two-blocks-per-CU shape (three full column tiles).
python tools/bench_n384.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
from tools.bench_tnet import timeit_rot
M, N, K = 50432, 384, 768
g = torch.Generator(device="cuda"); g.manual_seed(3)
rnd = lambda shape, sc=1.0, dt=torch.bfloat16: (torch.randn(shape, device="cuda", generator=g) * sc).to(dt)
As = [rnd((M, K)) for _ in range(6)]
W = rnd((N, K), K ** -0.5); bias = rnd((N,), 1.0, torch.float32)
Rs = [rnd((M, N)) for _ in range(6)]
outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in range(6)]
fns = [(lambda a=a, r=r, o=o: ops.gemm_nt(a, W, M, N, K, bias=bias, res=r, C_out=o)) for a, r, o in zip(As, Rs, outs)]
t = min(timeit_rot(fns) for _ in range(3))
ref = (As[0].float() @ W.float().t() + bias + Rs[0].float())
err = float((outs[0].float() - ref).abs().max() / ref.abs().max())
print(f"N=384 K=768 bias+res, FAST_NW={os.environ.get('DIST_AMD_FAST_NW', '8')}: {t * 1e6:7.1f} us   {2.0 * M * N * K / t / 1e12:6.1f} TF   rel err {err:.2e}", flush=True)
