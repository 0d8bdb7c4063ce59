"""e4m3 q | k | v: in_proj output stage and attention input stage, ViT-L/14 shapes, cold operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from bench_attn import timeit  # noqa (also prints the bf16 attention lines)
frames, Lt, heads, K = 256, 257, 16, 1024
M, d = frames * Lt, heads * 64
NS = 5
As = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(NS)]
W = (torch.randn(3 * d, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
qs = [ops.quant_rows_fp8(a) for a in As]
qw, sw = ops.quant_rows_fp8(W)
bias = torch.randn(3 * d, device="cuda")
o16 = [torch.empty(frames * heads * 3 * Lt, 64, dtype=torch.bfloat16, device="cuda") for _ in range(NS)]
o8 = [torch.empty(frames * heads * 3 * Lt, 64, dtype=torch.uint8, device="cuda") for _ in range(NS)]
sc, am = torch.tensor([2.0 ** -5], device="cuda"), torch.zeros(1, device="cuda")
om = ops.outmap(L.OM_HEADS, Lt, heads)
t = timeit([(lambda q=q, c=c: ops.gemm_nt(q[0], qw, M, 3 * d, K, bias=bias, C_out=c, ldc=64, omap=om, fp8=(q[1], sw))) for q, c in zip(qs, o16)])
print(f"in_proj fp8 -> bf16 head-major      : {t*1e6:7.1f} us")
t = timeit([(lambda q=q, c=c: ops.gemm_nt(q[0], qw, M, 3 * d, K, bias=bias, omap=om, out8=(c, sc, am), fp8=(q[1], sw))) for q, c in zip(qs, o8)])
print(f"in_proj fp8 -> e4m3 only head-major : {t*1e6:7.1f} us")
t = timeit([(lambda c=c: ops.attention(c, frames, Lt, heads, layout=L.QKV_HEADS)) for c in o16])
print(f"attention bf16 qkv -> bf16          : {t*1e6:7.1f} us")
t = timeit([(lambda c=c: ops.attention_out8(c, frames, Lt, heads, sc, am)) for c in o16])
print(f"attention bf16 qkv -> e4m3          : {t*1e6:7.1f} us")
t = timeit([(lambda c=c: ops.attention_fp8(c, sc, frames, Lt, heads)) for c in o8])
print(f"attention e4m3 qkv -> bf16          : {t*1e6:7.1f} us")
t = timeit([(lambda c=c: ops.attention_fp8(c, sc, frames, Lt, heads, out8_scale=sc, amax=am)) for c in o8])
print(f"attention e4m3 qkv -> e4m3          : {t*1e6:7.1f} us")
