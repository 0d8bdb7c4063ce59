"""ViT attention kernel alone, cold operands: B/16 (L = 197, 12 heads) and L/14 (L = 257, 16 heads) shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from tools.bench_tnet import timeit_rot as timeit
for tag, frames, Lt, heads in (("B/16 b=32", 256, 197, 12), ("L/14 b=8", 256, 257, 16), ("L/14 b=16", 512, 257, 16)):
    qs = [torch.randn(frames * heads * 3 * Lt, 64, device="cuda").to(torch.bfloat16) for _ in range(6)]
    t = timeit([(lambda q=q: ops.attention(q, frames, Lt, heads, layout=L.QKV_HEADS)) for q in qs])
    fl = 4.0 * frames * heads * Lt * Lt * 64
    by = frames * Lt * heads * 64 * 2 * 4
    print(f"attention {tag:10s} frames={frames} L={Lt} heads={heads}: {t*1e6:7.1f} us  {fl/t/1e12:6.1f} TF  {by/t/1e9:6.0f} GB/s", flush=True)
