"""A few launches of the ViT-B/16 attention (256 frames, L = 197, 12 heads, head-major q | k | v) for rocprofv3 passes (tools/r04_pmc_attn.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
frames, Lt, heads = 256, 197, 12
qs = [torch.randn(frames * heads * 3 * Lt, 64, device="cuda").to(torch.bfloat16) for _ in range(6)]
for _ in range(3):
    for q in qs:
        ops.attention(q, frames, Lt, heads, layout=L.QKV_HEADS)
torch.cuda.synchronize()
print("done")
