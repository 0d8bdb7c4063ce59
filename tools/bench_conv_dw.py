"""conv3x3 weight gradient (96 -> 96 channels, 14 x 14 plane, b = 32 x T = 16 frames): frame-resident kernel (conv_dw.hip) vs the generic tap-per-tile
kernel (DIST_AMD_CONV9=0), cold operands in rotation; block counts 96 (the engine's cap) and 256."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from tools.check_pp import timeit_rot
G, C, taps, frames = 14, 96, 9, 512
M = frames * G * G
sets = [(torch.randn(M, C, device="cuda").to(torch.bfloat16), torch.randn(M, C, device="cuda").to(torch.bfloat16)) for _ in range(6)]
part = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
out = torch.zeros(C, C, taps, device="cuda"); cs = torch.zeros(C, device="cuda")
bm = ops.rowmap(L.RM_SPATIAL, G, 0, 1)
for mb in (96, 128, 256, 64):
    fns = [(lambda a=a, b=b: ops.gemm_tn(a, b, out, M, C, C, taps=taps, bmap=bm, so_i=C * taps, so_tap=1, so_outer=taps, inner=1, colsum=cs, partial=part, max_blocks=mb)) for a, b in sets]
    t = timeit_rot(fns, reps=4)
    print(f"conv3x3 dW [{'generic' if os.environ.get('DIST_AMD_CONV9') == '0' else 'frame-resident'}] max_blocks {mb:3d}: {t*1e6:7.1f} us  {2*M*C*C*taps/t/1e12:6.1f} TF  ({2*M*C*2/t/1e9:6.0f} GB/s of operands)", flush=True)
