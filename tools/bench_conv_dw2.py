"""conv3x3 dW kernel: time against frames per block (64 blocks; 1, 2, 4, 8 frames each), cold (6 operand sets in rotation) and warm (one set)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from tools.check_pp import timeit_rot
G, C, taps = 14, 96, 9
part = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
out = torch.zeros(C, C, taps, device="cuda"); cs = torch.zeros(C, device="cuda")
bm = ops.rowmap(L.RM_SPATIAL, G, 0, 1)
for nsets in (6, 1):
    for frames in (64, 128, 256, 512, 1024):
        M = frames * G * G
        sets = [(torch.randn(M, C, device="cuda").to(torch.bfloat16), torch.randn(M, C, device="cuda").to(torch.bfloat16)) for _ in range(nsets)]
        fns = [(lambda a=a, b=b: ops.gemm_tn(a, b, out, M, C, C, taps=taps, bmap=bm, so_i=C * taps, so_tap=1, so_outer=taps, inner=1, colsum=cs, partial=part, max_blocks=64)) for a, b in sets]
        t = timeit_rot(fns, reps=4)
        print(f"{'cold' if nsets > 1 else 'warm'} frames {frames:5d} ({frames // 64} per block, 64 blocks): {t*1e6:7.1f} us (main + reduce)", flush=True)
