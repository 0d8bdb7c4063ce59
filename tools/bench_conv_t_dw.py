"""Temporal-tap weight gradients (conv_t_dw.hip) at the bench size: time per launch (main kernel + second phase), rotating operand sets.
Run once with DIST_AMD_CONV9=1 and once with 0 (the generic tap-per-tile kernel).  python tools/bench_conv_t_dw.py <tag>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from check_pp import timeit_rot
from dist_amd import ops

tag = sys.argv[1] if len(sys.argv) > 1 else ""
g = torch.Generator(device="cuda"); g.manual_seed(11)
part = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
for (name, clips, T, N, taps, K, lda, mb) in (("tn_fc1 96x96x3", 32, 16, 196, 3, 96, 96, 0), ("tf_fc2 96x96x3 (dY in 480-wide rows)", 32, 8, 197, 3, 96, 480, 0),
                                              ("stem 96x768x5", 32, 16, 196, 5, 768, 96, 0), ("tn_fc1, 256 blocks", 32, 16, 196, 3, 96, 96, 256)):
    M = clips * T * N
    sets = []
    for _ in range(4):
        Aw = torch.randn(M, lda, device="cuda", generator=g).to(torch.bfloat16)
        B = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        sets.append((Aw, B))
    out = torch.zeros(96, K * taps, dtype=torch.float32, device="cuda")
    cs = torch.zeros(96, dtype=torch.float32, device="cuda")
    bm = ops.rowmap(1, T * N, N, 1)
    lay = dict(so_i=K * taps, so_tap=K // 3, so_outer=(K // 3) * taps, inner=K // 3) if K == 768 else dict(so_i=K * taps, so_tap=1, so_outer=taps, inner=1)
    def call(Aw, B):
        A = Aw[:, lda - 96:]
        ops.gemm_tn(A, B, out, M, 96, K, taps=taps, bmap=bm, colsum=cs, partial=part, lda=lda, max_blocks=mb, **lay)
    t = timeit_rot([(lambda a=a, b=b: call(a, b)) for a, b in sets]) * 1e6
    gf = 2.0 * M * 96 * K * taps / 1e9
    print(f"[{tag}] {name:40s} {t:7.1f} us  ({gf / t * 1e-3:6.1f} TF, operands {(M * 96 + M * K) * 2 / 1e6:6.1f} MB)", flush=True)
