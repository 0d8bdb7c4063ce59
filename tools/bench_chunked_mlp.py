"""Does the 256 MB Infinity Cache keep a producer -> consumer tensor out of HBM when the pair runs over row chunks?
fc (768 -> 3072, QuickGELU) then proj (3072 -> 768, + residual) over M = 50432 rows, same buffers every iteration (as the
engine does), whole-M launches vs 2 / 4 / 8 row chunks (fc of a chunk, then proj of that chunk)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
dt = torch.bfloat16; M = 50432; d = 768
x = torch.randn(M, d, device="cuda").to(dt); hid = torch.empty(M, 4 * d, device="cuda", dtype=dt); y = torch.empty(M, d, device="cuda", dtype=dt)
W1 = (torch.randn(4 * d, d, device="cuda") * d ** -0.5).to(dt); b1 = torch.randn(4 * d, device="cuda")
W2 = (torch.randn(d, 4 * d, device="cuda") * (4 * d) ** -0.5).to(dt); b2 = torch.randn(d, device="cuda")
def run(nchunks):
    rows = ((M // nchunks + 255) // 256) * 256
    r0 = 0
    while r0 < M:
        r = min(rows, M - r0)
        ops.gemm_nt(x[r0:r0 + r], W1, r, 4 * d, d, bias=b1, C2_out=hid[r0:r0 + r])
        ops.gemm_nt(hid[r0:r0 + r], W2, r, d, 4 * d, bias=b2, res=x[r0:r0 + r], C_out=y[r0:r0 + r])
        r0 += r
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps)
    return best
for n in (1, 2, 3, 4, 6, 8, 1):
    t = timeit(lambda: run(n))
    print(f"chunks={n}: fc+proj {t*1e3:7.1f} us  ({M // n} rows / chunk, hidden chunk {M // n * 4 * d * 2 / 1e6:.0f} MB)", flush=True)
