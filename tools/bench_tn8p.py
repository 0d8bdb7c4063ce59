"""The large plain weight gradients on gemm_tn8p_kernel (two-group LDS-DMA, gemm_tn8p.hip) against gemm_tn_kernel (DIST_AMD_TN8P=0 in another
process): correctness against fp64 torch on the engine's three shapes (+ ragged row counts, split destination, swapped orientation), cold operands."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops


def timeit_rot(fns, reps=4):
    for f in fns: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for f in fns: f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * len(fns)) * 1e-3


dt = torch.bfloat16
part = torch.empty(16 << 20, device="cuda")
print("DIST_AMD_TN8P =", os.environ.get("DIST_AMD_TN8P", "(default 1)"), " DIST_AMD_TN8P_BLOCKS =", os.environ.get("DIST_AMD_TN8P_BLOCKS", "(default)"))
# ---- correctness
CASES = [] if '--no-check' in sys.argv else [(50432, 384, 768, 384, 768, 0), (50432, 384, 480, 384, 480, 384), (50432, 480, 384, 576, 384, 0),
                                    (8200, 384, 768, 384, 768, 0), (9001, 384, 480, 384, 480, 384), (12345, 480, 384, 576, 384, 0),
                                    (8192 + 70, 192, 256, 192, 256, 0), (20000, 384, 1024, 384, 1024, 0), (10000, 200, 200, 200, 200, 0)]
for (M, NI, K, lda, ldb, split) in CASES:
    g = torch.Generator(device="cuda"); g.manual_seed(M + NI)
    A = (torch.randn(M, lda, device="cuda", generator=g) + 0.05).to(dt)
    B = torch.randn(M, ldb, device="cuda", generator=g).to(dt)
    ref = A[:, :NI].double().t() @ B[:, :K].double()
    refb = A[:, :NI].double().sum(0)
    cs = torch.zeros(NI, device="cuda")
    if split:
        out = torch.zeros(NI, split, device="cuda"); out2 = torch.zeros(NI, K - split, device="cuda"); cs2 = torch.zeros(NI, device="cuda")
        ops.gemm_tn(A, B, out, M, NI, K, so_i=split, so_tap=0, colsum=cs, partial=part, out2=out2, split_c=split, so_i2=K - split, colsum2=cs2)
        got = torch.cat([out, out2], 1)
        eb2 = float((cs2.double() - refb).abs().max() / refb.abs().max())
    else:
        out = torch.zeros(NI, K, device="cuda")
        ops.gemm_tn(A, B, out, M, NI, K, colsum=cs, partial=part)
        got, eb2 = out, 0.0
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    eb = float((cs.double() - refb).abs().max() / refb.abs().max())
    # repeatability (bit for bit)
    out_b = torch.zeros(NI, K, device="cuda"); cs_b = torch.zeros(NI, device="cuda")
    ops.gemm_tn(A, B, out_b, M, NI, K, colsum=cs_b, partial=part)
    out_c = torch.zeros(NI, K, device="cuda"); cs_c = torch.zeros(NI, device="cuda")
    ops.gemm_tn(A, B, out_c, M, NI, K, colsum=cs_c, partial=part)
    same = bool((out_b == out_c).all()) and bool((cs_b == cs_c).all())
    print(f"M {M:6d} {NI}x{K} (ld {lda}/{ldb}, split {split}): max rel err {err:.2e}, bias grad {eb:.2e} {eb2:.2e}, repeatable {same}", flush=True)
    assert err < 2e-5 and eb < 2e-5 and eb2 < 2e-5, "MISMATCH"
    del A, B

# ---- timing, cold operands (10 operand sets in rotation); --blocks N: dist_gemm_tn_args.max_blocks (default 256: the lone launch's best form)
MB = int(sys.argv[sys.argv.index('--blocks') + 1]) if '--blocks' in sys.argv else 256
NSET = 10
M = 50432
for (NI, K, lda, ldb, tag) in [(384, 768, 384, 768, "in_lin"), (384, 480, 384, 480, "proj pair"), (480, 384, 576, 384, "[fc|fc1]"), (384, 384, 384, 384, "ffn_fc")]:
    As = [torch.randn(M, lda, device="cuda").to(dt) for _ in range(NSET)]
    Bs = [torch.randn(M, ldb, device="cuda").to(dt) for _ in range(NSET)]
    out = torch.zeros(NI, K, device="cuda"); cs = torch.zeros(NI, device="cuda")
    fns = [(lambda a=a, b=b: ops.gemm_tn(a, b, out, M, NI, K, colsum=cs, partial=part, max_blocks=MB)) for a, b in zip(As, Bs)]
    t = timeit_rot(fns)
    byts = (M * NI + M * K) * 2
    print(f"gemm_tn {tag:10s} {NI}x{K}: {t*1e6:8.1f} us  {2*M*NI*K/t/1e12:7.1f} TF = {2*M*NI*K/t/2.5e15:.3f} of peak (operands once: {byts/t/1e9:6.0f} GB/s)", flush=True)
    del As, Bs
