"""Large plain weight gradients (input_linear 384x768, c_proj pair 384x480, ffn.c_fc 384x384), cold operands: tile shapes of gemm_tn_kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L


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
NSET = 10
M = 50432
for (NI, K, tag) in [(384, 768, "in_lin"), (384, 480, "proj pair"), (384, 384, "ffn_fc"), (96, 384, "tf_fc1")]:
    As = [torch.randn(M, NI, device="cuda").to(dt) for _ in range(NSET)]
    Bs = [torch.randn(M, K, device="cuda").to(dt) for _ in range(NSET)]
    out = torch.zeros(NI, K, device="cuda"); cs = torch.zeros(NI, device="cuda"); part = torch.empty(16 << 20, device="cuda")
    fns = [(lambda a=a, b=b: ops.gemm_tn(a, b, out, M, NI, K, colsum=cs, partial=part)) for a, b in zip(As, Bs)]
    t = timeit_rot(fns)
    byts = (M * NI + M * K) * 2
    print(f"gemm_tn {tag:10s} {NI}x{K}: {t*1e6:8.1f} us  {2*M*NI*K/t/1e12:7.1f} TF  (operands once: {byts/t/1e9:6.0f} GB/s)", flush=True)
    # correctness against torch
    out.zero_(); cs.zero_()
    ops.gemm_tn(As[0], Bs[0], out, M, NI, K, colsum=cs, partial=part)
    ref = As[0].float().t() @ Bs[0].float()
    err = float((out - ref).abs().max() / ref.abs().max())
    errb = float((cs - As[0].float().sum(0)).abs().max() / As[0].float().sum(0).abs().max())
    print(f"    max rel err {err:.2e}, bias grad {errb:.2e}")
    del As, Bs
