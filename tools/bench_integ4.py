"""The fused IntegrationNetwork launches in the ENGINE's form at the bench size (T2I in front, I2T behind; training = xhat form, and inference), rotating
cold operands: python tools/bench_integ4.py [--bwd].  The tile form is chosen per process: DIST_AMD_INTEG_W4=0 -> 128-row tiles / 8 waves (integ.hip),
default -> 64-row tiles / 4 waves, two workgroups per CU (integ4.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from tools.bench_tnet import timeit_rot
from tests.test_integ_gpu import make, CI, C4

clips, t, Ltok = 32, 8, 197
rows, N = clips * t * Ltok, Ltok - 1
w, _ = make(1, 8, 16, seed=1)
wc = {k: v.cuda() for k, v in w.items()}
Wt = torch.randn(CI, C4, 2, 1, 1, device="cuda") * 0.07
Wi = torch.randn(C4, CI, device="cuda") * CI ** -0.5
bt = torch.randn(CI, device="cuda") * 0.1; cls = torch.randn(t, CI, device="cuda") * 0.5; bi = torch.randn(C4, device="cuda") * 0.1
pk = ops.integration_pack(wc, bwd=True, t2i_w=Wt, i2t_w=Wi)
NSET = 6
xs = [(torch.randn(rows, CI, device="cuda") * 1.5 + 0.3).to(torch.bfloat16) for _ in range(NSET)]
Xps = [(torch.randn(clips * 2 * t * N, C4, device="cuda") * 0.8).to(torch.bfloat16) for _ in range(NSET)]
tag = "W4=" + os.environ.get("DIST_AMD_INTEG_W4", "1")
for train in (True, False):
    kw = dict(xhat=True) if train else dict(train=False)
    outs = [ops.integration_fwd(x, pk, clips, t, Ltok, t2i=(xp, bt, cls), i2t_bias=bi, **kw) for x, xp in zip(xs, Xps)]
    fns = [(lambda x=x, xp=xp, o=o: ops.integration_fwd(x, pk, clips, t, Ltok, t2i=(xp, bt, cls), i2t_bias=bi, out=o, **kw)) for x, xp, o in zip(xs, Xps, outs)]
    tt = min(timeit_rot(fns) for _ in range(3))
    print(f"{tag} integration_fwd (T2I + I2T) train={train}: {tt*1e6:8.1f} us", flush=True)
if "--bwd" in sys.argv:
    saved = [ops.integration_fwd(x, pk, clips, t, Ltok, xhat=True) for x in xs]
    dRs = [(torch.randn(rows, CI, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(NSET)]
    dXn = [(torch.randn(clips * 2 * t * N, C4, device="cuda") * 0.3).to(torch.bfloat16) for _ in range(NSET)]
    pact = [(torch.randn(clips * 2 * t * N, C4, device="cuda") * 0.8).to(torch.bfloat16) for _ in range(NSET)]
    fns = [(lambda d=d, sv=sv, dx=dx, pa=pa: ops.integration_bwd(d, sv, pk, clips, t, Ltok, i2t_dXnext=dx, t2i_p=pa, t2i_dXnext=dx)) for d, sv, dx, pa in zip(dRs, saved, dXn, pact)]
    tt = min(timeit_rot(fns, reps=3) for _ in range(3))
    print(f"{tag} integration_bwd (I2T + T2I behind): {tt*1e6:8.1f} us (includes the wrapper's output allocations)", flush=True)
