"""A few launches of the fused IntegrationNetwork forward / backward in the engine's form at the bench size, for rocprofv3 passes (tools/r06_pmc_integ4.sh);
the tile form follows DIST_AMD_INTEG_W4 (see tools/bench_integ4.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
from tests.test_integ_gpu import make, CI, C4
clips, t, Ltok = 32, 8, 197
rows, N = clips * t * Ltok, Ltok - 1
w, _ = make(1, 8, 16, seed=1)
wc = {k: v.cuda() for k, v in w.items()}
Wt = torch.randn(CI, C4, 2, 1, 1, device="cuda") * 0.07
Wi = torch.randn(C4, CI, device="cuda") * CI ** -0.5
bt = torch.randn(CI, device="cuda") * 0.1; cls = torch.randn(t, CI, device="cuda") * 0.5; bi = torch.randn(C4, device="cuda") * 0.1
pk = ops.integration_pack(wc, bwd=True, t2i_w=Wt, i2t_w=Wi)
xs = [(torch.randn(rows, CI, device="cuda") * 1.5 + 0.3).to(torch.bfloat16) for _ in range(4)]
Xps = [(torch.randn(clips * 2 * t * N, C4, device="cuda") * 0.8).to(torch.bfloat16) for _ in range(4)]
dRs = [(torch.randn(rows, CI, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(4)]
outs = [ops.integration_fwd(x, pk, clips, t, Ltok, xhat=True, t2i=(xp, bt, cls), i2t_bias=bi) for x, xp in zip(xs, Xps)]
for _ in range(3):
    for x, xp, o, d in zip(xs, Xps, outs, dRs):
        ops.integration_fwd(x, pk, clips, t, Ltok, xhat=True, t2i=(xp, bt, cls), i2t_bias=bi, out=o)
        ops.integration_bwd(d, o, pk, clips, t, Ltok, i2t_dXnext=xp, t2i_p=xp, t2i_dXnext=xp)
torch.cuda.synchronize()
print("done")
