"""Fused IntegrationNetwork forward against the unfused sequence at the bench size: python tools/bench_integ.py [--check]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L
from tools.bench_tnet import timeit_rot
from tests.test_integ_gpu import make, reference, rel, CI, C4

clips, t, Ltok = 32, 8, 197
rows = clips * t * Ltok
w, Mp = make(clips, t, Ltok, seed=1)
wc = {k: v.cuda() for k, v in w.items()}
pk = ops.integration_pack(wc)
ln = tuple(wc[k].float().contiguous() for k in ("ln.weight", "ln.bias", "ln_temporal.weight", "ln_temporal.bias"))
NSET = 6
xs = [(torch.randn(rows, CI, device="cuda") * 1.5 + 0.3).to(torch.bfloat16) for _ in range(NSET)]
if "--check" in sys.argv:
    out = ops.integration_fwd(Mp.cuda(), pk, clips, t, Ltok, ln=ln)
    ref = reference(w, Mp, clips, t, Ltok)
    print("BM", os.environ.get("DIST_AMD_INTEG_BM", "128"), "R", rel(out["R"], ref["R"]), "h2", rel(out["zf_h2"][:, CI:], ref["h2"]), "Na", rel(out["Na"], ref["Na"]))
for train in (True, False):
    outs = [ops.integration_fwd(x, pk, clips, t, Ltok, ln=ln, train=train) for x in xs]
    fns = [(lambda x=x, o=o: ops.integration_fwd(x, pk, clips, t, Ltok, ln=ln, train=train, out=o)) for x, o in zip(xs, outs)]
    tt = timeit_rot(fns)
    fl = 2.0 * rows * (CI * (CI + C4) + 3 * C4 * C4 + (CI + C4) * CI)
    print(f"integration_fwd train={train}: {tt*1e6:8.1f} us  {fl/tt/1e12:7.1f} TF", flush=True)
