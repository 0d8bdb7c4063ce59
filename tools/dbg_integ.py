import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_integ_gpu import make, run, reference
for (clips, t, L) in [(2, 8, 40), (2, 8, 197)]:
    w, Mp = make(clips, t, L, seed=5)
    a = run(w, Mp, clips, t, L, train=True)
    b = run(w, Mp, clips, t, L, train=False)
    ref = reference(w, Mp, clips, t, L)["R"]
    ea = (a["R"].double().cpu() - ref).abs().max(1).values
    eb = (b["R"].double().cpu() - ref).abs().max(1).values
    d = (a["R"].float() - b["R"].float()).abs().sum(1).cpu()
    rows = torch.nonzero(d > 0).flatten().tolist()
    print(clips, t, L, "rows", rows)
    print(" err train ", [round(float(ea[r]), 4) for r in rows], "typical", float(ea.median()), float(ea.max()))
    print(" err infer ", [round(float(eb[r]), 4) for r in rows], "typical", float(eb.median()), float(eb.max()))
