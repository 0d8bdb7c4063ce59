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
if "--t2i" in sys.argv:
    N = Ltok - 1
    Wt = (torch.randn(CI, C4, 2, 1, 1, device="cuda") * 0.07)
    bt = torch.randn(CI, device="cuda") * 0.1; cls = torch.randn(t, CI, device="cuda") * 0.5
    pk2 = ops.integration_pack(wc, t2i_w=Wt)
    Xps = [(torch.randn(clips * 2 * t * N, C4, device="cuda") * 0.8).to(torch.bfloat16) for _ in range(NSET)]
    outs = [ops.integration_fwd(x, pk2, clips, t, Ltok, xhat=True, t2i=(xp, bt, cls)) for x, xp in zip(xs, Xps)]
    fns = [(lambda x=x, xp=xp, o=o: ops.integration_fwd(x, pk2, clips, t, Ltok, xhat=True, t2i=(xp, bt, cls), out=o)) for x, xp, o in zip(xs, Xps, outs)]
    print(f"integration_fwd with T2I in front (xhat form): {timeit_rot(fns)*1e6:8.1f} us", flush=True)
    outs = [ops.integration_fwd(x, pk2, clips, t, Ltok, xhat=True) for x in xs]
    fns = [(lambda x=x, o=o: ops.integration_fwd(x, pk2, clips, t, Ltok, xhat=True, out=o)) for x, o in zip(xs, outs)]
    print(f"integration_fwd (xhat form): {timeit_rot(fns)*1e6:8.1f} us", flush=True)
    Wf = Wt.reshape(CI, C4, 2).permute(0, 2, 1).reshape(CI, 2 * C4).to(torch.bfloat16).contiguous()
    Mps = [torch.empty_like(x) for x in xs]
    def t2i(x, xp, mp):
        ops.gemm_nt(xp, Wf, clips * t * N, CI, C4, taps=2, bias=bt, res=x, amap=ops.rowmap(L.RM_STRIDED, 2, N), omap=ops.outmap(L.OM_INSERTCLS, N), C_out=mp)
    fns = [(lambda x=x, xp=xp, mp=mp: t2i(x, xp, mp)) for x, xp, mp in zip(xs, Xps, Mps)]
    print(f"T2I GEMM alone: {timeit_rot(fns)*1e6:8.1f} us", flush=True)
if "--bwd" in sys.argv:
    pkb = ops.integration_pack(wc, bwd=True)
    saved = [ops.integration_fwd(x, pkb, clips, t, Ltok, xhat=True) for x in xs]
    dRs = [(torch.randn(rows, CI, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(NSET)]
    from dist_amd import lib as LL
    import ctypes as C
    outs = [ops.integration_bwd(d, sv, pkb, clips, t, Ltok, copy=True) for d, sv in zip(dRs, saved)]
    def call(d, sv, o):
        a = LL.IntegBwdArgs()
        p_ = lambda x: x.data_ptr()
        a.dR, a.zf_h2, a.Xhat, a.rstd = p_(d), p_(sv["zf_h2"]), p_(sv["Xhat"]), p_(sv["rstd"])
        a.B1, a.B2, a.B3 = p_(pkb["B1"]), p_(pkb["B2"]), p_(pkb["B3"])
        a.dzf_dh2, a.dh1, a.dMp, a.dM_copy = p_(o["dzf_dh2"]), p_(o["dh1"]), p_(o["dMp"]), p_(o["dM"])
        a.add_dR, a.clips, a.t, a.L, a.Ci, a.C4, a.tk, a.dtype = 0, clips, t, Ltok, CI, C4, 3, LL.BF16
        LL.check(LL.load().dist_op_integration_bwd(C.byref(a), ops._stream()))
    fns = [(lambda d=d, sv=sv, o=o: call(d, sv, o)) for d, sv, o in zip(dRs, saved, outs)]
    print(f"integration_bwd: {timeit_rot(fns)*1e6:8.1f} us", flush=True)
