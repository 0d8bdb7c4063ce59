"""The fixed part of a gemm_fast8p tile: time per launch over K (linear fit: rounds x (fixed + K-tiles x c)) for the ViT's in_proj / c_fc / out_proj
epilogues; run once per DIST_AMD_FAST_DBG value (timing-only library: 1 no output stores, 2 no epilogue).  python tools/bench_gemm_fixed.py <tag>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from check_pp import timeit_rot


def main(tag):
    from dist_amd import ops, lib as L
    dt = torch.bfloat16
    M = 50432
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    rnd = lambda shape, scale=1.0, dtype=dt: (torch.randn(shape, device="cuda", generator=g) * scale).to(dtype)
    for (N, kind) in ((2304, "lnfold_heads"), (3072, "lnfold_act"), (768, "res_rowstats"), (2304, "plain")):
        pts = []
        for K in (128, 256, 512, 768, 1536):
            As = [rnd((M, K)) for _ in range(4)]
            W = rnd((N, K), K ** -0.5)
            bias = rnd((N,), 1.0, torch.float32)
            stats = torch.stack([rnd((M,), 0.1, torch.float32), rnd((M,), 0.2, torch.float32).abs() + 0.5]).contiguous()
            colsum = W.float().sum(1).contiguous()
            res = rnd((M, N))
            outs = [torch.empty((M, N), device="cuda", dtype=dt) for _ in range(4)]
            rs = torch.zeros(N // 64, M, 2, device="cuda", dtype=torch.float32) if kind == "res_rowstats" else None
            def call(a, c):
                kw = dict(bias=bias)
                if "lnfold" in kind: kw["lnfold"] = (stats, colsum)
                if "act" in kind: kw["C2_out"] = c
                else: kw["C_out"] = c
                if kind.startswith("res"): kw["res"] = res
                if rs is not None: kw["rowstats"] = rs
                if "heads" in kind: kw["omap"] = ops.outmap(L.OM_HEADS, 197, N // 192); kw["ldc"] = 64
                ops.gemm_nt(a, W, M, N, K, **kw)
            t = timeit_rot([(lambda a=a, c=c: call(a, c)) for a, c in zip(As, outs)]) * 1e6
            pts.append((K // 64, t))
            del As, outs
        rounds = -(-(197 * (N // 256)) // 256)
        import numpy as np
        x = np.array([p[0] for p in pts], float); y = np.array([p[1] for p in pts], float) / rounds
        c, f = np.polyfit(x, y, 1)
        print(f"[{tag}] N={N:5d} {kind:13s} " + " ".join(f"K{int(k)*64}:{t:7.1f}" for k, t in pts) + f" | rounds {rounds}: fixed {f:5.2f} us + {c:5.3f} us per K-tile", flush=True)


if __name__ == "__main__":
    main(sys.argv[1])
