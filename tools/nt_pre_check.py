"""Bitwise A/B of the branch GEMM's epilogue: with and without the LDS-DMA prefetch of the residual / derivative tile
(DIST_AMD_NT_ROTATE=0 / 2), full-size shapes, every epilogue combination, repeated (race screen).
usage: python tools/nt_pre_check.py            (parent: spawns the two children and compares)"""
import sys, os, subprocess, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def child():
    import torch
    from dist_amd import ops, lib as L
    dt = torch.bfloat16
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    def rnd(*s): return torch.randn(*s, device="cuda", generator=g).to(dt)
    shapes = [(100352, 96, 96, 9, L.RM_SPATIAL, (14, 0)), (100352 - 37, 96, 96, 3, L.RM_SHIFT, (16 * 196, 196)), (50432, 384, 768, 1, 0, (0, 0)),
              (50432 - 5, 384, 96, 1, 0, (0, 0)), (50432, 96, 384, 1, 0, (0, 0)), (3136, 384, 96, 1, 0, (0, 0)), (6272, 96, 96, 9, L.RM_SPATIAL, (14, 0))]
    for (M, N, K, taps, mode, kw) in shapes:
        A = rnd(M, K); W = (rnd(N, taps * K).float() * (taps * K) ** -0.5).to(dt); bias = torch.randn(N, device="cuda", generator=g)
        R = rnd(M, N); X = rnd(M, N)
        am = ops.rowmap(mode, kw[0], kw[1])
        for combo in ["res", "res_inplace", "res+act", "aux", "aux+res", "aux_post+res", "none", "act"]:
            hs = set()
            for rep in range(4):
                C = torch.full((M + 8, N), 7.0, device="cuda", dtype=dt); C2 = torch.full((M + 8, N), 7.0, device="cuda", dtype=dt) if "act" in combo else None
                res = R if "res" in combo else None
                if combo == "res_inplace": C[:M].copy_(R); res = C
                aux = X if "aux" in combo else None
                ops.gemm_nt(A, W, M, N, K, taps=taps, bias=bias, res=res, aux=aux, C_out=C, C2_out=C2, amap=am, mulg_post="post" in combo)
                torch.cuda.synchronize()
                h = hashlib.md5(C.view(torch.int16).cpu().numpy().tobytes())
                if C2 is not None: h.update(C2.view(torch.int16).cpu().numpy().tobytes())
                hs.add(h.hexdigest())
            print(f"{M}x{N}x{K}x{taps} {combo} {'/'.join(sorted(hs))}", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child()
else:
    outs = []
    for r in ("2", "0"):
        env = dict(os.environ, DIST_AMD_NT_ROTATE=r)
        outs.append(subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().splitlines())
    bad = 0
    for a, b in zip(*outs):
        ok = a == b and "/" not in a.split()[-1]
        bad += not ok
        print(("ok   " if ok else "DIFF ") + a + ("" if ok else "   |   " + b))
    print("lines", len(outs[0]), len(outs[1]), "bad", bad)
