"""Fused TemporalNet forward (dist_op_temporal_net_fwd) against the unfused sequence it replaces (LayerNorm + conv_t GEMM + conv3x3 GEMM),
cold operands (every call works on another buffer set, > the 256 MB MALL in total), at the BASELINE geometries."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L


def timeit_rot(fns, reps=5):
    for f in fns: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for f in fns: f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * len(fns)) * 1e-3


def main():
    dt = torch.bfloat16
    NSET = 8
    for (clips, T, G, tag) in [(32, 16, 14, "config 2: b=32 T=16 14x14"), (32, 32, 14, "config 3: b=32 T=32 14x14"), (8, 64, 16, "config 4: b=8 T=64 16x16")]:
        Ct, N = 96, G * G
        rows = clips * T * N
        Xs = [(torch.randn(rows, Ct, device="cuda") * 1.3).to(dt) for _ in range(NSET)]
        W1 = ops.pack_conv_taps(torch.randn(Ct, Ct, 3, 1, 1, device="cuda") * 0.05); W2 = ops.pack_conv_taps(torch.randn(Ct, Ct, 1, 3, 3, device="cuda") * 0.03)
        b1, b2, lw, lb = (torch.randn(Ct, device="cuda") * 0.1 for _ in range(4))
        bufs = [[torch.empty(rows, Ct, device="cuda", dtype=dt) for _ in range(5)] for _ in range(NSET)]
        mean, rstd = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")

        def fused(i, uv):
            a = L.TnetArgs()
            U, z, V, p, Xp = bufs[i]
            a.X, a.W1, a.W2, a.b1, a.b2, a.ln_w, a.ln_b = (ops._p(v) for v in (Xs[i], W1, W2, b1, b2, lw, lb))
            a.z, a.p, a.Xp, a.U, a.V = ops._p(z), ops._p(p), ops._p(Xp), ops._p(U if uv else None), ops._p(V if uv else None)
            a.mean, a.rstd = ops._p(mean), ops._p(rstd)
            a.clips, a.T, a.G, a.Ct, a.tk, a.dtype, a.eps = clips, T, G, Ct, 3, L.BF16, 1e-5
            L.check(L.load().dist_op_temporal_net_fwd(a, ops._stream()))

        def unfused(i):
            U, z, V, p, Xp = bufs[i]
            ops.layernorm(Xs[i], lw, lb, y=U, mean=mean, rstd=rstd)
            ops.gemm_nt(U, W1, rows, Ct, Ct, taps=3, bias=b1, amap=ops.rowmap(L.RM_SHIFT, T * N, N, 1), C_out=z, C2_out=V)
            ops.gemm_nt(V, W2, rows, Ct, Ct, taps=9, bias=b2, res=Xs[i], amap=ops.rowmap(L.RM_SPATIAL, G, 0, 1), C_out=p, C2_out=Xp)

        flops = 2.0 * rows * Ct * Ct * 12
        for name, fn, nbytes in (("unfused (3 launches)", lambda i: unfused(i), 6 * rows * Ct * 2 + 3 * rows * Ct * 2),
                                 ("fused, U and V written", lambda i: fused(i, True), 6 * rows * Ct * 2),
                                 ("fused, z p X' only", lambda i: fused(i, False), 4 * rows * Ct * 2)):
            t = timeit_rot([(lambda i=i: fn(i)) for i in range(NSET)])
            print(f"{tag}: {name:24s} {t * 1e6:8.1f} us  {flops / t / 1e12:6.1f} TF  algorithmic bytes {nbytes / 1e6:6.1f} MB -> {nbytes / t / 1e9:6.0f} GB/s", flush=True)
        del Xs, bufs


if __name__ == "__main__" and "--bwd" not in sys.argv:
    main()


def bench_bwd():
    """the data-gradient backward: fused (two launches + reduce) vs the unfused sequence (two row-mapped GEMMs + LayerNorm backward)"""
    dt = torch.bfloat16
    NSET = 6
    for (clips, T, G, tag) in [(32, 16, 14, "config 2: b=32 T=16 14x14"), (8, 64, 16, "config 4: b=8 T=64 16x16")]:
        Ct, N = 96, G * G
        rows = clips * T * N
        mk = lambda: [(torch.randn(rows, Ct, device="cuda") * 0.7).to(dt) for _ in range(NSET)]
        Xs, zs, dps = mk(), mk(), mk()
        W1 = torch.randn(Ct, Ct, 3, 1, 1, device="cuda") * 0.05; W2 = torch.randn(Ct, Ct, 1, 3, 3, device="cuda") * 0.03
        W1b, W2b = ops.pack_conv_taps_dgrad(W1), ops.pack_conv_taps_dgrad(W2)
        lw = torch.randn(Ct, device="cuda") * 0.1 + 1
        mean, rstd = torch.randn(rows, device="cuda") * 0.1, torch.rand(rows, device="cuda") + 0.5
        dzs, dUs, dXs = [torch.empty(rows, Ct, device="cuda", dtype=dt) for _ in range(NSET)], [torch.empty(rows, Ct, device="cuda", dtype=dt) for _ in range(2)], \
            [torch.empty(rows, Ct, device="cuda", dtype=dt) for _ in range(NSET)]
        dg, db = torch.zeros(Ct, device="cuda"), torch.zeros(Ct, device="cuda")
        n = L.load().dist_op_temporal_net_bwd_scratch(clips, T, Ct)
        scratch = torch.empty(n, device="cuda")

        def fused(i, phase=0):
            a = L.TnetBwdArgs()
            a.dp, a.z, a.X, a.mean, a.rstd, a.ln_w, a.W1b, a.W2b = (ops._p(v) for v in (dps[i], zs[i], Xs[i], mean, rstd, lw, W1b, W2b))
            a.dz, a.dX, a.dgamma, a.dbeta, a.scratch, a.scratch_elems = ops._p(dzs[i]), ops._p(dXs[i]), ops._p(dg), ops._p(db), ops._p(scratch), n
            a.clips, a.T, a.G, a.Ct, a.tk, a.dtype, a.phase = clips, T, G, Ct, 3, L.BF16, phase
            L.check(L.load().dist_op_temporal_net_bwd(a, ops._stream()))

        def unfused(i, part=0):
            if part in (0, 1):
                ops.gemm_nt(dps[i], W2b, rows, Ct, Ct, taps=9, aux=zs[i], amap=ops.rowmap(L.RM_SPATIAL, G, 0, -1), C_out=dzs[i])
            if part in (0, 2):
                ops.gemm_nt(dzs[i], W1b, rows, Ct, Ct, taps=3, amap=ops.rowmap(L.RM_SHIFT, T * N, N, -1), C_out=dUs[i % 2])
                ops.layernorm_bwd(Xs[i], mean, rstd, dUs[i % 2], lw, dx=dXs[i], dw=dg, db=db, dx_add=dps[i])
        for name, fn in (("unfused: conv3x3^T * g'", lambda i: unfused(i, 1)), ("unfused: conv_t^T + LN bwd", lambda i: unfused(i, 2)),
                         ("fused phase 1 (dz)", lambda i: fused(i, 1)), ("fused phase 2 (dX, dg, db)", lambda i: fused(i, 2)), ("fused, all", lambda i: fused(i, 0))):
            t = timeit_rot([(lambda i=i: fn(i)) for i in range(NSET)])
            print(f"{tag}: {name:28s} {t * 1e6:8.1f} us", flush=True)


if __name__ == "__main__" and "--bwd" in sys.argv:
    bench_bwd()
