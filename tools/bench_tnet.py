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


if __name__ == "__main__":
    main()
