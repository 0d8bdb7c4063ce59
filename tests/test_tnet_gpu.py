"""Fused TemporalNet forward (dist_op_temporal_net_fwd, csrc/tnet.hip; reference models/module_zoo/branches/dist.py:48-65) through the C ABI:

  * against a plain torch fp64 restatement of the reference expression with the kernel's bf16 rounding points (the same points
    oracle/dist_oracle.py `temporal_net` has) on random inputs: every output tensor, every geometry class - the bench plane 14 x 14,
    ViT-L/14's 16 x 16, odd planes (3 x 3, 5 x 5: ragged last m-tile, taps that leave the plane everywhere), one frame / two frames
    (every temporal tap at a border), clip counts that are not a multiple of the 8 XCDs, Ct = 32 / 64 / 96, 1- / 3- / 5-tap kernels;
  * against the unfused sequence the engine ran before (dist_op_layernorm -> dist_op_gemm_nt SHIFT -> dist_op_gemm_nt SPATIAL): the
    same rounding points, so the results agree to bf16 rounding of accumulation-order differences;
  * race / repeatability at the full bench size (b = 32, T = 16): two launches give the same bits, and rows of a clip do not depend
    on the clips around it.
"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _gaps import record  # noqa: E402
from _oracle_ops import temporal_net_oracle, ulp_ratio  # noqa: E402


ULP_GATE = 4.0      # measured 2.0 
#      # tests/_oracle_ops.ulp_ratio; measured values in profiles/r04_parity_gaps.json


def bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def qgelu(x):
    return x * torch.sigmoid(1.702 * x)


def reference(X, W1, b1, W2, b2, lnw, lnb, clips, T, G, tk):
    """fp64 on the bf16-valued inputs, rounding where the kernel stores bf16 (U, z, V = g(bf16 z), p, X' = g(bf16 p)): Oracle.temporal_net of the
    PINNED oracle (oracle/dist_oracle.py, reference dist.py:48-65) - no test-local restatement of the module"""
    k, _, _ = temporal_net_oracle(X, W1, b1, W2, b2, lnw, lnb, clips, T, G, tk)
    return {"U": k["tn_U"], "z": k["tn_z"], "V": k["tn_V"], "p": k["tn_p"], "Xp": k["tn_out"]}


def make(clips, T, G, Ct, tk, seed=0):
    g = torch.Generator().manual_seed(seed)
    rows = clips * T * G * G
    X = (torch.randn(rows, Ct, generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    W1 = torch.randn(Ct, Ct, tk, 1, 1, generator=g) * (0.8 / (Ct * tk) ** 0.5)
    W2 = torch.randn(Ct, Ct, 1, 3, 3, generator=g) * (0.8 / (Ct * 9) ** 0.5)
    b1, b2 = torch.randn(Ct, generator=g) * 0.1, torch.randn(Ct, generator=g) * 0.1
    lnw, lnb = 1 + 0.2 * torch.randn(Ct, generator=g), 0.1 * torch.randn(Ct, generator=g)
    return X, W1, b1, W2, b2, lnw, lnb


def run_fused(t, clips, T, G, tk, save_uv=True):
    from dist_amd import ops
    X, W1, b1, W2, b2, lnw, lnb = (v.cuda() for v in t)
    return ops.temporal_net_fwd(X, ops.pack_conv_taps(W1), b1, ops.pack_conv_taps(W2), b2, lnw, lnb, clips, T, G, tk=tk, save_uv=save_uv)


CASES = [  # clips, T, G, Ct, tk
    (2, 4, 14, 96, 3), (1, 8, 16, 96, 3), (3, 4, 3, 32, 3), (2, 1, 5, 32, 3), (9, 2, 4, 32, 3), (1, 3, 7, 64, 3),
    (1, 5, 6, 96, 5), (2, 2, 14, 96, 1), (10, 3, 2, 96, 3), (1, 6, 1, 32, 3),
]


@pytest.mark.parametrize("clips,T,G,Ct,tk", CASES)
def test_fused_temporal_net_vs_fp64_reference(gpu_lib, clips, T, G, Ct, tk):
    t = make(clips, T, G, Ct, tk, seed=clips * 100 + T * 10 + G)
    out = run_fused(t, clips, T, G, tk)
    torch.cuda.synchronize()
    ref = reference(*t, clips, T, G, tk)
    worst = 0.0
    for k in ("U", "z", "V", "p", "Xp"):
        got, want = out[k].double().cpu(), ref[k]
        # a result that lands on the other side of a bf16 rounding boundary differs by one bf16 ulp (2^-8 relative); downstream tensors
        # inherit a few of those through the convolutions
        err = float((got - want).abs().max() / (want.abs().max() + 1e-9))
        mean_err = float((got - want).abs().mean() / (want.abs().mean() + 1e-9))
        worst = max(worst, err)
        assert err < 1.2e-2 and mean_err < 6e-4, (k, err, mean_err)
    # element-wise beside the range-relative metric (a wrong small element must not pass): every element within a few bf16 ulps of its own size
    ur = max(ulp_ratio(out[k], ref[k]) for k in ("U", "z", "V", "p", "Xp"))
    record(f"tnet.fwd.ulp_ratio.{clips}x{T}x{G}x{Ct}x{tk}", ur)
    assert ur < ULP_GATE, ur
    # LayerNorm statistics
    x = t[0].double()
    mu = x.mean(1)
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt()
    torch.testing.assert_close(out["mean"].double().cpu(), mu, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out["rstd"].double().cpu(), rstd, rtol=1e-4, atol=1e-6)
    record(f"tnet.fwd.vs_fp64.{clips}x{T}x{G}x{Ct}x{tk}", worst)


@pytest.mark.parametrize("clips,T,G,Ct", [(2, 4, 14, 96), (1, 4, 16, 96), (3, 2, 3, 32)])
def test_fused_temporal_net_matches_the_unfused_sequence(gpu_lib, clips, T, G, Ct):
    from dist_amd import lib as L, ops
    t = make(clips, T, G, Ct, 3, seed=5)
    fused = run_fused(t, clips, T, G, 3)
    X, W1, b1, W2, b2, lnw, lnb = (v.cuda() for v in t)
    rows, N = X.shape[0], G * G
    U = torch.empty_like(X); z = torch.empty_like(X); V = torch.empty_like(X); p = torch.empty_like(X); Xp = torch.empty_like(X)
    ops.layernorm(X, lnw, lnb, y=U)
    ops.gemm_nt(U, ops.pack_conv_taps(W1), rows, Ct, Ct, taps=3, bias=b1, amap=ops.rowmap(L.RM_SHIFT, T * N, N, 1), C_out=z, C2_out=V)
    ops.gemm_nt(V, ops.pack_conv_taps(W2), rows, Ct, Ct, taps=9, bias=b2, res=X, amap=ops.rowmap(L.RM_SPATIAL, G, 0, 1), C_out=p, C2_out=Xp)
    torch.cuda.synchronize()
    for k, ref in (("U", U), ("z", z), ("V", V), ("p", p), ("Xp", Xp)):
        got, want = fused[k].float(), ref.float()
        frac = float(((got - want).abs() > 0).float().mean())               # differing elements: one-ulp flips from the summation order
        err = float((got - want).abs().max() / (want.abs().max() + 1e-9))
        # (U: the fused kernel takes the variance as E[x^2] - mean^2 from packed-bf16 dot products, dist_op_layernorm in two passes:
        # a value next to a bf16 rounding boundary may land on the other side)
        assert err < 1.2e-2 and frac < (2e-3 if k == "U" else 0.08), (k, err, frac)


def test_fused_temporal_net_full_size_is_repeatable_and_clip_local(gpu_lib):
    """the bench size (b = 32, T = 16, 14 x 14, Ct = 96 -> 512 workgroups, two per CU): no races between the LDS phases (two launches give
    the same bits), and a clip's rows are the same bits when it is processed alone"""
    clips, T, G, Ct = 32, 16, 14, 96
    t = make(clips, T, G, Ct, 3, seed=11)
    a = run_fused(t, clips, T, G, 3)
    b = run_fused(t, clips, T, G, 3)
    torch.cuda.synchronize()
    for k in ("U", "z", "V", "p", "Xp", "mean", "rstd"):
        assert torch.equal(a[k], b[k]), k
    rows1 = T * G * G
    for clip in (0, 13, 31):
        t1 = (t[0][clip * rows1:(clip + 1) * rows1].contiguous(),) + t[1:]
        one = run_fused(t1, 1, T, G, 3, save_uv=False)
        for k in ("z", "p", "Xp"):
            assert torch.equal(one[k], a[k][clip * rows1:(clip + 1) * rows1]), (clip, k)
    # without the optional outputs the required ones are the same bits
    c = run_fused(t, clips, T, G, 3, save_uv=False)
    assert "U" not in c and all(torch.equal(c[k], a[k]) for k in ("z", "p", "Xp"))
    ref = reference(*[v[:rows1] if i == 0 else v for i, v in enumerate(t)], 1, T, G, 3)
    err = float((a["Xp"][:rows1].double().cpu() - ref["Xp"]).abs().max() / ref["Xp"].abs().max())
    assert err < 1.2e-2, err


# ---------------------------------------------------------------------------------------------------------------------------------
# backward: dist_op_temporal_net_bwd (dz = conv3x3^T(dp) * g'(z); dX = dp + LN'(conv_t^T(dz)); dgamma, dbeta)
# ---------------------------------------------------------------------------------------------------------------------------------
def qgelu_grad(x):
    s_ = torch.sigmoid(1.702 * x)
    return s_ * (1 + 1.702 * x * (1 - s_))


def reference_bwd(dp, z, X, W1, W2, lnw, lnb, clips, T, G, tk, b1=None, b2=None):
    """fp64 autograd through the pinned oracle's temporal_net with the rounding points of the backward kernels (fused=True: dz is rounded to bf16
    where tnet_bwd_spatial_kernel stores it): dp is the gradient at the pre-activation p, so the loss is sum(p * dp)"""
    Ct = X.shape[1]
    zero = torch.zeros(Ct)
    k, o, x = temporal_net_oracle(X, W1, zero if b1 is None else b1, W2, zero if b2 is None else b2, lnw, lnb, clips, T, G, tk, fused=True, requires_grad=True)
    pre = "dist_net.temporal_nets.0."
    # (tn_p is the rounded pre-activation in front of its gradient hook; tn_z the rounded z in front of its hook: its gradient is the ROUNDED dz)
    raw = k["_raw"]
    dz, dx, dg, db = torch.autograd.grad((raw["tn_p"] * dp.double().reshape(raw["tn_p"].shape)).sum(), (raw["tn_z"], x, o.p[pre + "ln.weight"], o.p[pre + "ln.bias"]))
    bfr = lambda v: v.to(torch.bfloat16).to(torch.float64)
    return {"dz": dz.reshape(-1, Ct), "dX": bfr(dx.reshape(-1, Ct)), "dgamma": dg, "dbeta": db}


@pytest.mark.parametrize("clips,T,G,Ct,tk", CASES)
def test_fused_temporal_net_backward_vs_fp64_reference(gpu_lib, clips, T, G, Ct, tk):
    from dist_amd import ops
    t = make(clips, T, G, Ct, tk, seed=clips * 100 + T * 10 + G + 1)
    X, W1, b1, W2, b2, lnw, lnb = t
    fwd = run_fused(t, clips, T, G, tk, save_uv=False)
    gen = torch.Generator().manual_seed(99)
    dp = (torch.randn(X.shape, generator=gen) * 0.5).to(torch.bfloat16)
    z = fwd["z"].cpu()
    pre = torch.full((Ct,), 0.25)
    out = ops.temporal_net_bwd(dp.cuda(), fwd["z"], X.cuda(), fwd["mean"], fwd["rstd"], lnw.cuda(), ops.pack_conv_taps_dgrad(W1).cuda(),
                               ops.pack_conv_taps_dgrad(W2).cuda(), clips, T, G, tk=tk, dgamma=pre.clone().cuda(), dbeta=pre.clone().cuda())
    torch.cuda.synchronize()
    ref = reference_bwd(dp, z, X, W1, W2, lnw, lnb, clips, T, G, tk, b1, b2)
    worst = 0.0
    for k in ("dz", "dX"):
        got, want = out[k].double().cpu(), ref[k]
        err = float((got - want).abs().max() / (want.abs().max() + 1e-9))
        mean_err = float((got - want).abs().mean() / (want.abs().mean() + 1e-9))
        worst = max(worst, err)
        assert err < 1.2e-2 and mean_err < 1e-3, (k, err, mean_err)
    for k in ("dgamma", "dbeta"):                    # accumulated INTO the given buffers (pre-filled with 0.25), fp32 sums over every row
        got, want = out[k].double().cpu() - 0.25, ref[k]
        err = float((got - want).abs().max() / (want.abs().max() + 1e-9))
        assert err < 5e-3, (k, err)
    record(f"tnet.bwd.vs_fp64.{clips}x{T}x{G}x{Ct}x{tk}", worst)


def test_fused_temporal_net_backward_matches_the_unfused_sequence(gpu_lib):
    """against the kernels the engine ran before: dist_op_gemm_nt (SPATIAL, sign -1, MULG z) -> dist_op_gemm_nt (SHIFT, sign -1) ->
    dist_op_layernorm_bwd (dx_add = dp), at the bench plane; and bit-repeatable (no atomics)"""
    from dist_amd import lib as L, ops
    clips, T, G, Ct = 3, 4, 14, 96
    t = make(clips, T, G, Ct, 3, seed=21)
    X, W1, b1, W2, b2, lnw, lnb = (v.cuda() for v in t)
    fwd = run_fused(t, clips, T, G, 3, save_uv=False)
    rows, N = X.shape[0], G * G
    dp = (torch.randn(rows, Ct, device="cuda") * 0.5).to(torch.bfloat16)
    W1b, W2b = ops.pack_conv_taps_dgrad(W1), ops.pack_conv_taps_dgrad(W2)
    a = ops.temporal_net_bwd(dp, fwd["z"], X, fwd["mean"], fwd["rstd"], lnw, W1b, W2b, clips, T, G)
    b = ops.temporal_net_bwd(dp, fwd["z"], X, fwd["mean"], fwd["rstd"], lnw, W1b, W2b, clips, T, G)
    for k in ("dz", "dX", "dgamma", "dbeta"):
        assert torch.equal(a[k], b[k]), k
    dz = torch.empty_like(X); dU = torch.empty_like(X); dX = torch.empty_like(X)
    ops.gemm_nt(dp, W2b, rows, Ct, Ct, taps=9, aux=fwd["z"], amap=ops.rowmap(L.RM_SPATIAL, G, 0, -1), C_out=dz)
    ops.gemm_nt(dz, W1b, rows, Ct, Ct, taps=3, amap=ops.rowmap(L.RM_SHIFT, T * N, N, -1), C_out=dU)
    dg, db = torch.zeros(Ct, device="cuda"), torch.zeros(Ct, device="cuda")
    ops.layernorm_bwd(X, fwd["mean"], fwd["rstd"], dU, lnw, dx=dX, dw=dg, db=db, dx_add=dp)
    torch.cuda.synchronize()
    for k, ref in (("dz", dz), ("dX", dX)):
        got, want = a[k].float(), ref.float()
        err = float((got - want).abs().max() / (want.abs().max() + 1e-9))
        frac = float(((got - want).abs() > 0).float().mean())
        assert err < 1.5e-2 and frac < 0.25, (k, err, frac)       # (the unfused path rounds dU to bf16 between its kernels, the fused one does not: measured 14 % one-ulp differences in dX)
    for k, ref in (("dgamma", dg), ("dbeta", db)):
        err = float((a[k] - ref).abs().max() / (ref.abs().max() + 1e-9))
        assert err < 8e-3, (k, err)


def test_fused_temporal_net_backward_phases_and_the_multi_layer_reduce(gpu_lib):
    """phase 1 (dz) + phase 2 (dX, dgamma, dbeta) == phase 0 bit for bit; phase 3 leaves the partial rows in the layer's scratch slice
    and ONE dist_op_temporal_net_bwd_reduce over two layers gives the same sums as the per-layer form, accumulated into the buffers"""
    from dist_amd import lib as L, ops
    clips, T, G, Ct = 2, 4, 14, 96
    lib = L.load()
    n = lib.dist_op_temporal_net_bwd_scratch(clips, T, Ct)
    scratch = torch.zeros(2 * n, dtype=torch.float32, device="cuda")
    whole, dgs, dbs = [], [], []
    for layer in range(2):
        t = make(clips, T, G, Ct, 3, seed=40 + layer)
        X, W1, b1, W2, b2, lnw, lnb = (v.cuda() for v in t)
        fwd = run_fused(t, clips, T, G, 3, save_uv=False)
        dp = (torch.randn(X.shape, device="cuda") * 0.5).to(torch.bfloat16)
        W1b, W2b = ops.pack_conv_taps_dgrad(W1), ops.pack_conv_taps_dgrad(W2)
        args = (dp, fwd["z"], X, fwd["mean"], fwd["rstd"], lnw, W1b, W2b, clips, T, G)
        a = ops.temporal_net_bwd(*args)
        p1 = ops.temporal_net_bwd(*args, phase=1)
        assert torch.equal(p1["dz"], a["dz"])
        p2 = ops.temporal_net_bwd(*args, phase=2, dz=p1["dz"])
        for k in ("dX", "dgamma", "dbeta"):
            assert torch.equal(p2[k], a[k]), k
        p3 = ops.temporal_net_bwd(*args, phase=3, dz=p1["dz"], scratch=scratch[layer * n:(layer + 1) * n])
        assert torch.equal(p3["dX"], a["dX"])
        assert float(p3["dgamma"].abs().max()) == 0.0 and float(p3["dbeta"].abs().max()) == 0.0      # not summed yet
        whole.append(a)
        dgs.append(torch.full((Ct,), 0.5, device="cuda")); dbs.append(torch.full((Ct,), -0.5, device="cuda"))
    ops.temporal_net_bwd_reduce(scratch, 2, clips, T, Ct, dgs, dbs)
    torch.cuda.synchronize()
    for layer in range(2):
        assert torch.equal(dgs[layer], whole[layer]["dgamma"] + 0.5), layer
        assert torch.equal(dbs[layer], whole[layer]["dbeta"] - 0.5), layer
