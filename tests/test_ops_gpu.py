"""Per-kernel parity tests (GPU): every HIP operator behind the C ABI vs a plain
PyTorch fp32/fp64 statement of the same op on the same seeded inputs.

fp32 mode (exact-f32 MFMA) is held to tight tolerances; bf16 mode is held to one
bf16 rounding of the output (inputs are identical bf16 values, accumulation is fp32).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def tol(dtype, scale=1.0):
    return dict(rtol=2e-5, atol=2e-5 * scale) if dtype == torch.float32 else dict(rtol=1.2e-2, atol=1.2e-2 * scale)


def rnd(shape, dtype, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(*shape, generator=g, device="cuda") * scale).to(dtype)


def qgelu(x):
    return x * torch.sigmoid(1.702 * x)


def qgelu_grad(x):
    s = torch.sigmoid(1.702 * x)
    return s * (1 + 1.702 * x * (1 - s))


# ---- python statement of the row maps (include/dist_amd.h) ------------------------------
def src_rows(mode, M, tap, taps, p0=0, p1=0, sign=1):
    m = torch.arange(M, device="cuda")
    if mode == "plain":
        return m, torch.ones_like(m, dtype=torch.bool)
    if mode == "shift":
        off = sign * (tap - taps // 2) * p1
        r = m % p0 + off
        return m + off, (r >= 0) & (r < p0)
    if mode == "spatial":
        g = p0
        dy, dx = (tap // 3 - 1) * sign, (tap % 3 - 1) * sign
        n = m % (g * g)
        y, x = n // g + dy, n % g + dx
        return m + dy * g + dx, (y >= 0) & (y < g) & (x >= 0) & (x < g)
    if mode == "strided":
        bj, n = m // p1, m % p1
        return (bj * p0 + tap) * p1 + n, torch.ones_like(m, dtype=torch.bool)
    if mode == "skipcls":
        bj, n = m // p0, m % p0
        return bj * (p0 + 1) + 1 + n, torch.ones_like(m, dtype=torch.bool)
    raise KeyError(mode)


MODES = {"plain": 0, "shift": 1, "spatial": 2, "strided": 3, "skipcls": 4}


def gather(A, mode, M, tap, taps, **kw):
    src, ok = src_rows(mode, M, tap, taps, **kw)
    out = A.double()[src.clamp(0, A.shape[0] - 1)]
    return out * ok[:, None]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("M,N,K", [(1100, 768, 768), (2048, 256, 64), (1024, 512, 3072), (1500, 448, 160), (1280, 256, 128), (4096, 2304, 768),
                                   (3152, 768, 768), (591, 2304, 768), (1000, 96, 96), (777, 384, 3072), (130, 1152, 384), (64, 512, 384), (257, 192, 96)])
def test_gemm_nt_plain(gpu_lib, dtype, M, N, K):
    from dist_amd import ops
    A, B = rnd((M, K), dtype, 1), rnd((N, K), dtype, 2, K ** -0.5)
    bias = rnd((N,), torch.float32, 3)
    res = rnd((M, N), dtype, 4)
    C, C2 = torch.empty(M, N, dtype=dtype, device="cuda"), torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, bias=bias, res=res, C_out=C, C2_out=C2)
    ref = A.double() @ B.double().t() + bias.double() + res.double()
    torch.testing.assert_close(C.double(), ref, **tol(dtype))
    torch.testing.assert_close(C2.double(), qgelu(C.double()), **tol(dtype))
    # activation-only output (no pre-activation kept) and MULG epilogue
    C3 = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, bias=bias, C2_out=C3)
    torch.testing.assert_close(C3.double(), qgelu(A.double() @ B.double().t() + bias.double()), **tol(dtype))
    aux = rnd((M, N), dtype, 5)
    C4 = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, aux=aux, C_out=C4)
    torch.testing.assert_close(C4.double(), (A.double() @ B.double().t()) * qgelu_grad(aux.double()), **tol(dtype))
    # derivative factor applied after bias and residual (backward through X' = g(p) fused into the producing GEMM)
    C5 = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, bias=bias, res=res, aux=aux, C_out=C5, mulg_post=True)
    torch.testing.assert_close(C5.double(), (A.double() @ B.double().t() + bias.double() + res.double()) * qgelu_grad(aux.double()), **tol(dtype))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("mode,kw,taps,N", [
    ("shift", dict(p0=8 * 9, p1=9), 3, 96), ("shift", dict(p0=8 * 9, p1=9, sign=-1), 3, 128), ("shift", dict(p0=6 * 16, p1=16), 5, 32),
    ("spatial", dict(p0=4), 9, 96), ("spatial", dict(p0=3, sign=-1), 9, 32), ("spatial", dict(p0=14), 9, 96),
    ("strided", dict(p0=2, p1=9), 2, 384), ("skipcls", dict(p0=9), 1, 96),
    # large enough for the LDS-DMA kernels (gemm_nt2.hip)
    ("shift", dict(p0=16 * 49, p1=49), 3, 96), ("shift", dict(p0=8 * 197, p1=197, sign=-1), 3, 128), ("shift", dict(p0=16 * 49, p1=49), 5, 96),
    ("spatial", dict(p0=14, sign=-1), 9, 96), ("strided", dict(p0=2, p1=196), 2, 384), ("skipcls", dict(p0=196), 1, 96)])
def test_gemm_nt_rowmaps(gpu_lib, dtype, mode, kw, taps, N):
    from dist_amd import ops, lib as L
    K = 96
    if mode == "shift":
        M = kw["p0"] * 5
    elif mode == "spatial":
        M = kw["p0"] ** 2 * 7
    else:
        M = kw.get("p1", kw["p0"]) * 11 if mode == "strided" else kw["p0"] * 11
    nn = kw.get("p1", 0) if mode == "strided" else kw["p0"]
    rowsA = {"strided": M * 2, "skipcls": (M // nn) * (nn + 1) if mode == "skipcls" else M}.get(mode, M)
    A = rnd((rowsA, K), dtype, 1)
    B = rnd((N, taps * K), dtype, 2, (taps * K) ** -0.5)
    bias = rnd((N,), torch.float32, 3)
    C = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, taps=taps, bias=bias, C_out=C,
                amap=ops.rowmap(MODES[mode], kw.get("p0", 0), kw.get("p1", 0), kw.get("sign", 1)))
    ref = bias.double().expand(M, N).clone()
    for tap in range(taps):
        ref += gather(A, mode, M, tap, taps, **kw) @ B.double()[:, tap * K:(tap + 1) * K].t()
    torch.testing.assert_close(C.double(), ref, **tol(dtype))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("grid,frames,sign", [(14, 5, 1), (14, 3, -1), (16, 3, 1)])
def test_conv3x3_frame_epilogues(gpu_lib, dtype, grid, frames, sign):
    """3x3 frame convolution, 96 channels (the spatial row map of dist_op_gemm_nt, 9 taps; bf16: the LDS-DMA kernel) with
    every epilogue the TemporalNet uses: bias + residual + second activated output (forward), activation derivative (backward)."""
    from dist_amd import ops
    K = N = 96
    M = grid * grid * frames
    A = rnd((M, K), dtype, 1)
    B = rnd((N, 9 * K), dtype, 2, (9 * K) ** -0.5)
    bias = rnd((N,), torch.float32, 3)
    res, aux = rnd((M, N), dtype, 4), rnd((M, N), dtype, 5)
    amap = ops.rowmap(MODES["spatial"], grid, 0, sign)
    conv = torch.zeros(M, N, dtype=torch.float64, device="cuda")
    for tap in range(9):
        conv += gather(A, "spatial", M, tap, 9, p0=grid, sign=sign) @ B.double()[:, tap * K:(tap + 1) * K].t()
    C, C2 = torch.empty(M, N, dtype=dtype, device="cuda"), torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, taps=9, bias=bias, res=res, C_out=C, C2_out=C2, amap=amap)
    torch.testing.assert_close(C.double(), conv + bias.double() + res.double(), **tol(dtype))
    torch.testing.assert_close(C2.double(), qgelu(C.double()), **tol(dtype))
    C3 = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, taps=9, bias=bias, C2_out=C3, amap=amap)
    torch.testing.assert_close(C3.double(), qgelu(conv + bias.double()), **tol(dtype))
    C4 = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, taps=9, aux=aux, C_out=C4, amap=amap)
    torch.testing.assert_close(C4.double(), conv * qgelu_grad(aux.double()), **tol(dtype))


@pytest.mark.parametrize("M,N,K", [(2048, 256, 192), (1536, 512, 320), (1300, 384, 448), (4096, 768, 1536), (2304, 1536, 384)])
def test_gemm_fast_two_group_loop_tile_counts(gpu_lib, M, N, K):
    """256x256x64 two-group main loop (gemm_fast8p_kernel): odd and even K-tile counts (3, 5, 7, 24, 6) walk the
    scalar-predicated tail (counted vmcnt 8 / 4 / 2 / 0), a half-empty column tile (N = 384) and ragged M."""
    from dist_amd import ops
    dtype = torch.bfloat16
    A, B = rnd((M, K), dtype, 11), rnd((N, K), dtype, 12, K ** -0.5)
    bias, res = rnd((N,), torch.float32, 13), rnd((M, N), dtype, 14)
    C = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, bias=bias, res=res, C_out=C)
    torch.testing.assert_close(C.double(), A.double() @ B.double().t() + bias.double() + res.double(), **tol(dtype))


@pytest.mark.parametrize("N,K", [(2304, 768), (768, 3072)])
def test_gemm_fast_two_group_loop_is_race_free_at_full_size(gpu_lib, N, K):
    """LDS-DMA data is ordered for a ds_read only by the issuing wave's counted vmcnt plus a barrier: a misplaced wait shows as
    RARE wrong tiles under memory load.  Full BASELINE row count (50 432 token rows), every block of the chip busy, 12 launches:
    all bit-identical to each other and equal to the fp32 reference."""
    from dist_amd import ops
    dtype = torch.bfloat16
    M = 50432
    A, B = rnd((M, K), dtype, 21), rnd((N, K), dtype, 22, K ** -0.5)
    bias = rnd((N,), torch.float32, 23)
    ref = (A.float() @ B.float().t() + bias).to(dtype)
    outs = []
    for _ in range(12):
        C = torch.empty(M, N, dtype=dtype, device="cuda")
        ops.gemm_nt(A, B, M, N, K, bias=bias, C_out=C)
        outs.append(C)
    torch.cuda.synchronize()
    for C in outs[1:]:
        assert torch.equal(C, outs[0])
    torch.testing.assert_close(outs[0].float(), ref.float(), **tol(dtype))


@pytest.mark.parametrize("M,N,K,taps,mode,kw", [(100352, 96, 96, 9, "spatial", dict(p0=14)), (50432 - 5, 384, 96, 1, "plain", {}), (50432, 96, 384, 1, "plain", {})])
def test_gemm_nt_epilogue_tile_prefetch_is_race_free_at_full_size(gpu_lib, M, N, K, taps, mode, kw):
    """The branch GEMM's epilogue stages in ring slots the last K-tile no longer reads, without a block barrier, and requests its first
    input tile (activation derivative, else residual) by LDS-DMA behind the last MFMA: full BASELINE row counts (three tile shapes),
    the in-place residual the engine uses, derivative + residual together, and a second activated output; 6 launches per case are
    bit-identical to each other, and the first / last rows equal the fp64 statement."""
    from dist_amd import ops
    dtype = torch.bfloat16
    A, B = rnd((M, K), dtype, 31), rnd((N, taps * K), dtype, 32, (taps * K) ** -0.5)
    bias, R, X = rnd((N,), torch.float32, 33), rnd((M, N), dtype, 34), rnd((M, N), dtype, 35)
    am = ops.rowmap(MODES[mode], kw.get("p0", 0), kw.get("p1", 0))
    rows = torch.cat([torch.arange(0, 400), torch.arange(M - 400, M)]).cuda()
    lin = sum(gather(A, mode, M, t, taps, **kw)[rows] @ B.double()[:, t * K:(t + 1) * K].t() for t in range(taps)) + bias.double()
    want = {"res_inplace": lin + R.double()[rows], "aux+res": lin * qgelu_grad(X.double()[rows]) + R.double()[rows], "res+act": lin + R.double()[rows]}
    for combo in ("res_inplace", "aux+res", "res+act"):
        outs = []
        for _ in range(6):
            C = R.clone() if combo == "res_inplace" else torch.full((M, N), 7.0, dtype=dtype, device="cuda")
            C2 = torch.full((M, N), 7.0, dtype=dtype, device="cuda") if combo == "res+act" else None
            ops.gemm_nt(A, B, M, N, K, taps=taps, bias=bias, res=C if combo == "res_inplace" else R, aux=X if combo == "aux+res" else None,
                        C_out=C, C2_out=C2, amap=am)
            outs.append((C, C2))
        torch.cuda.synchronize()
        for C, C2 in outs[1:]:
            assert torch.equal(C, outs[0][0]) and (C2 is None or torch.equal(C2, outs[0][1])), combo
        torch.testing.assert_close(outs[0][0][rows].double(), want[combo], **tol(dtype))
        if combo == "res+act":
            torch.testing.assert_close(outs[0][1][rows].double(), qgelu(outs[0][0][rows].double()), **tol(dtype))


def test_gemm_fast_patch_embed_maps(gpu_lib):
    """the ViT patch embedding as the 256x256 LDS-DMA kernel sees it: strided source rows (every alpha-th
    frame), rows inserted behind each frame's cls row, residual read at the destination."""
    from dist_amd import ops, lib as L
    dtype = torch.bfloat16
    Nn, alpha, nbj, K, N = 196, 2, 7, 768, 768
    M = nbj * Nn
    A = rnd((M * alpha, K), dtype, 1)
    B = rnd((N, K), dtype, 2, K ** -0.5)
    res = rnd((nbj * (Nn + 1), N), dtype, 3)
    C = torch.full((nbj * (Nn + 1), N), 7.0, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, N, K, res=res, C_out=C, amap=ops.rowmap(L.RM_STRIDED, alpha, Nn), omap=ops.outmap(L.OM_INSERTCLS, Nn))
    src = A.double().reshape(nbj, alpha, Nn, K)[:, 0].reshape(M, K)
    ref = (src @ B.double().t()).reshape(nbj, Nn, N) + res.double().reshape(nbj, Nn + 1, N)[:, 1:]
    Cr = C.double().reshape(nbj, Nn + 1, N)
    torch.testing.assert_close(Cr[:, 1:], ref, **tol(dtype))
    assert (Cr[:, 0] == 7.0).all()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Nn,nbj", [(9, 11), (196, 8)])
def test_gemm_nt_outmaps(gpu_lib, dtype, Nn, nbj):
    from dist_amd import ops, lib as L
    alpha, K = 2, 128
    M = nbj * Nn
    A = rnd((M, K), dtype, 1)
    # DUP: (bj, n) -> rows (bj*alpha + a, n), residual read at the destination
    B = rnd((96, K), dtype, 2, K ** -0.5)
    res = rnd((M * alpha, 96), dtype, 3)
    C = torch.zeros(M * alpha, 96, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, 96, K, res=res, C_out=C, omap=ops.outmap(L.OM_DUP, alpha, Nn))
    Y = (A.double() @ B.double().t()).reshape(nbj, 1, Nn, 96).expand(nbj, alpha, Nn, 96).reshape(M * alpha, 96)
    torch.testing.assert_close(C.double(), Y + res.double(), **tol(dtype))
    # INSERTCLS: (bj, n) -> row bj*(N+1) + 1 + n; cls rows untouched
    res2 = rnd((nbj * (Nn + 1), 96), dtype, 4)
    C = torch.full((nbj * (Nn + 1), 96), 7.0, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B, M, 96, K, res=res2, C_out=C, omap=ops.outmap(L.OM_INSERTCLS, Nn))
    Cr = C.double().reshape(nbj, Nn + 1, 96)
    torch.testing.assert_close(Cr[:, 1:], (A.double() @ B.double().t()).reshape(nbj, Nn, 96) + res2.double().reshape(nbj, Nn + 1, 96)[:, 1:], **tol(dtype))
    assert (Cr[:, 0] == 7.0).all()
    # SPLITCOLS: column block a of (bj, n) -> row (bj*alpha + a, n)
    B2 = rnd((alpha * 32, K), dtype, 5, K ** -0.5)
    res3 = rnd((M * alpha, 32), dtype, 6)
    C = torch.zeros(M * alpha, 32, dtype=dtype, device="cuda")
    ops.gemm_nt(A, B2, M, alpha * 32, K, res=res3, C_out=C, omap=ops.outmap(L.OM_SPLITCOLS, alpha, Nn, 32), ldc=32)
    Y = (A.double() @ B2.double().t()).reshape(nbj, Nn, alpha, 32).permute(0, 2, 1, 3).reshape(M * alpha, 32)
    torch.testing.assert_close(C.double(), Y + res3.double(), **tol(dtype))


@pytest.mark.parametrize("dtype,use_tr", [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("M,NI,K", [(3152, 384, 768), (1000, 96, 96), (5000, 96, 288), (333, 128, 64), (2048, 768, 384), (6272, 96, 96), (4000, 384, 96), (2500, 96, 384)])
def test_gemm_tn_plain(gpu_lib, dtype, use_tr, M, NI, K):
    from dist_amd import ops
    A, B = rnd((M, NI), dtype, 1), rnd((M, K), dtype, 2)
    out = torch.zeros(NI, K, dtype=torch.float32, device="cuda")
    cs = torch.zeros(NI, dtype=torch.float32, device="cuda")
    ops.gemm_tn(A, B, out, M, NI, K, use_tr=use_tr, colsum=cs)
    ref = A.double().t() @ B.double()
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-4 * M ** 0.5)
    torch.testing.assert_close(cs.double(), A.double().sum(0), rtol=1e-4, atol=1e-4 * M ** 0.5)   # fused bias gradient
    # two-phase reduction (partial tiles + reduce kernel) must give the same answer as the atomic epilogue
    out2 = torch.zeros(NI, K, dtype=torch.float32, device="cuda")
    part = torch.empty(8 << 20, dtype=torch.float32, device="cuda")
    ops.gemm_tn(A, B, out2, M, NI, K, use_tr=use_tr, partial=part)
    torch.testing.assert_close(out2.double(), ref, rtol=1e-4, atol=1e-4 * M ** 0.5)


@pytest.mark.parametrize("M,NI,K,lda,ldb,split", [(50432, 384, 768, 384, 768, 0), (9001, 384, 480, 384, 480, 384), (12345, 480, 384, 576, 384, 0),
                                                    (8262, 192, 256, 192, 256, 0), (20000, 384, 1024, 384, 1024, 0), (10000, 200, 200, 200, 200, 0)])
def test_gemm_tn_large_plain_two_group_kernel(gpu_lib, M, NI, K, lda, ldb, split):
    """the large plain weight gradients run on gemm_tn8p.hip (two wave groups, LDS-DMA ring, transpose reads; >= 8192 rows, both extents >= 192):
    ragged row counts (the descriptor zeroes rows behind a block's range), operands wider than the product (ld > NI / K), the split
    destination of the c_proj pair, the swapped orientation (480 x 384 runs as 384 x 480), the fused bias gradient - against fp64, and bit-repeatable."""
    from dist_amd import ops
    g = torch.Generator(device="cuda"); g.manual_seed(M + NI)
    A = (torch.randn(M, lda, device="cuda", generator=g) + 0.05).to(torch.bfloat16)
    B = torch.randn(M, ldb, device="cuda", generator=g).to(torch.bfloat16)
    part = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    ref = A[:, :NI].double().t() @ B[:, :K].double()
    refb = A[:, :NI].double().sum(0)

    def run():
        cs = torch.zeros(NI, device="cuda")
        if split:
            o1, o2, cs2 = torch.zeros(NI, split, device="cuda"), torch.zeros(NI, K - split, device="cuda"), torch.zeros(NI, device="cuda")
            ops.gemm_tn(A, B, o1, M, NI, K, so_i=split, so_tap=0, colsum=cs, partial=part, out2=o2, split_c=split, so_i2=K - split, colsum2=cs2)
            assert torch.equal(cs, cs2)
            return torch.cat([o1, o2], 1), cs
        o = torch.zeros(NI, K, device="cuda")
        ops.gemm_tn(A, B, o, M, NI, K, colsum=cs, partial=part)
        return o, cs
    out, cs = run()
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert float((cs.double() - refb).abs().max() / refb.abs().max()) < 2e-5
    out2, cs2 = run()
    assert torch.equal(out, out2) and torch.equal(cs, cs2)            # the second phase sums the row splits in a fixed order
    # accumulates INTO the destination; any block cap (dist_gemm_tn_args.max_blocks) gives the same sums up to fp32 order
    o = torch.full((NI, K), 0.5, device="cuda")
    ops.gemm_tn(A, B, o, M, NI, K, partial=part, max_blocks=256)
    assert float((o.double() - 0.5 - ref).abs().max() / ref.abs().max()) < 2e-5


def test_gemm_tn_large_plain_column_offset_views_end_at_the_allocation(gpu_lib):
    """dist_op_gemm_tn is public, and its operands may be column-offset views of wider buffers (the engine's [dzf | dh1 | dh2] rows).
    The buffer descriptors of the ring kernel must end with the view's last owned element: operands whose last row ends exactly at the end of
    their allocation (384 of 480 columns from column 96, 256 of 320 from column 64), ragged M, against fp64."""
    from dist_amd import ops
    M, NI, K, offa, offb = 9000 + 13, 384, 256, 96, 64
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    Abig = torch.randn(M, offa + NI, device="cuda", generator=g).to(torch.bfloat16)
    Bbig = torch.randn(M, offb + K, device="cuda", generator=g).to(torch.bfloat16)
    A, B = Abig[:, offa:], Bbig[:, offb:]
    part = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    o, cs = torch.zeros(NI, K, device="cuda"), torch.zeros(NI, device="cuda")
    ops.gemm_tn(A, B, o, M, NI, K, colsum=cs, partial=part, lda=offa + NI, ldb=offb + K)
    ref = A.double().t() @ B.double()
    assert float((o.double() - ref).abs().max() / ref.abs().max()) < 2e-5
    assert float((cs.double() - A.double().sum(0)).abs().max() / A.double().sum(0).abs().max()) < 2e-5


@pytest.mark.parametrize("dtype,use_tr", [(torch.float32, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("mode,kw,taps", [("shift", dict(p0=8 * 9, p1=9), 3), ("spatial", dict(p0=4), 9), ("strided", dict(p0=2, p1=9), 2),
                                          ("shift", dict(p0=16 * 49, p1=49), 3), ("spatial", dict(p0=14), 9), ("strided", dict(p0=2, p1=196), 2)])
def test_gemm_tn_taps_conv_layout(gpu_lib, dtype, use_tr, mode, kw, taps):
    """dW of a conv in the reference's [Co][Ci][taps] parameter layout."""
    from dist_amd import ops
    NI, K = 96, 32
    M = {"shift": kw["p0"] * 5, "spatial": kw["p0"] ** 2 * 12, "strided": kw.get("p1", 1) * 12}[mode]
    rowsB = M * 2 if mode == "strided" else M
    A, B = rnd((M, NI), dtype, 1), rnd((rowsB, K), dtype, 2)
    out = torch.zeros(NI, K, taps, dtype=torch.float32, device="cuda")
    ops.gemm_tn(A, B, out, M, NI, K, taps=taps, bmap=ops.rowmap(MODES[mode], kw.get("p0", 0), kw.get("p1", 0), 1),
                so_i=K * taps, so_tap=1, so_outer=taps, inner=1, use_tr=use_tr)
    for tap in range(taps):
        ref = A.double().t() @ gather(B, mode, M, tap, taps, **kw)
        torch.testing.assert_close(out[:, :, tap].double(), ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("frames,sign", [(8, 1), (37, -1), (512, 1), (203, 1)])
def test_gemm_tn_conv3x3_frame_resident_kernel(gpu_lib, frames, sign):
    """The 3x3 frame convolution's weight gradient (dist.py:54-60 c_fc2, 96 -> 96 channels on the 14 x 14 plane) runs on conv_dw.hip: all nine taps
    from one LDS-resident frame (padded-plane image, taps = row shifts, zeros from the buffer descriptor).  Against fp64 for every tap in the
    reference's [Co][Ci][taps] layout with the fused bias gradient, both tap directions, frame counts that leave blocks with unequal shares;
    accumulates INTO the destination; bit-repeatable; equal to the generic tap-per-tile kernel up to fp32 summation order."""
    from dist_amd import ops
    G, NI, K, taps = 14, 96, 96, 9
    M = frames * G * G
    g = torch.Generator(device="cuda"); g.manual_seed(frames)
    A = (torch.randn(M, NI, device="cuda", generator=g) + 0.03).to(torch.bfloat16)
    B = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    part = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    bm = ops.rowmap(MODES["spatial"], G, 0, sign)

    def run(init=0.0, **kw):
        out = torch.full((NI, K, taps), init, dtype=torch.float32, device="cuda")
        cs = torch.full((NI,), init, dtype=torch.float32, device="cuda")
        ops.gemm_tn(A, B, out, M, NI, K, taps=taps, bmap=bm, so_i=K * taps, so_tap=1, so_outer=taps, inner=1, colsum=cs, partial=part, **kw)
        return out, cs
    out, cs = run()
    scale = float((A.double().t() @ B.double()).abs().max())
    for tap in range(taps):
        ref = A.double().t() @ gather(B, "spatial", M, tap, taps, p0=G, sign=sign)
        assert float((out[:, :, tap].double() - ref).abs().max()) < 2e-5 * scale, tap
    refb = A.double().sum(0)
    assert float((cs.double() - refb).abs().max() / refb.abs().max()) < 2e-5
    out2, cs2 = run()
    assert torch.equal(out, out2) and torch.equal(cs, cs2)            # the second phase adds the blocks in index order
    out3, cs3 = run(init=0.25)
    assert float((out3 - 0.25 - out).abs().max()) < 1e-5 * scale and float((cs3 - 0.25 - cs).abs().max()) < 1e-4 * float(refb.abs().max())
    out4, _ = run(max_blocks=256)                                       # any block count: the same sums up to fp32 order
    assert float((out4 - out).abs().max()) < 2e-5 * scale


@pytest.mark.parametrize("clips,T,N,taps,K,sign,wide", [(5, 16, 196, 3, 96, 1, False), (3, 8, 197, 3, 96, -1, True), (2, 16, 196, 5, 768, 1, False),
                                                         (33, 16, 196, 3, 96, 1, False), (2, 5, 50, 5, 192, -1, True)])
def test_gemm_tn_temporal_taps_frame_ring_kernel(gpu_lib, clips, T, N, taps, K, sign, wide):
    """The temporal convolutions' weight gradients (dist.py:23-36 TemporalNet.c_fc1 / temporal_ffn.c_fc2: (3, 1, 1) Conv3d, 96 channels; dist.py:178-181 the
    (5, 1, 1) stem over the 768 patch columns) run on conv_t_dw.hip: a ring of frame slots in LDS, a tap = a slot, frames outside the clip = an empty
    buffer descriptor (taken for 5 taps / several 96-column tiles; the 3-tap 96 x 96 cases run the generic kernel).  Against fp64 for every tap in the reference's weight layouts (Conv3d [Co][Ci][taps]; the stem's [Co][3][taps][P][P]) with
    the fused bias gradient, both tap directions, dY as a column-offset view of wider rows (the engine's [dzf | dh2] buffer), more items than blocks;
    accumulates INTO the destination; bit-repeatable; equal to the generic tap-per-tile kernel up to fp32 summation order."""
    from dist_amd import ops
    NI = 96
    M = clips * T * N
    g = torch.Generator(device="cuda"); g.manual_seed(clips * 100 + taps)
    Aw = (torch.randn(M, NI + (384 if wide else 0), device="cuda", generator=g) + 0.03).to(torch.bfloat16)
    A = Aw[:, 384:] if wide else Aw
    B = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    part = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    bm = ops.rowmap(MODES["shift"], T * N, N, sign)
    stem = K == 768
    lay = dict(so_i=K * taps, so_tap=K // 3, so_outer=(K // 3) * taps, inner=K // 3) if stem else dict(so_i=K * taps, so_tap=1, so_outer=taps, inner=1)

    def run(init=0.0, **kw):
        out = torch.full((NI, K * taps), init, dtype=torch.float32, device="cuda")
        cs = torch.full((NI,), init, dtype=torch.float32, device="cuda")
        ops.gemm_tn(A, B, out, M, NI, K, taps=taps, bmap=bm, colsum=cs, partial=part, lda=Aw.shape[1], **lay, **kw)
        return out, cs
    out, cs = run()
    Ad = A.double()
    scale = float((Ad.t() @ B.double()).abs().max())
    for tap in range(taps):
        ref = Ad.t() @ gather(B, "shift", M, tap, taps, p0=T * N, p1=N, sign=sign)
        got = out.view(NI, 3, taps, K // 3)[:, :, tap].reshape(NI, K) if stem else out.view(NI, K, taps)[:, :, tap]
        assert float((got.double() - ref).abs().max()) < 2e-5 * scale, tap
    refb = Ad.sum(0)
    assert float((cs.double() - refb).abs().max() / refb.abs().max()) < 2e-5
    out2, cs2 = run()
    # bit-repeatable on both kernels: conv_t_dw.hip's second phase adds the blocks in index order; the generic kernel's (the 3-tap 96 x 96 cases) combines
    # its four split groups in LDS in a fixed order since round 6 (they used to meet in atomics)
    assert torch.equal(out, out2) and torch.equal(cs, cs2)
    out3, cs3 = run(init=0.25)
    assert float((out3 - 0.25 - out).abs().max()) < 2e-5 * scale and float((cs3 - 0.25 - cs).abs().max()) < 1e-4 * float(refb.abs().max())
    out4, _ = run(max_blocks=7)                                         # any block count: the same sums up to fp32 order
    assert float((out4 - out).abs().max()) < 2e-5 * scale


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("Nn", [9, 196])
def test_gemm_tn_skipcls_a(gpu_lib, dtype, Nn):
    from dist_amd import ops, lib as L
    nbj, NI, K = 13, 128, 96
    M = nbj * Nn
    A, B = rnd((nbj * (Nn + 1), NI), dtype, 1), rnd((M, K), dtype, 2)
    out = torch.zeros(NI, K, dtype=torch.float32, device="cuda")
    ops.gemm_tn(A, B, out, M, NI, K, amap=ops.rowmap(L.RM_SKIPCLS, Nn))
    ref = A.double().reshape(nbj, Nn + 1, NI)[:, 1:].reshape(M, NI).t() @ B.double()
    torch.testing.assert_close(out.double(), ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("rows,C", [(3152, 768), (1000, 96), (777, 384), (100, 1024), (50, 32), (64, 128)])
def test_layernorm_fwd_bwd(gpu_lib, dtype, rows, C):
    from dist_amd import ops
    x = rnd((rows, C), dtype, 1, 2.0)
    w, b = 1 + 0.1 * rnd((C,), torch.float32, 2), 0.1 * rnd((C,), torch.float32, 3)
    w2, b2 = 1 + 0.1 * rnd((C,), torch.float32, 4), 0.1 * rnd((C,), torch.float32, 5)
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    y2 = torch.empty_like(x)
    y = ops.layernorm(x, w, b, y2=y2, w2=w2, b2=b2, mean=mean, rstd=rstd)
    xd = x.double().requires_grad_(True)
    wd, bd, w2d, b2d = [t.double().requires_grad_(True) for t in (w, b, w2, b2)]
    ry = torch.nn.functional.layer_norm(xd, (C,), wd, bd, 1e-5)
    ry2 = torch.nn.functional.layer_norm(xd, (C,), w2d, b2d, 1e-5)
    torch.testing.assert_close(y.double(), ry, **tol(dtype))
    torch.testing.assert_close(y2.double(), ry2, **tol(dtype))
    dy, dy2 = rnd((rows, C), dtype, 6), rnd((rows, C), dtype, 7)
    (ry * dy.double() + ry2 * dy2.double()).sum().backward(retain_graph=True)
    dx0 = rnd((rows, C), dtype, 8)
    dx = dx0.clone()
    dw, db, dw2, db2 = [torch.zeros(C, device="cuda") for _ in range(4)]
    ops.layernorm_bwd(x, mean, rstd, dy, w, dy2=dy2, w2=w2, dx=dx, accumulate=True, dw=dw, db=db, dw2=dw2, db2=db2)
    torch.testing.assert_close(dx.double(), xd.grad + dx0.double(), **tol(dtype, 4.0))
    for got, ref in ((dw, wd.grad), (db, bd.grad), (dw2, w2d.grad), (db2, b2d.grad)):
        torch.testing.assert_close(got.double(), ref, rtol=1e-3, atol=2e-3 * rows ** 0.5)
    # two-phase parameter gradients (ABI 8: per-block partial sums + a fixed-order reduction, no atomics): same values, accumulated INTO the
    # given buffers, and bit-repeatable
    pre = 0.5
    outs = []
    for _ in range(2):
        dx_t = dx0.clone()
        g4 = [torch.full((C,), pre, device="cuda") for _ in range(4)]
        ops.layernorm_bwd(x, mean, rstd, dy, w, dy2=dy2, w2=w2, dx=dx_t, accumulate=True, dw=g4[0], db=g4[1], dw2=g4[2], db2=g4[3], two_phase=True)
        outs.append(g4)
    torch.testing.assert_close(dx_t.double(), xd.grad + dx0.double(), **tol(dtype, 4.0))
    for got, again, ref in zip(outs[0], outs[1], (wd.grad, bd.grad, w2d.grad, b2d.grad)):
        torch.testing.assert_close(got.double() - pre, ref, rtol=1e-3, atol=2e-3 * rows ** 0.5)
        assert torch.equal(got, again)
    g2 = [torch.zeros(C, device="cuda") for _ in range(2)]          # a subset of the arrays (single-input form)
    ops.layernorm_bwd(x, mean, rstd, dy, w, dx=dx_t, dw=g2[0], db=g2[1], two_phase=True)
    single = (ry * dy.double()).sum()
    gw, gb = torch.autograd.grad(single, (wd, bd))
    torch.testing.assert_close(g2[0].double(), gw, rtol=1e-3, atol=2e-3 * rows ** 0.5)
    torch.testing.assert_close(g2[1].double(), gb, rtol=1e-3, atol=2e-3 * rows ** 0.5)
    # out-of-place accumulate + second copy (used by the two-stream backward)
    dxo, dxc = torch.empty_like(x), torch.empty_like(x)
    ops.layernorm_bwd(x, mean, rstd, dy, w, dy2=dy2, w2=w2, dx=dxo, dx_add=dx0, dx_copy=dxc)
    torch.testing.assert_close(dxo.double(), xd.grad + dx0.double(), **tol(dtype, 4.0))
    assert torch.equal(dxo, dxc)


@pytest.mark.parametrize("dtype", DT)
def test_layernorm_addend(gpu_lib, dtype):
    from dist_amd import ops
    L_, C, frames = 10, 128, 6
    x = rnd((frames * L_, C), dtype, 1)
    pos = rnd((L_, C), torch.float32, 2)
    w, b = 1 + 0.1 * rnd((C,), torch.float32, 3), 0.1 * rnd((C,), torch.float32, 4)
    y = ops.layernorm(x, w, b, addend=pos, period=L_)
    ref = torch.nn.functional.layer_norm(x.double().reshape(frames, L_, C) + pos.double(), (C,), w.double(), b.double(), 1e-5)
    torch.testing.assert_close(y.double().reshape(frames, L_, C), ref, **tol(dtype))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("frames,L_,heads", [(6, 197, 12), (3, 257, 16), (5, 17, 2), (4, 10, 2), (2, 32, 1)])
def test_vit_attention(gpu_lib, dtype, frames, L_, heads):
    from dist_amd import ops
    d = heads * 64
    qkv = rnd((frames * L_, 3 * d), dtype, 1)
    out = ops.attention(qkv, frames, L_, heads)
    q, k, v = qkv.double().reshape(frames, L_, 3, heads, 64).permute(2, 0, 3, 1, 4)
    att = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    ref = (att @ v).permute(0, 2, 1, 3).reshape(frames * L_, d)
    torch.testing.assert_close(out.double(), ref, **(dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("frames,L_,heads", [(6, 197, 12), (3, 257, 16), (5, 17, 2)])
def test_vit_attention_head_major(gpu_lib, dtype, frames, L_, heads):
    """same attention on the [frame][head][q|k|v][L][64] layout the QKV GEMM writes with DIST_OM_HEADS"""
    from dist_amd import ops, lib as L
    d = heads * 64
    qkv = rnd((frames * L_, 3 * d), dtype, 1)
    hm = qkv.reshape(frames, L_, 3, heads, 64).permute(0, 3, 2, 1, 4).contiguous()
    out = ops.attention(hm, frames, L_, heads, layout=L.QKV_HEADS)
    ref = ops.attention(qkv, frames, L_, heads)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("frames,L_,heads,K", [(7, 197, 12, 768), (26, 197, 4, 256), (3, 17, 2, 64), (5, 50, 6, 128)])
def test_gemm_heads_outmap(gpu_lib, dtype, frames, L_, heads, K):
    """QKV projection written head-major (both the 256x256 LDS-DMA kernel and the generic kernel take this out-map)"""
    from dist_amd import ops, lib as L
    M, N = frames * L_, 3 * heads * 64
    A = rnd((M, K), dtype, 1)
    W = (rnd((N, K), torch.float32, 2) * K ** -0.5).to(dtype)
    bias = rnd((N,), torch.float32, 3)
    plain = torch.empty(M, N, dtype=dtype, device="cuda")
    ops.gemm_nt(A, W, M, N, K, bias=bias, C_out=plain)
    hm = torch.full((frames, heads, 3, L_, 64), float("nan"), dtype=dtype, device="cuda")
    ops.gemm_nt(A, W, M, N, K, bias=bias, C_out=hm, ldc=64, omap=ops.outmap(L.OM_HEADS, L_, heads))
    ref = plain.reshape(frames, L_, 3, heads, 64).permute(0, 3, 2, 1, 4)
    # the plain call may take another kernel (small-M split-K) with a different summation order: one rounding apart
    torch.testing.assert_close(hm.float(), ref.float(), **(dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)))
    assert not torch.isnan(hm.float()).any()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("B,S,C", [(16, 197, 384), (4, 8, 384), (6, 10, 128)])
def test_xattn1q(gpu_lib, dtype, B, S, C):
    from dist_amd import ops
    H = C // 64
    q, kv = rnd((B, C), dtype, 1), rnd((B * S, 2 * C), dtype, 2)
    o, probs = ops.xattn1q(q, kv, B, S, C)
    qd = q.double().requires_grad_(True)
    kvd = kv.double().requires_grad_(True)
    qh = qd.reshape(B, H, 1, 64)
    kh = kvd.reshape(B, S, 2, H, 64)[:, :, 0].permute(0, 2, 1, 3)
    vh = kvd.reshape(B, S, 2, H, 64)[:, :, 1].permute(0, 2, 1, 3)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / 8.0, dim=-1)
    ref = (att @ vh).reshape(B, C)
    torch.testing.assert_close(o.double(), ref, **tol(dtype))
    torch.testing.assert_close(probs.double(), att.reshape(B, H, S), rtol=1e-4, atol=1e-5)
    do = rnd((B, C), dtype, 3)
    (ref * do.double()).sum().backward()
    dq, dkv = ops.xattn1q_bwd(q, kv, probs, do, B, S, C)
    torch.testing.assert_close(dq.double(), qd.grad, **tol(dtype))
    torch.testing.assert_close(dkv.double(), kvd.grad, **tol(dtype))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("P,res", [(16, 64), (14, 56)])
def test_patchify(gpu_lib, dtype, P, res):
    from dist_amd import ops
    b, T = 2, 4
    video = rnd((b, 3, T, res, res), torch.float32, 1)
    out = ops.patchify(video, P, dtype)
    G = res // P
    ref = video.permute(0, 2, 1, 3, 4).reshape(b, T, 3, G, P, G, P).permute(0, 1, 3, 5, 2, 4, 6).reshape(b * T * G * G, 3 * P * P)
    assert out.shape[1] == (3 * P * P + 7) // 8 * 8
    torch.testing.assert_close(out[:, :3 * P * P].float(), ref.to(dtype).float(), rtol=0, atol=0)
    assert (out[:, 3 * P * P:] == 0).all()


@pytest.mark.parametrize("dtype", DT)
def test_elementwise_colsum(gpu_lib, dtype):
    from dist_amd import ops, lib as L
    a, b = rnd((1000, 96), dtype, 1), rnd((1000, 96), dtype, 2)
    torch.testing.assert_close(ops.add(a, b).double(), (a.double() + b.double()), **tol(dtype))
    torch.testing.assert_close(ops.gelu_bwd(a, b).double(), a.double() * qgelu_grad(b.double()), **tol(dtype))
    for rows, C in ((5000, 96), (3000, 384), (100, 768), (77, 32)):
        x = rnd((rows, C), dtype, 3)
        out = torch.zeros(C, device="cuda")
        ops.colsum(x, out, rows, C)
        torch.testing.assert_close(out.double(), x.double().sum(0), rtol=1e-4, atol=1e-3)
    Nn, nbj, C = 9, 7, 128
    x = rnd((nbj * (Nn + 1), C), dtype, 4)
    out = torch.zeros(C, device="cuda")
    ops.colsum(x, out, nbj * Nn, C, rmap=ops.rowmap(L.RM_SKIPCLS, Nn))
    torch.testing.assert_close(out.double(), x.double().reshape(nbj, Nn + 1, C)[:, 1:].sum((0, 1)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("b,E,K", [(4, 512, 174), (3, 64, 10), (2, 768, 400)])
def test_logits_loss(gpu_lib, dtype, b, E, K):
    from dist_amd import ops
    v = rnd((b, E), dtype, 1)
    text = rnd((K, E), torch.float32, 2)
    ls = torch.tensor(2.659, device="cuda")
    y = torch.softmax(rnd((b, K), torch.float32, 3), -1)
    logits, vid, loss, dv, dls = ops.logits_loss(v, text, ls, soft_target=y)
    vd = v.double().requires_grad_(True)
    lsd = ls.double().requires_grad_(True)
    vn = vd / vd.norm(dim=1, keepdim=True)
    tn = text.double() / text.double().norm(dim=1, keepdim=True)
    rl = lsd.exp() * vn @ tn.t()
    rloss = torch.sum(-y.double() * torch.log_softmax(rl, -1), -1).mean()
    rloss.backward()
    torch.testing.assert_close(logits.double(), rl, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(vid.double(), vn, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(loss.double(), rloss, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dv.double(), vd.grad, **tol(dtype, 0.05))
    torch.testing.assert_close(dls.double(), lsd.grad, rtol=1e-3, atol=1e-4)
    # externally supplied dlogits (the drop-in autograd path)
    dl = rnd((b, K), torch.float32, 4, 0.01)
    _, _, _, dv2, dls2 = ops.logits_loss(v, text, ls, dlogits_in=dl)
    vd.grad = None; lsd.grad = None
    vn = vd / vd.norm(dim=1, keepdim=True)
    ((lsd.exp() * vn @ tn.t()) * dl.double()).sum().backward()
    torch.testing.assert_close(dv2.double(), vd.grad, **tol(dtype, 0.05))
    torch.testing.assert_close(dls2.double(), lsd.grad, rtol=1e-3, atol=1e-4)


def test_adamw(gpu_lib):
    from dist_amd import ops
    n = 10000
    p0 = rnd((n,), torch.float32, 1)
    segs_spec = [(0, 3000, 3.2e-4, 1e-4), (3000, 3004, 1e-3, 0.0), (3004, n, 3.2e-4, 0.0)]
    ps = [torch.nn.Parameter(p0[b:e].clone()) for b, e, _, _ in segs_spec]
    opt = torch.optim.AdamW([{"params": [q], "lr": lr, "weight_decay": wd} for q, (_, _, lr, wd) in zip(ps, segs_spec)], betas=(0.9, 0.999))
    p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    segs = ops.make_segs(segs_spec, "cuda")
    for step in range(1, 4):
        g = rnd((n,), torch.float32, 10 + step)
        for q, (b, e, _, _) in zip(ps, segs_spec):
            q.grad = g[b:e].clone()
        opt.step()
        ops.adamw(p, g, m, v, segs, step)
        torch.testing.assert_close(p, torch.cat([q.data for q in ps]), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("M,N,K,act", [(2048, 768, 768, False), (1500, 2304, 768, False), (1111, 1024, 256, True)])
def test_layernorm_folded_into_gemm(gpu_lib, M, N, K, act):
    """DIST_EPI_LNFOLD (reference clip.py:160-176: ln_1 -> attn in_proj, ln_2 -> mlp.c_fc): the GEMM consumes the RAW rows with
    W diag(gamma) and normalises in its epilogue, v = rstd (acc - mean colsum) + (b + W beta); `dist_op_layernorm` with y = NULL
    only produces the row statistics.  Checked against fp32 LayerNorm -> Linear (-> QuickGELU) on the same bf16 inputs."""
    from dist_amd import ops, lib as L
    torch.manual_seed(0)
    dev = "cuda"
    x = (torch.randn(M, K, device=dev) * 1.7 + 0.8 + 3.0 * torch.randn(1, K, device=dev)).to(torch.bfloat16)   # rows far from zero mean
    W = torch.randn(N, K, device=dev) * K ** -0.5
    bias = torch.randn(N, device=dev)
    gamma = 1.0 + 0.3 * torch.randn(K, device=dev)
    beta = 0.2 * torch.randn(K, device=dev)
    Wp, cs, bf = ops.ln_fold(W, bias, gamma, beta)
    torch.testing.assert_close(Wp.float(), (W * gamma).to(torch.bfloat16).float(), rtol=0, atol=0)
    torch.testing.assert_close(cs, Wp.float().sum(1), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(bf, bias + W @ beta, rtol=1e-5, atol=1e-4)
    stats = torch.empty(2 * M, device=dev)
    assert ops.layernorm(x, gamma, beta, y=False, mean=stats[:M], rstd=stats[M:]) is None
    xf = x.float()
    mu, var = xf.mean(1), xf.var(1, unbiased=False)
    torch.testing.assert_close(stats[:M], mu, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stats[M:], torch.rsqrt(var + 1e-5), rtol=1e-4, atol=1e-5)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    out2 = torch.empty_like(out) if act else None
    ops.gemm_nt(x, Wp, M, N, K, bias=bf, lnfold=(stats, cs), C_out=out, C2_out=out2)
    ref = torch.nn.functional.layer_norm(xf, (K,), gamma, beta, 1e-5) @ W.t() + bias
    scale = float(ref.abs().max())
    assert float((out.float() - ref).abs().max()) < 1.5e-2 * scale, float((out.float() - ref).abs().max()) / scale
    if act:
        refa = ref * torch.sigmoid(1.702 * ref)
        assert float((out2.float() - refa).abs().max()) < 1.5e-2 * scale
    # not combinable with the activation-derivative epilogue, and only for shapes the LDS-DMA kernel takes
    with pytest.raises(L.DistError):
        ops.gemm_nt(x[:512], Wp, 512, N, K, bias=bf, lnfold=(stats, cs), C_out=out[:512])


@pytest.mark.parametrize("M,N,K", [(4096, 768, 768), (3152, 768, 3072), (1500, 1024, 1024)])
def test_row_statistics_from_the_producer_gemm(gpu_lib, M, N, K):
    """DIST_EPI_ROWSTATS / ln_part: the GEMM that writes the residual stream (reference clip.py:160-176, x = x + attn(..),
    x = x + mlp(..)) leaves, per row and 64-column slice, the sum and the sum of squares of the values it STORED;
    dist_op_ln_stats_from_partials turns them into the mean / rstd the next LayerNorm-folded GEMM consumes, so the rows are not re-read."""
    from dist_amd import ops, lib as L
    torch.manual_seed(1)
    dev, dt = "cuda", torch.bfloat16
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    bias = torch.randn(N, device=dev)
    res = (torch.randn(M, N, device=dev) * 1.5 + 0.7).to(dt)
    X = torch.empty(M, N, dtype=dt, device=dev)
    part = torch.full((N // 64, M, 2), float("nan"), device=dev)
    ops.gemm_nt(A, W, M, N, K, bias=bias, res=res, C_out=X, rowstats=part)
    X0 = torch.empty_like(X)
    ops.gemm_nt(A, W, M, N, K, bias=bias, res=res, C_out=X0)
    assert torch.equal(X, X0)                                     # the output itself is untouched by the extra epilogue
    xs = X.float().reshape(M, N // 64, 64)
    torch.testing.assert_close(part[:, :, 0].t(), xs.sum(2), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(part[:, :, 1].t(), (xs * xs).sum(2), rtol=1e-5, atol=1e-4)
    part2 = torch.empty_like(part)                                # fixed summation order: a second launch gives the same bits
    ops.gemm_nt(A, W, M, N, K, bias=bias, res=res, C_out=X0, rowstats=part2)
    assert torch.equal(part, part2)
    # dist_op_ln_stats_from_partials: the statistics dist_op_layernorm(y = NULL) computes from the rows, without the rows
    st = ops.ln_stats_from_partials(part, N)
    xf = X.float()
    torch.testing.assert_close(st[:M], xf.mean(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(st[M:], torch.rsqrt(xf.var(1, unbiased=False) + 1e-5), rtol=1e-4, atol=1e-5)
    assert torch.equal(st, ops.ln_stats_from_partials(part, N))
    # consumer: LayerNorm(X) -> Linear through the fold with those statistics
    N2 = 1024
    W2 = torch.randn(N2, N, device=dev) * N ** -0.5
    b2 = torch.randn(N2, device=dev)
    gamma = 1.0 + 0.3 * torch.randn(N, device=dev)
    beta = 0.2 * torch.randn(N, device=dev)
    Wp, cs, bf = ops.ln_fold(W2, b2, gamma, beta)
    out = torch.empty(M, N2, dtype=dt, device=dev)
    ops.gemm_nt(X, Wp, M, N2, N, bias=bf, lnfold=(st, cs), C_out=out)
    ref = torch.nn.functional.layer_norm(xf, (N,), gamma, beta, 1e-5) @ W2.t() + b2
    scale = float(ref.abs().max())
    assert float((out.float() - ref).abs().max()) < 1.5e-2 * scale
    # only the LDS-DMA kernel leaves partials: other shapes refuse instead of silently skipping them
    with pytest.raises(L.DistError):
        ops.gemm_nt(A[:512], W, 512, N, K, bias=bias, res=res[:512], C_out=X[:512], rowstats=part)


@pytest.mark.parametrize("kind,frames,L_,N,K", [("in_proj", 9, 197, 2304, 768), ("in_proj", 5, 257, 3072, 1024), ("c_fc", 6, 197, 3072, 768), ("proj", 7, 197, 768, 3072),
                                              ("proj", 6, 197, 768, 768), ("none", 11, 197, 384, 768), ("bias", 11, 197, 384, 768), ("bias_res", 11, 197, 384, 768),
                                              ("bias_res", 5, 257, 1024, 256)])
def test_gemm_compile_time_epilogues_vs_fp32(gpu_lib, kind, frames, L_, N, K):
    """The straight-line instantiations of the 256 x 256 kernel's epilogue (gemm_fast8p_kernel<false, CF>: the frozen ViT's in_proj = LayerNorm fold + head-major
    q | k | v, c_fc = LayerNorm fold + QuickGELU only, out_proj / c_proj = bias + residual + row statistics; the branch's none / bias / bias + residual; reference
    clip.py:155-176, dist.py:23-45) against fp32 torch on the same bf16 operands, with a ragged last row tile (M = frames x L is not a multiple of 256) and a
    half-empty column tile (N = 384): rows beyond M and columns beyond N are cut off by buffer descriptors there, not by masks."""
    from dist_amd import ops, lib as L
    torch.manual_seed(7)
    dev, dt = "cuda", torch.bfloat16
    M = frames * L_
    assert M % 256 and M >= 1024
    x = (torch.randn(M, K, device=dev) * 1.3 + 0.5).to(dt)
    W = torch.randn(N, K, device=dev) * K ** -0.5
    bias = torch.randn(N, device=dev)
    res = (torch.randn(M, N, device=dev) * 1.5).to(dt)
    Wb = W.to(dt)
    lin = x.float() @ Wb.float().t()
    guard = 3.0                                            # canary rows behind the output: nothing may be written past row M - 1
    if kind in ("in_proj", "c_fc"):
        gamma, beta = 1.0 + 0.3 * torch.randn(K, device=dev), 0.2 * torch.randn(K, device=dev)
        Wp, cs, bf = ops.ln_fold(W, bias, gamma, beta)
        stats = torch.empty(2 * M, device=dev)
        ops.layernorm(x, gamma, beta, y=False, mean=stats[:M], rstd=stats[M:])
        ref = torch.nn.functional.layer_norm(x.float(), (K,), gamma, beta, 1e-5) @ W.t() + bias
        scale = float(ref.abs().max())
        if kind == "in_proj":
            heads = N // 192
            buf = torch.full((frames + 1, heads, 3, L_, 64), guard, dtype=dt, device=dev)
            ops.gemm_nt(x, Wp, M, N, K, bias=bf, lnfold=(stats, cs), C_out=buf, ldc=64, omap=ops.outmap(L.OM_HEADS, L_, heads))
            got = buf[:frames].permute(0, 3, 2, 1, 4).reshape(M, N).float()
            assert bool((buf[frames] == guard).all())
        else:
            buf = torch.full((M + 64, N), guard, dtype=dt, device=dev)
            ops.gemm_nt(x, Wp, M, N, K, bias=bf, lnfold=(stats, cs), C2_out=buf[:M])
            got = buf[:M].float()
            ref = ref * torch.sigmoid(1.702 * ref)
            assert bool((buf[M:] == guard).all())
        assert float((got - ref).abs().max()) < 1.5e-2 * scale
        return
    buf = torch.full((M + 64, N), guard, dtype=dt, device=dev)
    if kind == "proj":
        part = torch.full((N // 64, M, 2), float("nan"), device=dev)
        ops.gemm_nt(x, Wb, M, N, K, bias=bias, res=res, C_out=buf[:M], rowstats=part)
        ref = lin + bias + res.float()
        xs = buf[:M].float().reshape(M, N // 64, 64)
        torch.testing.assert_close(part[:, :, 0].t(), xs.sum(2), rtol=1e-5, atol=1e-4)
        torch.testing.assert_close(part[:, :, 1].t(), (xs * xs).sum(2), rtol=1e-5, atol=1e-3)
    elif kind == "none":
        ops.gemm_nt(x, Wb, M, N, K, C_out=buf[:M]); ref = lin
    elif kind == "bias":
        ops.gemm_nt(x, Wb, M, N, K, bias=bias, C_out=buf[:M]); ref = lin + bias
    else:
        ops.gemm_nt(x, Wb, M, N, K, bias=bias, res=res, C_out=buf[:M]); ref = lin + bias + res.float()
    torch.testing.assert_close(buf[:M].float(), ref, rtol=1.2e-2, atol=1.2e-2 * float(ref.abs().max()) / 4)
    assert bool((buf[M:] == guard).all())
