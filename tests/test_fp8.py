"""fp8 operands of the frozen spatial branch (BASELINE config 5): dist_op_quant_rows_fp8 + the DIST_EPI_FP8 mode of
dist_op_gemm_nt (block-scaled e4m3 MFMA, 256x256 LDS-DMA kernel).  The reference has no fp8 path (oracle/fp8_oracle.py: parity
unpinned); the oracle is pinned to torch's float8_e4m3fn cast.

Not GPU: the oracle's quantisation round-trips and bounds.  GPU: the quantiser is bit-identical to the oracle; the GEMM equals the fp64
product of the dequantised operands within fp32-accumulation tolerance for every epilogue the ViT uses; the end-to-end quantisation
error against the unquantised product is within the e4m3 bound."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
import fp8_oracle as fo  # noqa: E402


def rnd(shape, seed, scale=1.0, dtype=torch.bfloat16, device="cuda"):
    g = torch.Generator(device=device).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=device) * scale).to(dtype)


def test_oracle_quantisation_bounds_and_edge_rows():
    x = rnd((64, 256), 1, 3.0, torch.float32, "cpu")
    x[3] = 0                                                        # all-zero row keeps scale 1 and zero bytes
    x[5, 7] = 1000.0                                                # an outlier sets the row's scale
    q, s = fo.quant_rows(x)
    assert q.dtype == torch.uint8 and s.dtype == torch.float32 and float(s[3]) == 1.0 and int(q[3].sum()) == 0
    d = fo.dequant(q, s)
    assert float(d[5, 7]) == pytest.approx(1000.0, rel=1e-6)        # amax maps to 448 exactly
    rel = (d - x.double()).abs() / x.double().abs().clamp_min(1e-30)
    big = x.abs() > (x.abs().amax(dim=1, keepdim=True) * 2.0 ** -6) # normal range of e4m3 below the row maximum: 3 mantissa bits
    assert float(rel[big].max()) <= 2.0 ** -4 + 1e-6
    assert not torch.isnan(d).any()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,K", [(1000, 768), (513, 1024), (64, 4096), (300, 3072), (7, 8)])
def test_hip_quantiser_is_bit_identical_to_the_oracle(gpu_lib, dtype, rows, K):
    from dist_amd import ops
    x = rnd((rows, K), 11, 2.5, dtype)
    x[1] = 0
    x[2, K // 2] = 3.0e4 if dtype == torch.float32 else 3.0e4
    q, s = ops.quant_rows_fp8(x)
    qo, so = fo.quant_rows(x.cpu())
    assert torch.equal(s.cpu(), so)
    assert torch.equal(q.cpu(), qo)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(2048, 768, 768), (1300, 2304, 768), (4096, 1024, 4096), (1024, 384, 256), (3000, 3072, 1024)])
def test_fp8_gemm_matches_the_dequantised_product(gpu_lib, M, N, K):
    from dist_amd import ops
    A, W = rnd((M, K), 21), rnd((N, K), 22, K ** -0.5)
    qa, sa = ops.quant_rows_fp8(A)
    qw, sw = ops.quant_rows_fp8(W)
    bias = rnd((N,), 23, 1.0, torch.float32)
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(qa, qw, M, N, K, bias=bias, C_out=C, fp8=(sa, sw))
    ref = fo.gemm(qa.cpu(), sa.cpu(), qw.cpu(), sw.cpu()) + bias.cpu().double()
    torch.testing.assert_close(C.cpu().double(), ref, rtol=1.2e-2, atol=1.2e-2)           # bf16 output rounding
    # the quantisation error itself, against the unquantised operands: ~2^-4 per element, averaged down by the K-sum
    exact = A.cpu().double() @ W.cpu().double().t() + bias.cpu().double()
    err = (ref - exact).abs().max() / exact.abs().max()
    assert float(err) < 0.05, float(err)


@pytest.mark.gpu
def test_fp8_gemm_epilogues_of_the_vit(gpu_lib):
    """residual + row statistics (out / proj), LayerNorm fold + head-major output (qkv), LayerNorm fold + QuickGELU-only output (fc)."""
    from dist_amd import ops, lib as L
    M, N, K = 197 * 12, 768, 768
    A, W = rnd((M, K), 31, 2.0), rnd((N, K), 32, K ** -0.5)
    qa, sa = ops.quant_rows_fp8(A)
    qw, sw = ops.quant_rows_fp8(W)
    lin = fo.gemm(qa.cpu(), sa.cpu(), qw.cpu(), sw.cpu())
    bias, R = rnd((N,), 33, 1.0, torch.float32), rnd((M, N), 34)
    # residual + ROWSTATS
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(N // 64, M, 2, device="cuda")
    ops.gemm_nt(qa, qw, M, N, K, bias=bias, res=R, C_out=C, rowstats=part, fp8=(sa, sw))
    torch.testing.assert_close(C.cpu().double(), lin + bias.cpu().double() + R.cpu().double(), rtol=1.2e-2, atol=1.2e-2)
    Cs = C.float().view(M, N // 64, 64)
    torch.testing.assert_close(part[..., 0].t(), Cs.sum(-1), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(part[..., 1].t(), (Cs * Cs).sum(-1), rtol=1e-4, atol=1e-3)
    # LayerNorm fold: v = rstd * (acc - mean * colsum) + bias, QuickGELU-only output
    stats = torch.stack([A.float().mean(1), (A.float().var(1, unbiased=False) + 1e-5).rsqrt()]).contiguous()
    colsum = fo.dequant(qw.cpu(), sw.cpu()).sum(1).float().cuda()
    C2 = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(qa, qw, M, N, K, bias=bias, C2_out=C2, lnfold=(stats, colsum), fp8=(sa, sw))
    v = stats[1].cpu().double()[:, None] * (lin - stats[0].cpu().double()[:, None] * colsum.cpu().double()[None, :]) + bias.cpu().double()
    torch.testing.assert_close(C2.cpu().double(), v * torch.sigmoid(1.702 * v), rtol=2e-2, atol=2e-2)
    # head-major q | k | v output
    heads, L_ = 4, 197
    N3 = 3 * 64 * heads
    W3 = rnd((N3, K), 35, K ** -0.5)
    qw3, sw3 = ops.quant_rows_fp8(W3)
    out = torch.full((12 * heads * 3 * L_, 64), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(qa, qw3, M, N3, K, C_out=out, ldc=64, omap=ops.outmap(L.OM_HEADS, L_, heads), fp8=(sa, sw3))
    ref = fo.gemm(qa.cpu(), sa.cpu(), qw3.cpu(), sw3.cpu()).view(12, L_, 3, heads, 64).permute(0, 3, 2, 1, 4).reshape(-1, 64)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1.2e-2, atol=1.2e-2)


@pytest.mark.gpu
def test_fp8_gemm_is_race_free_at_full_size_and_refuses_bad_shapes(gpu_lib):
    from dist_amd import ops, lib as L
    M, N, K = 50432, 2304, 768
    A, W = rnd((M, K), 41), rnd((N, K), 42, K ** -0.5)
    qa, sa = ops.quant_rows_fp8(A)
    qw, sw = ops.quant_rows_fp8(W)
    outs = []
    for _ in range(8):
        C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        ops.gemm_nt(qa, qw, M, N, K, C_out=C, fp8=(sa, sw))
        outs.append(C)
    torch.cuda.synchronize()
    assert all(torch.equal(C, outs[0]) for C in outs[1:])
    rows = torch.cat([torch.arange(0, 300), torch.arange(M - 300, M)])
    ref = fo.gemm(qa.cpu()[rows], sa.cpu()[rows], qw.cpu(), sw.cpu())
    torch.testing.assert_close(outs[0].cpu()[rows].double(), ref, rtol=1.2e-2, atol=1.2e-2)
    C = torch.empty(2048, 256, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(L.DistError):                                 # K not a multiple of 128
        ops.gemm_nt(qa[:2048, :192].contiguous(), qw[:256, :192].contiguous(), 2048, 256, 192, C_out=C, fp8=(sa, sw))
    with pytest.raises(L.DistError):                                 # fewer than 1024 rows: no fp8 kernel for small GEMMs
        ops.gemm_nt(qa[:512], qw[:256], 512, 256, K, C_out=C, fp8=(sa, sw))


# ---- the engine's fp8 frozen spatial branch (dist_config.vit_fp8) ---------------------------------------------------------------
def _engine(gname, b, vit_fp8):
    from dist_amd import synth
    from dist_amd.engine import Engine, config_from_geometry
    g = synth.geometry(gname)
    eng = Engine(config_from_geometry(g, b, torch.bfloat16, True, vit_fp8))
    sd = synth.state_dict(g)
    eng.load_state_dict(sd)
    video = torch.from_numpy(synth.video(g, b)).cuda()
    text = torch.from_numpy(synth.text_features(g)).cuda()
    tgt = torch.from_numpy(synth.soft_target(g, b)[0]).cuda()
    return g, eng, sd, video, text, tgt


def _rel(a, b):
    a, b = a.double().cpu().reshape(-1), torch.as_tensor(b).double().cpu().reshape(-1)
    return float((a - b).norm() / b.norm())


@pytest.mark.gpu
@pytest.mark.parametrize("mask", [15, 5])
def test_engine_fp8_vit_matches_the_oracle_with_the_same_quantisation_points(gpu_lib, mask):
    """BASELINE config 1 geometry (ViT-B/16 8+16f, b = 2; 3152 token rows: the fp8 kernel takes every GEMM of the frozen tower): the
    saved features of every block against the CPU oracle that quantises the same operands at the same points (per-row e4m3, folded
    LayerNorm), and the logits against the bf16 engine and the reference's fp32 golden."""
    from dist_oracle import Oracle
    g, eng, sd, video, text, tgt = _engine("b16_8+16f", 2, mask)
    loss, logits = eng.forward_backward(video, text, tgt)
    feats = [eng.debug(f"feat.{i}").clone().cpu().float() for i in range(g.layers)]
    o8, o16 = Oracle(g, sd, dtype=torch.float32, bf16=True, vit_fp8=mask), Oracle(g, sd, dtype=torch.float32, bf16=True)
    # block by block on the ENGINE's own input of the block: e4m3 has a 6 % rounding step, so bf16-level differences of an input flip
    # a few per cent of the roundings and decorrelate most of the quantisation noise - errors must not be allowed to accumulate
    for i in (1, 6, 11):
        x_in = feats[i - 1].reshape(2, g.t, g.L, g.d)
        with torch.no_grad():
            r8, r16 = o8.vit_block(x_in, i), o16.vit_block(x_in, i)
        e8, e16, q = _rel(feats[i], r8), _rel(feats[i], r16), _rel(r8, r16)
        print(f"fp8 mask {mask} block {i}: engine vs fp8 oracle {e8:.4f}, engine vs bf16 oracle {e16:.4f}, fp8 oracle vs bf16 oracle {q:.4f}")
        assert e8 < 0.012 and e8 < 0.6 * q and e16 > 0.8 * q, (i, e8, e16, q)      # the engine follows the fp8 oracle, not the bf16 one
        assert 0.003 < q < 0.05, q                                                   # the quantisation is visible and bounded
    g0, eng0, _, _, _, _ = _engine("b16_8+16f", 2, 0)
    loss0, logits0 = eng0.forward_backward(video, text, tgt)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "b16_b2.npz"))
    gap16 = float((logits.float() - logits0.float()).abs().max())
    gap32 = float((logits.cpu().double() - torch.from_numpy(gold["logits"]).double()).abs().max())
    print(f"fp8 mask {mask}: logits vs bf16 engine {gap16:.4f}, vs fp32 reference golden {gap32:.4f}")
    assert 0 < gap16 < 0.25 and gap32 < 0.35
    assert (logits.cpu().argmax(1) == torch.from_numpy(gold["logits"]).argmax(1)).all()
    assert torch.isfinite(eng.grads).all() and abs(float(loss) - float(gold["loss"])) < 0.05


@pytest.mark.gpu
def test_engine_fp8_producer_images_follow_the_oracle_after_the_calibration_pass(gpu_lib):
    """vit_fp8 = 31: the first pass after a pack runs the per-token quantisers and collects the tensor maxima; from the second pass on the
    out_proj / c_fc / c_proj epilogues write the e4m3 images themselves (per-tensor power-of-two scales of the previous pass) and the hidden
    tensor exists in e4m3 only.  Block by block on the engine's own block input against the oracle with the same quantisation points."""
    from dist_oracle import Oracle
    g, eng, sd, video, text, tgt = _engine("b16_8+16f", 2, 31)
    eng.forward_backward(video, text, tgt)                          # calibration pass (per-token quantisers)
    cal = [eng.debug(f"feat.{i}").clone().cpu().float() for i in range(g.layers)]
    loss, logits = eng.forward_backward(video, text, tgt)           # producers' images
    feats = [eng.debug(f"feat.{i}").clone().cpu().float() for i in range(g.layers)]
    assert not torch.equal(cal[5], feats[5])                        # the second pass really took the other path
    o8, o16 = Oracle(g, sd, dtype=torch.float32, bf16=True, vit_fp8=31), Oracle(g, sd, dtype=torch.float32, bf16=True)
    for i in (1, 6, 11):
        x_in = feats[i - 1].reshape(2, g.t, g.L, g.d)
        with torch.no_grad():
            r8, r16 = o8.vit_block(x_in, i), o16.vit_block(x_in, i)
        e8, e16, q = _rel(feats[i], r8), _rel(feats[i], r16), _rel(r8, r16)
        print(f"fp8 images block {i}: engine vs fp8 oracle {e8:.4f}, engine vs bf16 oracle {e16:.4f}, fp8 oracle vs bf16 oracle {q:.4f}")
        assert e8 < 0.012 and e8 < 0.6 * q and e16 > 0.8 * q, (i, e8, e16, q)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "b16_b2.npz"))
    gap32 = float((logits.cpu().double() - torch.from_numpy(gold["logits"]).double()).abs().max())
    print(f"fp8 images: logits vs fp32 reference golden {gap32:.4f}")
    assert gap32 < 0.35 and (logits.cpu().argmax(1) == torch.from_numpy(gold["logits"]).argmax(1)).all()
    loss3, logits3 = eng.forward_backward(video, text, tgt)         # third pass: scales from the second - same batch, same powers of two
    assert torch.equal(logits3, logits)


@pytest.mark.gpu
def test_engine_fp8_images_on_the_vit_l14_geometry(gpu_lib):
    """BASELINE config 5's tower (ViT-L/14: width 1024, 16 heads, 257 tokens per frame, 24 blocks) at T = 8, b = 1 (1028 token rows - just
    above the fp8 kernel's minimum): vit_fp8 = 31 after its calibration pass, two blocks against the oracle on the engine's own block inputs."""
    from dist_oracle import Oracle
    g, eng, sd, video, text, tgt = _engine("l14_tiny_t", 1, 31)
    eng.forward_backward(video, text, tgt)
    loss, logits = eng.forward_backward(video, text, tgt)
    feats = {i: eng.debug(f"feat.{i}").clone().cpu().float() for i in (2, 3, 22, 23)}
    o8, o16 = Oracle(g, sd, dtype=torch.float32, bf16=True, vit_fp8=31), Oracle(g, sd, dtype=torch.float32, bf16=True)
    for i in (3, 23):
        x_in = feats[i - 1].reshape(1, g.t, g.L, g.d)
        with torch.no_grad():
            r8, r16 = o8.vit_block(x_in, i), o16.vit_block(x_in, i)
        e8, e16, q = _rel(feats[i], r8), _rel(feats[i], r16), _rel(r8, r16)
        print(f"L/14 fp8 images block {i}: engine vs fp8 oracle {e8:.4f}, engine vs bf16 oracle {e16:.4f}, fp8 oracle vs bf16 oracle {q:.4f}")
        assert e8 < 0.012 and e8 < 0.6 * q and e16 > 0.8 * q, (i, e8, e16, q)
    g0, eng0, _, _, _, _ = _engine("l14_tiny_t", 1, 0)
    loss0, logits0 = eng0.forward_backward(video, text, tgt)
    gap = float((logits.float() - logits0.float()).abs().max())
    print(f"L/14 fp8 images: logits vs bf16 engine {gap:.4f}")
    assert 0 < gap < 0.3 and torch.isfinite(eng.grads).all() and abs(float(loss) - float(loss0)) < 0.05


@pytest.mark.gpu
def test_engine_fp8_mode_falls_back_to_bf16_for_shapes_the_kernel_does_not_take(gpu_lib):
    """tiny geometry (width 128: K below the fp8 kernel's 256): the mode is accepted and every GEMM runs in bf16 - bit-identical results"""
    g, eng, sd, video, text, tgt = _engine("tiny", 2, 31)
    _, logits = eng.forward_backward(video, text, tgt)
    _, logits = eng.forward_backward(video, text, tgt)
    g0, eng0, _, _, _, _ = _engine("tiny", 2, 0)
    _, logits0 = eng0.forward_backward(video, text, tgt)
    assert torch.equal(logits, logits0)


# ---- producer side: DIST_EPI_OUT8 (e4m3 image of a GEMM's output with a per-tensor scale) ------------------------------------------
def _e4m3_image(v_bf16, scale):
    """the header's statement: e4m3_rne(clamp(float(bf16(v)) / scale, -448, 448)) as bytes"""
    return (v_bf16.float().cpu() / float(scale)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


@pytest.mark.gpu
def test_scale_update_and_amax_ops(gpu_lib):
    from dist_amd import ops
    amax = torch.tensor([0.0, 448.0, 449.0, 1.0, 3.3e4, 112.0], device="cuda")
    scale = torch.zeros(6, device="cuda")
    ops.fp8_scale_update(amax, scale, 4.0)
    want = [2.0 ** -30, 4.0, 8.0, 2.0 ** -6, 512.0, 1.0]                       # smallest power of two >= max(amax, 2^-24) * 4 / 448
    got = scale.cpu().tolist()
    assert got == want and float(amax.abs().sum()) == 0                      # (an all-zero tensor: a floor whose square is a normal fp32 number)
    x = rnd((1000, 768), 5, 3.0)
    a = torch.zeros(1, device="cuda")
    ops.amax_(x, a)
    assert float(a) == float(x.float().abs().max())
    ops.amax_(x * 0.5, a)
    assert float(a) == float(x.float().abs().max())                            # a running maximum


@pytest.mark.gpu
@pytest.mark.parametrize("fp8_in", [False, True])
def test_out8_image_of_the_output(gpu_lib, fp8_in):
    """C8 beside C (residual stream: bias + residual + row statistics) and C8 instead of everything (QuickGELU'd hidden tensor), from a bf16
    and from an e4m3 GEMM; the collected maximum; then the image consumed by the next GEMM with its one scalar scale."""
    from dist_amd import ops
    M, N, K = 197 * 16, 1024, 768
    A, W = rnd((M, K), 51, 2.0), rnd((N, K), 52, K ** -0.5)
    bias, R = rnd((N,), 53, 1.0, torch.float32), rnd((M, N), 54)
    kw = {}
    Ain, Win = A, W
    if fp8_in:
        (Ain, sa), (Win, sw) = ops.quant_rows_fp8(A), ops.quant_rows_fp8(W)
        kw = dict(fp8=(sa, sw))
    scale, amax = torch.tensor([0.25], device="cuda"), torch.zeros(1, device="cuda")
    C = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    C8 = torch.zeros(M, N, dtype=torch.uint8, device="cuda")
    part = torch.zeros(N // 64, M, 2, device="cuda")
    ops.gemm_nt(Ain, Win, M, N, K, bias=bias, res=R, C_out=C, rowstats=part, out8=(C8, scale, amax), **kw)
    assert torch.equal(C8.cpu(), _e4m3_image(C, 0.25))                        # the image of exactly the stored bf16 values
    assert float(amax) == float(C.float().abs().max())
    Cs = C.float().view(M, N // 64, 64)
    torch.testing.assert_close(part[..., 0].t(), Cs.sum(-1), rtol=1e-4, atol=1e-3)   # the row statistics saw the bf16 tile, not the image
    # activation-only, e4m3 only
    H8 = torch.zeros(M, N, dtype=torch.uint8, device="cuda")
    amax.zero_()
    ops.gemm_nt(Ain, Win, M, N, K, bias=bias, out8=(H8, scale, amax), act_only8=True, **kw)
    H = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(Ain, Win, M, N, K, bias=bias, C2_out=H, **kw)                # the bf16 activated output of the same GEMM
    assert torch.equal(H8.cpu(), _e4m3_image(H, 0.25)) and float(amax) == float(H.float().abs().max())
    # consumer: the image as the A operand of an e4m3 GEMM, one scalar scale for all rows
    W2 = rnd((256, N), 55, N ** -0.5)
    qw2, sw2 = ops.quant_rows_fp8(W2)
    Y = torch.empty(M, 256, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(H8, qw2, M, 256, N, C_out=Y, fp8=(scale, sw2))
    ref = (H8.cpu().view(torch.float8_e4m3fn).double() * 0.25) @ fo.dequant(qw2.cpu(), sw2.cpu()).t()
    torch.testing.assert_close(Y.cpu().double(), ref, rtol=1.2e-2, atol=1.2e-2)
    # a saturating scale: values beyond 448 * scale clamp instead of becoming NaN
    tiny = torch.tensor([2.0 ** -12], device="cuda")
    ops.gemm_nt(Ain, Win, M, N, K, bias=bias, out8=(H8, tiny, None), act_only8=True, **kw)
    img = H8.cpu().view(torch.float8_e4m3fn).float()
    assert not torch.isnan(img).any() and float(img.max()) == 448.0


@pytest.mark.gpu
@pytest.mark.parametrize("frames,L_,heads", [(16, 197, 12), (6, 257, 16), (3, 50, 4)])
def test_attention_writes_its_output_as_e4m3(gpu_lib, frames, L_, heads):
    """dist_op_attention_out8: the image of exactly the bf16 rows dist_op_attention stores, with the caller's per-tensor scale; running maximum"""
    from dist_amd import ops, lib as L
    qkv = rnd((frames * heads * 3 * L_, 64), 61, 1.0)
    o = ops.attention(qkv, frames, L_, heads, layout=L.QKV_HEADS)
    scale, amax = torch.tensor([2.0 ** -6], device="cuda"), torch.zeros(1, device="cuda")
    o8 = ops.attention_out8(qkv, frames, L_, heads, scale, amax)
    assert torch.equal(o8.cpu(), _e4m3_image(o, 2.0 ** -6))
    assert float(amax) == float(o.float().abs().max())
    o8b = ops.attention_out8(qkv, frames, L_, heads, torch.tensor([2.0 ** -14], device="cuda"), None)      # saturates, no NaN
    img = o8b.cpu().view(torch.float8_e4m3fn).float()
    assert not torch.isnan(img).any() and float(img.abs().max()) == 448.0


@pytest.mark.gpu
@pytest.mark.parametrize("frames,L_,heads", [(12, 197, 12), (5, 257, 16)])
def test_qkv_as_e4m3_from_the_gemm_through_the_attention(gpu_lib, frames, L_, heads):
    """in_proj writes q | k | v head-major as e4m3 ONLY (DIST_EPI_OUT8 with DIST_OM_HEADS), dist_op_attention_fp8 widens the bytes to bf16
    on their way into LDS: with a power-of-two scale the result is bit-identical to dist_op_attention on the dequantised bf16 tensor."""
    from dist_amd import ops, lib as L
    M, d = frames * L_, heads * 64
    K = 768 if d == 768 else 1024
    A, W = rnd((M, K), 71, 1.5), rnd((3 * d, K), 72, K ** -0.5)
    bias = rnd((3 * d,), 73, 1.0, torch.float32)
    qa, sa = ops.quant_rows_fp8(A)
    qw, sw = ops.quant_rows_fp8(W)
    qkv16 = torch.empty(frames * heads * 3 * L_, 64, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(qa, qw, M, 3 * d, K, bias=bias, C_out=qkv16, ldc=64, omap=ops.outmap(L.OM_HEADS, L_, heads), fp8=(sa, sw))
    scale, amax = torch.tensor([2.0 ** -5], device="cuda"), torch.zeros(1, device="cuda")
    qkv8 = torch.zeros(frames * heads * 3 * L_, 64, dtype=torch.uint8, device="cuda")
    ops.gemm_nt(qa, qw, M, 3 * d, K, bias=bias, omap=ops.outmap(L.OM_HEADS, L_, heads), out8=(qkv8, scale, amax), fp8=(sa, sw))
    assert torch.equal(qkv8.cpu(), _e4m3_image(qkv16, 2.0 ** -5)) and float(amax) == float(qkv16.float().abs().max())
    deq = (qkv8.cpu().view(torch.float8_e4m3fn).float() * 2.0 ** -5).to(torch.bfloat16).cuda()       # exact in bf16
    want = ops.attention(deq, frames, L_, heads, layout=L.QKV_HEADS)
    got = ops.attention_fp8(qkv8, scale, frames, L_, heads)
    assert torch.equal(got, want)
    s2 = torch.tensor([2.0 ** -7], device="cuda")
    got8 = ops.attention_fp8(qkv8, scale, frames, L_, heads, out8_scale=s2, amax=amax.zero_())
    assert torch.equal(got8.cpu(), _e4m3_image(want, 2.0 ** -7)) and float(amax) == float(want.float().abs().max())
