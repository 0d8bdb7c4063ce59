"""GPU parity for the BASELINE configurations beyond the headline one (round 3; VERDICT r02 "Missing 1 / 5", "Weak 1 / 2").

  * ViT-L/14 geometry (config 4 / 5's tower and branch sizes) against the reference's own outputs: tests/golden/l14_t8.npz (T = 8)
    and tests/golden/l14_t64_b1.npz (the real 32+64f frame counts, one clip) - HIP engine in fp32 parity mode at north_star's
    rtol 1e-3 / atol 1e-4 on the logits, 2e-3 on every gradient norm, and the saved activations' checksums; bf16 mode bounded
    by measured gaps.
  * ViT-B/16 16+32f (config 3) the same way against tests/golden/b16_t32_b1.npz.
  * The FULL bench batches of configs 3, 4 and 5 (b = 32 / 8 / 16 with vit_fp8 = 31) through size-independent properties: the
    golden clip sits in row 0 of the batch and a clip's logits do not depend on the clips it is batched with, so row 0 must
    reproduce the one-clip run bit for bit ... up to tile-edge summation order; repeatability; finite, non-zero gradients.
  * bf16 fast path at a size where the LDS-DMA GEMM, the LayerNorm fold and the ROWSTATS statistics run (ViT-B/16 8+16f, b = 2)
    against the oracle with the SAME bf16 rounding points, tensor by tensor.

Every gate below is written as (measured on MI355X, gate ~ 2x measured); the measured values of the last run are dumped to
gpurun_out/parity_gaps.json so the gates can be re-derived.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _gaps import record  # noqa: E402

_SD_CACHE = {}


def _state_dict(g):
    """procedural weights, cached per geometry (ViT-L/14: 343 M values hashed on the host take ~15 s)"""
    from dist_amd import synth
    if g.name not in _SD_CACHE:
        _SD_CACHE.clear()                      # one geometry at a time: the L/14 dictionary is 1.4 GB
        _SD_CACHE[g.name] = synth.state_dict(g)
    return _SD_CACHE[g.name]


def build(gname, b, dtype, vit_fp8=0, batch_seeds=None):
    """engine + procedural inputs; with `batch_seeds` clip i of the batch is synth.video(g, 1, seed) - clip 0 is always the golden clip
    (seed 1), and the host never holds more than one clip's generator temporaries"""
    from dist_amd import synth
    from dist_amd.engine import Engine, config_from_geometry
    g = synth.geometry(gname)
    eng = Engine(config_from_geometry(g, b, dtype, True, vit_fp8))
    sd = _state_dict(g)
    eng.load_state_dict(sd)
    if batch_seeds is None:
        video = torch.from_numpy(synth.video(g, b)).cuda()
    else:
        video = torch.empty(b, 3, g.T, g.res, g.res, device="cuda")
        for i, s in enumerate(batch_seeds):
            video[i].copy_(torch.from_numpy(synth.video(g, 1, seed=s))[0])
    text = torch.from_numpy(synth.text_features(g)).cuda()
    tgt = torch.from_numpy(synth.soft_target(g, b)[0]).cuda()
    return g, eng, sd, video, text, tgt


def checksum_gaps(eng, gold, g, b):
    """engine activations against the golden checksums (mean, std, max-abs, first 16 values of the build's layout)"""
    worst_first, worst_stat = 0.0, 0.0
    for k in gold.files:
        if not k.startswith("act."):
            continue
        v = eng.debug(k[4:]).double().cpu().flatten()
        ref = gold[k]
        scale = float(np.abs(ref[3:]).max()) + 1e-9
        worst_first = max(worst_first, float(np.abs(v[:16].numpy() - ref[3:]).max()) / max(scale, float(ref[1])))
        worst_stat = max(worst_stat, abs(float(v.mean()) - ref[0]) / (float(ref[1]) + 1e-9), abs(float(v.std()) - ref[1]) / (float(ref[1]) + 1e-9))
    return worst_first, worst_stat


def gnorm_gaps(eng, gold):
    bad, worst = [], 0.0
    for k in gold.files:
        if k.startswith("gnorm."):
            n = k[6:]
            got, ref = float(eng.view(n, grad=True).double().norm()), float(gold[k])
            err = abs(got - ref) / (ref + 1e-12)
            worst = max(worst, err)
            bad.append((err, n, got, ref))
    bad.sort(reverse=True)
    return worst, bad


# ---------------------------------------------------------------------------------------------------------------------
# fp32 parity mode against the reference's outputs, per geometry
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gname,fname", [("l14_tiny_t", "l14_t8"), ("b16_16+32f", "b16_t32_b1"), ("l14_32+64f", "l14_t64_b1")])
def test_fp32_engine_vs_reference_golden(gpu_lib, gname, fname):
    """north_star's bar (logits rtol 1e-3 / atol 1e-4) on the geometries of BASELINE configs 3 and 4 / 5, one clip, exact-fp32 kernels."""
    g, eng, sd, video, text, tgt = build(gname, 1, torch.float32)
    loss, logits = eng.forward_backward(video, text, tgt)
    gold = np.load(os.path.join(GOLD, fname + ".npz"))
    ref = torch.from_numpy(gold["logits"]).double()
    record(f"fp32.{fname}.logits_maxabs", (logits.cpu().double() - ref).abs().max())
    torch.testing.assert_close(logits.cpu().double(), ref, rtol=1e-3, atol=1e-4)
    assert abs(float(loss) - float(gold["loss"])) < 1e-4
    worst, bad = gnorm_gaps(eng, gold)
    record(f"fp32.{fname}.gnorm_worst", worst)
    assert worst <= 2e-3, bad[:5]
    assert len(bad) == int(gold["n_grad_tensors"])
    wf, ws = checksum_gaps(eng, gold, g, 1)
    record(f"fp32.{fname}.act_first16", wf); record(f"fp32.{fname}.act_stats", ws)
    assert wf < 1e-3 and ws < 1e-3, (wf, ws)
    last = g.layers - 1
    assert float(eng.view(f"dist_net.integration2temporal_nets.{last}.linear_fuse.weight", grad=True).abs().max()) == 0.0


# measured on MI355X (round 3, profiles/r03_parity_gaps.json) -> gate ~ 2x measured.  The logits of these geometries span +-1.6 ... 1.8.
BF16_GATES = {
    #  name        : (logits max-abs, loss abs, worst gradient-norm rel)      measured
    "l14_t8":      (0.026, 0.004, 0.017),                                    # 0.0128, 0.0002, 0.0081
    "b16_t32_b1":  (0.027, 0.015, 0.016),                                    # 0.0134, 0.0072, 0.0076
    "l14_t64_b1":  (0.022, 0.005, 0.011),                                    # 0.0108, 0.0023, 0.0053
}


@pytest.mark.parametrize("gname,fname", [("l14_tiny_t", "l14_t8"), ("b16_16+32f", "b16_t32_b1"), ("l14_32+64f", "l14_t64_b1")])
def test_bf16_engine_vs_reference_golden(gpu_lib, gname, fname):
    """the performance mode (bf16 storage, fp32 accumulation) on the same geometries against the fp32 reference: the honest
    bf16-vs-fp32 gap, gated at about twice what was measured; arg-max identical."""
    g, eng, sd, video, text, tgt = build(gname, 1, torch.bfloat16)
    loss, logits = eng.forward_backward(video, text, tgt)
    gold = np.load(os.path.join(GOLD, fname + ".npz"))
    ref = torch.from_numpy(gold["logits"]).double()
    gap = record(f"bf16.{fname}.logits_maxabs", (logits.cpu().double() - ref).abs().max())
    lgap = record(f"bf16.{fname}.loss_abs", abs(float(loss) - float(gold["loss"])))
    worst, bad = gnorm_gaps(eng, gold)
    record(f"bf16.{fname}.gnorm_worst", worst)
    record(f"bf16.{fname}.logit_range", ref.abs().max())
    print(f"{fname} bf16: logits gap {gap:.4f} (range +-{float(ref.abs().max()):.2f}), loss gap {lgap:.5f}, worst gradient norm {worst:.4f} {bad[0][1]}")
    gl, gs, gg = BF16_GATES[fname]
    assert gap < gl and lgap < gs and worst < gg, (gap, lgap, bad[:5])
    assert (logits.cpu().argmax(1) == ref.argmax(1)).all()
    assert torch.isfinite(eng.grads).all()


# ---------------------------------------------------------------------------------------------------------------------
# the full bench batches of configs 3, 4, 5 through size-independent properties
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("gname,fname,b,fp8", [("b16_16+32f", "b16_t32_b1", 32, 0), ("l14_32+64f", "l14_t64_b1", 8, 0), ("l14_32+64f", "l14_t64_b1", 16, 31)])
def test_full_batch_of_configs_3_4_5(gpu_lib, gname, fname, b, fp8):
    """BASELINE config 3 (ViT-B/16 16+32f, b = 32), 4 (ViT-L/14 32+64f, b = 8) and 5 (the same with the e4m3 spatial branch, b = 16) at
    their full per-GPU batch.  Row 0 is the clip the reference golden was computed on."""
    from dist_amd import synth
    seeds = [1] + [100 + 7 * i for i in range(1, b)]
    g, eng, sd, video, text, tgt = build(gname, b, torch.bfloat16, fp8, batch_seeds=seeds)
    assert np.array_equal(synth.video(g, 1), video[:1].cpu().numpy())
    passes = 2 if fp8 else 1                                         # vit_fp8 & 16: the first pass after a pack calibrates the per-tensor scales
    for _ in range(passes):
        loss, logits = eng.forward_backward(video, text, tgt)
    logits, grads = logits.clone(), eng.grads.clone()
    assert torch.isfinite(logits).all() and torch.isfinite(grads).all() and float(grads.abs().max()) > 0 and np.isfinite(float(loss))
    # (1) against the reference's golden for the clip in row 0
    gold = np.load(os.path.join(GOLD, fname + ".npz"))
    ref = torch.from_numpy(gold["logits"]).double()
    gap = record(f"full.{fname}.b{b}.fp8_{fp8}.row0_vs_golden", (logits[:1].cpu().double() - ref).abs().max())
    assert gap < (0.17 if fp8 else 0.03), gap                       # measured 0.083 (e4m3 tower) / 0.0134, 0.0108 (bf16)
    assert int(logits[0].argmax()) == int(ref[0].argmax())
    # (2) batch invariance: the one-clip engine gives the same row (same kernels; only tile-edge summation order may differ).
    # The per-tensor e4m3 scales of image mode are statistics of the whole batch, so config 5 is compared in per-token mode (15).
    if not (fp8 & 16):
        g1, e1, _, _, _, _ = build(gname, 1, torch.bfloat16, fp8)
        l1, lg1 = e1.forward_backward(video[:1].contiguous(), text, tgt[:1].contiguous())
        inv = record(f"full.{fname}.b{b}.fp8_{fp8}.row0_vs_b1", (logits[:1].float() - lg1.float()).abs().max())
        assert inv <= 1e-6, inv                                      # measured 0.0: the same bits
        del e1
    # (3) repeatable: logits bit for bit, gradients up to fp32 atomic ordering
    loss_b, logits_b = eng.forward_backward(video, text, tgt)
    if fp8 & 16:
        # image mode takes its per-tensor power-of-two scales from the PREVIOUS pass's maxima: the pass after the calibration pass and
        # the pass after that may differ in a scale (DESIGN "fp8 frozen spatial branch": a clip's features depend on what ran before
        # it); from then on the same batch reproduces itself bit for bit
        rep = record(f"full.{fname}.b{b}.fp8_{fp8}.pass3_vs_pass2", (logits_b - logits).abs().max())
        assert rep < 0.05, rep
        logits, grads = logits_b.clone(), eng.grads.clone()
        loss_b, logits_b = eng.forward_backward(video, text, tgt)
    assert torch.equal(logits_b, logits)
    rel = float((eng.grads - grads).abs().max() / grads.abs().max())
    record(f"full.{fname}.b{b}.fp8_{fp8}.grad_repeat_rel", rel)
    assert rel < 5e-5, rel                                           # measured 1.4e-5 / 5e-6 (fp32 atomics' order)
    # (4) every tensor that must receive a gradient did; the optimizer moves the weights
    zero = [n for n, (off, shape, _) in eng.tables[0].items()
            if float(eng.view(n, grad=True).abs().max()) == 0.0 and f"integration2temporal_nets.{g.layers - 1}." not in n]
    assert not zero, zero[:5]
    before = eng.theta.clone()
    eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0)
    torch.cuda.synchronize()
    assert float((eng.theta - before).abs().max()) > 0


def test_config5_batch_invariance_in_per_token_mode(gpu_lib):
    """config 5's e4m3 tower with per-token scales (vit_fp8 = 15): a clip's logits do not depend on its batch (b = 4 vs b = 1)"""
    g, eng, sd, video, text, tgt = build("l14_32+64f", 4, torch.bfloat16, 15, batch_seeds=[1, 107, 114, 121])
    _, logits = eng.forward_backward(video, text, tgt)
    logits = logits.clone()
    g1, e1, _, _, _, _ = build("l14_32+64f", 1, torch.bfloat16, 15)
    _, lg1 = e1.forward_backward(video[:1].contiguous(), text, tgt[:1].contiguous())
    inv = record("full.l14_t64.fp8_15.row0_vs_b1", (logits[:1].float() - lg1.float()).abs().max())
    assert inv <= 1e-6, inv
    gold = np.load(os.path.join(GOLD, "l14_t64_b1.npz"))
    gap = record("full.l14_t64.fp8_15.row0_vs_golden", (logits[:1].cpu().double() - torch.from_numpy(gold["logits"]).double()).abs().max())
    assert gap < 0.17                                                # measured 0.083


# ---------------------------------------------------------------------------------------------------------------------
# bf16 fast path against the oracle with the same rounding points, at a size where the fast kernels run
# ---------------------------------------------------------------------------------------------------------------------
def _rel(a, b):
    a, b = a.double().cpu().reshape(-1), torch.as_tensor(b).double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


# (measured, gate): relative L2 error of one ViT block on the ENGINE's own block input; of the branch tensors end to end
#   measured (round 4, Oracle(bf16=True, fused=True): the fused kernels' rounding points): vit_block 0.0017, stem 5e-5, branch_act 0.0110,
#   logits 0.0107 (range +-6), loss 0.0006, worst gradient 0.0216 of its tensor's maximum = dist_net.temporal_stem.weight, then
#   temporal_nets.0.ln.weight, the T2I cls tokens of the last layers, temporal_nets.0/1 (gpurun_out/same_rounding_grad_worst.json ->
#   profiles/r04_parity_gaps.json): the tensors at the far END of the backward chain, i.e. twelve layers of bf16-stored gradients, not one
#   defective tensor; median 0.005.  With the unfused sequence's rounding points (round 3's oracle) the same run reads 0.0125 / 0.0111 / 0.035.
#   Every gate is 2x its measured value.  Noise floor: the oracle in fp32 against ITSELF in fp64 (same rounding points) differs by logits 0.0099,
#   activations 0.0107, worst gradient 0.027 (tools/oracle_rounding_noise.py -> profiles/r04_oracle_rounding_noise.json): the engine sits on it.
SAME_ROUNDING_GATES = {"vit_block": 3.5e-3, "stem": 1e-4, "branch_act": 0.022, "logits": 0.0215, "loss": 1.3e-3, "grad": 0.043}


def test_b16_bf16_fast_path_vs_oracle_with_the_same_rounding_points(gpu_lib):
    """ViT-B/16 8+16f, b = 2 (3152 token rows: the 256x256 LDS-DMA GEMM, the LayerNorm fold and the row statistics from the producing
    GEMM all run; the branch GEMMs run their LDS-DMA variants).  Oracle(bf16=True) rounds to bf16 wherever the kernels store bf16."""
    from dist_oracle import Oracle
    g, eng, sd, video, text, tgt = build("b16_8+16f", 2, torch.bfloat16)
    loss, logits = eng.forward_backward(video, text, tgt)
    feats = [eng.debug(f"feat.{i}").clone().cpu().float() for i in range(g.layers)]
    # fused=True: the rounding points of the FUSED kernels (LayerNorm fold in the ViT GEMMs; xhat / bf16(W diag gamma) / one c_proj accumulation in
    # the IntegrationNetwork; bf16 gradient tensors of the backward chain) - oracle/dist_oracle.py header.  The unfused points are measured beside
    # it (record only) to show what the flag buys.
    o = Oracle(g, sd, dtype=torch.float32, bf16=True, fused=True)
    # ---- frozen ViT, block by block on the engine's own input of the block (errors cannot hide in accumulated drift)
    worst = 0.0
    for i in (1, 4, 7, 11):
        x_in = feats[i - 1].reshape(2, g.t, g.L, g.d)
        with torch.no_grad():
            r = o.vit_block(x_in, i)
        e = _rel(feats[i], r)
        worst = max(worst, e)
        print(f"bf16 fast path, ViT block {i}: engine vs same-rounding oracle rel-L2 {e:.5f}")
    record("same_rounding.vit_block_worst", worst)
    assert worst < SAME_ROUNDING_GATES["vit_block"], worst
    # ---- whole step against the oracle's own forward / backward
    from dist_amd import synth
    ref = o.forward_backward(synth.video(g, 2), synth.text_features(g), synth.soft_target(g, 2)[0])
    worst_act, which = 0.0, ""
    for i in range(g.layers):
        # (M' of the earlier layers is never materialised by the fused kernels: dist_debug_tensor refuses "mid.i" for them)
        for name in (f"tn_out.{i}", f"int_out.{i}", f"x_temporal.{i}") + ((f"mid.{i}",) if i == g.layers - 1 else ()):
            e = _rel(eng.debug(name), ref["keep"][name].detach())
            if e > worst_act:
                worst_act, which = e, name
    e = _rel(eng.debug("stem"), ref["keep"]["stem"].detach())
    record("same_rounding.stem", e)
    record("same_rounding.branch_act_worst", worst_act)
    print(f"bf16 fast path: stem {e:.5f}, worst branch activation {which} {worst_act:.5f}")
    assert e < SAME_ROUNDING_GATES["stem"] and worst_act < SAME_ROUNDING_GATES["branch_act"], (e, which, worst_act)
    lgap = record("same_rounding.logits_maxabs", (logits.cpu().double() - ref["logits"].detach().double()).abs().max())
    sgap = record("same_rounding.loss_abs", abs(float(loss) - float(ref["loss"])))
    assert lgap < SAME_ROUNDING_GATES["logits"] and sgap < SAME_ROUNDING_GATES["loss"], (lgap, sgap)
    errs = []
    for n, gr in ref["grads"].items():
        if gr.abs().max() < 1e-6:
            continue
        got = eng.view(n, grad=True)
        errs.append((float((got.double().cpu() - gr.double()).abs().max() / (gr.abs().max() + 1e-12)), n))
    errs.sort(reverse=True)
    record("same_rounding.grad_worst_relmax", errs[0][0])
    record("same_rounding.grad_median_relmax", errs[len(errs) // 2][0])
    record("same_rounding.grad_worst_rel_l2", max(_rel(eng.view(n, grad=True), ref["grads"][n]) for _, n in errs))
    import json
    try:      # which tensors sit behind the worst gaps ("nobody can tell whether that is rounding or a defect in one tensor")
        json.dump({"worst_relmax": [[n, e] for e, n in errs[:12]]}, open(os.path.join(os.path.dirname(GOLD), "..", "gpurun_out", "same_rounding_grad_worst.json"), "w"), indent=1)
    except OSError:
        pass
    print(f"bf16 fast path: logits gap {lgap:.4f}, loss gap {sgap:.5f}, worst gradients {errs[:5]}, median {errs[len(errs) // 2][0]:.4f}")
    assert errs[0][0] < SAME_ROUNDING_GATES["grad"], errs[:5]
    # the unfused rounding points on the same engine run, for the record (no gate): what `fused=True` changes
    o2 = Oracle(g, sd, dtype=torch.float32, bf16=True)
    ref2 = o2.forward_backward(synth.video(g, 2), synth.text_features(g), synth.soft_target(g, 2)[0])
    record("same_rounding.unfused_points.logits_maxabs", (logits.cpu().double() - ref2["logits"].detach().double()).abs().max())
    record("same_rounding.unfused_points.branch_act_worst", max(_rel(eng.debug(nm), ref2["keep"][nm].detach()) for i in range(g.layers) for nm in (f"tn_out.{i}", f"int_out.{i}", f"x_temporal.{i}")))
    record("same_rounding.unfused_points.grad_worst_relmax", max(float((eng.view(n, grad=True).double().cpu() - gr.double()).abs().max() / (gr.abs().max() + 1e-12))
                                                                for n, gr in ref2["grads"].items() if gr.abs().max() >= 1e-6))


# ---------------------------------------------------------------------------------------------------------------------
# round 5 (VERDICT r04 "Missing 3"): engine-vs-oracle gradients at sizes where the bench's weight-gradient kernels are SELECTED
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_threads():
    """the torch-CPU oracle on a 256-core host: 32 threads (every small operator forks / joins; all cores is the slowest setting)"""
    old = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 32))
    return old


# measured on MI355X (round 5, profiles/r05_parity_gaps.json): logits 0.0135 (range +-6), loss 0.00029, worst gradient 0.0266 of its tensor's maximum
# (dist_net.temporal_stem.weight, then the T2I cls token of layer 10 and temporal_nets.0.ln.weight: the far end of the backward chain, as at b = 2),
# median 0.0049; the 60 gradients of the ring kernel: worst 0.0080.  Gates = 2x measured.
B8_GATES = {"logits": 0.027, "loss": 6e-4, "grad": 0.053, "grad_median": 0.010, "ring": 0.016}


def test_b16_bf16_b8_every_gradient_vs_oracle_with_the_same_rounding_points(gpu_lib):
    """ViT-B/16 8+16f, b = 8: 12 608 token rows >= 8192, so dist_op_gemm_tn takes gemm_tn8r_kernel (the two-group LDS-DMA ring kernel) for the
    three large plain weight gradients per layer, both weight-gradient kernels run under their 96-block grid caps and on both weight-gradient
    streams - the kernels of the bench step, which the b = 2 test above (3152 rows) never selects.  EVERY gradient tensor against
    Oracle(bf16=True, fused=True) (autograd over the restated reference graph, rounding where the kernels store bf16: runs/train.py:110)."""
    from dist_oracle import Oracle
    from dist_amd import synth
    b = 8
    g, eng, sd, video, text, tgt = build("b16_8+16f", b, torch.bfloat16)
    loss, logits = eng.forward_backward(video, text, tgt)
    logits = logits.clone()
    old = _oracle_threads()
    try:
        o = Oracle(g, sd, dtype=torch.float32, bf16=True, fused=True)
        ref = o.forward_backward(synth.video(g, b), synth.text_features(g), synth.soft_target(g, b)[0])
    finally:
        torch.set_num_threads(old)
    lgap = record("b8_same_rounding.logits_maxabs", (logits.cpu().double() - ref["logits"].detach().double()).abs().max())
    sgap = record("b8_same_rounding.loss_abs", abs(float(loss) - float(ref["loss"])))
    errs = []
    for n, gr in ref["grads"].items():
        if gr.abs().max() < 1e-6:
            continue
        got = eng.view(n, grad=True)
        errs.append((float((got.double().cpu() - gr.double()).abs().max() / (gr.abs().max() + 1e-12)), n))
    errs.sort(reverse=True)
    record("b8_same_rounding.grad_worst_relmax", errs[0][0])
    record("b8_same_rounding.grad_median_relmax", errs[len(errs) // 2][0])
    record("b8_same_rounding.n_grad_tensors", len(errs))
    print(f"b = 8 bf16 vs same-rounding oracle: logits gap {lgap:.4f}, loss gap {sgap:.5f}, worst gradients {errs[:6]}, median {errs[len(errs) // 2][0]:.4f}")
    import json
    try:
        json.dump({"worst_relmax": [[n, e] for e, n in errs[:12]]}, open(os.path.join(ROOT, "gpurun_out", "b8_same_rounding_grad_worst.json"), "w"), indent=1)
    except OSError:
        pass
    assert len(errs) >= 370, len(errs)                               # 381 gradient tensors, the last layer's I2T pair receives none
    assert lgap < B8_GATES["logits"] and sgap < B8_GATES["loss"], (lgap, sgap)
    assert errs[0][0] < B8_GATES["grad"], errs[:6]
    assert errs[len(errs) // 2][0] < B8_GATES["grad_median"], errs[len(errs) // 2]
    # the large plain gradients the ring kernel computes, by name: input_linears.i (384 x 768), the integration c_proj pair, [ffn.c_fc ; temporal_ffn.c_fc1]
    def ring_kernel(n):
        if n.startswith("dist_net.input_linears.") and n.endswith(".weight"):
            return True
        return n.startswith("dist_net.integration_nets.") and n.endswith((".ffn.c_fc.weight", ".ffn.c_proj.weight", ".temporal_ffn.c_fc1.weight", ".temporal_ffn.c_proj.weight"))
    big = [(e, n) for e, n in errs if ring_kernel(n)]
    record("b8_same_rounding.ring_kernel_grad_worst_relmax", max(big)[0])
    assert len(big) == 5 * g.layers and max(big)[0] < B8_GATES["ring"], max(big)


# measured (round 5, profiles/r05_parity_gaps.json): worst gradient norm 0.42 % (temporal_nets.6.ln.weight), median 0.08 %, logits 0.0146 -> gates = 2x measured
# round 6, element-wise: relative L2 distance of every gradient tensor, worst 2.56 % (temporal_stem.weight), median 0.82 % -> gates = 2x measured
B32_GATES = {"gnorm_worst": 0.0085, "gnorm_median": 0.0016, "logits": 0.03, "rel_l2_worst": 0.052, "rel_l2_median": 0.0165}


def test_b16_bf16_b32_gradient_norms_vs_fp32_oracle(gpu_lib):
    """The bench workload itself (BASELINE config 2: ViT-B/16 8+16f, b = 32, bf16): the norm of EVERY gradient tensor against the fp32 oracle
    (plain restatement of the reference, no bf16 rounding anywhere: autograd of runs/train.py:110) - the honest bf16-vs-fp32 gap of the step the
    bench times, with every bench-size kernel selected (gemm_tn8r_kernel, capped grids, two weight-gradient streams, the fused branch kernels)."""
    from dist_oracle import Oracle
    from dist_amd import synth
    b = 32
    g, eng, sd, video, text, tgt = build("b16_8+16f", b, torch.bfloat16)
    loss, logits = eng.forward_backward(video, text, tgt)
    logits = logits.clone()
    old = _oracle_threads()
    try:
        o = Oracle(g, sd, dtype=torch.float32)
        ref = o.forward_backward(synth.video(g, b), synth.text_features(g), synth.soft_target(g, b)[0])
    finally:
        torch.set_num_threads(old)
    lgap = record("b32_vs_fp32.logits_maxabs", (logits.cpu().double() - ref["logits"].detach().double()).abs().max())
    record("b32_vs_fp32.loss_abs", abs(float(loss) - float(ref["loss"])))
    errs, l2s = [], []
    for n, gr in ref["grads"].items():
        rn = float(gr.double().norm())
        if rn < 1e-9:
            continue
        mine = eng.view(n, grad=True).double().cpu()
        errs.append((abs(float(mine.norm()) - rn) / rn, n))
        # round 6 (at b = 32 only the norms were compared): every ELEMENT of every gradient - the relative L2 distance of the tensor
        l2s.append((float((mine - gr.double().reshape(mine.shape)).norm()) / rn, n))
    errs.sort(reverse=True)
    l2s.sort(reverse=True)
    record("b32_vs_fp32.grad_rel_l2_worst", l2s[0][0])
    record("b32_vs_fp32.grad_rel_l2_median", l2s[len(l2s) // 2][0])
    print(f"b = 32 bf16 vs fp32 oracle, element-wise: worst relative L2 distances {l2s[:6]}, median {l2s[len(l2s) // 2][0]:.5f}")
    assert l2s[0][0] < B32_GATES["rel_l2_worst"], l2s[:6]
    assert l2s[len(l2s) // 2][0] < B32_GATES["rel_l2_median"], l2s[len(l2s) // 2]
    record("b32_vs_fp32.gnorm_worst", errs[0][0])
    record("b32_vs_fp32.gnorm_median", errs[len(errs) // 2][0])
    record("b32_vs_fp32.n_grad_tensors", len(errs))
    print(f"b = 32 bf16 vs fp32 oracle: logits gap {lgap:.4f}, worst gradient norms {errs[:6]}, median {errs[len(errs) // 2][0]:.5f}")
    assert len(errs) >= 370, len(errs)
    assert lgap < B32_GATES["logits"] and (logits.cpu().argmax(1) == ref["logits"].detach().argmax(1)).float().mean() > 0.9
    assert errs[0][0] < B32_GATES["gnorm_worst"], errs[:6]
    assert errs[len(errs) // 2][0] < B32_GATES["gnorm_median"], errs[len(errs) // 2]
