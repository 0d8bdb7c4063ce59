"""End-to-end parity (GPU): the HIP engine behind the C ABI vs the CPU oracle on the
same procedural weights and frames, and vs the committed golden fixtures that were
generated from the reference itself (oracle/make_golden.py).

Tolerances (written here, as north_star asks):
  * fp32 mode  : logits rtol 1e-3 / atol 1e-4 against the reference goldens and the fp64 oracle
  * bf16 mode  : against the oracle with IDENTICAL bf16 rounding points, logits atol 3e-2 (logit
                 range ~ +-4; measured 0.023), gradients 4 % of each tensor's maximum (measured 1.9 %);
                 against the fp32 reference golden on ViT-B/16: logits 0.025 (measured 0.0116), every
                 gradient norm within 1.5 % (measured 0.72 %).  Measured values: profiles/r03_parity_gaps.json;
                 the larger-geometry gates live in tests/test_configs_gpu.py.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _gaps import record  # noqa: E402


def build(gname, b, dtype, use_tr=True):
    from dist_amd import synth
    from dist_amd.engine import Engine, config_from_geometry
    g = synth.geometry(gname)
    eng = Engine(config_from_geometry(g, b, dtype, use_tr))
    sd = synth.state_dict(g)
    eng.load_state_dict(sd)
    video = torch.from_numpy(synth.video(g, b)).cuda()
    text = torch.from_numpy(synth.text_features(g)).cuda()
    tgt = torch.from_numpy(synth.soft_target(g, b)[0]).cuda()
    return g, eng, sd, video, text, tgt


def oracle_run(g, sd, b, bf16=False):
    from dist_amd import synth
    from dist_oracle import Oracle
    o = Oracle(g, sd, dtype=torch.float64, bf16=bf16)
    return o.forward_backward(synth.video(g, b), synth.text_features(g), synth.soft_target(g, b)[0])


def rel_err(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def test_param_tables_match_reference_names(gpu_lib):
    from dist_amd import synth
    g, eng, sd, *_ = build("tiny", 2, torch.float32)
    names0 = set(eng.tables[0])
    assert names0 == set(synth.dist_net_shapes(g)), names0 ^ set(synth.dist_net_shapes(g))
    for n, (off, shape, grp) in eng.tables[0].items():
        assert tuple(shape) == tuple(synth.dist_net_shapes(g)[n]), n
    assert set(eng.tables[1]) == set(synth.visual_shapes(g))
    # known-answer optimizer group sizes for ViT-B/16 8+16f
    from dist_amd.engine import Engine, config_from_geometry
    gb = synth.geometry("b16_8+16f")
    e2 = Engine(config_from_geometry(gb, 1, torch.bfloat16))
    sizes = {}
    for n, (off, shape, grp) in e2.tables[0].items():
        c = sizes.setdefault(grp, [0, 0]); c[0] += 1; c[1] += int(np.prod(shape))
    assert sizes == {0: [16, 43776], 1: [16, 7077888], 2: [32, 19968], 3: [123, 11808768], 4: [196, 50784]}
    assert e2.theta.numel() == 19001184


@pytest.mark.parametrize("gname,b", [("tiny", 2), ("tiny3", 3), ("tiny3_sel", 3), ("tiny_ratio", 2)])
def test_fp32_forward_backward_vs_reference_golden(gpu_lib, gname, b):
    g, eng, sd, video, text, tgt = build(gname, b, torch.float32)
    loss, logits = eng.forward_backward(video, text, tgt)
    gold = np.load(os.path.join(GOLD, gname + ".npz"))
    torch.testing.assert_close(logits.cpu().double(), torch.from_numpy(gold["logits"]).double(), rtol=1e-3, atol=1e-4)
    assert abs(float(loss) - float(gold["loss"])) < 1e-4
    full = gold["grad.logit_scale"].ndim == 0 and gold["grad.dist_net.proj"].ndim == 2
    bad = []
    for k in gold.files:
        if not k.startswith("grad."):
            continue
        n = k[5:]
        got = eng.view(n, grad=True).cpu().double()
        if full:
            ref = torch.from_numpy(gold[k]).double()
            err = float((got - ref).abs().max() / (ref.abs().max() + 1e-9))
        else:
            err = abs(float(got.norm()) - float(gold["gnorm." + n])) / (float(gold["gnorm." + n]) + 1e-9)
        if err > 2e-3:
            bad.append((n, err))
    assert not bad, bad[:10]
    # tensors that must NOT receive a gradient (last layer's I2T, SURVEY.md §2.2 B*)
    last = g.nsel - 1                                   # (DiST layers: one per selected ViT block)
    assert float(eng.view(f"dist_net.integration2temporal_nets.{last}.linear_fuse.weight", grad=True).abs().max()) == 0.0


def test_fp32_intermediates_vs_reference_golden(gpu_lib):
    g, eng, sd, video, text, tgt = build("tiny", 2, torch.float32)
    eng.vit_forward(video)
    eng.branch_forward(text)
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    for k in gold.files:
        if not k.startswith("act."):
            continue
        n = k[4:]
        got = eng.debug(n).cpu().double()
        ref = torch.from_numpy(gold[k]).double().reshape(got.shape)
        assert rel_err(got, ref) < 1e-4, (n, rel_err(got, ref))


@pytest.mark.parametrize("use_tr", [True, False])
def test_bf16_vs_oracle_with_same_rounding_points(gpu_lib, use_tr):
    g, eng, sd, video, text, tgt = build("tiny", 2, torch.bfloat16, use_tr)
    loss, logits = eng.forward_backward(video, text, tgt)
    ref = oracle_run(g, sd, 2, bf16=True)
    record(f"tiny.bf16_same_rounding.tr{int(use_tr)}.logits_maxabs", (logits.cpu().double() - ref["logits"].detach()).abs().max())
    record(f"tiny.bf16_same_rounding.tr{int(use_tr)}.loss_abs", abs(float(loss) - float(ref["loss"])))
    torch.testing.assert_close(logits.cpu().double(), ref["logits"].detach(), rtol=0, atol=3e-2)
    assert abs(float(loss) - float(ref["loss"])) < 7e-3              # measured 0.0033
    worst = 0.0
    for n, gr in ref["grads"].items():
        if gr.abs().max() < 1e-6:
            continue
        worst = max(worst, rel_err(eng.view(n, grad=True), gr))
    record(f"tiny.bf16_same_rounding.tr{int(use_tr)}.grad_worst_relmax", worst)
    assert worst < 0.04, worst                                       # measured 0.0193
    # and the honest gap to the fp64 oracle without rounding (reported, loosely bounded)
    ref32 = oracle_run(g, sd, 2, bf16=False)
    gap = float((logits.cpu().double() - ref32["logits"].detach()).abs().max())
    print(f"bf16 vs fp64 logits max-abs gap: {gap:.4f}")
    assert gap < 0.15


def test_b16_logits_vs_reference_golden_fp32(gpu_lib):
    """BASELINE config 1 (ViT-B/16 8+16f, b=2): logits within rtol 1e-3 / atol 1e-4 of the reference."""
    g, eng, sd, video, text, tgt = build("b16_8+16f", 2, torch.float32)
    loss, logits = eng.forward_backward(video, text, tgt)
    gold = np.load(os.path.join(GOLD, "b16_b2.npz"))
    torch.testing.assert_close(logits.cpu().double(), torch.from_numpy(gold["logits"]).double(), rtol=1e-3, atol=1e-4)
    assert abs(float(loss) - float(gold["loss"])) < 1e-4
    bad = []
    for k in gold.files:
        if k.startswith("gnorm."):
            n = k[6:]
            got = float(eng.view(n, grad=True).double().norm())
            ref = float(gold[k])
            if abs(got - ref) > 2e-3 * ref + 1e-7:
                bad.append((n, got, ref))
    assert not bad, bad[:10]


def test_b16_bf16_vs_reference_golden(gpu_lib):
    """Same config in bf16 (the LDS-DMA kernels run at these sizes) against the fp32 reference: the
    honest bf16-vs-fp32 gap, gated at ~2x measured (logits are in +-6)."""
    g, eng, sd, video, text, tgt = build("b16_8+16f", 2, torch.bfloat16)
    loss, logits = eng.forward_backward(video, text, tgt)
    gold = np.load(os.path.join(GOLD, "b16_b2.npz"))
    gap = float((logits.cpu().double() - torch.from_numpy(gold["logits"]).double()).abs().max())
    print(f"B/16 bf16 vs fp32 reference logits max-abs gap: {gap:.4f}")
    record("b16_b2.bf16_vs_fp32_golden.logits_maxabs", gap)
    record("b16_b2.bf16_vs_fp32_golden.loss_abs", abs(float(loss) - float(gold["loss"])))
    assert gap < 0.025 and abs(float(loss) - float(gold["loss"])) < 0.006      # measured 0.0116 / 0.0025 (logits in +-6)
    assert (logits.cpu().argmax(1) == torch.from_numpy(gold["logits"]).argmax(1)).all()
    bad, errs = [], []
    for k in gold.files:
        if k.startswith("gnorm."):
            n = k[6:]
            got, ref = float(eng.view(n, grad=True).double().norm()), float(gold[k])
            errs.append(abs(got - ref) / (ref + 1e-12))
            if abs(got - ref) > 0.015 * ref + 1e-6:                  # measured worst 0.72 %
                bad.append((n, got, ref))
    errs.sort(reverse=True)
    record("b16_b2.bf16_vs_fp32_golden.gnorm_worst", errs[0]); record("b16_b2.bf16_vs_fp32_golden.gnorm_4th", errs[3])
    assert not bad, bad[:10]


def test_full_size_batch_is_consistent_with_the_golden_case(gpu_lib):
    """BASELINE config 2 (ViT-B/16 8+16f, bf16, b=32 per GPU - the bench workload) through size-independent properties:
    a clip's logits do not depend on the clips it is batched with, so rows 0-1 of the b=32 run must reproduce the b=2
    run that the reference golden pins; the run is repeatable; every gradient is finite and AdamW moves the weights."""
    from dist_amd import synth
    g, eng, sd, video, text, tgt = build("b16_8+16f", 32, torch.bfloat16)
    loss, logits = eng.forward_backward(video, text, tgt)
    logits = logits.clone()
    grads = eng.grads.clone()
    g2, eng2, _, _, _, _ = build("b16_8+16f", 2, torch.bfloat16)
    loss2, logits2 = eng2.forward_backward(video[:2].contiguous(), text, tgt[:2].contiguous())
    assert torch.equal(logits[:2], logits2)                                                  # same kernels, same rows, same summation order: the same bits
    assert np.array_equal(synth.video(g, 2), video[:2].cpu().numpy())                       # rows 0-1 ARE the golden clips
    gold = np.load(os.path.join(GOLD, "b16_b2.npz"))
    gap = float((logits[:2].cpu().double() - torch.from_numpy(gold["logits"]).double()).abs().max())
    assert gap < 0.025 and (logits[:2].cpu().argmax(1) == torch.from_numpy(gold["logits"]).argmax(1)).all()
    assert torch.isfinite(logits).all() and torch.isfinite(grads).all() and float(grads.abs().max()) > 0
    # repeatable: same inputs, same weights -> same logits bit for bit, gradients up to fp32 atomic ordering
    loss_b, logits_b = eng.forward_backward(video, text, tgt)
    assert torch.equal(logits_b, logits)
    rel = float((eng.grads - grads).abs().max() / grads.abs().max())
    assert rel < 1e-4, rel
    before = eng.theta.clone()
    eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0)
    torch.cuda.synchronize()
    assert float((eng.theta - before).abs().max()) > 0


def test_adamw_steps_vs_reference_golden(gpu_lib):
    """3 train steps (fwd+bwd+AdamW with the intended DiST groups) vs torch.optim.AdamW on the reference."""
    g, eng, sd, video, text, tgt = build("tiny", 2, torch.float32)
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    for step in range(1, 4):
        loss, _ = eng.forward_backward(video, text, tgt)
        if step > 1:
            assert abs(float(loss) - float(gold[f"loss_step{step}"])) < 2e-4
        eng.adamw_step(3.2e-4, 1e-4)
        if step in (1, 3):
            for k in gold.files:
                if k.startswith(f"w{step}."):
                    n = k[3:]
                    got, ref = eng.view(n).cpu(), torch.from_numpy(gold[k])
                    diff = (got - ref).abs()
                    # Adam's update is ~lr*sign(g): an element whose gradient is ~eps may flip, so a
                    # handful of elements may be off by up to 2*step learning rates; all others are tight.
                    assert float(diff.max()) <= 2 * step * 3.2e-4 * 1.05 + 1e-6, (n, float(diff.max()))
                    frac = float((diff > 2e-6 + 1e-4 * ref.abs()).float().mean())
                    assert frac < 2e-4, (n, frac)


def test_smaller_batch_than_capacity_and_errors(gpu_lib):
    from dist_amd import lib as L
    g, eng, sd, video, text, tgt = build("tiny", 4, torch.float32)
    eng.vit_forward(video[:2].contiguous())
    lg2, _ = eng.branch_forward(text)
    eng.vit_forward(video)
    lg4, _ = eng.branch_forward(text)
    torch.testing.assert_close(lg2, lg4[:2], rtol=1e-5, atol=1e-5)
    with pytest.raises(L.DistError):
        eng.vit_forward(torch.zeros(5, 3, g.T, g.res, g.res, device="cuda"))       # over capacity
    eng.vit_forward(video)
    with pytest.raises(L.DistError):
        eng.backward(torch.zeros(4, g.K, device="cuda"))                            # backward before branch_forward


def test_grad_ready_hook_covers_the_flat_buffer_once(gpu_lib):
    """the DP overlap hook: slices arrive tail (ada + head) first, then layers L-1..0, then the stem, and tile
    [0, total) exactly; gradients are already final when a slice is reported (checked against a hook-free run)."""
    g, eng, sd, video, text, tgt = build("tiny", 2, torch.float32)
    loss, _ = eng.forward_backward(video, text, tgt)
    ref = eng.grads.clone()
    slices, snaps = [], []

    def hook(b, e):
        slices.append((b, e))
        snaps.append(eng.grads[b:e].clone())          # enqueued behind the kernels that produced the slice
    eng.set_grad_ready_hook(hook)
    eng.forward_backward(video, text, tgt)
    eng.set_grad_ready_hook(None)
    assert len(slices) == g.layers + 2
    assert slices[0][1] == eng.theta.numel() and slices[-1][0] == 0
    assert all(slices[i][0] == slices[i + 1][1] for i in range(len(slices) - 1))
    for (b, e), snap in zip(slices, snaps):
        torch.testing.assert_close(snap, ref[b:e], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_vit_prefetch_pipeline_equals_the_serial_order(gpu_lib, dtype):
    """Software pipelining over batches (dist_vit_prefetch / dist_vit_adopt): the frozen ViT of batch n+1 runs into the
    spare feature slot on the handle's prefetch stream while branch forward / backward / AdamW of batch n run on the
    caller's stream.  Four training steps over alternating batches must give the same losses, logits and weights as
    the serial order vit_forward -> branch -> backward -> AdamW (the ViT is frozen: nothing it reads changes)."""
    from dist_amd import synth, lib as L
    g, e1, sd, video, text, tgt = build("tiny", 2, dtype)
    _, e2, *_ = build("tiny", 2, dtype)
    vids = [video, torch.from_numpy(synth.video(g, 2, seed=7)).cuda()]
    tgts = [tgt, torch.from_numpy(synth.soft_target(g, 2, seed=9)[0]).cuda()]
    steps = 4

    def train(eng, pipelined):
        out = []
        if pipelined:
            with pytest.raises(L.DistError):
                eng.vit_adopt()                                 # nothing prefetched yet
            eng.vit_forward(vids[0])
        for n in range(steps):
            split = (n % 2 == 1)                                # odd steps issue the pass in two parts (dist_vit_prefetch_layers)
            if pipelined:
                eng.vit_prefetch(vids[(n + 1) % 2], layer_end=1 if split else None)
                if split:
                    with pytest.raises(L.DistError):
                        eng.vit_adopt()                             # the pass is not complete yet
            else:
                eng.vit_forward(vids[n % 2])
            logits, _ = eng.branch_forward(text)
            loss, dl = eng.loss(tgts[n % 2])
            if pipelined and split:
                eng.vit_prefetch_more()                         # remaining layers beside the backward
            eng.backward(dl)
            eng.adamw_step(3.2e-4, 1e-4, lr_mult=1.0)
            out.append((float(loss), logits.clone()))
            if pipelined:
                eng.vit_adopt()
        torch.cuda.synchronize()
        return out

    a, b_ = train(e1, False), train(e2, True)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    for (la, ga), (lb, gb) in zip(a, b_):
        assert abs(la - lb) <= tol["atol"] + tol["rtol"] * abs(la), (la, lb)
        torch.testing.assert_close(ga, gb, **tol)
    # same kernels on the same data: only the atomics' summation order in the parameter gradients may differ.  Adam's update
    # is ~lr*sign(g), so an element whose gradient is ~eps may flip: a handful may be off by up to 2*steps learning rates.
    diff = (e1.theta - e2.theta).abs()
    assert float(diff.max()) <= 2 * steps * 3.2e-4 * 1.05 + 1e-6, float(diff.max())
    frac = float((diff > (2e-6 if dtype == torch.float32 else 1e-4) + 1e-4 * e1.theta.abs()).float().mean())
    assert frac < (1e-3 if dtype == torch.float32 else 2e-2), frac
    # the two slots really alternate: feat.0 of the current slot belongs to the batch adopted last
    e2.vit_forward(vids[steps % 2])
    ref = e2.debug("feat.0").clone()
    e2.vit_prefetch(vids[(steps + 1) % 2]); e2.vit_adopt()
    torch.cuda.synchronize()
    assert not torch.equal(ref, e2.debug("feat.0"))
    e2.vit_prefetch(vids[steps % 2]); e2.vit_adopt()
    torch.cuda.synchronize()
    torch.testing.assert_close(ref, e2.debug("feat.0"), rtol=0, atol=0)


def test_vit_layernorm_fold_survives_the_optimizer_step(gpu_lib):
    """The frozen ViT's LayerNorms are folded into the consuming GEMMs (DIST_EPI_LNFOLD, statistics from DIST_EPI_ROWSTATS partials):
    the normalised tensor is never written.  The per-step re-pack of the TRAINABLE weights must leave that alone - it used to reset
    the fold, so every ViT pass after the first optimizer step ran LayerNorm + GEMM again (and bench.py measured that)."""
    g, eng, sd, video, text, tgt = build("b16_8+16f", 2, torch.bfloat16)
    eng.vit_forward(video); eng.branch_forward(text)
    scratch = eng.debug("vit_ln_out")
    for after_step in (False, True):
        if after_step:
            _, dl = eng.loss(tgt); eng.backward(dl); eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0)
        scratch.fill_(float("nan"))
        eng.vit_forward(video); eng.branch_forward(text)
        torch.cuda.synchronize()
        assert bool(torch.isnan(scratch.float()).all()), f"the ViT wrote its LayerNorm output (after_step={after_step}): the fold is off"


def test_inference_mode_gives_the_same_logits_and_refuses_backward(gpu_lib):
    """dist_set_inference (the reference's torch.no_grad() evaluation loops): the forward writes nothing that only a backward pass reads.
    With the fused IntegrationNetwork kernel both modes run the same arithmetic (the logits are bit-identical on this geometry: gap 0.0; the
    unfused GEMM path applied the two QuickGELUs to the fp32 accumulators in inference mode: 0.005 on a +-6 range), so the gate only bounds
    the gap; backward refuses, and a training-mode forward afterwards reproduces the first one."""
    from dist_amd import lib as L
    g, eng, sd, video, text, tgt = build("b16_8+16f", 2, torch.bfloat16)
    loss, logits = eng.forward_backward(video, text, tgt)
    logits, grads = logits.clone(), eng.grads.clone()
    eng.set_inference(True)
    eng.vit_forward(video)
    lg_inf, vid_inf = eng.branch_forward(text)
    gap = record("inference_vs_training.logits_maxabs", (lg_inf - logits).abs().max())
    assert 0 <= gap < 0.02 and torch.equal(lg_inf.argmax(1), logits.argmax(1))
    _, dl = eng.loss(tgt)
    with pytest.raises(L.DistError, match="inference"):
        eng.backward(dl)
    eng.set_inference(False)
    loss2, logits2 = eng.forward_backward(video, text, tgt)
    assert torch.equal(logits2, logits)
    rel = float((eng.grads - grads).abs().max() / grads.abs().max())
    assert rel < 1e-4, rel


def test_backward_accumulates_with_the_layernorm_fold(gpu_lib):
    """dist_branch_backward(zero_grads=0) ADDS to the bound gradient buffer (include/dist_amd.h).  On ViT-B/16 the weight-gradient GEMMs of the two
    folded Linears leave G' = dz^T xhat and the unfold used to rewrite the slot in place - rescaling what an earlier pass had left there.
    Two accumulated backward passes of the same batch must give twice the gradient of one."""
    g, eng, sd, video, text, tgt = build("b16_8+16f", 2, torch.bfloat16)
    eng.vit_forward(video)
    eng.branch_forward(text)
    _, dl = eng.loss(tgt)
    eng.backward(dl, zero_grads=True)
    torch.cuda.synchronize()
    g1 = eng.grads.clone()
    eng.branch_forward(text)
    _, dl = eng.loss(tgt)
    eng.backward(dl, zero_grads=False)
    torch.cuda.synchronize()
    g2 = eng.grads.clone()
    worst = ("", 0.0)
    for n, (off, shape, _) in eng.tables[0].items():
        k = int(np.prod(shape)) if shape else 1
        a, b2 = g1[off:off + k].double(), g2[off:off + k].double()
        err = float((b2 - 2 * a).abs().max() / (2 * a.abs().max() + 1e-30))
        if err > worst[1]:
            worst = (n, err)
    assert worst[1] < 1e-4, worst          # (fp32 sums in another order; the defect was a factor gamma on two weights and wrong LayerNorm gradients)


def test_selected_layers_subset_intermediates_and_import(gpu_lib):
    """DIST.SELECTED_LAYERS = [0, 2] (dist_config.selected_mask = 0b101; reference dist.py:170-190, 226): DiST layer k reads ViT block selected[k].
    Every intermediate against the reference's own tensors (tests/golden/tiny3_sel.npz), the parameter table holds two DiST layers, and
    dist_features_import takes the two selected blocks only."""
    g, eng, sd, video, text, tgt = build("tiny3_sel", 3, torch.float32)
    assert eng.selected == [0, 2] and eng.cfg.selected_mask == 5
    assert not any(".2." in n and ("temporal_nets" in n or "integration_nets" in n or "input_linears" in n) for n in eng.tables[0])
    eng.vit_forward(video)
    logits, vid = eng.branch_forward(text)
    gold = np.load(os.path.join(GOLD, "tiny3_sel.npz"))
    torch.testing.assert_close(logits.cpu().double(), torch.from_numpy(gold["logits"]).double(), rtol=1e-3, atol=1e-4)
    for k in gold.files:
        if k.startswith("act."):
            got = eng.debug(k[4:]).cpu().double()
            assert rel_err(got, torch.from_numpy(gold[k]).double().reshape(got.shape)) < 1e-4, k
    # the caller's features for the selected blocks only, in the reference's [L, b*t, C] layout
    bt, L_ = 3 * g.t, g.N + 1
    mid = {i: eng.debug(f"feat.{i}").view(bt, L_, g.d).permute(1, 0, 2).contiguous().float().clone() for i in (0, 2)}
    other = torch.from_numpy(__import__("dist_amd.synth", fromlist=["x"]).video(g, 3, seed=9)).cuda()
    eng.vit_forward(other)                                             # the slot now holds another clip
    eng.import_features(mid, video)
    logits2, _ = eng.branch_forward(text)
    torch.testing.assert_close(logits2, logits, rtol=0, atol=0)
    # block 1 was not supplied, so the slot's feat.1 still holds the OTHER clip - it must not be readable as this clip's
    # (CLIP's lazy img_logits reads feat.<last>: with a selection that leaves the last block out it would silently return another clip's logits)
    eng.debug("feat.0"); eng.debug("feat.2")
    with pytest.raises(Exception, match="feat.1"):
        eng.debug("feat.1")
    eng.vit_forward(video)
    eng.debug("feat.1")                                                # a full ViT pass makes every block valid again
    with pytest.raises(Exception):
        eng.import_features({0: mid[0]}, video)                        # block 2 missing
    # bf16 engine of the same geometry: runs, finite, close
    g2, e2, _, v2, t2, y2 = build("tiny3_sel", 3, torch.bfloat16)
    loss, lg = e2.forward_backward(v2, t2, y2)
    assert torch.isfinite(lg).all() and float((lg.cpu().double() - torch.from_numpy(gold["logits"]).double()).abs().max()) < 0.15


def test_mlp_ratios_intermediates_and_bf16(gpu_lib):
    """DIST.TEMPORAL_CONV_MLP_RATIO = 2 / INTEGRATION_MLP_RATIO = 0.5 (dist_config.temporal_hidden / integration_hidden; reference dist.py:20-25, 51-58):
    every intermediate of the fp32 engine against the reference's own tensors (tests/golden/tiny_ratio.npz); the bf16 engine of the geometry runs the
    unfused kernel sequence and stays within bf16 distance."""
    g, eng, sd, video, text, tgt = build("tiny_ratio", 2, torch.float32)
    assert eng.cfg.temporal_hidden == 64 and eng.cfg.integration_hidden == 64
    assert tuple(eng.tables[0]["dist_net.temporal_nets.0.temporal_net.c_fc1.weight"][1]) == (64, 32, 3, 1, 1)
    assert tuple(eng.tables[0]["dist_net.integration_nets.1.ffn.c_proj.weight"][1]) == (128, 64)
    eng.vit_forward(video)
    logits, _ = eng.branch_forward(text)
    gold = np.load(os.path.join(GOLD, "tiny_ratio.npz"))
    torch.testing.assert_close(logits.cpu().double(), torch.from_numpy(gold["logits"]).double(), rtol=1e-3, atol=1e-4)
    for k in gold.files:
        if k.startswith("act."):
            got = eng.debug(k[4:]).cpu().double()
            assert rel_err(got, torch.from_numpy(gold[k]).double().reshape(got.shape)) < 1e-4, k
    g2, e2, _, v2, t2, y2 = build("tiny_ratio", 2, torch.bfloat16)
    loss, lg = e2.forward_backward(v2, t2, y2)
    assert torch.isfinite(lg).all() and float((lg.cpu().double() - torch.from_numpy(gold["logits"]).double()).abs().max()) < 0.15
    ref = oracle_run(g2, sd, 2, bf16=True)
    worst = max(rel_err(e2.view(n, grad=True), gr) for n, gr in ref["grads"].items() if gr.abs().max() >= 1e-6)
    assert worst < 0.06, worst
