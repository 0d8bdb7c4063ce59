"""Batch-mode Mixup / CutMix (reference dataset/utils/mixup.py:18-23,212-223).

Not GPU: the numpy oracle reproduces the golden vectors the REFERENCE's own class produced (oracle/make_golden_mixup.py),
bit for bit, and the evaluation-order trap of the in-place formula is pinned.
GPU: the HIP-backed drop-in class (dist_op_mixup / dist_op_cutmix / dist_op_mixup_target behind the C ABI) reproduces the same
vectors bit for bit from the same np.random seed, plus size-independent properties at the BASELINE batch.
"""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mixup_oracle as mo  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "mixup.npz"))
NCASES = int(GOLD["n_cases"])


def clips(seed, b, T, H, W):          # the recipe of oracle/make_golden_mixup.py (inputs are procedural, only outputs are stored)
    return np.random.Generator(np.random.PCG64(1000 + seed)).standard_normal((b, 3, T, H, W)).astype(np.float32)


def labels_of(seed, b, K):
    return np.random.Generator(np.random.PCG64(2000 + seed)).integers(0, K, size=b).astype(np.int64)


def case(ci):
    seed, b, T, H, W, K = (int(v) for v in GOLD[f"c{ci}_meta"])
    ma, ca, prob, sw, sm = (float(v) for v in GOLD[f"c{ci}_hyper"])
    return seed, b, T, H, W, K, ma, ca, prob, sw, sm


def cfg_of(K, ma, ca, prob, sw, sm):
    return NS(AUGMENTATION=NS(MIXUP=NS(ALPHA=ma, PROB=prob, SWITCH_PROB=sw, MODE="batch"), CUTMIX=NS(ENABLE=ca > 0.0, ALPHA=ca, MINMAX=None),
                              LABEL_SMOOTHING=sm), VIDEO=NS(HEAD=NS(NUM_CLASSES=K)))


@pytest.mark.parametrize("ci", range(NCASES))
def test_oracle_reproduces_the_reference_vectors(ci):
    seed, b, T, H, W, K, ma, ca, prob, sw, sm = case(ci)
    x, lab = clips(seed, b, T, H, W), labels_of(seed, b, K)
    np.random.seed(seed)
    lam, soft, _ = mo.mixup_call(x, lab, K, ma, ca, prob, sw, sm)
    assert lam == float(GOLD[f"c{ci}_lam"])
    assert np.array_equal(x, GOLD[f"c{ci}_x"])
    assert np.array_equal(soft, GOLD[f"c{ci}_soft"])


def test_golden_set_covers_mixup_cutmix_identity_and_odd_batches():
    kinds = set()
    for ci in range(NCASES):
        seed, b, T, H, W, K, ma, ca, prob, sw, sm = case(ci)
        np.random.seed(seed)
        lam, uc = mo.params_per_batch(ma, ca, prob, sw)
        kinds.add(("identity" if lam == 1.0 else "cutmix" if uc else "mixup", b % 2))
    assert {("mixup", 0), ("mixup", 1), ("cutmix", 0), ("cutmix", 1), ("identity", 0)} <= kinds


def test_draw_plan_makes_the_reference_draws_in_the_reference_order():
    """The product's host side (dist_amd.dataset.utils.mixup.draw_plan: one function, its own form) consumes numpy's global
    stream exactly as the pinned oracle does: same kind, same weight, same box, and the generator is left in the same state -
    over the golden cases and a sweep of seeds / hyper-parameters / frame sizes (every branch: skipped, mixup only, cutmix only,
    switch between the two, boxes cut by the border)."""
    from dist_amd.dataset.utils.mixup import draw_plan
    cases = [case(ci)[0:1] + case(ci)[3:5] + case(ci)[6:10] for ci in range(NCASES)]
    for seed in range(200):
        for H, W in ((224, 224), (7, 5), (64, 96)):
            for ma, ca, prob, sw in ((0.8, 1.0, 1.0, 0.5), (0.8, 0.0, 1.0, 0.5), (0.0, 1.0, 1.0, 0.5), (0.8, 1.0, 0.5, 0.9), (0.2, 0.3, 0.7, 0.1)):
                cases.append((seed, H, W, ma, ca, prob, sw))
    kinds = set()
    for seed, H, W, ma, ca, prob, sw in cases:
        np.random.seed(seed)
        lam, uc = mo.params_per_batch(ma, ca, prob, sw)
        box = None
        if lam != 1.0 and uc:
            box, lam = mo.cutmix_bbox_and_lam((H, W), lam)
        after_oracle = np.random.rand()
        np.random.seed(seed)
        plan = draw_plan(H, W, ma, ca, prob, sw)
        assert np.random.rand() == after_oracle, "generator state differs after the call"
        want = "none" if lam == 1.0 and box is None else ("cutmix" if uc else "mixup")
        assert plan.kind == want and plan.lam == lam and (plan.box == tuple(box) if box is not None else plan.box is None), (seed, H, W, plan, lam, box)
        kinds.add(plan.kind)
    assert kinds == {"none", "mixup", "cutmix"}
    with pytest.raises(AssertionError):
        np.random.seed(0); draw_plan(8, 8, 0.0, 0.0, 1.0, 0.5)


def test_in_place_formula_needs_the_flipped_copy_first():
    """reference :221-222 takes x.flip(0) BEFORE scaling x; `x.mul_(lam).add_(x.flip(0).mul_(1-lam))` scales first and mixes
    lam*x with lam*(1-lam)*flip(x) - the defect the round-1 class had."""
    x0 = torch.from_numpy(clips(0, 4, 1, 4, 4)); lam = 0.3
    good = x0.clone(); xf = good.flip(0).mul_(1.0 - lam); good.mul_(lam).add_(xf)
    bad = x0.clone(); bad.mul_(lam).add_(bad.flip(0).mul_(1.0 - lam))
    ora = x0.numpy().copy(); ora[...] = (ora * np.float32(lam)).astype(np.float32) + (ora[::-1] * np.float32(1.0 - lam)).astype(np.float32)
    assert np.array_equal(good.numpy(), ora)
    assert float((good - bad).abs().max()) > 0.1


def test_product_class_refuses_cpu_tensors():
    from dist_amd.dataset.utils.mixup import Mixup
    fn = Mixup(cfg_of(10, 0.8, 1.0, 1.0, 0.5, 0.1))
    np.random.seed(0)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        fn({"video": torch.zeros(2, 3, 1, 4, 4)}, torch.zeros(2, dtype=torch.long))


# ---------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(NCASES))
def test_hip_mixup_is_bit_identical_to_the_reference(gpu_lib, ci):
    from dist_amd.dataset.utils.mixup import Mixup
    seed, b, T, H, W, K, ma, ca, prob, sw, sm = case(ci)
    x = torch.from_numpy(clips(seed, b, T, H, W)).cuda()
    lab = torch.from_numpy(labels_of(seed, b, K)).cuda()
    fn = Mixup(cfg_of(K, ma, ca, prob, sw, sm))
    np.random.seed(seed)
    inputs, soft = fn({"video": x}, lab)
    assert inputs["video"] is x                                   # in place, like the reference
    assert np.array_equal(x.cpu().numpy(), GOLD[f"c{ci}_x"])
    assert np.array_equal(soft.cpu().numpy(), GOLD[f"c{ci}_soft"])


@pytest.mark.gpu
def test_hip_mixup_properties_at_the_baseline_batch(gpu_lib):
    """b = 32 clips of 3 x 16 x 224 x 224 (BASELINE config 2, 308 MB): oracle-sized checks do not reach this size, properties do."""
    from dist_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    x0 = torch.randn(32, 3, 16, 224, 224, generator=g, device="cuda")
    # cutmix is an involution, leaves everything outside the box untouched and swaps the box between clip i and clip 31-i
    x = x0.clone()
    ops.cutmix_(x, 30, 140, 17, 201)
    assert torch.equal(x[:, :, :, :30], x0[:, :, :, :30]) and torch.equal(x[:, :, :, 140:], x0[:, :, :, 140:])
    assert torch.equal(x[..., :17], x0[..., :17]) and torch.equal(x[..., 201:], x0[..., 201:])
    assert torch.equal(x[:, :, :, 30:140, 17:201], x0.flip(0)[:, :, :, 30:140, 17:201])
    ops.cutmix_(x, 30, 140, 17, 201)
    assert torch.equal(x, x0)
    # mixup equals the reference's three torch ops bit for bit, on every element
    lam = 0.37251
    x = x0.clone(); ops.mixup_(x, lam)
    ref = x0.clone(); xf = ref.flip(0).mul_(1.0 - lam); ref.mul_(lam).add_(xf)
    assert torch.equal(x, ref)
    del ref, xf
    # lam = 1 (the class never launches then) is the identity up to -0 / rounding of x*1 + y*0
    x = x0.clone(); ops.mixup_(x, 1.0)
    assert torch.equal(x, x0)
    # soft target: rows sum to 1, and equal the reference's tensor expression bit for bit
    lab = torch.randint(0, 174, (32,), generator=g, device="cuda")
    soft = ops.mixup_target(lab, 174, lam, 0.1)
    off = 0.1 / 174; on = 1.0 - 0.1 + off
    y1 = torch.full((32, 174), off, device="cuda").scatter_(1, lab.view(-1, 1), on)
    y2 = torch.full((32, 174), off, device="cuda").scatter_(1, lab.flip(0).view(-1, 1), on)
    assert torch.equal(soft, y1 * lam + y2 * (1.0 - lam))
    torch.testing.assert_close(soft.sum(1), torch.ones(32, device="cuda"), rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_hip_mixup_edge_cases(gpu_lib):
    from dist_amd import ops
    x0 = torch.randn(3, 3, 2, 8, 8, device="cuda")                # odd batch: the middle clip is mixed with itself / keeps its box
    x = x0.clone(); ops.cutmix_(x, 2, 6, 1, 5)
    assert torch.equal(x[1], x0[1]) and torch.equal(x[0, ..., 2:6, 1:5], x0[2, ..., 2:6, 1:5])
    x = x0.clone(); ops.cutmix_(x, 3, 3, 0, 8)                    # empty box: nothing moves
    assert torch.equal(x, x0)
    x = x0.clone(); ops.cutmix_(x, 0, 8, 0, 8)                    # whole frame: the batch is flipped
    assert torch.equal(x, x0.flip(0))
    x1 = torch.randn(1, 3, 2, 8, 8, device="cuda")                # a single clip pairs with itself
    y = x1.clone(); ops.mixup_(y, 0.25)
    r = x1.clone(); rf = r.flip(0).mul_(0.75); r.mul_(0.25).add_(rf)
    assert torch.equal(y, r)


# ---------------------------------------------------------------------------------------------- Mixup fused into the patch-row gather
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("b,T,res,P", [(4, 2, 64, 16), (5, 2, 64, 16), (3, 2, 56, 14), (1, 1, 32, 16)])
def test_patch_rows_of_the_mixed_batch_equal_mix_then_patchify(gpu_lib, dtype, b, T, res, P):
    """dist_op_patchify_mixed (Mixup fused into the patch-row gather) == dist_op_patchify after dist_op_mixup / dist_op_cutmix, bit for
    bit, in both storage types, for the band kernel (P % 8 == 0) and the element-wise one (ViT-L/14's P = 14), even / odd / single-clip batches - and the
    frames are left untouched."""
    from dist_amd import ops
    g = torch.Generator(device="cuda").manual_seed(b * 100 + P)
    x0 = torch.randn(b, 3, T, res, res, generator=g, device="cuda")
    for kind, lam, box in (("mixup", 0.37251, None), ("mixup", 0.9, None), ("cutmix", 0.5, (res // 8, res - res // 4, 3, res // 2 + 1)), ("cutmix", 0.5, (0, res, 0, res)),
                           ("cutmix", 0.5, (7, 7, 0, 9)), ("none", 1.0, None)):
        ref = x0.clone()
        if kind == "mixup":
            ops.mixup_(ref, lam)
        elif kind == "cutmix":
            ops.cutmix_(ref, *box)
        want = ops.patchify(ref, P, dtype)
        x = x0.clone()
        got = ops.patchify_mixed(x, P, dtype, kind, lam, box)
        assert torch.equal(got, want), (kind, lam, box)
        assert torch.equal(x, x0)                                    # only read


@pytest.mark.gpu
def test_patch_rows_of_the_mixed_batch_at_the_baseline_batch(gpu_lib):
    """b = 32 x 16 frames x 224^2 (BASELINE config 2): the fused gather against mix-then-gather, every patch row."""
    from dist_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    x0 = torch.randn(32, 3, 16, 224, 224, generator=g, device="cuda")
    ref = x0.clone(); ops.mixup_(ref, 0.6180339)
    assert torch.equal(ops.patchify_mixed(x0, 16, torch.bfloat16, "mixup", 0.6180339), ops.patchify(ref, 16, torch.bfloat16))
    ref = x0.clone(); ops.cutmix_(ref, 30, 140, 17, 201)
    assert torch.equal(ops.patchify_mixed(x0, 16, torch.bfloat16, "cutmix", 0.5, (30, 140, 17, 201)), ops.patchify(ref, 16, torch.bfloat16))


@pytest.mark.gpu
def test_engine_consumes_a_deferred_mix_plan(gpu_lib):
    """TRAIN.FUSE_MIXUP: Mixup leaves its plan on the clip tensor, the next ViT pass over that tensor (vit_forward or the pipelined vit_prefetch) applies it while
    gathering the patch rows - the same features and logits as mixing the clips in place first; the plan is consumed once; the soft target is unchanged."""
    from types import SimpleNamespace as NS
    from dist_amd import synth
    from dist_amd.dataset.utils.mixup import Mixup
    from dist_amd.engine import Engine, config_from_geometry
    g = synth.geometry("tiny")
    eng = Engine(config_from_geometry(g, 4, torch.float32))
    eng.load_state_dict(synth.state_dict(g))
    text = torch.from_numpy(synth.text_features(g)).cuda()
    x0 = torch.from_numpy(synth.video(g, 4, seed=3)).cuda()
    lab = torch.arange(4, device="cuda") % g.K
    for ma, ca in ((0.8, 0.0), (0.0, 1.0)):
        cfgs = [cfg_of(g.K, ma, ca, 1.0, 0.5, 0.1) for _ in range(2)]
        cfgs[1].TRAIN = NS(FUSE_MIXUP=True)
        outs = []
        for cfg, how in zip(cfgs, ("forward", "prefetch")):
            fn = Mixup(cfg)
            x = x0.clone()
            np.random.seed(7)
            inputs, soft = fn({"video": x}, lab)
            if fn.fuse:
                assert torch.equal(x, x0) and getattr(x, "_dist_mix", None) is not None          # the frames are not written, the plan travels with them
                eng.vit_prefetch(x); eng.vit_adopt()
                assert getattr(x, "_dist_mix", None) is None                                     # consumed
            else:
                assert not torch.equal(x, x0)
                eng.vit_forward(x)
            logits, _ = eng.branch_forward(text)
            outs.append((logits.clone(), soft.clone(), eng.debug("feat.0").clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
