"""Evaluation side (reference utils/metrics.py:100-159, utils/meters.py:24-176, base_blocks.py:573-585).

Not GPU: the numpy oracle reproduces what the REFERENCE's own `topks_correct` / `topk_errors` / `TestMeter` returned on seeded
inputs (tests/golden/meters.npz, made by oracle/make_golden_meters.py).
GPU: dist_op_topk_correct / dist_op_ensemble_update / dist_op_softmax_rows behind the C ABI, through the drop-in
`dist_amd.utils.metrics` and `dist_amd.utils.meters.TestMeter`, reproduce the same vectors: counts, labels, view counts and the
"sum" / "max" score tables bit for bit (integer and ordered fp32 work), softmax within 2 ulp-scale (1e-6 relative, fp32)."""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import meters_oracle as mo  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "meters.npz"))
NM, NTK = int(GOLD["n_meter"]), int(GOLD["n_topk"])


# ---- the input recipes of oracle/make_golden_meters.py (inputs are procedural; only outputs are stored) --------------------
def softmax32(z):
    return torch.softmax(torch.from_numpy(z), dim=-1).numpy()


def scores(seed, n, K, kind="softmax"):
    z = (np.random.Generator(np.random.PCG64(7000 + seed)).standard_normal((n, K)) * 3.0).astype(np.float32)
    return z if kind == "logits" else softmax32(z)


def int_draws(seed, hi, n):
    return np.random.Generator(np.random.PCG64(9000 + seed)).integers(0, hi, size=n).astype(np.int64)


def meter_inputs(seed, V, views, K, bs, frac):
    order = np.random.Generator(np.random.PCG64(8000 + seed)).permutation(V * views)
    order = order[: max(1, int(round(len(order) * frac)))]
    vid_label = int_draws(seed, K, V)
    if seed == 0:
        vid_label[0] = 0
    p = scores(seed, len(order), K)
    return [(p[i:i + bs], vid_label[order[i:i + bs] // views], order[i:i + bs]) for i in range(0, len(order), bs)]


def meter_case(ci):
    seed, V, views, K, bs, is_max, pct = (int(v) for v in GOLD[f"m{ci}_meta"])
    return seed, V, views, K, bs, ("max" if is_max else "sum"), pct / 100.0


# ---- oracle vs the reference's outputs (CPU) -------------------------------------------------------------------------------
@pytest.mark.parametrize("ci", range(NTK))
def test_oracle_topk_reproduces_the_reference(ci):
    seed, n, K = (int(v) for v in GOLD[f"t{ci}_meta"])
    ks = tuple(int(k) for k in GOLD[f"t{ci}_ks"])
    p, lab = scores(100 + seed, n, K), int_draws(100 + seed, K, n)
    got = mo.topks_correct(p, lab, ks)
    assert np.array_equal(np.array(got, np.float32), GOLD[f"t{ci}_correct"])
    err = [np.float32((np.float32(1.0) - x / np.float32(n)) * np.float32(100.0)) for x in got]
    np.testing.assert_allclose(np.array(err, np.float32), GOLD[f"t{ci}_errors"], rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("ci", range(NM))
def test_oracle_meter_reproduces_the_reference(ci):
    seed, V, views, K, bs, method, frac = meter_case(ci)
    m = mo.TestMeterOracle(V, views, K, method)
    for p, lab, ids in meter_inputs(seed, V, views, K, bs, frac):
        m.update_stats(p, lab, ids)
    assert np.array_equal(m.video_preds, GOLD[f"m{ci}_video_preds"])          # same fp32 additions in the same order
    assert np.array_equal(m.video_labels, GOLD[f"m{ci}_video_labels"])
    assert np.array_equal(m.clip_count, GOLD[f"m{ci}_clip_count"])
    ks = (1, 5) if K >= 5 else (1, 2)
    acc = np.array([x / np.float32(V) * np.float32(100.0) for x in mo.topks_correct(m.video_preds, m.video_labels, ks)], np.float32)
    np.testing.assert_allclose(acc, GOLD[f"m{ci}_acc"], rtol=1e-6)


def test_oracle_softmax_and_golden_set_coverage():
    np.testing.assert_allclose(mo.softmax_rows(scores(55, 9, 174, "logits")), GOLD["s_softmax"], rtol=2e-6, atol=1e-9)
    kinds = {meter_case(ci)[5] for ci in range(NM)}
    assert kinds == {"sum", "max"} and any(meter_case(ci)[6] < 1.0 for ci in range(NM))       # both ensembles, one incomplete test set
    assert any(int(GOLD[f"m{ci}_meta"][3]) == 174 for ci in range(NM)) and any(int(GOLD[f"m{ci}_meta"][3]) == 400 for ci in range(NM))


def test_oracle_meter_flags_what_the_reference_asserts():
    m = mo.TestMeterOracle(2, 2, 4)
    p = np.full((1, 4), 0.25, np.float32)
    m.update_stats(p, np.array([3]), np.array([0]))
    with pytest.raises(AssertionError):
        m.update_stats(p, np.array([2]), np.array([1]))                        # second view of video 0 with another label
    m2 = mo.TestMeterOracle(2, 2, 4)
    m2.update_stats(p, np.array([0]), np.array([0]))
    m2.update_stats(p, np.array([2]), np.array([1]))                           # stored label 0 is never compared (reference :97)
    with pytest.raises(IndexError):
        m2.update_stats(p, np.array([1]), np.array([4]))


def test_product_modules_refuse_cpu_tensors():
    from dist_amd.utils import metrics
    with pytest.raises(RuntimeError):
        metrics.topks_correct(torch.zeros(2, 5), torch.zeros(2, dtype=torch.long), (1, 5))
    if not torch.cuda.is_available():
        from dist_amd.utils.meters import TestMeter
        with pytest.raises(RuntimeError):
            TestMeter(None, 4, 2, 5, 1)


# ---- HIP path vs oracle and golden (GPU) -----------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(NTK))
def test_hip_topk_matches_the_reference(gpu_lib, ci):
    from dist_amd.utils import metrics
    seed, n, K = (int(v) for v in GOLD[f"t{ci}_meta"])
    ks = tuple(int(k) for k in GOLD[f"t{ci}_ks"])
    p, lab = torch.from_numpy(scores(100 + seed, n, K)).cuda(), torch.from_numpy(int_draws(100 + seed, K, n)).cuda()
    got = metrics.topks_correct(p, lab, ks)
    assert all(g.dim() == 0 and g.dtype == torch.float32 for g in got)        # the reference returns 0-dim float tensors
    assert np.array_equal(np.array([float(g) for g in got], np.float32), GOLD[f"t{ci}_correct"])
    np.testing.assert_allclose(np.array([float(e) for e in metrics.topk_errors(p, lab, ks)], np.float32), GOLD[f"t{ci}_errors"], rtol=1e-6, atol=1e-5)
    acc = metrics.topk_accuracies(p, lab, ks)
    np.testing.assert_allclose(np.array([float(a) for a in acc]) + GOLD[f"t{ci}_errors"], 100.0, rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(NM))
def test_hip_meter_is_bit_identical_to_the_reference(gpu_lib, ci):
    from dist_amd.utils.meters import TestMeter
    seed, V, views, K, bs, method, frac = meter_case(ci)
    m = TestMeter(NS(LOG_PERIOD=0), V, views, K, 1, method)
    for p, lab, ids in meter_inputs(seed, V, views, K, bs, frac):
        m.update_stats(torch.from_numpy(p).cuda(), torch.from_numpy(lab).cuda(), torch.from_numpy(ids).cuda())
    assert np.array_equal(m.video_preds.cpu().numpy(), GOLD[f"m{ci}_video_preds"])
    assert np.array_equal(m.video_labels.cpu().numpy(), GOLD[f"m{ci}_video_labels"])
    assert np.array_equal(m.clip_count.cpu().numpy(), GOLD[f"m{ci}_clip_count"])
    ks = (1, 5) if K >= 5 else (1, 2)
    stats = m.finalize_metrics(ks)
    assert stats["split"] == "test_final"
    for k, a in zip(ks, GOLD[f"m{ci}_acc"]):
        assert stats[f"top{k}_acc"] == "{:.2f}".format(float(a))
    m.reset()
    assert float(m.video_preds.abs().sum()) == 0 and int(m.clip_count.sum()) == 0 and int(m.video_labels.sum()) == 0


@pytest.mark.gpu
def test_hip_meter_flags_label_mismatch_and_bad_ids(gpu_lib):
    from dist_amd.utils.meters import TestMeter
    p = torch.full((1, 4), 0.25, device="cuda")
    lab = lambda v: torch.tensor([v], device="cuda")
    m = TestMeter(None, 2, 2, 4, 1)
    m.update_stats(p, lab(3), lab(0))
    m.update_stats(p, lab(2), lab(1))
    with pytest.raises(AssertionError):
        m.finalize_metrics((1, 2))
    m = TestMeter(None, 2, 2, 4, 1)
    m.update_stats(p, lab(0), lab(0))
    m.update_stats(p, lab(2), lab(1))                                          # stored label 0: not compared, as in the reference
    m.check()
    m.update_stats(p, lab(1), lab(4))
    with pytest.raises(IndexError):
        m.check()
    with pytest.raises(NotImplementedError):
        TestMeter(None, 2, 2, 4, 1, "mean")


@pytest.mark.gpu
def test_hip_topk_tie_rule_and_properties_at_full_size(gpu_lib):
    from dist_amd import ops
    # exactly equal scores rank by class index (torch.topk leaves this open; the oracle states the rule)
    p = torch.tensor([[0.1, 0.3, 0.3, 0.3, 0.2, 0.3], [0.5, 0.5, 0.0, 0.0, 0.0, 0.0], [0.0] * 6], device="cuda")
    lab = torch.tensor([3, 1, 5], device="cuda")
    got = ops.topk_correct(p, lab, (1, 2, 3, 6)).cpu().numpy()
    want = np.array(mo.topks_correct(p.cpu().numpy(), lab.cpu().numpy(), (1, 2, 3, 6)), np.float32)
    assert np.array_equal(got, want) and np.array_equal(want, np.array([0, 1, 2, 3], np.float32))
    # full evaluation batch (8 ranks x 32 clips gathered, K400): against the oracle, and monotone in k; out-of-range labels never count
    n, K = 256, 400
    pp, ll = scores(77, n, K), int_draws(77, K, n)
    ll[:3] = [-1, K, K + 5]
    got = ops.topk_correct(torch.from_numpy(pp).cuda(), torch.from_numpy(ll).cuda(), (1, 5, 50, K)).cpu().numpy()
    assert np.array_equal(got, np.array(mo.topks_correct(pp, ll, (1, 5, 50, K)), np.float32))
    assert got[0] <= got[1] <= got[2] <= got[3] == n - 3
    # bf16 predictions are read as their fp32 values
    pb = torch.from_numpy(pp).cuda().bfloat16()
    assert np.array_equal(ops.topk_correct(pb, torch.from_numpy(ll).cuda(), (1, 5)).cpu().numpy(),
                          np.array(mo.topks_correct(pb.float().cpu().numpy(), ll, (1, 5)), np.float32))


@pytest.mark.gpu
def test_hip_softmax_head(gpu_lib):
    from dist_amd import ops
    from dist_amd.models.base.base_blocks import ClipVideoTextIdentity
    z = scores(55, 9, 174, "logits")
    y = ops.softmax_rows(torch.from_numpy(z).cuda())
    np.testing.assert_allclose(y.cpu().numpy(), GOLD["s_softmax"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(y.sum(dim=1).cpu().numpy(), 1.0, rtol=1e-6)
    head = ClipVideoTextIdentity(NS(VIDEO=NS(HEAD=NS(ACTIVATION="softmax")))).eval()
    out, passthrough = head({"logits_per_image": torch.from_numpy(z).cuda().view(9, 1, 174)})
    assert torch.equal(out, y) and "logits_per_image" in passthrough
    head.train()
    out, _ = head({"logits_per_image": torch.from_numpy(z).cuda().view(9, 1, 174)})
    assert torch.equal(out.cpu(), torch.from_numpy(z))                          # training: raw logits (reference :582-584)
    big = ops.softmax_rows(torch.from_numpy(scores(56, 256, 400, "logits") * 30).cuda())   # large logits: no overflow
    assert torch.isfinite(big).all() and float((big.sum(dim=1) - 1).abs().max()) < 1e-5
