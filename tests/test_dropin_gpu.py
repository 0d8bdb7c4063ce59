"""GPU: the reference-shaped Python surface (registries, build_model, ClipVisionTextTransformer, CLIP,
DiSTNetwork, ClipVideoTextIdentity, construct_optimizer, checkpoints, runs/run.py) on top of the C ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TINY = os.path.join(ROOT, "configs", "projects", "dist", "smoke", "tiny.yaml")


def tiny_cfg(*extra):
    from dist_amd.utils.config import Config
    return Config(load=True, argv=["--cfg", TINY] + list(extra))


def test_registries_and_model_surface(gpu_lib):
    from dist_amd.models.base.backbone import BACKBONE_REGISTRY
    from dist_amd.models.base.base_blocks import HEAD_REGISTRY
    from dist_amd.models.base.clip import ATTEN_BLOCK_REGISTRY
    from dist_amd.models.base.models import MODEL_REGISTRY
    assert BACKBONE_REGISTRY.get("ClipVisionTextTransformer") is not None
    assert HEAD_REGISTRY.get("ClipVideoTextIdentity") is not None
    assert ATTEN_BLOCK_REGISTRY.get("ResidualAttentionBlockMid") is not None
    assert MODEL_REGISTRY.get("clip") is None          # falls back to BaseVideoModel, as in the reference


def test_model_forward_backward_matches_reference_golden(gpu_lib):
    """fp32 parity mode through the nn.Module / autograd surface vs the reference's own outputs."""
    from dist_amd import synth
    from dist_amd.models.base.builder import build_model
    from dist_amd.models.utils import losses
    cfg = tiny_cfg("TRAIN.FP32_PARITY", "true", "TRAIN.BATCH_SIZE", "2")
    model, ema = build_model(cfg)
    assert ema is None
    g = synth.geometry("tiny")
    sd = {"backbone.base_encoder." + k: torch.from_numpy(v) for k, v in synth.state_dict(g).items()}
    from dist_amd.utils.checkpoint import normalize_state_dict
    model.backbone.base_encoder.load_state_dict(normalize_state_dict({"model_state": sd}), strict=False)
    names = [n for n, _ in model.named_parameters()]
    assert "backbone.base_encoder.dist_net.temporal_stem.weight" in names
    assert "backbone.base_encoder.visual.transformer.resblocks.1.attn.in_proj_weight" in names
    clip = model.backbone.base_encoder
    clip.text_features = torch.from_numpy(synth.text_features(g)).cuda()          # cached text features (clip.py:437-452)
    clip.text_logits = clip.text_features
    video = torch.from_numpy(synth.video(g, 2)).cuda()
    tgt = torch.from_numpy(synth.soft_target(g, 2)[0]).cuda()
    texts = torch.zeros(g.K, 77, dtype=torch.long, device="cuda")
    model.train()
    preds, out = model({"video": video, "texts": texts})
    assert preds.shape == (2, g.K) and out["logits_per_image"].shape == (2, 1, g.K) and out["vid_logits"].shape == (2, 1, g.E)
    gold = np.load(os.path.join(GOLD, "tiny.npz"))
    torch.testing.assert_close(preds.detach().cpu().double(), torch.from_numpy(gold["logits"]).double(), rtol=1e-3, atol=1e-4)
    # img_logits: the frozen ViT's own per-frame embedding ln_post(cls) @ proj, [b*t, E] (reference clip.py:291-298,532) - round 2 returned None
    assert out["img_logits"].shape == (2 * g.t, g.E) and not out["img_logits"].requires_grad
    torch.testing.assert_close(out["img_logits"].cpu().double(), torch.from_numpy(gold["img_logits"]).double(), rtol=1e-3, atol=1e-4)
    loss, _, _ = losses.calculate_loss(cfg, preds, out, {"supervised": tgt}, 0)
    assert abs(float(loss.detach()) - float(gold["loss"])) < 1e-4
    loss.backward()
    pd = dict(model.named_parameters())
    for n in ("dist_net.temporal_stem.weight", "dist_net.proj", "dist_net.integration_nets.1.ln.bias", "logit_scale"):
        got = pd["backbone.base_encoder." + n].grad.cpu().double()
        ref = torch.from_numpy(gold["grad." + n]).double()
        assert float((got - ref).abs().max() / (ref.abs().max() + 1e-12)) < 2e-3, n
    assert all(p.grad is None for n, p in pd.items() if ".visual." in n)           # frozen ViT: no gradients
    model.eval()
    with torch.no_grad():
        p_eval, _ = model({"video": video, "texts": texts})
    torch.testing.assert_close(p_eval.sum(1).cpu(), torch.ones(2), rtol=1e-4, atol=1e-4)   # head softmax at eval


def test_text_tower_matches_the_reference_golden(gpu_lib):
    """`encode_text` / `cache_text` (reference clip.py:420-452; the frozen text tower runs once per label set and feeds every logit):
    the reference's own text features for a procedural tower and procedural label tokens (oracle/make_golden_text.py ->
    tests/golden/text_tiny.npz) against this repository's restatement, built through the same `clip.build_model` shape inference."""
    from dist_amd import synth
    from dist_amd.models.base import clip as C
    g = synth.geometry("tiny")
    gold = np.load(os.path.join(GOLD, "text_tiny.npz"))
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.state_dict(g).items()}
    sd.update({k: torch.from_numpy(v.copy()) for k, v in synth.text_tower_state_dict(embed=g.E).items()})
    model = C.build_model(tiny_cfg("TRAIN.FP32_PARITY", "true", "TRAIN.BATCH_SIZE", "2"), dict(sd)).cuda()
    assert model.context_length == int(gold["context"]) and model.transformer.layers == int(gold["layers"])
    assert model.transformer.resblocks[0].attn.num_heads == int(gold["heads"])
    tokens = torch.from_numpy(synth.label_tokens(g.K)).cuda()
    with torch.no_grad():
        feats, eot, _ = model.encode_text(tokens)
    torch.testing.assert_close(feats.cpu().double(), torch.from_numpy(gold["text_features"]).double(), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(eot.cpu().double(), torch.from_numpy(gold["eot_features"]).double(), rtol=1e-4, atol=1e-5)
    cached, _, _ = model.cache_text(tokens)
    again, _, _ = model.cache_text(tokens)
    assert cached.data_ptr() == again.data_ptr()                                   # computed once per label set
    torch.testing.assert_close(cached.cpu(), feats.cpu().float(), rtol=0, atol=0)
    # and these features drive the logits of a forward pass
    video = torch.from_numpy(synth.video(g, 2)).cuda()
    out = model(video.permute(0, 2, 1, 3, 4).reshape(2 * g.T, 3, g.res, g.res).contiguous(), tokens)
    v = out["vid_logits"][:, 0].double().cpu()
    tf = torch.from_numpy(gold["text_features"]).double()
    want = float(np.exp(np.log(1 / 0.07))) * v @ (tf / tf.norm(dim=1, keepdim=True)).t()
    torch.testing.assert_close(out["logits_per_image"].double().cpu(), want, rtol=1e-3, atol=1e-4)


def test_optimizer_groups_and_fused_step(gpu_lib):
    from dist_amd.models.base.builder import build_model
    from dist_amd.models.utils import optimizer as optim
    cfg = tiny_cfg("TRAIN.FP32_PARITY", "true")
    model, _ = build_model(cfg)
    groups = optim.construct_DiST_optimizer(model, cfg)
    assert [g["weight_decay"] for g in groups] == [0.0, 1e-4, 0.0, 1e-4, 0.0]
    assert all(g["lr_mult"] == 10.0 for g in groups)
    n_params = sum(len(g["params"]) for g in groups)
    assert n_params == len(model.backbone.base_encoder.engine.tables[0])
    opt = optim.construct_optimizer(model, cfg)
    optim.set_lr(opt, optim.get_epoch_lr(0.5, cfg))
    eng = model.backbone.base_encoder.engine
    before = eng.theta.clone()
    eng.grads.normal_()
    opt.step()
    assert float((eng.theta - before).abs().max()) > 0
    # a vanilla torch optimizer on the parameter views is picked up as well (re-pack by version stamp)
    p = dict(model.named_parameters())["backbone.base_encoder.dist_net.proj"]
    assert p.data_ptr() == eng.view("dist_net.proj").data_ptr()


def test_checkpoint_roundtrip_and_reference_key_spellings(gpu_lib, tmp_path):
    from dist_amd.models.base.builder import build_model
    from dist_amd.utils import checkpoint as cu
    cfg = tiny_cfg()
    model, _ = build_model(cfg)
    eng = model.backbone.base_encoder.engine
    w0 = eng.view("dist_net.proj").clone()
    path = cu.save_checkpoint(str(tmp_path), model, None, 0, cfg)
    ck = torch.load(path, map_location="cpu")
    assert "backbone.base_encoder.dist_net.proj" in ck["model_state"] and not any(".visual." in k for k in ck["model_state"])
    eng.view("dist_net.proj").zero_()
    cu.load_checkpoint(path, model)
    torch.testing.assert_close(eng.view("dist_net.proj"), w0)
    # process_dist_cpkt.py spelling: ladder_net.* inside model_state
    renamed = {"model_state": {k.replace("dist_net.", "ladder_net."): v for k, v in ck["model_state"].items()}}
    p2 = str(tmp_path / "old.pyth")
    torch.save(renamed, p2)
    eng.view("dist_net.proj").zero_()
    cu.load_checkpoint(p2, model)
    torch.testing.assert_close(eng.view("dist_net.proj"), w0)


def test_runs_entry_point_trains_and_tests(gpu_lib, tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "runs", "run.py"), "--cfg", TINY, "OUTPUT_DIR", str(tmp_path)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "top1_acc" in out.stdout and "Finish running" in out.stdout
    assert os.path.isdir(os.path.join(str(tmp_path), "checkpoints"))


def test_train_loop_pipelines_the_frozen_vit_without_changing_results(gpu_lib):
    """runs/train.py with TRAIN.PIPELINE_VIT (look-ahead of one batch: the frozen-ViT pass of batch n+1 runs beside the step
    of batch n through CLIP.prefetch_video / adopt_prefetched) gives the per-iteration statistics of the serial loop."""
    sys.path.insert(0, ROOT)
    from runs import train as T

    def run(pipe, host=False):
        cfg = tiny_cfg("TRAIN.FP32_PARITY", "true", "TRAIN.PIPELINE_VIT", "true" if pipe else "false", "TRAIN.EVAL_PERIOD", "0",
                       "TRAIN.CHECKPOINT_PERIOD", "0", "OPTIMIZER.MAX_EPOCH", "1", "DATA.SYNTHETIC_HOST", "true" if host else "false")
        logs = []
        orig = T.train_epoch

        def spy(*a, **k):
            return orig(*a, **k, log=logs.append)
        T.train_epoch = spy
        try:
            T.train(cfg)
        finally:
            T.train_epoch = orig
        return logs

    a, b = run(False), run(True)
    assert len(a) == len(b) == 4
    for x, y in zip(a, b):
        assert abs(x["loss"] - y["loss"]) < 1e-4 * max(1.0, abs(x["loss"])), (x, y)
        assert x["top1_err"] == y["top1_err"] and x["lr"] == y["lr"]
    # round 6: the same batches handed over in HOST memory (the reference loader's form): pinned staging + a copy stream two batches ahead
    # (dist_amd/utils/staging.py) - the same clips reach the same kernels: the same statistics (up to the fp32 atomics of the parity mode), pipelined or not
    for c in (run(True, host=True), run(False, host=True)):
        assert len(c) == 4
        for x, y in zip(b, c):
            assert abs(x["loss"] - y["loss"]) < 1e-4 * max(1.0, abs(x["loss"])) and x["top1_err"] == y["top1_err"] and x["lr"] == y["lr"], (x, y)


def _tiny_model(*extra):
    from dist_amd import synth
    from dist_amd.models.base.builder import build_model
    from dist_amd.utils.checkpoint import normalize_state_dict
    cfg = tiny_cfg("TRAIN.FP32_PARITY", "true", "TRAIN.BATCH_SIZE", "2", *extra)
    model, _ = build_model(cfg)
    g = synth.geometry("tiny")
    sd = {"backbone.base_encoder." + k: torch.from_numpy(v) for k, v in synth.state_dict(g).items()}
    model.backbone.base_encoder.load_state_dict(normalize_state_dict({"model_state": sd}), strict=False)
    clip = model.backbone.base_encoder
    clip.text_features = torch.from_numpy(synth.text_features(g)).cuda()
    clip.text_logits = clip.text_features
    return cfg, model, clip, g


def test_distnetwork_forward_consumes_the_callers_mid_feat(gpu_lib):
    """reference dist.py:222-247: DiSTNetwork.forward(input) reads input['mid_feat']['img'][layer_id] ([L, b*t, C]) and input['images'].
    Round 3 ignored them and used the engine's own slot.  Proof that the CALLER's tensors are consumed: the features of clip B (taken from a
    pass over B, in the reference's layout) handed over while the engine's slot holds the pass over clip A must give B's embedding."""
    from dist_amd import synth
    cfg, model, clip, g = _tiny_model()
    eng = clip.engine
    L_, bt = g.N + 1, 2 * g.t
    vA = torch.from_numpy(synth.video(g, 2, seed=1)).cuda()
    vB = torch.from_numpy(synth.video(g, 2, seed=7)).cuda()
    texts = torch.zeros(g.K, 77, dtype=torch.long, device="cuda")
    model.eval()
    with torch.no_grad():
        outB = clip.forward_video(vB, texts)
        want_logits, want_vid = outB["logits_per_image"].clone(), outB["vid_logits"][:, 0].clone()
        # block outputs of the pass over B in the reference's layout [L, b*t, width] (clip.py:282-300: sequence first)
        midB = {i: eng.debug(f"feat.{i}").view(bt, L_, g.d).permute(1, 0, 2).contiguous().float().clone() for i in range(g.layers)}
        outA = clip.forward_video(vA, texts)                                   # the slot now holds clip A
        assert float((outA["logits_per_image"] - want_logits).abs().max()) > 1e-3
        inp = {"mid_feat": {"img": midB}, "images": vB.permute(0, 2, 1, 3, 4).reshape(2 * g.T, 3, g.res, g.res).contiguous(),
               "text_features": clip.text_features}
        vid, inp2 = clip.dist_net(inp)
    assert inp2 is inp
    torch.testing.assert_close(vid, want_vid, rtol=0, atol=0)                  # fp32 mode: the copies are exact, the kernels the same
    torch.testing.assert_close(inp["logits_per_image"], want_logits, rtol=0, atol=0)
    # without mid_feat the engine's own pass is used (what CLIP.forward does)
    with torch.no_grad():
        vid_own, _ = clip.dist_net({"text_features": clip.text_features})
    torch.testing.assert_close(vid_own, want_vid, rtol=0, atol=0)               # (the slot still holds B's imported features)
    # bf16 tensors and wrong shapes
    with pytest.raises(Exception):
        clip.dist_net({"mid_feat": {"img": {i: midB[i][:, :1] for i in midB}}, "images": vB, "text_features": clip.text_features})
    with pytest.raises(Exception):
        clip.dist_net({"mid_feat": {"img": midB}, "text_features": clip.text_features})      # frames missing


def test_train_loop_loss_is_the_engines_dist_loss(gpu_lib):
    """runs/train.py computes the loss through models/utils/losses.calculate_loss: with the engine's logits that is dist_loss (the launch bench.py
    times), not a torch expression - same value and the same gradients as the torch form."""
    from dist_amd import synth
    from dist_amd.models.utils import losses
    cfg, model, clip, g = _tiny_model()
    video = torch.from_numpy(synth.video(g, 2)).cuda()
    tgt = torch.from_numpy(synth.soft_target(g, 2)[0]).cuda()
    texts = torch.zeros(g.K, 77, dtype=torch.long, device="cuda")
    model.train()
    pd = dict(model.named_parameters())
    grads = []
    for use_engine in (True, False):
        for p in pd.values():
            p.grad = None
        preds, out = model({"video": video, "texts": texts})
        assert out["_dist_engine"] is clip.engine
        if use_engine:
            calls = []
            orig = clip.engine.loss
            clip.engine.loss = lambda t: (calls.append(1), orig(t))[1]
            loss, _, _ = losses.calculate_loss(cfg, preds, out, {"supervised": tgt}, 0)
            clip.engine.loss = orig
            assert calls == [1]                                                # went through dist_loss
        else:
            loss = losses.SoftTargetCrossEntropy()(preds, tgt)                 # the torch expression
        loss.backward()
        grads.append((float(loss.detach()), {n: p.grad.clone() for n, p in pd.items() if p.grad is not None}))
    assert abs(grads[0][0] - grads[1][0]) < 1e-5
    # hard labels take the same kernel on their one-hot form
    preds, out = model({"video": video, "texts": texts})
    hard = tgt.argmax(1)
    lh, names, _ = losses.calculate_loss(cfg, preds, out, {"supervised": hard}, 0)
    assert "cross_entropy" in names and abs(float(lh.detach()) - float(torch.nn.functional.cross_entropy(preds.detach(), hard))) < 1e-5
    assert grads[0][1].keys() == grads[1][1].keys() and len(grads[0][1]) > 100
    for n in grads[0][1]:
        a, b = grads[0][1][n].double(), grads[1][1][n].double()
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-9, n


def test_engine_loss_is_only_taken_for_the_last_forwards_own_logits(gpu_lib):
    """dist_loss reads the video embedding the LAST branch forward left in the engine.  If another forward ran between producing
    `out` and calculate_loss (two views, an evaluation pass), or `preds` does not come from that dictionary's logits, the loss must be the torch
    expression on `preds` - never silently the other batch's."""
    from dist_amd import synth
    from dist_amd.models.utils import losses
    cfg, model, clip, g = _tiny_model()
    va = torch.from_numpy(synth.video(g, 2)).cuda()
    vb = torch.from_numpy(synth.video(g, 2, seed=9)).cuda()
    tgt = torch.from_numpy(synth.soft_target(g, 2)[0]).cuda()
    texts = torch.zeros(g.K, 77, dtype=torch.long, device="cuda")
    model.train()
    calls = []
    orig = clip.engine.loss
    clip.engine.loss = lambda t: (calls.append(1), orig(t))[1]
    try:
        preds_a, out_a = model({"video": va, "texts": texts})
        preds_b, out_b = model({"video": vb, "texts": texts})                 # the engine now holds clip B's embedding
        want_a = float(losses.SoftTargetCrossEntropy()(preds_a.detach(), tgt))
        la, _, _ = losses.calculate_loss(cfg, preds_a, out_a, {"supervised": tgt}, 0)
        assert calls == [] and abs(float(la.detach()) - want_a) < 1e-6        # stale stamp -> torch expression, clip A's loss
        lb, _, _ = losses.calculate_loss(cfg, preds_b, out_b, {"supervised": tgt}, 0)
        assert calls == [1]                                                    # the last forward's own logits -> dist_loss
        assert abs(float(lb.detach()) - float(losses.SoftTargetCrossEntropy()(preds_b.detach(), tgt))) < 1e-5
        # logits that are not derived from this dictionary (a detached copy) take the torch expression too
        preds_c, out_c = model({"video": va, "texts": texts})
        foreign = preds_c.detach().clone().requires_grad_(True)
        lc, _, _ = losses.calculate_loss(cfg, foreign, out_c, {"supervised": tgt}, 0)
        assert calls == [1] and abs(float(lc.detach()) - want_a) < 1e-5
        lc.backward()
        assert foreign.grad is not None
    finally:
        clip.engine.loss = orig


def test_img_logits_is_lazy_and_guarded(gpu_lib):
    from dist_amd import synth
    cfg, model, clip, g = _tiny_model()
    video = torch.from_numpy(synth.video(g, 2)).cuda()
    texts = torch.zeros(g.K, 77, dtype=torch.long, device="cuda")
    with torch.no_grad():
        out = clip.forward_video(video, texts)
        assert "img_logits" in out and not dict.__contains__(out, "img_logits")      # not computed yet
        v = out["img_logits"]
        assert v.shape == (2 * g.t, g.E) and dict.__contains__(out, "img_logits")
        out2 = clip.forward_video(video.clone(), texts)
        clip.forward_video(video, texts)                                               # the slot moves on
        with pytest.raises(Exception):
            out2["img_logits"]
