"""Checkpoint key compatibility (reference process_dist_cpkt.py:10-30, utils/checkpoint.py:277-347).

Not GPU: `dist_amd.utils.checkpoint.rename_model_state` / `normalize_state_dict` against tests/golden/ckpt_rename.json, the output of
the REFERENCE's own `rename_model_state` over every dist_net tensor name in its pre-release spelling (oracle/make_golden_ckpt.py).
GPU: a checkpoint written in the pre-release spelling (and with the DDP `module.` prefix) loads into the engine with no dist_net
tensor left unmatched, and round-trips bit for bit."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "ckpt_rename.json")))


def test_rename_table_matches_the_reference_for_every_tensor_name():
    from dist_amd import synth
    from dist_amd.utils import checkpoint as cu
    assert sum("ladder_net" in k for k in GOLD) >= 383 and not any("ladder_net" in v for v in GOLD.values())
    for old, new in GOLD.items():
        assert cu.rename_key(old) == new, (old, cu.rename_key(old), new)
    got = cu.rename_model_state({k: i for i, k in enumerate(GOLD)})
    assert list(got.keys()) == [GOLD[k] for k in GOLD] and list(got.values()) == list(range(len(GOLD)))      # order and values kept
    # every tensor the engine owns is reachable from the pre-release spelling
    g = synth.geometry("b16_8+16f")
    released = {v[len(cu.PREFIX):] for v in GOLD.values()}
    assert set(synth.dist_net_shapes(g)) <= released and len(synth.dist_net_shapes(g)) == 383


def test_normalize_strips_wrappers_and_prefixes():
    from dist_amd.utils import checkpoint as cu
    sd = {"model_state": {"module.backbone.base_encoder.ladder_net.s2t_fuse_nets.3.linear_fuse.weight": 1,
                          "backbone.base_encoder.ladder_net.proj": 2, "backbone.base_encoder.visual.proj": 3,
                          "backbone.base_encoder.dist_net.input_linears.0.bias": 4, "logit_scale": 5}}
    assert cu.normalize_state_dict(sd) == {"dist_net.integration2temporal_nets.3.linear_fuse.weight": 1, "dist_net.proj": 2, "visual.proj": 3,
                                           "dist_net.input_linears.0.bias": 4, "logit_scale": 5}
    assert cu.normalize_state_dict({"visual.proj": 7}) == {"visual.proj": 7}                                   # a bare CLIP state-dict passes through


@pytest.mark.gpu
def test_pre_release_checkpoint_loads_completely(gpu_lib, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_dropin_gpu import tiny_cfg
    from dist_amd.models.base.builder import build_model
    from dist_amd.utils import checkpoint as cu
    cfg = tiny_cfg()
    model, _ = build_model(cfg)
    eng = model.backbone.base_encoder.engine
    path = cu.save_checkpoint(str(tmp_path), model, None, 0, cfg)
    ck = torch.load(path, map_location="cpu")
    inverse = {v: k for k, v in GOLD.items() if "ladder_net" in k}
    old = {"module." + inverse.get(k, k): v for k, v in ck["model_state"].items()}
    assert sum("ladder_net" in k for k in old) == len(eng.tables[0]) and any("final_temporal_nets" in k for k in old)
    p2 = str(tmp_path / "pre_release.pyth")
    torch.save({"model_state": old}, p2)
    want = {n: eng.view(n).clone() for n in eng.tables[0]}
    for n in eng.tables[0]:
        eng.view(n).zero_()
    epoch = cu.load_checkpoint(p2, model)
    assert epoch == -1                                                          # no "epoch" entry (reference :341-345)
    missing, unexpected = cu.load_checkpoint.last_mismatch
    assert not [k for k in missing if k.startswith("dist_net.")] and not unexpected
    for n, w in want.items():
        assert torch.equal(eng.view(n), w), n
    cu.load_checkpoint(p2, model, strict=True)                                  # a complete dist_net-only file is a complete file
    half = {k: v for k, v in old.items()}
    half["module.backbone.base_encoder.visual.proj"] = torch.zeros_like(model.backbone.base_encoder.visual.proj)
    p3 = str(tmp_path / "half_visual.pyth"); torch.save({"model_state": half}, p3)
    with pytest.raises(KeyError, match="visual"):
        cu.load_checkpoint(p3, model, strict=True)                              # SOME frozen-ViT tensors but not all of them
    with pytest.raises(AssertionError):
        cu.load_checkpoint(str(tmp_path / "nope.pyth"), model)


@pytest.mark.gpu
def test_strict_load_of_a_dist_net_only_checkpoint(gpu_lib, tmp_path):
    """`save_checkpoint(full=False)` writes dist_net.* (+ logit_scale) only; `load_checkpoint(strict=True)` accepts exactly that - the text
    tower / frozen-ViT keys of the model that the file does not carry are not an error - and still refuses a file with a dist_net tensor
    missing or a stray dist_net / ladder_net key (the docstring's contract; round 2 raised on ANY mismatch)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_dropin_gpu import tiny_cfg
    from dist_amd.models.base.builder import build_model
    from dist_amd.utils import checkpoint as cu
    cfg = tiny_cfg()
    model, _ = build_model(cfg)
    eng = model.backbone.base_encoder.engine
    path = cu.save_checkpoint(str(tmp_path), model, None, 0, cfg, full=False)
    want = eng.theta.clone()
    eng.theta.zero_()
    cu.load_checkpoint(path, model, None, strict=True)
    assert torch.equal(eng.theta, want)
    ck = torch.load(path, map_location="cpu")
    victim = next(k for k in ck["model_state"] if "temporal_stem.weight" in k)
    broken = dict(ck, model_state={k: v for k, v in ck["model_state"].items() if k != victim})
    p2 = str(tmp_path / "missing.pyth"); torch.save(broken, p2)
    with pytest.raises(KeyError, match="temporal_stem"):
        cu.load_checkpoint(p2, model, None, strict=True)
    stray = dict(ck, model_state=dict(ck["model_state"], **{"backbone.base_encoder.dist_net.no_such_tensor": torch.zeros(1)}))
    p3 = str(tmp_path / "stray.pyth"); torch.save(stray, p3)
    with pytest.raises(KeyError, match="no_such_tensor"):
        cu.load_checkpoint(p3, model, None, strict=True)
    cu.load_checkpoint(p3, model, None, strict=False)                  # non-strict like the reference: reported, not raised
    assert "dist_net.no_such_tensor" in cu.load_checkpoint.last_mismatch[1]


def _module_from_state_dict(sd):
    """nested nn.Modules whose state_dict() is exactly `sd` (parameter names with dots become sub-modules), scriptable"""
    class Node(torch.nn.Module):
        def forward(self):
            return 0
    root = Node()
    for name, t in sd.items():
        mod = root
        parts = name.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, Node())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], torch.nn.Parameter(t.clone(), requires_grad=False))
    return root


@pytest.mark.gpu
def test_openai_clip_torchscript_file_loads(gpu_lib, tmp_path):
    """The published CLIP weights are TorchScript archives (`ViT-B-16.pt`): the reference reads them with torch.jit.load(...).state_dict()
    (clip.py:624-627).  No such file exists in this environment, so one is WRITTEN here - a scripted module whose state-dict is a procedural
    CLIP state-dict (visual tower + text tower) - and `clip.load(cfg)` must build the model from it: shape inference, every frozen tensor in place."""
    from dist_amd import synth
    from dist_amd.models.base import clip as C
    from tests.test_dropin_gpu import tiny_cfg
    g = synth.geometry("tiny")
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.state_dict(g).items() if not k.startswith("dist_net.")}
    sd.update({k: torch.from_numpy(v.copy()) for k, v in synth.text_tower_state_dict(embed=g.E).items()})
    sd["logit_scale"] = sd["logit_scale"].reshape(())
    path = os.path.join(str(tmp_path), "ViT-tiny.pt")
    torch.jit.save(torch.jit.script(_module_from_state_dict(sd)), path)
    assert set(torch.jit.load(path, map_location="cpu").state_dict()) == set(sd)
    cfg = tiny_cfg("TRAIN.FP32_PARITY", "true", "TRAIN.BATCH_SIZE", "2", "VIDEO.BACKBONE.PRETRAIN_WEIGHT_PATH", path, "VIDEO.BACKBONE.SYNTHETIC_INIT", "false")
    model = C.load(cfg)
    got = model.state_dict()
    for k, v in sd.items():
        if k.startswith(("visual.", "transformer.", "token_embedding", "positional_embedding", "ln_final", "text_projection")):
            assert k in got and torch.equal(got[k].cpu().float().reshape(v.shape), v.float()), k
    assert model.engine.cfg.width == g.d and model.engine.cfg.layers == g.layers and model.engine.cfg.embed_dim == g.E
    # and it runs: the frozen towers feed a forward pass
    video = torch.from_numpy(synth.video(g, 2)).cuda()
    tokens = torch.from_numpy(synth.label_tokens(g.K)).cuda()
    with torch.no_grad():
        out = model.cuda()(video.permute(0, 2, 1, 3, 4).reshape(2 * g.T, 3, g.res, g.res).contiguous(), tokens)
    assert out["logits_per_image"].shape == (2, g.K) and torch.isfinite(out["logits_per_image"]).all()
