#!/usr/bin/env python3
"""Generate golden OUTPUT vectors from the real reference.  TEST INFRASTRUCTURE.

Runs ONLY in the build container (needs /root/reference, read-only).  It imports the
reference's own `models.base.clip` with stub modules for the third-party packages
the image lacks (timm, torchvision, simplejson, oss2: SURVEY.md Appendix B), builds
the reference CLIP+DiST model from a procedurally generated CLIP-format state-dict
(dist_amd/synth.py), runs the reference forward / backward on procedural frames and
writes ONLY outputs (logits, loss, gradients, intermediate activations or their
checksums) to tests/golden/*.npz.  No reference source is copied; the inputs are
regenerated from dist_amd/synth.py wherever the fixtures are consumed.

    python oracle/make_golden.py [name ...]   # writes tests/golden/{tiny,tiny3,tiny3_sel,b16_b2,l14_t8,b16_t32_b1,l14_t64_b1}.npz
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dist_amd import synth  # noqa: E402

REF = "/root/reference"


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()

        def forward(self, x):
            return x

    timm = mod("timm")
    timm_models = mod("timm.models")
    timm_models.__path__ = []
    mod("timm.models.layers", trunc_normal_=trunc_normal_, drop_path=lambda x, p=0.0, training=False: x,
        to_2tuple=lambda x: (x, x), DropPath=DropPath)
    mod("timm.models.registry", register_model=lambda f: f)
    timm.models = timm_models
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils", make_grid=None, save_image=None)
    mod("simplejson", dumps=lambda *a, **k: "")
    mod("oss2")


class NS(dict):
    """attribute-access config node (the reference only does cfg.A.B.C / hasattr)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def make_cfg(g):
    return NS(
        DATA=NS(NUM_INPUT_FRAMES=g.T, SPARSE_SAMPLE_ALPHA=g.alpha),
        TRAIN=NS(HALF_PRECISION=False),
        TEST=NS(),
        VIDEO=NS(
            BACKBONE=NS(FREEZE_TEXT=True, FREEZE_VISUAL=True, RECORD_VIS_MID_FEAT=True,
                        ATTEN_BLOCK="ResidualAttentionBlockMid", META_ARCH_NAME="ViT-B-16",
                        DIST=NS(INTEGRATION_DIM=g.Ci, TEMPORAL_DIM=g.Ct, TEMPORAL_KERNEL_SIZE=g.tk,
                                TEMPORAL_CONV_MLP_RATIO=g.tn_ratio, INTEGRATION_MLP_RATIO=g.ffn_ratio,
                                INTEGRATION_TEMPORAL_MLP_RATIO=g.int_t_ratio, ADA_POOLING_LAYERS=g.ada,
                                SELECTED_LAYERS=list(g.selected), S_PATCH_SIZE=g.patch,
                                T_PATCH_SIZE=g.tpatch)),
            HEAD=NS(NUM_CLASSES=g.K)),
    )


def text_tower_stub(g):
    """Smallest text tower build_model() can infer shapes from; it is never run
    (text features are preset, reference clip.py:441-451)."""
    w = 64
    sd = {
        "text_projection": torch.zeros(w, g.E), "positional_embedding": torch.zeros(8, w),
        "token_embedding.weight": torch.zeros(16, w), "ln_final.weight": torch.ones(w), "ln_final.bias": torch.zeros(w),
    }
    p = "transformer.resblocks.0."
    sd.update({p + "attn.in_proj_weight": torch.zeros(3 * w, w), p + "attn.in_proj_bias": torch.zeros(3 * w),
               p + "attn.out_proj.weight": torch.zeros(w, w), p + "attn.out_proj.bias": torch.zeros(w),
               p + "ln_1.weight": torch.ones(w), p + "ln_1.bias": torch.zeros(w),
               p + "ln_2.weight": torch.ones(w), p + "ln_2.bias": torch.zeros(w),
               p + "mlp.c_fc.weight": torch.zeros(4 * w, w), p + "mlp.c_fc.bias": torch.zeros(4 * w),
               p + "mlp.c_proj.weight": torch.zeros(w, 4 * w), p + "mlp.c_proj.bias": torch.zeros(w)})
    return sd


def build_reference(g, seed=0, dtype=torch.float32):
    from models.base import clip  # the reference's own module
    sd_np = synth.state_dict(g, seed)
    sd = {k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}
    sd.update(text_tower_stub(g))
    model = clip.build_model(make_cfg(g), dict(sd))
    missing = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing[0] if k.startswith(("visual.", "dist_net."))], missing[0]
    assert not missing[1], missing[1]
    model.prediction_fusion_enable = False        # SURVEY §0 defect 1 work-around (instance attr)
    model = model.to(dtype)
    model.train()
    return model, sd_np


def checksum(t):
    t = t.detach().double().flatten()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item()] + t[:16].tolist(), dtype=np.float64)


def run(gname, b, full, dtype=torch.float32, with_steps=False):
    g = synth.geometry(gname)
    model, sd_np = build_reference(g, dtype=dtype)
    video = torch.from_numpy(synth.video(g, b)).to(dtype)
    tf = torch.from_numpy(synth.text_features(g)).to(dtype)
    tgt, _ = synth.soft_target(g, b)
    tgt = torch.from_numpy(tgt).to(dtype)
    model.text_features = tf.clone()
    model.text_logits = tf.clone()
    frames = video.permute(0, 2, 1, 3, 4).reshape(b * g.T, 3, g.res, g.res)       # backbone.py:228-233
    texts = torch.zeros(g.K, 77, dtype=torch.long)

    inter = {}

    def hook(name):
        def f(mod, inp, out):
            inter[name] = out[0] if isinstance(out, tuple) else out
        return f
    hs = []
    for i in range(g.nsel):                      # one DiST layer per SELECTED ViT block (dist.py:170-190)
        hs.append(model.dist_net.temporal_nets[i].register_forward_hook(hook(f"tn_out.{i}")))
        hs.append(model.dist_net.integration_nets[i].register_forward_hook(hook(f"int_out.{i}")))
    for i in range(g.layers):
        hs.append(model.visual.transformer.resblocks[i].register_forward_hook(hook(f"feat.{i}")))
    hs.append(model.dist_net.temporal_stem.register_forward_hook(hook("stem")))

    out = model(frames, texts)
    logits = out["logits_per_image"]
    loss = torch.sum(-tgt * torch.nn.functional.log_softmax(logits, dim=-1), dim=-1).mean()   # losses.py:29-31
    loss.backward()
    for h in hs:
        h.remove()

    res = {"logits": logits.detach().numpy(), "vid_logits": out["vid_logits"].detach().numpy()[:, 0],
           "loss": np.array(loss.item()),
           # the frozen ViT's own clip-free embedding of every sampled frame: ln_post(cls) @ proj, [b*t, E] (clip.py:291-298,532)
           "img_logits": out["img_logits"].detach().numpy()}
    # intermediates converted to the build's layout
    L, t = g.L, g.t
    for k, v in inter.items():
        v = v.detach()
        if k.startswith("feat.") or k.startswith("int_out."):          # [L, b*t, C] -> [b,t,L,C]
            v = v.reshape(L, b, t, -1).permute(1, 2, 0, 3)
        else:                                                            # [b,Ct,T,H,W] -> [b,T,N,Ct]
            v = v.permute(0, 2, 3, 4, 1).reshape(b, g.T, g.N, -1)
        res["act." + k] = v.numpy() if full else checksum(v)
    ngrad = 0
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        ngrad += 1
        res["grad." + n] = p.grad.numpy().copy() if full else checksum(p.grad)
        if not full:
            res["gnorm." + n] = np.array(p.grad.double().norm().item())
    res["n_grad_tensors"] = np.array(ngrad)

    if with_steps:
        # fixture "(2)" of SURVEY §8(c): weights after 1 and 3 AdamW steps with the INTENDED
        # param groups (the released constructor is broken), lr 3.2e-4, wd 1e-4 / 0.
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from dist_oracle import dist_param_groups
        shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
        gid = dist_param_groups(shapes)
        groups = [{"params": [], "weight_decay": wd} for wd in (0.0, 1e-4, 0.0, 1e-4, 0.0)]
        for n, p in model.named_parameters():
            if n in gid:
                groups[gid[n]]["params"].append(p)
        res["group_sizes"] = np.array([[len(gr["params"]), sum(p.numel() for p in gr["params"])] for gr in groups])
        opt = torch.optim.AdamW(groups, lr=3.2e-4, betas=(0.9, 0.999), weight_decay=0.0)
        watch = ["dist_net.temporal_stem.weight", "dist_net.proj", "dist_net.adapooling_nets.0.positional_embedding",
                 "dist_net.integration_nets.0.ffn.c_fc.weight", "dist_net.integration_nets.1.ln.bias",
                 "dist_net.adapooling_nets.1.spatial_transformer.attn.in_proj_weight"]
        pd = dict(model.named_parameters())
        for step in range(1, 4):
            if step > 1:
                opt.zero_grad()
                lg = model(frames, texts)["logits_per_image"]
                ls = torch.sum(-tgt * torch.nn.functional.log_softmax(lg, dim=-1), dim=-1).mean()
                ls.backward()
                res[f"loss_step{step}"] = np.array(ls.item())
            opt.step()
            if step in (1, 3):
                for n in watch:
                    res[f"w{step}." + n] = pd[n].detach().numpy().copy()
    return res


def main():
    assert os.path.isdir(REF), "the reference is only present in the build container"
    _stub_modules()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    jobs = [("tiny", "tiny", 2, True, True), ("tiny3", "tiny3", 3, False, False),
            # SELECTED_LAYERS = [0, 2] of three ViT blocks (round 4): the reference takes any subset (dist.py:170-190); full tensors
            ("tiny3_sel", "tiny3_sel", 3, True, False),
            # TEMPORAL_CONV_MLP_RATIO = 2, INTEGRATION_MLP_RATIO = 0.5 (round 4): hidden widths other than the released yamls' (dist.py:20-25, 51-58)
            ("tiny_ratio", "tiny_ratio", 2, True, False),
            ("b16_b2", "b16_8+16f", 2, False, False), ("l14_t8", "l14_tiny_t", 1, False, False),
            # BASELINE configs 3 and 4 / 5 at their real frame counts (T = 32, T = 64), one clip: the temporal branch at the sizes the bench
            # configurations run (round 3; the full-batch GPU tests compare rows of a b = 32 / 8 / 16 run with these through batch invariance)
            ("b16_t32_b1", "b16_16+32f", 1, False, False), ("l14_t64_b1", "l14_32+64f", 1, False, False)]
    only = sys.argv[1:]
    for fname, gname, b, full, steps in jobs:
        if only and fname not in only:
            continue
        res = run(gname, b, full, with_steps=steps)
        # fixtures are stored in fp32 (tiny: full tensors) / fp64 (checksums) to stay small
        res = {k: (v.astype(np.float32) if (full and v.dtype == np.float64 and v.ndim > 0) else v) for k, v in res.items()}
        path = os.path.join(out_dir, fname + ".npz")
        np.savez_compressed(path, **res)
        print(fname, "->", path, os.path.getsize(path) // 1024, "KiB; loss", float(res["loss"]),
              "grad tensors", int(res["n_grad_tensors"]))


if __name__ == "__main__":
    main()
