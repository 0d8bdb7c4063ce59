"""CLIP + DiST model behind the reference's module surface (reference models/base/clip.py).

Same class / function names, constructor inference rules and state-dict keys as the reference
(`build_model(cfg, state_dict)` infers width / layers / patch / resolution / embed dim from tensor
shapes, clip.py:564-592; `load(cfg)` reads a `.pyth` state-dict or an OpenAI TorchScript archive,
clip.py:614-629), but every tensor operation of the hot path runs in libdist_amd.so:

  * `visual.*` (frozen ViT) and `dist_net.*` parameters are nn.Parameters that are VIEWS into the engine's
    flat fp32 buffers, so `state_dict()` / `load_state_dict()` / torch optimizers see the reference layout;
  * `CLIP.forward` = one autograd.Function: dist_vit_forward + dist_branch_forward in forward,
    dist_branch_backward in backward (gradients only for dist_net.* and logit_scale, exactly the tensors
    the reference's autograd reaches because the ViT runs under eval()+no_grad, clip.py:454-458);
  * the frozen text tower runs once per label set and is cached (clip.py:437-452): it is outside the hot
    path and kept as plain torch modules.
"""
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from ... import lib as L
from ... import ops
from ...engine import Engine
from ...utils.registry import Registry

ATTEN_BLOCK_REGISTRY = Registry("AttentionBlock")


class LayerNorm(nn.LayerNorm):
    """fp32-internal LayerNorm (reference clip.py:181-187); used by the cached text tower only."""

    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


class QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class ResidualAttentionBlock(nn.Module):
    """Text-tower block (reference clip.py:112-135)."""

    def __init__(self, d_model, n_head, attn_mask=None, cfg=None, layer_id=0):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask

    def forward(self, x):
        m = self.attn_mask.to(dtype=x.dtype, device=x.device) if self.attn_mask is not None else None
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=m)[0]
        return x + self.mlp(self.ln_2(x))


class TextTransformer(nn.Module):
    """The text tower's block stack under the reference's attribute name (`transformer.resblocks.{i}.*`, clip.py:204-215): the
    OpenAI-CLIP state-dict keys must land on these modules.  (Until round 3 this was a bare nn.Sequential - keys `transformer.{i}.*`
    - so a non-strict load silently left the text tower at its random initialisation; tests/golden/text_tiny.npz pins it now.)"""

    def __init__(self, width, layers, heads, attn_mask=None):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x):
        return self.resblocks(x)


def causal_mask(context_length):
    """additive attention mask of the text tower (clip.py:411-417): -inf above the diagonal"""
    return torch.empty(context_length, context_length).fill_(float("-inf")).triu_(1)


def encode_text_with(m, text):
    """reference clip.py:420-435 on any module `m` that carries token_embedding, positional_embedding, transformer, ln_final and
    text_projection: (features [K, E], the end-of-text token's transformer output [K, width])"""
    x = m.token_embedding(text) + m.positional_embedding
    x = m.transformer(x.permute(1, 0, 2)).permute(1, 0, 2)                 # NLD -> LND -> NLD
    eot = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]                  # the end token has the largest id of its row
    return m.ln_final(eot) @ m.text_projection, eot


class TextTower(nn.Module):
    """The frozen text tower alone (same attribute / state-dict names as on CLIP): used where no GPU engine is needed."""

    def __init__(self, embed_dim, context_length, vocab_size, width, heads, layers):
        super().__init__()
        self.context_length = context_length
        self.transformer = TextTransformer(width, layers, heads, causal_mask(context_length))
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width).normal_(std=0.01))
        self.ln_final = LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, embed_dim).normal_(std=width ** -0.5))

    def encode_text(self, text):
        return encode_text_with(self, text)


@ATTEN_BLOCK_REGISTRY.register()
class ResidualAttentionBlockMid(nn.Module):
    """Parameter container of one frozen ViT block (reference clip.py:150-178); the arithmetic
    (LN, packed QKV GEMM, per-frame attention, MLP, mid_feat capture) runs inside dist_vit_forward."""

    def __init__(self, d_model, n_head, attn_mask=None, cfg=None, layer_id=0):
        super().__init__()
        self.layer_id = layer_id


class _FlatParams(nn.Module):
    """nn.Module whose parameters are views into an engine flat buffer, registered under dotted names."""

    def _adopt(self, engine, prefix, kind, requires_grad):
        for name in engine.tables[kind]:
            if not name.startswith(prefix):
                continue
            parts = name[len(prefix):].split(".")
            mod = self
            for p in parts[:-1]:
                if not hasattr(mod, p):
                    mod.add_module(p, nn.Module())
                mod = getattr(mod, p)
            mod.register_parameter(parts[-1], nn.Parameter(engine.view(name), requires_grad=requires_grad))


class VisionTransformer(_FlatParams):
    """Frozen CLIP ViT (reference clip.py:218-300): parameters live in the engine's `visual` buffer."""

    def __init__(self, cfg, engine, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.output_dim = input_resolution, output_dim
        self.num_frames = cfg.DATA.NUM_INPUT_FRAMES
        self._adopt(engine, "visual.", 1, requires_grad=False)

    def forward(self, x, others=None):
        raise L.DistError("the frozen ViT runs inside CLIP.forward (dist_vit_forward); call the CLIP module")


class DiSTParams(_FlatParams):
    pass


class _DistFunction(torch.autograd.Function):
    """forward: frozen ViT + DiST branch + cosine logits in HIP; backward: the hand-derived branch backward."""

    @staticmethod
    def forward(ctx, engine, video, text_features, logit_scale, *dist_params):
        engine.sync_packed(sum(p._version for p in dist_params))
        if not engine.has_features_for(video):            # not prefetched during the previous step (CLIP.prefetch_video)
            engine.vit_forward(video)
        logits, vid = engine.branch_forward(text_features)
        ctx.engine = engine
        ctx.n = len(dist_params)
        ctx.mark_non_differentiable(vid)
        return logits, vid

    @staticmethod
    def backward(ctx, dlogits, _dvid):
        eng = ctx.engine
        eng.backward(dlogits.contiguous().float(), zero_grads=True)
        grads = tuple(eng.view(n, grad=True) for n in eng.tables[0])
        return (None, None, None, eng.dlogit_scale.view(()).clone()) + grads


class _OutputDict(dict):
    """The reference's output dictionary (clip.py:532) with `img_logits` computed on first access."""
    _lazy_img = None

    def _img(self):
        clip, stamp = self._lazy_img
        if clip.engine._feat_stamp != stamp:
            raise L.DistError("img_logits was requested after the engine's feature slot moved on to another batch; read it before the next forward / adopt")
        v = clip.image_logits()
        dict.__setitem__(self, "img_logits", v)
        return v

    def __getitem__(self, k):
        if k == "img_logits" and not dict.__contains__(self, k) and self._lazy_img is not None:
            return self._img()
        return dict.__getitem__(self, k)

    def __contains__(self, k):
        return k == "img_logits" or dict.__contains__(self, k)

    def get(self, k, default=None):
        return self[k] if k in self else default

    def keys(self):
        return list(dict.keys(self)) + ([] if dict.__contains__(self, "img_logits") else ["img_logits"])


class CLIP(nn.Module):
    def __init__(self, cfg, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        from ..module_zoo.branches.dist import DiSTNetwork, engine_config
        self.cfg = cfg
        self.context_length = context_length
        self.num_frames = cfg.DATA.NUM_INPUT_FRAMES
        self.freeze_text = cfg.VIDEO.BACKBONE.FREEZE_TEXT
        self.freeze_visual = cfg.VIDEO.BACKBONE.FREEZE_VISUAL
        self.num_classes = cfg.VIDEO.HEAD.NUM_CLASSES
        if not (self.freeze_text and self.freeze_visual):
            raise L.DistError("dist_amd implements the DiST recipe: FREEZE_TEXT and FREEZE_VISUAL must be true")
        self.text_features = None
        self.text_logits = None
        self.prediction_fusion_enable = False       # the released reference reads this undefined attribute (clip.py:519)
        batch = _per_gpu_batch(cfg)
        dtype = torch.float32 if getattr(cfg.TRAIN, "FP32_PARITY", False) else torch.bfloat16
        self.engine = Engine(engine_config(cfg, vision_width, vision_layers, vision_patch_size, image_resolution, embed_dim, batch, dtype))
        self.visual = VisionTransformer(cfg, self.engine, image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        self.dist_net = DiSTNetwork(cfg, d_model=vision_width, width=vision_width, output_dim=embed_dim, engine=self.engine)
        self.logit_scale = nn.Parameter(self.engine.logit_scale.view(()))
        # frozen text tower (clip.py:359-371), plain torch: runs once per label set
        self.transformer = TextTransformer(transformer_width, transformer_layers, transformer_heads, self.build_attention_mask())
        self.vocab_size = vocab_size
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width).normal_(std=0.01))
        self.ln_final = LayerNorm(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim).normal_(std=transformer_width ** -0.5))
        for p in list(self.transformer.parameters()) + [self.token_embedding.weight, self.positional_embedding, self.text_projection] + \
                list(self.ln_final.parameters()):
            p.requires_grad_(False)
        self._dist_names = list(self.engine.tables[0])
        self._dist_params = [self._param_by_name(n) for n in self._dist_names]

    def _param_by_name(self, name):
        mod = self
        parts = name.split(".")
        for p in parts[:-1]:
            mod = getattr(mod, p)
        return getattr(mod, parts[-1])

    def build_attention_mask(self):
        return causal_mask(self.context_length)

    @property
    def dtype(self):
        return torch.float32

    def encode_text(self, text, others=None):
        feats, eot = encode_text_with(self, text)
        return feats, eot, others

    def cache_text(self, text, others=None):
        """reference clip.py:437-452: text features are computed once (no_grad) and reused."""
        if others is not None and "label_embeddings" in others:
            return others["label_embeddings"], None, others
        if self.text_features is None or text.size(0) != self.text_features.size(0):
            self.transformer.eval(); self.ln_final.eval()
            with torch.no_grad():
                tf, tl, others = self.encode_text(text, others)
            self.text_features, self.text_logits = tf.float().contiguous(), tl.clone()
        return self.text_features, self.text_logits, others

    def forward(self, image, text, others=None):
        """image: [b*T, 3, H, W] as the reference backbone passes it (backbone.py:228-233), or [b,3,T,H,W]."""
        if text is None:
            raise L.DistError("the DiST path needs label texts (reference backbone.py:250 is broken without them)")
        if image.dim() == 4:
            bt, c, h, w = image.shape
            image = image.view(bt // self.num_frames, self.num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        return self.forward_video(image, text, others)

    def prefetch_video(self, video):
        """Software pipelining over batches (dist_vit_prefetch): the ViT is frozen, so the pass over the NEXT batch can run on
        the engine's low-priority stream beside this batch's branch forward / backward / optimizer step.  Call with the
        next batch's (already augmented) clip before `forward` of the current one, then `adopt_prefetched()` after the
        optimizer step; `forward` of that clip then skips its ViT pass.  The tensor must not be written in between."""
        conv = video.contiguous().float()
        if conv is not video and getattr(video, "_dist_mix", None) is not None:      # (a deferred Mixup plan travels with the tensor the engine reads)
            conv._dist_mix = video._dist_mix
            del video._dist_mix
        self.engine.vit_prefetch(conv)
        self._pf_pair = (video, conv)

    def adopt_prefetched(self):
        self.engine.vit_adopt()
        self._cur_pair, self._pf_pair = self._pf_pair, None

    def forward_video(self, video, text, others=None):
        text_features, _, others = self.cache_text(text, others)
        pair = getattr(self, "_cur_pair", None)
        if pair is not None and pair[0] is video:
            video = pair[1]
        else:
            conv = video.contiguous().float()
            if conv is not video and getattr(video, "_dist_mix", None) is not None:
                conv._dist_mix = video._dist_mix
                del video._dist_mix
            video = conv
        # under torch.no_grad() (eval_epoch, perform_test) the engine keeps nothing for a backward pass (dist_set_inference)
        infer = not torch.is_grad_enabled()
        if infer != getattr(self, "_infer", False):
            self.engine.set_inference(infer)
            self._infer = infer
        logits, vid = _DistFunction.apply(self.engine, video, text_features, self.logit_scale, *self._dist_params)
        # `img_logits` (the frozen ViT's own frame embeddings) is read by no DiST configuration (zero_shot_test / prediction fusion are off): it is
        # computed on first access, from the feature slot this forward left (a gather, a LayerNorm, a GEMM and a copy less per train step)
        out = _OutputDict({"logits_per_image": logits, "logits_per_text": logits.t(), "vid_logits": vid[:, None, :]})
        out._lazy_img = (self, self.engine._feat_stamp)
        dict.__setitem__(out, "_dist_engine", self.engine)      # for models/utils/losses.py: the loss of these logits is dist_loss ...
        dict.__setitem__(out, "_dist_branch_stamp", self.engine._branch_stamp)      # ... as long as no other branch forward ran in between
        return out

    @torch.no_grad()
    def image_logits(self):
        """`img_logits` of the reference's output dictionary (clip.py:291-298,503,532): the frozen ViT's own embedding of every sampled
        frame, ln_post(cls token of the last block) @ visual.proj, [b*t, E] - NOT normalised (the reference only normalises it inside
        the zero-shot / prediction-fusion branch).  The cls rows come from the engine's saved mid_feat of the last block; LayerNorm and
        the projection are the HIP operators."""
        eng = self.engine
        d, L = eng.cfg.width, (eng.cfg.resolution // eng.cfg.patch) ** 2 + 1
        feat = eng.debug(f"feat.{eng.cfg.layers - 1}")                       # [b*t*L, d] of the current batch
        cls = feat.view(-1, L, d)[:, 0, :].contiguous()
        y = torch.empty_like(cls)
        ops.layernorm(cls, eng.view("visual.ln_post.weight"), eng.view("visual.ln_post.bias"), y=y)
        key = eng.visual._version
        if getattr(self, "_projT_key", None) != key:                          # frozen: transposed working copy made once per load
            self._projT = eng.view("visual.proj").t().contiguous().to(eng.dtype)
            self._projT_key = key
        out = torch.empty(cls.shape[0], eng.cfg.embed_dim, dtype=eng.dtype, device=cls.device)
        ops.gemm_nt(y, self._projT, cls.shape[0], eng.cfg.embed_dim, d, C_out=out)
        return out.float()

    def load_state_dict(self, state_dict, strict=True, first_init=False):
        out = super().load_state_dict(state_dict, strict=strict)
        self.engine.pack(3)
        return out


def _per_gpu_batch(cfg):
    n = max(1, int(getattr(cfg, "NUM_GPUS", 1)) * int(getattr(cfg, "NUM_SHARDS", 1)))
    tr = int(cfg.TRAIN.BATCH_SIZE) // n if hasattr(cfg, "TRAIN") and hasattr(cfg.TRAIN, "BATCH_SIZE") else 1
    te = int(cfg.TEST.BATCH_SIZE) // n if hasattr(cfg, "TEST") and hasattr(cfg.TEST, "BATCH_SIZE") else 1
    return max(1, tr, te)


def build_model(cfg, state_dict):
    """Shape inference as reference clip.py:564-600."""
    if "visual.proj" not in state_dict:
        raise L.DistError("only the ViT CLIP variants are on the DiST path (ModifiedResNet is dead code in the reference)")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    image_resolution = vision_patch_size * grid_size
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_heads = max(1, transformer_width // 64)
    transformer_layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP(cfg, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size,
                 context_length, vocab_size, transformer_width, transformer_heads, transformer_layers)
    for key in ("input_resolution", "context_length", "vocab_size"):
        state_dict.pop(key, None)
    sd = {k: (torch.as_tensor(v) if not torch.is_tensor(v) else v) for k, v in state_dict.items()}
    res = model.load_state_dict(sd, strict=False, first_init=True)
    # non-strict like the reference (the dist_net.* tensors are not in a CLIP checkpoint) - but a frozen-tower key that found no module
    # is a naming bug, never a choice: refuse instead of training on a randomly initialised tower
    frozen = ("visual.", "transformer.", "token_embedding.", "ln_final.")
    lost = [k for k in res.unexpected_keys if k.startswith(frozen) or k in ("positional_embedding", "text_projection", "logit_scale")]
    unset = [k for k in res.missing_keys if k.startswith(frozen) or k in ("positional_embedding", "text_projection")]
    if lost or unset:
        raise L.DistError(f"CLIP state-dict does not match the frozen towers: unexpected {lost[:6]}, missing {unset[:6]}")
    return model.eval()


def synthetic_state_dict(cfg):
    """Random-init CLIP-format state-dict for runs without a checkpoint (no network in this environment):
    VIDEO.BACKBONE.META_ARCH_NAME picks the geometry."""
    from ... import synth
    name = str(getattr(cfg.VIDEO.BACKBONE, "META_ARCH_NAME", "ViT-B-16"))
    T = cfg.DATA.NUM_INPUT_FRAMES
    if "L-14" in name or "L/14" in name:
        g = synth.Geometry(synth.geometry("l14_32+64f"), T=T).derived()
    elif "tiny" in name.lower():
        g = synth.Geometry(synth.geometry("tiny"), T=T).derived()
    else:
        g = synth.Geometry(synth.geometry("b16_8+16f"), T=T).derived()
    g["K"] = cfg.VIDEO.HEAD.NUM_CLASSES
    g["ada"] = cfg.VIDEO.BACKBONE.DIST.ADA_POOLING_LAYERS
    sd = {k: torch.from_numpy(v) for k, v in synth.state_dict(g).items() if k.startswith("visual.") or k == "logit_scale"}
    w = 64 if "tiny" in name.lower() else 512
    sd.update({"text_projection": torch.randn(w, g.E) * w ** -0.5, "positional_embedding": torch.randn(77, w) * 0.01,
               "token_embedding.weight": torch.randn(49408 if w == 512 else 64, w) * 0.02,
               "ln_final.weight": torch.ones(w), "ln_final.bias": torch.zeros(w)})
    for i in range(2 if w == 64 else 12):
        p = f"transformer.resblocks.{i}."
        sd.update({p + "attn.in_proj_weight": torch.randn(3 * w, w) * w ** -0.5, p + "attn.in_proj_bias": torch.zeros(3 * w),
                   p + "attn.out_proj.weight": torch.randn(w, w) * w ** -0.5, p + "attn.out_proj.bias": torch.zeros(w),
                   p + "ln_1.weight": torch.ones(w), p + "ln_1.bias": torch.zeros(w), p + "ln_2.weight": torch.ones(w), p + "ln_2.bias": torch.zeros(w),
                   p + "mlp.c_fc.weight": torch.randn(4 * w, w) * (2 * w) ** -0.5, p + "mlp.c_fc.bias": torch.zeros(4 * w),
                   p + "mlp.c_proj.weight": torch.randn(w, 4 * w) * w ** -0.5, p + "mlp.c_proj.bias": torch.zeros(w)})
    return sd


def load(cfg):
    """reference clip.py:614-629 (minus the OSS download): `.pyth` = pickled state-dict, otherwise a TorchScript archive."""
    path = cfg.VIDEO.BACKBONE.PRETRAIN_WEIGHT_PATH
    local = getattr(cfg.VIDEO.BACKBONE, "LOCAL_PRETRAIN_WEIGHT_PATH", None)
    if local and os.path.exists(local):
        path = local
    if path and os.path.exists(path):
        if str(path).endswith(".pyth"):
            state_dict = torch.load(path, map_location="cpu")
        else:
            state_dict = torch.jit.load(path, map_location="cpu").state_dict()
    elif getattr(cfg.VIDEO.BACKBONE, "SYNTHETIC_INIT", False):
        state_dict = synthetic_state_dict(cfg)
    else:
        raise FileNotFoundError(f"CLIP weights not found at {path}; set VIDEO.BACKBONE.SYNTHETIC_INIT true for random-init runs")
    return build_model(cfg, state_dict)
