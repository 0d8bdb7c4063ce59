"""Procedural (counter-hash) synthetic weights and inputs.

Used by bench.py / smoke() (random-init weights + synthetic frames, there is no
network for checkpoints or datasets) and by the tests and the oracle.  Weights
and inputs are never stored: every tensor is a pure function of
(name, shape, seed), so the container that generates golden OUTPUTS from the
reference and the GPU box that checks the HIP path regenerate bit-identical
inputs from this file alone.

Integer hash (murmur3 finaliser on a 64-bit counter) -> 24-bit mantissa ->
uniform in [-1, 1) -> scaled.  Only integer ops and one exact float multiply, so
the values are bit-identical on any IEEE machine.
"""
import zlib

import numpy as np


def _mix(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xFF51AFD7ED558CCD)
    x ^= x >> np.uint64(33)
    x *= np.uint64(0xC4CEB9FE1A85EC53)
    x ^= x >> np.uint64(33)
    return x


def uniform(name, shape, seed=0):
    """float32 array, i.i.d.-looking uniform in [-1, 1), a function of (name, seed, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF) | (np.uint64(seed & 0xFFFFFFFF) << np.uint64(32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h = _mix(idx * np.uint64(0x9E3779B97F4A7C15) + _mix(np.array([key], dtype=np.uint64))[0])
    m = (h >> np.uint64(40)).astype(np.float32)          # 24 bits, exact in fp32
    u = m * np.float32(2.0 ** -23) - np.float32(1.0)     # exact
    return u.reshape(shape)


SQRT3 = 1.7320508075688772


def gaussian_like(name, shape, std, seed=0):
    """Uniform with the requested standard deviation (uniform[-1,1) has var 1/3)."""
    return (uniform(name, shape, seed) * np.float32(std * SQRT3)).astype(np.float32)


# --------------------------------------------------------------------------------------
# model geometry + state-dict (reference key names, SURVEY.md §8(b) "State-dict layout")
# --------------------------------------------------------------------------------------

class Geometry(dict):
    """Plain attribute dict describing one DiST configuration."""
    __getattr__ = dict.__getitem__

    def derived(self):
        g = self
        g["grid"] = g.res // g.patch
        g["N"] = g.grid * g.grid
        g["L"] = g.N + 1
        g["t"] = g.T // g.alpha
        g["heads"] = g.d // 64
        g["C4"] = int(g.Ci * g.int_t_ratio)
        # hidden widths of TemporalNet / IntegrationNetwork.ffn: DIST.TEMPORAL_CONV_MLP_RATIO, DIST.INTEGRATION_MLP_RATIO (1 in every released yaml)
        g["tn_ratio"] = g.get("tn_ratio", 1); g["ffn_ratio"] = g.get("ffn_ratio", 1)
        g["Ch"] = int(g.Ct * g["tn_ratio"]); g["Cf"] = int(g.Ci * g["ffn_ratio"])
        g["iheads"] = g.Ci // 64
        # VIDEO.BACKBONE.DIST.SELECTED_LAYERS (reference dist.py:170-190, 226): the ViT blocks whose outputs feed the branch, one DiST layer each
        g["selected"] = tuple(g.get("selected") or range(g.layers))
        g["nsel"] = len(g["selected"])
        return g


def geometry(name):
    base = dict(alpha=2, tk=3, tpatch=5, int_t_ratio=0.25, ada=2)
    if name == "tiny":
        # SURVEY §8(c) fixture (1): small enough that the reference runs in ~0.1 s
        g = Geometry(base, name="tiny", d=128, layers=2, patch=16, res=64, Ci=128, Ct=32, T=8, K=10, E=64)
    elif name == "tiny3":
        # odd grid (3x3 patches -> L=10): ragged rows for every tile shape
        g = Geometry(base, name="tiny3", d=128, layers=3, patch=16, res=48, Ci=128, Ct=32, T=4, K=7, E=64)
    elif name == "tiny3_sel":
        # a real SUBSET of the ViT blocks: blocks 0 and 2 of three feed two DiST layers (every released yaml selects all; the reference takes any)
        g = Geometry(base, name="tiny3_sel", d=128, layers=3, patch=16, res=48, Ci=128, Ct=32, T=4, K=7, E=64, selected=(0, 2))
    elif name == "tiny_ratio":
        # MLP ratios other than 1 (TEMPORAL_CONV_MLP_RATIO 2, INTEGRATION_MLP_RATIO 0.5): the reference takes any (dist.py:20-25, 51-58)
        g = Geometry(base, name="tiny_ratio", d=128, layers=2, patch=16, res=48, Ci=128, Ct=32, T=4, K=7, E=64, tn_ratio=2, ffn_ratio=0.5)
    elif name == "b16_8+16f":
        g = Geometry(base, name="b16_8+16f", d=768, layers=12, patch=16, res=224, Ci=384, Ct=96, T=16, K=174, E=512)
    elif name == "b16_16+32f":
        g = Geometry(base, name="b16_16+32f", d=768, layers=12, patch=16, res=224, Ci=384, Ct=96, T=32, K=174, E=512)
    elif name == "l14_tiny_t":
        # ViT-L/14 geometry (S_PATCH_SIZE=14) at T=8 to keep the CPU run short
        g = Geometry(base, name="l14_tiny_t", d=1024, layers=24, patch=14, res=224, Ci=384, Ct=96, T=8, K=400, E=768, ada=4)
    elif name == "l14_32+64f":
        g = Geometry(base, name="l14_32+64f", d=1024, layers=24, patch=14, res=224, Ci=384, Ct=96, T=64, K=400, E=768, ada=4)
    else:
        raise KeyError(name)
    return g.derived()


def dist_net_shapes(g):
    """Every dist_net.* tensor (reference models/module_zoo/branches/dist.py:165-202)."""
    s = {}
    Ct, Ci, d, t, C4, Ch, Cf = g.Ct, g.Ci, g.d, g.t, g.C4, g.Ch, g.Cf
    s["dist_net.temporal_stem.weight"] = (Ct, 3, g.tpatch, g.patch, g.patch)
    s["dist_net.temporal_stem.bias"] = (Ct,)
    for i in range(len(g.selected)):
        s[f"dist_net.input_linears.{i}.weight"] = (Ci, d)
        s[f"dist_net.input_linears.{i}.bias"] = (Ci,)
        s[f"dist_net.integration2temporal_nets.{i}.linear_fuse.weight"] = (Ct, Ci)
        s[f"dist_net.integration2temporal_nets.{i}.linear_fuse.bias"] = (Ct,)
        s[f"dist_net.temporal2integration_nets.{i}.cls_token"] = (1, 1, t, Ci)
        s[f"dist_net.temporal2integration_nets.{i}.linear_fuse.weight"] = (Ci, Ct, g.alpha, 1, 1)
        s[f"dist_net.temporal2integration_nets.{i}.linear_fuse.bias"] = (Ci,)
        p = f"dist_net.temporal_nets.{i}."
        s[p + "temporal_net.c_fc1.weight"] = (Ch, Ct, g.tk, 1, 1)
        s[p + "temporal_net.c_fc1.bias"] = (Ch,)
        s[p + "temporal_net.c_fc2.weight"] = (Ct, Ch, 1, 3, 3)
        s[p + "temporal_net.c_fc2.bias"] = (Ct,)
        s[p + "ln.weight"] = (Ct,)
        s[p + "ln.bias"] = (Ct,)
        p = f"dist_net.integration_nets.{i}."
        s[p + "ffn.c_fc.weight"] = (Cf, Ci)
        s[p + "ffn.c_fc.bias"] = (Cf,)
        s[p + "ffn.c_proj.weight"] = (Ci, Cf)
        s[p + "ffn.c_proj.bias"] = (Ci,)
        s[p + "temporal_ffn.c_fc1.weight"] = (C4, Ci, 1, 1, 1)
        s[p + "temporal_ffn.c_fc1.bias"] = (C4,)
        s[p + "temporal_ffn.c_fc2.weight"] = (C4, C4, g.tk, 1, 1)
        s[p + "temporal_ffn.c_fc2.bias"] = (C4,)
        s[p + "temporal_ffn.c_proj.weight"] = (Ci, C4, 1, 1, 1)
        s[p + "temporal_ffn.c_proj.bias"] = (Ci,)
        s[p + "ln.weight"] = (Ci,)
        s[p + "ln.bias"] = (Ci,)
        s[p + "ln_temporal.weight"] = (Ci,)
        s[p + "ln_temporal.bias"] = (Ci,)
    for a in range(g.ada):
        p = f"dist_net.adapooling_nets.{a}."
        s[p + "positional_embedding"] = (1, t, Ci)
        for tr in ("temporal_transformer", "spatial_transformer"):
            s[p + tr + ".attn.in_proj_weight"] = (3 * Ci, Ci)
            s[p + tr + ".attn.in_proj_bias"] = (3 * Ci,)
            s[p + tr + ".attn.out_proj.weight"] = (Ci, Ci)
            s[p + tr + ".attn.out_proj.bias"] = (Ci,)
            s[p + tr + ".ln_1.weight"] = (Ci,)
            s[p + tr + ".ln_1.bias"] = (Ci,)
        for m in ("output_map_cls_token", "output_map_spatial_cls_token"):
            s[p + m + ".c_fc.weight"] = (4 * Ci, Ci)
            s[p + m + ".c_fc.bias"] = (4 * Ci,)
            s[p + m + ".c_proj.weight"] = (Ci, 4 * Ci)
            s[p + m + ".c_proj.bias"] = (Ci,)
        for m in ("ln_out_temp_cls_token", "ln_out_spat_cls_token"):
            s[p + m + ".weight"] = (Ci,)
            s[p + m + ".bias"] = (Ci,)
    s["dist_net.proj_spatial_cls_token.weight"] = (Ci, d)
    s["dist_net.proj_spatial_cls_token.bias"] = (Ci,)
    s["dist_net.ln_post.weight"] = (Ci,)
    s["dist_net.ln_post.bias"] = (Ci,)
    s["dist_net.proj"] = (Ci, g.E)
    s["dist_net.aggregated_cls_token"] = (1, 1, Ci)
    s["dist_net.aggregated_spatial_cls_token"] = (1, 1, Ci)
    return s


def visual_shapes(g):
    """visual.* tensors of the OpenAI-CLIP ViT (reference models/base/clip.py:218-247)."""
    d = g.d
    s = {
        "visual.conv1.weight": (d, 3, g.patch, g.patch),
        "visual.class_embedding": (d,),
        "visual.positional_embedding": (g.L, d),
        "visual.ln_pre.weight": (d,), "visual.ln_pre.bias": (d,),
        "visual.ln_post.weight": (d,), "visual.ln_post.bias": (d,),
        "visual.proj": (d, g.E),
    }
    for i in range(g.layers):
        p = f"visual.transformer.resblocks.{i}."
        s[p + "attn.in_proj_weight"] = (3 * d, d)
        s[p + "attn.in_proj_bias"] = (3 * d,)
        s[p + "attn.out_proj.weight"] = (d, d)
        s[p + "attn.out_proj.bias"] = (d,)
        s[p + "ln_1.weight"] = (d,); s[p + "ln_1.bias"] = (d,)
        s[p + "ln_2.weight"] = (d,); s[p + "ln_2.bias"] = (d,)
        s[p + "mlp.c_fc.weight"] = (4 * d, d); s[p + "mlp.c_fc.bias"] = (4 * d,)
        s[p + "mlp.c_proj.weight"] = (d, 4 * d); s[p + "mlp.c_proj.bias"] = (d,)
    return s


def _std_for(name, shape):
    leaf = name.split(".")[-1]
    if name.endswith(("ln.weight", "ln_1.weight", "ln_2.weight", "ln_pre.weight", "ln_post.weight",
                      "ln_temporal.weight", "ln_out_temp_cls_token.weight", "ln_out_spat_cls_token.weight")):
        return None  # handled as 1 + 0.1 u
    if len(shape) == 1 or leaf in ("bias", "in_proj_bias"):
        return 0.05
    if leaf in ("cls_token", "positional_embedding", "aggregated_cls_token",
                "aggregated_spatial_cls_token", "class_embedding"):
        return 0.1
    if leaf == "proj":                       # [in, out] matrices
        return 0.8 / np.sqrt(shape[0])
    fan_in = int(np.prod(shape[1:]))
    return 0.8 / np.sqrt(fan_in)


def make_tensor(name, shape, seed=0):
    std = _std_for(name, shape)
    if std is None:
        return (np.float32(1.0) + np.float32(0.1) * uniform(name, shape, seed)).astype(np.float32)
    return gaussian_like(name, shape, std, seed)


def state_dict(g, seed=0):
    """name -> float32 ndarray for visual.* + dist_net.* + logit_scale."""
    sd = {}
    shapes = {}
    shapes.update(visual_shapes(g))
    shapes.update(dist_net_shapes(g))
    for k, shp in shapes.items():
        sd[k] = make_tensor(k, shp, seed)
    sd["logit_scale"] = np.array(np.log(1.0 / 0.07), dtype=np.float32)
    return sd


def text_tower_state_dict(width=64, layers=2, vocab=96, context=77, embed=64, seed=5):
    """Procedural CLIP text tower (reference models/base/clip.py:359-371,420-435 key names) for the encode_text fixture:
    small enough that the reference encodes a label set in milliseconds on the CPU."""
    sd = {"token_embedding.weight": gaussian_like("token_embedding.weight", (vocab, width), 0.3, seed),
          "positional_embedding": gaussian_like("text.positional_embedding", (context, width), 0.1, seed),
          "ln_final.weight": (np.float32(1.0) + np.float32(0.1) * uniform("ln_final.weight", (width,), seed)).astype(np.float32),
          "ln_final.bias": gaussian_like("ln_final.bias", (width,), 0.05, seed),
          "text_projection": gaussian_like("text_projection", (width, embed), 0.8 / np.sqrt(width), seed)}
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        for name, shape in (("attn.in_proj_weight", (3 * width, width)), ("attn.out_proj.weight", (width, width)),
                            ("mlp.c_fc.weight", (4 * width, width)), ("mlp.c_proj.weight", (width, 4 * width))):
            sd[p + name] = gaussian_like("text." + p + name, shape, 0.8 / np.sqrt(shape[1]), seed)
        for name, n in (("attn.in_proj_bias", 3 * width), ("attn.out_proj.bias", width), ("mlp.c_fc.bias", 4 * width), ("mlp.c_proj.bias", width),
                        ("ln_1.bias", width), ("ln_2.bias", width)):
            sd[p + name] = gaussian_like("text." + p + name, (n,), 0.05, seed)
        for name in ("ln_1.weight", "ln_2.weight"):
            sd[p + name] = (np.float32(1.0) + np.float32(0.1) * uniform("text." + p + name, (width,), seed)).astype(np.float32)
    return sd


def label_tokens(K, vocab=96, context=77, seed=6):
    """[K, context] int64 token rows shaped like CLIP's tokenizer output: start token, 2-9 word tokens, the end token = the LARGEST id
    of the row (encode_text takes the feature at argmax, reference clip.py:430), zero padding."""
    u = (uniform("label_tokens", (K, context), seed).astype(np.float64) + 1.0) * 0.5
    t = np.zeros((K, context), dtype=np.int64)
    for r in range(K):
        n = 2 + int(u[r, 0] * 8)
        t[r, 0] = vocab - 2
        t[r, 1:1 + n] = 1 + (u[r, 1:1 + n] * (vocab - 4)).astype(np.int64)
        t[r, 1 + n] = vocab - 1
    return t


def video(g, b, seed=1):
    """Synthetic mean/std-normalised frames [b,3,T,H,W] ~ unit variance."""
    return gaussian_like("video", (b, 3, g.T, g.res, g.res), 1.0, seed)


def text_features(g, seed=2):
    """[K,E] text embeddings (the frozen, cached text tower's output; an input fixture)."""
    tf = gaussian_like("text_features", (g.K, g.E), 1.0, seed).astype(np.float64)
    tf /= np.linalg.norm(tf, axis=1, keepdims=True)
    return tf.astype(np.float32)


def soft_target(g, b, seed=3, smoothing=0.1, lam=0.7):
    """Mixup(label-smoothed one-hot) target as dataset/utils/mixup.py:18-23 builds it."""
    u = uniform("labels", (b,), seed)
    y = np.minimum(((u + 1.0) * 0.5 * g.K).astype(np.int64), g.K - 1)
    off, on = smoothing / g.K, 1.0 - smoothing + smoothing / g.K
    one = np.full((b, g.K), off, dtype=np.float32)
    one[np.arange(b), y] = on
    flip = one[::-1]
    return (one * np.float32(lam) + flip * np.float32(1.0 - lam)).astype(np.float32), y
