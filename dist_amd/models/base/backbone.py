"""Backbone registry + `ClipVisionTextTransformer` (reference models/base/backbone.py:218-256)."""
import torch.nn as nn

from ...utils.registry import Registry

BACKBONE_REGISTRY = Registry("Backbone")


@BACKBONE_REGISTRY.register()
class ClipVisionTextTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        from . import clip
        self.base_encoder = clip.load(cfg)

    def forward(self, x):
        """x: {"video": [b,3,T,H,W], "texts": [K,77]} -> the reference's output dict with
        `logits_per_image` reshaped to [b,1,K] (backbone.py:238-241).  The reference permutes the clip to
        [b*T,3,H,W] first (a 308 MB copy at b=32); the kernels read [b,3,T,H,W] directly."""
        if "texts" not in x:
            raise ValueError("DiST needs x['texts'] (the text-less branch of the reference raises TypeError, SURVEY.md §0)")
        video = x["video"]
        b = video.shape[0]
        out = self.base_encoder.forward_video(video, x["texts"])
        out["logits_per_image"] = out["logits_per_image"].reshape(b, 1, -1)
        return out

    def get_num_layers(self):
        return self.base_encoder.engine.cfg.layers, 0
