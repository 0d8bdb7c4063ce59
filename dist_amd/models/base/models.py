"""`BaseVideoModel` = backbone + head (reference models/base/models.py:12-67)."""
import torch.nn as nn

from ...utils.registry import Registry
from .backbone import BACKBONE_REGISTRY
from .base_blocks import HEAD_REGISTRY

MODEL_REGISTRY = Registry("Model")


class BaseVideoModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.backbone = BACKBONE_REGISTRY.get(cfg.VIDEO.BACKBONE.META_ARCH)(cfg=cfg)
        self.head = HEAD_REGISTRY.get(cfg.VIDEO.HEAD.NAME)(cfg=cfg)

    def forward(self, x):
        return self.head(self.backbone(x))

    # software pipelining over batches (runs/train.py): frozen-ViT pass of the next batch beside this batch's step
    def prefetch(self, x_next):
        self.backbone.base_encoder.prefetch_video(x_next["video"])

    def adopt(self):
        self.backbone.base_encoder.adopt_prefetched()
