"""`build_model(cfg, gpu_id=None) -> (model, model_ema)` (reference models/base/builder.py:19-74).

Data parallelism differs by design: the reference wraps the model in DistributedDataParallel with
find_unused_parameters=True and reduces all 168.6 M parameters; here the model stays unwrapped and
`DistGradSync` reduces the flat dist_net gradient buffer only (76 MB for ViT-B/16) over RCCL."""
import torch

from ...utils import distributed as du
from .models import MODEL_REGISTRY, BaseVideoModel


class DistGradSync:
    """Call `.reduce()` after backward: one sum all-reduce over the flat dist_net gradients; the
    1/world average is applied by the fused AdamW (grad_scale)."""

    def __init__(self, model):
        self.engine = model.backbone.base_encoder.engine
        self.world = du.get_world_size()

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def reduce(self):
        if self.world > 1:
            torch.distributed.all_reduce(self.engine.grads)


def build_model(cfg, gpu_id=None):
    if MODEL_REGISTRY.get(cfg.MODEL.NAME) is None:
        model = BaseVideoModel(cfg)          # MODEL.NAME 'clip' is unregistered in the reference too (builder.py:30-32)
    else:
        model = MODEL_REGISTRY.get(cfg.MODEL.NAME)(cfg)
    if not torch.cuda.is_available():
        raise RuntimeError("dist_amd needs a GPU: there is no CPU implementation of the hot path")
    cur = torch.cuda.current_device() if gpu_id is None else gpu_id
    model = model.cuda(device=cur) if any(p.device.type != "cuda" for p in model.parameters()) else model
    model.grad_sync = DistGradSync(model)
    return model, None
