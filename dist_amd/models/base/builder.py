"""`build_model(cfg, gpu_id=None) -> (model, model_ema)` (reference models/base/builder.py:19-74).

Data parallelism differs by design: the reference wraps the model in DistributedDataParallel with
find_unused_parameters=True and reduces all 168.6 M parameters; here the model stays unwrapped and
`DistGradSync` reduces the flat dist_net gradient buffer only (76 MB for ViT-B/16) over RCCL."""
import torch

from ...utils import distributed as du
from .models import MODEL_REGISTRY, BaseVideoModel


class DistGradSync(du.GradReducer):
    """Data-parallel gradient exchange of a built model: the bucketed all-reduces of the flat dist_net gradient
    buffer are started from the engine's gradient-ready hook DURING `loss.backward()`; `.reduce()` after backward
    only flushes the last bucket and makes the compute stream wait for the side stream."""

    def __init__(self, model):
        super().__init__(model.backbone.base_encoder.engine, du.get_world_size())

    def reduce(self):
        if self.world <= 1:
            return
        if not self.overlap:
            torch.distributed.all_reduce(self.eng.grads)
            return
        if self._pending is not None:
            self._send(*self._pending)
            self._pending = None
        done = torch.cuda.Event()
        done.record(self.comm)
        torch.cuda.current_stream().wait_event(done)


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
