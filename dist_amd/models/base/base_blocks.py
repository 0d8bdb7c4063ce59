"""Head registry + `ClipVideoTextIdentity` (reference models/base/base_blocks.py:541-585): parameter-free,
mean over dim 1, softmax only at eval (dist_op_softmax_rows on device logits)."""
import torch.nn as nn

from ... import ops
from ...utils.registry import Registry

HEAD_REGISTRY = Registry("Head")
STEM_REGISTRY = Registry("Stem")
BRANCH_REGISTRY = Registry("Branch")


@HEAD_REGISTRY.register()
class ClipVideoTextIdentity(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        act = getattr(cfg.VIDEO.HEAD, "ACTIVATION", "softmax")
        if act == "softmax":
            self.activation = nn.Softmax(dim=-1)
        elif act == "sigmoid":
            self.activation = nn.Sigmoid()
        elif act == "identity":
            self.activation = nn.Identity()
        else:
            raise NotImplementedError(f"{act} is not supported as an activation function.")

    def forward(self, x):
        out = x["logits_per_image"].mean(dim=1) if isinstance(x, dict) else x.mean(dim=1)
        if not self.training:
            if isinstance(self.activation, nn.Softmax) and out.is_cuda and out.dim() == 2:
                out = ops.softmax_rows(out).to(out.dtype)        # computed in fp32, returned in the input dtype like nn.Softmax(dim=-1)
            else:
                out = self.activation(out)
        return out, x
