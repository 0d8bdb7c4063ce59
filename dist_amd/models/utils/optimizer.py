"""Optimizer construction for DiST (reference models/utils/optimizer.py:23-91,138-214).

`construct_DiST_optimizer` builds the five parameter groups the released code evidently intends (it is
broken as shipped: misplaced brackets raise TypeError): cls_token / positional_embedding
without weight decay, ada-pooling weights / biases, other weights / biases; every group uses
lr * NEW_NET_LRMULT.  `construct_optimizer` returns `FusedDistAdamW`, a torch.optim.Optimizer whose
step() is ONE multi-tensor HIP kernel over the flat parameter buffer (dist_op_adamw) followed by the
re-pack of the bf16 working weights."""
import torch

from . import lr_policy


def construct_DiST_optimizer(model, cfg):
    groups = {k: [] for k in ("no_wd", "ada_w", "ada_b", "w", "b")}
    for name, p in model.named_parameters():
        if not p.requires_grad or "dist_net" not in name:
            continue
        one_d = "bias" in name or p.dim() == 1
        if name.endswith("cls_token") or name.endswith("positional_embedding"):
            groups["no_wd"].append(p)
        elif "adapooling_nets" in name:
            groups["ada_b" if one_d else "ada_w"].append(p)
        else:
            groups["b" if one_d else "w"].append(p)
    wd, mult = cfg.OPTIMIZER.NEW_NET_WEIGHT_DECAY, cfg.OPTIMIZER.NEW_NET_LRMULT
    out = []
    for key, decay in (("no_wd", 0.0), ("ada_w", wd), ("ada_b", 0.0), ("w", wd), ("b", 0.0)):
        if groups[key]:
            out.append({"params": groups[key], "weight_decay": decay, "lr_mult": mult})
    return out


class FusedDistAdamW(torch.optim.Optimizer):
    """AdamW (betas (0.9, 0.999) hard-coded like reference optimizer.py:67-73) over the engine's flat buffers."""

    def __init__(self, params, engine, lr, weight_decay, grad_sync=None):
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay, lr_mult=1.0))
        self.engine = engine
        self.grad_sync = grad_sync

    @torch.no_grad()
    def step(self, closure=None):
        g0 = self.param_groups[0]
        lr = g0["lr"] / max(g0.get("lr_mult", 1.0), 1e-30)          # set_lr() stored lr * lr_mult
        wd = max(g["weight_decay"] for g in self.param_groups)
        scale = self.grad_sync.grad_scale if self.grad_sync is not None else 1.0
        self.engine.adamw_step(lr, wd, lr_mult=g0.get("lr_mult", 1.0), grad_scale=scale)

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=True)          # the engine zeroes its flat gradient buffer at each backward


def construct_optimizer(model, cfg):
    if cfg.OPTIMIZER.OPTIM_METHOD != "adamw":
        raise NotImplementedError("the DiST recipe uses adamw")
    params = construct_DiST_optimizer(model, cfg)
    engine = model.backbone.base_encoder.engine
    return FusedDistAdamW(params, engine, cfg.OPTIMIZER.BASE_LR, cfg.OPTIMIZER.NEW_NET_WEIGHT_DECAY, getattr(model, "grad_sync", None))


def get_epoch_lr(cur_epoch, cfg):
    return lr_policy.get_lr_at_epoch(cfg, cur_epoch)


def set_lr(optimizer, new_lr):
    """reference optimizer.py:200-214"""
    for g in optimizer.param_groups:
        if g.get("lr_reduce"):
            g["lr"] = new_lr / 10
        elif "lr_mult" in g:
            g["lr"] = new_lr * g["lr_mult"]
        else:
            g["lr"] = new_lr
