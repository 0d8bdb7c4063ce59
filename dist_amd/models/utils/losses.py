"""Losses of the DiST path (reference models/utils/losses.py:20-31,52-119)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class SoftTargetCrossEntropy(nn.Module):
    """mean_b sum_k -target * log_softmax(x) (reference losses.py:20-31, timm's SoftTargetCrossEntropy)."""

    def forward(self, x, target):
        return torch.sum(-target * F.log_softmax(x, dim=-1), dim=-1).mean()


def calculate_loss(cfg, preds, logits, labels, cur_epoch):
    """reference losses.py:52-119 for the supervised DiST case: soft-target CE when mixup / label smoothing
    produced soft labels, plain CE otherwise.  Returns (loss, {name: loss}, weight)."""
    target = labels["supervised"] if isinstance(labels, dict) else labels
    if target.dtype in (torch.float32, torch.float16, torch.bfloat16) and target.dim() == 2:
        loss = SoftTargetCrossEntropy()(preds, target)
        name = "soft_target"
    else:
        loss = F.cross_entropy(preds, target)
        name = "cross_entropy"
    return loss, {name: loss}, None
