"""Losses of the DiST path (reference models/utils/losses.py:20-31,52-119)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _EngineSoftTargetCE(torch.autograd.Function):
    """Soft-target CE of the engine's own logits through dist_loss (logits_loss_kernel: loss and dlogits in one launch) - the call
    bench.py times, so the shipped train loop and the benchmarked step are one kernel sequence.  `preds` only carries the graph: the
    kernel reads the video embedding the branch forward left in the engine."""

    @staticmethod
    def forward(ctx, preds, target, engine):
        loss, dlogits = engine.loss(target.contiguous().float())
        ctx.save_for_backward(dlogits)
        ctx.shape = preds.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return (g * dlogits).view(ctx.shape), None, None


def _engine_of(out, preds):
    """The engine whose LAST branch forward produced `preds`, or None.  dist_loss reads the video embedding that forward left in the engine, so the
    HIP path is only taken when (a) the output dictionary's stamp is still the engine's branch-forward counter (no other forward - a second view,
    an evaluation pass - ran in between) and (b) `preds` is computed from this dictionary's `logits_per_image` (its autograd graph reaches that
    tensor's node within a few steps: the head's mean / activation).  Anything else takes the torch expression on `preds` itself."""
    if not isinstance(out, dict) or not torch.is_grad_enabled():
        return None
    engine, stamp = out.get("_dist_engine"), out.get("_dist_branch_stamp")
    if engine is None or stamp is None or stamp != getattr(engine, "_branch_stamp", None):
        return None
    src = out["logits_per_image"]
    if preds is src:
        return engine
    root, todo, seen = src.grad_fn, [preds.grad_fn], 0
    while todo and seen < 16 and root is not None:
        fn = todo.pop()
        if fn is None:
            continue
        if fn is root:
            return engine
        seen += 1
        todo.extend(f for f, _ in fn.next_functions)
    return None


class SoftTargetCrossEntropy(nn.Module):
    """mean_b sum_k -target * log_softmax(x) (reference losses.py:20-31, timm's SoftTargetCrossEntropy).  `engine`: the logits are the ones
    that engine's branch forward just produced -> dist_loss (HIP); tensors of any other origin (unit tests on the host) take the
    torch expression."""

    def forward(self, x, target, engine=None):
        if engine is not None and x.is_cuda and x.dim() == 2 and x.shape[0] == engine.b:
            return _EngineSoftTargetCE.apply(x, target, engine)
        return torch.sum(-target * F.log_softmax(x, dim=-1), dim=-1).mean()


def calculate_loss(cfg, preds, logits, labels, cur_epoch):
    """reference losses.py:52-119 for the supervised DiST case: soft-target CE when mixup / label smoothing
    produced soft labels, plain CE otherwise.  Returns (loss, {name: loss}, weight)."""
    target = labels["supervised"] if isinstance(labels, dict) else labels
    if target.dtype in (torch.float32, torch.float16, torch.bfloat16) and target.dim() == 2:
        # `logits` is the backbone's output dictionary (models.py:forward -> head returns (preds, x)); CLIP.forward_video leaves its engine in it
        engine = _engine_of(logits, preds)
        loss = SoftTargetCrossEntropy()(preds, target, engine)
        name = "soft_target"
    else:
        # hard labels (no mixup / label smoothing): the same HIP loss on their one-hot form when the logits are the engine's
        engine = _engine_of(logits, preds)
        if engine is not None and preds.is_cuda and preds.dim() == 2 and preds.shape[0] == engine.b and target.dim() == 1:
            loss = SoftTargetCrossEntropy()(preds, F.one_hot(target.long(), preds.shape[1]).float(), engine)
        else:
            loss = F.cross_entropy(preds, target)
        name = "cross_entropy"
    return loss, {name: loss}, None
