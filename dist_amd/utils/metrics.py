"""top-k helpers (reference utils/metrics.py:100-133)."""
import torch


def topks_correct(preds, labels, ks):
    assert preds.size(0) == labels.size(0)
    _, top = torch.topk(preds, max(ks), dim=1, largest=True, sorted=True)
    rep = labels.view(1, -1).expand_as(top.t())
    correct = top.t().eq(rep)
    return [correct[:k].reshape(-1).float().sum() for k in ks]


def topk_errors(preds, labels, ks):
    return [(1.0 - x / preds.size(0)) * 100.0 for x in topks_correct(preds, labels, ks)]


def topk_accuracies(preds, labels, ks):
    return [(x / preds.size(0)) * 100.0 for x in topks_correct(preds, labels, ks)]
