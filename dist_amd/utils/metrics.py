"""top-k helpers with the reference's names and return types (utils/metrics.py:100-159), computed by dist_op_topk_correct on the
predictions where they are (device tensors): no topk / sort / host copy per iteration.  The predictions must be on the GPU -
there is no CPU path in the product (the numpy restatement used by the tests lives in oracle/meters_oracle.py); labels may
arrive as host tensors the way the reference's loaders yield them (runs/train.py:165, runs/test.py:133) and are moved to the
predictions' device."""
from .. import ops


def topks_correct(preds, labels, ks):
    """list of 0-dim fp32 tensors: number of rows whose label is among the top-k scores, one per k in `ks`."""
    if not preds.is_cuda:
        raise RuntimeError("dist_amd.utils.metrics needs the predictions on the GPU (the HIP library computes the ranks)")
    labels = labels.to(preds.device, non_blocking=True)
    assert preds.size(0) == labels.size(0), "Batch dim of predictions and labels must match"
    out = ops.topk_correct(preds, labels, tuple(ks))
    return [out[i] for i in range(len(ks))]


def topk_errors(preds, labels, ks):
    return [(1.0 - x / preds.size(0)) * 100.0 for x in topks_correct(preds, labels, ks)]


def topk_accuracies(preds, labels, ks):
    return [(x / preds.size(0)) * 100.0 for x in topks_correct(preds, labels, ks)]
