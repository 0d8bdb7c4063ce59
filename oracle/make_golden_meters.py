"""Golden vectors for the evaluation side: runs the REFERENCE's own `utils.metrics.topks_correct` / `topk_errors` and
`utils.meters.TestMeter` (imported from /root/reference; the only stand-in is `simplejson` -> json, which the imported code path
never calls) plus torch's nn.Softmax on seeded inputs, and stores inputs-by-recipe and OUTPUTS in tests/golden/meters.npz.
Runs only in the build container; nothing of the reference ships."""
import json
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "meters.npz")


def scores(seed, n, K, kind="softmax"):
    """continuous fp32 scores without ties, from a counter-based generator"""
    g = np.random.Generator(np.random.PCG64(7000 + seed))
    z = (g.standard_normal((n, K)) * 3.0).astype(np.float32)
    return z if kind == "logits" else torch.softmax(torch.from_numpy(z), dim=-1).numpy()


def int_draws(seed, hi, n):
    return np.random.Generator(np.random.PCG64(9000 + seed)).integers(0, hi, size=n).astype(np.int64)


# (seed, videos, views per video, classes, batch, method, fraction of clips delivered)
METER_CASES = [(0, 6, 3, 10, 4, "sum", 1.0), (1, 6, 3, 10, 4, "max", 1.0), (2, 16, 30, 174, 32, "sum", 1.0), (3, 8, 12, 400, 7, "sum", 1.0),
               (4, 9, 4, 174, 5, "sum", 0.7), (5, 5, 2, 3, 10, "max", 1.0)]
TOPK_CASES = [(0, 32, 174, (1, 5)), (1, 7, 400, (1, 5)), (2, 256, 10, (1, 5)), (3, 5, 6, (1, 3, 6)), (4, 1, 174, (1, 5))]


def meter_inputs(seed, V, views, K, bs, frac):
    order = np.random.Generator(np.random.PCG64(8000 + seed)).permutation(V * views)
    order = order[: max(1, int(round(len(order) * frac)))]
    vid_label = int_draws(seed, K, V)
    if seed == 0:
        vid_label[0] = 0                                         # label 0: the reference's consistency assert never looks at it
    p = scores(seed, len(order), K)
    batches = [(p[i:i + bs], vid_label[order[i:i + bs] // views], order[i:i + bs]) for i in range(0, len(order), bs)]
    return batches


def main():
    sys.path.insert(0, "/root/reference")
    sj = types.ModuleType("simplejson"); sj.dumps = json.dumps; sj.loads = json.loads
    sys.modules["simplejson"] = sj
    import utils.metrics as rmetrics
    from utils.meters import TestMeter
    out = {"n_meter": np.int64(len(METER_CASES)), "n_topk": np.int64(len(TOPK_CASES))}
    for ci, (seed, n, K, ks) in enumerate(TOPK_CASES):
        p, lab = scores(100 + seed, n, K), int_draws(100 + seed, K, n)
        tc = rmetrics.topks_correct(torch.from_numpy(p), torch.from_numpy(lab), ks)
        te = rmetrics.topk_errors(torch.from_numpy(p), torch.from_numpy(lab), ks)
        out[f"t{ci}_meta"] = np.array([seed, n, K], np.int64)
        out[f"t{ci}_ks"] = np.array(ks, np.int64)
        out[f"t{ci}_correct"] = np.array([float(x) for x in tc], np.float32)
        out[f"t{ci}_errors"] = np.array([float(x) for x in te], np.float32)
    for ci, (seed, V, views, K, bs, method, frac) in enumerate(METER_CASES):
        cfg = types.SimpleNamespace(LOG_PERIOD=10 ** 9)
        m = TestMeter(cfg, V, views, K, 1, method)
        for p, lab, ids in meter_inputs(seed, V, views, K, bs, frac):
            m.update_stats(torch.from_numpy(p), torch.from_numpy(lab), torch.from_numpy(ids))
        tc = rmetrics.topks_correct(m.video_preds, m.video_labels, (1, 5) if K >= 5 else (1, 2))   # what finalize_metrics computes (:157-163)
        out[f"m{ci}_meta"] = np.array([seed, V, views, K, bs, method == "max", int(frac * 100)], np.int64)
        out[f"m{ci}_video_preds"] = m.video_preds.numpy().copy()
        out[f"m{ci}_video_labels"] = m.video_labels.numpy().copy()
        out[f"m{ci}_clip_count"] = m.clip_count.numpy().copy()
        out[f"m{ci}_acc"] = np.array([float(x / m.video_preds.size(0) * 100.0) for x in tc], np.float32)
    z = scores(55, 9, 174, "logits")
    out["s_softmax"] = torch.nn.Softmax(dim=-1)(torch.from_numpy(z)).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
