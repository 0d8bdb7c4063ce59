"""TEST INFRASTRUCTURE ONLY (never imported by the product path): numpy restatement of the evaluation-side arithmetic.

  * softmax_rows     - nn.Softmax(dim=-1) of the eval head (reference models/base/base_blocks.py:573-585), fp32
  * topks_correct    - reference utils/metrics.py:100-129 (torch.topk + compare with the label)
  * TestMeterOracle  - reference utils/meters.py:24-176 TestMeter (update_stats :82-112, finalize_metrics :141-170)

Pinned by tests/golden/meters.npz, which oracle/make_golden_meters.py produced by running the reference's own
`utils.metrics.topks_correct` and `utils.meters.TestMeter` on seeded inputs (tests/test_meters.py).  Under exactly equal scores
torch.topk's order is unspecified; this restatement (and the HIP kernel) rank equal scores by class index - the golden inputs are
continuous random scores without ties, plus one tie case that is checked against this rule only."""
import numpy as np


def softmax_rows(x):
    x = np.asarray(x, dtype=np.float32)
    e = np.exp(x - x.max(axis=-1, keepdims=True), dtype=np.float32)
    return (e / e.sum(axis=-1, keepdims=True, dtype=np.float32)).astype(np.float32)


def label_ranks(preds, labels):
    """rank of every row's label in a stable descending sort of the row"""
    preds = np.asarray(preds, dtype=np.float32)
    labels = np.asarray(labels, dtype=np.int64)
    n, K = preds.shape
    s = preds[np.arange(n), np.clip(labels, 0, K - 1)][:, None]
    j = np.arange(K)[None, :]
    r = ((preds > s) | ((preds == s) & (j < labels[:, None]))).sum(axis=1)
    return np.where((labels >= 0) & (labels < K), r, K)


def topks_correct(preds, labels, ks):
    r = label_ranks(preds, labels)
    return [np.float32((r < k).sum()) for k in ks]


def topk_accuracies(preds, labels, ks):
    n = np.float32(len(labels))
    return [np.float32(x / n * np.float32(100.0)) for x in topks_correct(preds, labels, ks)]


class TestMeterOracle:
    __test__ = False

    def __init__(self, num_videos, num_clips, num_cls, ensemble_method="sum"):
        self.num_clips, self.method = num_clips, ensemble_method
        self.video_preds = np.zeros((num_videos, num_cls), np.float32)
        self.video_labels = np.zeros(num_videos, np.int64)
        self.clip_count = np.zeros(num_videos, np.int64)

    def update_stats(self, preds, labels, clip_ids):
        preds = np.asarray(preds, np.float32)
        for i in range(preds.shape[0]):                          # the reference's loop order: fp32 additions happen clip by clip
            vid = int(clip_ids[i]) // self.num_clips
            if vid < 0 or vid >= len(self.video_labels):
                raise IndexError("clip id outside the meter")
            if self.video_labels[vid] > 0:
                assert self.video_labels[vid] == labels[i], "views of one video carry different labels"
            self.video_labels[vid] = labels[i]
            if self.method == "sum":
                self.video_preds[vid] = self.video_preds[vid] + preds[i]
            else:
                self.video_preds[vid] = np.maximum(self.video_preds[vid], preds[i])
            self.clip_count[vid] += 1

    def finalize_metrics(self, ks=(1, 5)):
        n = np.float32(self.video_preds.shape[0])
        return {"top{}_acc".format(k): "{:.2f}".format(float(x / n * np.float32(100.0))) for k, x in zip(ks, topks_correct(self.video_preds, self.video_labels, ks))}
