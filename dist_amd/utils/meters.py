"""`TestMeter` - the multi-view ensemble of the test loop (reference utils/meters.py:24-176), same constructor, methods and
attributes; the state lives on the device and `update_stats` is one dist_op_ensemble_update launch per iteration (the reference
copies predictions / labels / ids to the host and loops over the clips in Python).  The label-consistency assert and the
clip-id range check of the reference are collected in a device flag word and raised by `finalize_metrics` (or `check()`)."""
import json
import time

import torch

from .. import ops
from . import metrics


class TestMeter(object):
    __test__ = False                                           # not a pytest class

    def __init__(self, cfg, num_videos, num_clips, num_cls, overall_iters, ensemble_method="sum", device=None):
        if ensemble_method not in ("sum", "max"):
            raise NotImplementedError("Ensemble Method {} is not supported".format(ensemble_method))
        if not torch.cuda.is_available():
            raise RuntimeError("dist_amd TestMeter needs a GPU (state and updates are device-side)")
        self.cfg = cfg
        self.num_clips = num_clips
        self.overall_iters = overall_iters
        self.ensemble_method = ensemble_method
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self.video_preds = torch.zeros((num_videos, num_cls), device=dev)
        self.video_labels = torch.zeros((num_videos,), dtype=torch.long, device=dev)
        self.clip_count = torch.zeros((num_videos,), dtype=torch.long, device=dev)
        self.clip_indices = torch.linspace(0, num_videos - 1, num_videos).long()
        self._err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.model_ema_enabled = False
        self._t0 = time.perf_counter()
        self._dt = 0.0
        self.reset()

    def reset(self):
        self.clip_count.zero_()
        self.video_preds.zero_()
        self.video_labels.zero_()
        self._err.zero_()

    def update_stats(self, preds, labels, clip_ids):
        """preds [N, C] on the device; labels [N], clip_ids [N] on the device or - as the reference's loaders yield them - on the
        host (moved here; no synchronisation either way)."""
        dev = self.video_preds.device
        labels, clip_ids = labels.to(dev, non_blocking=True), clip_ids.to(dev, non_blocking=True)
        ops.ensemble_update(self.video_preds, self.video_labels, self.clip_count, preds, labels, clip_ids, self.num_clips,
                            ops.ENSEMBLE_SUM if self.ensemble_method == "sum" else ops.ENSEMBLE_MAX, self._err)

    def check(self):
        e = int(self._err.item())
        if e & 2:
            raise IndexError("clip id outside [0, num_videos * num_clips)")
        if e & 1:
            raise AssertionError("views of one video carry different labels")

    def iter_tic(self):
        self._t0 = time.perf_counter()

    def iter_toc(self):
        self._dt = time.perf_counter() - self._t0

    def log_iter_stats(self, cur_iter):
        period = getattr(self.cfg, "LOG_PERIOD", 0) if self.cfg is not None else 0
        if not period or (cur_iter + 1) % period != 0:
            return
        print(json.dumps({"split": "test_iter" if not self.model_ema_enabled else "ema_test_iter", "cur_iter": "{}".format(cur_iter + 1),
                          "time_diff": self._dt}))

    def finalize_metrics(self, ks=(1, 5)):
        """top-k accuracy over ALL videos (videos no view reached count as wrong unless their label is the arg-max of zeros,
        exactly as in the reference); returns the reference's log record."""
        self.check()
        incomplete = (self.clip_count != self.num_clips).nonzero().view(-1).tolist()
        if incomplete:
            seen = ", ".join(f"video {i} has {int(self.clip_count[i])}" for i in incomplete[:32])
            print(f"TestMeter: {len(incomplete)} videos did not receive all {self.num_clips} views ({seen}{', ...' if len(incomplete) > 32 else ''})")
        num_topks_correct = metrics.topks_correct(self.video_preds, self.video_labels, ks)
        topks = [(x / self.video_preds.size(0)) * 100.0 for x in num_topks_correct]
        stats = {"split": "test_final" if not self.model_ema_enabled else "ema_test_final"}
        for k, topk in zip(ks, topks):
            stats["top{}_acc".format(k)] = "{:.{prec}f}".format(float(topk), prec=2)
        return stats

    def set_model_ema_enabled(self, model_ema_enabled):
        self.model_ema_enabled = model_ema_enabled
