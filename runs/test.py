#!/usr/bin/env python3
"""Multi-view evaluation (reference runs/test.py:24-178,181-322, utils/meters.py:24-176): forward under
no_grad (the head applies softmax at eval), all-gather of (preds, labels, video ids), per-video score
summation over NUM_ENSEMBLE_VIEWS x NUM_SPATIAL_CROPS views, top-1 / top-5."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dist_amd.dataset.synthetic import build_loader, label_texts
from dist_amd.models.base.builder import build_model
from dist_amd.utils import checkpoint as cu
from dist_amd.utils import distributed as du
from dist_amd.utils.meters import TestMeter


@torch.no_grad()
def perform_test(test_loader, model, test_meter, cfg, texts):
    model.eval()
    # one batch of look-ahead (TRAIN.PIPELINE_VIT): the frozen-ViT pass of the next batch runs beside the branch forward,
    # the all-gather and the device-side meter update of this one (dist_vit_prefetch / dist_vit_adopt)
    pipe = bool(getattr(cfg.TRAIN, "PIPELINE_VIT", True)) and hasattr(model, "prefetch")
    it = iter(test_loader)
    nxt = next(it, None)
    while nxt is not None:
        inputs, labels, video_idx, _ = nxt
        nxt = next(it, None)
        if pipe and nxt is not None:
            model.prefetch(nxt[0])
        inputs["texts"] = texts
        preds, _ = model(inputs)
        if pipe and nxt is not None:
            model.adopt()
        lab, idx = labels["supervised"].to(preds.device, non_blocking=True), video_idx.to(preds.device, non_blocking=True)   # host tensors from a real loader
        preds, lab, idx = du.all_gather([preds, lab, idx])
        test_meter.update_stats(preds, lab, idx)                # device-side ensemble: no host copy, no synchronisation per iteration
    out = test_meter.finalize_metrics()
    test_meter.reset()
    return {k: (float(v) if k.startswith("top") else v) for k, v in out.items()}


def test(cfg):
    np.random.seed(cfg.RANDOM_SEED)                             # reference runs/test.py:190-191
    torch.manual_seed(cfg.RANDOM_SEED)
    model, _ = build_model(cfg)
    if getattr(cfg.TEST, "CHECKPOINT_FILE_PATH", ""):
        cu.load_checkpoint(cfg.TEST.CHECKPOINT_FILE_PATH, model, None)
    loader = build_loader(cfg, "test")
    views = cfg.TEST.NUM_ENSEMBLE_VIEWS * cfg.TEST.NUM_SPATIAL_CROPS
    assert len(loader.dataset) % views == 0
    meter = TestMeter(cfg, len(loader.dataset) // views, views, cfg.VIDEO.HEAD.NUM_CLASSES, len(loader),
                      getattr(cfg.DATA, "ENSEMBLE_METHOD", "sum"))                      # reference runs/test.py:240-248
    texts = label_texts(cfg, vocab=model.backbone.base_encoder.vocab_size)
    out = perform_test(loader, model, meter, cfg, texts)
    if du.is_master_proc():
        print(out)
    du.synchronize()
    return out
