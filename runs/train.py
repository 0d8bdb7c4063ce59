#!/usr/bin/env python3
"""Train entry point with the reference's step semantics (reference runs/train.py:40-206,331-431):
mixup -> forward -> soft-target CE -> zero_grad / backward -> (gradient all-reduce of dist_net only)
-> AdamW with a per-iteration cosine + warm-up learning rate -> one fused metric reduce."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from dist_amd.dataset.synthetic import build_loader, label_texts
from dist_amd.dataset.utils.mixup import Mixup
from dist_amd.models.base.builder import build_model
from dist_amd.models.utils import losses
from dist_amd.models.utils import optimizer as optim
from dist_amd.utils import checkpoint as cu
from dist_amd.utils import distributed as du
from dist_amd.utils import metrics


def train_epoch(train_loader, model, model_ema, optimizer, cur_epoch, mixup_fn, cfg, texts, log=print):
    model.train()
    data_size = len(train_loader)
    t0 = time.time()
    stats = {}
    # One batch of look-ahead: the loader's next batch is fetched (and mixed) before this batch's step, so the frozen-ViT pass
    # over it runs beside this step (TRAIN.PIPELINE_VIT, default on; the order of loader / mixup draws is unchanged).
    pipe = bool(getattr(cfg.TRAIN, "PIPELINE_VIT", True)) and hasattr(model, "prefetch")
    it = iter(train_loader)
    # Host-resident batches (what the reference's loader yields; runs/train.py:81-101 copies them inside the step): the copy of batch n+2 is started beside
    # step n on a copy stream (dist_amd/utils/staging.py), so that step n+1 finds the frames of the batch it prefetches resident.  One more batch of
    # look-ahead on the LOADER only: mixup still draws in batch order, right before the batch's ViT pass is issued.
    stager = [None]

    def stage(item):
        if item is not None and not item[0]["video"].is_cuda:
            if stager[0] is None:
                from dist_amd.utils.staging import HostStager
                stager[0] = HostStager(depth=3)
            item[0]["_ticket"] = stager[0].submit(item[0]["video"])
        return item

    def fetch(item):
        if item is None:
            return None
        if "_ticket" in item[0]:
            item[0]["video"] = stager[0].wait(item[0]["_ticket"])     # (copied a step ago: nothing waits)
        if mixup_fn is not None:                                       # reference runs/train.py:92-93
            _, item[1]["supervised"] = mixup_fn(item[0], item[1]["supervised"].to(item[0]["video"].device))   # the clips are mixed in place
        else:
            item[1]["supervised"] = item[1]["supervised"].to(item[0]["video"].device)
        return item

    ahead = stage(next(it, None))
    nxt = fetch(ahead)
    ahead = stage(next(it, None))
    cur_iter = -1
    while nxt is not None:
        cur_iter += 1
        inputs, labels, _, meta = nxt
        nxt = fetch(ahead)
        ahead = stage(next(it, None))
        if pipe and nxt is not None:
            model.prefetch(nxt[0])
        inputs["texts"] = texts
        lr = optim.get_epoch_lr(cur_epoch + cfg.TRAIN.NUM_FOLDS * float(cur_iter) / data_size, cfg)
        optim.set_lr(optimizer, lr)
        preds, logits = model(inputs)
        if "_ticket" in inputs:
            stager[0].release(inputs.pop("_ticket"))               # the forward waited for this batch's features: its frames are dead, the buffer may be refilled
        loss, _, _ = losses.calculate_loss(cfg, preds, logits, labels, cur_epoch + cfg.TRAIN.NUM_FOLDS * float(cur_iter) / data_size)
        optimizer.zero_grad()
        loss.backward()
        model.grad_sync.reduce()
        optimizer.step()
        if pipe and nxt is not None:
            model.adopt()
        hard = labels["supervised"].argmax(dim=1) if labels["supervised"].dim() == 2 else labels["supervised"]
        top1, top5 = metrics.topk_errors(preds.detach(), hard, (1, 5))
        loss_r, top1, top5 = du.all_reduce([loss.detach(), top1, top5])          # one packed collective, one host sync
        stats = {"epoch": cur_epoch, "iter": cur_iter, "loss": float(loss_r), "top1_err": float(top1), "top5_err": float(top5), "lr": lr}
        if not torch.isfinite(loss_r):
            raise RuntimeError("ERROR: Got NaN losses")                             # reference utils/misc.py:25-32
        if du.is_master_proc() and (cur_iter % max(1, cfg.LOG_PERIOD) == 0):
            clips = (cur_iter + 1) * inputs["video"].shape[0] * du.get_world_size()
            log({**stats, "clips_per_s": round(clips / (time.time() - t0), 1)})
    return stats


@torch.no_grad()
def eval_epoch(val_loader, model, cur_epoch, cfg, texts):
    model.eval()
    n, c1, c5 = 0, 0.0, 0.0
    pipe = bool(getattr(cfg.TRAIN, "PIPELINE_VIT", True)) and hasattr(model, "prefetch")
    it = iter(val_loader)
    nxt = next(it, None)
    while nxt is not None:
        inputs, labels, _, _ = nxt
        nxt = next(it, None)
        if pipe and nxt is not None:
            model.prefetch(nxt[0])             # frozen ViT of the next batch beside this batch's branch forward
        inputs["texts"] = texts
        preds, _ = model(inputs)
        if pipe and nxt is not None:
            model.adopt()
        k1, k5 = metrics.topks_correct(preds, labels["supervised"], (1, 5))
        k1, k5 = du.all_reduce([k1, k5], average=False)
        c1 += float(k1); c5 += float(k5); n += preds.size(0) * du.get_world_size()
    return {"epoch": cur_epoch, "top1_acc": 100.0 * c1 / max(n, 1), "top5_acc": 100.0 * c5 / max(n, 1)}


def train(cfg):
    np.random.seed(cfg.RANDOM_SEED)                             # reference runs/train.py:340-342 (Mixup draws from np.random)
    torch.manual_seed(cfg.RANDOM_SEED)
    model, model_ema = build_model(cfg)
    optimizer = optim.construct_optimizer(model, cfg)
    start_epoch = 0
    if cfg.TRAIN.AUTO_RESUME and os.path.isdir(cu.get_checkpoint_dir(cfg.OUTPUT_DIR)) and os.listdir(cu.get_checkpoint_dir(cfg.OUTPUT_DIR)):
        start_epoch = cu.load_checkpoint(cu.get_last_checkpoint(cfg.OUTPUT_DIR), model, optimizer) + 1
    elif getattr(cfg.TRAIN, "CHECKPOINT_FILE_PATH", ""):
        cu.load_checkpoint(cfg.TRAIN.CHECKPOINT_FILE_PATH, model, None)
    train_loader = build_loader(cfg, "train")
    val_loader = build_loader(cfg, "val") if cfg.TRAIN.EVAL_PERIOD > 0 else None
    texts = label_texts(cfg, vocab=model.backbone.base_encoder.vocab_size)
    mixup_fn = None
    if cfg.AUGMENTATION.MIXUP.ENABLE or cfg.AUGMENTATION.CUTMIX.ENABLE:
        mixup_fn = Mixup(cfg)                                  # reference runs/train.py:388; draws from np.random like the reference
    assert (cfg.OPTIMIZER.MAX_EPOCH - start_epoch) % cfg.TRAIN.NUM_FOLDS == 0, "Total training epochs should be divisible by cfg.TRAIN.NUM_FOLDS."
    stats = {}
    for cur_epoch in range(start_epoch, cfg.OPTIMIZER.MAX_EPOCH, cfg.TRAIN.NUM_FOLDS):
        stats = train_epoch(train_loader, model, model_ema, optimizer, cur_epoch, mixup_fn, cfg, texts)
        if du.is_master_proc() and cfg.TRAIN.CHECKPOINT_PERIOD > 0 and (cur_epoch + cfg.TRAIN.NUM_FOLDS) % cfg.TRAIN.CHECKPOINT_PERIOD == 0:
            cu.save_checkpoint(cfg.OUTPUT_DIR, model, optimizer, cur_epoch + cfg.TRAIN.NUM_FOLDS - 1, cfg)
        if val_loader is not None and (cur_epoch + cfg.TRAIN.NUM_FOLDS) % cfg.TRAIN.EVAL_PERIOD == 0:
            ev = eval_epoch(val_loader, model, cur_epoch, cfg, texts)
            if du.is_master_proc():
                print(ev)
        if getattr(cfg.TRAIN, "MAX_STEPS_DEBUG", 0):
            break
    return stats
