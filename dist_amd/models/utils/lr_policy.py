"""Per-iteration learning-rate schedule (reference models/utils/lr_policy.py:10-44)."""
import math


def lr_func_cosine(cfg, cur_epoch):
    return cfg.OPTIMIZER.BASE_LR * (math.cos(math.pi * cur_epoch / cfg.OPTIMIZER.MAX_EPOCH) + 1.0) * 0.5


def get_lr_func(policy):
    if policy != "cosine":
        raise NotImplementedError(f"Unknown LR policy: {policy} (the DiST yamls use cosine)")
    return lr_func_cosine


def get_lr_at_epoch(cfg, cur_epoch):
    lr = get_lr_func(cfg.OPTIMIZER.LR_POLICY)(cfg, cur_epoch)
    if cur_epoch < cfg.OPTIMIZER.WARMUP_EPOCHS:
        lr_start = cfg.OPTIMIZER.WARMUP_START_LR
        lr_end = get_lr_func(cfg.OPTIMIZER.LR_POLICY)(cfg, cfg.OPTIMIZER.WARMUP_EPOCHS)
        alpha = (lr_end - lr_start) / cfg.OPTIMIZER.WARMUP_EPOCHS
        lr = cur_epoch * alpha + lr_start
    return lr
