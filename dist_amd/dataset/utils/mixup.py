"""GPU Mixup / CutMix with soft targets (reference dataset/utils/mixup.py:18-23,103-319, batch mode).

Pairs clips within the rank's batch (`x.flip(0)`), so data-parallel ranks exchange nothing."""
import numpy as np
import torch


def one_hot(x, num_classes, on_value=1.0, off_value=0.0):
    x = x.long().view(-1, 1)
    return torch.full((x.size(0), num_classes), off_value, device=x.device).scatter_(1, x, on_value)


def mixup_target(target, num_classes, lam=1.0, smoothing=0.0):
    off = smoothing / num_classes
    on = 1.0 - smoothing + off
    y1 = one_hot(target, num_classes, on, off)
    y2 = one_hot(target.flip(0), num_classes, on, off)
    return y1 * lam + y2 * (1.0 - lam)


def rand_bbox(img_shape, lam, rng):
    h, w = img_shape[-2:]
    ratio = np.sqrt(1 - lam)
    ch, cw = int(h * ratio), int(w * ratio)
    cy, cx = rng.integers(0, h), rng.integers(0, w)
    yl, yh = np.clip(cy - ch // 2, 0, h), np.clip(cy + ch // 2, 0, h)
    xl, xh = np.clip(cx - cw // 2, 0, w), np.clip(cx + cw // 2, 0, w)
    return yl, yh, xl, xh


class Mixup:
    def __init__(self, mixup_alpha=1.0, cutmix_alpha=0.0, cutmix_minmax=None, prob=1.0, switch_prob=0.5, mode="batch",
                 correct_lam=True, label_smoothing=0.1, num_classes=1000, seed=0):
        self.mixup_alpha, self.cutmix_alpha = mixup_alpha, cutmix_alpha
        self.mix_prob, self.switch_prob = prob, switch_prob
        self.label_smoothing, self.num_classes = label_smoothing, num_classes
        self.correct_lam = correct_lam
        self.rng = np.random.default_rng(seed)
        assert mode == "batch", "the DiST yamls use MIXUP.MODE: batch"

    def _params(self):
        lam, use_cutmix = 1.0, False
        if self.rng.random() < self.mix_prob:
            if self.mixup_alpha > 0.0 and self.cutmix_alpha > 0.0:
                use_cutmix = self.rng.random() < self.switch_prob
                lam = self.rng.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix else self.rng.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.mixup_alpha > 0.0:
                lam = self.rng.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.cutmix_alpha > 0.0:
                use_cutmix, lam = True, self.rng.beta(self.cutmix_alpha, self.cutmix_alpha)
        return float(lam), use_cutmix

    def __call__(self, x, target):
        """x: [b,3,T,H,W] (modified in place), target: [b] int labels -> (x, soft target [b,K])"""
        assert x.shape[0] % 2 == 0, "batch size should be even when using mixup"
        lam, use_cutmix = self._params()
        if lam != 1.0:
            if use_cutmix:
                yl, yh, xl, xh = rand_bbox(x.shape, lam, self.rng)
                x[..., yl:yh, xl:xh] = x.flip(0)[..., yl:yh, xl:xh]
                if self.correct_lam:
                    lam = 1.0 - (yh - yl) * (xh - xl) / float(x.shape[-2] * x.shape[-1])
            else:
                x.mul_(lam).add_(x.flip(0).mul_(1.0 - lam))
        return x, mixup_target(target, self.num_classes, lam, self.label_smoothing)
