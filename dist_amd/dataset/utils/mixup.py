"""Batch-mode Mixup / CutMix with soft targets on the GPU, drop-in for the reference's class
(dataset/utils/mixup.py:103-325; call site runs/train.py:92-93, built at runs/train.py:388 as `Mixup(cfg)`).

Same constructor (`Mixup(cfg)`), same call (`mixup_fn(inputs: {"video": [b,3,T,H,W]}, target) -> (inputs, soft_target)`),
same random stream: lam, the cutmix switch and the box are drawn from numpy's GLOBAL generator in the reference's order
(:160-176, :43-64), so after `np.random.seed(s)` both produce the same parameters.  The tensor work - the in-place mix of
the rank's clips (clip i with clip b-1-i, so ranks exchange nothing) and the soft target - runs as HIP kernels behind the C ABI
(dist_op_mixup / dist_op_cutmix / dist_op_mixup_target), bit-identical to the reference's torch ops
(tests/golden/mixup.npz).  There is no CPU path: a CPU tensor raises.

Only MODE: batch is implemented (every DiST yaml uses it, configs/projects/dist/vit_*.yaml:22); the reference's
elem / pair modes use `np.bool` (:146,191), which current numpy no longer has.
"""
import numpy as np
import torch

from ... import ops


def mixup_target(target, num_classes, lam=1.0, smoothing=0.0, device="cuda"):
    """y1 * lam + y2 * (1 - lam) over smoothed one-hot rows of target / target.flip(0) (reference :18-23)."""
    return ops.mixup_target(_on_gpu(target, "target"), num_classes, lam, smoothing)


def label_smoothing_target(target, num_classes, smoothing=0.0, device="cuda"):
    """smoothed one-hot rows (reference :25-29) = mixup_target with lam = 1 (y1*1 + y2*0 is exact)."""
    return ops.mixup_target(_on_gpu(target, "target"), num_classes, 1.0, smoothing)


def label_smoothing(cfg, target):
    """reference :31-40"""
    if isinstance(target, dict):
        return {k: label_smoothing_target(v, cfg.VIDEO.HEAD.NUM_CLASSES[i], cfg.AUGMENTATION.LABEL_SMOOTHING) for i, (k, v) in enumerate(target.items())}
    return label_smoothing_target(target, cfg.VIDEO.HEAD.NUM_CLASSES, cfg.AUGMENTATION.LABEL_SMOOTHING)


def rand_bbox(img_shape, lam, margin=0.0, count=None):
    """reference :43-64 (same np.random calls in the same order)"""
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = np.random.randint(0 + margin_y, img_h - margin_y, size=count)
    cx = np.random.randint(0 + margin_x, img_w - margin_x, size=count)
    yl = np.clip(cy - cut_h // 2, 0, img_h)
    yh = np.clip(cy + cut_h // 2, 0, img_h)
    xl = np.clip(cx - cut_w // 2, 0, img_w)
    xh = np.clip(cx + cut_w // 2, 0, img_w)
    return yl, yh, xl, xh


def rand_bbox_minmax(img_shape, minmax, count=None):
    """reference :67-86"""
    assert len(minmax) == 2
    img_h, img_w = img_shape[-2:]
    cut_h = np.random.randint(int(img_h * minmax[0]), int(img_h * minmax[1]), size=count)
    cut_w = np.random.randint(int(img_w * minmax[0]), int(img_w * minmax[1]), size=count)
    yl = np.random.randint(0, img_h - cut_h, size=count)
    xl = np.random.randint(0, img_w - cut_w, size=count)
    return yl, yl + cut_h, xl, xl + cut_w


def cutmix_bbox_and_lam(img_shape, lam, ratio_minmax=None, correct_lam=True, count=None):
    """reference :89-100"""
    if ratio_minmax is not None:
        yl, yu, xl, xu = rand_bbox_minmax(img_shape, ratio_minmax, count=count)
    else:
        yl, yu, xl, xu = rand_bbox(img_shape, lam, count=count)
    if correct_lam or ratio_minmax is not None:
        bbox_area = (yu - yl) * (xu - xl)
        lam = 1.0 - bbox_area / float(img_shape[-2] * img_shape[-1])
    return (yl, yu, xl, xu), lam


def _on_gpu(t, what):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise RuntimeError(f"Mixup: {what} must be a GPU tensor (the mix runs as HIP kernels; there is no CPU path)")
    return t


class Mixup:
    """reference :103-325, MODE: batch."""

    def __init__(self, cfg):
        aug = cfg.AUGMENTATION
        self.mixup_alpha = aug.MIXUP.ALPHA
        self.cutmix_alpha = aug.CUTMIX.ALPHA if aug.CUTMIX.ENABLE else 0.0
        self.cutmix_minmax = (aug.CUTMIX.MINMAX or None) if aug.CUTMIX.ENABLE else None
        if self.cutmix_minmax is not None:
            assert len(self.cutmix_minmax) == 2
            self.cutmix_alpha = 1.0
        self.mix_prob = aug.MIXUP.PROB
        self.switch_prob = aug.MIXUP.SWITCH_PROB
        self.label_smoothing = aug.LABEL_SMOOTHING
        self.num_classes = cfg.VIDEO.HEAD.NUM_CLASSES
        self.mode = aug.MIXUP.MODE
        if self.mode != "batch":
            raise NotImplementedError("AUGMENTATION.MIXUP.MODE: only `batch` (every DiST yaml) is implemented")
        self.correct_lam = True
        self.mixup_enabled = True

    def _params_per_batch(self):
        """reference :160-176"""
        lam = 1.0
        use_cutmix = False
        if self.mixup_enabled and np.random.rand() < self.mix_prob:
            if self.mixup_alpha > 0.0 and self.cutmix_alpha > 0.0:
                use_cutmix = np.random.rand() < self.switch_prob
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha) if use_cutmix else \
                    np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.mixup_alpha > 0.0:
                lam_mix = np.random.beta(self.mixup_alpha, self.mixup_alpha)
            elif self.cutmix_alpha > 0.0:
                use_cutmix = True
                lam_mix = np.random.beta(self.cutmix_alpha, self.cutmix_alpha)
            else:
                assert False, "One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true."
            lam = float(lam_mix)
        return lam, use_cutmix

    def _mix_batch(self, x):
        """reference :212-223; x is mixed in place by dist_op_cutmix / dist_op_mixup"""
        _on_gpu(x, "video")
        lam, use_cutmix = self._params_per_batch()
        if lam == 1.0:
            return 1.0
        if use_cutmix:
            (yl, yh, xl, xh), lam = cutmix_bbox_and_lam(x.shape, lam, ratio_minmax=self.cutmix_minmax, correct_lam=self.correct_lam)
            ops.cutmix_(x, int(yl), int(yh), int(xl), int(xh))
        else:
            ops.mixup_(x, lam)
        return lam

    def __call__(self, x, target):
        """reference :303-325 for {"video": tensor}"""
        assert isinstance(x, dict)
        if "video" in x and torch.is_tensor(x["video"]) and "flow" not in x:
            lam = self._mix_batch(x["video"])
        else:
            raise NotImplementedError("Mixup: only {'video': tensor} inputs are on the DiST path")
        if isinstance(target, dict):
            target_ = {k: mixup_target(v, self.num_classes[i], lam, self.label_smoothing) for i, (k, v) in enumerate(target.items())}
        else:
            target_ = mixup_target(target, self.num_classes, lam, self.label_smoothing)
        return x, target_
