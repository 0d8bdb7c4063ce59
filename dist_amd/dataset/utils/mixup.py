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
from typing import NamedTuple, Optional, Tuple

import numpy as np
import torch

from ... import ops


def mixup_target(target, num_classes, lam=1.0, smoothing=0.0, device="cuda"):
    """y1 * lam + y2 * (1 - lam) over smoothed one-hot rows of target / target.flip(0) (reference :18-23)."""
    return ops.mixup_target(_labels_on_gpu(target), num_classes, lam, smoothing)


def label_smoothing_target(target, num_classes, smoothing=0.0, device="cuda"):
    """smoothed one-hot rows (reference :25-29) = mixup_target with lam = 1 (y1*1 + y2*0 is exact)."""
    return ops.mixup_target(_labels_on_gpu(target), num_classes, 1.0, smoothing)


def label_smoothing(cfg, target):
    """reference :31-40"""
    if isinstance(target, dict):
        return {k: label_smoothing_target(v, cfg.VIDEO.HEAD.NUM_CLASSES[i], cfg.AUGMENTATION.LABEL_SMOOTHING) for i, (k, v) in enumerate(target.items())}
    return label_smoothing_target(target, cfg.VIDEO.HEAD.NUM_CLASSES, cfg.AUGMENTATION.LABEL_SMOOTHING)


class MixPlan(NamedTuple):
    """What one call does to the batch: kind "none" | "mixup" | "cutmix", the mixing weight that also builds the soft target,
    and (cutmix) the box [y0, y1) x [x0, x1) that is swapped between clip i and clip b-1-i."""
    kind: str
    lam: float
    box: Optional[Tuple[int, int, int, int]] = None


def _span(centre, extent, limit):
    """[centre - extent // 2, centre + extent // 2) cut to [0, limit]"""
    half = extent // 2
    return max(centre - half, 0), min(centre + half, limit)


def draw_plan(height, width, mixup_alpha, cutmix_alpha, mix_prob, switch_prob, minmax=None, enabled=True):
    """Every random decision of one batch-mode call, drawn from numpy's GLOBAL generator in the reference's order
    (dataset/utils/mixup.py:160-176 then :43-64 / :67-86; pinned draw by draw by tests/golden/mixup.npz):
      1. rand()                     apply at all?            (skipped when disabled)
      2. rand()                     cutmix instead of mixup? (only when both alphas are positive)
      3. beta(a, a)                 the mixing weight, a = the chosen mode's alpha
      4. cutmix box: randint(0, H), randint(0, W) -> centre of a box with sides int(H r), int(W r), r = sqrt(1 - lam), clipped by the
         border; or, with `minmax`, randint for the two side lengths then for the two corners.
    The weight that reaches the soft target is the area actually swapped (the reference's correct_lam=True, :96-99)."""
    if not enabled or not (np.random.rand() < mix_prob):
        return MixPlan("none", 1.0)
    both = mixup_alpha > 0.0 and cutmix_alpha > 0.0
    if not both and not (mixup_alpha > 0.0 or cutmix_alpha > 0.0):
        raise AssertionError("One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true.")
    cut = (np.random.rand() < switch_prob) if both else not (mixup_alpha > 0.0)
    alpha = cutmix_alpha if cut else mixup_alpha
    lam = float(np.random.beta(alpha, alpha))
    if lam == 1.0:
        return MixPlan("none", 1.0)
    if not cut:
        return MixPlan("mixup", lam)
    if minmax is not None:
        lo, hi = minmax
        box_h = int(np.random.randint(int(height * lo), int(height * hi)))
        box_w = int(np.random.randint(int(width * lo), int(width * hi)))
        y0 = int(np.random.randint(0, height - box_h))
        x0 = int(np.random.randint(0, width - box_w))
        y1, x1 = y0 + box_h, x0 + box_w
    else:
        r = np.sqrt(1 - lam)
        box_h, box_w = int(height * r), int(width * r)
        cy = int(np.random.randint(0, height))
        cx = int(np.random.randint(0, width))
        (y0, y1), (x0, x1) = _span(cy, box_h, height), _span(cx, box_w, width)
    return MixPlan("cutmix", 1.0 - ((y1 - y0) * (x1 - x0)) / float(height * width), (y0, y1, x0, x1))


def _on_gpu(t, what):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise RuntimeError(f"Mixup: {what} must be a GPU tensor (the mix runs as HIP kernels; there is no CPU path)")
    return t


def _labels_on_gpu(t):
    """labels arrive from the loader as host tensors in the reference (runs/train.py:92-93): they are moved, the clips are not"""
    if torch.is_tensor(t) and not t.is_cuda and torch.cuda.is_available():
        t = t.cuda(non_blocking=True)
    return _on_gpu(t, "target")


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
        # TRAIN.FUSE_MIXUP (dist_amd): the clips are NOT mixed in memory - the plan is left on the clip tensor (`video._dist_mix`) and the frozen ViT's patch-row
        # gather applies it while it reads the frames (Engine.vit_forward / vit_prefetch -> dist_vit_mix_next -> dist_op_patchify_mixed: the same arithmetic, one
        # pass over 308 MB instead of three).  The soft target is built as always.  Off = the reference's in-place mix.
        self.fuse = bool(getattr(getattr(cfg, "TRAIN", None), "FUSE_MIXUP", False))

    def plan(self, shape):
        """the batch's MixPlan for a clip tensor of `shape` [..., H, W] (consumes the global numpy stream like the reference's call)"""
        return draw_plan(int(shape[-2]), int(shape[-1]), self.mixup_alpha, self.cutmix_alpha, self.mix_prob, self.switch_prob,
                         minmax=self.cutmix_minmax, enabled=self.mixup_enabled)

    def apply(self, x, plan):
        """x [b, 3, T, H, W] fp32 on the GPU, mixed in place (reference :212-223): dist_op_cutmix / dist_op_mixup"""
        _on_gpu(x, "video")
        if self.fuse and plan.kind != "none" and x.dtype == torch.float32 and x.is_contiguous():
            x._dist_mix = plan                      # consumed by the next ViT pass over this tensor
            return plan.lam
        if plan.kind == "cutmix":
            ops.cutmix_(x, *plan.box)
        elif plan.kind == "mixup":
            ops.mixup_(x, plan.lam)
        return plan.lam

    def __call__(self, x, target):
        """reference :303-325 for {"video": tensor}"""
        assert isinstance(x, dict)
        if "video" in x and torch.is_tensor(x["video"]) and "flow" not in x:
            lam = self.apply(x["video"], self.plan(x["video"].shape))
        else:
            raise NotImplementedError("Mixup: only {'video': tensor} inputs are on the DiST path")
        if isinstance(target, dict):
            target_ = {k: mixup_target(v, self.num_classes[i], lam, self.label_smoothing) for i, (k, v) in enumerate(target.items())}
        else:
            target_ = mixup_target(target, self.num_classes, lam, self.label_smoothing)
        return x, target_
