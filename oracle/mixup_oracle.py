"""CPU restatement (numpy) of the reference's batch-mode Mixup / CutMix and soft target.

TEST INFRASTRUCTURE ONLY: imported by tests/ and oracle/make_golden_mixup.py, never by the product path
(dist_amd.dataset.utils.mixup runs the HIP kernels dist_op_mixup / dist_op_cutmix / dist_op_mixup_target and
raises without a GPU).

Follows /root/reference/dataset/utils/mixup.py:
  one_hot / mixup_target          :13-23
  rand_bbox                       :43-64
  cutmix_bbox_and_lam             :89-100
  Mixup._params_per_batch         :160-176
  Mixup._mix_batch                :212-223
  Mixup.__call__ (video only)     :303-325
Random numbers come from numpy's GLOBAL generator in the reference's call order (np.random.rand, np.random.rand,
np.random.beta, np.random.randint, np.random.randint), so after np.random.seed(s) this restatement, the reference and
the HIP-backed class draw the same lam and box.  Pinned by tests/golden/mixup.npz (oracle/make_golden_mixup.py runs the
reference's own class on the same seeds).

Arithmetic follows torch's: a Python-float scalar multiplies an fp32 tensor as float32(scalar); every tensor op rounds
to fp32 once.
"""
import numpy as np


def one_hot(x, num_classes, on_value=1.0, off_value=0.0):                       # :13-15
    out = np.full((len(x), num_classes), np.float32(off_value), dtype=np.float32)
    out[np.arange(len(x)), np.asarray(x, dtype=np.int64)] = np.float32(on_value)
    return out


def mixup_target(target, num_classes, lam=1.0, smoothing=0.0):                  # :18-23
    off_value = smoothing / num_classes
    on_value = 1.0 - smoothing + off_value
    target = np.asarray(target, dtype=np.int64)
    y1 = one_hot(target, num_classes, on_value, off_value)
    y2 = one_hot(target[::-1], num_classes, on_value, off_value)
    return (y1 * np.float32(lam)).astype(np.float32) + (y2 * np.float32(1.0 - lam)).astype(np.float32)


def rand_bbox(img_shape, lam, margin=0.0):                                      # :43-64 (count=None)
    ratio = np.sqrt(1 - lam)
    img_h, img_w = img_shape[-2:]
    cut_h, cut_w = int(img_h * ratio), int(img_w * ratio)
    margin_y, margin_x = int(margin * cut_h), int(margin * cut_w)
    cy = np.random.randint(0 + margin_y, img_h - margin_y)
    cx = np.random.randint(0 + margin_x, img_w - margin_x)
    yl = np.clip(cy - cut_h // 2, 0, img_h)
    yh = np.clip(cy + cut_h // 2, 0, img_h)
    xl = np.clip(cx - cut_w // 2, 0, img_w)
    xh = np.clip(cx + cut_w // 2, 0, img_w)
    return int(yl), int(yh), int(xl), int(xh)


def cutmix_bbox_and_lam(img_shape, lam, correct_lam=True):                      # :89-100 (ratio_minmax=None: the DiST yamls leave MINMAX empty)
    yl, yu, xl, xu = rand_bbox(img_shape, lam)
    if correct_lam:
        bbox_area = (yu - yl) * (xu - xl)
        lam = 1.0 - bbox_area / float(img_shape[-2] * img_shape[-1])
    return (yl, yu, xl, xu), lam


def params_per_batch(mixup_alpha, cutmix_alpha, mix_prob, switch_prob, enabled=True):   # :160-176
    lam, use_cutmix = 1.0, False
    if enabled and np.random.rand() < mix_prob:
        if mixup_alpha > 0.0 and cutmix_alpha > 0.0:
            use_cutmix = np.random.rand() < switch_prob
            lam_mix = np.random.beta(cutmix_alpha, cutmix_alpha) if use_cutmix else np.random.beta(mixup_alpha, mixup_alpha)
        elif mixup_alpha > 0.0:
            lam_mix = np.random.beta(mixup_alpha, mixup_alpha)
        elif cutmix_alpha > 0.0:
            use_cutmix = True
            lam_mix = np.random.beta(cutmix_alpha, cutmix_alpha)
        else:
            assert False, "One of mixup_alpha > 0., cutmix_alpha > 0., cutmix_minmax not None should be true."
        lam = float(lam_mix)
    return lam, bool(use_cutmix)


def mix_batch(x, mixup_alpha, cutmix_alpha, mix_prob, switch_prob):             # :212-223; x: fp32 [b,3,T,H,W], modified in place
    lam, use_cutmix = params_per_batch(mixup_alpha, cutmix_alpha, mix_prob, switch_prob)
    info = {"lam_raw": lam, "use_cutmix": use_cutmix, "bbox": (0, 0, 0, 0)}
    if lam == 1.0:
        return 1.0, info
    if use_cutmix:
        (yl, yh, xl, xh), lam = cutmix_bbox_and_lam(x.shape, lam)
        info["bbox"] = (yl, yh, xl, xh)
        x[:, :, :, yl:yh, xl:xh] = x[::-1][:, :, :, yl:yh, xl:xh].copy()
    else:
        x_flipped = (x[::-1] * np.float32(1.0 - lam)).astype(np.float32)
        x[...] = (x * np.float32(lam)).astype(np.float32) + x_flipped
    return lam, info


def mixup_call(x, target, num_classes, mixup_alpha=0.8, cutmix_alpha=1.0, mix_prob=1.0, switch_prob=0.5, label_smoothing=0.1):
    """Mixup.__call__ for {"video": x} (:303-325): returns (lam, soft target, info); x is modified in place."""
    assert x.dtype == np.float32
    lam, info = mix_batch(x, mixup_alpha, cutmix_alpha, mix_prob, switch_prob)
    return lam, mixup_target(target, num_classes, lam, label_smoothing), info
