"""Golden vectors for batch-mode Mixup / CutMix: runs the REFERENCE's own class (loaded from
/root/reference/dataset/utils/mixup.py; numpy + torch only) on seeded small inputs and stores inputs-by-recipe and OUTPUTS in
tests/golden/mixup.npz.  Runs only in the build container; nothing of the reference ships.

Per case: np.random.seed(seed); x = procedural fp32 clips [b,3,T,H,W]; labels; then exactly what Mixup.__call__ does for
{"video": x} (dataset/utils/mixup.py:303-325): lam = self._mix_batch(x); target = mixup_target(labels, K, lam, smoothing) -
with device='cpu' spelled out, because the reference's default device='cuda' (:13,18) cannot run here.
"""
import importlib.util
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

REF = "/root/reference/dataset/utils/mixup.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mixup.npz")


def procedural_clips(seed, b, T, H, W):
    """fp32 frames ~ N(0,1) from a counter-based generator (no dependence on np.random's global state)."""
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    return g.standard_normal((b, 3, T, H, W)).astype(np.float32)


def procedural_labels(seed, b, K):
    g = np.random.Generator(np.random.PCG64(2000 + seed))
    return g.integers(0, K, size=b).astype(np.int64)


CASES = [  # (seed, b, T, H, W, K, mixup_alpha, cutmix_alpha(enable), prob, switch_prob, smoothing)
    (0, 4, 2, 16, 16, 10, 0.8, 1.0, 1.0, 0.5, 0.1),
    (1, 4, 2, 16, 16, 10, 0.8, 1.0, 1.0, 0.5, 0.1),
    (2, 4, 2, 16, 16, 10, 0.8, 1.0, 1.0, 0.5, 0.1),
    (3, 6, 1, 12, 20, 7, 0.8, 1.0, 1.0, 0.5, 0.1),
    (4, 5, 2, 16, 16, 10, 0.8, 1.0, 1.0, 0.5, 0.1),     # odd batch: the middle clip is mixed with itself
    (5, 4, 2, 16, 16, 10, 0.8, 0.0, 1.0, 0.5, 0.0),     # mixup only, no smoothing
    (6, 4, 2, 16, 16, 10, 0.0, 1.0, 1.0, 0.5, 0.1),     # cutmix only
    (7, 4, 2, 16, 16, 10, 0.8, 1.0, 0.0, 0.5, 0.1),     # prob 0: lam = 1, nothing moves
    (8, 4, 2, 16, 16, 174, 0.8, 1.0, 1.0, 0.5, 0.1),    # SSV2 class count
    (11, 5, 2, 16, 16, 174, 0.8, 1.0, 1.0, 0.5, 0.1),   # cutmix, odd batch: the middle clip keeps its box
    (13, 6, 1, 14, 18, 400, 0.8, 1.0, 1.0, 0.5, 0.1),   # cutmix with a large box clipped by the frame border (lam correction), K400
    (21, 4, 2, 16, 16, 10, 0.8, 1.0, 1.0, 0.5, 0.1),    # cutmix with lam close to 1: a small box
]


def main():
    spec = importlib.util.spec_from_file_location("ref_mixup", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {"n_cases": np.int64(len(CASES))}
    kinds = []
    for ci, (seed, b, T, H, W, K, ma, ca, prob, sw, sm) in enumerate(CASES):
        cfg = NS(AUGMENTATION=NS(MIXUP=NS(ALPHA=ma, PROB=prob, SWITCH_PROB=sw, MODE="batch"),
                                 CUTMIX=NS(ENABLE=ca > 0.0, ALPHA=ca, MINMAX=None), LABEL_SMOOTHING=sm),
                 VIDEO=NS(HEAD=NS(NUM_CLASSES=K)))
        fn = ref.Mixup(cfg)
        x = torch.from_numpy(procedural_clips(seed, b, T, H, W))
        labels = torch.from_numpy(procedural_labels(seed, b, K))
        np.random.seed(seed)
        lam = fn._mix_batch(x)                                                   # reference :212-223 (modifies x in place)
        soft = ref.mixup_target(labels, K, lam, sm, device="cpu")               # reference :18-23
        out[f"c{ci}_meta"] = np.array([seed, b, T, H, W, K], dtype=np.int64)
        out[f"c{ci}_hyper"] = np.array([ma, ca, prob, sw, sm], dtype=np.float64)
        out[f"c{ci}_lam"] = np.float64(lam)
        out[f"c{ci}_x"] = x.numpy().copy()
        out[f"c{ci}_soft"] = soft.numpy().astype(np.float32)
        kinds.append("identity" if lam == 1.0 else "mixed")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", kinds)


if __name__ == "__main__":
    main()
