#!/usr/bin/env python3
"""Golden OUTPUT of the reference's text tower.  TEST INFRASTRUCTURE, build container only (needs /root/reference, read-only).

Builds the reference CLIP (models/base/clip.py `build_model`) from the procedural tiny visual / dist_net weights plus a PROCEDURAL
text tower (dist_amd/synth.py `text_tower_state_dict`), runs the reference's own `encode_text` / `cache_text` (clip.py:420-452) on
procedural label tokens and writes the text features to tests/golden/text_tiny.npz.  Inputs are regenerated from dist_amd/synth.py
wherever the fixture is consumed; nothing of the reference is copied.

    python oracle/make_golden_text.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from dist_amd import synth  # noqa: E402
import make_golden as mg  # noqa: E402


def main():
    assert os.path.isdir(mg.REF), "the reference is only present in the build container"
    mg._stub_modules()
    sys.path.insert(0, mg.REF)
    from models.base import clip
    g = synth.geometry("tiny")
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.state_dict(g).items()}
    sd.update({k: torch.from_numpy(v.copy()) for k, v in synth.text_tower_state_dict(embed=g.E).items()})
    model = clip.build_model(mg.make_cfg(g), dict(sd))
    missing = model.load_state_dict(sd, strict=False)
    assert not [k for k in missing[0] if not k.startswith("dist_net.") or k in sd], missing[0]
    assert not missing[1], missing[1]
    model.eval()
    tokens = torch.from_numpy(synth.label_tokens(g.K))
    with torch.no_grad():
        feats, eot, _ = model.encode_text(tokens, None)                 # clip.py:420-435
        cached, cached_eot, _ = model.cache_text(tokens, None)          # clip.py:437-452
    assert torch.equal(feats, cached)
    out = os.path.join(ROOT, "tests", "golden", "text_tiny.npz")
    np.savez_compressed(out, text_features=feats.numpy(), eot_features=eot.numpy(), context=np.array(model.context_length),
                        heads=np.array(model.transformer.resblocks[0].attn.num_heads), layers=np.array(len(model.transformer.resblocks)))
    print("wrote", out, feats.shape, float(feats.abs().mean()))


if __name__ == "__main__":
    main()
