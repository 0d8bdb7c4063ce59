"""CPU: the frozen text tower (`encode_text`, reference models/base/clip.py:420-435; attention mask :411-417; block :112-135) against
the REFERENCE's own output for a procedural tower and procedural label tokens (tests/golden/text_tiny.npz, oracle/make_golden_text.py).
The tower is plain torch in the product (it runs once per label set, off the per-step path), so it is checked here without a GPU; the
same function on the full CLIP module is checked in tests/test_dropin_gpu.py.  Round 2 had this unpinned - and wrong: the block stack
was registered as `transformer.{i}` instead of `transformer.resblocks.{i}`, so a CLIP checkpoint's text weights never loaded."""
import os

import numpy as np
import torch

from dist_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_text_tower_reproduces_the_reference_features():
    from dist_amd.models.base.clip import TextTower
    gold = np.load(os.path.join(GOLD, "text_tiny.npz"))
    g = synth.geometry("tiny")
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth.text_tower_state_dict(embed=g.E).items()}
    tower = TextTower(g.E, int(gold["context"]), sd["token_embedding.weight"].shape[0], sd["ln_final.weight"].shape[0], int(gold["heads"]), int(gold["layers"]))
    res = tower.load_state_dict(sd, strict=True)                # STRICT: every reference key has a module, every module a key
    assert not res.missing_keys and not res.unexpected_keys
    tower.eval()
    tokens = torch.from_numpy(synth.label_tokens(g.K))
    assert (tokens.argmax(1) >= 3).all() and (tokens[:, 0] == tokens[0, 0]).all()
    with torch.no_grad():
        feats, eot = tower.encode_text(tokens)
    np.testing.assert_allclose(feats.numpy(), gold["text_features"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(eot.numpy(), gold["eot_features"], rtol=1e-4, atol=1e-5)
    # causal: tokens behind the end token do not influence its feature
    t2 = tokens.clone()
    n = int(t2[0].argmax())
    t2[0, n + 1:n + 4] = 5
    with torch.no_grad():
        f2, _ = tower.encode_text(t2)
    np.testing.assert_allclose(f2[0].numpy(), feats[0].numpy(), rtol=1e-5, atol=1e-6)


def test_state_dict_key_names_are_the_openai_clip_ones():
    from dist_amd.models.base.clip import TextTower
    names = set(TextTower(64, 77, 96, 64, 1, 2).state_dict())
    assert {"token_embedding.weight", "positional_embedding", "ln_final.weight", "ln_final.bias", "text_projection",
            "transformer.resblocks.0.attn.in_proj_weight", "transformer.resblocks.1.mlp.c_proj.bias", "transformer.resblocks.0.ln_2.weight"} <= names
    assert names == set(synth.text_tower_state_dict())
