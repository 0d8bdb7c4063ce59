"""Checkpoints (reference utils/checkpoint.py:102-143,277-347,452-576; process_dist_cpkt.py:10-30).

Loads the reference's published layout ({"model_state": {"backbone.base_encoder.<clip keys>": ...}},
the pre-release `ladder_net.*` module names included: process_dist_cpkt.py's table, pinned by tests/golden/ckpt_rename.json) and OpenAI CLIP state-dicts; saves ONLY dist_net (+ the
optimizer moments) instead of the whole 168 M-parameter model the reference writes every time."""
import os

import torch

PREFIX = "backbone.base_encoder."


def get_checkpoint_dir(path_to_job):
    return os.path.join(path_to_job, "checkpoints")


def make_checkpoint_dir(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    os.makedirs(d, exist_ok=True)
    return d


def get_path_to_checkpoint(path_to_job, epoch):
    return os.path.join(get_checkpoint_dir(path_to_job), "checkpoint_epoch_{:05d}.pyth".format(epoch))


def get_last_checkpoint(path_to_job):
    d = get_checkpoint_dir(path_to_job)
    names = sorted(f for f in os.listdir(d) if "checkpoint" in f) if os.path.isdir(d) else []
    assert names, f"No checkpoints found in '{d}'."
    return os.path.join(d, names[-1])


# pre-release module names of the branch -> released names (reference process_dist_cpkt.py:10-30); longest prefix first
_LADDER = (
    ("ladder_net.input_map_feat_nets", "dist_net.input_linears"),
    ("ladder_net.final_temporal_nets", "dist_net.adapooling_nets"),
    ("ladder_net.s2t_fuse_nets", "dist_net.integration2temporal_nets"),
    ("ladder_net.t2s_fuse_nets", "dist_net.temporal2integration_nets"),
    ("ladder_net.spatial_nets", "dist_net.integration_nets"),
    ("ladder_net.", "dist_net."),
)


def rename_key(k):
    """one checkpoint key in either spelling -> the released spelling (prefixes such as `backbone.base_encoder.` stay)."""
    i = k.find("ladder_net.")
    if i < 0:
        return k
    head, tail = k[:i], k[i:]
    for old, new in _LADDER:
        if tail.startswith(old):
            return head + new + tail[len(old):]
    return k


def rename_model_state(model_state):
    """drop-in for the reference's process_dist_cpkt.rename_model_state (same result, order kept)."""
    return {rename_key(k): v for k, v in model_state.items()}


def normalize_state_dict(sd):
    """reference key spellings -> CLIP keys (visual.*, dist_net.*, logit_scale, text tower): unwraps {"model_state": ...},
    strips the DDP `module.` and the `backbone.base_encoder.` prefixes, applies the pre-release -> released rename."""
    if "model_state" in sd:
        sd = sd["model_state"]
    out = {}
    for k, v in sd.items():
        if k.startswith("module."):
            k = k[7:]
        if k.startswith(PREFIX):
            k = k[len(PREFIX):]
        out[rename_key(k)] = v
    return out


def save_checkpoint(path_to_job, model, optimizer, epoch, cfg, full=False):
    eng = model.backbone.base_encoder.engine
    names = list(eng.tables[0]) + (list(eng.tables[1]) if full else [])
    state = {PREFIX + n: eng.view(n).detach().cpu().clone() for n in names}
    state[PREFIX + "logit_scale"] = eng.logit_scale.view(()).cpu().clone()
    ck = {"epoch": epoch, "model_state": state, "cfg": cfg.to_dict() if hasattr(cfg, "to_dict") else None,
          "optimizer_state": {"step": eng.step_count,
                              "exp_avg": None if eng.exp_avg is None else eng.exp_avg.cpu(),
                              "exp_avg_sq": None if eng.exp_avg_sq is None else eng.exp_avg_sq.cpu()}}
    make_checkpoint_dir(path_to_job)
    path = get_path_to_checkpoint(path_to_job, epoch + 1)
    torch.save(ck, path)
    return path


def load_checkpoint(path, model, optimizer=None, strict=False):
    """reference utils/checkpoint.py:277-347: non-strict load, the two mismatch lists are reported (and kept in
    `load_checkpoint.last_mismatch`).  `strict=True` raises when a dist_net tensor of the model is missing from the file, when the file
    carries SOME visual.* tensors but not all of them (a dist_net-only file - what `save_checkpoint(full=False)` writes - is complete:
    the frozen towers keep the CLIP weights they were built from), or when a dist_net / ladder_net key of the file is left over.
    Text-tower and logit_scale keys never make a strict load fail."""
    assert os.path.exists(path), "Checkpoint '{}' not found".format(path)
    ck = torch.load(path, map_location="cpu")
    sd = normalize_state_dict(ck)
    clip = model.backbone.base_encoder
    missing = clip.load_state_dict(sd, strict=False)
    load_checkpoint.last_mismatch = (list(missing.missing_keys), list(missing.unexpected_keys))
    for what, keys in (("model", missing.missing_keys), ("checkpoint", missing.unexpected_keys)):
        print("Keys in {} not matched: {}{}".format(what, len(keys), " (" + ", ".join(keys[:4]) + (", ..." if len(keys) > 4 else "") + ")" if keys else ""))
    if strict:
        # only what the docstring names: a file that carries dist_net (+ visual) only - what save_checkpoint(full=False) writes - is complete
        # although the text tower / logit_scale keys of the model are "missing"
        has_visual = any(k.startswith("visual.") for k in sd)
        miss = [k for k in missing.missing_keys if k.startswith("dist_net.") or (has_visual and k.startswith("visual."))]
        left = [k for k in missing.unexpected_keys if k.startswith(("dist_net.", "ladder_net."))]
        if miss or left:
            raise KeyError(f"checkpoint mismatch: missing {miss[:8]}{' ...' if len(miss) > 8 else ''}, unexpected {left[:8]}{' ...' if len(left) > 8 else ''}")
    eng = clip.engine
    if optimizer is not None and isinstance(ck, dict) and ck.get("optimizer_state"):
        st = ck["optimizer_state"]
        eng.step_count = int(st.get("step", 0))
        if st.get("exp_avg") is not None:
            eng.exp_avg = st["exp_avg"].to(eng.device)
            eng.exp_avg_sq = st["exp_avg_sq"].to(eng.device)
    return ck.get("epoch", -1) if isinstance(ck, dict) else -1


load_checkpoint.last_mismatch = ([], [])
