"""The operator-level GPU tests of the fused kernels take their references from the PINNED oracle (oracle/dist_oracle.py, checked against the
reference's own outputs by tests/test_oracle_golden.py) instead of restating the reference modules again: this file only renames the tests'
parameter dictionaries to the reference's state-dict names and reshapes rows <-> [b, frames, tokens, channels]."""
import os
import sys
from types import SimpleNamespace

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from dist_oracle import Oracle  # noqa: E402


def integration_oracle(w, Mp_rows, clips, t, Ltok, bf16=False, fused=False, requires_grad=False):
    """IntegrationNetwork (reference dist.py:16-45) through Oracle.integration_net on rows [clips*t*Ltok, Ci]; `w`: the module's parameter names
    ("ln.weight", "ffn.c_fc.weight", ...).  Returns (keep dict of [rows, C] tensors, the M' leaf)."""
    Ci = Mp_rows.shape[1]
    g = SimpleNamespace(layers=1, tk=w["temporal_ffn.c_fc2.weight"].shape[2])
    o = Oracle(g, {"dist_net.integration_nets.0." + k: v for k, v in w.items()}, dtype=torch.float64, bf16=bf16, fused=fused)
    x = Mp_rows.double().reshape(clips, t, Ltok, Ci).clone()
    if requires_grad:
        x.requires_grad_(True)
    keep = {}
    o.integration_net(x, 0, keep)
    flat = {k.split(".")[0]: v.reshape(-1, v.shape[-1]) for k, v in keep.items()}
    return flat, x


def temporal_net_oracle(X_rows, W1, b1, W2, b2, lnw, lnb, clips, T, G, tk, fused=False, requires_grad=False):
    """TemporalNet (reference dist.py:48-65) through Oracle.temporal_net in fp64 with the bf16 rounding points of the kernels (bf16=True).
    Returns (keep dict of [rows, Ct] tensors, the oracle, the X leaf)."""
    Ct = X_rows.shape[1]
    g = SimpleNamespace(layers=1, tk=tk, T=T, grid=G, Ct=Ct)
    pre = "dist_net.temporal_nets.0."
    params = {pre + "ln.weight": lnw, pre + "ln.bias": lnb, pre + "temporal_net.c_fc1.weight": W1, pre + "temporal_net.c_fc1.bias": b1,
              pre + "temporal_net.c_fc2.weight": W2, pre + "temporal_net.c_fc2.bias": b2}
    o = Oracle(g, params, dtype=torch.float64, bf16=True, fused=fused)
    x = X_rows.double().reshape(clips, T, G * G, Ct).clone()
    if requires_grad:
        x.requires_grad_(True)
        for k in (pre + "ln.weight", pre + "ln.bias"):
            o.p[k].requires_grad_(True)
    keep = {}
    o.temporal_net(x, 0, keep)
    flat = {k.split(".")[0]: v.reshape(-1, Ct) for k, v in keep.items()}
    flat["_raw"] = {k.split(".")[0]: v for k, v in keep.items()}      # the graph's own tensors (for autograd.grad w.r.t. an intermediate)
    return flat, o, x


def ulp_ratio(got, want, rms_floor=2.0 ** -8):
    """element-wise gate beside the range-relative one: max over the elements of |got - want| / (ulp_bf16(|want|) + rms_floor * rms(want)) - a wrong
    SMALL element shows here (the range-relative metric max|d| / max|want| lets it pass); the rms term is the noise floor a sum of bf16-rounded
    terms has whatever its own size."""
    got, want = got.double().cpu().reshape(-1), torch.as_tensor(want).double().cpu().reshape(-1)
    ulp = torch.exp2(torch.floor(torch.log2(want.abs().clamp_min(1e-30))) - 7)
    rms = float(want.pow(2).mean().sqrt())
    return float(((got - want).abs() / (ulp + rms_floor * rms)).max())
