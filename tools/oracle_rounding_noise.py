"""How far do two evaluations of the SAME bf16-rounded graph drift apart when only the arithmetic BETWEEN the rounding points differs?
Oracle(bf16=True, fused=True) in fp32 against fp64 on ViT-B/16 8+16f, b = 2 (CPU, a few minutes): the noise floor any bf16 implementation
has against any other - the engine's gap to this oracle (profiles/r04_parity_gaps.json same_rounding.*) is to be read against it."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from dist_amd import synth
from dist_oracle import Oracle
torch.set_num_threads(8)
g = synth.geometry("b16_8+16f"); sd = synth.state_dict(g)
v, t, y = synth.video(g, 2), synth.text_features(g), synth.soft_target(g, 2)[0]
t0 = time.time()
a = Oracle(g, sd, dtype=torch.float32, bf16=True, fused=True).forward_backward(v, t, y)
b = Oracle(g, sd, dtype=torch.float64, bf16=True, fused=True).forward_backward(v, t, y)
out = {"logits_maxabs": float((a["logits"].double() - b["logits"]).abs().max()), "loss_abs": abs(float(a["loss"]) - float(b["loss"]))}
rel = lambda p, q: float((p.double() - q.double()).norm() / (q.double().norm() + 1e-30))
out["branch_act_worst"] = max(rel(a["keep"][k].detach(), b["keep"][k].detach()) for k in a["keep"] if k.split(".")[0] in ("tn_out", "int_out", "x_temporal"))
errs = sorted(((float((a["grads"][k].double() - b["grads"][k]).abs().max() / (b["grads"][k].abs().max() + 1e-12)), k) for k in b["grads"] if b["grads"][k].abs().max() >= 1e-6), reverse=True)
out["grad_worst_relmax"] = errs[0][0]; out["grad_worst_tensors"] = [[k, e] for e, k in errs[:6]]; out["grad_median_relmax"] = errs[len(errs) // 2][0]
out["seconds"] = round(time.time() - t0, 1)
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_oracle_rounding_noise.json"), "w"), indent=1)
