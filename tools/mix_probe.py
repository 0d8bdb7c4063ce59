"""The pipelined train step with the loop's batch-mode mixup in front of every ViT pass: mode none | unfused (dist_op_mixup in place, then the patch gather) |
fused (dist_vit_mix_next: the mix applied while the patch rows are gathered).  usage: python tools/mix_probe.py <mode>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth, ops
from dist_amd.engine import Engine, config_from_geometry
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
g = synth.geometry("b16_8+16f"); b = 32
eng = Engine(config_from_geometry(g, b, torch.bfloat16, True, 0))
eng.load_state_dict(synth.state_dict(g))
videos = [torch.from_numpy(synth.video(g, b, seed=1 + 100 * k)).cuda() for k in range(2)]
text = torch.from_numpy(synth.text_features(g)).cuda()
tgts = [torch.from_numpy(synth.soft_target(g, b, seed=3 + 100 * k)[0]).cuda() for k in range(2)]
it = [0]
def step():
    n = it[0]; it[0] += 1
    v = videos[(n + 1) % 2]
    if mode == "unfused": ops.mixup_(v, 0.6180339)
    elif mode == "fused": eng.vit_mix_next("mixup", 0.6180339)
    eng.vit_prefetch(v)
    eng.branch_forward(text)
    _, dl = eng.loss(tgts[n % 2])
    eng.backward(dl)
    eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0)
    eng.vit_adopt()
eng.vit_forward(videos[0])
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize()
print(f"mix_probe {mode}: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms/step", flush=True)
