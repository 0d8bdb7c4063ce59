"""Host enqueue time per step vs GPU time per step (is the launch path the bottleneck?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry
g = synth.geometry("b16_8+16f"); b = 32
eng = Engine(config_from_geometry(g, b, torch.bfloat16)); eng.load_state_dict(synth.state_dict(g))
video = torch.from_numpy(synth.video(g, b)).cuda(); text = torch.from_numpy(synth.text_features(g)).cuda()
tgt = torch.from_numpy(synth.soft_target(g, b)[0]).cuda()
pipe = len(sys.argv) < 2 or sys.argv[1] != "serial"
if pipe: eng.vit_forward(video)
def step(parts):
    t = [time.perf_counter()]
    if pipe: eng.vit_prefetch(video)
    else: eng.vit_forward(video)
    t.append(time.perf_counter()); eng.branch_forward(text)
    t.append(time.perf_counter()); _, dl = eng.loss(tgt)
    t.append(time.perf_counter()); eng.backward(dl)
    t.append(time.perf_counter()); eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0)
    if pipe: eng.vit_adopt()
    t.append(time.perf_counter())
    for k in range(5): parts[k] += t[k + 1] - t[k]
for _ in range(5): step([0] * 5)
torch.cuda.synchronize()
parts = [0.0] * 5; n = 20
t0 = time.perf_counter()
for _ in range(n): step(parts)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tg = time.perf_counter() - t0
print(("pipelined" if pipe else "serial"), "host enqueue %.2f ms/step  (vit %.2f | branch fwd %.2f | loss %.2f | backward %.2f | adamw+pack %.2f)   GPU-complete %.2f ms/step" % (th / n * 1e3, *[p / n * 1e3 for p in parts], tg / n * 1e3))
