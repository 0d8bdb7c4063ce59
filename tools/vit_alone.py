"""The frozen-ViT pass alone on the handle's prefetch stream (optionally CU-masked: DIST_AMD_PF_CUMASK), and the chain alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry
g = synth.geometry("b16_8+16f"); b = 32
torch.cuda.set_stream(torch.cuda.Stream())
eng = Engine(config_from_geometry(g, b, torch.bfloat16)); eng.load_state_dict(synth.state_dict(g))
video = torch.from_numpy(synth.video(g, b)).cuda(); text = torch.from_numpy(synth.text_features(g)).cuda()
tgt = torch.from_numpy(synth.soft_target(g, b)[0]).cuda()
eng.vit_forward(video)
def vit():
    eng.vit_prefetch(video); eng.vit_adopt()
def chain():
    eng.branch_forward(text); _, dl = eng.loss(tgt); eng.backward(dl); eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0)
def both():
    eng.vit_prefetch(video); chain(); eng.vit_adopt()
for name, fn in (("vit alone (prefetch stream)", vit), ("chain alone", chain), ("both", both)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(15): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 15 * 1e3:.2f} ms")
