"""Wall time of the step phases (HIP events on the caller's stream), ViT-B/16 8+16f b=32 bf16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry
g = synth.geometry("b16_8+16f"); b = 32
eng = Engine(config_from_geometry(g, b, torch.bfloat16)); eng.load_state_dict(synth.state_dict(g))
video = torch.from_numpy(synth.video(g, b)).cuda(); text = torch.from_numpy(synth.text_features(g)).cuda()
tgt = torch.from_numpy(synth.soft_target(g, b)[0]).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
acc = [0.0] * 5
for it in range(13):
    ev[0].record(); eng.vit_forward(video); ev[1].record(); eng.branch_forward(text); ev[2].record()
    _, dl = eng.loss(tgt); ev[3].record(); eng.backward(dl); ev[4].record(); eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0); ev[5].record()
    torch.cuda.synchronize()
    if it >= 3:
        for k in range(5): acc[k] += ev[k].elapsed_time(ev[k + 1]) / 10
print("vit_forward %.2f ms | branch_forward (after vit) %.2f | loss %.2f | backward %.2f | adamw+pack %.2f | total %.2f" % (*acc, sum(acc)))
# isolated: branch forward alone (ViT finished), backward alone
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng.vit_forward(video); torch.cuda.synchronize(); e0.record(); eng.branch_forward(text); e1.record(); torch.cuda.synchronize()
print("branch_forward alone %.2f ms" % e0.elapsed_time(e1))
