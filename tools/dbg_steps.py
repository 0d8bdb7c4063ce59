import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry
def p(*a): print(*a, flush=True)
for dt in (torch.float32, torch.bfloat16):
    g = synth.geometry("tiny"); b = 2
    eng = Engine(config_from_geometry(g, b, dt)); eng.load_state_dict(synth.state_dict(g))
    p(dt, "engine ready")
    video = torch.from_numpy(synth.video(g, b)).cuda(); text = torch.from_numpy(synth.text_features(g)).cuda()
    tgt = torch.from_numpy(synth.soft_target(g, b)[0]).cuda()
    torch.cuda.synchronize(); p("inputs ok")
    eng.vit_forward(video); torch.cuda.synchronize(); p("vit ok")
    eng.branch_forward(text); torch.cuda.synchronize(); p("branch ok")
    loss, dl = eng.loss(tgt); torch.cuda.synchronize(); p("loss ok", float(loss))
    eng.backward(dl); torch.cuda.synchronize(); p("backward ok")
    eng.adamw_step(1e-3, 1e-4); torch.cuda.synchronize(); p("adamw ok")
