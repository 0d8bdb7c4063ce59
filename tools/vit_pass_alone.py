"""N frozen-ViT passes with nothing else on the GPU (for `rocprofv3 --kernel-trace --stats`: the ALONE duration of the dominant GEMM)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
g = synth.geometry("b16_8+16f"); b = 32
eng = Engine(config_from_geometry(g, b, torch.bfloat16)); eng.load_state_dict(synth.state_dict(g))
video = torch.from_numpy(synth.video(g, b)).cuda()
for _ in range(n):
    eng.vit_forward(video)
    torch.cuda.synchronize()
print("done")
