"""Frozen-ViT pass alone (no branch, no other stream): bf16 vs the fp8 modes.  usage: python tools/bench_vit_pass.py [config] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry
gname = sys.argv[1] if len(sys.argv) > 1 else "l14_32+64f"
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = synth.geometry(gname)
sd = synth.state_dict(g)
video = torch.from_numpy(synth.video(g, b)).cuda()
for mask in (0, 5, 15, 31):
    eng = Engine(config_from_geometry(g, b, torch.bfloat16, True, mask))
    eng.load_state_dict(sd)
    for _ in range(3): eng.vit_forward(video)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): eng.vit_forward(video)
    torch.cuda.synchronize()
    print(f"{gname} b={b} vit_fp8={mask:2d}: {(time.perf_counter()-t0)/5*1e3:7.2f} ms per frozen-ViT pass", flush=True)
    del eng
