"""Weight re-pack of the trainable tensors (dist_pack_weights what = 2, once per optimizer step) and of everything (what = 3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry
g = synth.geometry("b16_8+16f")
eng = Engine(config_from_geometry(g, 32, torch.bfloat16))
eng.load_state_dict(synth.state_dict(g))
for what in (2, 3):
    for _ in range(3): eng.pack(what)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.pack(what)
    torch.cuda.synchronize()
    print(f"dist_pack_weights(what={what}): {(time.perf_counter()-t0)/20*1e6:7.1f} us")
