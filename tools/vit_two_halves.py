"""Does the frozen-ViT pass lose time to the last, partly filled round of its N = 768 GEMMs (591 tiles on 256 CUs = 2.31 rounds)?  Two half batches
(16 clips each, own engines / workspaces) on TWO streams fill each other's tails; one 32-clip pass on one stream cannot.
usage: python tools/vit_two_halves.py   -> ms per 32 clips, both ways"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry

g = synth.geometry("b16_8+16f")
sd = synth.state_dict(g)


def mk(b):
    e = Engine(config_from_geometry(g, b, torch.bfloat16, True, 0))
    e.load_state_dict(sd)
    return e


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


full = mk(32)
v32 = torch.from_numpy(synth.video(g, 32, seed=1)).cuda()
t_full = timeit(lambda: full.vit_forward(v32))
print(f"one 32-clip pass, one stream: {t_full:.3f} ms", flush=True)
del full
torch.cuda.empty_cache()
h = [mk(16), mk(16)]
vs = [v32[:16].contiguous(), v32[16:].contiguous()]
prio = int(os.environ.get("HALF_PRIO", "0"))
ss = [torch.cuda.Stream(priority=prio), torch.cuda.Stream(priority=prio)]


def both():
    for e, v, s in zip(h, vs, ss):
        with torch.cuda.stream(s):
            e.vit_forward(v)


t_two = timeit(both)
print(f"two 16-clip passes, two streams: {t_two:.3f} ms per 32 clips  ({t_full / t_two:.3f} x)", flush=True)
t_seq = timeit(lambda: (h[0].vit_forward(vs[0]), h[1].vit_forward(vs[1])))
print(f"two 16-clip passes, one stream:  {t_seq:.3f} ms per 32 clips", flush=True)
