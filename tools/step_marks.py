"""Schedule of the pipelined step WITHOUT a profiler: device-side marks (dist_marks_*) of steady-state steps.
usage: python tools/step_marks.py [reps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry
g = synth.geometry("b16_8+16f"); b = 32
eng = Engine(config_from_geometry(g, b, torch.bfloat16)); eng.load_state_dict(synth.state_dict(g))
videos = [torch.from_numpy(synth.video(g, b, seed=1 + 100 * k)).cuda() for k in range(2)]
text = torch.from_numpy(synth.text_features(g)).cuda(); tgt = torch.from_numpy(synth.soft_target(g, b)[0]).cuda()
n = [0]
def step():
    eng.vit_prefetch(videos[(n[0] + 1) % 2]); n[0] += 1
    eng.branch_forward(text); _, dl = eng.loss(tgt); eng.backward(dl)
    eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0); eng.vit_adopt()
eng.vit_forward(videos[0])
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("un-instrumented: %.2f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
eng.marks_enable(True)
acc = {}
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for r in range(reps):
    for _ in range(6): step()            # run ahead a few steps, then read the marks of the last one
    m = eng.marks_read()
    for k, v in m.items(): acc.setdefault(k, []).append(v)
print("marks of the last of 6 queued steps, ms since that step's vit_begin (median of %d):" % reps)
for k in eng.MARKS:
    v = sorted(acc[k]); print(f"   {k:10s} {v[len(v)//2]:8.2f}   (min {v[0]:.2f} max {v[-1]:.2f})")
