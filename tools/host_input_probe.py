"""What does feeding the step from host memory cost, and why?  The pipelined train step of bench.py with the batches resident, plus per step:
  mode none   nothing (baseline)            mode tiny  a 4 KB H2D copy on a fifth stream          mode full  the 308 MB copy on a fifth stream, result unused
  mode main   the 308 MB copy on the step's own stream (serial)            mode ev    `full` + the wait / release events of dist_amd/utils/staging.py
usage: python tools/host_input_probe.py <mode>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry
mode = sys.argv[1] if len(sys.argv) > 1 else "none"
g = synth.geometry("b16_8+16f"); b = 32
eng = Engine(config_from_geometry(g, b, torch.bfloat16, True, 0))
eng.load_state_dict(synth.state_dict(g))
videos = [torch.from_numpy(synth.video(g, b, seed=1 + 100 * k)).cuda() for k in range(2)]
text = torch.from_numpy(synth.text_features(g)).cuda()
tgts = [torch.from_numpy(synth.soft_target(g, b, seed=3 + 100 * k)[0]).cuda() for k in range(2)]
host = videos[0].cpu().pin_memory()
dev = torch.empty_like(videos[0])
cs = torch.cuda.Stream()
it = [0]
stager, tk = None, {}
if mode.startswith("stager"):
    from dist_amd.utils.staging import HostStager
    stager = HostStager(depth=3, host_ordered=os.environ.get("HOST_ORDERED", "1") == "1")
    hosts = [v.cpu().pin_memory() for v in videos]
    acc = {"submit": 0.0, "wait": 0.0, "release": 0.0}
    def timed(name, fn):
        def w(*a, **k):
            t = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t; return r
        return w
    stager.submit, stager.wait, stager.release = timed("submit", stager.submit), timed("wait", stager.wait), timed("release", stager.release)
def step():
    n = it[0]; it[0] += 1
    if stager is not None:
        # stager0: submit only; stager1: + wait / release, the ViT still reads the resident tensors; stager2: the ViT reads the staged copies (= bench.py --host-input)
        pos, pin = os.environ.get("POS", "mid"), (True if os.environ.get("PINNED") == "1" else None)
        if n + 1 not in tk: tk[n + 1] = stager.submit(hosts[(n + 1) % 2], pin)
        if pos == "start": tk[n + 2] = stager.submit(hosts[(n + 2) % 2], pin)
        v = videos[(n + 1) % 2]
        if mode != "stager0":
            w = stager.wait(tk[n + 1])
            if mode == "stager2": v = w
        eng.vit_prefetch(v)
        if pos == "mid": tk[n + 2] = stager.submit(hosts[(n + 2) % 2], pin)
        eng.branch_forward(text)
        if mode != "stager0" and n in tk: stager.release(tk.pop(n))
        _, dl = eng.loss(tgts[n % 2]); eng.backward(dl); eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0)
        if pos == "end": tk[n + 2] = stager.submit(hosts[(n + 2) % 2], pin)
        eng.vit_adopt()
        return
    if mode in ("tiny", "full", "ev"):
        with torch.cuda.stream(cs):
            if mode == "tiny": dev.view(-1)[:1024].copy_(host.view(-1)[:1024], non_blocking=True)
            else: dev.copy_(host, non_blocking=True)
            if mode == "ev": ev = torch.cuda.Event(); ev.record(cs)
        if mode == "ev": torch.cuda.current_stream().wait_event(ev)
    elif mode == "main":
        dev.copy_(host, non_blocking=True)
    eng.vit_prefetch(videos[(n + 1) % 2])
    eng.branch_forward(text)
    _, dl = eng.loss(tgts[n % 2])
    eng.backward(dl)
    eng.adamw_step(3.2e-5, 1e-4, lr_mult=10.0)
    eng.vit_adopt()
eng.vit_forward(videos[0])
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print(f"host_input_probe {mode}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/step" + (f"  host time per step: " + ", ".join(f"{k} {v / 25 * 1e3:.3f} ms" for k, v in acc.items()) if stager is not None else ""), flush=True)
