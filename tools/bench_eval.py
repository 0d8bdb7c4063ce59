"""Multi-view evaluation loop at the bench workload (ViT-B/16 8+16f, b = 32 per GPU, bf16, synthetic clips resident in HBM):
forward only (frozen ViT of batch n+1 beside the branch forward of batch n), eval softmax, score ensemble.
Three variants of what happens to the predictions of every iteration:
  device : dist_op_softmax_rows + dist_op_ensemble_update (dist_amd.utils.meters.TestMeter) - no host synchronisation
  host   : what the reference's loop does (runs/test.py:133-150, utils/meters.py:82-112): preds / labels / ids to the host, a Python
           loop over the clips (restated inline here for timing only)
  none   : forward only
usage: python tools/bench_eval.py [iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth, ops
from dist_amd.engine import Engine, config_from_geometry
from dist_amd.utils.meters import TestMeter

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = synth.geometry("b16_8+16f")
b, views = 32, 30
eng = Engine(config_from_geometry(g, b, torch.bfloat16))
eng.load_state_dict(synth.state_dict(g))
videos = [torch.from_numpy(synth.video(g, b, seed=1 + 100 * k)).cuda() for k in range(2)]
text = torch.from_numpy(synth.text_features(g)).cuda()
K = eng.cfg.num_classes
V = (iters + 8) * b // views + 1


def run(kind, pipelined=True):
    meter = TestMeter(None, V, views, K, iters) if kind == "device" else None
    hp, hl, hc = torch.zeros(V, K), torch.zeros(V, dtype=torch.long), torch.zeros(V, dtype=torch.long)
    if pipelined:
        eng.vit_forward(videos[0])
    def it(n):
        if pipelined:
            eng.vit_prefetch(videos[(n + 1) % 2])
        else:
            eng.vit_forward(videos[n % 2])
        logits, _ = eng.branch_forward(text)
        ids = n * b + torch.arange(b, device="cuda")
        labels = ((ids // views) * 7919) % K
        if kind == "device":
            meter.update_stats(ops.softmax_rows(logits.view(b, -1)), labels, ids)
        elif kind == "host":
            p, l, c = torch.softmax(logits.view(b, -1).float(), dim=-1).cpu(), labels.cpu(), ids.cpu()
            for i in range(b):
                vid = int(c[i]) // views
                hl[vid] = l[i]; hp[vid] += p[i]; hc[vid] += 1
        if pipelined:
            eng.vit_adopt()
    for n in range(4):
        it(n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for n in range(4, 4 + iters):
        it(n)
    if meter is not None:
        meter.finalize_metrics()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return dt


for kind in ("none", "device", "host", "device", "host", "none"):
    dt = run(kind)
    print(f"eval loop, predictions -> {kind:6s}: {dt*1e3:7.2f} ms / iteration  {b/dt:8.1f} clips/s", flush=True)
dt = run("device", pipelined=False)
print(f"eval loop, serial order, device meter: {dt*1e3:7.2f} ms / iteration  {b/dt:8.1f} clips/s")
# the meter update alone (8 ranks x 32 clips gathered, K = 174 / 400)
for n, Kk in ((32, 174), (256, 174), (256, 400)):
    m = TestMeter(None, 4096, views, Kk, 1)
    p = torch.rand(n, Kk, device="cuda"); ids = torch.arange(n, device="cuda"); lab = ((ids // views) * 7919) % Kk
    for _ in range(3): m.update_stats(p, lab, ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): m.update_stats(p, lab, ids)
    torch.cuda.synchronize()
    print(f"dist_op_ensemble_update n={n} K={Kk}: {(time.perf_counter()-t0)/50*1e6:7.1f} us")
    pc, lc, ic = p.cpu(), lab.cpu(), ids.cpu(); hp = torch.zeros(4096, Kk); hl = torch.zeros(4096, dtype=torch.long)
    t0 = time.perf_counter()
    for _ in range(5):
        for i in range(n):
            vid = int(ic[i]) // views; hl[vid] = lc[i]; hp[vid] += pc[i]
    print(f"   the reference's host loop over the same batch: {(time.perf_counter()-t0)/5*1e6:7.1f} us (+ the device->host copies and their synchronisation)")
