"""Forward-only iteration (frozen ViT of batch n+1 beside the branch forward of batch n, inference mode) at the bench size: what bench.py's
`forward_only` object times, alone, so that stream-priority / order knobs can be swept for THIS number.
usage: python tools/fwd_only.py [--serial] [--iters N] [--train] [--vit-only] [--branch-only]   -> one line: ms per iteration"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import synth
from dist_amd.engine import Engine, config_from_geometry

serial = "--serial" in sys.argv
iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 30
g = synth.geometry("b16_8+16f"); b = 32
eng = Engine(config_from_geometry(g, b, torch.bfloat16, True, 0))
eng.load_state_dict(synth.state_dict(g))
videos = [torch.from_numpy(synth.video(g, b, seed=1 + 100 * k)).cuda() for k in range(2)]
text = torch.from_numpy(synth.text_features(g)).cuda()
eng.set_inference("--train" not in sys.argv)
vit_only, branch_only = "--vit-only" in sys.argv, "--branch-only" in sys.argv


def fwd(n):
    if vit_only:
        eng.vit_forward(videos[n % 2]); return
    if branch_only:
        eng.branch_forward(text); return
    if serial:
        eng.vit_forward(videos[n % 2])
    else:
        eng.vit_prefetch(videos[(n + 1) % 2])
    eng.branch_forward(text)
    if not serial:
        eng.vit_adopt()


eng.vit_forward(videos[0])
for n in range(5):
    fwd(n)
torch.cuda.synchronize()
t0 = time.perf_counter()
for n in range(iters):
    fwd(n)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / iters * 1e3
gf = 325.73 - 8 * 196 * 768 * 768 * 2 / 1e9
knobs = {k: v for k, v in os.environ.items() if k.startswith("DIST_AMD_") and k != "DIST_AMD_LIB"}
tag = "vit-only" if vit_only else "branch-only" if branch_only else "serial" if serial else "pipelined"
print(f"fwd_only {tag} {knobs}: {ms:.3f} ms/iter" + ("" if (vit_only or branch_only) else f"  path_mfma_frac {b / ms * gf / 2500.0:.4f}"), flush=True)
