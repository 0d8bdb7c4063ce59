import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
def timeit_rot(fns, reps=3):
    for f in fns: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        for f in fns: f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * len(fns)) * 1e-3
dt = torch.bfloat16
for rows, C in [(50432, 768), (50432, 384), (100352, 96)]:
    xs = [torch.randn(rows, C, device="cuda").to(dt) for _ in range(12)]; ys = [torch.empty_like(x) for x in xs]
    w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    mean = torch.empty(rows, device="cuda"); rstd = torch.empty(rows, device="cuda")
    t = timeit_rot([(lambda x=x, y=y: ops.layernorm(x, w, b, y=y, mean=mean, rstd=rstd)) for x, y in zip(xs, ys)])
    print(f"ln_fwd {rows}x{C}: {t*1e6:7.1f} us {2*rows*C*2/t/1e9:7.0f} GB/s")
    dw = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    t = timeit_rot([(lambda x=x, y=y: ops.layernorm_bwd(x, mean, rstd, y, w, dx=y, dw=dw, db=db)) for x, y in zip(xs, ys)])
    print(f"ln_bwd {rows}x{C}: {t*1e6:7.1f} us {3*rows*C*2/t/1e9:7.0f} GB/s")
