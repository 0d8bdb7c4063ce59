"""ln_bwd: cost of the parameter-gradient tail (with / without dw, db), dual-input variant included."""
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
for rows, C in [(50432, 384), (100352, 96)]:
    xs = [torch.randn(rows, C, device="cuda").to(dt) for _ in range(8)]; ys = [torch.randn_like(x) for x in xs]; y2 = [torch.randn_like(x) for x in xs]
    outs = [torch.empty_like(x) for x in xs]; cps = [torch.empty_like(x) for x in xs]
    w = torch.ones(C, device="cuda"); w2 = torch.ones(C, device="cuda")
    mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda")
    dw = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); dw2 = torch.zeros(C, device="cuda"); db2 = torch.zeros(C, device="cuda")
    t = timeit_rot([(lambda x=x, y=y, o=o: ops.layernorm_bwd(x, mean, rstd, y, w, dx=o)) for x, y, o in zip(xs, ys, outs)])
    print(f"{rows}x{C} single, no dw/db : {t*1e6:6.1f} us {3*rows*C*2/t/1e9:6.0f} GB/s")
    t = timeit_rot([(lambda x=x, y=y, o=o: ops.layernorm_bwd(x, mean, rstd, y, w, dx=o, dw=dw, db=db)) for x, y, o in zip(xs, ys, outs)])
    print(f"{rows}x{C} single, dw/db    : {t*1e6:6.1f} us {3*rows*C*2/t/1e9:6.0f} GB/s")
    t = timeit_rot([(lambda x=x, y=y, z=z, o=o, c=c: ops.layernorm_bwd(x, mean, rstd, y, w, dy2=z, w2=w2, dx=o, dx_copy=c)) for x, y, z, o, c in zip(xs, ys, y2, outs, cps)])
    print(f"{rows}x{C} dual+copy, no dw : {t*1e6:6.1f} us {5*rows*C*2/t/1e9:6.0f} GB/s")
    t = timeit_rot([(lambda x=x, y=y, z=z, o=o, c=c: ops.layernorm_bwd(x, mean, rstd, y, w, dy2=z, w2=w2, dx=o, dx_copy=c, dw=dw, db=db, dw2=dw2, db2=db2)) for x, y, z, o, c in zip(xs, ys, y2, outs, cps)])
    print(f"{rows}x{C} dual+copy, 4 grads: {t*1e6:6.1f} us {5*rows*C*2/t/1e9:6.0f} GB/s")
