"""Batch-mode Mixup / CutMix kernels at the BASELINE batch (32 clips x 3 x 16 x 224 x 224 fp32 = 308 MB): HBM-bound streaming.
Algorithmic bytes: mixup reads and writes every element once (2 x 308 MB); cutmix reads and writes the box of every plane;
the reference's torch ops (flip copy, mul_, mul_, add_) next to it on the same GPU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops
def timeit(f, reps=10):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): f()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps * 1e-3)
    return best
x = torch.randn(32, 3, 16, 224, 224, device="cuda"); nbytes = x.numel() * 4
t = timeit(lambda: ops.mixup_(x, 0.5))
print(f"dist_op_mixup   b=32: {t*1e6:7.1f} us  {2*nbytes/t/1e9:7.0f} GB/s of 8000 peak ({2*nbytes/t/8e12:.2f}; float4 copy reaches 6290)")
def ref_mix():
    xf = x.flip(0).mul_(0.5); x.mul_(0.5).add_(xf)
t2 = timeit(ref_mix)
print(f"reference torch ops (flip, mul_, mul_, add_) on this GPU: {t2*1e6:7.1f} us  ({t2/t:.1f}x)")
yl, yh, xl, xh = 40, 180, 30, 190
box = 32 * 3 * 16 * (yh - yl) * (xh - xl) * 4
t = timeit(lambda: ops.cutmix_(x, yl, yh, xl, xh))
print(f"dist_op_cutmix  box {yh-yl}x{xh-xl}: {t*1e6:7.1f} us  {2*box/t/1e9:7.0f} GB/s")
def ref_cut():
    x[:, :, :, yl:yh, xl:xh] = x.flip(0)[:, :, :, yl:yh, xl:xh]
t2 = timeit(ref_cut)
print(f"reference torch ops (flip copy + slice assignment): {t2*1e6:7.1f} us  ({t2/t:.1f}x)")
lab = torch.randint(0, 174, (32,), device="cuda")
t = timeit(lambda: ops.mixup_target(lab, 174, 0.5, 0.1))
print(f"dist_op_mixup_target 32 x 174: {t*1e6:7.1f} us (launch-bound)")
