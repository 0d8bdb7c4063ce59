"""Host -> device copy of one 32-clip fp32 batch (308 MB) from pinned memory: idle GPU, and beside a running GEMM loop on another stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
n = 32 * 3 * 16 * 224 * 224
h = torch.empty(n, dtype=torch.float32).pin_memory()
d = torch.empty(n, dtype=torch.float32, device="cuda")
s = torch.cuda.Stream()
def copy_ms(reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(reps):
            d.copy_(h, non_blocking=True)
    s.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
copy_ms(2)
ms = copy_ms()
print(f"H2D 308 MB pinned, idle GPU: {ms:.2f} ms = {n * 4 / ms / 1e6:.1f} GB/s", flush=True)
a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
for _ in range(3): (a @ b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): c = a @ b
torch.cuda.synchronize()
base = (time.perf_counter() - t0) / 20 * 1e3
t0 = time.perf_counter()
with torch.cuda.stream(s):
    for _ in range(3): d.copy_(h, non_blocking=True)
for _ in range(20): c = a @ b
torch.cuda.synchronize()
both = (time.perf_counter() - t0) * 1e3
print(f"20 x 8192^3 bf16 matmul alone {base * 20:.1f} ms; with 3 H2D copies beside them {both:.1f} ms", flush=True)
ph = torch.empty(n, dtype=torch.float32)
t0 = time.perf_counter(); d.copy_(ph); torch.cuda.synchronize(); print(f"H2D 308 MB PAGEABLE: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
