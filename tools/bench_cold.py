"""Cold-cache micro-benchmarks: every call works on a different buffer set (> 256 MB MALL in total),
which is how the kernels run inside a train step (5 GB of saved activations stream through HBM)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L

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
NSET = 12
def mk(*shape): return [torch.randn(*shape, device="cuda").to(dt) for _ in range(NSET)]

print("== TN (dW) cold ==")
for (M, NI, K, taps, mode, kw, tag) in [(100352, 96, 96, 9, L.RM_SPATIAL, (14, 0), "tn_fc2"), (100352, 96, 96, 3, L.RM_SHIFT, (16*196, 196), "tn_fc1"),
                                        (50432, 384, 768, 1, 0, (0, 0), "dWi"), (50432, 384, 384, 1, 0, (0, 0), "dWffn"), (50432, 384, 96, 1, 0, (0,0), "dW3"),
                                        (50432, 96, 96, 3, L.RM_SHIFT, (8*197, 197), "tf_fc2")]:
    As, Bs = mk(M, NI), mk(M, K)
    out = torch.zeros(NI, K * taps, device="cuda"); cs = torch.zeros(NI, device="cuda"); part = torch.empty(9 << 20, device="cuda")
    fns = [(lambda a=a, b=b: ops.gemm_tn(a, b, out, M, NI, K, taps=taps, bmap=ops.rowmap(mode, kw[0], kw[1]), colsum=cs, partial=part)) for a, b in zip(As, Bs)]
    t = timeit_rot(fns)
    byts = (M * NI + M * K) * 2
    print(f"gemm_tn {tag:8s} M={M} NI={NI} K={K} taps={taps}: {t*1e6:8.1f} us  {2*M*NI*K*taps/t/1e12:7.1f} TF  (min HBM {byts/t/1e9:6.0f} GB/s)", flush=True)
    del As, Bs

print("== NT cold ==")
for (M, N, K, taps, mode, kw, tag, extra) in [(100352, 96, 96, 9, L.RM_SPATIAL, (14, 0), "conv3x3", "res+act"), (100352, 96, 96, 3, L.RM_SHIFT, (16*196, 196), "conv_t", "act"),
                                               (50432, 384, 768, 1, 0, (0, 0), "in_lin", "res"), (50432, 384, 384, 1, 0, (0, 0), "ffn_fc", "act"),
                                               (50432, 96, 384, 1, 0, (0, 0), "tf_fc1", ""), (50432, 384, 96, 1, 0, (0, 0), "tf_proj", "res"),
                                               (50432, 768, 768, 1, 0, (0, 0), "vit_out", "res"), (50432, 2304, 768, 1, 0, (0, 0), "vit_qkv", ""),
                                               (50432, 3072, 768, 1, 0, (0, 0), "vit_fc", "act"), (50432, 768, 3072, 1, 0, (0, 0), "vit_proj", "res")]:
    As = mk(M, K); W = (torch.randn(N, taps * K, device="cuda") * (taps * K) ** -0.5).to(dt)
    Cs, C2s, Rs = mk(M, N), mk(M, N) if "act" in extra else [None] * NSET, mk(M, N) if "res" in extra else [None] * NSET
    bias = torch.randn(N, device="cuda")
    fns = [(lambda a=a, c=c, c2=c2, r=r: ops.gemm_nt(a, W, M, N, K, taps=taps, bias=bias, res=r, C_out=c, C2_out=c2, amap=ops.rowmap(mode, kw[0], kw[1])))
           for a, c, c2, r in zip(As, Cs, C2s, Rs)]
    t = timeit_rot(fns)
    byts = (M * K + M * N * (1 + ("act" in extra) + ("res" in extra))) * 2
    print(f"gemm_nt {tag:8s} M={M} N={N} K={K} taps={taps} {extra:8s}: {t*1e6:8.1f} us  {2*M*N*K*taps/t/1e12:7.1f} TF  (min HBM {byts/t/1e9:6.0f} GB/s)", flush=True)
    del As, Cs, C2s, Rs

print("== LN / attention cold ==")
xs = mk(50432, 768); ys = mk(50432, 768); w = torch.ones(768, device="cuda"); b = torch.zeros(768, device="cuda")
t = timeit_rot([(lambda x=x, y=y: ops.layernorm(x, w, b, y=y)) for x, y in zip(xs, ys)])
print(f"layernorm 50432x768: {t*1e6:8.1f} us {2*xs[0].numel()*2/t/1e9:7.0f} GB/s")
del xs, ys
qs = mk(256 * 197, 2304)
t = timeit_rot([(lambda q=q: ops.attention(q, 256, 197, 12)) for q in qs])
print(f"attention (rows layout): {t*1e6:8.1f} us {4*256*12*197*197*64/t/1e12:7.1f} TF")
t = timeit_rot([(lambda q=q: ops.attention(q, 256, 197, 12, layout=L.QKV_HEADS)) for q in qs])
print(f"attention (head-major):  {t*1e6:8.1f} us {4*256*12*197*197*64/t/1e12:7.1f} TF")
