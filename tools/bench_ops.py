"""Micro-benchmarks of the hot kernels at BASELINE config-2 shapes (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dist_amd import ops, lib as L

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3

def main():
    dt = torch.bfloat16
    dev = "cuda"
    print("device", torch.cuda.get_device_name(0))
    for (M, N, K, tag) in [(50432, 2304, 768, "qkv"), (50432, 768, 768, "out"), (50432, 3072, 768, "fc"), (50432, 768, 3072, "proj"),
                           (50432, 384, 768, "input_linear"), (50432, 384, 384, "ffn"), (100352, 96, 96, "n96")]:
        A = torch.randn(M, K, device=dev).to(dt); B = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        C = torch.empty(M, N, device=dev, dtype=dt)
        bias = torch.randn(N, device=dev)
        t = timeit(lambda: ops.gemm_nt(A, B, M, N, K, bias=bias, C_out=C))
        t2 = timeit(lambda: torch.nn.functional.linear(A, B))
        print(f"gemm_nt {tag:14s} M={M} N={N} K={K}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF   (hipBLASLt via torch: {t2*1e6:8.1f} us {2*M*N*K/t2/1e12:7.1f} TF)")
    # conv taps
    M, Ct = 100352, 96
    A = torch.randn(M, Ct, device=dev).to(dt)
    for taps, mode, kw in [(3, L.RM_SHIFT, (16 * 196, 196)), (9, L.RM_SPATIAL, (14, 0))]:
        B = torch.randn(Ct, taps * Ct, device=dev).to(dt)
        C = torch.empty(M, Ct, device=dev, dtype=dt)
        t = timeit(lambda: ops.gemm_nt(A, B, M, Ct, Ct, taps=taps, amap=ops.rowmap(mode, kw[0], kw[1]), C_out=C))
        print(f"conv taps={taps}: {t*1e6:8.1f} us  {2*M*Ct*Ct*taps/t/1e12:7.1f} TF")
    # dW
    for (M, NI, K, tag) in [(50432, 384, 768, "dWi"), (50432, 384, 384, "dWffn"), (100352, 96, 96, "dW96")]:
        A = torch.randn(M, NI, device=dev).to(dt); B = torch.randn(M, K, device=dev).to(dt)
        out = torch.zeros(NI, K, device=dev)
        for tr in (0, 1):
            t = timeit(lambda: ops.gemm_tn(A, B, out, M, NI, K, use_tr=tr))
            print(f"gemm_tn {tag:8s} tr={tr} M={M} NI={NI} K={K}: {t*1e6:8.1f} us  {2*M*NI*K/t/1e12:7.1f} TF")
    # attention
    frames, Ltok, heads = 256, 197, 12
    qkv = torch.randn(frames * Ltok, 3 * 768, device=dev).to(dt)
    t = timeit(lambda: ops.attention(qkv, frames, Ltok, heads))
    fl = 4 * frames * heads * Ltok * Ltok * 64
    print(f"attention frames={frames}: {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF")
    # layernorm
    x = torch.randn(50432, 768, device=dev).to(dt); w = torch.ones(768, device=dev); b = torch.zeros(768, device=dev)
    y = torch.empty_like(x)
    t = timeit(lambda: ops.layernorm(x, w, b, y=y))
    print(f"layernorm 50432x768: {t*1e6:8.1f} us {2*x.numel()*2/t/1e9:7.1f} GB/s")

if __name__ == "__main__":
    main()
