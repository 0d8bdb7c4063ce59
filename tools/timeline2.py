"""Per-queue activity of ONE step (0.5 ms bins) from a rocprofv3 kernel-trace DB, plus per-queue gap statistics."""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end, queue_id from kernels order by start"))
ad = [r for r in rows if "adamw" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
t0, t1 = ad[k][2], ad[k + 1][2]
win = [r for r in rows if r[2] > t0 and r[1] < t1]
qs = sorted({r[3] for r in win})
binw = 0.5e6
nb = int((t1 - t0) / binw) + 1
def tag(n):
    for key, c in (("gemm_fast", "F"), ("gemm_nt", "n"), ("gemm_tn", "t"), ("tn_reduce", "r"), ("attn", "A"), ("ln_fwd", "l"), ("ln_bwd", "b"), ("gemm_small", "s"), ("adamw", "W"), ("pack", "P"), ("patchify", "p")):
        if key in n: return c
    return "."
print(f"step {k}: {(t1-t0)/1e6:.2f} ms; one column = 0.5 ms; digit = busy tenths of the bin, letter row = dominant kernel")
for q in qs:
    busy = [0.0] * nb; dom = [defaultdict(float) for _ in range(nb)]
    for r in win:
        if r[3] != q: continue
        s, e = max(r[1], t0), min(r[2], t1)
        b = int((s - t0) / binw)
        while s < e:
            be = t0 + (b + 1) * binw
            d = min(e, be) - s
            busy[b] += d; dom[b][tag(r[0])] += d
            s = min(e, be); b += 1
    print(f"q{q} busy ", "".join(str(min(9, int(x / binw * 10))) for x in busy))
    print(f"q{q} kern ", "".join((max(d.items(), key=lambda x: x[1])[0] if d else " ") for d in dom))
    ks = sorted([r for r in win if r[3] == q], key=lambda r: r[1])
    gaps = [max(0, ks[i + 1][1] - ks[i][2]) for i in range(len(ks) - 1)]
    if gaps:
        big = sorted(gaps)[-5:]
        print(f"      {len(ks)} launches, busy {sum(r[2]-r[1] for r in ks)/1e6:.2f} ms, gaps: total {sum(gaps)/1e6:.2f} ms, median {sorted(gaps)[len(gaps)//2]/1e3:.1f} us, >20us: {sum(1 for g in gaps if g > 20e3)}, largest {[round(g/1e3) for g in big]} us")
