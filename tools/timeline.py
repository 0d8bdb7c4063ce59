"""Concurrency analysis of a rocprofv3 kernel trace (results DB): how the step's wall time splits into
idle / one kernel / several kernels in flight, per-queue busy time, and which kernels run alone."""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = next((c for c in ("queue_id", "stream_id", "queue") if c in cols), None)
rows = list(cur.execute(f"select name, start, end, {qcol or '0'} from kernels order by start"))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:60]
# steady-state window: between the adamw kernels number lo and hi
ad = [r for r in rows if "adamw" in r[0]]
lo, hi = int(sys.argv[2]) if len(sys.argv) > 2 else 6, int(sys.argv[3]) if len(sys.argv) > 3 else 12
t0, t1 = ad[lo][2], ad[hi][2]
nsteps = hi - lo
win = [r for r in rows if r[1] >= t0 and r[2] <= t1]
print(f"window: {nsteps} steps, {(t1-t0)/1e6/nsteps:.3f} ms/step, {len(win)/nsteps:.0f} launches/step")
ev = []
for i, r in enumerate(win):
    ev.append((r[1], 1, i)); ev.append((r[2], -1, i))
ev.sort()
hist = defaultdict(float); alone = defaultdict(float); active = set(); last = t0
for t, d, i in ev:
    dt = t - last
    hist[min(len(active), 6)] += dt
    if len(active) == 1:
        alone[short(win[next(iter(active))][0])] += dt
    last = t
    if d > 0: active.add(i)
    else: active.discard(i)
hist[0] += t1 - last
print("kernels in flight -> ms/step:", {k: round(v/1e6/nsteps, 3) for k, v in sorted(hist.items())})
print("alone time by kernel (ms/step):")
for k, v in sorted(alone.items(), key=lambda x: -x[1])[:12]:
    print(f"   {v/1e6/nsteps:7.3f}  {k}")
busy = defaultdict(float); cnt = defaultdict(int)
for r in win:
    busy[r[3]] += r[2] - r[1]; cnt[r[3]] += 1
print("per-queue busy ms/step (sum of kernel durations):")
for q, v in sorted(busy.items(), key=lambda x: -x[1]):
    names = defaultdict(float)
    for r in win:
        if r[3] == q: names[short(r[0])] += r[2] - r[1]
    top = ", ".join(f"{n.split('<')[0][:28]} {t/1e6/nsteps:.2f}" for n, t in sorted(names.items(), key=lambda x: -x[1])[:3])
    print(f"   q{q}: {v/1e6/nsteps:7.3f} ms  {cnt[q]/nsteps:5.0f} launches   [{top}]")
# per step: end of the last ViT GEMM vs end of adamw
print("per kernel-name totals in window (ms/step):")
tot = defaultdict(float)
for r in win: tot[short(r[0])] += r[2] - r[1]
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:14]:
    print(f"   {v/1e6/nsteps:7.3f}  {k}")
