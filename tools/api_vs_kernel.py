"""Where does the host stand relative to the GPU?  rocprofv3 --hip-trace --kernel-trace DB: for one steady-state step, the host
time of every HIP API call that takes > 50 us, and for the step's first kernels per queue: host launch time vs device start."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print("tables/views:", [t for t in tabs if not t.startswith("rocpd_")][:40])
kern = list(cur.execute("select name, start, end, queue_id from kernels order by start"))
ad = [r for r in kern if "adamw" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
t0, t1 = ad[k][2], ad[k + 1][2]
print(f"step {k}: device {(t1-t0)/1e6:.2f} ms")
# API regions
cols = [r[1] for r in cur.execute("pragma table_info(regions)")]
print("regions cols:", cols)
api = list(cur.execute("select name, start, end from regions order by start"))
win = [a for a in api if a[1] >= t0 - 25e6 and a[1] <= t1]
print(f"{len(win)} API calls between step start - 25 ms and step end")
print("calls longer than 100 us (start relative to the device-side step start):")
for a in win:
    if a[2] - a[1] > 100e3: print(f"   {(a[1]-t0)/1e3:10.1f} us  +{(a[2]-a[1])/1e3:9.1f} us  {a[0]}")
# launches: pair the n-th hipLaunchKernel-like call with the n-th kernel is unreliable across streams; instead print the host
# time of launch calls around the device step start
la = [a for a in api if "Launch" in a[0]]
import bisect
starts = [a[1] for a in la]
i0 = bisect.bisect_left(starts, t0 - 25e6); i1 = bisect.bisect_left(starts, t1)
print(f"launch calls in window: {i1 - i0}")
# histogram of launch-call host times in 1 ms bins relative to device step start
from collections import Counter
h = Counter(int((s - t0) // 1e6) for s in starts[i0:i1])
print("launch calls per ms (bin = ms relative to device step start):", dict(sorted(h.items())))
