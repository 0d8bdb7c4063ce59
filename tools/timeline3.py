"""List every kernel of ONE step's first milliseconds from a rocprofv3 kernel-trace DB: start, duration, queue, grid, name.
usage: python tools/timeline3.py <results.db> [step] [ms]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = list(cur.execute("select name, start, end, queue_id, grid_x, workgroup_x from kernels order by start"))
ad = [r for r in rows if "adamw" in r[0]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ms = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
t0 = ad[k][2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:46]
for r in rows:
    if r[1] >= ad[k][1] - 0.2e6 and r[1] < t0 + ms * 1e6:
        print(f"{(r[1]-t0)/1e3:9.1f} us  +{(r[2]-r[1])/1e3:7.1f}  q{r[3]}  {r[4]//max(r[5],1):6d} blk  {short(r[0])}")
