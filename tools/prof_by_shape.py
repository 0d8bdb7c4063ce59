"""Summarise a rocprofv3 kernel-trace DB by (kernel, grid, block): per-step time of every distinct launch shape.
usage: python tools/prof_by_shape.py <results.db> <steps> [rows]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = list(cur.execute(
    "select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3 "
    "from kernels group by name, grid_x, workgroup_x order by 5 desc"))
tot = sum(r[4] for r in rows)
print(f"total kernel time {tot/1e3:.1f} ms over {steps} steps = {tot/1e3/steps:.2f} ms/step\n")
print("| % | calls/step | avg us | min us | ms/step | blocks | threads | kernel |\n|---|---|---|---|---|---|---|---|")
for r in rows[:top]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
    nm = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", nm)[:70]
    print(f"| {r[4]/tot*100:.1f} | {r[3]/steps:.1f} | {r[5]:.1f} | {r[6]:.1f} | {r[4]/1e3/steps:.2f} | {r[1]//max(r[2],1)} | {r[2]} | `{nm}` |")
