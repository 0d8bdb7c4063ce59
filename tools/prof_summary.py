"""Summarise a rocprofv3 results DB (kernel trace) into a markdown table."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e3:.1f} ms over {steps} steps = {tot/1e3/steps:.2f} ms/step\n")
print("| % | calls/step | avg us | min us | max us | ms/step | kernel |\n|---|---|---|---|---|---|---|")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
    nm = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", nm)[:100]
    print(f"| {r[2]/tot*100:.1f} | {r[1]/steps:.1f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {r[2]/1e3/steps:.2f} | `{nm}` |")
