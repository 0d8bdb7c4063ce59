"""Aggregate FETCH_SIZE / WRITE_SIZE counter CSVs (rocprofv3 --pmc, separate passes) per kernel name -> MB per step.
FETCH_SIZE is doubled (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md HBM section); units are KB."""
import csv, glob, os, re, sys
from collections import defaultdict
def load(d, counter):
    out = defaultdict(float); n = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter: continue
            out[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    return out, n
fd, wd, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
F, n = load(fd, "FETCH_SIZE"); W, _ = load(wd, "WRITE_SIZE")
def short(s):
    s = re.sub(r"\(anonymous namespace\)::", "", s); s = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", s); return s[:70]
rows = []
for k in set(F) | set(W):
    rd = F.get(k, 0) * 2 * 1024 / 1e6 / steps; wr = W.get(k, 0) * 1024 / 1e6 / steps
    rows.append((rd + wr, rd, wr, n.get(k, 0) / steps, short(k)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"total HBM-side traffic {tot/1e3:.2f} GB per step (reads x2-corrected {sum(r[1] for r in rows)/1e3:.2f} GB, writes {sum(r[2] for r in rows)/1e3:.2f} GB)\n")
print("| MB/step | read MB | write MB | launches/step | MB/launch | kernel |\n|---|---|---|---|---|---|")
for r in rows[:24]:
    print(f"| {r[0]:.0f} | {r[1]:.0f} | {r[2]:.0f} | {r[3]:.1f} | {r[0]/max(r[3],1e-9):.1f} | `{r[4]}` |")
if len(sys.argv) > 4:
    import json
    json.dump({"bytes_per_step": round(tot * 1e6), "read_bytes_per_step": round(sum(r[1] for r in rows) * 1e6),
               "write_bytes_per_step": round(sum(r[2] for r in rows) * 1e6), "steps": steps,
               "note": "rocprofv3 --pmc FETCH_SIZE (x2: gfx950 tallies 128-byte requests at 64 B) and --pmc WRITE_SIZE, separate passes, all kernels of python3 bench.py --steps 2 --warmup 1 --no-pipeline; includes Infinity-Cache hits"},
              open(sys.argv[4], "w"))
