"""Per-launch PMC counter value by (kernel, grid) from a rocprofv3 --pmc ... --output-format csv directory.
usage: python tools/pmc_csv.py <dir> <COUNTER> [scale]   (scale 2 for FETCH_SIZE on gfx950: 128-B requests tallied at 64 B; values are KB)"""
import csv, glob, os, sys
from collections import defaultdict
d, ctr = sys.argv[1], sys.argv[2]
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(list)
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != ctr: continue
            name = r["Kernel_Name"].split("(")[0][-40:]
            acc[(name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))].append(float(r["Counter_Value"]))
for (name, blocks), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) * scale < 2000: continue
    print(f"{name:42s} {blocks:6d} blk  {len(v):3d} launches  {sum(v)/len(v)*scale*1024/1e6:9.1f} MB per launch")
