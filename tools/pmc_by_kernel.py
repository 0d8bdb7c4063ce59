"""FETCH_SIZE / WRITE_SIZE per launch by kernel name (rocprofv3 --pmc csv directories of the two passes).
FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md HBM section); units are KB.
usage: python tools/pmc_by_kernel.py <fetch dir> <write dir> <name filter> [out.json]"""
import csv, glob, json, os, re, sys
from collections import defaultdict
def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc
F, W = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
flt = sys.argv[3]
out = {}
print("| kernel | launches | read MB / launch (x2-corrected) | write MB / launch | total MB |\n|---|---|---|---|---|")
for k in sorted(set(F) | set(W)):
    if flt not in k: continue
    rd = sum(F.get(k, [0])) / max(len(F.get(k, [0])), 1) * 2 * 1024 / 1e6
    wr = sum(W.get(k, [0])) / max(len(W.get(k, [0])), 1) * 1024 / 1e6
    nm = re.sub(r"\(anonymous namespace\)::", "", k); nm = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", nm)[:90]
    print(f"| `{nm}` | {len(F.get(k, []))} | {rd:.1f} | {wr:.1f} | {rd + wr:.1f} |")
    out[nm] = {"read_mb": round(rd, 2), "write_mb": round(wr, 2), "launches": len(F.get(k, []))}
if len(sys.argv) > 4: json.dump(out, open(sys.argv[4], "w"), indent=1)
