"""Mean counter values per launch by (kernel, grid) for several counters of one rocprofv3 --pmc csv directory.
usage: python tools/pmc_generic.py <dir> <name filter>"""
import csv, glob, os, sys
from collections import defaultdict
d, flt = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if flt not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"][:48], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in acc.items():
    print(key)
    for c, v in sorted(cs.items()):
        print(f"     {c:28s} {sum(v)/len(v):16.0f}")
