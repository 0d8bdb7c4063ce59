"""profiles/rNN_pmc_fast_gemm.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_fast.py.
usage: python tools/pmc_fast_json.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, os, sys
from collections import defaultdict
def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and "gemm_fast" in r["Kernel_Name"]:
                acc[int(r["Grid_Size"]) // int(r["Workgroup_Size"])].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return acc
F, W = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
M = 50432
shapes = [("qkv", 2304, 768, 1773), ("out", 768, 768, 591), ("fc", 3072, 768, 2364), ("proj", 768, 3072, 591)]
per = []
for tag, N, K, blocks in shapes:
    f = sorted(F[blocks]); w = sorted(W[blocks])
    if blocks == 591:          # out (K = 768) launches come before proj (K = 3072) in tools/pmc_fast.py
        f = f[:3] if tag == "out" else f[3:]; w = w[:3] if tag == "out" else w[3:]
    fb = sum(v for _, v in f) / len(f) * 2 * 1024; wb = sum(v for _, v in w) / len(w) * 1024
    per.append({"shape": tag, "M": M, "N": N, "K": K, "fetch_bytes": round(fb), "write_bytes": round(wb),
                "algorithmic_read_bytes": (M * K + N * K) * 2 + N * 4, "algorithmic_write_bytes": M * N * 2})
out = {"kernel": "gemm_fast8p_kernel (256x256x64, two wave groups)",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/pmc_fast.py",
       "corrections": "FETCH_SIZE reported in KB, doubled (gfx950 counts 128-B requests at 64 B); WRITE_SIZE in KB, uncorrected",
       "per_shape": per,
       "traffic_bytes_per_launch_avg": round(sum(p["fetch_bytes"] + p["write_bytes"] for p in per) / len(per))}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for p in per:
    print(f"{p['shape']:5s} fetch {p['fetch_bytes']/1e6:7.1f} MB (algorithmic {p['algorithmic_read_bytes']/1e6:6.1f})  write {p['write_bytes']/1e6:7.1f} MB (algorithmic {p['algorithmic_write_bytes']/1e6:6.1f})")
print("mean bytes per launch", out["traffic_bytes_per_launch_avg"] / 1e6, "MB")
