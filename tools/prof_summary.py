"""Summarise a rocprofv3 results DB (kernel trace) into a markdown table.
usage: prof_summary.py <results.db> <steps> [rows] [--json out.json --alone-db alone.db --alone-passes N]
`steps` is the REAL number of steps the profiled command executed (warm-up + timed): every per-step column is a plain division by it."""
import json, re, sqlite3, sys
args = sys.argv[1:]
opt = {}
while "--json" in args or "--alone-db" in args or "--alone-passes" in args or "--command" in args:
    for k in ("--json", "--alone-db", "--alone-passes", "--command"):
        if k in args:
            i = args.index(k); opt[k] = args[i + 1]; del args[i:i + 2]
db = sqlite3.connect(args[0]); cur = db.cursor()
steps = int(args[1]) if len(args) > 1 else 1
nrows = int(args[2]) if len(args) > 2 else 30
Q = "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc"
rows = list(cur.execute(Q))
tot = sum(r[2] for r in rows)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:100]
print(f"total kernel time {tot/1e3:.1f} ms over {steps} steps = {tot/1e3/steps:.2f} ms/step (sum over streams: kernels of different streams overlap)\n")
print("| % | calls/step | avg us | min us | max us | ms/step | kernel |\n|---|---|---|---|---|---|---|")
for r in rows[:nrows]:
    print(f"| {r[2]/tot*100:.1f} | {r[1]/steps:.1f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {r[2]/1e3/steps:.2f} | `{short(r[0])}` |")
if "--json" in opt:
    dom = next(r for r in rows if "gemm_fast8p_kernel" in r[0] and "true" not in r[0].split("gemm_fast8p_kernel")[1][:8])
    out = {"dominant_kernel": dom[0], "command": opt.get("--command", ""), "steps_profiled": steps,
           "in_situ_launches": dom[1], "in_situ_launches_per_step": round(dom[1] / steps, 2), "in_situ_avg_us": round(dom[3], 1),
           "total_kernel_ms_per_step": round(tot / 1e3 / steps, 2)}
    fam = lambda key: sum(r[2] for r in rows if key(r[0])) / 1e3 / steps
    out["ms_per_step_in_situ"] = {"gemm_tn_family": round(fam(lambda n: "gemm_tn" in n or "tn_reduce" in n or "tn8p_reduce" in n), 3),
                                  "gemm_fast8p": round(fam(lambda n: "gemm_fast8p" in n), 3), "attn": round(fam(lambda n: "attn_kernel" in n), 3),
                                  "integ": round(fam(lambda n: "integ_" in n), 3), "tnet": round(fam(lambda n: "tnet_" in n), 3)}
    if "--alone-db" in opt:
        adb = sqlite3.connect(opt["--alone-db"])
        arows = list(adb.cursor().execute(Q))
        ad = next(r for r in arows if r[0] == dom[0])
        out.update({"alone_launches": ad[1], "alone_avg_us": round(ad[3], 1), "alone_passes": int(opt.get("--alone-passes", 0)),
                    "alone_command": "rocprofv3 --kernel-trace --stats -- python3 tools/vit_pass_alone.py " + opt.get("--alone-passes", "")})
    json.dump(out, open(opt["--json"], "w"), indent=1)
