"""Summarise a rocprofv3 results DB (kernel trace) into a markdown table.
usage: prof_summary.py <results.db> <steps> [rows] [--json out.json --alone-db alone.db --alone-passes N --serial-db serial.db --serial-steps N --wall-ms T]
`steps` is the REAL number of steps the profiled command executed (warm-up + timed): every per-step column is a plain division by it.

--serial-db: the kernel trace of the SAME step with every kernel alone on the GPU (DIST_AMD_SERIAL=3 python bench.py --no-pipeline ...: one stream).
With it the table gains, per kernel, its alone-time, the stretch in situ / alone, its blocks per launch and its share of the step's
CU-TIME FLOOR = sum over kernels of (alone us x launches per step x min(1, blocks / 256)): the time the step would take if the 256 CUs were
packed perfectly with the kernels as they are (a kernel of >= 256 blocks holds the whole chip for its alone-time, a 96-block kernel 96 / 256 of
it) - next to the measured wall time (--wall-ms) it says how much of the step is schedule and how much is the kernels themselves."""
import json, re, sqlite3, sys
args = sys.argv[1:]
opt = {}
OPTS = ("--json", "--alone-db", "--alone-passes", "--command", "--serial-db", "--serial-steps", "--wall-ms")
while any(k in args for k in OPTS):
    for k in OPTS:
        if k in args:
            i = args.index(k); opt[k] = args[i + 1]; del args[i:i + 2]
db = sqlite3.connect(args[0]); cur = db.cursor()
steps = int(args[1]) if len(args) > 1 else 1
nrows = int(args[2]) if len(args) > 2 else 30
Q = ("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
     "avg((grid_x*1.0/workgroup_x)*(grid_y*1.0/workgroup_y)*(grid_z*1.0/workgroup_z)) from kernels group by name order by 3 desc")
rows = list(cur.execute(Q))
tot = sum(r[2] for r in rows)
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:100]
print(f"total kernel time {tot/1e3:.1f} ms over {steps} steps = {tot/1e3/steps:.2f} ms/step (sum over streams: kernels of different streams overlap)\n")
serial = None
if "--serial-db" in opt:
    sdb = sqlite3.connect(opt["--serial-db"])
    ssteps = int(opt.get("--serial-steps", steps))
    serial = {r[0]: r for r in sdb.cursor().execute(Q)}
    floor_rows, floor_ms, alone_ms = [], 0.0, 0.0
    for r in rows:
        a = serial.get(r[0])
        if a is None:
            continue
        share = min(1.0, r[6] / 256.0)
        cu_ms = a[3] * (r[1] / steps) * share / 1e3            # alone us x launches per step x CU share
        floor_ms += cu_ms; alone_ms += a[3] * (r[1] / steps) / 1e3
        floor_rows.append((r, a, share, cu_ms))
    wall = float(opt["--wall-ms"]) if "--wall-ms" in opt else None
    print(f"CU-time floor of the step (sum of alone us x launches/step x min(1, blocks/256)): {floor_ms:.2f} ms; sum of alone-times {alone_ms:.2f} ms/step; "
          f"sum of in-situ times {tot/1e3/steps:.2f} ms/step" + (f"; measured wall {wall:.2f} ms/step = {wall/floor_ms:.2f} x the floor" if wall else "") + "\n")
    print("| % | calls/step | in-situ us | alone us | stretch | blocks | CU share | CU-time ms/step | in-situ ms/step | kernel |\n|---|---|---|---|---|---|---|---|---|---|")
    for r, a, share, cu_ms in floor_rows[:nrows]:
        print(f"| {r[2]/tot*100:.1f} | {r[1]/steps:.1f} | {r[3]:.1f} | {a[3]:.1f} | {r[3]/a[3]:.2f} | {r[6]:.0f} | {share:.2f} | {cu_ms:.3f} | {r[2]/1e3/steps:.2f} | `{short(r[0])}` |")
else:
    print("| % | calls/step | avg us | min us | max us | ms/step | kernel |\n|---|---|---|---|---|---|---|")
    for r in rows[:nrows]:
        print(f"| {r[2]/tot*100:.1f} | {r[1]/steps:.1f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {r[2]/1e3/steps:.2f} | `{short(r[0])}` |")
if "--json" in opt:
    # the dominant kernel = gemm_fast8p_kernel<false, *>: the bf16 two-group GEMM with its epilogue instantiations (one loop, compile-time epilogues) taken together
    is_dom = lambda n: "gemm_fast8p_kernel" in n and "true" not in n.split("gemm_fast8p_kernel")[1][:8]
    def agg(rs, name="gemm_fast8p_kernel<false, *> (the bf16 two-group GEMM, all epilogue instantiations)"):
        rs = [r for r in rs if is_dom(r[0])]
        calls, total = sum(r[1] for r in rs), sum(r[2] for r in rs)
        return (name, calls, total, total / max(calls, 1), min(r[4] for r in rs), max(r[5] for r in rs), sum(r[6] * r[1] for r in rs) / max(calls, 1))
    dom = agg(rows)
    out = {"dominant_kernel": dom[0], "command": opt.get("--command", ""), "steps_profiled": steps,
           "in_situ_launches": dom[1], "in_situ_launches_per_step": round(dom[1] / steps, 2), "in_situ_avg_us": round(dom[3], 1),
           "total_kernel_ms_per_step": round(tot / 1e3 / steps, 2)}
    fam = lambda key: sum(r[2] for r in rows if key(r[0])) / 1e3 / steps
    out["ms_per_step_in_situ"] = {"gemm_tn_family": round(fam(lambda n: "gemm_tn" in n or "tn_reduce" in n or "tn8p_reduce" in n or "conv3x3_dw" in n), 3),
                                  "gemm_fast8p": round(fam(lambda n: "gemm_fast8p" in n), 3), "attn": round(fam(lambda n: "attn_kernel" in n), 3),
                                  "integ": round(fam(lambda n: "integ_" in n), 3), "tnet": round(fam(lambda n: "tnet_" in n), 3)}
    if serial is not None:
        out["cu_time_floor_ms"] = round(floor_ms, 3)
        out["cu_time_floor_note"] = ("sum over kernels of (alone us x launches per step x min(1, blocks / 256)); alone us from the single-stream trace "
                                     "(DIST_AMD_SERIAL=3, --no-pipeline), launches and blocks from the timed loop's trace")
        out["sum_alone_ms_per_step"] = round(alone_ms, 3)
        out["wall_ms_per_step"] = wall
        out["stretch"] = {short(r[0])[:60]: round(r[3] / a[3], 3) for r, a, _, _ in floor_rows[:16]}
        if any(is_dom(n) for n in serial):
            out["dominant_serial_avg_us"] = round(agg(serial.values())[3], 1)
        out["dominant_instantiations"] = {short(r[0])[:70]: {"launches_per_step": round(r[1] / steps, 2), "in_situ_avg_us": round(r[3], 1),
                                                             "alone_avg_us": round(serial[r[0]][3], 1) if r[0] in serial else None} for r in rows if is_dom(r[0])}
    if "--alone-db" in opt:
        adb = sqlite3.connect(opt["--alone-db"])
        arows = list(adb.cursor().execute(Q))
        ad = agg(arows)
        out.update({"alone_launches": ad[1], "alone_avg_us": round(ad[3], 1), "alone_passes": int(opt.get("--alone-passes", 0)),
                    "alone_command": "rocprofv3 --kernel-trace --stats -- python3 tools/vit_pass_alone.py " + opt.get("--alone-passes", "")})
    json.dump(out, open(opt["--json"], "w"), indent=1)
