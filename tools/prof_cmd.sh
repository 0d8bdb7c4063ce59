# rocprofv3 kernel-trace stats of one command: bash tools/prof_cmd.sh <tag> <python args...>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o p -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("| % | calls | avg us | min us | max us | kernel |\n|---|---|---|---|---|---|")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:40]:
    print(f"| {100*float(r['TotalDurationNs'])/tot:.1f} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | `{r['Name'][:110]}` |")
PY
