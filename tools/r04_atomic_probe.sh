cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_atomic_probe.txt; : > $out
for mode in partial atomic; do
  rm -rf gpurun_out/kt_tn
  DIST_AMD_TN8P=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_tn -o p -- python3 tools/tn_atomic_probe.py $mode > gpurun_out/kt_tn.log 2>&1
  f=$(find gpurun_out/kt_tn -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$mode" >> $out <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tn" in r["Name"] and "at::" not in r["Name"]]
for r in rows:
    print(f"   {sys.argv[2]:8s} {r['Calls']:>4} calls  avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}  {r['Name'][:80]}")
PY
done
rm -rf gpurun_out/kt_tn
cat $out
