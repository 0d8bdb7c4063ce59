# the weight-gradient kernel variants side by side (DIST_AMD_TN8P = 0 old kernel, 1 two 64-row buffers, 16 / 20 ring slots): correctness, cold timing, kernel trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_tn8p_modes.txt; : > $out
for m in 1 0; do
  echo "=== DIST_AMD_TN8P=$m" >> $out
  DIST_AMD_TN8P=$m timeout 300 python3 tools/bench_tn8p.py >> $out 2>&1
  for shape in "384 768" "384 480" "480 384"; do
    rm -rf gpurun_out/kt_tn
    DIST_AMD_TN8P=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_tn -o p -- python3 tools/pmc_tn_shape.py $shape > gpurun_out/kt_tn.log 2>&1
    f=$(find gpurun_out/kt_tn -name "*kernel_stats.csv" | head -1)
    python3 - "$f" "$shape" >> $out <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "tn" in r["Name"] and "at::" not in r["Name"]]
for r in rows:
    print(f"   trace {sys.argv[2]:8s} {r['Calls']:>4} calls  avg {float(r['AverageNs'])/1e3:7.1f} us  min {float(r['MinNs'])/1e3:7.1f}  {r['Name'][:80]}")
PY
  done
done
rm -rf gpurun_out/kt_tn
