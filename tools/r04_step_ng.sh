# the step under forced column-tile group counts of the dominant GEMM (DIST_AMD_FAST_NG; 0 = the launcher's traffic model), alternating on one box
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_step_ng.txt; : > $out
run() { r=$(env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"); echo "$* : $r" >> $out; }
for i in 1 2; do
  for g in 0 1 2 3 4; do run DIST_AMD_FAST_NG=$g; done
done
cat $out
