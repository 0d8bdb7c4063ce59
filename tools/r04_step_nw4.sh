# the step with the two-blocks-per-CU 256x128 GEMM shape (DIST_AMD_FAST_NW=4) against the default two-group 256x256 kernel, alternating on one box
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_step_nw4.txt; : > $out
run() { r=$(env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"); echo "$* : $r" >> $out; }
for i in 1 2; do
  run DIST_AMD_FAST_NW=0
  run DIST_AMD_FAST_NW=4
done
cat $out
