# the train step under the weight-gradient kernel variants, alternating on ONE box
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_step_tn.txt; : > $out
run() { r=$(env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); echo "$* : $r" >> $out; }
for i in 1 2; do
  run DIST_AMD_TN8P=0
  run DIST_AMD_TN8P=1
  run DIST_AMD_TN8P=16
  run DIST_AMD_TN8P=20
  run DIST_AMD_TN8P_BLOCKS=128
  run DIST_AMD_TN8P=20 DIST_AMD_TN8P_BLOCKS=192
  run DIST_AMD_TN8P=1 DIST_AMD_TN8P_BLOCKS=128
done
cat $out
