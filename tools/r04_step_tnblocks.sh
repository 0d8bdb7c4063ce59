# the step under the block caps of the two weight-gradient kernels (alternating on one box)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_step_tnblocks2.txt; : > $out
run() { r=$(env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"); echo "$* : $r" >> $out; }
for i in 1 2; do
  run DIST_AMD_TN_BLOCKS=128
  run DIST_AMD_TN_BLOCKS=96
  run DIST_AMD_TN_BLOCKS=64
  run DIST_AMD_TN_BLOCKS=128 DIST_AMD_TN8P_BLOCKS=96
  run DIST_AMD_TN_BLOCKS=128 DIST_AMD_TN8P_BLOCKS=64
  run DIST_AMD_TN_BLOCKS=96 DIST_AMD_TN8P_BLOCKS=96
done
cat $out
