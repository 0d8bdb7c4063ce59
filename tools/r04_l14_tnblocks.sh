# ViT-L/14 32+64f (config 4, b = 8): the step under the block caps of the two weight-gradient kernels (alternating on one box)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_l14_tnblocks.txt; : > $out
run() { r=$(env "$@" timeout 300 python bench.py --config l14_32+64f --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"); echo "$* : $r" >> $out; }
for i in 1 2; do
  run DIST_AMD_TN_BLOCKS=96 DIST_AMD_TN8P_BLOCKS=96
  run DIST_AMD_TN_BLOCKS=128 DIST_AMD_TN8P_BLOCKS=128
  run DIST_AMD_TN_BLOCKS=64 DIST_AMD_TN8P_BLOCKS=64
  run DIST_AMD_TN_BLOCKS=192 DIST_AMD_TN8P_BLOCKS=192
  run DIST_AMD_TN_BLOCKS=96 DIST_AMD_TN8P_BLOCKS=160
  run DIST_AMD_TN_BLOCKS=160 DIST_AMD_TN8P_BLOCKS=96
done
cat $out
