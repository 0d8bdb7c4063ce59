# round 6: the 4-wave / 64-row IntegrationNetwork forward against the 8-wave / 128-row one: parity, time per launch, step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{


for rep in 1 2; do
  DIST_AMD_INTEG_W4=0 python tools/bench_integ4.py
  python tools/bench_integ4.py
done
bash tools/ab_step.sh DIST_AMD_INTEG_W4 0 1 3
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r06_integ4.log
