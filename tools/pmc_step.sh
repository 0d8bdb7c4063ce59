# whole-step HBM traffic by kernel: two separate PMC passes over 3 steps of bench.py (--no-pipeline: exactly one ViT pass per step;
# the counter collection serialises the kernels anyway)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcstep_$c -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-pipeline > gpurun_out/pmcstep_$c.log 2>&1
  echo rc=$? >> gpurun_out/pmcstep_$c.log
done
find gpurun_out/pmcstep_FETCH_SIZE gpurun_out/pmcstep_WRITE_SIZE -name "*.csv" | head
python3 tools/pmc_step.py gpurun_out/pmcstep_FETCH_SIZE gpurun_out/pmcstep_WRITE_SIZE 3 gpurun_out/pmc_step_traffic.json
