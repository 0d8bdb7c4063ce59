# everything profiles/r06_* is built from, on ONE box: GPU test suite, the round snapshot + PMC passes, forward budget, configurations 3 / 4 / 5
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/gputest_r06.log; tail -2 gpurun_out/gputest_r06.log
bash tools/snapshot_all.sh r06
bash tools/r06_fwd_budget.sh
bash tools/r06_configs.sh
