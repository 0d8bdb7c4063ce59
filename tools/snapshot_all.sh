# everything profiles/r0N_* is built from, in one call: bash tools/snapshot_all.sh <tag>
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
bash tools/snapshot.sh $tag
bash tools/pmc_step.sh > gpurun_out/pmc_step_traffic.md 2>&1
bash tools/pmc_pass.sh FETCH_SIZE; bash tools/pmc_pass.sh WRITE_SIZE
python3 tools/pmc_fast_json.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_fast_gemm.json > gpurun_out/pmc_fast_gemm.md 2>&1
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmcstep_FETCH_SIZE gpurun_out/pmcstep_WRITE_SIZE gpurun_out/prof_$tag gpurun_out/prof_${tag}_alone gpurun_out/prof_${tag}_serial
tail -1 gpurun_out/bench_$tag.json | cut -c1-300
