cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_e -o e -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_e.log 2>&1
python tools/prof_summary.py gpurun_out/prof_e/e_results.db 16 40 > gpurun_out/prof_e_summary.md 2>&1
