cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_f.json 2> gpurun_out/bench_f.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_f -o f -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_f.log 2>&1
python tools/prof_summary.py gpurun_out/prof_f/f_results.db 16 40 > gpurun_out/prof_f_summary.md 2>&1
