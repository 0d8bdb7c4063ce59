cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_d.json 2> gpurun_out/bench_d.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_d -o d -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_d.log 2>&1
python tools/prof_summary.py gpurun_out/prof_d/d_results.db 16 40 > gpurun_out/prof_d_summary.md 2>&1
