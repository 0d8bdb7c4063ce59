# BASELINE config 4 (ViT-L/14 32+64f, b = 8): bench line + rocprofv3 kernel stats of the timed loop
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --config l14_32+64f --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_l14.json 2> gpurun_out/bench_l14.err
CMD="python3 bench.py --config l14_32+64f --batch 8 --steps 8 --warmup 2 --no-cpu-baseline --no-serial-ref --no-roofline"
rm -rf gpurun_out/prof_l14
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_l14 -o l14 -- $CMD > gpurun_out/prof_l14.log 2>&1
python tools/prof_summary.py gpurun_out/prof_l14/l14_results.db 10 40 > gpurun_out/prof_l14_summary.md 2>&1
rm -f gpurun_out/prof_l14/l14_results.db
tail -1 gpurun_out/bench_l14.json | cut -c1-400
