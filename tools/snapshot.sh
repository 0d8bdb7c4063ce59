# round snapshot: bench line + rocprofv3 kernel stats + concurrency summary  (usage: bash tools/snapshot.sh <tag>)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-pipeline --no-cpu-baseline > gpurun_out/bench_${tag}_serial.json 2>> gpurun_out/bench_$tag.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-ref > gpurun_out/prof_$tag.log 2>&1
python tools/prof_summary.py gpurun_out/prof_$tag/${tag}_results.db 20 40 > gpurun_out/prof_${tag}_summary.md 2>&1
python tools/timeline.py gpurun_out/prof_$tag/${tag}_results.db 5 11 > gpurun_out/prof_${tag}_timeline.txt 2>&1
rm -f gpurun_out/prof_$tag/${tag}_results.db
