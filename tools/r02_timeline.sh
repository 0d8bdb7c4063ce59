cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=${1:-t}
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-ref --no-roofline > gpurun_out/prof_$tag.log 2>&1
python tools/timeline.py gpurun_out/prof_$tag/${tag}_results.db 5 11 > gpurun_out/r02_${tag}_timeline.txt 2>&1
python tools/timeline2.py gpurun_out/prof_$tag/${tag}_results.db > gpurun_out/r02_${tag}_timeline2.txt 2>&1
python tools/prof_by_shape.py gpurun_out/prof_$tag/${tag}_results.db 14 50 > gpurun_out/r02_${tag}_by_shape.md 2>&1
rm -f gpurun_out/prof_$tag/${tag}_results.db
cat gpurun_out/r02_${tag}_timeline.txt | tail -40
