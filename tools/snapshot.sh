# round snapshot (usage: bash tools/snapshot.sh <tag>):
#   gpurun_out/bench_<tag>.json          the driver's command (default flags)
#   gpurun_out/bench_<tag>_serial.json   --no-pipeline
#   gpurun_out/prof_<tag>_summary.md     rocprofv3 --kernel-trace --stats of the TIMED LOOP ONLY (no CPU leg, no serial reference, no roofline / forward-only
#                                        extras): 5 warm-up + 20 timed steps = 25 steps (+ the pipeline prologue's one extra ViT pass), divided by 25
#   gpurun_out/prof_<tag>_stats.json     dominant kernel: in_situ_avg_us (that loop) and alone_avg_us (10 lone ViT passes, tools/vit_pass_alone.py);
#                                        cu_time_floor_ms: the packing bound of the step from a third trace with every kernel alone (DIST_AMD_SERIAL=3 --no-pipeline)
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-pipeline --no-cpu-baseline > gpurun_out/bench_${tag}_serial.json 2>> gpurun_out/bench_$tag.err
CMD="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serial-ref --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- $CMD > gpurun_out/prof_$tag.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_alone -o alone -- python3 tools/vit_pass_alone.py 10 > gpurun_out/prof_${tag}_alone.log 2>&1
# every kernel of the step ALONE on the GPU (one stream, serial order): the alone-times of the CU-time floor (10 + 3 steps)
DIST_AMD_SERIAL=3 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_serial -o serial -- python3 bench.py --steps 10 --warmup 3 --no-pipeline --no-cpu-baseline --no-serial-ref --no-roofline > gpurun_out/prof_${tag}_serial.log 2>&1
WALL=$(python -c "import json; print(json.load(open('gpurun_out/bench_$tag.json'))['ms_per_step'])")
python tools/prof_summary.py gpurun_out/prof_$tag/${tag}_results.db 25 45 --json gpurun_out/prof_${tag}_stats.json --command "rocprofv3 --kernel-trace --stats -- $CMD" \
    --alone-db gpurun_out/prof_${tag}_alone/alone_results.db --alone-passes 10 \
    --serial-db gpurun_out/prof_${tag}_serial/serial_results.db --serial-steps 13 --wall-ms $WALL > gpurun_out/prof_${tag}_summary.md 2>&1
python tools/timeline.py gpurun_out/prof_$tag/${tag}_results.db 5 11 > gpurun_out/prof_${tag}_timeline.txt 2>&1
rm -f gpurun_out/prof_$tag/${tag}_results.db gpurun_out/prof_${tag}_alone/alone_results.db gpurun_out/prof_${tag}_serial/serial_results.db
