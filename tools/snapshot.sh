# round snapshot (usage: bash tools/snapshot.sh <tag> [bench.py arguments of another configuration, e.g. --config l14_32+64f --batch 8]):
#   gpurun_out/bench_<tag>.json          the driver's command (default flags + the extra arguments)
#   gpurun_out/bench_<tag>_serial.json   --no-pipeline
#   gpurun_out/prof_<tag>_summary.md     rocprofv3 --kernel-trace --stats of the TIMED LOOP ONLY (no CPU leg, no serial reference, no roofline / forward-only
#                                        extras): W warm-up + K timed steps (+ the pipeline prologue's one extra ViT pass), divided by W + K
#   gpurun_out/prof_<tag>_stats.json     dominant kernel: in_situ_avg_us (that loop) and alone_avg_us (10 lone ViT passes, tools/vit_pass_alone.py; headline configuration only);
#                                        cu_time_floor_ms: the packing bound of the step from a third trace with every kernel alone (DIST_AMD_SERIAL=3 --no-pipeline)
tag=${1:-x}; shift
EXTRA="$*"
K=${SNAP_STEPS:-20}; W=${SNAP_WARMUP:-5}; KS=${SNAP_SERIAL_STEPS:-10}; WS=3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps $K --warmup $W $EXTRA > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
timeout 400 python bench.py --steps $K --warmup $W --no-pipeline --no-cpu-baseline $EXTRA > gpurun_out/bench_${tag}_serial.json 2>> gpurun_out/bench_$tag.err
CMD="python3 bench.py --steps $K --warmup $W --no-cpu-baseline --no-serial-ref --no-roofline $EXTRA"
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- $CMD > gpurun_out/prof_$tag.log 2>&1
ALONE=""
if [ -z "$EXTRA" ]; then
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_alone -o alone -- python3 tools/vit_pass_alone.py 10 > gpurun_out/prof_${tag}_alone.log 2>&1
  ALONE="--alone-db gpurun_out/prof_${tag}_alone/alone_results.db --alone-passes 10"
fi
# every kernel of the step ALONE on the GPU (one stream, serial order): the alone-times of the CU-time floor
DIST_AMD_SERIAL=3 timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_serial -o serial -- python3 bench.py --steps $KS --warmup $WS --no-pipeline --no-cpu-baseline --no-serial-ref --no-roofline $EXTRA > gpurun_out/prof_${tag}_serial.log 2>&1
WALL=$(python -c "import json; print(json.loads(open('gpurun_out/bench_$tag.json').read().strip().splitlines()[-1])['ms_per_step'])")
python tools/prof_summary.py gpurun_out/prof_$tag/${tag}_results.db $((K + W)) 45 --json gpurun_out/prof_${tag}_stats.json --command "rocprofv3 --kernel-trace --stats -- $CMD" \
    $ALONE --serial-db gpurun_out/prof_${tag}_serial/serial_results.db --serial-steps $((KS + WS)) --wall-ms $WALL > gpurun_out/prof_${tag}_summary.md 2>&1
python tools/timeline.py gpurun_out/prof_$tag/${tag}_results.db 5 11 > gpurun_out/prof_${tag}_timeline.txt 2>&1
rm -rf gpurun_out/prof_$tag gpurun_out/prof_${tag}_alone gpurun_out/prof_${tag}_serial
