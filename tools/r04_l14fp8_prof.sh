# BASELINE config 5 (ViT-L/14 32+64f, e4m3 spatial branch, b = 16): rocprofv3 kernel stats of the timed loop
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --config l14_32+64f --batch 16 --vit-fp8 31 --steps 6 --warmup 2 --no-cpu-baseline --no-serial-ref --no-roofline"
rm -rf gpurun_out/prof_l14f8
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_l14f8 -o l14 -- $CMD > gpurun_out/prof_l14f8.log 2>&1
python tools/prof_summary.py gpurun_out/prof_l14f8/l14_results.db 8 30 > gpurun_out/prof_l14f8_summary.md 2>&1
rm -rf gpurun_out/prof_l14f8
head -24 gpurun_out/prof_l14f8_summary.md
