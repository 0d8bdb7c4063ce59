# forward-only iteration (frozen ViT of batch n+1 beside the branch forward of batch n, inference mode): kernel statistics in situ and alone, CU-time floor
#  -> gpurun_out/r06_fwd_kernel_stats.{md,json}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/fwd_only.py --iters 30 2>&1 | grep fwd_only > gpurun_out/r06_fwd_only.txt
WALL=$(python -c "import re; print(re.search(r': ([0-9.]+) ms/iter', open('gpurun_out/r06_fwd_only.txt').read()).group(1))")
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fwd -o fwd -- python3 tools/fwd_only.py --iters 20 > gpurun_out/prof_fwd.log 2>&1
DIST_AMD_SERIAL=3 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_fwds -o fwds -- python3 tools/fwd_only.py --serial --iters 10 > gpurun_out/prof_fwds.log 2>&1
python tools/prof_summary.py gpurun_out/prof_fwd/fwd_results.db 25 40 --json gpurun_out/r06_fwd_kernel_stats.json --command "rocprofv3 --kernel-trace --stats -- python3 tools/fwd_only.py --iters 20" \
    --serial-db gpurun_out/prof_fwds/fwds_results.db --serial-steps 15 --wall-ms $WALL > gpurun_out/r06_fwd_kernel_stats.md 2>&1
rm -rf gpurun_out/prof_fwd gpurun_out/prof_fwds
head -8 gpurun_out/r06_fwd_kernel_stats.md
