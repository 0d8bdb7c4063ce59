cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --hip-trace --kernel-trace -d gpurun_out/prof_api -o api -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-ref --no-roofline > gpurun_out/prof_api.log 2>&1
python tools/api_vs_kernel.py gpurun_out/prof_api/api_results.db 7 > gpurun_out/r02_api_vs_kernel.txt 2>&1
rm -f gpurun_out/prof_api/api_results.db
cat gpurun_out/r02_api_vs_kernel.txt | cut -c1-250 | head -70
