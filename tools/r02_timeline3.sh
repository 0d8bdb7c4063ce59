cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_t3 -o t3 -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-serial-ref --no-roofline > gpurun_out/prof_t3.log 2>&1
python tools/timeline3.py gpurun_out/prof_t3/t3_results.db 7 6.5 > gpurun_out/r02_t3_first_ms.txt 2>&1
rm -f gpurun_out/prof_t3/t3_results.db
grep -v "q4" gpurun_out/r02_t3_first_ms.txt | head -60
