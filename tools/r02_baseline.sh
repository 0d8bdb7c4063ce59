# round-2 baseline: phase times, bench line, serial-order kernel trace by launch shape
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/phase_times.py > gpurun_out/r02_phase_times.txt 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_base_bench.json 2> gpurun_out/r02_base_bench.err
export DIST_AMD_SERIAL=3
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02s -o s -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipeline --no-serial-ref > gpurun_out/prof_r02s.log 2>&1
python tools/prof_by_shape.py gpurun_out/prof_r02s/s_results.db 8 80 > gpurun_out/r02_serial_by_shape.md 2>&1
python tools/prof_summary.py gpurun_out/prof_r02s/s_results.db 8 40 > gpurun_out/r02_serial_summary.md 2>&1
rm -f gpurun_out/prof_r02s/s_results.db
tail -3 gpurun_out/r02_phase_times.txt; cat gpurun_out/r02_base_bench.json | cut -c1-400
