cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export DIST_AMD_SERIAL=3
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02s -o s -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipeline --no-serial-ref --no-roofline > gpurun_out/prof_r02s.log 2>&1
python tools/prof_by_shape.py gpurun_out/prof_r02s/s_results.db 8 60 > gpurun_out/r02_serial_by_shape2.md 2>&1
rm -f gpurun_out/prof_r02s/s_results.db
head -50 gpurun_out/r02_serial_by_shape2.md
