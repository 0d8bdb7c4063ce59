cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
DIST_AMD_FAST_DBG=4 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-serial-ref --no-roofline 2>&1 | grep "fast8p" > gpurun_out/r05_fast8p_shapes.log
cat gpurun_out/r05_fast8p_shapes.log
unset DIST_AMD_LIB
bash tools/r05_spec_check.sh 2>&1 | grep -E "RESULT|s1b"
