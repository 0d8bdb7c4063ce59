cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
DIST_AMD_FAST_DBG=4 python bench.py --config l14_32+64f --batch 16 --vit-fp8 31 --steps 3 --warmup 2 --no-cpu-baseline --no-serial-ref --no-roofline 2>&1 | grep "fast8p" > gpurun_out/r05_fast8p_shapes_fp8.log
cat gpurun_out/r05_fast8p_shapes_fp8.log
