cd /root/repo; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_tnet_gpu.py -x -q -m gpu > gpurun_out/r05_run2_tests.log 2>&1; tail -3 gpurun_out/r05_run2_tests.log
bash tools/r05_step_ab.sh DIST_AMD_CONV9 3 2>&1 | tee gpurun_out/r05_step_ab_conv9.log
