# the two bench-size parity tests of round 5 + what they measured
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "b8_every_gradient or b32_gradient_norms" -s > gpurun_out/r05_parity.log 2>&1
tail -30 gpurun_out/r05_parity.log
cat gpurun_out/parity_gaps.json | head -40
