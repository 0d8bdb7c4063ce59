# what a kernel's CU-time is worth in the step: timing-only library, DIST_AMD_SKIP 32 = no ViT attention (1.13 ms of CU-time), 64 = no c_fc GEMM (3.17 ms), 1 = no weight gradients
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
for v in 0 32 64 1 0 32 64 1; do DIST_AMD_SKIP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serial-ref --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip=$v ms/step', d['ms_per_step'])"; done 2>&1 | tee gpurun_out/r05_skip_calib.log
