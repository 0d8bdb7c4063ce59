# the stem's weight gradient on conv_t_dw.hip in the step (timing-only library: DIST_AMD_CONVT5=0 -> the generic kernel), alternating on one box
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
for v in 1 0 1 0 1 0; do DIST_AMD_CONVT5=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serial-ref --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('convt5=$v ms/step', d['ms_per_step'])"; done 2>&1 | tee gpurun_out/r05_step_convt.log
