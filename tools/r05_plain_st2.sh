# streaming hint off for ONE of the ViT's GEMM outputs (does its consumer find it in the Infinity Cache?): the step, timing-only library
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
for rep in 1 2; do for v in 0 2 3 4; do DIST_AMD_FAST_PLAIN_ST=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serial-ref --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain_st=$v ms/step', d['ms_per_step'])"; done; done 2>&1 | tee gpurun_out/r05_plain_st2.log
