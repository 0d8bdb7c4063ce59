# priority of the frozen ViT's stream in the pipelined step (timing-only library): 1 = lowest (shipped), 0 = default, 2 = highest; with the side streams at lowest / default
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
for rep in 1 2; do for cfg in "1 1" "0 1" "2 1" "0 0" "2 0"; do set -- $cfg
DIST_AMD_PF_PRIO=$1 DIST_AMD_SIDE_PRIO=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serial-ref --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pf_prio=$1 side_prio=$2 ms/step', d['ms_per_step'])"; done; done 2>&1 | tee gpurun_out/r05_pf_prio.log
