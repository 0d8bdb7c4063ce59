# block caps of the two weight-gradient kernel families in the step (product selectors), alternating on one box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for rep in 1 2; do for cfg in "96 96" "80 96" "96 80" "112 96" "96 112" "128 128" "80 80"; do set -- $cfg
DIST_AMD_TN_BLOCKS=$1 DIST_AMD_TN8P_BLOCKS=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serial-ref --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tn=$1 tn8p=$2 ms/step', d['ms_per_step'])"; done; done 2>&1 | tee gpurun_out/r06_blocks.log
