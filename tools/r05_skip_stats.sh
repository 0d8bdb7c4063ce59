# how much of the step is the ViT chain's 24 ln_stats_from_partials launches (timing-only library, DIST_AMD_SKIP=128: the launches are skipped, statistics stale)
cd /root/repo; mkdir -p gpurun_out
. tools/measure_build.sh
for v in 0 128 0 128 0 128; do DIST_AMD_SKIP=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-serial-ref --no-roofline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip=$v ms/step', d['ms_per_step'])"; done 2>&1 | tee gpurun_out/r05_skip_stats.log
