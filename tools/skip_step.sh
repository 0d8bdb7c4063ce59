#!/bin/bash
# sensitivity of the pipelined step to whole kernel families (DIST_AMD_SKIP bits, results wrong): how much step time each family holds
mkdir -p gpurun_out; . tools/measure_build.sh      # DIST_AMD_SKIP only exists in the timing-only library
for v in ${@:-0 1 16 2 4 8 32 64 0}; do
  r=$(DIST_AMD_SKIP=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-serial-ref 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "DIST_AMD_SKIP=$v: $r ms"
done
