#!/bin/bash
# timing-only ablation of the fused IntegrationNetwork forward (results are wrong with a knob set): tools/integ_ablate.sh [BM]
mkdir -p gpurun_out
export DIST_AMD_BUILD_DEFS="-DDIST_INTEG_ABLATE"
. tools/measure_build.sh      # the debug switches only exist in the timing-only library
BM=${1:-128}
for d in 0 1 2 4 8 16 32 3 19 23 63 59; do
  echo "dbg=$d: $(DIST_AMD_INTEG_BM=$BM DIST_AMD_INTEG_DBG=$d python tools/bench_integ.py 2>/dev/null | tr '\n' ' ')"
done
